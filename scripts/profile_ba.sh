#!/bin/bash
# kernel-trace stats of the stand-alone batched local BA (scripts/ba_pipeline_time.py B: distinct windows, set-up + solve + update; BA_SCRIPT=ba_batch_time.py for the replicated solve-only form); output gpurun_out/prof_ba_<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ba}
B=${2:-64}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_ba_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/scripts/${BA_SCRIPT:-ba_pipeline_time.py} $B > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
tail -3 $O/kt.log
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.search(r"rocprim::\w+::detail::wrapped_(\w+?)_config", n)
    if m: return "rocprim_" + m.group(1)
    h = n.split("(", 1)[0]
    while True:
        t = re.sub(r"<[^<>]*>", "", h)
        if t == h: break
        h = t
    m = re.search(r"(\w+)\s*$", h)
    return m.group(1) if m else n
agg = {}
for r in rows:
    k = short(r["Name"]); a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':34s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k:34s} {c:7d} {t/1e6:10.2f} {t/c/1e3:10.1f} {100*t/tot:6.1f}")
PY
find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -delete
