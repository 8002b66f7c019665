#!/bin/bash
# SQ issue / stall / LDS counters of the batched local BA alone (scripts/ba_pipeline_time.py 64), two rocprofv3 --pmc passes
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_ba_sq
rm -rf $O && mkdir -p $O
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $O/set$i -- python3 $R/scripts/${BA_SCRIPT:-ba_pipeline_time.py} 64 3 > $O/set$i.log 2>&1 || { echo "set failed: $SET"; tail -3 $O/set$i.log; continue; }
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$O/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    v = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k}: waves {v.get('SQ_WAVES', 0):.0f}  per-launch wave_cycles {wc:.3g}  busy_cycles {v.get('SQ_BUSY_CYCLES', 0):.3g}")
    for c in sorted(v):
        if c not in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            print(f"    {c:28s} {v[c]:14.4g}   /wave_cycles {v[c] / wc:6.3f}   /wave {v[c] / max(v.get('SQ_WAVES', 1), 1):10.1f}")
PY
find $O -name "*agent_info*" -delete
