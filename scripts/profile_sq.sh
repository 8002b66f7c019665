#!/bin/bash
# SQ issue/stall counters of the hot kernels (one rocprofv3 --pmc pass, 8 SQ slots) for the same bench command.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sq}
shift
ARGS=${@:---steps 30 --warmup 6 --no-cpu-baseline --no-roofline --no-ba}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob("$O/sq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
print("kernel launches waves/launch wave_cycles active_any active_valu wait_any wait_inst valu_insts/wave")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:12]:
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    print(f"{k:24s} {cnt[k]:5d} {v['SQ_WAVES']/max(cnt[k],1):9.0f} {wc/max(cnt[k],1):12.0f} {v['SQ_ACTIVE_INST_ANY']/wc:6.2f} {v['SQ_ACTIVE_INST_VALU']/wc:6.2f} {v['SQ_WAIT_ANY']/wc:6.2f} {v['SQ_WAIT_INST_ANY']/wc:6.2f} {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):9.0f}")
PY
find $O -name "*agent_info*" -delete
