#!/bin/bash
# texture-addresser / vector-L1 counters of the KLT kernels (scripts/klt_time.py, nothing else on the GPU): is the
# kernel bound by the rate of vector memory instructions (TA busy) rather than by VALU issue or HBM?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_ta
rm -rf $O && mkdir -p $O
# (a set with a counter this build does not know aborts rocprofv3 and then hangs: only sets that were seen to work)
for SET in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $O/$tag -- python3 $R/scripts/klt_time.py 64 > $O/$tag.log 2>&1 || { echo "set failed: $SET"; tail -3 $O/$tag.log; continue; }
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$O/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in ("klt_stage1_kernel", "level_kernel", "clahe_lut_kernel", "level0_clahe_tiled_kernel"):
    if k in acc:
        print(k, {c: round(v / max(cnt[k][c], 1), 1) for c, v in sorted(acc[k].items())})
PY
find $O -name "*agent_info*" -delete
