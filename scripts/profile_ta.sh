#!/bin/bash
# texture-addresser / vector-L1 counters of the KLT kernels (scripts/klt_time.py, nothing else on the GPU): is the
# kernel bound by the rate of vector memory instructions (TA busy) rather than by VALU issue or HBM?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_ta
rm -rf $O && mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|SQ_INSTS_VMEM[A-Z_]*\|SQ_INSTS_LDS\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_ACTIVE_INST_VMEM\|SQ_ACTIVE_INST_LDS" $O/avail.txt | sort -u > $O/names.txt
wc -l $O/names.txt
for SET in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $O/$tag -- python3 $R/scripts/klt_time.py 64 > $O/$tag.log 2>&1 || { echo "set failed: $SET"; tail -3 $O/$tag.log; continue; }
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
