#!/bin/bash
# where do the KLT waves wait?  L2 request latency, TLB, LDS / vector-memory / scalar activity, instruction fetch
# (scripts/klt_time.py 64, nothing else on the GPU; one rocprofv3 --pmc pass per counter set)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_stalls
rm -rf $O && mkdir -p $O
i=0
for SET in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $O/set$i -- python3 $R/scripts/klt_time.py 64 > $O/set$i.log 2>&1 || { echo "set failed: $SET"; tail -3 $O/set$i.log; continue; }
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$O/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in ("klt_stage1_kernel",):
    if k in acc:
        for c, v in sorted(acc[k].items()):
            print(f"{k} {c:40s} {v / max(cnt[k][c], 1):16.1f}")
PY
find $O -name "*agent_info*" -delete
