#!/bin/bash
# MFMA evidence north_star asks for.  The matrix-core sites of the path are the Cholesky of the reduced camera system:
# ba_chol_mfma_kernel (one workgroup per window, panel update on v_mfma_f64_16x16x4_f64; m < 400) and the trailing update of
# the multi-workgroup form (ba_chol_syrk_kernel; m >= 400 -- the config-4 window, 100 KF / 20 k landmarks).
# One rocprofv3 --pmc pass per counter set (program directly after --) of the stand-alone batched solve.
# usage: profile_mfma.sh                         config-4 windows, 8 per batch  -> gpurun_out/prof_mfma
#        MFMA_KFS=50 MFMA_LMS=10000 MFMA_B=64 MFMA_TAG=bench profile_mfma.sh   bench-size windows -> gpurun_out/prof_mfma_bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_mfma${MFMA_TAG:+_$MFMA_TAG}
NB=${MFMA_B:-8}
rm -rf $O && mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i "mfma" | sed 's/^[ \t]*//' | sort -u | head -40 > $O/mfma_counters_available.txt
export BA_KFS=${MFMA_KFS:-100} BA_LMS=${MFMA_LMS:-20000} BA_DEV=1
for SET in "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $SET --output-format csv -d $O/set$i -- python3 $R/scripts/ba_batch_time.py $NB > $O/set$i.log 2>&1 || { echo "set failed: $SET"; tail -3 $O/set$i.log; }
done
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/scripts/ba_batch_time.py $NB > $O/kt.log 2>&1 || tail -3 $O/kt.log
tail -2 $O/kt.log
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$O/set*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
dur = {}
for fn in glob.glob("$O/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"<[^<>]*>", "", r["Name"].replace("(anonymous namespace)::", "").split("(")[0]).split()[-1]
        d = dur.setdefault(k, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
print("kernels with a non-zero MFMA counter (sums over the run; launches; total us):")
for k in sorted(acc):
    v = acc[k]
    if not any(v.get(c, 0) > 0 for c in v if "MFMA" in c):
        continue
    print(f"  {k}: " + ", ".join(f"{c}={v[c]:.4g}" for c in sorted(v)) + f"; launches {max(cnt[k].values())}; kernel-trace {dur.get(k, [0, 0])[0]} calls {dur.get(k, [0, 0])[1] / 1e3:.1f} us")
    n = v.get("SQ_INSTS_VALU_MFMA_F64", 0) or v.get("SQ_INSTS_MFMA", 0)
    if n and k in dur and dur[k][1] > 0:
        # counters were summed over the launches of the pmc run, durations over the launches of the kernel-trace run (same program)
        flops = n * 16 * 16 * 4 * 2
        print(f"    v_mfma_f64_16x16x4: {n:.4g} wave-instructions x 2048 flop = {flops / 1e9:.3f} GFLOP in {dur[k][1] / 1e3:.1f} us of kernel time = "
              f"{flops / dur[k][1]:.2f} GFLOP/s... = {flops / dur[k][1] / 1e3:.4f} TFLOP/s of 78.6 TFLOP/s f64 matrix peak = {100 * flops / dur[k][1] / 1e3 / 78.6:.3f} %")
tot = sum(d[1] for d in dur.values())
print(f"all kernels of the run: {tot / 1e3:.0f} us; ba_chol_syrk_kernel share {100 * dur.get('ba_chol_syrk_kernel', [0, 0])[1] / max(tot, 1):.1f} %, "
      f"ba_chol_mfma_kernel share {100 * dur.get('ba_chol_mfma_kernel', [0, 0])[1] / max(tot, 1):.1f} %")
PY
find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -delete
