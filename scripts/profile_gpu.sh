#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and, in SEPARATE passes, the HBM PMC counters for the
# same bench command.  Outputs under gpurun_out/prof_*; scripts/summarize_prof.py turns them into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
shift
ARGS=${@:---steps 60 --warmup 6 --no-cpu-baseline --no-roofline}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $ARGS > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
# summarised HERE: the per-dispatch counter CSVs are tens of MB and gpurun merges at most 64 MiB back
mkdir -p $R/gpurun_out/prof_summary_$TAG
python3 $R/scripts/summarize_prof.py $TAG $R/gpurun_out/prof_summary_$TAG > $R/gpurun_out/prof_summary_$TAG/summary.txt 2>&1
find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
