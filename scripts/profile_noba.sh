#!/bin/bash
# kernel-trace stats of the front-end alone (bench.py --no-ba): launch durations without the local-BA worker's batches
# sharing the device -- the figure bench.py's roofline block measures with hipEvents after the worker has stopped.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_noba}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ba --no-euroc-like --no-hard-stream > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
tail -c 400 $O/kt.log
find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -delete
