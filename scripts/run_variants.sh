#!/bin/bash
# bench every library build under ov2slam_amd/lib/variants/ (kernel experiments) plus the default one
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${@:---steps 300 --seqs 64 --no-cpu-baseline --no-ba}
for lib in "" $R/ov2slam_amd/lib/variants/*.so; do
  echo "== ${lib:-default}"
  OV2SLAM_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py $ARGS > /tmp/b.json 2> /tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
  python3 $R/scripts/show_bench.py /tmp/b.json | head -3
done
