#!/bin/bash
# bench every library build under ov2slam_amd/lib/variants/ (kernel experiments) plus the default one
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" $R/ov2slam_amd/lib/variants/*.so; do
  for S in 16 64; do
    echo "== ${lib:-default} seqs=$S"
    OV2SLAM_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 300 --seqs $S --no-cpu-baseline --no-ba --no-roofline > /tmp/b.json 2> /tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
    python3 $R/scripts/show_bench.py /tmp/b.json | head -1
  done
done
