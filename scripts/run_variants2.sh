#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" $R/ov2slam_amd/lib/variants/*.so; do
  echo "== ${lib:-default}"
  for B in 16 64; do OV2SLAM_HIP_LIB=$lib timeout -k 10 200 python3 $R/scripts/klt_time.py $B | grep stage1 || exit 1; done
done
