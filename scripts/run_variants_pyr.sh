#!/bin/bash
# isolated pyramid timing (scripts/pyr_time.py) for every library build under ov2slam_amd/lib/variants/ plus the default
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" $R/ov2slam_amd/lib/variants/*.so; do
  echo "== ${lib:-default}"
  OV2SLAM_HIP_LIB=$lib timeout -k 10 120 python3 $R/scripts/pyr_time.py 2>&1 | grep "B=64"
done
