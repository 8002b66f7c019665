#!/bin/bash
# isolated KLT timing (scripts/klt_time.py) for every library build under ov2slam_amd/lib/variants/ plus the default
R=${GRAFT_REPO_ROOT:-/root/repo}
SEQS=${@:-64 16}
for lib in "" $R/ov2slam_amd/lib/variants/*.so; do
  for S in $SEQS; do
    echo "== ${lib:-default} seqs=$S: $(OV2SLAM_HIP_LIB=$lib timeout -k 10 120 python3 $R/scripts/klt_time.py $S 2>&1 | grep klt_stage1)"
  done
done
