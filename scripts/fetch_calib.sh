#!/bin/bash
# FETCH_SIZE of a kernel that reads a known byte count with the KLT staging pattern -> gpurun_out/fetch_calib.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_fetch_calib
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python3 $R/scripts/fetch_calib.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
F=$(find $O -name "*counter_collection.csv" | head -1)
python3 - "$F" "$O/run.log" <<'PY' | tee $R/gpurun_out/fetch_calib.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dbg_rowload16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
nbytes = int([l for l in open(sys.argv[2]) if l.startswith("bytes_per_launch")][0].split()[1])
vals = [float(r["Counter_Value"]) for r in rows]
kib = sum(vals) / len(vals)
print(f"dbg_rowload16_kernel: {len(vals)} launches, bytes read per launch (known) {nbytes}, FETCH_SIZE {kib:.0f} KiB = {kib * 1024:.0f} B")
print(f"FETCH_SIZE x 1024 / bytes = {kib * 1024 / nbytes:.4f}   (0.5 => the guide's x2 correction applies to this pattern; 1.0 => raw bytes)")
PY
find $O -name "*agent_info*" -delete
