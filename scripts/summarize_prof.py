#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/profile_gpu.sh (gpurun_out/prof_<tag>/) into the committed summaries:
  profiles/<tag>_kernel_stats.csv   -- `rocprofv3 --kernel-trace --stats` per-kernel table (names shortened)
  profiles/<tag>_pmc.csv            -- per kernel: mean FETCH_SIZE / WRITE_SIZE (KiB, raw) per launch
  profiles/pmc_summary.json         -- per kernel HBM bytes per launch, read by bench.py for roofline.traffic
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced
streams, so read bytes = 2 x FETCH_SIZE x 1024 there; WRITE_SIZE is exact for streaming stores.  Our kernels load
bytes/words/dwords (not 16 B/lane): calibrated on level0_kernel (reads exactly w*h bytes per image with dword loads,
raw FETCH_SIZE = 0.90 of that with the remainder L2 hits) the RAW figure is the right one here, so `traffic` uses
raw; the x2 figure is kept beside it.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """'void (anonymous namespace)::klt_stage2_kernel<9>(ov2_pyr_view, ...)' -> 'klt_stage2_kernel' (the name
    bench.py's hipEvent table uses); template arguments are dropped, instantiations of one kernel are pooled."""
    n = name.strip('"').replace("(anonymous namespace)::", "")
    m = re.search(r"rocprim::\w+::detail::wrapped_(\w+?)_config", n)
    if m:   # hipCUB / rocPRIM device-wide primitives (key sorts and scans of the BA program build)
        return "rocprim_" + m.group(1)
    head = n.split("(", 1)[0]
    while True:
        t = re.sub(r"<[^<>]*>", "", head)
        if t == head:
            break
        head = t
    m = re.search(r"(\w+)\s*$", head)
    return m.group(1) if m else n


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")   # on the GPU box: a directory under gpurun_out/ (what gpurun merges back)
    os.makedirs(out, exist_ok=True)
    # gpurun merges every call's outputs into gpurun_out/: only the NEWEST file of a pass belongs to the current state
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]
    ks = newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        with open(ks[0]) as f, open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as g:
            r, w = csv.reader(f), csv.writer(g)
            for i, row in enumerate(r):
                if i:
                    row[0] = short(row[0])
                w.writerow(row)
    pmc = {}
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(lambda: [0.0, 0])
        for fn in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != ctr:
                        continue
                    a = acc[short(row["Kernel_Name"])]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
        for k, (s, n) in acc.items():
            pmc.setdefault(k, {})[ctr] = s / max(n, 1)
            pmc[k]["launches_" + ctr] = n
    with open(os.path.join(out, f"{tag}_pmc.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_per_launch_raw", "WRITE_SIZE_KiB_per_launch",
                    "hbm_bytes_per_launch_raw", "hbm_bytes_per_launch_fetch_x2"])
        summ = {}
        for k, v in sorted(pmc.items()):
            fe, wr = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            raw, cor = (fe + wr) * 1024.0, (2 * fe + wr) * 1024.0
            w.writerow([k, v.get("launches_FETCH_SIZE", 0), f"{fe:.1f}", f"{wr:.1f}", f"{raw:.0f}", f"{cor:.0f}"])
            summ[k] = {"hbm_bytes_per_launch": raw, "hbm_bytes_per_launch_fetch_x2": cor, "fetch_KiB": fe, "write_KiB": wr,
                       "source": f"profiles/{tag}_pmc.csv"}
    json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read())
    print(open(os.path.join(out, f"{tag}_pmc.csv")).read())


if __name__ == "__main__":
    main()
