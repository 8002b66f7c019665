#!/usr/bin/env python3
"""rewrite the headline block and the per-kernel table of profiles/README.md from profiles/r01_bench.json"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
rows = []
for k, v in b["kernels"].items():
    alg = v.get("alg_bytes_per_launch") or 0
    extra = f'{v["achieved_Tops"]:.1f} Tiop/s = {100 * v["valu_frac"]:.0f} % of the vector peak' if "achieved_Tops" in v else ""
    rows.append(f'| `{k}` | {v["launches"]} | {v["avg_us"]:.1f} | {alg / 1e6:.1f} | {v["achieved_GBs"]:.0f} | {extra} |')
table = "| kernel | launches | avg µs | algorithmic MB / launch | GB/s | |\n|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n"
p = os.path.join(ROOT, "profiles", "README.md")
s = open(p).read()
a, e = s.index("| kernel | launches |"), s.index("\nHBM traffic per launch")
s = s[:a] + table + s[e:]
rl, ba = b["roofline"], b["local_ba"]
hdr = f'''* **{b["value"]:.0f} tracked frames/s** ({b["ms_per_step"]:.3f} ms per step of 64 frames), CPU port of the same workload on one host core:
  {b["cpu_baseline"]["value"]:.1f} frames/s; concurrent local BA (50 KF / 10 k landmarks / 128 k residual blocks):
  {ba["value"]:.0f} LM iterations/s = {ba["solves_per_sec"]:.1f} solves/s (stand-alone: 6.4 ms per solve), CPU port {ba["cpu_baseline"]["value"]:.1f} iterations/s;
  local-BA set-up (`local_ba.setup`): hash-map walk {ba["setup"]["hash_map_walk_ms"]:.2f} ms, device map scans {ba["setup"]["device_map_scans_ms"]:.2f} ms.
* `roofline`: dominant kernel `klt_stage1_kernel` (the combined KLT launch: keypoints with a prior on 2 levels + keypoints without
  on the full pyramid), {rl["achieved"]:.0f} GB/s of algorithmic bytes = {100 * rl["frac"]:.1f} % of 8 TB/s at {rl["avg_launch_us"]:.0f} µs per launch (161 µs isolated);
  measured HBM traffic {rl["traffic"] / 1e6:.0f} MB per launch vs {rl["alg_bytes_per_launch"] / 1e6:.0f} MB algorithmic ({rl["traffic"] / rl["alg_bytes_per_launch"]:.2f}x: no wasted re-reads — at 64 x 2 MB
  per pyramid the working set no longer fits the caches, so the window rows do come from HBM). The kernel is bound by integer
  VALU issue (76 % of the SIMD cycles busy; it was texture-addresser bound before the windows went through LDS, DESIGN.md §7):
  `roofline.valu` = {rl["valu"]["achieved"]:.1f} Tiop/s of window arithmetic = {100 * rl["valu"]["frac"]:.0f} % of the vector peak.
'''
a, e = s.index("* **"), s.index("\nPer kernel (hipEvent averages")
s = s[:a] + hdr + s[e:]
open(p, "w").write(s)
print(hdr)
