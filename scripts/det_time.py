#!/usr/bin/env python3
"""wall time of ov2_detect_grid_batch (host pointers in/out, two internal syncs) vs its kernels"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = fe.Context(0)
S = synth.StereoStream()
ims = fe.Images(ctx, B, 752, 480)
for b in range(B):
    ims.upload(b, S.left(3 * (b % 5)))
pyr = fe.preprocess_images(ctx, ims)
base = synth.grid_keypoints(2048)
rng = np.random.default_rng(1)
cur = [base[rng.uniform(size=len(base)) < 0.85] for _ in range(B)]
th = np.full(B, 0.001)
for _ in range(3):
    out = fe.detect_grid_batch(ctx, pyr, 13, fe.OV2_DETECT_MINEIG if hasattr(fe, "OV2_DETECT_MINEIG") else 1, th.copy(), cur)
ctx.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); out = fe.detect_grid_batch(ctx, pyr, 13, 1, th.copy(), cur); ts.append(time.perf_counter() - t0)
ctx.kernel_timing(True); ctx.kernel_times()
out = fe.detect_grid_batch(ctx, pyr, 13, 1, th.copy(), cur)
kt = ctx.kernel_times(); ctx.kernel_timing(False)
print(f"B={B}: wall {1e3*min(ts):.3f} ms (median {1e3*np.median(ts):.3f}); kernels {sum(v[0] for v in kt.values()):.3f} ms:", {k: round(v[0], 3) for k, v in kt.items()})
