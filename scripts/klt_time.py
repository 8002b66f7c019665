#!/usr/bin/env python3
"""isolated timing of the two KLT stage kernels (nothing else on the GPU): python scripts/klt_time.py [seqs [frame_gap [prior_sigma]]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GAP = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # stream frames between the two images
SIGMA = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0  # prior error (px)
N = 2048
ctx = fe.Context(0)
S = synth.StereoStream()
i0, i1 = fe.Images(ctx, B, 752, 480), fe.Images(ctx, B, 752, 480)
base = synth.grid_keypoints(N)
ks, ps, hs = [], [], []
for b in range(B):
    t0, t1 = GAP * (b % 5), GAP * (b % 5) + GAP
    i0.upload(b, S.left(t0)); i1.upload(b, S.left(t1))
    pri, has = synth.make_priors(base, S.flow(t0, t1, base), sigma=SIGMA, seed=b)
    ks.append(base); ps.append(pri); hs.append(has)
p0, p1 = fe.preprocess_images(ctx, i0), fe.preprocess_images(ctx, i1)
n = B * N
d_k, d_p, d_h = ctx.to_device(np.concatenate(ks)), ctx.to_device(np.concatenate(ps)), ctx.to_device(np.concatenate(hs))
d_i = ctx.to_device(np.repeat(np.arange(B, dtype=np.int32), N))
d_o, d_s, d_r, d_w = ctx.empty((n, 2), np.float32), ctx.empty((n,), np.uint8), ctx.empty((B,), np.int32), ctx.empty((2 * n,), np.uint32)
trk = fe.FeatureTracker(ctx, 30, 0.01)
for _ in range(3):
    trk.kltTracking_dev(p0, p1, 9, 3, 30.0, 0.5, d_k, d_p, d_h, d_o, d_s, n, d_i, d_r, d_w)
ctx.synchronize()
w = d_w.get()
it, ps_ = (w & 0xffff), (w >> 16)
print(f"B={B}: tracked {d_s.get().mean():.3f}; 2-level passes on {np.count_nonzero(ps_[:n])} kps ({it[:n].sum()} iterations, {ps_[:n].sum()} level passes); "
      f"full-pyramid passes on {np.count_nonzero(ps_[n:])} kps ({it[n:].sum()} iterations, {ps_[n:].sum()} level passes)")
a = it[:n][ps_[:n] > 0].astype(np.float64)
for g in (4, 8):
    m = a[:len(a) // g * g].reshape(-1, g)
    print(f"  divergence estimate, {g} keypoints per wave: mean iterations {a.mean():.2f}, mean of per-wave max {m.max(1).mean():.2f} "
          f"(lock-step efficiency {a.mean() / m.max(1).mean():.2f}); histogram of iterations {np.bincount(a.astype(int))[:32].tolist()}")
ctx.kernel_timing(True); ctx.kernel_times()
R = 20
for _ in range(R):
    trk.kltTracking_dev(p0, p1, 9, 3, 30.0, 0.5, d_k, d_p, d_h, d_o, d_s, n, d_i, d_r, None)
kt = ctx.kernel_times(); ctx.kernel_timing(False)
for k, v in kt.items():
    print(f"  {k}: {v[0] / v[1] * 1e3:.1f} us")
