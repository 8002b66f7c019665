"""host-side enqueue cost per front-end call (the bench loop is host-bound when this exceeds the GPU time per step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth
import bench
ctx = fe.Context(0)
wl = bench.Workload(ctx, fe, synth, 64, 2048, 8, seed=1)
for _ in range(50):
    wl.step(6)
ctx.synchronize()
N = 600
t = {"pre": 0.0, "klt": 0.0, "rel": 0.0}
prev = None
t0 = time.perf_counter()
for i in range(N):
    c = i % wl.L
    a = time.perf_counter()
    cur = fe.preprocess_images(ctx, wl.left[c], True, 3.0, 9, 3)
    b = time.perf_counter()
    if prev is not None:
        wl.trk.kltTracking_dev(prev, cur, 9, 3, 30.0, 0.5, wl.kps[c], wl.pri[c], wl.has[c], wl.out_xy, wl.out_st, wl.n, wl.img_idx, wl.p3p, None)
        d = time.perf_counter()
        prev.release()
        e = time.perf_counter()
        t["klt"] += d - b; t["rel"] += e - d
    t["pre"] += b - a
    prev = cur
    if i % 100 == 99:
        ctx.synchronize()
tot = time.perf_counter() - t0
ctx.synchronize()
print({k: round(1e6 * v / N, 1) for k, v in t.items()}, "us per call; loop", round(1e6 * tot / N, 1), "us per step (with a sync every 100)")
