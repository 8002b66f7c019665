"""stand-alone timing of ov2_ba_solve_batch: B copies of the bench window (50 KF / 10 k landmarks) per call.
usage: python scripts/ba_batch_time.py [B ...]      (BA_DEV=1: windows resident in HBM, ov2_ba_solve_batch_dev)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, local_ba, synth_ba

Bs = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
kf, lm = int(os.environ.get("BA_KFS", 50)), int(os.environ.get("BA_LMS", 10000))
ctx = fe.Context(0)
P0 = synth_ba.make_window(kf, lm, inv_depth=True, seed=20211, max_obs=7)
opt = local_ba.Optimizer(ctx)
dev = bool(int(os.environ.get("BA_DEV", "0")))
for B in Bs:
    best, its = 1e9, 0
    Ds = [local_ba.DeviceBaProblem(ctx, P0) for _ in range(B)] if dev else None
    for rep in range(4):
        if dev:
            for D in Ds:
                D.reset()
            ctx.synchronize()
            t = time.perf_counter()
            R = opt.localBA_batch_dev(Ds)
            dt = time.perf_counter() - t
            its = sum(r.n_log - 1 - (1 if r.l2_done else 0) for r in R)
        else:
            Ps = [P0.copy() for _ in range(B)]
            t = time.perf_counter()
            R = opt.localBA_batch(Ps, want_flags=False)
            dt = time.perf_counter() - t
            its = sum(sum(r.summary()["iterations"]) for r in R)
        best = min(best, dt)
    print(f"B={B:4d}: {1e3 * best:8.2f} ms per batch, {1e3 * best / B:7.3f} ms per window, {its / best:9.0f} LM it/s, "
          f"{B / best:8.0f} solves/s  ({P0.n_res} residual blocks per window, {its // B} LM iterations per window)", flush=True)
