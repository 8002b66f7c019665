#!/usr/bin/env python3
"""host-phase trace of ov2_ba_solve (OV2_BA_TRACE=1) on the bench window"""
import sys, os, time
os.environ["OV2_BA_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov2slam_amd import frontend as fe, local_ba, synth_ba
ctx = fe.Context(0)
opt = local_ba.Optimizer(ctx)
for (nkf, nlm) in [(50, 10000), (100, 20000)]:
    P0 = synth_ba.make_window(nkf, nlm, inv_depth=True, max_obs=7, seed=20211)
    for _ in range(4):
        P = P0.copy(); t0 = time.perf_counter(); R = opt.localBA(P); print(f"kf={nkf}: {1e3*(time.perf_counter()-t0):.2f} ms total, iters {R.summary()['iterations']}", flush=True)
