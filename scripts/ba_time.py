#!/usr/bin/env python3
"""times ov2_ba_solve on synthetic windows and prints the per-kernel hipEvent breakdown (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, local_ba, synth_ba

ctx = fe.Context(0)
opt = local_ba.Optimizer(ctx)
for (nkf, nlm, inv, mo) in [(20, 2000, True, 12), (50, 10000, True, 7), (100, 20000, True, 7), (100, 20000, False, 7)]:
    P0 = synth_ba.make_window(nkf, nlm, inv_depth=inv, max_obs=mo)
    opt.localBA(P0.copy())                       # warm-up
    ts = []
    for _ in range(3):
        P = P0.copy()
        t0 = time.perf_counter(); R = opt.localBA(P); ts.append(time.perf_counter() - t0)
    s = R.summary()
    iters = sum(s["iterations"])
    print(f"kf={nkf} lm={nlm} inv={inv} res={P0.n_res}: {min(ts)*1e3:.2f} ms/solve, LM iters {s['iterations']} -> "
          f"{iters/min(ts):.0f} it/s  outliers {s['outliers']}", flush=True)
    ctx.kernel_timing(True); ctx.kernel_times()
    P = P0.copy(); t0 = time.perf_counter(); opt.localBA(P); el = time.perf_counter() - t0
    kt = ctx.kernel_times(); ctx.kernel_timing(False)
    tot = sum(v[0] for v in kt.values())
    print(f"   instrumented {el*1e3:.2f} ms, kernels {tot:.2f} ms:", ", ".join(f"{k[3:-7] if k.startswith('ba_') else k} {v[0]:.3f}ms/{v[1]}" for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])), flush=True)
