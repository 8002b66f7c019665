#!/usr/bin/env python3
"""times ov2_pnp_solve_batch (host pointers in, pose out) against the CPU oracle on the same frames (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth_ba
from ov2slam_amd.multi_view_geometry import MultiViewGeometry
from oracle import oracle_py as O

ctx = fe.Context(0)
mvg = MultiViewGeometry(ctx)
for B, n in [(1, 300), (1, 2048), (16, 2048), (64, 2048), (256, 2048)]:
    frames = [synth_ba.make_pnp(n, seed=3 * b + 1) for b in range(B)]
    args = ([p["unpx"] for p in frames], [p["wpts"] for p in frames], np.stack([p["Twc0"] for p in frames]), 5, 5.9915,
            True, True, np.stack([p["K"] for p in frames]))
    mvg.ceresPnP_batch(*args)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ok, T, outs, it = mvg.ceresPnP_batch(*args); ts.append(time.perf_counter() - t0)
    ctx.kernel_timing(True); ctx.kernel_times()
    mvg.ceresPnP_batch(*args)
    kt = ctx.kernel_times(); ctx.kernel_timing(False)
    m = min(B, 8)
    t0 = time.perf_counter()
    for p in frames[:m]:
        O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"])
    cpu = (time.perf_counter() - t0) / m
    k = kt.get("pnp_kernel", (0, 0))[0]
    print(f"B={B} n={n}: wall {min(ts)*1e3:.3f} ms ({B/min(ts):.0f} poses/s incl. marshalling+H2D/D2H), kernel {k:.3f} ms "
          f"({B/(k*1e-3) if k else 0:.0f} poses/s), LM iters {it[0].tolist()}, CPU oracle {cpu*1e3:.3f} ms/pose "
          f"({1/cpu:.0f} poses/s, 1 thread)", flush=True)
