#!/usr/bin/env python3
"""ov2_triangulate_pairs_dev: device time per launch (hipEvent) for batches of keypoint pairs, against the algorithmic
bytes (92 B read + 57 B written per pair with world points and parallax) -- GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth_tri

ctx = fe.Context(0)
L = ctx.lib
for n in (2048, 131072, 2000000):
    s = synth_tri.make_pairs(min(n, 20000), seed=1, G=8)
    rep = (n + len(s["grp"]) - 1) // len(s["grp"])
    t = lambda a: np.ascontiguousarray(np.tile(a, (rep,) + (1,) * (a.ndim - 1))[:n])
    d = {k: ctx.to_device(t(s[k])) for k in ("grp", "bv_a", "bv_b", "unpx_a", "unpx_b")}
    dT, dW = ctx.to_device(s["T_ab"]), ctx.to_device(s["Twc_a"])
    pt, wpt, par, st = ctx.empty((n, 3), np.float64), ctx.empty((n, 3), np.float64), ctx.empty((n,), np.float64), ctx.empty((n,), np.uint8)
    Ka, Kb = np.ascontiguousarray(s["K_a"]), np.ascontiguousarray(s["K_b"])
    call = lambda: L.ov2_triangulate_pairs_dev(ctx.h, n, 0, 8, dT.ptr, dW.ptr, d["grp"].ptr, d["bv_a"].ptr, d["bv_b"].ptr,
                                               d["unpx_a"].ptr, d["unpx_b"].ptr, Ka.ctypes.data, Kb.ctypes.data, 3.0, pt.ptr,
                                               wpt.ptr, par.ptr, st.ptr)
    for _ in range(3):
        assert call() == 0
    ctx.synchronize()
    ctx.kernel_timing(True); ctx.kernel_times()
    R = 20
    for _ in range(R):
        call()
    kt = ctx.kernel_times(); ctx.kernel_timing(False)
    us = kt["tri_kernel"][0] / R * 1e3
    print(f"n={n}: {us:.1f} us per launch, {149.0 * n / us / 1e3:.0f} GB/s of algorithmic bytes", flush=True)
