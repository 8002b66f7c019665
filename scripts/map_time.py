#!/usr/bin/env python3
"""set-up stage of Optimizer::localBA: hash-map walk (C++ host mirror of the reference) vs scans of the device map
mirror (ov2_map_local_ba_setup), wall clock of the whole call incl. the two synchronisations, the D2H of the flat
problem and the host-side id maps; plus the device time of the scans alone (hipEvent).  GPU box."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, host_map, synth_ba

ctx = fe.Context(0)
L = host_map.lib()
for n_kf, n_lm in ((50, 10000), (120, 30000), (250, 60000)):
    P = synth_ba.make_window(n_kf, n_lm, inv_depth=True, seed=1, max_obs=7)
    hm = host_map.HostMap(P)
    hm.attach_device(ctx)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    def run(fn, reps=20):
        fn(hm.h, hm.newkf, C.byref(a), C.byref(b), C.byref(c))
        t = time.perf_counter()
        for _ in range(reps):
            fn(hm.h, hm.newkf, C.byref(a), C.byref(b), C.byref(c))
        return (time.perf_counter() - t) / reps * 1e3
    t_walk = run(L.ov2h_local_ba_setup)
    dims = (a.value, b.value, c.value)
    t_dev = run(L.ov2h_local_ba_setup_dev)
    assert dims == (a.value, b.value, c.value)
    ctx.kernel_timing(True); ctx.kernel_times()
    R = 10
    for _ in range(R):
        L.ov2h_local_ba_setup_dev(hm.h, hm.newkf, C.byref(a), C.byref(b), C.byref(c))
    kt = ctx.kernel_times(); ctx.kernel_timing(False)
    k_us = kt.get("map_setup_kernels", (0, 0))
    n_obs = P.n_res + len(P.lm)
    print(f"map {n_kf} KF / {n_lm} landmarks / ~{n_obs} observation rows -> problem {dims[0]} poses, {dims[1]} landmarks, "
          f"{dims[2]} residual blocks: hash-map walk {t_walk:.2f} ms, device set-up {t_dev:.3f} ms per call "
          f"({k_us[0] / R * 1e3:.0f} us in {k_us[1] // R} launches)", flush=True)
