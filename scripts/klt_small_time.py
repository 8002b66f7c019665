"""EuRoC-sized tracking calls: per-frame chain (CLAHE + pyramid build + two-stage KLT [+ ceresPnP]) for `seqs` sequences of
`kps` keypoints, with the 3-lane (in-kernel Scharr, no gradient planes) and the 16-lane (gradient planes on demand) kernels.
usage: python scripts/klt_small_time.py [kps] [seqs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ov2slam_amd import frontend as fe, synth

kps = int(sys.argv[1]) if len(sys.argv) > 1 else 308
seqs_list = [int(a) for a in sys.argv[2:]] or [1, 8]
ctx = fe.Context(0)
for seqs in seqs_list:
    wl = bench.Workload(ctx, fe, synth, seqs, kps, 8, seed=synth.SEED_IMG, det_cell=35)
    for pnp in (False, True):
        if pnp:
            wl.enable_pnp(seed=7)
        for lanes in (0, 3, 16):
            ctx.set_klt_lanes(lanes)
            for _ in range(3 * wl.L):
                wl.step(10 ** 9)
            ctx.synchronize()
            n = 600
            t = time.perf_counter()
            for _ in range(n):
                wl.step(10 ** 9)
            ctx.synchronize()
            thr = (time.perf_counter() - t) / n
            lat = []
            for _ in range(100):
                ctx.synchronize()
                t = time.perf_counter()
                wl.step(10 ** 9)
                ctx.synchronize()
                lat.append(time.perf_counter() - t)
            print(f"seqs {seqs:3d} kps {kps} pnp {int(pnp)} lanes {lanes:2d}: {1e6 * thr:7.1f} us per frame-batch back to back, "
                  f"{1e6 * np.median(lat):7.1f} us latency of one synchronised frame-batch", flush=True)
    ctx.set_klt_lanes(0)
