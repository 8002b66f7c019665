"""How uneven is the tracking work inside a wave?  Per keypoint the kernels return (LK iterations | level passes << 16) of both
stages (Workload.step(want_work=True)); waves of the three-lane kernel hold 20 consecutive keypoints of the compacted list.
Prints, for the headline stream and the hard stream: mean iterations per keypoint, mean over waves of the slowest keypoint,
and their ratio -- the factor a perfectly balanced schedule could gain on the iteration part.
usage: python scripts/klt_tail_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ov2slam_amd import frontend as fe, synth

ctx = fe.Context(0)
for name, gap, sig in (("headline", 3, 1.0), ("hard", 9, 3.0)):
    wl = bench.Workload(ctx, fe, synth, 16, 2048, 8, seed=synth.SEED_IMG, gap=gap, prior_sigma=sig)
    wl.detect = False
    its, passes = [], []
    for k in range(6):
        wl.step(10 ** 9, want_work=True)
        ctx.synchronize()
        if k:
            w = wl.work.get()[: wl.n]            # stage 1 (the long launch)
            its.append((w & 0xffff).astype(np.int64)); passes.append((w >> 16).astype(np.int64))
    # would last frame's work predict this frame's?  group frame k's keypoints by their work in frame k - 1
    for G in (20,):
        r = []
        for k in range(1, len(its)):
            lv = (passes[k] > 0) & (passes[k - 1] > 0)
            order = np.argsort(its[k - 1][lv], kind="stable")
            v = its[k][lv][order]
            v = v[: len(v) // G * G].reshape(-1, G)
            r.append(v.max(1).mean() / v.mean())
        print(f"   {name}: waves of {G} formed by the PREVIOUS frame's work: slowest / mean = {np.mean(r):.2f}")
    it, ps = np.concatenate(its), np.concatenate(passes)
    live = ps > 0
    print(f"{name}: {live.mean():.3f} of the keypoints tracked in stage 1; iterations per keypoint mean {it[live].mean():.2f}, "
          f"p50 {np.percentile(it[live], 50):.0f}, p90 {np.percentile(it[live], 90):.0f}, p99 {np.percentile(it[live], 99):.0f}, max {it[live].max()}; "
          f"level passes per keypoint {ps[live].mean():.2f}")
    for G in (20, 8, 4):
        v = it[live]
        v = v[: len(v) // G * G].reshape(-1, G)
        print(f"   waves of {G:2d} keypoints (list order): mean of the slowest keypoint {v.max(1).mean():.2f} = {v.max(1).mean() / v.mean():.2f} x the mean; "
              f"sorted by work first: {np.sort(it[live])[: len(v) * G].reshape(-1, G).max(1).mean() / v.mean():.2f} x")
    del wl
