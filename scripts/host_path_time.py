#!/usr/bin/env python3
"""latency of the HOST-buffer entry points a single-sequence integration calls per frame (PCIe + synchronisation
included): ov2_pyramid_build on a 752x480 image, ov2_klt_track_fb on 2048 keypoints.  GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth

ctx = fe.Context(0)
S = synth.StereoStream()
I0, I1 = S.left(0), S.left(3)
kps = synth.grid_keypoints(2048)
trk = fe.FeatureTracker(ctx, 30, 0.01)
p0 = fe.preprocess_image(ctx, I0)
for _ in range(5):
    fe.preprocess_image(ctx, I1).release()
R = 50
t = time.perf_counter()
for _ in range(R):
    fe.preprocess_image(ctx, I1).release()
ctx.synchronize()
t_pyr = (time.perf_counter() - t) / R * 1e3
p1 = fe.preprocess_image(ctx, I1)
for _ in range(5):
    trk.fbKltTracking(p0, p1, 9, 3, 30.0, 0.5, kps, kps)
t = time.perf_counter()
for _ in range(R):
    trk.fbKltTracking(p0, p1, 9, 3, 30.0, 0.5, kps, kps)
t_klt = (time.perf_counter() - t) / R * 1e3
print(f"ov2_pyramid_build (752x480 host image, CLAHE, 4 levels): {t_pyr:.3f} ms per call; "
      f"ov2_klt_track_fb (2048 host keypoints, fwd-bwd, 4 levels): {t_klt:.3f} ms per call", flush=True)
