#!/usr/bin/env python3
"""per-level timing of the pyramid kernels (hipEvent per launch) for batches of 752x480 images (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, synth

ctx = fe.Context(0)
S = synth.StereoStream()
img = S.left(0)
Bs = [int(a) for a in sys.argv[1:]]   # with arguments: only those batch sizes, full pyramid only (profiling)
for B in (Bs or (1, 16, 64)):
    ims = fe.Images(ctx, B, 752, 480)
    for b in range(B):
        ims.upload(b, img)
    prev = {}
    for nl in ((3,) if Bs else (0, 1, 2, 3)):
        for _ in range(3):
            fe.preprocess_images(ctx, ims, True, 3.0, 9, nl).release()
        ctx.synchronize()
        ctx.kernel_timing(True); ctx.kernel_times()
        R = 20
        for _ in range(R):
            fe.preprocess_images(ctx, ims, True, 3.0, 9, nl).release()
        kt = ctx.kernel_times(); ctx.kernel_timing(False)
        lv = kt.get("level_kernel", (0, 0))[0] / R * 1e3
        nlv = kt.get("level_kernel", (0, 0))[1] // R
        print(f"B={B} nlevels={nl}: clahe_lut {kt['clahe_lut_kernel'][0]/R*1e3:.1f} us, level0 {kt.get('level0_kernel',(0,0))[0]/R*1e3:.1f} us, "
              f"level kernels total {lv:.1f} us ({nlv} launches)", flush=True)
