"""closed loop through C++ (ov2::SlamManager of libov2host.so) on the synthetic plane scene: ms per frame, ATE against ground
truth, beside the Python loop over the same ABI.  usage: python scripts/closed_loop_cpp.py [frames] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, host_map, slam_loop, synth_scene as sc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scene = sc.PlaneScene(n)
t0 = time.perf_counter()
L = [scene.left(t) for t in range(n)]
R = [scene.right(t) for t in range(n)]
print(f"{n} stereo frames rendered in {time.perf_counter() - t0:.1f} s", flush=True)
gt = [scene.pose(t) for t in range(n)]
ctx = fe.Context(0)
out = {"frames": n, "scene": "textured plane, 752x480 rectified stereo, ~2 cm / frame", "keypoints_per_frame": "<= 308 (cell 35)"}


def run_cpp(policy, device_map, brief=False):
    cl = host_map.CppSlam(ctx, sc.K4, sc.BASELINE, sc.W, sc.H, policy=policy, device_map=device_map)
    if brief:   # use_brief + bdo_track_localmap (the reference's YAML defaults); the BRIEF test table is a stand-in for opencv_contrib's
        from ov2slam_amd import mapper
        cl.set_brief(mapper.random_brief_pattern(3))
    per, extra = [], {}
    try:
        for t in range(n):
            t1 = time.perf_counter()
            cl.step(0.05 * t, L[t], R[t])
            per.append(time.perf_counter() - t1)
        if brief:
            inv, _ = cl.check_map()
            extra = dict(map_matching=dict(keyframes=len(cl.kf_stats), described_per_keyframe_mean=float(np.mean([k["described"] for k in cl.kf_stats])),
                                           local_map_points_offered_mean=float(np.mean([k["local"] for k in cl.kf_stats])),
                                           merges_total=int(sum(k["matched"] for k in cl.kf_stats)), map_invariant_violations=inv))
    finally:
        cl.close()
    per = np.array(per)
    kf = np.array([bool(s["kf"]) for s in cl.stats])
    warm = np.arange(n) >= 20
    return dict(ms_per_frame_mean=1e3 * float(per[warm].mean()), ms_per_frame_median=1e3 * float(np.median(per[warm])),
                ms_per_non_keyframe_median=1e3 * float(np.median(per[warm & ~kf])), ms_per_keyframe_median=1e3 * float(np.median(per[warm & kf])),
                keyframes=int(kf.sum()), local_bas=int(sum(int(s["ba"]) for s in cl.stats)),
                tracked_mean=float(np.mean([s["tracked"] for s in cl.stats[1:]])), ate_rmse_m=slam_loop.ate_rmse(cl.traj, gt), **extra), cl


out["cpp_slam_loop_policy"], c1 = run_cpp("slam_loop", False)
out["cpp_reference_policies"], c2 = run_cpp(None, True)
out["cpp_reference_policies_with_brief_and_local_map_matching"], c3 = run_cpp(None, True, brief=True)
pl = slam_loop.SlamLoop(slam_loop.HipBackend(ctx), sc.K4, sc.BASELINE, sc.W, sc.H)
per = []
for t in range(n):
    t1 = time.perf_counter()
    pl.step(t, L[t], lambda q: R[q])
    per.append(time.perf_counter() - t1)
per = np.array(per)
out["python_loop_same_abi"] = dict(ms_per_frame_mean=1e3 * float(per[20:].mean()), ms_per_frame_median=1e3 * float(np.median(per[20:])),
                                   ate_rmse_m=slam_loop.ate_rmse(pl.traj, gt))
d = max(float(np.abs(a[:3] - b[:3]).max()) for a, b in zip(c1.traj, pl.traj))
out["cpp_vs_python_max_translation_diff_m"] = d
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
