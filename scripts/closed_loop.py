"""closed-loop run of the synthetic stereo sequence through the C ABI (GPU) and through the oracle (CPU), N frames:
per-frame pose agreement, trajectory RMSE against ground truth, TUM files under gpurun_out/ (SURVEY.md 8 row g).
usage: python scripts/closed_loop.py [N=200]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ov2slam_amd import frontend as fe, slam_loop, synth_scene
from oracle import oracle_py as O
from closed_loop_oracle import OracleBackend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scene = synth_scene.PlaneScene(N)
ctx = fe.Context(0)
mk = lambda b: slam_loop.SlamLoop(b, synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H)
gl = mk(slam_loop.HipBackend(ctx))
# frames rendered up front; the two loops run one after the other (interleaving them lets the GPU clock down during
# every CPU step, which shows up as milliseconds of wake-up latency in the next ABI call)
lefts = [scene.left(t) for t in range(N)]
rights = {}
def right(t):
    if t not in rights:
        rights[t] = scene.right(t)
    return rights[t]
for t in range(N):                     # which frames become keyframes is decided by the loop: render those on demand once
    gl.step(t, lefts[t], right)
    if t % 20 == 19:
        print(f"frame {t + 1}: tracked {gl.stats[-1]['tracked']} kfs {len(gl.kfs)} lms {len(gl.lms)}", flush=True)
# timed passes on fresh loops over the now complete frame store
gl2, ol = mk(slam_loop.HipBackend(ctx)), mk(OracleBackend(O))
tg = tc = 0.0
for t in range(N):
    a = time.perf_counter(); gl2.step(t, lefts[t], right); b = time.perf_counter()
    if t >= 10: tg += b - a            # steady state: the first frames carry allocations
for t in range(N):
    a = time.perf_counter(); ol.step(t, lefts[t], right); b = time.perf_counter()
    if t >= 10: tc += b - a
assert all(np.array_equal(x, y) for x, y in zip(gl.traj, gl2.traj))   # the GPU loop is deterministic
gt = [scene.pose(t) for t in range(N)]
dpos = max(np.abs(a[:3] - b[:3]).max() for a, b in zip(gl.traj, ol.traj))
dq = max(np.abs(a[3:] - b[3:]).max() for a, b in zip(gl.traj, ol.traj))
out = dict(frames=N, keyframes=len(gl.kfs), landmarks=len(gl.lms),
           max_abs_pose_diff_gpu_vs_oracle=dict(translation_m=dpos, quaternion=dq),
           same_track_counts=[s["tracked"] for s in gl.stats] == [s["tracked"] for s in ol.stats],
           ate_rmse_m=dict(gpu=slam_loop.ate_rmse(gl.traj, gt), oracle=slam_loop.ate_rmse(ol.traj, gt)),
           path_length_m=float(sum(np.linalg.norm(gt[k + 1][:3] - gt[k][:3]) for k in range(N - 1))),
           loop_ms_per_frame_after_warmup=dict(gpu_abi=1e3 * tg / max(N - 10, 1), oracle_cpu=1e3 * tc / max(N - 10, 1),
                                               note="whole Python loop incl. the host-side problem assembly; scripts/closed_loop_profile.py splits it"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
slam_loop.write_tum(os.path.join(ROOT, "gpurun_out", "closed_loop_gpu.tum"), gl.traj)
slam_loop.write_tum(os.path.join(ROOT, "gpurun_out", "closed_loop_oracle.tum"), ol.traj)
slam_loop.write_tum(os.path.join(ROOT, "gpurun_out", "closed_loop_gt.tum"), gt)
print(json.dumps(out))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "closed_loop.json"), "w"), indent=1)
