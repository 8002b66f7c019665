"""where the wall time of the single-camera closed loop (host-pointer ABI) goes: python scripts/closed_loop_profile.py [N=120]"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ov2slam_amd import frontend as fe, slam_loop, synth_scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
scene = synth_scene.PlaneScene(N)
ctx = fe.Context(0)
gl = slam_loop.SlamLoop(slam_loop.HipBackend(ctx), synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H)
imgs = [(scene.left(t), scene.right(t)) for t in range(N)]
for t in range(10):
    gl.step(t, imgs[t][0], lambda k: imgs[k][1])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for t in range(10, N):
    gl.step(t, imgs[t][0], lambda k: imgs[k][1])
pr.disable()
dt = time.perf_counter() - t0
print(f"{N - 10} frames in {dt:.3f} s = {1e3 * dt / (N - 10):.2f} ms per frame, {len(gl.kfs)} keyframes")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
