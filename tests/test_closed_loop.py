"""Closed-loop synthetic stereo sequence (SURVEY.md 8 row g; no EuRoC data exists here or on the GPU box): the
SlamManager-like loop of ov2slam_amd/slam_loop.py -- preprocess, KLT with motion-model priors, ceresPnP, and per keyframe
detection, stereo matching, triangulation, local BA -- on exact renderings of a textured plane with a known trajectory.
CPU test: the loop over the oracle follows the ground truth.  GPU test: the loop through the C ABI gives the same poses
as the loop over the oracle, frame by frame (north_star: BA pose states within 1e-4 relative)."""
import numpy as np
import pytest

from ov2slam_amd import slam_loop, synth_scene
from closed_loop_oracle import OracleBackend


def _run(backend, scene, n):
    loop = slam_loop.SlamLoop(backend, synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H)
    for t in range(n):
        loop.step(t, scene.left(t), scene.right)
    return loop


@pytest.fixture(scope="module")
def scene():
    return synth_scene.PlaneScene(40)


@pytest.fixture(scope="module")
def oracle_loop(oracle, scene):
    return _run(OracleBackend(oracle), scene, 16)


def test_oracle_loop_follows_ground_truth(oracle_loop, scene):
    loop = oracle_loop
    gt = [scene.pose(t) for t in range(len(loop.traj))]
    assert all(s["tracked"] > 150 for s in loop.stats[1:])
    assert len(loop.kfs) == 4 and all("ba" in s for s in loop.stats if s["kf"] and s["frame"] > 0)
    assert slam_loop.ate_rmse(loop.traj, gt) < 0.01                       # 1 cm over 16 frames of ~2 cm motion each
    # landmarks sit on the plane
    X = np.array(list(loop.lms.values()))
    assert len(X) > 200 and np.median(np.abs(X @ scene.nrm - scene.d)) < 0.05


@pytest.mark.gpu
def test_hip_loop_matches_oracle_loop(ctx, oracle_loop, scene):
    gl = _run(slam_loop.HipBackend(ctx), scene, 16)
    ol = oracle_loop
    for t, (a, b) in enumerate(zip(gl.traj, ol.traj)):
        assert np.abs(a[:3] - b[:3]).max() <= 1e-4 * max(1.0, np.abs(b[:3]).max()), t
        assert np.abs(a[3:] - b[3:]).max() <= 1e-4, t
    assert [s["tracked"] for s in gl.stats] == [s["tracked"] for s in ol.stats]
    assert sorted(gl.lms) == sorted(ol.lms)
