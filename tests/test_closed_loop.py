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


@pytest.mark.gpu
def test_cpp_loop_matches_the_python_loop(ctx, scene):
    """ov2::SlamManager (libov2host.so: visualTracking -> createKeyframe -> Mapper::run -> local BA, all in C++ over the C
    ABI) with the fixed stand-ins of SlamLoop for the keyframe decision / BA window reproduces the Python loop over the same
    ABI pose by pose (the Python loop in turn equals the loop over the CPU oracle, test above): same track counts, same
    landmark ids, poses to 1e-9 (the two differ in the order their hash maps hand keypoints to ceresPnP and in how they
    compose the motion prediction)."""
    from ov2slam_amd import host_map
    n = 16
    pl = _run(slam_loop.HipBackend(ctx), scene, n)
    cl = host_map.CppSlam(ctx, synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H, policy="slam_loop")
    try:
        for t in range(n):
            cl.step(0.05 * t, scene.left(t), scene.right(t))
        ids, xyz = cl.landmarks()
    finally:
        cl.close()   # before the session's context goes away
    assert [int(s["tracked"]) for s in cl.stats[1:]] == [len(k) for k in _kps_per_frame(pl)][1:]
    assert [bool(s["kf"]) for s in cl.stats] == [s["kf"] for s in pl.stats]
    for t, (a, b) in enumerate(zip(cl.traj, pl.traj)):
        assert np.abs(a[:3] - b[:3]).max() <= 1e-9 and np.abs(np.abs(a[3:]) - np.abs(b[3:])).max() <= 1e-9, (t, a, b)
    assert ids.tolist() == sorted(pl.lms)
    assert np.abs(xyz - np.array([pl.lms[i] for i in sorted(pl.lms)])).max() < 1e-8


def _kps_per_frame(loop):
    """keypoints in the current frame at the END of every step of a SlamLoop (after PnP outliers, detection and BA removals)"""
    return loop.kps_log


@pytest.mark.gpu
def test_cpp_loop_with_the_reference_policies_follows_ground_truth(ctx, scene):
    """the same driver with the reference's own heuristics (checkNewKfReq, se3 motion model, covisibility local BA through
    the device map mirror, rectified disparity triangulation): no oracle loop has those policies, so the check is ground truth"""
    from ov2slam_amd import host_map
    cl = host_map.CppSlam(ctx, synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H, policy=None, device_map=True)
    n = 64   # the scene moves slowly: the reference's rule for this stream is 'stereo and more than 1 s since the last keyframe'
    try:
        for t in range(n):
            cl.step(0.05 * t, scene.left(t), scene.right(t))
        ids, xyz = cl.landmarks()
    finally:
        cl.close()
    gt = [scene.pose(t) for t in range(n)]
    assert slam_loop.ate_rmse(cl.traj, gt) < 0.01
    assert sum(int(s["kf"]) for s in cl.stats) >= 3 and sum(int(s["ba"]) for s in cl.stats) >= 2
    assert all(s["tracked"] > 150 for s in cl.stats[1:])
    assert len(ids) > 200 and np.median(np.abs(xyz @ scene.nrm - scene.d)) < 0.05


class _DevHandle:
    """a borrowed ov2_map* with DeviceMap's download()"""

    def __init__(self, ctx, h):
        from ov2slam_amd import device_map as DM
        self.ctx, self.L, self.h = ctx, ctx.lib, h
        self._p = DM.DeviceMap._p
        self.download = lambda: DM.DeviceMap.download(self)


@pytest.mark.gpu
def test_cpp_loop_with_brief_matches_lost_points_back_into_the_map(ctx, scene):
    """use_brief / bdo_track_localmap (on in every parameter file of the reference): keyframes describe their keypoints (BRIEF
    with a caller-supplied test table) and Mapper::matchingToLocalMap merges re-detected points into the map points they
    duplicate (MapManager::mergeMapPoints).  A grey band painted over two frames makes the tracker lose a stripe of
    keypoints; the next keyframes re-detect the stripe, and the old map points -- still in the local map of the covisible
    keyframes -- must take the new observations over.  Checked: merges happen, the host map keeps its invariants (every
    observation two-sided, descriptors only from observers), the device mirror equals the host map after the edits
    (a merged observation = a dead row + an appended one), and the trajectory stays on the ground truth."""
    from ov2slam_amd import device_map as DM, host_map, mapper
    n = 40
    cl = host_map.CppSlam(ctx, synth_scene.K4, synth_scene.BASELINE, synth_scene.W, synth_scene.H, policy="slam_loop", kf_every=5,
                          ba_window=0, device_map=True)
    cl.set_brief(mapper.random_brief_pattern(3))
    try:
        for t in range(n):
            il, ir = scene.left(t).copy(), scene.right(t).copy()
            if t in (11, 12, 13):
                il[:, 300:460] = 128
            cl.step(0.05 * t, il, ir)
        inv, total = cl.check_map()
        host = cl.export_map()
        cl.flush_device()
        dev = DM.canonical_state(_DevHandle(ctx, cl.device_handle()).download())
    finally:
        cl.close()
    ks = cl.kf_stats
    assert len(ks) == 8 and all(k["described"] > 100 for k in ks) and all(k["local"] > 0 for k in ks[2:])
    assert sum(k["matched"] for k in ks) >= 10, ks
    assert inv["kp_without_mp"] == inv["kp_not_listed"] == inv["observer_without_kp"] == inv["desc_without_observer"] == 0, inv
    gt = [scene.pose(t) for t in range(n)]
    assert slam_loop.ate_rmse(cl.traj, gt) < 0.01
    kf_h, lm_h, ob_h = host
    kf_d, lm_d, ob_d = dev
    assert sorted(kf_h) == sorted(kf_d) and all(np.allclose(kf_h[k], kf_d[k], atol=1e-12) for k in kf_h)
    assert sorted(lm_h) == sorted(lm_d)
    assert set(ob_h) == set(ob_d)
    assert all(bool(ob_h[o]) == bool(ob_d[o]) for o in ob_h)
