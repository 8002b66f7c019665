"""Device map mirror (ov2_map, csrc/map.hip): the set-up stage of Optimizer::localBA as linear scans of the observation
table equals the reference's hash-map walk (src/optimizer.cpp:43-430, restated in ov2slam_amd/host Optimizer::setupLocalBA)
-- same keyframes with the same constness, same landmarks / anchors, same multiset of residual blocks, same bad
landmarks -- on fresh maps, after incremental edits pushed through the MapManager hooks, and after a local BA's own
update stage.  Index orders differ by design (ascending ids here), so problems are compared keyed by kfid / lmid."""
import ctypes as C

import numpy as np
import pytest

from ov2slam_amd import _lib, host_map, synth_ba

pytestmark = pytest.mark.gpu


def canon(pb):
    if pb["aborted"]:
        return "aborted"
    poses = {int(k): (int(c), tuple(p)) for k, c, p in zip(pb["pose_kfid"], pb["pose_const"], pb["pose"])}
    lms = {int(l): (tuple(v), int(a), tuple(uv)) for l, v, a, uv in zip(pb["lm_lmid"], pb["lm"], pb["lm_anchor_kfid"], pb["lm_anchor_uv"])}
    res = sorted((int(t), int(k), int(l), float(uv[0]), float(uv[1]))
                 for t, k, l, uv in zip(pb["res_type"], pb["res_kfid"], pb["res_lmid"], pb["res_uv"]))
    return poses, lms, res


def assert_same(hm, hd=None, tol=0.0):
    """hm: map set up by the hash-map walk, hd: identical map set up through its device mirror (the same object when
    no isBad() landmark is around: MapPoint::isBad() clears is3d_ as a side effect (src/map_point.cpp:219), so two
    set-ups of ONE map do not see the same state)"""
    hd = hd or hm
    a = hm.setup_local_ba(dev=False)
    bad_a = hm.bad_lmids()
    b = hd.setup_local_ba(dev=True)
    bad_b = hd.bad_lmids()
    assert np.array_equal(bad_a, bad_b), "isBad() landmarks differ"
    ca, cb = canon(a), canon(b)
    if ca == "aborted" or cb == "aborted":
        assert ca == cb
        return a
    pa, la, ra = ca
    pb_, lb, rb = cb
    assert {k: v[0] for k, v in pa.items()} == {k: v[0] for k, v in pb_.items()}, "keyframes / constness differ"
    for k in pa:   # tol > 0: the two maps went through two (differently ordered, equally valid) solves
        assert np.allclose(pa[k][1], pb_[k][1], rtol=0, atol=tol), "poses differ"
    assert ra == rb, "residual blocks differ"
    assert sorted(la) == sorted(lb), "landmark sets differ"
    for l in la:
        va, aa, ua = la[l]
        vb, ab, ub = lb[l]
        assert aa == ab and ua == ub
        assert np.allclose(va, vb, rtol=max(1e-12, tol), atol=tol), (l, va, vb)   # 1/z: R'(p - t) here, (Tcw * p).z there
    return a


@pytest.mark.parametrize("inv_depth", [True, False])
@pytest.mark.parametrize("n_kf,n_lm,nmin", [(8, 300, 25), (30, 4000, 25), (50, 10000, 25), (30, 3000, 200), (12, 500, 10 ** 6)])
def test_device_setup_equals_hash_map_walk(ctx, inv_depth, n_kf, n_lm, nmin):
    P = synth_ba.make_window(n_kf, n_lm, inv_depth=inv_depth, seed=n_kf + n_lm, max_obs=7)
    hm = host_map.HostMap(P, nmin_covscore=nmin)
    hm.attach_device(ctx)
    a = assert_same(hm)
    if nmin == 10 ** 6:
        assert a["aborted"]
    elif n_kf >= 30:
        assert 0 < int(np.sum(a["pose_const"])) and len(a["pose_kfid"]) <= n_kf   # a real window: constants present


@pytest.mark.parametrize("inv_depth", [True, False])
def test_gauge_fix_in_mono_mode(ctx, inv_depth):
    """mono: two constant keyframes are required (src/optimizer.cpp:65-68); only kfid 0 is constant by the walk, so the
    smallest optimised kfid is fixed as well (:394-407) -- on both sides"""
    P = synth_ba.make_window(8, 300, inv_depth=inv_depth, seed=4)
    hm = host_map.HostMap(P, stereo=False)
    hm.attach_device(ctx)
    a = assert_same(hm)
    assert {int(k) for k, c in zip(a["pose_kfid"], a["pose_const"]) if c} == {0, 1}


@pytest.mark.parametrize("inv_depth", [True, False])
def test_incremental_edits_reach_the_device(ctx, inv_depth):
    P = synth_ba.make_window(20, 2000, inv_depth=inv_depth, seed=77, max_obs=6)
    hw, hd = host_map.HostMap(P), host_map.HostMap(P)
    hd.attach_device(ctx)
    a = assert_same(hw, hd)
    rng = np.random.default_rng(5)
    # drop observations, whole landmarks, clear isobs_ on a third, and bring a few landmarks down to ONE observer that
    # the current frame does not see: MapPoint::isBad()
    pick = rng.choice(len(a["res_kfid"]), 400, replace=False)
    gone = set(int(l) for l in rng.choice(a["lm_lmid"], 60, replace=False))
    lonely = [int(l) for l in a["lm_lmid"][1::50] if int(l) not in gone][:12]
    for h in (hw, hd):
        for i in pick:
            h.remove_obs(a["res_kfid"][i], a["res_lmid"][i])
        for l in gone:
            h.remove_landmark(l)
        for l in a["lm_lmid"][::3]:
            if int(l) not in gone:
                h.set_isobs(l, 0)
        for l in lonely:
            ks = sorted({int(k) for k, ll in zip(a["res_kfid"], a["res_lmid"]) if int(ll) == l} |
                        ({int(a["lm_anchor_kfid"][list(a["lm_lmid"]).index(l)])} if inv_depth else set()))
            for k in ks[:-1]:
                h.remove_obs(k, l)
            h.set_isobs(l, 0)
    b = assert_same(hw, hd)
    assert len(b["res_type"]) < len(a["res_type"]) and len(b["lm_lmid"]) < len(a["lm_lmid"])
    assert len(hd.bad_lmids()) > 0
    # and once more: the cleared is3d_ flags went to the device with the next flush
    assert_same(hw, hd)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_local_ba_through_the_mirror_equals_the_walk(ctx, inv_depth):
    """Estimator::applyLocalBA with the device set-up == with the host walk; its update stage (poses, landmarks,
    removed observations, demoted stereo keypoints) reaches the mirror, so the NEXT set-up agrees too."""
    P = synth_ba.make_window(12, 800, inv_depth=inv_depth, seed=31)
    h0, h1 = host_map.HostMap(P), host_map.HostMap(P)
    h1.attach_device(ctx)
    r0, r1 = h0.apply_local_ba(ctx), h1.apply_local_ba(ctx)
    assert r0[0] == 0 and r1[0] == 0
    assert r0[1:3] == r1[1:3]
    assert r1[3] == pytest.approx(r0[3], rel=1e-9)
    for k in range(len(P.pose)):
        assert np.allclose(h0.pose(k), h1.pose(k), atol=1e-9)
    assert_same(h0, h1, tol=1e-8)


def test_growth_and_argument_errors(ctx):
    L = ctx.lib
    m = C.c_void_p()
    assert L.ov2_map_create(ctx.h, 4, 8, 16, C.byref(m)) == 0
    T = np.array([0, 0, 0, 0, 0, 0, 1.0])
    lm = np.arange(10, dtype=np.int32)
    uv = np.zeros((10, 2))
    vp = lambda a: a.ctypes.data
    neg = np.array([-1], np.int32)
    assert L.ov2_map_add_keyframe(m, -1, vp(T), 0, None, None, None, None, None) != 0
    assert L.ov2_map_add_keyframe(m, 1, vp(T), 1, vp(neg), vp(uv), None, None, None) != 0
    # ids and counts beyond the initial capacities grow the tables
    assert L.ov2_map_add_keyframe(m, 7, vp(T), 0, None, None, None, None, None) == 0
    for k in (1, 2, 3):
        assert L.ov2_map_add_keyframe(m, k, vp(T), 10, vp(lm), vp(uv), None, None, None) == 0
    out = (C.c_byte * 256)()
    assert L.ov2_map_local_ba_setup(m, 900, 25, 1, 1, None, out) != 0      # unknown keyframe
    # no landmark is alive yet: nb3dkps = 0 < nmin_covscore -> aborted, not an error
    assert L.ov2_map_local_ba_setup(m, 2, 25, 1, 1, None, out) == 0
    assert np.frombuffer(out, np.int32, 1)[0] == 1
    L.ov2_map_destroy(m)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_tables_grow_from_tiny_capacities(ctx, inv_depth):
    P = synth_ba.make_window(14, 900, inv_depth=inv_depth, seed=12, max_obs=6)
    hm = host_map.HostMap(P)
    hm.attach_device(ctx, max_kf=2, max_lm=3, max_obs=5)     # everything has to grow, several times
    a = assert_same(hm)
    assert len(a["res_type"]) > 5000


@pytest.mark.parametrize("inv_depth", [True, False])
def test_observation_table_is_squeezed_when_mostly_dead(ctx, inv_depth):
    """removals leave tombstones; once fewer than half of the rows are live the set-up squeezes the table (stable), and
    every later set-up still equals the hash-map walk; further removals and an explicit ov2_map_compact likewise"""
    P = synth_ba.make_window(40, 6000, inv_depth=inv_depth, seed=91, max_obs=7)
    hm = host_map.HostMap(P)
    hm.attach_device(ctx)
    rows0, cap0, n0 = hm.device_rows()
    assert rows0 > 4096 and n0 == 0
    rng = np.random.default_rng(5)
    for l in rng.permutation(len(P.lm))[:int(0.7 * len(P.lm))]:
        hm.remove_landmark(int(l))
    assert_same(hm)                              # reads the table with its tombstones, then squeezes it
    rows1, cap1, n1 = hm.device_rows()
    assert n1 == 1 and rows1 < 0.5 * rows0 and cap1 == cap0
    assert_same(hm)                              # the squeezed table gives the same problem
    assert hm.device_rows() == (rows1, cap1, 1)  # nothing dead any more: no second squeeze
    # single observations go too; the next squeeze happens only when half of the rows are dead again
    a = hm.setup_local_ba(dev=False)
    pairs = sorted({(int(k), int(l)) for k, l in zip(a["res_kfid"], a["res_lmid"])})
    for k, l in pairs[::3]:
        hm.remove_obs(k, l)
    assert_same(hm)
    assert hm.device_rows()[2] == 1


from ov2slam_amd.device_map import SetupC as _SetupC   # noqa: E402  (ov2_local_ba_setup)


def test_long_sequence_keeps_the_observation_table_bounded(ctx):
    """3000 keyframes through the raw hooks, a sliding window of 40 of them alive (older keyframes and their landmarks are
    removed as MapManager::removeKeyframe / removeMapPoint would): the table is squeezed again and again, stays within a
    small multiple of the live observations instead of growing with the sequence, and the last set-up equals the set-up of
    a fresh map holding only the live window."""
    L = ctx.lib
    vp = lambda a: a.ctypes.data
    rng = np.random.default_rng(3)
    NKF, WIN, PER, NEW = 3000, 40, 120, 12          # keyframes, live window, keypoints per keyframe, new landmarks per keyframe

    def frame(k):
        lm = np.arange(max(0, NEW * k - (PER - NEW)), NEW * (k + 1), dtype=np.int32)[-PER:]   # the newest PER landmark ids
        T = np.array([0.1 * k, 0, 0, 0, 0, 0, 1.0])
        uv = np.ascontiguousarray(rng.uniform(20, 700, (len(lm), 2)))
        return T, lm, uv

    def add(m, k):
        T, lm, uv = frame(k)
        xyz = np.ascontiguousarray(np.stack([0.1 * k + 0.01 * (lm % 7), 0.02 * (lm % 5), 3.0 + 0.001 * lm], 1))
        st = np.full(len(lm), 1 | 2 | 8, np.uint8)   # alive, 3D, keypoints 3D
        assert L.ov2_map_set_landmarks(m, len(lm), vp(lm), vp(xyz), vp(st)) == 0
        assert L.ov2_map_add_keyframe(m, k, vp(T), len(lm), vp(lm), vp(uv), None, None, None) == 0

    m = C.c_void_p()
    assert L.ov2_map_create(ctx.h, 64, 1024, 4096, C.byref(m)) == 0
    out = _SetupC()
    for k in range(NKF):
        add(m, k)
        if k >= WIN:
            old = k - WIN
            assert L.ov2_map_remove_keyframe(m, old) == 0
            gone = np.arange(NEW * old, NEW * (old + 1), dtype=np.int32) - (PER - NEW)   # landmarks only the removed keyframes saw
            gone = gone[gone >= 0]
            if len(gone):
                assert L.ov2_map_remove_landmarks(m, len(gone), vp(gone)) == 0
        if k % 25 == 24:   # the last of these is the set-up of the newest keyframe; a second call on the same map would see the
            assert L.ov2_map_local_ba_setup(m, k, 25, 1, 0, None, C.byref(out)) == 0   # is3d_ flags MapPoint::isBad() cleared in the first
    assert (NKF - 1) % 25 == 24
    rows, cap, nsq = C.c_int(), C.c_int(), C.c_int()
    assert L.ov2_map_obs_rows(m, C.byref(rows), C.byref(cap), C.byref(nsq)) == 0
    live = WIN * PER
    assert nsq.value >= 10 and rows.value <= 2.6 * live and cap.value <= 8 * live, (rows.value, cap.value, nsq.value)
    got = (out.aborted, out.n_pose, out.n_lm, out.n_res)
    # the same window in a fresh map
    f = C.c_void_p()
    assert L.ov2_map_create(ctx.h, NKF + 8, NEW * (NKF + 1) + 8, 8192, C.byref(f)) == 0
    for k in range(NKF - WIN, NKF):
        add(f, k)
    gone = np.arange(0, NEW * (NKF - WIN) - (PER - NEW), dtype=np.int32)
    ref = _SetupC()
    assert L.ov2_map_local_ba_setup(f, NKF - 1, 25, 1, 0, None, C.byref(ref)) == 0
    assert got == (ref.aborted, ref.n_pose, ref.n_lm, ref.n_res) and not out.aborted and out.n_res > 1000
    L.ov2_map_destroy(m)
    L.ov2_map_destroy(f)


# ---- batched, device-resident set-up (ov2_map_local_ba_setup_batch) and the device-side update stage ---------------
from ov2slam_amd import device_map as DM   # noqa: E402
from ov2slam_amd import local_ba   # noqa: E402


def _f32(P):
    """pixels rounded to float32, as Keypoint::unpx_ / runpx_ hold them (cv::Point2f)"""
    Q = P.copy()
    Q.res_uv = Q.res_uv.astype(np.float32).astype(np.float64)
    if Q.lm_anchor_uv is not None:
        Q.lm_anchor_uv = Q.lm_anchor_uv.astype(np.float32).astype(np.float64)
    return Q


def _single_setup(ctx, m, inv, nmin=25):
    """ov2_map_local_ba_setup (one map, host form) as a dict keyed like device_map.fetch_view"""
    out = DM.SetupC()
    K = np.ascontiguousarray(synth_ba.K_L)
    assert ctx.lib.ov2_map_local_ba_setup(m.h, m.newkf, nmin, 1, int(inv), K.ctypes.data, C.byref(out)) == 0
    if out.aborted:
        return dict(aborted=True)
    e = 1 if inv else 3

    def arr(ptr, shape, dt):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n,)).reshape(shape).copy() if n else np.zeros(shape, dt)
    P, NL, R, NB = out.n_pose, out.n_lm, out.n_res, out.n_bad
    kfid, lmid = arr(out.pose_kfid, (P,), np.int32), arr(out.lm_lmid, (NL,), np.int32)
    anch = arr(out.lm_anchor_pose, (NL,), np.int32)
    return dict(aborted=False, pose_kfid=kfid, pose_const=arr(out.pose_const, (P,), np.uint8), pose=arr(out.pose, (P, 7), np.float64),
                lm_lmid=lmid, lm=arr(out.lm, (NL, e), np.float64), lm_anchor_kfid=np.where(anch >= 0, kfid[np.maximum(anch, 0)], -1),
                lm_anchor_uv=arr(out.lm_anchor_uv, (NL, 2), np.float64), res_type=arr(out.res_type, (R,), np.uint8),
                res_kfid=kfid[arr(out.res_pose, (R,), np.int32)], res_lmid=lmid[arr(out.res_lm, (R,), np.int32)],
                res_uv=arr(out.res_uv, (R, 2), np.float64), res_sigma=arr(out.res_sigma, (R,), np.float64),
                bad_lmid=np.sort(arr(out.bad_lmid, (NB,), np.int32)))


@pytest.mark.parametrize("inv_depth", [True, False])
def test_batched_setup_equals_single_calls(ctx, inv_depth):
    """B maps of different sizes in one sync-free chain == B single calls, array for array (same kernels, map = blockIdx.y);
    one of the maps aborts (too few 3D keypoints), one holds isBad() landmarks"""
    shapes = [(8, 300), (30, 4000), (14, 900), (50, 10000), (3, 20), (20, 2000)]
    probs = [synth_ba.make_window(k, l, inv_depth=inv_depth, seed=100 + k + l, max_obs=7) for k, l in shapes]
    a = [DM.DeviceMap.from_problem(ctx, P, isobs="newest") for P in probs]
    b = [DM.DeviceMap.from_problem(ctx, P, isobs="newest") for P in probs]
    for P, ma, mb in zip(probs, a, b):   # a few landmarks down to ONE observer the current frame does not see: isBad()
        kf, lm, _, _, _ = DM.observations_of(P)
        cnt = np.bincount(lm, minlength=len(P.lm))
        seen_now = np.zeros(len(P.lm), bool)
        seen_now[lm[kf == len(P.pose) - 1]] = True
        lonely = np.flatnonzero((cnt == 2) & ~seen_now)[:8]
        first = np.array([np.flatnonzero(lm == l)[0] for l in lonely], np.int64)
        if len(first):
            for m in (ma, mb):
                m.remove_obs(kf[first], lm[first])
    views = DM.setup_batch(ctx, a, inv_depth=inv_depth, calib_l=synth_ba.K_L)
    n_bad = 0
    for k, (ma, mb, v) in enumerate(zip(a, b, views)):
        got, ref = DM.fetch_view(ctx, v, inv_depth), _single_setup(ctx, mb, inv_depth)
        assert got["aborted"] == ref["aborted"], k
        if ref["aborted"]:
            assert shapes[k] == (3, 20)
            continue
        for key in ref:
            if key != "aborted":
                assert np.array_equal(got[key], ref[key]), (k, key)
        assert not got["outlier"].any()
        n_bad += len(ref["bad_lmid"])
        assert len(ref["res_type"]) > 100
    assert n_bad > 0, "no isBad() landmark in any map: the bad list went untested"
    # a second batched call on the same maps (scratch re-zeroed, blocks reused) gives the same problems again, minus the
    # landmarks isBad() has meanwhile demoted (is3d_ cleared: src/map_point.cpp:219)
    views2 = DM.setup_batch(ctx, a, inv_depth=inv_depth, calib_l=synth_ba.K_L)
    for v1, v2 in zip(views, views2):
        assert (v1.aborted, v1.n_pose) == (v2.aborted, v2.n_pose) and v1.n_res <= v2.n_res <= v1.n_res + 2 * v1.n_bad


def _solve_on_device(ctx, maps, views, proto, inv_depth):
    pcs, rcs = DM.problems_of(views, proto, inv_depth)
    o = local_ba.default_options()
    st = ctx.lib.ov2_ba_solve_batch_dev(ctx.h, len(maps), pcs, C.byref(o), rcs)
    assert st == 0, ctx.lib.ov2_last_error(ctx.h)
    return rcs


@pytest.mark.parametrize("inv_depth", [True, False])
def test_device_update_equals_host_update(ctx, inv_depth):
    """set-up -> solve -> update entirely on the device tables == Estimator::applyLocalBA of the C++ host mirror (hash-map
    objects, updateAfterLocalBA = src/optimizer.cpp:741-882) followed by the flush of its edits: same poses, same
    surviving landmarks with the same points and states, same surviving observations with the same stereo flags"""
    P = _f32(synth_ba.make_window(14, 1200, inv_depth=inv_depth, seed=41, outlier_frac=0.08))
    hm = host_map.HostMap(P)
    hm.attach_device(ctx)
    dm = DM.DeviceMap.from_problem(ctx, P, isobs="all")
    # landmarks the current frame does not see any more (isobs_ = false) can be culled: every third one, on both sides
    off = np.arange(0, len(P.lm), 3, dtype=np.int32)
    for l in off:
        hm.set_isobs(int(l), 0)
    dm.set_landmarks(off, None, np.full(len(off), DM.LM_ALIVE | DM.LM_3D | DM.LM_KP3D, np.uint8))
    hm.flush_device()
    start = DM.canonical_state(dm.download())
    _assert_states_close(DM.canonical_state(DM.DeviceMap.download(_Handle(ctx, hm.device_handle()))), start, 1e-13)

    r = hm.apply_local_ba(ctx)
    assert r[0] == 0 and r[1] > 0
    hm.flush_device()
    ref = DM.canonical_state(DM.DeviceMap.download(_Handle(ctx, hm.device_handle())))

    views = DM.setup_batch(ctx, [dm], inv_depth=inv_depth, calib_l=P.calib_l)
    rcs = _solve_on_device(ctx, [dm], views, P, inv_depth)
    assert (rcs[0].n_outliers_pass1, rcs[0].n_outliers_pass2) == (r[1], r[2])
    upd = DM.update_batch(ctx, [dm], views, cur_kfid=[dm.newkf])[0]
    got = DM.canonical_state(dm.download())

    _assert_states_close(ref, got, 1e-9)
    kf_g, lm_g, ob_g = got
    # what the device reports for replay is exactly the difference between the two states
    kf_s, lm_s, ob_s = start
    assert set(upd["removed_lmid"].tolist()) == set(lm_s) - set(lm_g)
    assert len(upd["removed_lmid"]) > 0 and len(upd["removed_obs"]) > 0 and len(upd["stereo_off"]) > 0
    gone = {(int(k), int(l)) for k, l in upd["removed_obs"]}   # (some of their landmarks were removed as well afterwards)
    assert gone >= {o for o in ob_s if o not in ob_g and o[1] in lm_g} and not (gone & set(ob_g)) and gone <= set(ob_s)
    demoted = {(int(k), int(l)) for k, l in upd["stereo_off"]}
    assert {o for o in ob_g if ob_s[o] and not ob_g[o]} <= demoted


def _assert_states_close(ref, got, tol):
    kf_r, lm_r, ob_r = ref
    kf_g, lm_g, ob_g = got
    assert sorted(kf_r) == sorted(kf_g)
    for k in kf_r:
        assert np.allclose(kf_r[k], kf_g[k], rtol=0, atol=tol), k
    assert sorted(lm_r) == sorted(lm_g), "surviving landmarks differ"
    for l in lm_r:
        assert lm_r[l][1] == lm_g[l][1], (l, lm_r[l][1], lm_g[l][1])
        assert np.allclose(lm_r[l][0], lm_g[l][0], rtol=tol, atol=tol), l
    assert ob_r == ob_g, "surviving observations / stereo flags differ"


class _Handle:
    """a borrowed ov2_map* with DeviceMap's download()"""

    def __init__(self, ctx, h):
        self.ctx, self.L, self.h = ctx, ctx.lib, h
    _p = staticmethod(DM.DeviceMap._p)


def test_batch_of_distinct_windows_setup_solve_update_and_restore(ctx):
    """the keyframe job of bench.py: B distinct maps -> batched set-up -> ov2_ba_solve_batch_dev on the device views ->
    device update; every window's solve is bitwise the solve of the same flat problem alone, the update changes the
    tables, ov2_map_restore_state_batch brings every table back, and the next job repeats the first bit for bit"""
    shapes = [(10, 500, 0.02, 0.1), (16, 1500, 0.10, 1.0), (12, 900, 0.05, 4.0), (7, 260, 0.15, 1.0)]
    probs = [synth_ba.make_window(k, l, inv_depth=True, seed=7 * k + l, outlier_frac=f, pose_noise=(0.02 * g, np.deg2rad(0.5) * g))
             for k, l, f, g in shapes]
    maps = [DM.DeviceMap.from_problem(ctx, P, isobs="newest") for P in probs]
    for m in maps:
        m.save_state()
    before = [DM.canonical_state(m.download()) for m in maps]
    views = DM.setup_batch(ctx, maps, calib_l=synth_ba.K_L)
    flat = [DM.fetch_view(ctx, v, True) for v in views]
    rcs = _solve_on_device(ctx, maps, views, probs[0], True)
    logs = [[(i.cost, i.radius, i.step_is_successful) for i in r.log[:r.n_log]] for r in rcs]
    solved = [DM.fetch_view(ctx, v, True) for v in views]
    for k, (f, s) in enumerate(zip(flat, solved)):
        # the same flat problem through the host form, alone
        q = synth_ba.BaProblem(synth_ba.K_L, synth_ba.K_R, probs[0].T_rl, 1, f["pose"], f["pose_const"], f["lm"],
                               np.searchsorted(f["pose_kfid"], f["lm_anchor_kfid"]).astype(np.int32), f["lm_anchor_uv"], f["res_type"],
                               np.searchsorted(f["pose_kfid"], f["res_kfid"]).astype(np.int32),
                               np.searchsorted(f["lm_lmid"], f["res_lmid"]).astype(np.int32), f["res_uv"], f["res_sigma"])
        r1 = local_ba.Optimizer(ctx).localBA(q)
        assert np.array_equal(q.pose.view(np.uint64), s["pose"].view(np.uint64)), k
        assert np.array_equal(q.lm.view(np.uint64), s["lm"].view(np.uint64)), k
        assert np.array_equal(r1.outlier, s["outlier"]), k
    DM.update_batch(ctx, maps, views, want_lists=False)
    after = [DM.canonical_state(m.download()) for m in maps]
    assert all(a != b for a, b in zip(after, before))
    DM.restore_state_batch(ctx, maps)
    assert [DM.canonical_state(m.download()) for m in maps] == before
    views = DM.setup_batch(ctx, maps, calib_l=synth_ba.K_L)
    rcs2 = _solve_on_device(ctx, maps, views, probs[0], True)
    assert [[(i.cost, i.radius, i.step_is_successful) for i in r.log[:r.n_log]] for r in rcs2] == logs


def test_batch_with_aborted_and_empty_maps(ctx):
    """edge cases of the batched keyframe job: a map that aborts (too few 3D keypoints in the new keyframe, src/optimizer.cpp:61-63)
    and an EMPTY map (no keyframe, no observation) ride in the same batch as two real windows -- their windows reach the
    solver with no blocks, the update leaves their tables untouched, and the real windows come out exactly as they do in a
    batch of their own; B = 0 is a no-op; a map twice in a batch / an update without a set-up are refused"""
    shapes = [(10, 500), (3, 20), (14, 900)]
    probs = [synth_ba.make_window(k, l, inv_depth=True, seed=11 * k + l, outlier_frac=0.05) for k, l in shapes]
    maps = [DM.DeviceMap.from_problem(ctx, P, isobs="newest") for P in probs]
    empty = DM.DeviceMap(ctx, 4, 4, 4)
    empty.newkf = 0
    maps.insert(2, empty)
    alone = [DM.DeviceMap.from_problem(ctx, probs[i], isobs="newest") for i in (0, 2)]
    before = [DM.canonical_state(m.download()) for m in maps]

    L = ctx.lib
    assert L.ov2_map_local_ba_setup_batch(ctx.h, 0, None, None, 25, 1, 1, None, None) == 0
    assert L.ov2_map_local_ba_update_batch(ctx.h, 0, None, None, None, None) == 0
    with pytest.raises(Exception):   # no set-up ran on these maps yet
        DM.update_batch(ctx, maps, (DM.SetupC * len(maps))(), want_lists=False)
    with pytest.raises(Exception):
        DM.setup_batch(ctx, [maps[0], maps[0]], calib_l=synth_ba.K_L)

    views = DM.setup_batch(ctx, maps, calib_l=synth_ba.K_L)
    assert [bool(v.aborted) for v in views] == [False, True, True, False]
    rcs = _solve_on_device(ctx, maps, views, probs[0], True)
    assert rcs[1].n_log == 0 and rcs[2].n_log == 0 and rcs[0].n_log > 1 and rcs[3].n_log > 1
    upd = DM.update_batch(ctx, maps, views, cur_kfid=[m.newkf for m in maps])
    after = [DM.canonical_state(m.download()) for m in maps]
    assert after[1] == before[1] and after[2] == before[2] and after[2] == ({}, {}, {})
    assert all(len(upd[i][k]) == 0 for i in (1, 2) for k in upd[i])
    assert after[0] != before[0] and after[3] != before[3]

    v2 = DM.setup_batch(ctx, alone, calib_l=synth_ba.K_L)
    _solve_on_device(ctx, alone, v2, probs[0], True)
    DM.update_batch(ctx, alone, v2, cur_kfid=[m.newkf for m in alone], want_lists=False)
    assert DM.canonical_state(alone[0].download()) == after[0]
    assert DM.canonical_state(alone[1].download()) == after[3]
