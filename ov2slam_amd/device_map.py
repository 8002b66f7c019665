"""ctypes mirror of the device map mirror's raw hooks (ov2_map_*, csrc/map.hip) for maps that live ONLY on the device:
no C++ Frame / MapPoint graph beside them.  Used by bench.py (one map per sequence: set-up -> solve -> update of
Optimizer::localBA without the host in the data path, reference src/optimizer.cpp:43-430, 439-735, 741-882) and by the
tests of the batched set-up / update stages.  Plumbing only."""
import ctypes as C

import numpy as np

from . import ba_types as T
from . import synth_ba
from .ba_types import BaProblemC, BaResultC, dp, i32p, u8p
from .frontend import _check

LM_ALIVE, LM_3D, LM_OBS, LM_KP3D = 1, 2, 4, 8
OBS_ALIVE, OBS_STEREO = 1, 2


class SetupC(C.Structure):
    """ov2_local_ba_setup"""
    _fields_ = [("aborted", C.c_int32), ("n_pose", C.c_int32), ("n_lm", C.c_int32), ("n_res", C.c_int32), ("n_bad", C.c_int32),
                ("pose_kfid", C.c_void_p), ("pose_const", C.c_void_p), ("pose", C.c_void_p), ("lm_lmid", C.c_void_p),
                ("lm", C.c_void_p), ("lm_anchor_pose", C.c_void_p), ("lm_anchor_uv", C.c_void_p), ("res_type", C.c_void_p),
                ("res_pose", C.c_void_p), ("res_lm", C.c_void_p), ("res_uv", C.c_void_p), ("res_sigma", C.c_void_p),
                ("bad_lmid", C.c_void_p), ("res_outlier", C.c_void_p)]


class UpdateC(C.Structure):
    """ov2_local_ba_update"""
    _fields_ = [("n_removed_lm", C.c_int32), ("n_removed_obs", C.c_int32), ("n_stereo_off", C.c_int32),
                ("removed_lmid", C.c_void_p), ("removed_obs", C.c_void_p), ("stereo_off", C.c_void_p)]


def observations_of(prob):
    """the (keyframe, landmark) observations behind a flat BaProblem, as Optimizer::localBA read them off the map:
    returns kf, lm (int32), unpx (n x 2), stereo (uint8), runpx (n x 2), sorted by (kf, lm)"""
    rt, rp, rl = np.asarray(prob.res_type), np.asarray(prob.res_pose, np.int64), np.asarray(prob.res_lm, np.int64)
    uv = np.asarray(prob.res_uv, np.float64).reshape(-1, 2)
    nl = len(prob.lm)
    left = (rt == T.L_XYZ) | (rt == T.L_INV)
    k_of = rp.copy()
    if prob.inv_depth:
        ranch = rt == T.RANCH_INV
        k_of[ranch] = np.asarray(prob.lm_anchor_pose, np.int64)[rl[ranch]]
    nk = len(prob.pose)
    key = k_of * nl + rl
    # left observations: the left blocks + (inverse depth) the anchor observation of every landmark
    lk, luv = key[left], uv[left]
    if prob.inv_depth:
        ak = np.asarray(prob.lm_anchor_pose, np.int64) * nl + np.arange(nl)
        lk = np.concatenate([lk, ak]); luv = np.concatenate([luv, np.asarray(prob.lm_anchor_uv, np.float64).reshape(-1, 2)])
    order = np.argsort(lk, kind="stable")
    lk, luv = lk[order], luv[order]
    assert len(np.unique(lk)) == len(lk) and nk * nl < 2 ** 62
    rk, ruv = key[~left], uv[~left]
    pos = np.searchsorted(lk, rk)
    assert np.all(lk[pos] == rk), "a right-camera block without its left observation"
    stereo = np.zeros(len(lk), np.uint8); stereo[pos] = 1
    run = np.zeros((len(lk), 2)); run[pos] = ruv
    return (lk // nl).astype(np.int32), (lk % nl).astype(np.int32), luv, stereo, run


def world_points_of(prob):
    """initial world points of the landmarks (what MapPoint::ptxyz_ holds before the local BA)"""
    if not prob.inv_depth:
        return np.asarray(prob.lm, np.float64).reshape(-1, 3).copy()
    a = np.asarray(prob.lm_anchor_pose, np.int64)
    z = 1.0 / np.asarray(prob.lm, np.float64).reshape(-1)
    u = np.asarray(prob.lm_anchor_uv, np.float64).reshape(-1, 2)
    K = prob.calib_l
    pc = np.stack([z * (u[:, 0] - K[2]) / K[0], z * (u[:, 1] - K[3]) / K[1], z], 1)
    out = np.zeros((len(z), 3))
    for k in np.unique(a):
        R = synth_ba.quat_to_rot(prob.pose[k, 3:])
        sel = a == k
        out[sel] = pc[sel] @ R.T + prob.pose[k, :3]
    return out


class DeviceMap:
    """one ov2_map filled through the raw hooks"""

    def __init__(self, ctx, max_kf, max_lm, max_obs):
        self.ctx, self.L = ctx, ctx.lib
        h = C.c_void_p()
        _check(ctx.h, self.L.ov2_map_create(ctx.h, int(max_kf), int(max_lm), int(max_obs), C.byref(h)))
        self.h = h
        self.newkf = -1

    @classmethod
    def from_problem(cls, ctx, prob, isobs="newest"):
        """the map a flat BaProblem was read from: its keyframes, landmarks (at their initial world points) and
        observations.  isobs: 'all' (every MapPoint::isobs_ set, as tests/HostMap builds them), 'newest' (only the landmarks
        the newest keyframe observes: what a live sequence looks like) or 'none'."""
        kf, lm, un, st, run = observations_of(prob)
        nk, nl = len(prob.pose), len(prob.lm)
        m = cls(ctx, nk + 8, nl + 8, len(kf) + 64)
        m.prob, m.newkf = prob, nk - 1
        xyz = world_points_of(prob)
        state = np.full(nl, LM_ALIVE | LM_3D | LM_KP3D, np.uint8)
        if isobs == "all":
            state |= LM_OBS
        elif isobs == "newest":
            state[lm[kf == nk - 1]] |= LM_OBS
        m.set_landmarks(np.arange(nl, dtype=np.int32), xyz, state)
        cut = np.searchsorted(kf, np.arange(nk + 1))
        for k in range(nk):
            a, b = cut[k], cut[k + 1]
            m.add_keyframe(k, prob.pose[k], lm[a:b], un[a:b], run[a:b], st[a:b])
        return m

    def close(self):
        if getattr(self, "h", None):
            self.L.ov2_map_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def add_keyframe(self, kfid, Twc, lmid, unpx, runpx=None, is_stereo=None, scale=None):
        Twc = np.ascontiguousarray(Twc, np.float64)
        lmid = np.ascontiguousarray(lmid, np.int32)
        unpx = np.ascontiguousarray(unpx, np.float64)
        runpx = None if runpx is None else np.ascontiguousarray(runpx, np.float64)
        is_stereo = None if is_stereo is None else np.ascontiguousarray(is_stereo, np.uint8)
        scale = None if scale is None else np.ascontiguousarray(scale, np.int32)
        _check(self.ctx.h, self.L.ov2_map_add_keyframe(self.h, int(kfid), self._p(Twc), len(lmid), self._p(lmid), self._p(unpx),
                                                       self._p(runpx), self._p(is_stereo), self._p(scale)))

    def set_landmarks(self, lmid, xyz, state):
        lmid = np.ascontiguousarray(lmid, np.int32)
        xyz = None if xyz is None else np.ascontiguousarray(xyz, np.float64)
        state = np.ascontiguousarray(state, np.uint8)
        _check(self.ctx.h, self.L.ov2_map_set_landmarks(self.h, len(lmid), self._p(lmid), self._p(xyz), self._p(state)))

    def remove_obs(self, kfid, lmid):
        k, l = np.ascontiguousarray(kfid, np.int32), np.ascontiguousarray(lmid, np.int32)
        _check(self.ctx.h, self.L.ov2_map_remove_obs(self.h, len(k), self._p(k), self._p(l)))

    def save_state(self):
        _check(self.ctx.h, self.L.ov2_map_save_state(self.h))

    def download(self):
        """dict of the tables (host copies)"""
        nk, nl, no = C.c_int(), C.c_int(), C.c_int()
        _check(self.ctx.h, self.L.ov2_map_download(self.h, C.byref(nk), C.byref(nl), C.byref(no), *([None] * 9)))
        K, L, N = nk.value, nl.value, no.value
        d = dict(kf_pose=np.zeros((K, 7)), kf_state=np.zeros(K, np.uint8), lm_xyz=np.zeros((L, 3)), lm_state=np.zeros(L, np.uint8),
                 obs_kf=np.zeros(N, np.int32), obs_lm=np.zeros(N, np.int32), obs_flag=np.zeros(N, np.uint8), obs_uv=np.zeros((N, 2)),
                 obs_ruv=np.zeros((N, 2)))
        _check(self.ctx.h, self.L.ov2_map_download(self.h, None, None, None, *[self._p(d[k]) for k in (
            "kf_pose", "kf_state", "lm_xyz", "lm_state", "obs_kf", "obs_lm", "obs_flag", "obs_uv", "obs_ruv")]))
        return d


def canonical_state(d):
    """a downloaded map (DeviceMap.download) keyed by ids: poses of the live keyframes, (point, state) of the live landmarks,
    the set of live observations with their stereo flag"""
    kfs = {int(k): tuple(d["kf_pose"][k]) for k in np.flatnonzero(d["kf_state"])}
    lms = {int(l): (tuple(d["lm_xyz"][l]), int(d["lm_state"][l])) for l in np.flatnonzero(d["lm_state"] & LM_ALIVE)}
    live = (d["obs_flag"] & OBS_ALIVE).astype(bool)
    if len(live):
        live &= d["kf_state"][d["obs_kf"]].astype(bool) & (d["lm_state"][d["obs_lm"]] & LM_ALIVE).astype(bool)
    obs = {(int(k), int(l)): int(f & OBS_STEREO) for k, l, f in zip(d["obs_kf"][live], d["obs_lm"][live], d["obs_flag"][live])}
    return kfs, lms, obs


def restore_state_batch(ctx, maps):
    hs = (C.c_void_p * len(maps))(*[m.h for m in maps])
    _check(ctx.h, ctx.lib.ov2_map_restore_state_batch(ctx.h, len(maps), hs))


def setup_batch(ctx, maps, newkf=None, nmin_covscore=25, nmin_cst_kfs=1, inv_depth=True, calib_l=None):
    """ov2_map_local_ba_setup_batch: returns the array of SetupC device views (one synchronisation for all maps)"""
    B = len(maps)
    hs = (C.c_void_p * B)(*[m.h for m in maps])
    nk = np.ascontiguousarray([m.newkf for m in maps] if newkf is None else newkf, np.int32)
    K = None if calib_l is None else np.ascontiguousarray(np.broadcast_to(np.asarray(calib_l, np.float64), (B, 4)))
    out = (SetupC * B)()
    _check(ctx.h, ctx.lib.ov2_map_local_ba_setup_batch(ctx.h, B, hs, nk.ctypes.data_as(C.c_void_p), int(nmin_covscore),
                                                       int(nmin_cst_kfs), int(bool(inv_depth)),
                                                       None if K is None else K.ctypes.data_as(C.c_void_p), out))
    return out


def problems_of(views, proto, inv_depth):
    """ov2_ba_problem[B] (+ ov2_ba_result[B] with the outlier flags inside the maps' blocks) over the device views; proto: a
    BaProblem carrying the calibrations / extrinsic shared by the maps"""
    B = len(views)
    pcs, rcs = (BaProblemC * B)(), (BaResultC * B)()
    for b, v in enumerate(views):
        pc = proto.as_c()
        pc.inv_depth = int(bool(inv_depth))
        pc.n_pose, pc.n_lm, pc.n_res = (0, 0, 0) if v.aborted else (v.n_pose, v.n_lm, v.n_res)
        pc.pose, pc.pose_const, pc.lm = C.cast(v.pose, dp), C.cast(v.pose_const, u8p), C.cast(v.lm, dp)
        pc.lm_anchor_pose, pc.lm_anchor_uv = C.cast(v.lm_anchor_pose, i32p), C.cast(v.lm_anchor_uv, dp)
        pc.res_type, pc.res_pose, pc.res_lm = C.cast(v.res_type, u8p), C.cast(v.res_pose, i32p), C.cast(v.res_lm, i32p)
        pc.res_uv, pc.res_sigma = C.cast(v.res_uv, dp), C.cast(v.res_sigma, dp)
        pcs[b] = pc
        rcs[b].outlier = C.cast(v.res_outlier, u8p)
    return pcs, rcs


def update_batch(ctx, maps, views, cur_kfid=None, want_lists=True):
    """ov2_map_local_ba_update_batch with the outlier flags the solve left in the maps' blocks; returns per map
    dict(removed_lmid, removed_obs (n x 2: kfid, lmid), stereo_off (n x 2)) or None (asynchronous)"""
    B = len(maps)
    hs = (C.c_void_p * B)(*[m.h for m in maps])
    outl = (C.c_void_p * B)(*[None if v.aborted else v.res_outlier for v in views])
    ck = None if cur_kfid is None else np.ascontiguousarray(cur_kfid, np.int32)
    out = (UpdateC * B)() if want_lists else None
    _check(ctx.h, ctx.lib.ov2_map_local_ba_update_batch(ctx.h, B, hs, outl, None if ck is None else ck.ctypes.data_as(C.c_void_p), out))
    if not want_lists:
        return None
    res = []
    for u in out:
        def fetch(ptr, n, cols):
            a = np.zeros((n, cols) if cols > 1 else (n,), np.int32)
            if n:
                _check(ctx.h, ctx.lib.ov2_memcpy_d2h(ctx.h, a.ctypes.data_as(C.c_void_p), ptr, a.nbytes))
            return a
        res.append(dict(removed_lmid=np.sort(fetch(u.removed_lmid, u.n_removed_lm, 1)),
                        removed_obs=fetch(u.removed_obs, u.n_removed_obs, 2), stereo_off=fetch(u.stereo_off, u.n_stereo_off, 2)))
    return res


def fetch_view(ctx, v, inv_depth):
    """the flat problem of a device view as host arrays keyed like HostMap.setup_local_ba (reference ids, not indices)"""
    def get(ptr, shape, dt):
        a = np.zeros(shape, dt)
        if a.nbytes:
            _check(ctx.h, ctx.lib.ov2_memcpy_d2h(ctx.h, a.ctypes.data_as(C.c_void_p), ptr, a.nbytes))
        return a
    if v.aborted:
        return dict(aborted=True)
    e = 1 if inv_depth else 3
    P, NL, R, NB = v.n_pose, v.n_lm, v.n_res, v.n_bad
    kfid, lmid = get(v.pose_kfid, P, np.int32), get(v.lm_lmid, NL, np.int32)
    anch = get(v.lm_anchor_pose, NL, np.int32)
    return dict(aborted=False, pose_kfid=kfid, pose_const=get(v.pose_const, P, np.uint8), pose=get(v.pose, (P, 7), np.float64),
                lm_lmid=lmid, lm=get(v.lm, (NL, e), np.float64), lm_anchor_kfid=np.where(anch >= 0, kfid[np.maximum(anch, 0)], -1),
                lm_anchor_uv=get(v.lm_anchor_uv, (NL, 2), np.float64), res_type=get(v.res_type, R, np.uint8),
                res_kfid=kfid[get(v.res_pose, R, np.int32)], res_lmid=lmid[get(v.res_lm, R, np.int32)],
                res_uv=get(v.res_uv, (R, 2), np.float64), res_sigma=get(v.res_sigma, R, np.float64),
                bad_lmid=np.sort(get(v.bad_lmid, NB, np.int32)), outlier=get(v.res_outlier, R, np.uint8))
