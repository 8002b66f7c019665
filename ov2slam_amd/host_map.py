"""ctypes driver of the C++ host mirror (libov2host.so, ov2slam_amd/host/): builds a Frame/MapPoint graph, runs
Optimizer::setupLocalBA on the CPU or Estimator::applyLocalBA on the GPU.  Used by tests and bring-up only."""
import ctypes as C
import os

import numpy as np

from . import ba_types as T
from . import synth_ba

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(_HERE, "lib", "libov2host.so"))
        dp, ip, u8 = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
        L.ov2h_map_create.restype = C.c_void_p
        L.ov2h_map_create.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_int, C.c_int]
        L.ov2h_map_destroy.argtypes = [C.c_void_p]
        L.ov2h_map_add_keyframe.argtypes = [C.c_void_p, C.c_int, dp]
        L.ov2h_map_add_landmark.argtypes = [C.c_void_p, C.c_int, dp, C.c_int]
        L.ov2h_map_add_obs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        L.ov2h_map_finalize.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_local_ba_setup.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.ov2h_map_attach_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ov2h_local_ba_setup_dev.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.ov2h_map_device_rows.argtypes = [C.c_void_p, ip, ip, ip]
        L.ov2h_map_device_handle.argtypes = [C.c_void_p]
        L.ov2h_map_device_handle.restype = C.c_void_p
        L.ov2h_map_flush_device.argtypes = [C.c_void_p]
        L.ov2h_set_distortion.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]
        L.ov2h_undistort.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.ov2h_project_dist.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_float)]
        L.ov2h_range_ba_setup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip]
        L.ov2h_full_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, ip, ip, C.POINTER(C.c_double), ip]
        L.ov2h_local_pose_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, dp, C.POINTER(C.c_double), ip]
        L.ov2h_full_pose_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, dp, C.POINTER(C.c_uint8), C.POINTER(C.c_double)]
        L.ov2h_loose_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, ip, C.POINTER(C.c_double)]
        L.ov2h_map_remove_obs.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ov2h_map_remove_landmark.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_map_set_isobs.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ov2h_map_bad_lmids.argtypes = [C.c_void_p, ip, C.c_int]
        L.ov2h_local_ba_get.argtypes = [C.c_void_p, ip, u8, dp, ip, dp, ip, dp, u8, ip, ip, dp]
        L.ov2h_apply_local_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, ip, ip, dp]
        L.ov2h_ba_worker_create.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int]
        L.ov2h_ba_worker_create.restype = C.c_void_p
        L.ov2h_ba_worker_set_device_resident.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_ba_worker_set_device_resident.restype = None
        L.ov2h_ba_worker_submit_all.argtypes = [C.c_void_p]
        L.ov2h_ba_worker_submit_all.restype = None
        L.ov2h_ba_worker_set_counting.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_ba_worker_set_counting.restype = None
        L.ov2h_ba_worker_stats.argtypes = [C.c_void_p, dp]
        L.ov2h_ba_worker_stats.restype = None
        L.ov2h_ba_worker_destroy.argtypes = [C.c_void_p]
        L.ov2h_ba_worker_destroy.restype = None
        L.ov2h_ba_pipeline_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, ip, dp, C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.ov2h_slam_set_brief.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.ov2h_slam_set_brief.restype = None
        L.ov2h_slam_kf_stats.argtypes = [C.c_void_p, dp]
        L.ov2h_slam_kf_stats.restype = None
        L.ov2h_slam_device_handle.argtypes = [C.c_void_p]
        L.ov2h_slam_device_handle.restype = C.c_void_p
        L.ov2h_slam_flush_device.argtypes = [C.c_void_p]
        L.ov2h_slam_check_map.argtypes = [C.c_void_p, ip]
        L.ov2h_slam_export_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, ip, dp, ip, dp, u8, ip, ip, u8]
        L.ov2h_slam_export_map.restype = None
        L.ov2h_mp_new.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ov2h_mp_new.restype = C.c_void_p
        L.ov2h_mp_add_obs.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_mp_add_obs.restype = None
        L.ov2h_mp_add_desc.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ov2h_mp_add_desc.restype = None
        L.ov2h_mp_remove_obs.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_mp_remove_obs.restype = None
        L.ov2h_mp_state.argtypes = [C.c_void_p, ip, u8, C.c_int, ip, C.POINTER(C.c_float)]
        L.ov2h_mp_free.argtypes = [C.c_void_p]
        L.ov2h_mp_free.restype = None
        L.ov2h_feloop_create.argtypes = [C.c_void_p, C.c_void_p]
        L.ov2h_feloop_create.restype = C.c_void_p
        L.ov2h_feloop_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, ip]
        L.ov2h_feloop_destroy.argtypes = [C.c_void_p]
        L.ov2h_feloop_destroy.restype = None
        L.ov2h_ba_pipeline_create.restype = C.c_void_p
        for fn in (L.ov2h_ba_pipeline_submit_all, L.ov2h_ba_pipeline_destroy):
            fn.argtypes, fn.restype = [C.c_void_p], None
        L.ov2h_ba_pipeline_set_counting.argtypes, L.ov2h_ba_pipeline_set_counting.restype = [C.c_void_p, C.c_int], None
        L.ov2h_ba_pipeline_stats.argtypes, L.ov2h_ba_pipeline_stats.restype = [C.c_void_p, dp], None
        L.ov2h_slam_create.argtypes = [C.c_void_p, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int]
        L.ov2h_slam_create.restype = C.c_void_p
        L.ov2h_slam_add_stereo.argtypes = [C.c_void_p, C.c_double, u8, u8, C.c_int, C.c_int]
        L.ov2h_slam_pose.argtypes, L.ov2h_slam_pose.restype = [C.c_void_p, dp], None
        L.ov2h_slam_stats.argtypes, L.ov2h_slam_stats.restype = [C.c_void_p, dp], None
        L.ov2h_slam_landmarks.argtypes = [C.c_void_p, C.c_int, ip, dp]
        L.ov2h_slam_destroy.argtypes, L.ov2h_slam_destroy.restype = [C.c_void_p], None
        L.ov2h_compute_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, ip]
        L.ov2h_get_pose.argtypes = [C.c_void_p, C.c_int, dp]
        L.ov2h_get_landmark.argtypes = [C.c_void_p, C.c_int, dp, ip]
        L.ov2h_count_keypoints.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.ov2h_structure_only_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, ip, dp, ip]
        fpp = C.POINTER(C.c_float)
        L.ov2h_map_add_kp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, dp]
        L.ov2h_frame_init_grid.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ov2h_map_forget_landmark.argtypes = [C.c_void_p, C.c_int]
        L.ov2h_set_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ov2h_stereo_matching.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u8, u8, C.c_int, C.c_int]
        L.ov2h_klt_tracking.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u8, u8, C.c_int, C.c_int, ip]
        L.ov2h_get_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, fpp, u8, u8, fpp]
        L.ov2h_get_frl.argtypes = [C.c_void_p, C.c_int, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HostMap:
    """a MapManager + keyframes + map points built from a flat BaProblem (inverse-depth or XYZ window)."""

    def __init__(self, prob, nmin_covscore=25, stereo=True):
        L = lib()
        self.prob = prob
        T_rl = np.eye(4)
        T_rl[:3, :3] = synth_ba.quat_to_rot(prob.T_rl[3:])
        T_rl[:3, 3] = prob.T_rl[:3]
        T_lr = np.linalg.inv(T_rl)
        t_lr7 = np.ascontiguousarray(synth_ba.pose7(T_lr[:3, :3], T_lr[:3, 3]))
        self.h = L.ov2h_map_create(int(stereo), prob.inv_depth, _dp(prob.calib_l), _dp(prob.calib_r), _dp(t_lr7), 752, 480,
                                   nmin_covscore)
        for k in range(len(prob.pose)):
            L.ov2h_map_add_keyframe(self.h, k, _dp(np.ascontiguousarray(prob.pose[k])))
        # observations per (kf, lm) from the residual list
        obs = {}
        for i in range(prob.n_res):
            t, k, l = int(prob.res_type[i]), int(prob.res_pose[i]), int(prob.res_lm[i])
            if t == T.RANCH_INV:
                k = int(prob.lm_anchor_pose[l])
            o = obs.setdefault((k, l), {})
            if t in (T.L_XYZ, T.L_INV):
                o["l"] = prob.res_uv[i]
            else:
                o["r"] = prob.res_uv[i]
        self.xyz0 = np.zeros((len(prob.lm), 3))
        for l in range(len(prob.lm)):
            if prob.inv_depth:
                a = int(prob.lm_anchor_pose[l])
                obs.setdefault((a, l), {})["l"] = prob.lm_anchor_uv[l]
                z = 1.0 / prob.lm[l, 0]
                u, v = prob.lm_anchor_uv[l]
                pc = z * np.array([(u - prob.calib_l[2]) / prob.calib_l[0], (v - prob.calib_l[3]) / prob.calib_l[1], 1.0])
                xyz = synth_ba.quat_to_rot(prob.pose[a, 3:]) @ pc + prob.pose[a, :3]
            else:
                xyz = prob.lm[l]
            self.xyz0[l] = xyz
        anchor = {}
        for (k, l) in obs:
            anchor[l] = min(anchor.get(l, 10 ** 9), k)
        for l in range(len(prob.lm)):
            if l in anchor:
                L.ov2h_map_add_landmark(self.h, l, _dp(np.ascontiguousarray(self.xyz0[l])), anchor[l])
        for (k, l), o in sorted(obs.items()):
            ul = o["l"]
            ur = o.get("r")
            L.ov2h_map_add_obs(self.h, k, l, ul[0], ul[1], int(ur is not None), 0.0 if ur is None else ur[0],
                               0.0 if ur is None else ur[1])
        self.newkf = len(prob.pose) - 1
        assert L.ov2h_map_finalize(self.h, self.newkf) == 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().ov2h_map_destroy(self.h)
            self.h = None

    def attach_device(self, ctx, max_kf=None, max_lm=None, max_obs=None):
        """MapManager::attachDevice: mirror the whole map into an ov2_map (HBM tables); later set-ups can run there."""
        n_obs = self.prob.n_res + len(self.prob.lm)
        rc = lib().ov2h_map_attach_device(self.h, ctx.h, max_kf or len(self.prob.pose) + 8, max_lm or len(self.prob.lm) + 8,
                                          max_obs or n_obs + 64)
        if rc != 0:
            raise RuntimeError(f"attachDevice failed (status {rc})")

    def set_distortion(self, cam, model, coeffs):
        """CameraCalibration::Dcv_ of the left (0) / right (1) camera: model 'pinhole' (k1 k2 p1 p2 [k3]) or 'fisheye' (k1..k4)"""
        d = np.ascontiguousarray(coeffs, np.float64)
        assert lib().ov2h_set_distortion(self.h, int(cam), 1 if model == "fisheye" else 0, len(d), _dp(d)) == 0

    def undistort(self, cam, x, y):
        out = (C.c_float * 2)()
        assert lib().ov2h_undistort(self.h, int(cam), float(x), float(y), out) == 0
        return np.array([out[0], out[1]], np.float32)

    def project_dist(self, cam, pc):
        out = (C.c_float * 2)()
        assert lib().ov2h_project_dist(self.h, int(cam), _dp(np.ascontiguousarray(pc, np.float64)), out) == 0
        return np.array([out[0], out[1]], np.float32)

    def device_handle(self):
        """the ov2_map* of the attached device mirror"""
        return C.c_void_p(lib().ov2h_map_device_handle(self.h))

    def flush_device(self):
        assert lib().ov2h_map_flush_device(self.h) == 0

    def device_rows(self):
        """(rows, capacity, compactions) of the device mirror's observation table"""
        r, c, n = C.c_int(), C.c_int(), C.c_int()
        assert lib().ov2h_map_device_rows(self.h, C.byref(r), C.byref(c), C.byref(n)) == 0
        return r.value, c.value, n.value

    def remove_obs(self, kfid, lmid):
        lib().ov2h_map_remove_obs(self.h, int(kfid), int(lmid))

    def remove_landmark(self, lmid):
        lib().ov2h_map_remove_landmark(self.h, int(lmid))

    def set_isobs(self, lmid, isobs):
        """MapPoint::isobs_ (seen by the current frame)"""
        assert lib().ov2h_map_set_isobs(self.h, int(lmid), int(isobs)) == 0

    def bad_lmids(self):
        buf = np.zeros(max(1, len(self.prob.lm)), np.int32)
        n = lib().ov2h_map_bad_lmids(self.h, buf.ctypes.data_as(C.POINTER(C.c_int)), len(buf))
        return np.sort(buf[:n])

    def setup_local_ba(self, dev=False):
        """Optimizer::setupLocalBA (CPU hash-map walk) or, dev=True, Optimizer::setupLocalBADevice (scans of the device
        map mirror). returns dict of the flat problem keyed by reference ids."""
        L = lib()
        npose, nlm, nres = C.c_int(), C.c_int(), C.c_int()
        fn = L.ov2h_local_ba_setup_dev if dev else L.ov2h_local_ba_setup
        rc = fn(self.h, self.newkf, C.byref(npose), C.byref(nlm), C.byref(nres))
        if rc < 0:
            raise RuntimeError(f"local BA set-up failed ({rc})")
        return self._read_problem(rc, npose, nlm, nres)

    def setup_range_ba(self, kf_lo, kf_hi, kf_obs_max=2 ** 31 - 1, min_obs=0):
        """Optimizer::setupRangeBA: the set-up stage of fullBA (0, last, no observer filter, min_obs 3) / looseBA"""
        npose, nlm, nres = C.c_int(), C.c_int(), C.c_int()
        lib().ov2h_range_ba_setup(self.h, int(kf_lo), int(kf_hi), int(kf_obs_max), int(min_obs), C.byref(npose), C.byref(nlm), C.byref(nres))
        return self._read_problem(0, npose, nlm, nres)

    def full_ba(self, ctx, robust=True):
        """Optimizer::fullBA on the GPU. returns (status, outliers pass 1, pass 2, final cost, logged iterations)"""
        n1, n2, fc, it = C.c_int(), C.c_int(), C.c_double(), C.c_int()
        st = lib().ov2h_full_ba(self.h, ctx.h, int(robust), C.byref(n1), C.byref(n2), C.byref(fc), C.byref(it))
        return st, n1.value, n2.value, fc.value, it.value

    def loose_ba(self, ctx, inikfid, nkfid, robust=True):
        """Optimizer::looseBA on the GPU. returns (status, flagged observations, final cost)"""
        n1, fc = C.c_int(), C.c_double()
        st = lib().ov2h_loose_ba(self.h, ctx.h, int(inikfid), int(nkfid), int(robust), C.byref(n1), C.byref(fc))
        return st, n1.value, fc.value

    def local_pose_graph(self, ctx, newkf, kfloop_id, newTwc):
        """Optimizer::localPoseGraph. returns (1 accepted / 0 rejected, final cost, logged iterations)"""
        fc, nl = C.c_double(), C.c_int()
        T = np.ascontiguousarray(newTwc, np.float64)
        rc = lib().ov2h_local_pose_graph(self.h, ctx.h, int(newkf), int(kfloop_id), _dp(T), C.byref(fc), C.byref(nl))
        if rc < 0:
            raise RuntimeError(f"localPoseGraph failed ({rc})")
        return rc, fc.value, nl.value

    def full_pose_graph(self, ctx, Twc, Tpc, iskf):
        """Optimizer::fullPoseGraph on (n x 7) poses, (n x 7) relative poses prev -> cur, keyframe flags. returns (ok, Twc, cost)"""
        Tw = np.ascontiguousarray(Twc, np.float64).copy()
        Tp = np.ascontiguousarray(Tpc, np.float64)
        kf = np.ascontiguousarray(iskf, np.uint8)
        fc = C.c_double()
        rc = lib().ov2h_full_pose_graph(self.h, ctx.h, len(Tw), _dp(Tw), _dp(Tp), kf.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(fc))
        if rc < 0:
            raise RuntimeError(f"fullPoseGraph failed ({rc})")
        return rc == 1, Tw, fc.value

    def _read_problem(self, rc, npose, nlm, nres):
        L = lib()
        e = 1 if self.prob.inv_depth else 3
        ip, u8 = C.POINTER(C.c_int), C.POINTER(C.c_uint8)
        out = dict(aborted=rc == 1, pose_kfid=np.zeros(npose.value, np.int32), pose_const=np.zeros(npose.value, np.uint8),
                   pose=np.zeros((npose.value, 7)), lm_lmid=np.zeros(nlm.value, np.int32), lm=np.zeros((nlm.value, e)),
                   lm_anchor_kfid=np.zeros(nlm.value, np.int32), lm_anchor_uv=np.zeros((nlm.value, 2)),
                   res_type=np.zeros(nres.value, np.uint8), res_kfid=np.zeros(nres.value, np.int32),
                   res_lmid=np.zeros(nres.value, np.int32), res_uv=np.zeros((nres.value, 2)))
        L.ov2h_local_ba_get(self.h, out["pose_kfid"].ctypes.data_as(ip), out["pose_const"].ctypes.data_as(u8), _dp(out["pose"]),
                            out["lm_lmid"].ctypes.data_as(ip), _dp(out["lm"]), out["lm_anchor_kfid"].ctypes.data_as(ip),
                            _dp(out["lm_anchor_uv"]), out["res_type"].ctypes.data_as(u8), out["res_kfid"].ctypes.data_as(ip),
                            out["res_lmid"].ctypes.data_as(ip), _dp(out["res_uv"]))
        return out

    def apply_local_ba(self, ctx):
        """Estimator::applyLocalBA on the GPU. returns (status, outliers pass1, pass2, final cost)."""
        n1, n2, fc = C.c_int(), C.c_int(), C.c_double()
        st = lib().ov2h_apply_local_ba(self.h, ctx.h, self.newkf, C.byref(n1), C.byref(n2), C.byref(fc))
        return st, n1.value, n2.value, fc.value

    def compute_pose(self, ctx, kfid, Twc_init):
        """VisualFrontEnd::computePose with keyframe `kfid` as the current frame. returns (status, p3p requested)."""
        req = C.c_int()
        st = lib().ov2h_compute_pose(self.h, ctx.h, kfid, _dp(np.ascontiguousarray(Twc_init, np.float64)), C.byref(req))
        return st, bool(req.value)

    def structure_only_ba(self, ctx, lmids):
        """Optimizer::structureOnlyBA(vlm2optids) (src/optimizer.cpp:2594-2781). returns (status, final cost, LM iterations)"""
        ids = np.ascontiguousarray(lmids, np.int32)
        cost, it = C.c_double(0), C.c_int(0)
        st = lib().ov2h_structure_only_ba(self.h, ctx.h, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cost), C.byref(it))
        return st, cost.value, it.value

    def pose(self, kfid):
        out = np.zeros(7)
        assert lib().ov2h_get_pose(self.h, kfid, _dp(out)) == 0
        return out

    def landmark(self, lmid):
        out, n = np.zeros(3), C.c_int()
        rc = lib().ov2h_get_landmark(self.h, lmid, _dp(out), C.byref(n))
        return (out, n.value) if rc == 0 else (None, 0)

    def counts(self, kfid):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().ov2h_count_keypoints(self.h, kfid, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


class EstimatorWorker:
    """the reference's Estimator thread (src/estimator.cpp:32-98) as a NATIVE thread of libov2host.so with its own
    high-priority HIP context: waits for keyframes and solves the windows of ALL sequences that have one pending in one
    ov2_ba_solve_batch call (at most max_batch).  Python only submits keyframes and reads the counters, so the worker never
    competes for the interpreter lock."""

    def __init__(self, device, problem, nseq, robust_mono_th=5.9915, max_batch=64, high_priority=True, device_resident=False):
        self.problem = problem              # keeps the arrays alive during the deep copy
        pc = problem.as_c()
        self.h = lib().ov2h_ba_worker_create(device, C.addressof(pc), robust_mono_th, nseq, max_batch, int(bool(high_priority)))
        if not self.h:
            raise RuntimeError("ov2h_ba_worker_create failed (no GPU?)")
        if device_resident:   # windows kept in HBM, solved by ov2_ba_solve_batch_dev (before the first submission)
            lib().ov2h_ba_worker_set_device_resident(self.h, 1)

    def submit_all(self):
        lib().ov2h_ba_worker_submit_all(self.h)

    def set_counting(self, on):
        lib().ov2h_ba_worker_set_counting(self.h, int(bool(on)))

    def stats(self):
        out = np.zeros(7)
        lib().ov2h_ba_worker_stats(self.h, _dp(out))
        return dict(solves=int(out[0]), iters=int(out[1]), dropped=int(out[2]), submitted=int(out[3]), busy_s=float(out[4]),
                    last_status=int(out[5]), batches=int(out[6]))

    def close(self):
        if getattr(self, "h", None):
            lib().ov2h_ba_worker_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class CppSlam:
    """ov2::SlamManager of libov2host.so (host/ov2_slam.hpp): the closed loop in C++ -- VisualFrontEnd::visualTracking per
    frame, MapManager::createKeyframe + Mapper::run + Estimator::applyLocalBA per keyframe -- driven one stereo frame per call.
    policy: None = the reference's own heuristics (checkNewKfReq, covisibility local BA, se3 motion model, its triangulation);
    'slam_loop' = the fixed stand-ins of ov2slam_amd.slam_loop.SlamLoop (keyframe every kf_every frames, BA over the last
    ba_window keyframes, ...), which makes the two loops comparable pose by pose."""

    STAT_KEYS = ("frame", "tracked", "n3d", "kf", "new_kps", "stereo", "n_lm", "ba", "ba_res", "ba_it_robust", "ba_it_l2",
                 "ba_outliers", "ba_cost0", "ba_cost1", "keyframes", "landmarks")

    def __init__(self, ctx, K4, baseline, w, h, cell=35, rectified=True, policy=None, kf_every=5, ba_window=8, ba_fixed=2,
                 device_map=False):
        pol = np.zeros(6, np.int32)
        if policy == "slam_loop":
            pol[:] = [kf_every, ba_window, ba_fixed, 1, 1, 1]
        elif policy is not None:
            raise ValueError(policy)
        K = np.ascontiguousarray(K4, np.float64)
        self.w, self.h, self.ctx = w, h, ctx
        self.h_ = lib().ov2h_slam_create(ctx.h, _dp(K), float(baseline), w, h, cell, int(bool(rectified)),
                                         pol.ctypes.data_as(C.POINTER(C.c_int)), int(bool(device_map)))
        if not self.h_:
            raise RuntimeError("ov2h_slam_create failed")
        self.traj, self.stats = [], []

    def step(self, time, img_left, img_right):
        u8 = C.POINTER(C.c_uint8)
        a, b = np.ascontiguousarray(img_left, np.uint8), np.ascontiguousarray(img_right, np.uint8)
        st = lib().ov2h_slam_add_stereo(self.h_, float(time), a.ctypes.data_as(u8), b.ctypes.data_as(u8), self.w, self.h)
        if st != 0:
            raise RuntimeError(f"SlamManager::addNewStereoImages failed ({st}): {self.ctx.lib.ov2_last_error(self.ctx.h)}")
        T, s = np.zeros(7), np.zeros(16)
        lib().ov2h_slam_pose(self.h_, _dp(T))
        lib().ov2h_slam_stats(self.h_, _dp(s))
        self.traj.append(T)
        self.stats.append(dict(zip(self.STAT_KEYS, s.tolist())))
        if getattr(self, "kf_stats", None) is not None and self.stats[-1]["kf"]:
            k = np.zeros(3)
            lib().ov2h_slam_kf_stats(self.h_, _dp(k))
            self.kf_stats.append(dict(frame=int(self.stats[-1]["frame"]), described=int(k[0]), local=int(k[1]), matched=int(k[2])))
        return T

    def set_brief(self, pattern, use_brief=True, track_localmap=True, fmax_desc_dist=0.0, fmax_proj_pxdist=0.0):
        """use_brief / bdo_track_localmap of the YAML: keyframes describe their keypoints (BRIEF-32 with the caller's 256 x 4 int8
        test table -- opencv_contrib's is not in the reference tree) and run Mapper::matchingToLocalMap + mergeMapPoints"""
        pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8).reshape(256, 4)
        lib().ov2h_slam_set_brief(self.h_, None if pat is None else pat.ctypes.data, int(bool(use_brief)), int(bool(track_localmap)),
                                  float(fmax_desc_dist), float(fmax_proj_pxdist))
        self.kf_stats = []

    def check_map(self):
        """(violations of the host map's invariants, total) -- see ov2h_slam_check_map"""
        v = np.zeros(6, np.int32)
        tot = lib().ov2h_slam_check_map(self.h_, v.ctypes.data_as(C.POINTER(C.c_int)))
        return dict(zip(("kp_without_mp", "kp_not_listed", "observer_without_kp", "covisibility", "mp3d_without_desc", "desc_without_observer"),
                        v.tolist())), int(tot)

    def export_map(self, cap_kf=4096, cap_lm=1 << 18, cap_obs=1 << 20):
        """the host map keyed by ids: ({kfid: pose}, {lmid: (xyz, state bits)}, {(kfid, lmid): stereo})"""
        ip, u8 = C.POINTER(C.c_int), C.POINTER(C.c_uint8)
        n = np.zeros(3, np.int32)
        kf_id, kf_pose = np.zeros(cap_kf, np.int32), np.zeros((cap_kf, 7))
        lm_id, lm_xyz, lm_st = np.zeros(cap_lm, np.int32), np.zeros((cap_lm, 3)), np.zeros(cap_lm, np.uint8)
        ok, ol, os_ = np.zeros(cap_obs, np.int32), np.zeros(cap_obs, np.int32), np.zeros(cap_obs, np.uint8)
        lib().ov2h_slam_export_map(self.h_, cap_kf, cap_lm, cap_obs, n.ctypes.data_as(ip), kf_id.ctypes.data_as(ip), _dp(kf_pose),
                                   lm_id.ctypes.data_as(ip), _dp(lm_xyz), lm_st.ctypes.data_as(u8), ok.ctypes.data_as(ip),
                                   ol.ctypes.data_as(ip), os_.ctypes.data_as(u8))
        assert n[0] <= cap_kf and n[1] <= cap_lm and n[2] <= cap_obs
        kfs = {int(k): tuple(p) for k, p in zip(kf_id[:n[0]], kf_pose[:n[0]])}
        lms = {int(l): (tuple(x), int(s)) for l, x, s in zip(lm_id[:n[1]], lm_xyz[:n[1]], lm_st[:n[1]])}
        obs = {(int(k), int(l)): int(s) for k, l, s in zip(ok[:n[2]], ol[:n[2]], os_[:n[2]])}
        return kfs, lms, obs

    def device_handle(self):
        return lib().ov2h_slam_device_handle(self.h_)

    def flush_device(self):
        st = lib().ov2h_slam_flush_device(self.h_)
        if st != 0:
            raise RuntimeError(f"MapManager::flushDevice failed ({st})")

    def landmarks(self, cap=1 << 16):
        ids, xyz = np.zeros(cap, np.int32), np.zeros((cap, 3))
        n = lib().ov2h_slam_landmarks(self.h_, cap, ids.ctypes.data_as(C.POINTER(C.c_int)), _dp(xyz))
        return ids[:n].copy(), xyz[:n].copy()

    def close(self):
        if getattr(self, "h_", None):
            lib().ov2h_slam_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()


class EstimatorPipeline:
    """the Estimator threads of `len(maps)` SLAM instances as ONE native thread that runs the WHOLE of Optimizer::localBA per
    keyframe job on device-resident maps (ov2slam_amd/device_map.DeviceMap, all created on `ctx`): set-up
    (ov2_map_local_ba_setup_batch, src/optimizer.cpp:43-430) -> solve (ov2_ba_solve_batch_dev, :439-735) -> update
    (ov2_map_local_ba_update_batch, :741-882) for every sequence that has a keyframe pending, in one batch.  `ctx` and the
    maps belong to the thread until close(); the caller must have saved the maps' state (every job starts from it)."""

    def __init__(self, ctx, maps, proto, robust_mono_th=5.9915, max_batch=64):
        self.ctx, self.maps, self.proto = ctx, maps, proto
        n = len(maps)
        hs = (C.c_void_p * n)(*[m.h for m in maps])
        nk = np.ascontiguousarray([m.newkf for m in maps], np.int32)
        K = np.ascontiguousarray(np.tile(np.asarray(proto.calib_l, np.float64), (n, 1)))
        pc = proto.as_c()
        self.h = lib().ov2h_ba_pipeline_create(ctx.h, n, hs, nk.ctypes.data_as(C.POINTER(C.c_int)), _dp(K), C.addressof(pc),
                                               robust_mono_th, int(proto.inv_depth), max_batch)
        if not self.h:
            raise RuntimeError("ov2h_ba_pipeline_create failed")

    def submit_all(self):
        lib().ov2h_ba_pipeline_submit_all(self.h)

    def set_counting(self, on):
        lib().ov2h_ba_pipeline_set_counting(self.h, int(bool(on)))

    def stats(self):
        o = np.zeros(40)
        lib().ov2h_ba_pipeline_stats(self.h, _dp(o))
        return dict(solves=int(o[0]), iters=int(o[1]), dropped=int(o[2]), submitted=int(o[3]), busy_s=float(o[4]),
                    last_status=int(o[5]), batches=int(o[6]), setup_s=float(o[7]), solve_s=float(o[8]), update_s=float(o[9]),
                    slowest_sum=int(o[10]), res_blocks=int(o[11]), aborted=int(o[12]), iter_blocks=int(o[13]),
                    hist_robust=[int(x) for x in o[16:24]], hist_l2=[int(x) for x in o[24:40]])

    def close(self):
        if getattr(self, "h", None):
            lib().ov2h_ba_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class _FeLoopCfg(C.Structure):   # ov2h_feloop_cfg (host/ov2_host_capi.cpp)
    _fields_ = [(k, C.c_int32) for k in ("L", "B", "n", "win", "nlvl", "use_clahe", "tiles_x", "tiles_y")] + \
               [(k, C.c_float) for k in ("clahe_clip", "err_th", "fb_th", "eps")] + \
               [(k, C.c_int32) for k in ("max_iter", "detect", "det_cell", "det_ncur", "det_cap", "pnp")] + \
               [(k, C.c_void_p) for k in ("left", "right", "kps", "pri", "st_pri", "has", "st_has", "img_idx", "out_xy", "out_st", "p3p",
                                          "pnp_off", "pnp_unpx", "pnp_wpts", "pnp_K", "pnp_T0", "pnp_T", "pnp_outl", "pnp_rem", "pnp_ok")] + \
               [("pnp_T_bytes", C.c_uint64)] + \
               [(k, C.c_void_p) for k in ("det_thresh", "det_cur", "det_img", "det_nout", "det_out")]


class FrameLoop:
    """native per-frame driver of a bench Workload (bench.py): the same sequence of ABI calls as Workload.step -- pyramid,
    kltTracking, ceresPnP; right pyramid, stereoMatching, detector and a job for every Estimator pipeline on keyframes --
    enqueued from C++ (ov2h_feloop_run), so that small streams are not bound by ~15 us of interpreter per call.  Holds
    references to the workload's device arrays; everything stays asynchronous."""

    def __init__(self, ctx, wl, win, nlvl, tiles, pipelines=()):
        self.ctx, self.wl = ctx, wl
        L = wl.L
        arr = lambda xs: (C.c_void_p * L)(*xs)
        self._keep = dict(left=arr([i.h_ for i in wl.left]), right=arr([i.h_ for i in wl.right]),
                          kps=arr([a.ptr for a in wl.kps]), pri=arr([a.ptr for a in wl.pri]), st_pri=arr([a.ptr for a in wl.st_pri]),
                          has=arr([a.ptr for a in wl.has]), st_has=arr([a.ptr for a in wl.st_has]))
        c = _FeLoopCfg()
        c.L, c.B, c.n, c.win, c.nlvl, c.use_clahe, c.tiles_x, c.tiles_y = L, wl.B, wl.n, win, nlvl, 1, tiles[0], tiles[1]
        c.clahe_clip, c.err_th, c.fb_th, c.eps, c.max_iter = 3.0, 30.0, 0.5, wl.trk.fmax_px_precision, wl.trk.nmax_iter
        c.detect, c.det_cell, c.det_ncur, c.det_cap = int(bool(wl.detect)), wl.det_cell, wl.det_ncur, wl.det_cap
        for k, v in self._keep.items():
            setattr(c, k, C.cast(v, C.c_void_p))
        c.img_idx, c.out_xy, c.out_st, c.p3p = wl.img_idx.ptr, wl.out_xy.ptr, wl.out_st.ptr, wl.p3p.ptr
        c.det_thresh, c.det_cur, c.det_img = wl.d_det_thresh.ptr, wl.d_det_cur.ptr, wl.d_det_img.ptr
        c.det_nout, c.det_out = wl.d_det_nout.ptr, wl.d_det_out.ptr
        if wl.pnp:
            q = wl.pnp
            c.pnp = 1
            c.pnp_off, c.pnp_unpx, c.pnp_wpts, c.pnp_K = q["off"].ptr, q["unpx"].ptr, q["wpts"].ptr, q["K"].ptr
            c.pnp_T0, c.pnp_T, c.pnp_T_bytes = q["T0"].ptr, q["T"].ptr, q["T"].nbytes
            c.pnp_outl, c.pnp_rem, c.pnp_ok = q["outl"].ptr, q["rem"].ptr, q["ok"].ptr
        self.cfg = c
        self.pipes = (C.c_void_p * max(1, len(pipelines)))(*[p.h for p in pipelines])
        self.n_pipes = len(pipelines)
        self.h = lib().ov2h_feloop_create(ctx.h, C.addressof(c))
        if not self.h:
            raise RuntimeError("ov2h_feloop_create failed")

    def run(self, steps, kf_every):
        """enqueue `steps` frames; returns the number of keyframes among them"""
        nk = C.c_int(0)
        st = lib().ov2h_feloop_run(self.h, int(steps), int(kf_every), self.pipes, self.n_pipes, C.byref(nk))
        if st != 0:
            raise RuntimeError(f"ov2h_feloop_run: status {st}: {self.ctx.lib.ov2_last_error(self.ctx.h).decode()}")
        return nk.value

    def close(self):
        if getattr(self, "h", None):
            lib().ov2h_feloop_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class FrontEndFrame:
    """one keyframe with 2D / 3D keypoints in the C++ host mirror (Frame + MapManager + CameraCalibration pair), to drive
    MapManager::stereoMatching and VisualFrontEnd::kltTracking through libov2host.so.  Test / bring-up only."""

    def __init__(self, K, baseline, w, h, Twc=None, ncellsize=35, stereo_rect=False, klt_use_prior=True, nklt_pyr_lvl=3,
                 nklt_win_size=9):
        L = lib()
        K = np.ascontiguousarray(K, np.float64)
        t_lr7 = np.ascontiguousarray([baseline, 0, 0, 0, 0, 0, 1.0])    # right camera in the left frame (Tc0ci)
        self.h = L.ov2h_map_create(1, 1, _dp(K), _dp(K), _dp(t_lr7), w, h, 25)
        self.w, self.h_img, self.kfid = w, h, 0
        T = np.ascontiguousarray([0, 0, 0, 0, 0, 0, 1.0] if Twc is None else Twc, np.float64)
        L.ov2h_map_add_keyframe(self.h, 0, _dp(T))
        L.ov2h_set_params(self.h, int(klt_use_prior), int(stereo_rect), nklt_pyr_lvl, nklt_win_size)
        L.ov2h_frame_init_grid(self.h, 0, ncellsize)

    def add_keypoint(self, lmid, px, xyz=None):
        x = None if xyz is None else np.ascontiguousarray(xyz, np.float64)
        lib().ov2h_map_add_kp(self.h, 0, int(lmid), float(px[0]), float(px[1]), int(xyz is not None), None if x is None else _dp(x))

    def forget_landmark(self, lmid):
        lib().ov2h_map_forget_landmark(self.h, int(lmid))

    def stereo_matching(self, ctx, img_left, img_right):
        u8 = C.POINTER(C.c_uint8)
        a, b = np.ascontiguousarray(img_left, np.uint8), np.ascontiguousarray(img_right, np.uint8)
        return lib().ov2h_stereo_matching(self.h, ctx.h, 0, a.ctypes.data_as(u8), b.ctypes.data_as(u8), self.w, self.h_img)

    def klt_tracking(self, ctx, img_prev, img_cur):
        u8 = C.POINTER(C.c_uint8)
        a, b = np.ascontiguousarray(img_prev, np.uint8), np.ascontiguousarray(img_cur, np.uint8)
        req = C.c_int(0)
        st = lib().ov2h_klt_tracking(self.h, ctx.h, 0, a.ctypes.data_as(u8), b.ctypes.data_as(u8), self.w, self.h_img, C.byref(req))
        return st, bool(req.value)

    def keypoints(self, cap=65536):
        """dict lmid -> (px (2,), is3d, is_stereo, rpx (2,))"""
        fpp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        lm, px, rp = np.zeros(cap, np.int32), np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        i3, ist = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        n = lib().ov2h_get_keypoints(self.h, 0, cap, lm.ctypes.data_as(C.POINTER(C.c_int)), px.ctypes.data_as(fpp),
                                     i3.ctypes.data_as(u8), ist.ctypes.data_as(u8), rp.ctypes.data_as(fpp))
        assert 0 <= n <= cap
        return {int(lm[k]): (px[k].copy(), bool(i3[k]), bool(ist[k]), rp[k].copy()) for k in range(n)}

    def frl(self):
        F = np.zeros(9)
        lib().ov2h_get_frl(self.h, 0, _dp(F))
        return F.reshape(3, 3)

    def close(self):
        if getattr(self, "h", None):
            lib().ov2h_map_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
