"""ctypes binding of the CPU oracle (libov2oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (ov2slam_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int)
i64p = C.POINTER(C.c_int64)


def build(force=False):
    so = os.path.join(_HERE, "libov2oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def use_native(on=True):
    """bench.py's cpu_baseline leg only: switch to a build of the same sources with -O3 -march=native for THIS host
    (made on demand by `make native`; SURVEY.md 8d asks for the CPU restatement at -O3 -march=native).  The default
    build stays the checker of the tests (-O2, portable).  Returns True if the native build is in use."""
    global _LIB
    so = os.path.join(_HERE, "libov2oracle_native.so")
    if on:
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        except Exception:
            return False
        _LIB = None
        lib(so)
        return True
    _LIB = None
    lib()
    return False


def lib(path=None):
    global _LIB
    if _LIB is None:
        so = path or os.path.join(_HERE, "libov2oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.ov2o_pyramid_build.restype = C.c_void_p
        L.ov2o_pyramid_build.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ov2o_pyramid_free.argtypes = [C.c_void_p]
        L.ov2o_pyr_nlevels.argtypes = [C.c_void_p]
        L.ov2o_pyr_level_info.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p]
        L.ov2o_pyr_image.restype = u8p
        L.ov2o_pyr_image.argtypes = [C.c_void_p, C.c_int]
        L.ov2o_pyr_grad.restype = C.POINTER(C.c_int16)
        L.ov2o_pyr_grad.argtypes = [C.c_void_p, C.c_int]
        L.ov2o_clahe.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, u8p, C.c_int]
        L.ov2o_calc_optical_flow_pyr_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, f32p, u8p, f32p,
                                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, i32p]
        L.ov2o_fb_klt_tracking.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                           C.c_int, C.c_float, C.c_int, f32p, f32p, u8p, i64p]
        L.ov2o_klt_tracking_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                              C.c_int, C.c_float, C.c_int, f32p, f32p, u8p, f32p, u8p, i32p]
        L.ov2o_set_num_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def set_num_threads(n):
    """threads over the points of one LK call (OpenCV's parallel_for_); results do not depend on it"""
    lib().ov2o_set_num_threads(int(n))


def _p(a, t):
    return a.ctypes.data_as(t)


class Pyramid:
    """cv::buildOpticalFlowPyramid restatement (padded images + Scharr gradients)."""

    def __init__(self, img, win=9, max_level=3):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self._h = lib().ov2o_pyramid_build(_p(img, u8p), w, h, w, win, max_level)
        if not self._h:
            raise MemoryError("ov2o_pyramid_build failed")
        self.nlevels = lib().ov2o_pyr_nlevels(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ov2o_pyramid_free(self._h)
            self._h = None

    def level(self, l):
        """returns (img[(h+2p),(w+2p)] u8, grad[(h+2p),(w+2p),2] i16, w, h, pad), copies."""
        w, h, p, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().ov2o_pyr_level_info(self._h, l, C.byref(w), C.byref(h), C.byref(p), C.byref(s))
        w, h, p, s = w.value, h.value, p.value, s.value
        rows = h + 2 * p
        img = np.ctypeslib.as_array(lib().ov2o_pyr_image(self._h, l), shape=(rows, s)).copy()
        grad = np.ctypeslib.as_array(lib().ov2o_pyr_grad(self._h, l), shape=(rows, s, 2)).copy()
        return img, grad, w, h, p


def clahe(img, clip=3.0, tiles_x=15, tiles_y=9):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty_like(img)
    lib().ov2o_clahe(_p(img, u8p), w, h, w, clip, tiles_x, tiles_y, _p(out, u8p), w)
    return out


def calc_optical_flow_pyr_lk(prev, nxt, prev_xy, next_xy, win=9, max_level=3, max_iter=30, eps=0.01,
                             min_eig_thr=1e-4):
    prev_xy = np.ascontiguousarray(prev_xy, dtype=np.float32)
    out = np.ascontiguousarray(next_xy, dtype=np.float32).copy()
    n = prev_xy.shape[0]
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    ml = min(max_level, prev.nlevels - 1, nxt.nlevels - 1)
    iters = np.zeros((n, ml + 1), np.int32)
    lib().ov2o_calc_optical_flow_pyr_lk(prev._h, nxt._h, n, _p(prev_xy, f32p), _p(out, f32p), _p(status, u8p),
                                        _p(err, f32p), win, max_level, max_iter, eps, min_eig_thr,
                                        _p(iters, i32p))
    return out, status, err, iters


def fb_klt_tracking(prev, cur, kps_xy, priors_xy, win=9, nlevels=3, err_th=30.0, fb_th=0.5, max_iter=30,
                    eps=0.01):
    """FeatureTracker::fbKltTracking. returns (priors_out, status, total_lk_iterations)."""
    kps = np.ascontiguousarray(kps_xy, dtype=np.float32)
    pri = np.ascontiguousarray(priors_xy, dtype=np.float32).copy()
    n = kps.shape[0]
    status = np.zeros(n, np.uint8)
    total = C.c_int64(0)
    lib().ov2o_fb_klt_tracking(prev._h, cur._h, win, nlevels, err_th, fb_th, max_iter, eps, n, _p(kps, f32p),
                               _p(pri, f32p), _p(status, u8p), C.byref(total))
    return pri, status, total.value


def klt_tracking_frame(prev, cur, kps_xy, prior_xy, has_prior, win=9, nlevels=3, err_th=30.0, fb_th=0.5,
                       max_iter=30, eps=0.01):
    kps = np.ascontiguousarray(kps_xy, dtype=np.float32)
    pri = np.ascontiguousarray(prior_xy, dtype=np.float32)
    hp = np.ascontiguousarray(has_prior, dtype=np.uint8)
    n = kps.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    p3p = C.c_int(0)
    lib().ov2o_klt_tracking_frame(prev._h, cur._h, win, nlevels, err_th, fb_th, max_iter, eps, n, _p(kps, f32p),
                                  _p(pri, f32p), _p(hp, u8p), _p(out, f32p), _p(st, u8p), C.byref(p3p))
    return out, st, bool(p3p.value)


# ---------------------------------------------------------------------------------------------------
# BA half of the oracle

class _ResEvalC(C.Structure):
    _fields_ = [("r", C.c_double * 2), ("Jk", C.c_double * 12), ("Ja", C.c_double * 12), ("Jl", C.c_double * 6),
                ("chi2", C.c_double), ("depth_positive", C.c_int)]


class _BsProblemC(C.Structure):
    _fields_ = [("R", C.c_int), ("E", C.c_int), ("F", C.c_int), ("maxf", C.c_int), ("n_rows", C.c_int),
                ("n_e", C.c_int), ("n_f", C.c_int), ("row_e", i32p), ("row_f", i32p), ("Je", f64p), ("Jf", f64p),
                ("b", f64p), ("D", f64p)]


def _ba_lib():
    L = lib()
    if not getattr(L, "_ba_bound", False):
        from ov2slam_amd import ba_types as T
        L.ov2o_se3_exp.argtypes = [f64p, f64p]
        L.ov2o_se3_plus.argtypes = [f64p, f64p, f64p]
        L.ov2o_ba_eval_residual.argtypes = [C.POINTER(T.BaProblemC), f64p, f64p, C.c_int, C.c_int, C.POINTER(_ResEvalC)]
        L.ov2o_huber.argtypes = [C.c_double, C.c_double, f64p]
        L.ov2o_schur_solve.argtypes = [C.POINTER(_BsProblemC), f64p, f64p, f64p]
        L.ov2o_schur_solve.restype = C.c_int
        L.ov2o_ba_default_options.argtypes = [C.POINTER(T.BaOptionsC), C.c_float]
        L.ov2o_ba_solve.argtypes = [C.POINTER(T.BaProblemC), C.POINTER(T.BaOptionsC), C.POINTER(T.BaResultC)]
        L.ov2o_pnp_solve.argtypes = [C.c_int, f64p, f64p, i32p, f64p, f64p, C.c_int, C.c_float, C.c_int, C.c_int,
                                     u8p, i32p]
        L.ov2o_pnp_solve.restype = C.c_int
        L.ov2o_pg_eval_edge.argtypes = [f64p, f64p, f64p, f64p, f64p, f64p]
        L.ov2o_pose_graph_solve.argtypes = [C.POINTER(T.PgProblemC), C.POINTER(T.BaOptionsC), C.POINTER(T.PgResultC)]
        L.ov2o_pose_graph_solve.restype = C.c_int
        L._ba_bound = True
    return L


def se3_exp(d):
    d = np.ascontiguousarray(d, np.float64)
    out = np.zeros(7)
    _ba_lib().ov2o_se3_exp(_p(d, f64p), _p(out, f64p))
    return out


def se3_plus(x, d):
    x, d = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(d, np.float64)
    out = np.zeros(7)
    _ba_lib().ov2o_se3_plus(_p(x, f64p), _p(d, f64p), _p(out, f64p))
    return out


def ba_eval_residual(prob, i, poses=None, lms=None, want_jac=True):
    """returns dict(r, Jk(2,6), Ja(2,6), Jl(2,e), chi2, depth_positive) of residual i at (poses, lms)."""
    L = _ba_lib()
    poses = prob.pose if poses is None else np.ascontiguousarray(poses, np.float64)
    lms = prob.lm if lms is None else np.ascontiguousarray(lms, np.float64)
    pc = prob.as_c()
    ev = _ResEvalC()
    L.ov2o_ba_eval_residual(C.byref(pc), _p(poses, f64p), _p(lms, f64p), i, int(want_jac), C.byref(ev))
    e = 1 if prob.inv_depth else 3
    return dict(r=np.array(ev.r[:]), Jk=np.array(ev.Jk[:]).reshape(2, 6), Ja=np.array(ev.Ja[:]).reshape(2, 6),
                Jl=np.array(ev.Jl[:2 * e]).reshape(2, e), chi2=ev.chi2, depth_positive=bool(ev.depth_positive))


def huber(a, s):
    rho = np.zeros(3)
    _ba_lib().ov2o_huber(a, s, _p(rho, f64p))
    return rho


def schur_solve(R, E, F, row_e, row_f, Je, Jf, b, n_e, n_f, D=None):
    """generic block-sparse Schur least squares; returns (rc, S, rhs, x)."""
    L = _ba_lib()
    row_e = np.ascontiguousarray(row_e, np.int32)
    row_f = np.ascontiguousarray(row_f, np.int32)
    maxf = row_f.shape[1]
    Je, Jf, b = (np.ascontiguousarray(a, np.float64) for a in (Je, Jf, b))
    p = _BsProblemC(R, E, F, maxf, len(row_e), n_e, n_f, _p(row_e, i32p), _p(row_f, i32p), _p(Je, f64p), _p(Jf, f64p),
                    _p(b, f64p), None)
    if D is not None:
        D = np.ascontiguousarray(D, np.float64)
        p.D = _p(D, f64p)
    m = n_f * F
    S, rhs, x = np.zeros((m, m)), np.zeros(m), np.zeros(n_e * E + m)
    rc = L.ov2o_schur_solve(C.byref(p), _p(S, f64p), _p(rhs, f64p), _p(x, f64p))
    return rc, S, rhs, x


def ba_default_options(robust_mono_th=5.9915):
    from ov2slam_amd import ba_types as T
    o = T.BaOptionsC()
    _ba_lib().ov2o_ba_default_options(C.byref(o), robust_mono_th)
    return o


def ba_solve(prob, options=None):
    """Optimizer::localBA numerical core on a BaProblem (updated in place). returns BaResult."""
    from ov2slam_amd import ba_types as T
    L = _ba_lib()
    o = options if options is not None else ba_default_options()
    res = T.BaResult(prob.n_res)
    pc = prob.as_c()
    rc = L.ov2o_ba_solve(C.byref(pc), C.byref(o), C.byref(res.c))
    assert rc == 0
    return res


def pg_eval_edge(pose_i, pose_j, T_ij, want_jac=True):
    """LeftSE3RelativePoseError::Evaluate on one edge. returns (r[6], Ji[6,6], Jj[6,6])"""
    a, b, t = (np.ascontiguousarray(v, np.float64) for v in (pose_i, pose_j, T_ij))
    r, Ji, Jj = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    _ba_lib().ov2o_pg_eval_edge(_p(a, f64p), _p(b, f64p), _p(t, f64p), _p(r, f64p), _p(Ji, f64p) if want_jac else None,
                                _p(Jj, f64p) if want_jac else None)
    return r, Ji, Jj


def pg_default_options(max_iters=10, function_tolerance=1e-4):
    """the options of Optimizer::localPoseGraph (src/optimizer.cpp:2442-2446) on the trust-region defaults"""
    o = ba_default_options()
    o.max_iters, o.function_tolerance = max_iters, function_tolerance
    return o


def pose_graph_solve(prob, options=None):
    """LM solve of a PgProblem (poses updated in place). returns PgResultC"""
    from ov2slam_amd import ba_types as T
    o = options if options is not None else pg_default_options()
    res = T.PgResultC()
    pc = prob.as_c()
    assert _ba_lib().ov2o_pose_graph_solve(C.byref(pc), C.byref(o), C.byref(res)) == 0
    return res


def pnp_solve(unpx, wpts, K, Twc, scales=None, max_iters=5, chi2th=5.9915, use_robust=True, l2_after_robust=True):
    """MultiViewGeometry::ceresPnP on one frame. returns (success, Twc_out, outlier mask, (it_robust, it_l2))."""
    L = _ba_lib()
    unpx = np.ascontiguousarray(unpx, np.float64).reshape(-1, 2)
    wpts = np.ascontiguousarray(wpts, np.float64).reshape(-1, 3)
    n = len(unpx)
    K = np.ascontiguousarray(K, np.float64)
    T = np.array(Twc, np.float64).copy()
    out = np.zeros(max(n, 1), np.uint8)
    it = np.zeros(2, np.int32)
    sc = None if scales is None else np.ascontiguousarray(scales, np.int32)
    ok = L.ov2o_pnp_solve(n, _p(unpx, f64p), _p(wpts, f64p), None if sc is None else _p(sc, i32p), _p(K, f64p),
                          _p(T, f64p), max_iters, chi2th, int(use_robust), int(l2_after_robust), _p(out, u8p),
                          _p(it, i32p))
    return bool(ok), T, out[:n].astype(bool), (int(it[0]), int(it[1]))


def triangulate_pairs(T_ab, bv_a, bv_b, unpx_a, unpx_b, K_a, K_b, max_reproj_err, method=0, Twc_a=None, grp=None,
                      want_parallax=False):
    """Mapper::triangulateStereo / triangulateTemporal bodies. returns dict(pt_a, wpt, parallax, status)."""
    L = lib()
    f32p = C.POINTER(C.c_float)
    L.ov2o_triangulate_pairs.argtypes = [C.c_int, C.c_int, C.c_int, f64p, f64p, i32p, f64p, f64p, f32p, f32p, f64p, f64p,
                                         C.c_float, f64p, f64p, f64p, u8p]
    L.ov2o_triangulate_pairs.restype = C.c_int
    T_ab = np.ascontiguousarray(T_ab, np.float64).reshape(-1, 7)
    G = len(T_ab)
    bv_a, bv_b = np.ascontiguousarray(bv_a, np.float64).reshape(-1, 3), np.ascontiguousarray(bv_b, np.float64).reshape(-1, 3)
    ua, ub = np.ascontiguousarray(unpx_a, np.float32).reshape(-1, 2), np.ascontiguousarray(unpx_b, np.float32).reshape(-1, 2)
    n = len(bv_a)
    W = None if Twc_a is None else np.ascontiguousarray(Twc_a, np.float64).reshape(-1, 7)
    g = None if grp is None else np.ascontiguousarray(grp, np.int32)
    Ka, Kb = np.ascontiguousarray(K_a, np.float64), np.ascontiguousarray(K_b, np.float64)
    pt, st = np.zeros((max(n, 1), 3)), np.zeros(max(n, 1), np.uint8)
    wpt = None if W is None else np.zeros((max(n, 1), 3))
    par = np.zeros(max(n, 1)) if want_parallax else None
    rc = L.ov2o_triangulate_pairs(n, int(method), G, _p(T_ab, f64p), None if W is None else _p(W, f64p),
                                  None if g is None else _p(g, i32p), _p(bv_a, f64p), _p(bv_b, f64p), _p(ua, f32p), _p(ub, f32p),
                                  _p(Ka, f64p), _p(Kb, f64p), float(max_reproj_err), _p(pt, f64p),
                                  None if wpt is None else _p(wpt, f64p), None if par is None else _p(par, f64p), _p(st, u8p))
    if rc != 0:
        raise ValueError("pose-pair index out of range")
    return dict(pt_a=pt[:n], wpt=None if wpt is None else wpt[:n], parallax=None if par is None else par[:n], status=st[:n])


# ---------------------------------------------------------------------------------------------------
# detectors (keyframe rate)

def _det_lib():
    L = lib()
    if not getattr(L, "_det_bound", False):
        L.ov2o_detect_single_scale.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p, f64p,
                                               C.c_int, i32p, f32p]
        L.ov2o_detect_grid_fast.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p, i32p,
                                            C.c_int, i32p, f32p]
        L.ov2o_corner_subpix.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_double]
        L.ov2o_draw_disc_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint8]
        L.ov2o_min_eig_cell.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.ov2o_fast_score.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ov2o_fast_score.restype = C.c_int
        L._det_bound = True
    return L


def _roi(roi, w, h):
    return np.ascontiguousarray([0, 0, w, h] if roi is None else roi, dtype=np.int32)


def detect_single_scale(img, cell, cur_xy, dmaxquality, roi=None, subpix=True):
    """FeatureExtractor::detectSingleScale. returns (pts (n,2) f32, new dmaxquality)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cur = np.ascontiguousarray(cur_xy, np.float32).reshape(-1, 2)
    out = np.zeros(((h // cell) * (w // cell) * 2 + 2, 2), np.float32)
    n, q, r = C.c_int(0), C.c_double(dmaxquality), _roi(roi, w, h)
    _det_lib().ov2o_detect_single_scale(_p(img, u8p), w, h, w, cell, len(cur), _p(cur, f32p), _p(r, i32p), C.byref(q),
                                        int(subpix), C.byref(n), _p(out, f32p))
    return out[:n.value].copy(), q.value


def detect_grid_fast(img, cell, cur_xy, nfast_th, roi=None, subpix=True):
    """FeatureExtractor::detectGridFAST. returns (pts (n,2) f32, new nfast_th)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cur = np.ascontiguousarray(cur_xy, np.float32).reshape(-1, 2)
    out = np.zeros(((h // cell) * (w // cell) + 2, 2), np.float32)
    n, t, r = C.c_int(0), C.c_int(nfast_th), _roi(roi, w, h)
    _det_lib().ov2o_detect_grid_fast(_p(img, u8p), w, h, w, cell, len(cur), _p(cur, f32p), _p(r, i32p), C.byref(t),
                                     int(subpix), C.byref(n), _p(out, f32p))
    return out[:n.value].copy(), t.value


def corner_subpix(img, pts, half_win=3, max_iter=30, eps=0.01):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.ascontiguousarray(pts, np.float32).reshape(-1, 2).copy()
    _det_lib().ov2o_corner_subpix(_p(img, u8p), w, h, w, len(out), _p(out, f32p), half_win, max_iter, eps)
    return out


def draw_disc(mask, cx, cy, radius, value=0):
    h, w = mask.shape
    _det_lib().ov2o_draw_disc_u8(_p(mask, u8p), w, h, cx, cy, radius, value)
    return mask


def min_eig_cell(img, x0, y0, cell):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((cell, cell), np.float32)
    _det_lib().ov2o_min_eig_cell(_p(img, u8p), w, h, w, x0, y0, cell, _p(out, f32p))
    return out


def fast_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    return _det_lib().ov2o_fast_score(_p(img, u8p), img.shape[1], x, y, threshold)


# ---------------------------------------------------------------------------------------------------
# stereo matching pieces (ov2_oracle_stereo.c)

def _st_lib():
    L = lib()
    if not getattr(L, "_st_bound", False):
        L.ov2o_get_rect_sub_pix_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, u8p]
        L.ov2o_line_min_sad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
        L.ov2o_line_min_sad_img.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                            f32p, f32p]
        L.ov2o_sampson_distance.argtypes = [f64p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.ov2o_sampson_distance.restype = C.c_float
        L.ov2o_stereo_matching.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                           C.c_int, f32p, f32p, u8p, f32p, C.c_int, f64p, C.c_void_p, f32p, u8p]
        L._st_bound = True
    return L


def get_rect_sub_pix_u8(img, ww, wh, cx, cy):
    """cv::getRectSubPix 8U -> 8U"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((wh, ww), np.uint8)
    _st_lib().ov2o_get_rect_sub_pix_u8(_p(img, u8p), w, h, w, ww, wh, float(cx), float(cy), _p(out, u8p))
    return out


def line_min_sad(left, right, level, pts_xy, nwinsize=7, go_left=True):
    """FeatureTracker::getLineMinSAD on level `level` of two oracle pyramids. returns (xprior, l1err)."""
    pts = np.ascontiguousarray(pts_xy, np.float32).reshape(-1, 2)
    n = len(pts)
    xp, er = np.zeros(n, np.float32), np.zeros(n, np.float32)
    _st_lib().ov2o_line_min_sad(left._h, right._h, level, nwinsize, int(bool(go_left)), n, _p(pts, f32p), _p(xp, f32p),
                                _p(er, f32p))
    return xp, er


def line_min_sad_img(iml, imr, x, y, nwinsize=7, go_left=True):
    iml, imr = np.ascontiguousarray(iml, np.uint8), np.ascontiguousarray(imr, np.uint8)
    h, w = iml.shape
    xp, er = C.c_float(0), C.c_float(0)
    _st_lib().ov2o_line_min_sad_img(_p(iml, u8p), _p(imr, u8p), w, h, w, float(x), float(y), nwinsize, int(bool(go_left)),
                                    C.byref(xp), C.byref(er))
    return xp.value, er.value


def sampson_distance(F, l, r):
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    return float(_st_lib().ov2o_sampson_distance(_p(F, f64p), float(l[0]), float(l[1]), float(r[0]), float(r[1])))


def stereo_matching(left, right, kps_xy, prior_xy, has_prior, win=9, nlevels=3, err_th=30.0, fb_th=0.5, max_iter=30,
                    eps=0.01, lunpx=None, rectified=True, F_rl=None, right_cam=None):
    """tracking + gate of MapManager::stereoMatching on flat arrays. returns (right pixels, status)."""
    kps = np.ascontiguousarray(kps_xy, np.float32).reshape(-1, 2)
    pri = np.ascontiguousarray(prior_xy, np.float32).reshape(-1, 2)
    hp = np.ascontiguousarray(has_prior, np.uint8)
    n = len(kps)
    lu = None if lunpx is None else np.ascontiguousarray(lunpx, np.float32).reshape(-1, 2)
    F = np.zeros(9) if F_rl is None else np.ascontiguousarray(F_rl, np.float64).reshape(9)
    out, st = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8)
    _st_lib().ov2o_stereo_matching(left._h, right._h, win, nlevels, err_th, fb_th, max_iter, eps, n, _p(kps, f32p),
                                   _p(pri, f32p), _p(hp, u8p), None if lu is None else _p(lu, f32p), int(bool(rectified)),
                                   _p(F, f64p), None if right_cam is None else C.cast(C.addressof(right_cam), C.c_void_p),
                                   _p(out, f32p), _p(st, u8p))
    return out, st.astype(bool)


def cam_undistort(cam, pts):
    """CameraCalibration::undistortImagePoint for (n, 2) float32 pixels"""
    L = _st_lib()
    L.ov2o_cam_undistort.argtypes = [C.c_void_p, C.c_float, C.c_float, f32p, f32p]
    L.ov2o_cam_undistort.restype = None
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros_like(pts)
    a, b = C.c_float(), C.c_float()
    for i, (u, v) in enumerate(pts):
        L.ov2o_cam_undistort(None if cam is None else C.addressof(cam), float(u), float(v), C.byref(a), C.byref(b))
        out[i] = (a.value, b.value)
    return out


# ---------------------------------------------------------------------------------------------------
# keyframe descriptors and map matching (ov2_oracle_match.c)

def describe_brief(img, pts, pattern):
    """FeatureExtractor::describeBRIEF with a caller-supplied test table. returns (desc (n,32), valid (n,) bool)"""
    L = lib()
    L.ov2o_describe_brief.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.POINTER(C.c_int8), u8p, u8p]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    pat = np.ascontiguousarray(pattern, np.int8).reshape(256, 4)
    n = len(pts)
    desc, valid = np.zeros((max(n, 1), 32), np.uint8), np.zeros(max(n, 1), np.uint8)
    L.ov2o_describe_brief(_p(img, u8p), w, h, w, n, _p(pts, f32p), pat.ctypes.data_as(C.POINTER(C.c_int8)), _p(desc, u8p), _p(valid, u8p))
    return desc[:n], valid[:n].astype(bool)


def match_to_map(inp, fmaxprojerr=2.0, fdistratio=0.2):
    """Mapper::matchToMap on a ov2slam_amd.mapper.MatchInput. returns (match_cand, match_dist)"""
    L = lib()
    L.ov2o_match_to_map.argtypes = [C.c_void_p, C.c_float, C.c_float, i32p, f32p]
    n = inp.c.n_kp
    mc, md = np.full(max(n, 1), -1, np.int32), np.zeros(max(n, 1), np.float32)
    L.ov2o_match_to_map(C.addressof(inp.c), fmaxprojerr, fdistratio, _p(mc, i32p), _p(md, f32p))
    return mc[:n], md[:n]
