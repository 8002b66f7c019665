"""Host-side mirror of MultiViewGeometry::ceresPnP (reference include/multi_view_geometry.hpp:104,
src/multi_view_geometry.cpp:492-586) over the C ABI.  The LM loop, the chi2 flags and the L2 re-solve run inside one
HIP kernel (csrc/pnp.hip); this file only marshals arrays.  No arithmetic happens here."""
import numpy as np

from .frontend import _check


def _vp(a):
    return a.ctypes.data


class MultiViewGeometry:
    """mirror of the reference's static MultiViewGeometry helpers for the pose-refinement path."""

    def __init__(self, ctx):
        self.ctx = ctx

    def ceresPnP(self, vunkps, vwpts, Twc, nmaxiter, chi2th, buse_robust, bapply_l2_after_robust, fx, fy, cx, cy,
                 vscales=None):
        """one frame.  returns (success, Twc (7,) [t, qx qy qz qw], voutliersidx (sorted int array))."""
        ok, T, out, _ = self.ceresPnP_batch([vunkps], [vwpts], np.asarray(Twc, np.float64).reshape(1, 7), nmaxiter,
                                            chi2th, buse_robust, bapply_l2_after_robust,
                                            np.array([[fx, fy, cx, cy]], np.float64),
                                            None if vscales is None else [vscales])
        return bool(ok[0]), T[0], np.flatnonzero(out[0]).astype(np.int32)

    def triangulate_pairs(self, T_ab, bv_a, bv_b, unpx_a, unpx_b, K_a, K_b, max_reproj_err, method=0, Twc_a=None, grp=None,
                          want_parallax=False):
        """per-keypoint bodies of Mapper::triangulateStereo / triangulateTemporal (src/mapper.cpp:191-461) through
        ov2_triangulate_pairs: MultiViewGeometry::triangulate (mid-point, method 0) or the rectified disparity form
        (method 1) + depth / reprojection gates + world projection + parallax. returns dict(pt_a, wpt, parallax, status)."""
        T_ab = np.ascontiguousarray(T_ab, np.float64).reshape(-1, 7)
        bv_a, bv_b = np.ascontiguousarray(bv_a, np.float64).reshape(-1, 3), np.ascontiguousarray(bv_b, np.float64).reshape(-1, 3)
        if len(bv_a) != len(bv_b):
            raise ValueError("bv_a / bv_b sizes differ")
        ua, ub = np.ascontiguousarray(unpx_a, np.float32).reshape(-1, 2), np.ascontiguousarray(unpx_b, np.float32).reshape(-1, 2)
        n = len(bv_a)
        W = None if Twc_a is None else np.ascontiguousarray(Twc_a, np.float64).reshape(-1, 7)
        g = None if grp is None else np.ascontiguousarray(grp, np.int32)
        Ka, Kb = np.ascontiguousarray(K_a, np.float64), np.ascontiguousarray(K_b, np.float64)
        pt, st = np.zeros((max(n, 1), 3)), np.zeros(max(n, 1), np.uint8)
        wpt = None if W is None else np.zeros((max(n, 1), 3))
        par = np.zeros(max(n, 1)) if want_parallax else None
        c = self.ctx
        _check(c.h, c.lib.ov2_triangulate_pairs(c.h, n, int(method), len(T_ab), _vp(T_ab), None if W is None else _vp(W),
                                                None if g is None else _vp(g), _vp(bv_a), _vp(bv_b), _vp(ua), _vp(ub), _vp(Ka),
                                                _vp(Kb), float(max_reproj_err), _vp(pt), None if wpt is None else _vp(wpt),
                                                None if par is None else _vp(par), _vp(st)))
        return dict(pt_a=pt[:n], wpt=None if wpt is None else wpt[:n], parallax=None if par is None else par[:n], status=st[:n])

    def ceresPnP_batch_dev(self, B, d_off, d_unpx, d_wpts, d_scales, d_K, d_Twc, nmaxiter, chi2th, buse_robust,
                           bapply_l2_after_robust, d_outlier, d_removed, d_success, d_iters=None):
        """device-resident, asynchronous form (ov2_pnp_solve_batch_dev): DeviceArrays in, nothing synchronised."""
        p = lambda a: None if a is None else a.ptr
        c = self.ctx
        _check(c.h, c.lib.ov2_pnp_solve_batch_dev(c.h, int(B), p(d_off), p(d_unpx), p(d_wpts), p(d_scales), p(d_K), p(d_Twc),
                                                  int(nmaxiter), float(chi2th), int(bool(buse_robust)),
                                                  int(bool(bapply_l2_after_robust)), p(d_outlier), p(d_removed),
                                                  p(d_success), p(d_iters)))

    def ceresPnP_batch(self, unpx_list, wpts_list, Twc, nmaxiter, chi2th, buse_robust, bapply_l2_after_robust, K,
                       scales_list=None):
        """B independent frames in one launch (one workgroup each).
        returns (success (B,) bool, Twc (B,7), [outlier mask per frame], iters (B,2))."""
        B = len(unpx_list)
        n_pts = np.array([len(u) for u in unpx_list], np.int32)
        cat = lambda xs, k, dt: (np.concatenate([np.asarray(x, dt).reshape(-1, k) for x in xs]) if B and n_pts.sum()
                                 else np.zeros((0, k), dt))
        unpx = np.ascontiguousarray(cat(unpx_list, 2, np.float64))
        wpts = np.ascontiguousarray(cat(wpts_list, 3, np.float64))
        if len(wpts) != len(unpx):
            raise ValueError("vunkps.size() != vwpts.size()")       # the reference asserts (:500)
        sc = None if scales_list is None else np.ascontiguousarray(cat(scales_list, 1, np.int32).ravel())
        K = np.ascontiguousarray(np.asarray(K, np.float64).reshape(B, 4))
        T = np.ascontiguousarray(np.array(Twc, np.float64).reshape(B, 7))
        n = int(n_pts.sum())
        outl = np.zeros(max(n, 1), np.uint8)
        ok = np.zeros(max(B, 1), np.int32)
        it = np.zeros((max(B, 1), 2), np.int32)
        c = self.ctx
        _check(c.h, c.lib.ov2_pnp_solve_batch(c.h, B, _vp(n_pts) if B else None, _vp(unpx) if n else None,
                                              _vp(wpts) if n else None, None if sc is None or not n else _vp(sc),
                                              _vp(K) if B else None, _vp(T) if B else None, int(nmaxiter),
                                              float(chi2th), int(bool(buse_robust)),
                                              int(bool(bapply_l2_after_robust)), _vp(outl), _vp(ok), _vp(it)))
        off = np.concatenate([[0], np.cumsum(n_pts)])
        return ok[:B].astype(bool), T, [outl[off[b]:off[b + 1]].astype(bool) for b in range(B)], it[:B]
