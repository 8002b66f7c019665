"""Host-side mirror of Optimizer::localBA's solve stage (reference include/optimizer.hpp:42,
src/optimizer.cpp:439-735) over the C ABI: the flat BaProblem (ba_types.py) goes to ov2_ba_solve, which runs the
LM / Schur / Cholesky kernels on the MI355X and returns updated poses + landmarks, chi2 flags and the iteration log.
Plumbing only; no arithmetic happens here."""
import ctypes as C

from . import _lib
import numpy as np

from .ba_types import BaOptionsC, BaProblemC, BaResult, BaResultC, dp, i32p, u8p
from .frontend import _check


def default_options(robust_mono_th=5.9915):
    """the ceres::Solver::Options Optimizer::localBA sets (src/optimizer.cpp:439-468) + Ceres defaults."""
    o = BaOptionsC()
    _lib.load().ov2_ba_default_options(C.byref(o), robust_mono_th)
    return o


class Optimizer:
    """mirror of the reference Optimizer for the localBA path."""

    def __init__(self, ctx, robust_mono_th=5.9915, apply_l2_after_robust=True):
        self.ctx = ctx
        self.options = default_options(robust_mono_th)
        self.options.l2_refine = int(bool(apply_l2_after_robust))

    def localBA(self, problem, buse_robust_cost=True, options=None):
        """solves `problem` (BaProblem) in place; returns BaResult (flags, costs, iteration log)."""
        o = options if options is not None else self.options
        if not buse_robust_cost:
            o = BaOptionsC.from_buffer_copy(o)
            o.huber_delta = 0.0
        res = BaResult(problem.n_res)
        pc = problem.as_c()
        _check(self.ctx.h, self.ctx.lib.ov2_ba_solve(self.ctx.h, C.byref(pc), C.byref(o), C.byref(res.c)))
        return res

    def localBA_batch(self, problems, buse_robust_cost=True, options=None, want_flags=True):
        """B independent windows in one call (ov2_ba_solve_batch): every BaProblem is solved in place; returns the list of
        BaResult.  The reference runs one Estimator thread per SLAM instance; this is that, for the windows pending at the
        same time on one GPU.  want_flags=False skips the per-residual outputs (chi2 / depth / outlier arrays)."""
        o = options if options is not None else self.options
        if not buse_robust_cost:
            o = BaOptionsC.from_buffer_copy(o)
            o.huber_delta = 0.0
        B = len(problems)
        res = [BaResult(p.n_res) for p in problems]
        pcs = (BaProblemC * B)(*[p.as_c() for p in problems])
        rcs = (BaResultC * B)()
        for k, r in enumerate(res):
            if want_flags:
                rcs[k].chi2, rcs[k].depth_positive, rcs[k].outlier = r.c.chi2, r.c.depth_positive, r.c.outlier
        _check(self.ctx.h, self.ctx.lib.ov2_ba_solve_batch(self.ctx.h, B, pcs, C.byref(o), rcs))
        for k, r in enumerate(res):
            keep = (r.c.chi2, r.c.depth_positive, r.c.outlier)
            C.memmove(C.byref(r.c), C.byref(rcs[k]), C.sizeof(BaResultC))
            r.c.chi2, r.c.depth_positive, r.c.outlier = keep
        return res

    def localBA_batch_dev(self, dev_problems, buse_robust_cost=True, options=None):
        """ov2_ba_solve_batch_dev: the windows (DeviceBaProblem) and their per-residual outputs stay in device memory;
        returns the list of BaResultC (scalars + iteration logs; the flags are in each problem's device arrays)."""
        o = options if options is not None else self.options
        if not buse_robust_cost:
            o = BaOptionsC.from_buffer_copy(o)
            o.huber_delta = 0.0
        B = len(dev_problems)
        pcs = (BaProblemC * B)(*[p.c for p in dev_problems])
        rcs = (BaResultC * B)()
        for k, p in enumerate(dev_problems):
            rcs[k].chi2 = C.cast(p.chi2.ptr, dp)
            rcs[k].depth_positive = C.cast(p.depth_positive.ptr, u8p)
            rcs[k].outlier = C.cast(p.outlier.ptr, u8p)
        _check(self.ctx.h, self.ctx.lib.ov2_ba_solve_batch_dev(self.ctx.h, B, pcs, C.byref(o), rcs))
        return list(rcs)


class DeviceBaProblem:
    """device-resident copy of a BaProblem (what a device-side producer such as the map mirror would leave in HBM):
    the arrays ov2_ba_problem points at, uploaded once, + device arrays for the per-residual outputs."""

    def __init__(self, ctx, P):
        self.ctx, self.host = ctx, P
        self.pose0, self.lm0 = P.pose.copy(), P.lm.copy()
        up = ctx.to_device
        self.pose, self.pose_const, self.lm = up(P.pose), up(P.pose_const), up(P.lm)
        self.lm_anchor_pose = None if P.lm_anchor_pose is None else up(P.lm_anchor_pose)
        self.lm_anchor_uv = None if P.lm_anchor_uv is None else up(P.lm_anchor_uv)
        self.res_type, self.res_pose, self.res_lm, self.res_uv = up(P.res_type), up(P.res_pose), up(P.res_lm), up(P.res_uv)
        self.res_sigma = None if P.res_sigma is None else up(P.res_sigma)
        n = P.n_res
        self.chi2 = ctx.empty(max(n, 1), np.float64)
        self.depth_positive = ctx.empty(max(n, 1), np.uint8)
        self.outlier = ctx.empty(max(n, 1), np.uint8)
        c = P.as_c()
        cast = lambda a, t: C.cast(a.ptr, t) if a is not None else t()
        c.pose, c.pose_const, c.lm = cast(self.pose, dp), cast(self.pose_const, u8p), cast(self.lm, dp)
        c.lm_anchor_pose, c.lm_anchor_uv = cast(self.lm_anchor_pose, i32p), cast(self.lm_anchor_uv, dp)
        c.res_type, c.res_pose, c.res_lm = cast(self.res_type, u8p), cast(self.res_pose, i32p), cast(self.res_lm, i32p)
        c.res_uv, c.res_sigma = cast(self.res_uv, dp), cast(self.res_sigma, dp)
        self.c = c

    def reset(self):
        """state back to what the window was created with (a fresh window for the next solve)"""
        self.pose.set(self.pose0)
        self.lm.set(self.lm0)

    def download(self):
        """(pose, lm, chi2, depth_positive, outlier) as numpy arrays"""
        n = self.host.n_res
        return (self.pose.get(), self.lm.get(), self.chi2.get()[:n], self.depth_positive.get()[:n], self.outlier.get()[:n])
