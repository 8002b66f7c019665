"""Host-side mirror of Optimizer::localBA's solve stage (reference include/optimizer.hpp:42,
src/optimizer.cpp:439-735) over the C ABI: the flat BaProblem (ba_types.py) goes to ov2_ba_solve, which runs the
LM / Schur / Cholesky kernels on the MI355X and returns updated poses + landmarks, chi2 flags and the iteration log.
Plumbing only; no arithmetic happens here."""
import ctypes as C

from . import _lib
from .ba_types import BaOptionsC, BaProblemC, BaResult, BaResultC
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
