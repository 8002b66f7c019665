"""Host-side mirror of Optimizer::localBA's solve stage (reference include/optimizer.hpp:42,
src/optimizer.cpp:439-735) over the C ABI: the flat BaProblem (ba_types.py) goes to ov2_ba_solve, which runs the
LM / Schur / Cholesky kernels on the MI355X and returns updated poses + landmarks, chi2 flags and the iteration log.
Plumbing only; no arithmetic happens here."""
import ctypes as C

from . import _lib
from .ba_types import BaOptionsC, BaResult
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
