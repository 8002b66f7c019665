"""Host-side mirror of the Ceres problems of Optimizer::localPoseGraph / fullPoseGraph (reference
src/optimizer.cpp:2346-2592, 2783-2870) over the C ABI: a PgProblem (ba_types.py) goes to ov2_pose_graph_solve, which runs
the whole LM loop on the MI355X.  Plumbing only."""
import ctypes as C

from . import _lib
from .ba_types import BaOptionsC, PgResultC
from .frontend import _check


def default_options(max_iters=10, function_tolerance=1e-4):
    """options.max_num_iterations / function_tolerance of localPoseGraph (:2445-2446) on the Ceres trust-region defaults;
    fullPoseGraph: 100, 1e-6 (:2821-2824)"""
    o = BaOptionsC()
    _lib.load().ov2_ba_default_options(C.byref(o), 5.9915)
    o.max_iters, o.function_tolerance = max_iters, function_tolerance
    return o


def solve(ctx, problem, options=None):
    """solves `problem` (PgProblem) in place; returns PgResultC (costs, termination, iteration log)"""
    o = options if options is not None else default_options()
    res = PgResultC()
    pc = problem.as_c()
    _check(ctx.h, ctx.lib.ov2_pose_graph_solve(ctx.h, C.byref(pc), C.byref(o), C.byref(res)))
    return res
