"""ctypes mirror of the BA PODs in include/ov2slam_hip.h (ov2_ba_problem / ov2_ba_options / ov2_ba_result) and a
numpy-side container that owns the arrays they point at.  Shared by the product binding and the oracle binding
(types only)."""
import ctypes as C

import numpy as np

L_XYZ, R_XYZ, L_INV, R_INV, RANCH_INV = 0, 1, 2, 3, 4
MAX_LOG = 40
TERM = {0: "max_iter", 1: "function_tolerance", 2: "parameter_tolerance", 3: "gradient_tolerance", 4: "min_radius",
        5: "failure", 6: "skipped"}

dp = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)


class CamModelC(C.Structure):
    """ov2_cam_model: lens model of a camera (model 0 none, 1 radial-tangential k1 k2 p1 p2 [k3], 2 fisheye k1..k4)"""
    _fields_ = [("K", C.c_double * 4), ("model", C.c_int32), ("n_coeffs", C.c_int32), ("D", C.c_double * 5)]

    @classmethod
    def make(cls, K4, model, coeffs):
        m = cls()
        m.K[:] = [float(v) for v in K4]
        m.model = {"none": 0, "pinhole": 1, "radtan": 1, "fisheye": 2}.get(model, model)
        c = [float(v) for v in coeffs][:5]
        m.n_coeffs = len(c)
        m.D[:] = c + [0.0] * (5 - len(c))
        return m


class BaProblemC(C.Structure):
    _fields_ = [("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7),
                ("inv_depth", C.c_int32), ("n_pose", C.c_int32), ("pose", dp), ("pose_const", u8p),
                ("n_lm", C.c_int32), ("lm", dp), ("lm_anchor_pose", i32p), ("lm_anchor_uv", dp),
                ("n_res", C.c_int32), ("res_type", u8p), ("res_pose", i32p), ("res_lm", i32p), ("res_uv", dp),
                ("res_sigma", dp)]


class BaOptionsC(C.Structure):
    _fields_ = [("huber_delta", C.c_double), ("chi2_th", C.c_double), ("max_iters", C.c_int32),
                ("l2_refine", C.c_int32), ("l2_max_iters", C.c_int32), ("function_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("min_relative_decrease", C.c_double), ("parameter_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("jacobi_scaling", C.c_int32),
                ("max_consecutive_invalid_steps", C.c_int32)]


class BaIterC(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost_change", C.c_double), ("radius", C.c_double),
                ("relative_decrease", C.c_double), ("model_cost_change", C.c_double), ("step_is_valid", C.c_int32),
                ("step_is_successful", C.c_int32)]


class BaResultC(C.Structure):
    _fields_ = [("chi2", dp), ("depth_positive", u8p), ("outlier", u8p), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("l2_initial_cost", C.c_double), ("l2_final_cost", C.c_double),
                ("n_log", C.c_int32), ("n_log_robust", C.c_int32), ("termination", C.c_int32),
                ("l2_termination", C.c_int32), ("l2_done", C.c_int32), ("n_outliers_pass1", C.c_int32),
                ("n_outliers_pass2", C.c_int32), ("log", BaIterC * MAX_LOG)]


class PgProblemC(C.Structure):
    _fields_ = [("n_pose", C.c_int32), ("pose", dp), ("pose_const", u8p), ("n_edge", C.c_int32), ("edge_i", i32p),
                ("edge_j", i32p), ("T_ij", dp)]


class PgResultC(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("termination", C.c_int32), ("n_log", C.c_int32),
                ("log", BaIterC * MAX_LOG)]


class PgProblem:
    """a pose graph: poses (n x 7 Twc, solved in place), constness, edges (i, j) with the measured T_ij (7)"""

    def __init__(self, pose, pose_const, edge_i, edge_j, T_ij):
        c = np.ascontiguousarray
        self.pose = c(pose, np.float64).reshape(-1, 7).copy()
        self.pose_const = c(pose_const, np.uint8)
        self.edge_i, self.edge_j = c(edge_i, np.int32), c(edge_j, np.int32)
        self.T_ij = c(T_ij, np.float64).reshape(-1, 7)
        assert len(self.pose) == len(self.pose_const) and len(self.edge_i) == len(self.edge_j) == len(self.T_ij)

    def copy(self):
        return PgProblem(self.pose, self.pose_const, self.edge_i, self.edge_j, self.T_ij)

    def as_c(self):
        p = PgProblemC()
        p.n_pose, p.pose, p.pose_const = len(self.pose), self.pose.ctypes.data_as(dp), self.pose_const.ctypes.data_as(u8p)
        p.n_edge = len(self.edge_i)
        p.edge_i, p.edge_j = self.edge_i.ctypes.data_as(i32p), self.edge_j.ctypes.data_as(i32p)
        p.T_ij = self.T_ij.ctypes.data_as(dp)
        return p


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


class BaProblem:
    """owns the numpy arrays of one local-BA window (flat mirror of what Optimizer::localBA assembles)."""

    def __init__(self, calib_l, calib_r, T_rl, inv_depth, pose, pose_const, lm, lm_anchor_pose, lm_anchor_uv,
                 res_type, res_pose, res_lm, res_uv, res_sigma=None):
        c = np.ascontiguousarray
        self.calib_l = c(calib_l, np.float64)
        self.calib_r = c(calib_r, np.float64)
        self.T_rl = c(T_rl, np.float64)
        self.inv_depth = int(bool(inv_depth))
        self.pose = c(pose, np.float64).reshape(-1, 7).copy()
        self.pose_const = c(pose_const, np.uint8)
        self.lm = c(lm, np.float64).reshape(-1, 1 if self.inv_depth else 3).copy()
        self.lm_anchor_pose = None if lm_anchor_pose is None else c(lm_anchor_pose, np.int32)
        self.lm_anchor_uv = None if lm_anchor_uv is None else c(lm_anchor_uv, np.float64).reshape(-1, 2)
        self.res_type = c(res_type, np.uint8)
        self.res_pose = c(res_pose, np.int32)
        self.res_lm = c(res_lm, np.int32)
        self.res_uv = c(res_uv, np.float64).reshape(-1, 2)
        self.res_sigma = None if res_sigma is None else c(res_sigma, np.float64)
        assert len(self.pose_const) == len(self.pose)
        assert len(self.res_type) == len(self.res_pose) == len(self.res_lm) == len(self.res_uv)

    def copy(self):
        return BaProblem(self.calib_l, self.calib_r, self.T_rl, self.inv_depth, self.pose, self.pose_const, self.lm,
                         self.lm_anchor_pose, self.lm_anchor_uv, self.res_type, self.res_pose, self.res_lm, self.res_uv,
                         self.res_sigma)

    @property
    def n_res(self):
        return len(self.res_type)

    def as_c(self):
        p = BaProblemC()
        p.calib_l[:] = self.calib_l.tolist()
        p.calib_r[:] = self.calib_r.tolist()
        p.T_rl[:] = self.T_rl.tolist()
        p.inv_depth = self.inv_depth
        p.n_pose, p.pose, p.pose_const = len(self.pose), _ptr(self.pose, dp), _ptr(self.pose_const, u8p)
        p.n_lm, p.lm = len(self.lm), _ptr(self.lm, dp)
        p.lm_anchor_pose, p.lm_anchor_uv = _ptr(self.lm_anchor_pose, i32p), _ptr(self.lm_anchor_uv, dp)
        p.n_res = self.n_res
        p.res_type, p.res_pose, p.res_lm = _ptr(self.res_type, u8p), _ptr(self.res_pose, i32p), _ptr(self.res_lm, i32p)
        p.res_uv, p.res_sigma = _ptr(self.res_uv, dp), _ptr(self.res_sigma, dp)
        return p


class BaResult:
    def __init__(self, n_res):
        self.chi2 = np.zeros(n_res, np.float64)
        self.depth_positive = np.zeros(n_res, np.uint8)
        self.outlier = np.zeros(n_res, np.uint8)
        self.c = BaResultC()
        self.c.chi2, self.c.depth_positive, self.c.outlier = _ptr(self.chi2, dp), _ptr(self.depth_positive, u8p), _ptr(self.outlier, u8p)

    @property
    def log(self):
        return [dict(cost=i.cost, cost_change=i.cost_change, radius=i.radius, relative_decrease=i.relative_decrease,
                     model_cost_change=i.model_cost_change, valid=bool(i.step_is_valid), ok=bool(i.step_is_successful))
                for i in self.c.log[:self.c.n_log]]

    def summary(self):
        c = self.c
        return dict(initial_cost=c.initial_cost, final_cost=c.final_cost, termination=TERM.get(c.termination),
                    l2_done=bool(c.l2_done), l2_initial_cost=c.l2_initial_cost, l2_final_cost=c.l2_final_cost,
                    l2_termination=TERM.get(c.l2_termination), outliers=(c.n_outliers_pass1, c.n_outliers_pass2),
                    iterations=(c.n_log_robust - 1, c.n_log - c.n_log_robust - (1 if c.l2_done else 0)))
