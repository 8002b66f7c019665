import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe, local_ba, synth_ba
from oracle import oracle_py as O
ctx = fe.Context(0)
P = synth_ba.make_window(6, 80, inv_depth=True, seed=23)
Pc = P.copy()
Rg = local_ba.Optimizer(ctx).localBA(P)
Rc = O.ba_solve(Pc)
print(Rg.summary()); print(Rc.summary())
bad = np.flatnonzero((np.abs(P.lm - Pc.lm) > 1e-4 * np.maximum(np.abs(Pc.lm), 1e-3)).ravel())
print("bad lms", bad, P.lm[bad].ravel(), Pc.lm[bad].ravel())
for l in bad[:5]:
    rows = np.flatnonzero(P.res_lm == l)
    print(l, "rows", rows, "types", P.res_type[rows], "outlier g", Rg.outlier[rows], "c", Rc.outlier[rows], "chi2 g", Rg.chi2[rows], "c", Rc.chi2[rows])
print("outlier equal", np.array_equal(Rg.outlier, Rc.outlier), (Rg.outlier != Rc.outlier).sum())
for a, b in zip(Rg.log, Rc.log):
    print(a["cost"], b["cost"], a["radius"], b["radius"], a["ok"], b["ok"])
