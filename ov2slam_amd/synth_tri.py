"""synthetic two-view scenes for the triangulation tests / smoke: stereo rig or temporal pairs, pinhole cameras."""
import numpy as np

from . import synth_ba


def make_pairs(n, seed=0, G=1, noise_px=0.3, rectified=False, outlier_frac=0.1):
    """returns dict: T_ab (G,7), Twc_a (G,7), grp (n,), bv_a, bv_b (n,3), unpx_a, unpx_b (n,2 float32), K_a, K_b (4,),
    X_a (n,3) ground-truth points in view a (noise-free ones triangulate back to it)."""
    rng = np.random.default_rng(seed)
    K = np.array([458.654, 457.296, 367.215, 248.375])
    T_ab, Twc = np.zeros((G, 7)), np.zeros((G, 7))
    for g in range(G):
        if rectified:
            R, t = np.eye(3), np.array([0.11, 0.0, 0.0])
        else:
            R, _ = synth_ba.se3_exp(np.concatenate([np.zeros(3), rng.normal(0, 0.03, 3)]))
            t = np.array([0.11, 0.0, 0.0]) + rng.normal(0, 0.02, 3) if G == 1 else rng.normal(0, 0.25, 3)
        T_ab[g] = synth_ba.pose7(R, t)
        Rw, tw = synth_ba.se3_exp(rng.normal(0, 0.3, 6))
        Twc[g] = synth_ba.pose7(Rw, tw)
    grp = rng.integers(0, G, n).astype(np.int32)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(1.5, 12, n)], 1)
    n_bad = int(outlier_frac * n)
    X[:n_bad // 2, 2] = rng.uniform(-4, 0.05, n_bad // 2)          # behind / too close to view a
    ua, ub, fa, fb = np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(n):
        R = synth_ba.quat_to_rot(T_ab[grp[i], 3:])
        Xb = R.T @ (X[i] - T_ab[grp[i], :3])
        ua[i] = K[:2] * X[i, :2] / X[i, 2] + K[2:]
        ub[i] = K[:2] * Xb[:2] / Xb[2] + K[2:]
    ua += rng.normal(0, noise_px, ua.shape)
    ub += rng.normal(0, noise_px, ub.shape)
    ub[n_bad // 2:n_bad] += rng.uniform(8, 30, (n_bad - n_bad // 2, 2))   # mismatches: reprojection gate
    ua32, ub32 = ua.astype(np.float32), ub.astype(np.float32)
    for i in range(n):   # Keypoint::bv_ = iK * (unpx, 1), normalised (src/frame.cpp computeKeypoint)
        fa[i] = [(ua32[i, 0] - K[2]) / K[0], (ua32[i, 1] - K[3]) / K[1], 1.0]
        fb[i] = [(ub32[i, 0] - K[2]) / K[0], (ub32[i, 1] - K[3]) / K[1], 1.0]
    fa /= np.linalg.norm(fa, axis=1, keepdims=True)
    fb /= np.linalg.norm(fb, axis=1, keepdims=True)
    return dict(T_ab=T_ab, Twc_a=Twc, grp=grp, bv_a=fa, bv_b=fb, unpx_a=ua32, unpx_b=ub32, K_a=K, K_b=K.copy(), X_a=X, n_bad=n_bad)
