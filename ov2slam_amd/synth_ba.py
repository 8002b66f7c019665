"""Seeded synthetic local-BA windows: keyframes on a lawn-mower trajectory, EuRoC intrinsics and
stereo extrinsic (reference parameters_files/accurate/euroc/euroc_stereo.yaml:21-58), landmarks seen by up to 12
keyframes, pixel noise, gross outliers, perturbed initial state, oldest keyframes fixed.  The residual list is laid
out with the rules of Optimizer::localBA (reference src/optimizer.cpp:219-392).  numpy only."""
import numpy as np

from .ba_types import BaProblem, L_XYZ, R_XYZ, L_INV, R_INV, RANCH_INV

SEED_BA = 20211
W, H = 752, 480
K_L = np.array([458.654, 457.296, 367.215, 248.375])
K_R = np.array([457.587, 456.134, 379.999, 255.238])
BODY_T_CAM0 = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                        [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                        [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                        [0., 0., 0., 1.]])
BODY_T_CAM1 = np.array([[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
                        [0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024],
                        [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038],
                        [0., 0., 0., 1.]])


def rot_to_quat(R):
    """rotation matrix -> (x, y, z, w)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0, 0.0, 0.0, 0.0]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose7(R, t):
    return np.concatenate([t, rot_to_quat(R)])


def se3_exp(d):
    """numpy SE3 exp of [upsilon, omega] -> (R, t) (Rodrigues; independent of the oracle's C code)"""
    u, w = d[:3], d[3:]
    th = np.linalg.norm(w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + O, u.copy()
    R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th ** 2 * O @ O
    V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * O + (th - np.sin(th)) / th ** 3 * O @ O
    return R, V @ u


def stereo_extrinsic():
    T_lr = np.linalg.inv(BODY_T_CAM0) @ BODY_T_CAM1   # left <- right
    T_rl = np.linalg.inv(T_lr)
    return pose7(T_rl[:3, :3], T_rl[:3, 3]), T_rl


def make_window(n_kf=20, n_lm=2000, inv_depth=True, seed=SEED_BA, max_obs=12, px_noise=0.5, outlier_frac=0.05,
                pose_noise=(0.02, np.deg2rad(0.5)), depth_noise=0.03, fixed_frac=0.2, stereo=True, spacing=0.4,
                return_gt=False, obs_pick="nearest"):
    """obs_pick: which `max_obs` of the keyframes that see a landmark keep their observation -- 'nearest' (in index, to the
    keyframe that generated it: short tracks, the newest keyframe is covisible with a handful of others) or 'random' (tracks
    broken by occlusion: every pair of keyframes shares landmarks, so the covisibility walk of Optimizer::localBA,
    reference src/optimizer.cpp:128-190, optimises the whole window)"""
    rng = np.random.default_rng(seed)
    T_rl7, T_rl = stereo_extrinsic()
    # lawn-mower trajectory in the world XY plane, camera looking along its heading, yaw wobble +-15 deg
    per_row = max(2, int(np.ceil(np.sqrt(n_kf * 2.5))))
    Rs, ts = [], []
    for k in range(n_kf):
        row, col = divmod(k, per_row)
        x = (col if row % 2 == 0 else per_row - 1 - col) * spacing
        y = row * spacing * 2.0
        # the camera keeps looking along +x (yaw wobble only) while the rig sweeps the rows, so that every keyframe
        # shares landmarks with the fixed ones -- a reversed heading per row would split the window into
        # disconnected, gauge-free clusters, which a real covisibility window never is
        heading = np.deg2rad(15.0) * np.sin(0.9 * k)
        f = np.array([np.cos(heading), np.sin(heading), 0.0])
        d = np.array([0.0, 0.0, -1.0])
        r = np.cross(d, f)
        Rs.append(np.stack([r, d, f], axis=1))           # columns = camera x (right), y (down), z (forward)
        ts.append(np.array([x, y, 1.5 + 0.05 * np.sin(0.7 * k)]))
    Rs, ts = np.array(Rs), np.array(ts)

    def project(K, Xc):
        return np.stack([K[0] * Xc[..., 0] / Xc[..., 2] + K[2], K[1] * Xc[..., 1] / Xc[..., 2] + K[3]], -1)

    # landmarks: back-project a random pixel of a random keyframe at depth 2..25 m, then collect its observers
    lm_xyz, obs = [], []   # obs[l] = list of (kf, uv_l, uv_r or None)
    tries = 0
    while len(lm_xyz) < n_lm and tries < n_lm * 20:
        tries += 1
        k0 = int(rng.integers(n_kf))
        uv = rng.uniform([20, 20], [W - 20, H - 20])
        z = rng.uniform(2.0, 25.0)
        Xc = np.array([(uv[0] - K_L[2]) / K_L[0] * z, (uv[1] - K_L[3]) / K_L[1] * z, z])
        Xw = Rs[k0] @ Xc + ts[k0]
        Xcs = np.einsum("kji,kj->ki", Rs, Xw[None, :] - ts)       # R^T (Xw - t) for every keyframe
        ok = Xcs[:, 2] > 0.5
        px = project(K_L, np.where(ok[:, None], Xcs, [0, 0, 1.0]))
        ok &= (px[:, 0] > 5) & (px[:, 0] < W - 5) & (px[:, 1] > 5) & (px[:, 1] < H - 5) & (Xcs[:, 2] < 40.0)
        ids = np.nonzero(ok)[0]
        if len(ids) < 2:
            continue
        if len(ids) > max_obs and obs_pick == "random":
            ids = np.sort(rng.choice(ids, max_obs, replace=False))
        elif len(ids) > max_obs:   # keep the observers closest (in index) to the generating keyframe
            ids = np.sort(ids[np.argsort(np.abs(ids - k0), kind="stable")[:max_obs]])
        lst = []
        for k in ids:
            ul = px[k] + rng.normal(0, px_noise, 2)
            ur = None
            if stereo:
                Xr = T_rl[:3, :3] @ Xcs[k] + T_rl[:3, 3]
                pr = project(K_R, Xr)
                if Xr[2] > 0.5 and 5 < pr[0] < W - 5 and 5 < pr[1] < H - 5:
                    ur = pr + rng.normal(0, px_noise, 2)
            if rng.uniform() < outlier_frac:
                ul = ul + rng.uniform(5, 30, 2) * rng.choice([-1, 1], 2)
            if ur is not None and rng.uniform() < outlier_frac:
                ur = ur + rng.uniform(5, 30, 2) * rng.choice([-1, 1], 2)
            lst.append((int(k), ul, ur))
        lm_xyz.append(Xw)
        obs.append(lst)
    lm_xyz = np.array(lm_xyz)
    n_lm = len(lm_xyz)

    # initial state: GT o exp(noise) for the free keyframes, the fixed (oldest) ones stay at GT
    n_fixed = max(1, int(round(fixed_frac * n_kf)))
    pose_const = np.zeros(n_kf, np.uint8)
    pose_const[:n_fixed] = 1
    poses_gt = np.array([pose7(Rs[k], ts[k]) for k in range(n_kf)])
    poses = poses_gt.copy()
    Ri, ti = Rs.copy(), ts.copy()
    for k in range(n_fixed, n_kf):
        d = np.concatenate([rng.normal(0, pose_noise[0], 3), rng.normal(0, pose_noise[1], 3)])
        dR, dt = se3_exp(d)
        Ri[k] = dR @ Rs[k]
        ti[k] = dR @ ts[k] + dt
        poses[k] = pose7(Ri[k], ti[k])

    res_type, res_pose, res_lm, res_uv = [], [], [], []
    anchor_pose = np.zeros(n_lm, np.int32)
    anchor_uv = np.zeros((n_lm, 2))
    lm0 = np.zeros((n_lm, 1 if inv_depth else 3))
    lm_gt = np.zeros_like(lm0)
    for l, lst in enumerate(obs):
        ka, ua, ura = lst[0]                       # anchor = first observing keyframe (smallest kfid)
        anchor_pose[l] = ka
        anchor_uv[l] = ua
        Xc_gt = Rs[ka].T @ (lm_xyz[l] - ts[ka])
        scale = 1.0 + rng.normal(0, depth_noise)
        if inv_depth:
            lm0[l, 0] = 1.0 / (Xc_gt[2] * scale)
            lm_gt[l, 0] = 1.0 / Xc_gt[2]
            if ura is not None:                    # src/optimizer.cpp:269-284
                res_type.append(RANCH_INV); res_pose.append(ka); res_lm.append(l); res_uv.append(ura)
            rest = lst[1:]
        else:
            lm0[l] = Ri[ka] @ (Xc_gt * scale) + ti[ka]
            lm_gt[l] = lm_xyz[l]
            rest = lst
        for k, ul, ur in rest:
            res_type.append(L_INV if inv_depth else L_XYZ); res_pose.append(k); res_lm.append(l); res_uv.append(ul)
            if ur is not None:
                res_type.append(R_INV if inv_depth else R_XYZ); res_pose.append(k); res_lm.append(l); res_uv.append(ur)
    prob = BaProblem(K_L, K_R, T_rl7, inv_depth, poses, pose_const, lm0, anchor_pose if inv_depth else None,
                     anchor_uv if inv_depth else None, res_type, res_pose, res_lm, np.array(res_uv))
    if return_gt:
        return prob, dict(poses=poses_gt, lm=lm_gt, xyz=lm_xyz)
    return prob


def sequence_window_specs(nseq, seed, n_kf=50, n_lm=10000, spread=0.2):
    """nseq DISTINCT local-BA windows around a nominal size (one per SLAM instance of a batch): keyframes and landmarks
    drawn within +-spread of the nominal, outlier rate 1-12 %, initial pose / depth error spread around the default -- so that the
    windows of a batch differ in size, sparsity and in the LM iterations they take (initial error log-uniform over 0.1-4x the
    default: from a map that is nearly converged to one far off).  Returns make_window kwargs."""
    rng = np.random.default_rng(seed)
    specs = []
    for b in range(nseq):
        f = float(np.exp(rng.uniform(np.log(0.1), np.log(4.0))))
        specs.append(dict(n_kf=int(round(n_kf * rng.uniform(1 - spread, 1 + spread))),
                          n_lm=int(round(n_lm * rng.uniform(1 - spread, 1 + spread))), inv_depth=True, seed=int(seed + 1000 + 17 * b),
                          max_obs=7, outlier_frac=float(rng.uniform(0.01, 0.12)), pose_noise=(0.02 * f, np.deg2rad(0.5) * f),
                          depth_noise=float(np.exp(rng.uniform(np.log(0.003), np.log(0.1)))), obs_pick="random"))
    return specs


def _make_window_kw(kw):
    return make_window(**kw)


def make_windows_parallel(specs, procs):
    """make_window for every spec on a pool of SPAWNED processes (numpy only; safe to call before or after the parent has
    initialised the GPU, since the children never inherit its state)"""
    if procs <= 1 or len(specs) <= 1:
        return [make_window(**kw) for kw in specs]
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(procs, len(specs)), mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_make_window_kw, specs, chunksize=1))


def make_pnp(n=300, seed=SEED_BA, px_noise=0.5, outlier_frac=0.1, rot_pert=0.02, trans_pert=0.05, behind=0,
             with_scales=False):
    """one pose-refinement problem of VisualFrontEnd::computePose (reference src/visual_front_end.cpp:716-830):
    n world points in front of a camera at a random pose, undistorted pixels with noise + gross outliers, `behind`
    points placed behind the camera, and a perturbed initial pose (the constant-velocity prediction).
    returns dict(unpx (n,2), wpts (n,3), K (4,), Twc0 (7,), Twc_gt (7,), gt_outlier (n,) bool, scales or None)."""
    rng = np.random.default_rng(seed)
    R, t = se3_exp(np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-0.6, 0.6, 3)]))
    px = np.stack([rng.uniform(10, W - 10, n), rng.uniform(10, H - 10, n)], 1)
    z = rng.uniform(1.5, 12.0, n)
    Xc = np.stack([(px[:, 0] - K_L[2]) / K_L[0] * z, (px[:, 1] - K_L[3]) / K_L[1] * z, z], 1)
    if behind:
        Xc[:behind, 2] *= -1.0
    wpts = Xc @ R.T + t
    scales = rng.integers(0, 3, n).astype(np.int32) if with_scales else None
    sig = np.ones(n) if scales is None else 2.0 ** scales
    unpx = px + rng.normal(0, px_noise, (n, 2)) * sig[:, None]
    gt_out = np.zeros(n, bool)
    n_out = int(round(outlier_frac * n))
    if n_out:
        idx = rng.choice(n, n_out, replace=False)
        unpx[idx] += rng.uniform(15, 60, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
        gt_out[idx] = True
    gt_out[:behind] = True
    dR, dt = se3_exp(np.concatenate([rng.normal(0, trans_pert, 3), rng.normal(0, rot_pert, 3)]))
    return dict(unpx=unpx, wpts=wpts, K=K_L.copy(), Twc0=pose7(dR @ R, dR @ t + dt), Twc_gt=pose7(R, t),
                gt_outlier=gt_out, scales=scales)
