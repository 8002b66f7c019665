"""A closed-loop stereo SLAM run over the C ABI: the SlamManager-like sequence of the reference
(src/ov2slam.cpp:152-250 run(): VisualFrontEnd::visualTracking per frame, Mapper::run + Estimator::applyLocalBA per
keyframe) with every arithmetic stage behind a `backend` --

    per frame     preprocess (CLAHE + pyramid)                    VisualFrontEnd::preprocessImage   :1143-1177
                  kltTracking with motion-model priors            VisualFrontEnd::kltTracking       :132-275
                  ceresPnP on the 3-D keypoints                    VisualFrontEnd::computePose       :657-830
    per keyframe  grid detector on the free cells                 MapManager::extractKeypoints      src/map_manager.cpp:286-340
                  stereo matching (3-D / SAD priors, row gate)    MapManager::stereoMatching        :367-611
                  stereo triangulation of the 2-D keypoints       Mapper::triangulateStereo         src/mapper.cpp:346-461
                  local BA on the last keyframes                  Optimizer::localBA                src/optimizer.cpp:34-897

HipBackend runs those stages on the MI355X through libov2hip.so; tests drive the SAME loop with a backend made of the
CPU oracle's functions and compare the two trajectories pose by pose (SURVEY.md 8 row g: there is no EuRoC data, so
this synthetic sequence with exact ground truth stands in for "results match the reference CPU path on identical
frames").  The loop itself is bookkeeping (ids, lists, the flat BA problem); no arithmetic of the path happens here.
Simplifications against the reference, all deterministic: a keyframe every `kf_every` frames instead of the parallax /
track-count heuristics of checkNewKfReq; the local-BA window is the last `ba_window` keyframes (oldest `ba_fixed`
constant) instead of the covisibility walk (tested separately, tests/test_map_gpu.py); no P3P / map tracking / loop closing."""
import numpy as np

from . import synth_ba
from .ba_types import BaProblem, L_INV, R_INV, RANCH_INV


def pose_inv(T):
    R = synth_ba.quat_to_rot(T[3:])
    return synth_ba.pose7(R.T, -R.T @ T[:3])


def pose_mul(A, B):
    Ra, Rb = synth_ba.quat_to_rot(A[3:]), synth_ba.quat_to_rot(B[3:])
    return synth_ba.pose7(Ra @ Rb, Ra @ B[:3] + A[:3])


def project(K4, Tcw, X):
    """pinhole projection as CameraCalibration::projectCamToImage (src/camera_calibration.cpp:243-252): float32 pixels"""
    R, t = synth_ba.quat_to_rot(Tcw[3:]), Tcw[:3]
    pc = np.asarray(X, np.float64) @ R.T + t
    invz = 1.0 / pc[:, 2]
    return np.stack([K4[0] * (pc[:, 0] * invz) + K4[2], K4[1] * (pc[:, 1] * invz) + K4[3]], -1).astype(np.float32), pc[:, 2]


class HipBackend:
    """the stages on the GPU (C ABI)"""

    def __init__(self, ctx, cell=35, dmaxquality=0.001):
        from . import frontend as fe, local_ba
        from .multi_view_geometry import MultiViewGeometry
        self.fe, self.ctx = fe, ctx
        self.trk = fe.FeatureTracker(ctx, 30, 0.01)
        self.ext = fe.FeatureExtractor(ctx, nmaxdist=cell, dmaxquality=dmaxquality)
        self.mvg = MultiViewGeometry(ctx)
        self.opt = local_ba.Optimizer(ctx)

    def preprocess(self, img):
        return self.fe.preprocess_image(self.ctx, img)

    def klt_tracking(self, prev, cur, kps, priors, has):
        xy, st, _ = self.trk.kltTracking(prev, cur, 9, 3, 30.0, 0.5, kps, priors, has)
        return xy, st

    def pnp(self, unpx, wpts, Twc, K4):
        # the reference's signature takes the intrinsics as float (include/multi_view_geometry.hpp:88-93)
        return self.mvg.ceresPnP(unpx, wpts, Twc, 5, 5.9915, True, True, *np.float32(K4).astype(np.float64))

    def detect(self, pyr, img, cur_kps):
        return self.ext.detectSingleScale(pyr, cur_kps)

    def line_min_sad(self, lpyr, rpyr, pts):
        return self.trk.getLineMinSAD(lpyr, rpyr, 3, pts, 7, True)[0]

    def stereo(self, lpyr, rpyr, kps, priors, has):
        return self.trk.stereoMatching(lpyr, rpyr, 9, 3, 30.0, 0.5, kps, priors, has, rectified=True)

    def triangulate(self, T_lr, bv_l, bv_r, ul, ur, K4, Twc):
        r = self.mvg.triangulate_pairs(T_lr, bv_l, bv_r, ul, ur, K4, K4, 3.0, method=0, Twc_a=Twc)
        return r["wpt"], r["status"]

    def ba(self, problem):
        return self.opt.localBA(problem)


class SlamLoop:
    def __init__(self, backend, K4, baseline, w, h, kf_every=5, ba_window=8, ba_fixed=2, cell=35):
        self.b, self.K, self.base, self.w, self.h = backend, np.asarray(K4, np.float64), float(baseline), w, h
        self.kf_every, self.ba_window, self.ba_fixed, self.cell = kf_every, ba_window, ba_fixed, cell
        self.T_lr = np.array([baseline, 0, 0, 0, 0, 0, 1.0])                  # right camera in the left frame
        self.T_rl = np.array([-baseline, 0, 0, 0, 0, 0, 1.0])
        self.prev_pyr = None
        self.kps = {}            # lmid -> float32 (2,) pixel in the current left image
        self.lms = {}            # lmid -> world point (3,) for 3-D landmarks
        self.next_id = 0
        self.Twc = np.array([0, 0, 0, 0, 0, 0, 1.0])
        self.Twc_prev = None
        self.kfs = []            # dict(kfid, Twc, obs {lmid: (ul (2,), ur (2,) or None)})
        self.traj = []
        self.stats = []
        self.kps_log = []        # ids of the keypoints in the current frame at the end of every step

    # ---- helpers
    def _bv(self, px):
        px = np.asarray(px, np.float64).reshape(-1, 2)
        v = np.stack([(px[:, 0] - self.K[2]) / self.K[0], (px[:, 1] - self.K[3]) / self.K[1], np.ones(len(px))], -1)
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    def _in_image(self, p):
        return (p[:, 0] >= 0) & (p[:, 1] >= 0) & (p[:, 0] < self.w) & (p[:, 1] < self.h)

    # ---- one frame
    def step(self, t, img_left, img_right_fn):
        b = self.b
        cur_pyr = b.preprocess(img_left)
        st = dict(frame=t, tracked=0, n3d=0, pnp_out=0, kf=False)
        if self.prev_pyr is not None and self.kps:
            # motion model: constant velocity on the last two poses (src/visual_front_end.cpp:95-106)
            Tpred = self.Twc if self.Twc_prev is None else pose_mul(self.Twc, pose_mul(pose_inv(self.Twc_prev), self.Twc))
            ids = sorted(self.kps)
            kps = np.float32([self.kps[i] for i in ids])
            pri, has = kps.copy(), np.zeros(len(ids), np.uint8)
            i3 = [k for k, i in enumerate(ids) if i in self.lms]
            if i3:
                proj, z = project(self.K, pose_inv(Tpred), np.array([self.lms[ids[k]] for k in i3]))
                ok = self._in_image(proj) & (z > 0)
                for k, p, o in zip(i3, proj, ok):
                    if o:
                        pri[k], has[k] = p, 1
            xy, stt = b.klt_tracking(self.prev_pyr, cur_pyr, kps, pri, has)
            self.kps = {i: xy[k].copy() for k, i in enumerate(ids) if stt[k]}
            st["tracked"] = len(self.kps)
            # pose from the 3-D keypoints (ceresPnP, src/visual_front_end.cpp:791)
            i3 = [i for i in sorted(self.kps) if i in self.lms]
            st["n3d"] = len(i3)
            self.Twc_prev = self.Twc
            if len(i3) >= 4:
                unpx = np.float64([self.kps[i] for i in i3])
                wpts = np.array([self.lms[i] for i in i3])
                ok, T, outl = b.pnp(unpx, wpts, Tpred, self.K)
                if ok:
                    self.Twc = np.array(T)
                    for k in outl:                    # outliers leave the frame (:806-812)
                        self.kps.pop(i3[int(k)], None)
                    st["pnp_out"] = len(outl)
                else:
                    self.Twc = Tpred
            else:
                self.Twc = Tpred
        if t % self.kf_every == 0:
            self._keyframe(t, cur_pyr, img_left, img_right_fn(t), st)
        self.prev_pyr = cur_pyr
        self.traj.append(self.Twc.copy())
        self.stats.append(st)
        self.kps_log.append(sorted(self.kps))
        return self.Twc

    # ---- keyframe: detect, stereo match, triangulate, local BA
    def _keyframe(self, t, lpyr, img_left, img_right, st):
        b = self.b
        st["kf"] = True
        cur = np.float32([self.kps[i] for i in sorted(self.kps)]).reshape(-1, 2)
        new = b.detect(lpyr, img_left, cur)
        for p in new:
            self.kps[self.next_id] = np.float32(p)
            self.next_id += 1
        rpyr = b.preprocess(img_right)
        ids = sorted(self.kps)
        kps = np.float32([self.kps[i] for i in ids])
        pri, has = kps.copy(), np.zeros(len(ids), np.uint8)
        Tcw = pose_inv(self.Twc)
        i3 = [k for k, i in enumerate(ids) if i in self.lms]
        if i3:                                                      # 3-D keypoints: reprojection into the right camera
            proj, z = project(self.K, pose_mul(self.T_rl, Tcw), np.array([self.lms[ids[k]] for k in i3]))
            ok = self._in_image(proj) & (z > 0)
            for k, p, o in zip(i3, proj, ok):
                if o:
                    pri[k], has[k] = p, 1
        i2 = [k for k in range(len(ids)) if not has[k]]
        if i2:                                                      # rectified rig: SAD prior on the coarsest level (:419-436)
            xp = b.line_min_sad(lpyr, rpyr, (kps[i2] * np.float32(0.125)).astype(np.float32))
            for k, x in zip(i2, xp):
                x = np.float32(x * np.float32(8.0))
                if 0 <= x <= kps[k, 0]:
                    pri[k, 0] = x
        rxy, sst = b.stereo(lpyr, rpyr, kps, pri, has)
        obs = {i: (kps[k].copy(), rxy[k].copy() if sst[k] else None) for k, i in enumerate(ids)}
        # triangulate the stereo keypoints that have no 3-D point yet (Mapper::triangulateStereo)
        cand = [k for k, i in enumerate(ids) if sst[k] and i not in self.lms]
        if cand:
            ul, ur = kps[cand], rxy[cand]
            wpt, tst = b.triangulate(self.T_lr, self._bv(ul), self._bv(ur), ul, ur, self.K, self.Twc)
            for k, X, s in zip(cand, wpt, tst):
                if s == 0:
                    self.lms[ids[k]] = np.array(X)
        st["stereo"], st["new_kps"], st["n_lm"] = int(np.sum(sst)), len(new), len(self.lms)
        self.kfs.append(dict(kfid=len(self.kfs), frame=t, Twc=self.Twc.copy(), obs=obs))
        if len(self.kfs) >= 2:
            self._local_ba(st)

    def _local_ba(self, st):
        """flat problem of the last keyframes exactly as Optimizer::localBA lays it out (anchored inverse depth,
        src/optimizer.cpp:219-392), solve, write back (:741-882: poses, landmarks, outlier observations removed)"""
        win = self.kfs[-self.ba_window:]
        nfix = self.ba_fixed if len(win) > self.ba_fixed else 1
        poses = np.array([k["Twc"] for k in win])
        const = np.zeros(len(win), np.uint8)
        const[:nfix] = 1
        lm_ids = sorted({i for k in win for i in k["obs"] if i in self.lms})
        lm_index, lm_par, lm_anch, lm_auv = {}, [], [], []
        rt, rp, rl, ruv, rkey = [], [], [], [], []
        for i in lm_ids:
            seen = [(p, k["obs"][i]) for p, k in enumerate(win) if i in k["obs"]]
            if len(seen) < 2 and seen[0][1][1] is None:
                continue                                            # a single mono observation constrains nothing
            pa, (ua, _) = seen[0]
            Ta = pose_inv(win[pa]["Twc"])
            z = (synth_ba.quat_to_rot(Ta[3:]) @ self.lms[i] + Ta[:3])[2]
            if not z > 0:
                continue
            l = len(lm_par)
            lm_index[i] = l
            lm_par.append(1.0 / z); lm_anch.append(pa); lm_auv.append(np.float64(ua))
            for p, (ul, ur) in seen:
                if p == pa:
                    if ur is not None:
                        rt.append(RANCH_INV); rp.append(p); rl.append(l); ruv.append(np.float64(ur)); rkey.append((p, i, 1))
                else:
                    rt.append(L_INV); rp.append(p); rl.append(l); ruv.append(np.float64(ul)); rkey.append((p, i, 0))
                    if ur is not None:
                        rt.append(R_INV); rp.append(p); rl.append(l); ruv.append(np.float64(ur)); rkey.append((p, i, 1))
        if not rt:
            return
        P = BaProblem(self.K, self.K, self.T_rl, True, poses, const, np.array(lm_par).reshape(-1, 1), np.int32(lm_anch),
                      np.array(lm_auv).reshape(-1, 2), np.uint8(rt), np.int32(rp), np.int32(rl), np.array(ruv).reshape(-1, 2))
        R = self.b.ba(P)
        st["ba"] = dict(n_res=P.n_res, iters=R.summary()["iterations"], cost=(R.c.initial_cost, R.c.l2_final_cost if R.c.l2_done else R.c.final_cost),
                        outliers=int((R.outlier > 0).sum()))
        for p, k in enumerate(win):
            if not const[p]:
                k["Twc"] = P.pose[p].copy()
        for i, l in lm_index.items():                               # landmark back to world coordinates through its anchor
            pa = int(P.lm_anchor_pose[l])
            Twa = win[pa]["Twc"] if const[pa] else P.pose[pa]
            zi = 1.0 / P.lm[l, 0]
            ua = P.lm_anchor_uv[l]
            pc = np.array([(ua[0] - self.K[2]) / self.K[0] * zi, (ua[1] - self.K[3]) / self.K[1] * zi, zi])
            self.lms[i] = synth_ba.quat_to_rot(Twa[3:]) @ pc + Twa[:3]
        for j in np.flatnonzero(R.outlier):                         # removeMapPointObs of the flagged observations
            p, i, is_right = rkey[int(j)]
            ul, ur = win[p]["obs"].get(i, (None, None))
            if ul is None:
                continue
            if is_right:
                win[p]["obs"][i] = (ul, None)
            else:
                del win[p]["obs"][i]
                if win[p] is self.kfs[-1]:
                    self.kps.pop(i, None)
        self.Twc = self.kfs[-1]["Twc"].copy()


def write_tum(path, poses, dt=0.05):
    """TUM trajectory format of the reference's logger (include/logger.hpp:135-160): timestamp tx ty tz qx qy qz qw"""
    with open(path, "w") as f:
        for k, T in enumerate(poses):
            f.write(f"{k * dt:.6f} " + " ".join(f"{v:.9f}" for v in T) + "\n")


def ate_rmse(poses, gt):
    """absolute trajectory error (translation RMSE, m) -- both trajectories start at the identity, no alignment"""
    d = np.array([p[:3] for p in poses]) - np.array([g[:3] for g in gt])
    return float(np.sqrt((d * d).sum(1).mean()))
