"""Synthetic 3-D scene for closed-loop runs (there is no EuRoC data here or on the GPU box): a textured plane seen by a
rectified stereo rig (EuRoC-like intrinsics, 11 cm baseline) that moves on a smooth known trajectory.  Every image is an
exact perspective rendering of the plane (ray / plane intersection + bilinear texture lookup), so optical flow, stereo
disparity, triangulated depth and the camera trajectory all have analytic ground truth.  numpy only, seeded."""
import numpy as np

from . import synth, synth_ba

W, H = synth.IMG_W, synth.IMG_H
K4 = np.array([458.654, 457.296, 367.215, 248.375])
BASELINE = 0.110074


def pose_inv(T):
    R = synth_ba.quat_to_rot(T[3:])
    return synth_ba.pose7(R.T, -R.T @ T[:3])


def pose_mul(A, B):
    Ra, Rb = synth_ba.quat_to_rot(A[3:]), synth_ba.quat_to_rot(B[3:])
    return synth_ba.pose7(Ra @ Rb, Ra @ B[:3] + A[:3])


class PlaneScene:
    """plane z = depth (world frame = first left camera), texture = band-limited noise, `n` frames"""

    def __init__(self, n=200, depth=5.0, seed=synth.SEED_IMG, tilt=0.08):
        self.n, self.depth = n, depth
        self.tex = synth.base_texture(2400, 3200, seed)
        self.s = K4[0] / depth * 1.0            # texture px per metre on the plane: ~1 texture px per image px
        self.c = np.array([1600.0, 1200.0])
        # plane normal slightly tilted so that depth varies over the image (n . X = d)
        nrm = np.array([np.sin(tilt), 0.0, np.cos(tilt)])
        self.nrm, self.d = nrm / np.linalg.norm(nrm), depth * np.cos(tilt)
        # in-plane axes for the texture lookup
        self.ax = np.cross([0.0, 1.0, 0.0], self.nrm); self.ax /= np.linalg.norm(self.ax)
        self.ay = np.cross(self.nrm, self.ax)
        self.o = self.nrm * self.d              # plane point closest to the world origin

    def pose(self, t):
        """Twc of the left camera at frame t: [t, qx qy qz qw]; identity at t = 0"""
        a = 0.035 * t
        tr = np.array([0.45 * np.sin(a), 0.18 * (1 - np.cos(0.7 * a)), 0.25 * np.sin(0.5 * a)])
        yaw, pitch = 0.06 * np.sin(0.8 * a), 0.03 * np.sin(0.6 * a)
        Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        return synth_ba.pose7(Ry @ Rx, tr)

    def _render(self, R, c):
        ys, xs = np.mgrid[0:H, 0:W]
        rays = np.stack([(xs - K4[2]) / K4[0], (ys - K4[3]) / K4[1], np.ones_like(xs, np.float64)], -1).reshape(-1, 3) @ R.T
        lam = (self.d - self.nrm @ c) / (rays @ self.nrm)
        P = c[None, :] + lam[:, None] * rays - self.o[None, :]
        u, v = self.c[0] + self.s * (P @ self.ax), self.c[1] + self.s * (P @ self.ay)
        img = synth._bilinear(self.tex, u, v).reshape(H, W)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def left(self, t):
        T = self.pose(t)
        return self._render(synth_ba.quat_to_rot(T[3:]), T[:3])

    def right(self, t):
        T = self.pose(t)
        R = synth_ba.quat_to_rot(T[3:])
        return self._render(R, T[:3] + R @ np.array([BASELINE, 0.0, 0.0]))

    def world_of_pixel(self, t, xy, right=False):
        """3-D world point seen at left (or right) pixel xy (n,2) of frame t"""
        T = self.pose(t)
        R = synth_ba.quat_to_rot(T[3:])
        c = T[:3] + (R @ np.array([BASELINE, 0.0, 0.0]) if right else 0.0)
        xy = np.asarray(xy, np.float64)
        rays = np.stack([(xy[:, 0] - K4[2]) / K4[0], (xy[:, 1] - K4[3]) / K4[1], np.ones(len(xy))], -1) @ R.T
        lam = (self.d - self.nrm @ c) / (rays @ self.nrm)
        return c[None, :] + lam[:, None] * rays

    def project(self, t, X, right=False):
        T = self.pose(t)
        R = synth_ba.quat_to_rot(T[3:])
        c = T[:3] + (R @ np.array([BASELINE, 0.0, 0.0]) if right else 0.0)
        pc = (np.asarray(X, np.float64) - c[None, :]) @ R
        return np.stack([K4[0] * pc[:, 0] / pc[:, 2] + K4[2], K4[1] * pc[:, 1] / pc[:, 2] + K4[3]], -1)
