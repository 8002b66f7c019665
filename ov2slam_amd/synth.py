"""Seeded synthetic inputs for the hot path: a 752x480 stereo texture stream with
analytically known optical flow, jittered-grid keypoints, and priors split like
VisualFrontEnd::kltTracking (reference src/visual_front_end.cpp:142-184).  numpy only, no GPU.
There is no EuRoC data in the build or on the GPU box; this is the stand-in of the same shape.
"""
import numpy as np

IMG_W, IMG_H = 752, 480
SEED_IMG = 20210


def _box5(a):
    k = 5
    c = np.cumsum(np.pad(a, ((k // 2 + 1, k // 2), (0, 0)), mode="reflect"), axis=0)
    a = (c[k:] - c[:-k]) / k
    c = np.cumsum(np.pad(a, ((0, 0), (k // 2 + 1, k // 2)), mode="reflect"), axis=1)
    return (c[:, k:] - c[:, :-k]) / k


def base_texture(h=IMG_H + 160, w=IMG_W + 160, seed=SEED_IMG):
    """band-limited noise: uniform u8 noise box-blurred 3x with 5x5, stretched to [16,240] (float64)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(h, w)).astype(np.float64)
    for _ in range(3):
        a = _box5(a)
    a = (a - a.min()) / (a.max() - a.min())
    return 16.0 + a * (240.0 - 16.0)


def _bilinear(tex, u, v):
    h, w = tex.shape
    u = np.clip(u, 0.0, w - 1.001)
    v = np.clip(v, 0.0, h - 1.001)
    x0 = np.floor(u).astype(np.int64)
    y0 = np.floor(v).astype(np.int64)
    a = u - x0
    b = v - y0
    return ((1 - a) * (1 - b) * tex[y0, x0] + a * (1 - b) * tex[y0, x0 + 1] +
            (1 - a) * b * tex[y0 + 1, x0] + a * b * tex[y0 + 1, x0 + 1])


class StereoStream:
    """frame t = base texture under a smooth sub-pixel translation + 0.2 %/frame zoom; right image =
    left shifted by a (row-dependent) disparity of 8..40 px, rectified-like."""

    def __init__(self, w=IMG_W, h=IMG_H, seed=SEED_IMG):
        self.w, self.h = w, h
        self.tex = base_texture(h + 160, w + 160, seed)
        self.c_tex = np.array([(w + 160) / 2.0, (h + 160) / 2.0])
        self.c_img = np.array([w / 2.0, h / 2.0])

    def _pose(self, t):
        s = 1.0 / (1.002 ** t)
        d = np.array([2.3 * np.sin(0.07 * t), 1.7 * np.cos(0.05 * t)])
        return s, d

    def to_tex(self, t, xy):
        s, d = self._pose(t)
        return self.c_tex + s * (np.asarray(xy, np.float64) - self.c_img) + d

    def from_tex(self, t, uv):
        s, d = self._pose(t)
        return self.c_img + (np.asarray(uv, np.float64) - self.c_tex - d) / s

    def flow(self, t0, t1, xy):
        """ground-truth position in frame t1 of points xy (n,2) given in frame t0."""
        return self.from_tex(t1, self.to_tex(t0, xy))

    def disparity(self, y):
        return 8.0 + 32.0 * (np.asarray(y, np.float64) / (self.h - 1))

    def left(self, t):
        ys, xs = np.mgrid[0:self.h, 0:self.w]
        uv = self.to_tex(t, np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64))
        img = _bilinear(self.tex, uv[:, 0], uv[:, 1]).reshape(self.h, self.w)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def right(self, t):
        ys, xs = np.mgrid[0:self.h, 0:self.w]
        xs = xs.astype(np.float64) + self.disparity(ys)   # x_r = x_l - d  <=>  sample left at x_r + d
        uv = self.to_tex(t, np.stack([xs.ravel(), ys.ravel().astype(np.float64)], 1))
        img = _bilinear(self.tex, uv[:, 0], uv[:, 1]).reshape(self.h, self.w)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def stereo_gt(self, xy):
        xy = np.asarray(xy, np.float64).copy()
        xy[:, 0] -= self.disparity(xy[:, 1])
        return xy


def grid_keypoints(n, w=IMG_W, h=IMG_H, border=16, jitter=3.0, seed=SEED_IMG + 7):
    """~n points on a jittered grid inside a border (float32 (n,2))."""
    rng = np.random.default_rng(seed)
    aspect = (w - 2 * border) / (h - 2 * border)
    ny = max(1, int(round(np.sqrt(n / aspect))))
    nx = max(1, int(np.ceil(n / ny)))
    gx = border + (np.arange(nx) + 0.5) * (w - 2 * border) / nx
    gy = border + (np.arange(ny) + 0.5) * (h - 2 * border) / ny
    pts = np.stack(np.meshgrid(gx, gy), -1).reshape(-1, 2)[:n]
    pts = pts + rng.uniform(-jitter, jitter, size=pts.shape)
    pts[:, 0] = np.clip(pts[:, 0], border, w - 1 - border)
    pts[:, 1] = np.clip(pts[:, 1], border, h - 1 - border)
    return pts.astype(np.float32)


def make_priors(kps, gt_next, frac_prior=0.7, sigma=1.0, seed=SEED_IMG + 11):
    """70 % of the points carry a motion-model prior = GT + N(0, 1 px) (tracked on 2 levels by the
    reference), 30 % start from the previous position (full pyramid).  returns (prior_xy, has_prior)."""
    rng = np.random.default_rng(seed)
    n = kps.shape[0]
    has = rng.uniform(size=n) < frac_prior
    pri = np.where(has[:, None], gt_next + rng.normal(0.0, sigma, size=(n, 2)), kps).astype(np.float32)
    return pri, has.astype(np.uint8)
