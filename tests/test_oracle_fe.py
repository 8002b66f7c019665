"""CPU tests of the front-end oracle itself (no GPU): self-consistency on the synthetic stream and a
brute-force numpy re-derivation of the integer stages.  OpenCV is not in the container and the reference
holds no fixture for this path, so the KLT oracle is 'parity unpinned' (see oracle/ov2_oracle.h); these
tests pin it against independent numpy formulas of the published algorithm instead."""
import numpy as np
import pytest

from ov2slam_amd import synth


def _reflect(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def test_pyrdown_and_scharr_against_numpy(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    P = oracle.Pyramid(img, 9, 1)
    i0, g0, w, h, p = P.level(0)
    assert (w, h, p) == (53, 37, 9)
    # level 0 interior and REFLECT_101 border
    assert np.array_equal(i0[p:p + h, p:p + w], img)
    ys = _reflect(np.arange(-p, h + p), h)
    xs = _reflect(np.arange(-p, w + p), w)
    assert np.array_equal(i0, img[np.ix_(ys, xs)])
    # Scharr with numpy (int32)
    a = img.astype(np.int32)
    pad = a[np.ix_(_reflect(np.arange(-1, h + 1), h), _reflect(np.arange(-1, w + 1), w))]
    t0 = 3 * (pad[:-2] + pad[2:]) + 10 * pad[1:-1]
    t1 = pad[2:] - pad[:-2]
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, :-2] + t1[:, 2:]) + 10 * t1[:, 1:-1]
    assert np.array_equal(g0[p:p + h, p:p + w, 0], ix) and np.array_equal(g0[p:p + h, p:p + w, 1], iy)
    assert not g0[:p].any() and not g0[:, :p].any() and not g0[p + h:].any() and not g0[:, p + w:].any()
    # pyrDown with numpy
    i1, _, w1, h1, _ = P.level(1)
    assert (w1, h1) == (27, 19)
    k = np.array([1, 4, 6, 4, 1])
    exp = np.zeros((h1, w1), np.int64)
    for j in range(5):
        for i in range(5):
            yy = _reflect(2 * np.arange(h1) - 2 + j, h)
            xx = _reflect(2 * np.arange(w1) - 2 + i, w)
            exp += k[j] * k[i] * a[np.ix_(yy, xx)]
    assert np.array_equal(i1[p:p + h1, p:p + w1], ((exp + 128) >> 8).astype(np.uint8))


def test_pyramid_early_stop(oracle):
    img = np.zeros((40, 30), np.uint8)
    assert oracle.Pyramid(img, 9, 3).nlevels == 2   # 15x20 ok, next 8x10 <= 9 stops (buildOpticalFlowPyramid)


def test_clahe_properties(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(60, 120, size=(480, 752), dtype=np.uint8)
    out = oracle.clahe(img, 3.0, 15, 9)
    assert out.shape == img.shape and out.std() > img.std()        # contrast is stretched
    flat = np.full((480, 752), 77, np.uint8)
    f = oracle.clahe(flat, 3.0, 15, 9)
    assert len(np.unique(f)) == 1                                   # constant in, constant out
    # clip=0 disables clipping -> plain tile histogram equalisation; LUT must be monotone: order preserved per tile centre
    o2 = oracle.clahe(img, 0.0, 1, 1)
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(o2.ravel()[order].astype(int)) >= 0).all()


def _clahe_numpy(img, clip, tiles_x, tiles_y):
    """independent numpy restatement of cv::CLAHE::apply for 8-bit images (OpenCV modules/imgproc/src/clahe.cpp as
    described in SURVEY Appendix A): REFLECT_101 extension to a multiple of the tile grid, per-tile histogram, clip +
    uniform redistribution + residual stride, cumulative LUT scaled by 255 / tile area (round half to even, saturate),
    bilinear blend of the four surrounding tile LUTs in float32."""
    h, w = img.shape
    ew, eh = w, h
    if w % tiles_x or h % tiles_y:
        ew, eh = w + (tiles_x - w % tiles_x), h + (tiles_y - h % tiles_y)
    ext = np.pad(img, ((0, eh - h), (0, ew - w)), mode="reflect")
    tw, th = ew // tiles_x, eh // tiles_y
    area = tw * th
    limit = max(int(float(clip) * area / 256), 1) if clip > 0 else 0
    luts = np.zeros((tiles_y, tiles_x, 256), np.uint8)
    scale = np.float32(255.0) / np.float32(area)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if limit > 0:
                excess = int(np.maximum(hist - limit, 0).sum())
                hist = np.minimum(hist, limit)
                hist += excess // 256
                res = excess % 256
                if res:
                    step = max(256 // res, 1)
                    idx = np.arange(0, 256, step)[:res]
                    hist[idx] += 1
            cdf = np.cumsum(hist).astype(np.float32) * scale
            luts[ty, tx] = np.clip(np.rint(cdf), 0, 255).astype(np.uint8)
    inv_tw, inv_th = np.float32(1.0) / np.float32(tw), np.float32(1.0) / np.float32(th)
    xs, ys = np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32)
    txf, tyf = xs * inv_tw - np.float32(0.5), ys * inv_th - np.float32(0.5)
    tx1, ty1 = np.floor(txf).astype(np.int32), np.floor(tyf).astype(np.int32)
    xa, ya = (txf - tx1.astype(np.float32)), (tyf - ty1.astype(np.float32))
    xa1, ya1 = np.float32(1.0) - xa, np.float32(1.0) - ya
    tx2, ty2 = np.minimum(tx1 + 1, tiles_x - 1), np.minimum(ty1 + 1, tiles_y - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = img.astype(np.int64)
    Y1, Y2 = ty1[:, None], ty2[:, None]
    X1, X2 = tx1[None, :], tx2[None, :]
    f = lambda Y, X: luts[Y, X, v].astype(np.float32)
    top = f(Y1, X1) * xa1[None, :] + f(Y1, X2) * xa[None, :]
    bot = f(Y2, X1) * xa1[None, :] + f(Y2, X2) * xa[None, :]
    res = top * ya1[:, None] + bot * ya[:, None]
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,tiles,clip", [(752, 480, (15, 9), 3.0), (320, 240, (8, 8), 2.0), (97, 61, (3, 2), 40.0),
                                            (128, 96, (4, 4), 0.5), (200, 120, (4, 3), 0.0)])
def test_clahe_against_independent_numpy(oracle, w, h, tiles, clip):
    rng = np.random.default_rng(w + h)
    img = (np.linspace(0, 255, w)[None, :] * 0.6 + rng.integers(0, 100, size=(h, w))).clip(0, 255).astype(np.uint8)
    assert np.array_equal(oracle.clahe(img, clip, tiles[0], tiles[1]), _clahe_numpy(img, clip, tiles[0], tiles[1]))


def test_lk_recovers_known_subpixel_shift(oracle, stream):
    I0, I1 = stream.left(0), stream.left(10)
    P0, P1 = oracle.Pyramid(oracle.clahe(I0)), oracle.Pyramid(oracle.clahe(I1))
    kps = synth.grid_keypoints(600)
    gt = stream.flow(0, 10, kps)
    out, st, iters = oracle.fb_klt_tracking(P0, P1, kps, kps)
    e = np.linalg.norm(out[st > 0] - gt[st > 0], axis=1)
    assert st.mean() > 0.97 and np.median(e) < 0.15 and e.max() < 1.5   # 2 % zoom over 10 frames: translational LK model error
    assert iters > 0
    # identity pair: zero motion, every point survives the forward-backward gate
    out2, st2, _ = oracle.fb_klt_tracking(P0, P0, kps, kps)
    assert st2.all() and np.abs(out2 - kps).max() < 1e-3


def _lk_numpy_level0(Ii, Ig, Ji, pad, w, h, pt, guess, win=9, max_iter=30, eps=0.01, min_eig_thr=1e-4):
    """independent restatement of one LKTrackerInvoker level in numpy: float32 scalars in the order
    the C expressions evaluate, exact integer sums converted to float32 once (the oracle's documented choice).
    Ii / Ji: padded u8 planes, Ig: padded (Ix, Iy) int16 plane.  returns (x, y) float32, status, err."""
    f = np.float32
    half = f((win - 1) * 0.5)
    scale = f(1.0) / f(1 << 20)

    def weights(fx, fy):
        w00 = int(np.rint((f(1) - fx) * (f(1) - fy) * f(1 << 14)))
        w01 = int(np.rint(fx * (f(1) - fy) * f(1 << 14)))
        w10 = int(np.rint((f(1) - fx) * fy * f(1 << 14)))
        return w00, w01, w10, (1 << 14) - w00 - w01 - w10

    px, py = f(pt[0]) - half, f(pt[1]) - half
    ipx, ipy = int(np.floor(px)), int(np.floor(py))
    if ipx < -win or ipx >= w or ipy < -win or ipy >= h:
        return (f(guess[0]), f(guess[1])), 0, f(0)
    w00, w01, w10, w11 = weights(px - f(ipx), py - f(ipy))
    ys, xs = np.mgrid[0:win, 0:win]
    r, c = ipy + pad + ys, ipx + pad + xs
    tap = lambda P: P[r, c].astype(np.int64) * w00 + P[r, c + 1].astype(np.int64) * w01 + P[r + 1, c].astype(np.int64) * w10 + \
        P[r + 1, c + 1].astype(np.int64) * w11
    I = (tap(Ii) + (1 << 8)) >> 9
    Ix = (tap(Ig[..., 0]) + (1 << 13)) >> 14
    Iy = (tap(Ig[..., 1]) + (1 << 13)) >> 14
    tof = lambda v: f(float(int(v))) * scale           # exact integer -> double -> float32, then the 2^-20 scale
    A11, A12, A22 = tof((Ix * Ix).sum()), tof((Ix * Iy).sum()), tof((Iy * Iy).sum())
    D = A11 * A22 - A12 * A12
    min_eig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f(4) * A12 * A12)) / f(2 * win * win)
    if min_eig < f(min_eig_thr) or D < f(1.1920929e-07):
        return (f(guess[0]), f(guess[1])), 0, min_eig
    D = f(1) / D
    nx, ny = f(guess[0]) - half, f(guess[1]) - half
    ox, oy = f(guess[0]), f(guess[1])
    e = float(np.float32(eps))
    pdx = pdy = f(0)
    for j in range(max_iter):
        inx, iny = int(np.floor(nx)), int(np.floor(ny))
        if inx < -win or inx >= w or iny < -win or iny >= h:
            return (ox, oy), 0, min_eig
        w00, w01, w10, w11 = weights(nx - f(inx), ny - f(iny))
        r, c = iny + pad + ys, inx + pad + xs
        Jv = (tap(Ji) + (1 << 8)) >> 9
        diff = Jv - I
        b1, b2 = tof((diff * Ix).sum()), tof((diff * Iy).sum())
        dx, dy = (A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D
        nx, ny = nx + dx, ny + dy
        ox, oy = nx + half, ny + half
        if float(dx) * float(dx) + float(dy) * float(dy) <= e * e:
            break
        if j > 0 and abs(dx + pdx) <= f(0.01) and abs(dy + pdy) <= f(0.01):
            ox, oy = ox - dx * f(0.5), oy - dy * f(0.5)
            break
        pdx, pdy = dx, dy
    return (ox, oy), 1, min_eig


@pytest.mark.parametrize("win", [5, 9, 11])
def test_lk_level_against_independent_numpy(oracle, stream, win):
    """one pyramid level of calcOpticalFlowPyrLK (maxLevel = 0): positions bit-equal, status and min-eigenvalue equal"""
    rng = np.random.default_rng(win)
    I0, I1 = stream.left(0), stream.left(4)
    I0 = I0.copy(); I0[200:260, 300:420] = 90            # a flat block: min-eigenvalue rejection
    P0, P1 = oracle.Pyramid(I0, 11, 0), oracle.Pyramid(I1, 11, 0)
    i0, g0, w, h, pad = P0.level(0)
    i1, _, _, _, _ = P1.level(0)
    n = 90
    pts = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)
    pts[:6] = [[330, 230], [0, 0], [w - 1, h - 1], [0.4, h - 0.6], [w + 20, 50], [-14, 100]]
    guess = (pts + rng.uniform(-1.5, 1.5, pts.shape)).astype(np.float32)
    guess[7] = [w + 40, 10]                                # the search window leaves the image: status 0
    out, st, err, _ = oracle.calc_optical_flow_pyr_lk(P0, P1, pts, guess, win=win, max_level=0)
    for k in range(n):
        (x, y), s, e = _lk_numpy_level0(i0, g0, i1, pad, w, h, pts[k], guess[k], win=win)
        assert s == st[k], k
        if s:
            assert np.float32(e) == err[k], k
        assert (np.float32(x), np.float32(y)) == (out[k, 0], out[k, 1]), (k, x, y, out[k])
    assert st.sum() > 40 and st[0] == 0


def test_lk_pyramid_chain_against_independent_numpy(oracle, stream):
    """the coarse-to-fine chaining of calcOpticalFlowPyrLK (prev = kp / 2^l, the guess enters at maxLevel as
    prior / 2^l and is doubled between levels, status can only be cleared at level 0, err = min-eigenvalue of the last
    level that ran) on top of the per-level numpy restatement above"""
    rng = np.random.default_rng(3)
    P0, P1 = oracle.Pyramid(stream.left(0), 9, 3), oracle.Pyramid(stream.left(9), 9, 3)
    lv0 = [P0.level(l) for l in range(4)]
    lv1 = [P1.level(l) for l in range(4)]
    n = 60
    pts = np.stack([rng.uniform(2, 750, n), rng.uniform(2, 478, n)], 1).astype(np.float32)
    pts[:4] = [[0, 0], [751, 479], [1.5, 470.2], [749.0, 3.0]]
    guess = (pts + rng.uniform(-4, 4, pts.shape)).astype(np.float32)
    out, st, err, _ = oracle.calc_optical_flow_pyr_lk(P0, P1, pts, guess, win=9, max_level=3)
    f = np.float32
    for k in range(n):
        status, e, nxt = 1, f(0), None
        for l in (3, 2, 1, 0):
            sc = f(1.0) / f(1 << l)
            i0, g0, w, h, pad = lv0[l]
            i1 = lv1[l][0]
            p = (pts[k, 0] * sc, pts[k, 1] * sc)
            g = (guess[k, 0] * sc, guess[k, 1] * sc) if l == 3 else (nxt[0] * f(2), nxt[1] * f(2))
            nxt, s, el = _lk_numpy_level0(i0, g0, i1, pad, w, h, p, g)
            if s or el != 0:
                e = el
            if l == 0 and not s:
                status = 0
        assert status == st[k], k
        assert (f(nxt[0]), f(nxt[1])) == (out[k, 0], out[k, 1]), (k, nxt, out[k])
        if status:
            assert f(e) == err[k], k


def test_fb_wrapper_gates(oracle, stream):
    I0 = stream.left(0).copy()
    I0[200:300, 200:400] = 90
    P0 = oracle.Pyramid(I0)
    kps = np.array([[300.0, 250.0], [0.2, 0.3], [100.0, 100.0]], np.float32)
    out, st, _ = oracle.fb_klt_tracking(P0, P0, kps, kps)
    assert list(st) == [0, 0, 1]       # flat -> minEig reject ; (0.2,0.3) fails inBorder ; textured ok
    o, s, _ = oracle.fb_klt_tracking(P0, P0, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert o.shape == (0, 2) and s.shape == (0,)


def test_klt_tracking_frame_batches(oracle, stream):
    P0, P1 = oracle.Pyramid(stream.left(0)), oracle.Pyramid(stream.left(10))
    kps = synth.grid_keypoints(500)
    gt = stream.flow(0, 10, kps)
    pri, has = synth.make_priors(kps, gt, sigma=1.0)
    out, st, p3p = oracle.klt_tracking_frame(P0, P1, kps, pri, has)
    assert not p3p and st.mean() > 0.95
    pri, has = synth.make_priors(kps, gt, sigma=25.0)
    out, st, p3p = oracle.klt_tracking_frame(P0, P1, kps, pri, has)
    assert p3p and st.mean() > 0.9      # priors dropped, full pyramid recovers
