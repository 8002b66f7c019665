"""CPU tests of the detector oracle (no GPU): pieces against independent numpy/scipy formulas, end-to-end sanity on
synthetic corners.  OpenCV is not available => parity unpinned for these (see oracle/ov2_oracle_det.c header)."""
import numpy as np

from ov2slam_amd import synth


def test_filled_disc_is_the_midpoint_circle(oracle):
    m = np.ones((41, 41), np.uint8)
    oracle.draw_disc(m, 20, 20, 8, 0)
    ys, xs = np.nonzero(m == 0)
    d2 = (ys - 20) ** 2 + (xs - 20) ** 2
    assert d2.max() <= 8 * 8 + 8 and (m[20, 12:29] == 0).all() and m[20, 11] == 1 and m[20, 29] == 1
    assert (m == m[::-1]).all() and (m == m[:, ::-1]).all() and (m == m.T).all()      # 8-fold symmetry
    inner = np.add.outer((np.arange(41) - 20) ** 2, (np.arange(41) - 20) ** 2) <= 7 * 7
    assert (m[inner] == 0).all()
    c = np.ones((10, 10), np.uint8)                       # clipping at the image border
    oracle.draw_disc(c, 0, 9, 3, 0)
    assert c[9, 0] == 0 and c[9, 3] == 0 and c[6, 0] == 0 and c[9, 4] == 1


def test_min_eig_cell_against_float64_formula(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(120, 160), dtype=np.uint8)
    x0, y0, n = 35, 70, 35
    got = oracle.min_eig_cell(img, x0, y0, n)
    # independent float64 restatement: blur (parent pixels), Sobel/box with reflect101 at the cell border
    pad = np.pad(img.astype(np.int64), 1, mode="reflect")
    k = np.array([1, 2, 1])
    bl = sum(k[j] * k[i] * pad[y0 + j:y0 + j + n, x0 + i:x0 + i + n] for j in range(3) for i in range(3))
    bl = ((bl + 8) >> 4).astype(np.float64)
    p = np.pad(bl, 1, mode="reflect")
    s = 1.0 / (4 * 3 * 255)
    dx = s * ((p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2]))
    dy = s * ((p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:]))

    def box(a):
        q = np.pad(a, 1, mode="reflect")
        return sum(q[j:j + n, i:i + n] for j in range(3) for i in range(3))
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    assert np.allclose(got, ref, rtol=2e-4, atol=1e-7)
    assert got.min() > -1e-6


def test_fast_score_definition(oracle):
    """cornerScore == the largest threshold at which the pixel is still a 9/16 corner"""
    rng = np.random.default_rng(1)
    img = (rng.integers(0, 256, size=(40, 40)) // 32 * 32).astype(np.uint8)
    img[10:, 10:] = np.minimum(img[10:, 10:].astype(int) + 120, 255).astype(np.uint8)
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]

    def is_corner(x, y, t):
        v = int(img[y, x])
        ring = [int(img[y + dy[k], x + dx[k]]) for k in range(16)]
        for s in range(16):
            seg = [ring[(s + k) % 16] for k in range(9)]
            if all(p > v + t for p in seg) or all(p < v - t for p in seg):
                return True
        return False
    n = 0
    for y in range(3, 37):
        for x in range(3, 37):
            s = oracle.fast_score(img, x, y, 10)
            if not is_corner(x, y, 10):
                assert s == 0
                continue
            n += 1
            assert s >= 10 and is_corner(x, y, s) and not is_corner(x, y, s + 1)
    assert n > 20


def _corner_image(w=752, h=480, seed=3):
    """checkerboard-like synthetic image with known corner positions + mild noise"""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    img = 60 + 120 * (((xs // 47) + (ys // 41)) % 2) + rng.normal(0, 2.0, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_detect_single_scale_end_to_end(oracle):
    img = _corner_image()
    pts, q = oracle.detect_single_scale(img, 35, np.zeros((0, 2), np.float32), 0.001)
    ncells = (480 // 35) * (752 // 35)
    assert 0.3 * ncells < len(pts) <= ncells
    # detections sit on checkerboard corners (multiples of 47 / 41) after cornerSubPix
    ex = np.minimum(pts[:, 0] % 47, 47 - pts[:, 0] % 47)
    ey = np.minimum(pts[:, 1] % 41, 41 - pts[:, 1] % 41)
    assert np.median(np.hypot(ex, ey)) < 1.0
    # occupied cells are skipped and existing keypoints mask a disc of radius cell/4
    pts2, _ = oracle.detect_single_scale(img, 35, pts, 0.001)
    if len(pts2):
        d = np.linalg.norm(pts2[:, None, :] - np.rint(pts)[None, :, :], axis=2).min(1)
        assert d.min() > 35 // 4 - 1.5
    cells1 = set(map(tuple, (pts // 35).astype(int)))
    cells2 = set(map(tuple, (np.rint(pts2) // 35).astype(int))) if len(pts2) else set()
    assert len(pts2) < len(pts)
    # threshold adaptation (:418-423): > 90 % of the free cells filled -> x1.5 ; < 33 % -> /2
    assert q in (0.001, 0.0015, 0.0005)
    _, qflat = oracle.detect_single_scale(np.full((480, 752), 90, np.uint8), 35, np.zeros((0, 2), np.float32), 0.001)
    assert qflat == 0.0005
    # roi: detections outside are dropped
    roi = [100, 50, 400, 300]
    pr, _ = oracle.detect_single_scale(img, 35, np.zeros((0, 2), np.float32), 0.001, roi=roi, subpix=False)
    assert len(pr) and (pr[:, 0] >= 100).all() and (pr[:, 0] < 500).all() and (pr[:, 1] >= 50).all() and (pr[:, 1] < 350).all()


def test_detect_grid_fast_end_to_end(oracle, stream):
    img = stream.left(0)
    pts, th = oracle.detect_grid_fast(img, 50, np.zeros((0, 2), np.float32), 10, subpix=False)
    ncells = (480 // 50) * (752 // 50)
    assert 0 < len(pts) <= ncells and th in (6, 10, 15)
    assert np.array_equal(pts, np.rint(pts))
    for x, y in pts.astype(int):
        assert oracle.fast_score(img, x, y, 10) >= 20
        assert (x % 50) % 4 >= 2            # the reference's float-mask-read-as-bytes quirk
    flat, th2 = oracle.detect_grid_fast(np.full((480, 752), 100, np.uint8), 50, np.zeros((0, 2), np.float32), 10)
    assert len(flat) == 0 and th2 == 6      # < 50 % of the empty cells -> threshold * 0.66 (int)


def test_corner_subpix_converges_on_a_synthetic_corner(oracle):
    h, w = 80, 80
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = 40.3, 37.6
    # smooth-edged quadrant corner at (cx, cy)
    img = 50 + 150 / (1 + np.exp(-(xs - cx) * 2)) * (1 / (1 + np.exp(-(ys - cy) * 2)))
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    out = oracle.corner_subpix(img, np.array([[41.0, 38.0], [39.0, 37.0]], np.float32))
    assert np.abs(out - [cx, cy]).max() < 0.6
    far = oracle.corner_subpix(img, np.array([[10.0, 10.0]], np.float32))     # flat: singular system -> unchanged
    assert np.allclose(far, [[10.0, 10.0]])


def _corner_subpix_numpy(img, pt, hw=3, max_iter=30, eps=0.01):
    """independent numpy restatement of cv::cornerSubPix (win = (hw, hw), no zero zone) from its published algorithm:
    getRectSubPix of a (2hw+3)^2 float patch around the current estimate (bilinear, BORDER_REPLICATE), central
    differences, weights exp(-x^2/hw^2) exp(-y^2/hw^2), gradient-orthogonality normal equations in double; float64
    sums in natural order (the oracle's strided 16-leaf order differs in the last bits only)."""
    h, w = img.shape
    win = 2 * hw + 1
    bw = win + 2
    m1 = np.exp(-((np.arange(win) - hw) ** 2).astype(np.float32) * np.float32(1.0 / (hw * hw))).astype(np.float32)
    mask = np.outer(m1, m1).astype(np.float32)
    cT = np.array(pt, np.float32)
    cI = cT.copy()
    it = 0
    while True:
        c = cI - np.float32((bw - 1) * 0.5)
        ip = np.floor(c).astype(np.int64)
        a, b = np.float32(c[0] - ip[0]), np.float32(c[1] - ip[1])
        ys = np.clip(ip[1] + np.arange(bw + 1), 0, h - 1)
        xs = np.clip(ip[0] + np.arange(bw + 1), 0, w - 1)
        src = img[np.ix_(ys, xs)].astype(np.float32)
        one = np.float32(1)
        a11, a12, a21, a22 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
        patch = src[:-1, :-1] * a11 + src[:-1, 1:] * a12 + src[1:, :-1] * a21 + src[1:, 1:] * a22
        gx = (patch[1:-1, 2:] - patch[1:-1, :-2]).astype(np.float64)
        gy = (patch[2:, 1:-1] - patch[:-2, 1:-1]).astype(np.float64)
        m = mask.astype(np.float64)
        gxx, gxy, gyy = gx * gx * m, gx * gy * m, gy * gy * m
        py, px = np.mgrid[-hw:hw + 1, -hw:hw + 1].astype(np.float64)
        A, B, Cc = gxx.sum(), gxy.sum(), gyy.sum()
        bb1, bb2 = (gxx * px + gxy * py).sum(), (gxy * px + gyy * py).sum()
        det = A * Cc - B * B
        if abs(det) <= np.finfo(np.float64).eps ** 2:
            break
        sc = 1.0 / det
        n = np.array([cI[0] + Cc * sc * bb1 - B * sc * bb2, cI[1] - B * sc * bb1 + A * sc * bb2]).astype(np.float32)
        d = n - cI
        err = float(d[0] * d[0] + d[1] * d[1])
        cI = n
        if cI[0] < 0 or cI[0] >= w or cI[1] < 0 or cI[1] >= h:
            break
        it += 1
        if not (it < max_iter and err > eps * eps):
            break
    if abs(cI[0] - cT[0]) > hw or abs(cI[1] - cT[1]) > hw:
        cI = cT
    return cI


def test_corner_subpix_against_independent_numpy(oracle, stream):
    img = stream.left(2)
    det, _ = oracle.detect_single_scale(img, 35, np.zeros((0, 2), np.float32), 0.001, subpix=False)
    assert len(det) > 100
    pts = det[:120].astype(np.float32)
    pts[:4] = [[1.0, 1.0], [750.0, 478.0], [0.0, 240.0], [375.5, 0.2]]          # patches reaching over the border: REPLICATE
    got = oracle.corner_subpix(img, pts)
    want = np.stack([_corner_subpix_numpy(img, p) for p in pts])
    d = np.abs(got - want).max(1)
    assert np.median(d) < 1e-5 and d.max() < 2e-3, (np.median(d), d.max())
