"""GPU parity tests (through the C ABI) of the keyframe-rate detectors against the CPU oracle: bit-exact point lists
(integer arg-max positions, fp32 cornerSubPix positions) and identical threshold adaptation."""
import numpy as np
import pytest

from ov2slam_amd import frontend as fe

pytestmark = pytest.mark.gpu


def _corner_image(w=752, h=480, seed=3):
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    img = 60 + 120 * (((xs // 47) + (ys // 41)) % 2) + rng.normal(0, 2.0, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subpix", [False, True])
@pytest.mark.parametrize("cell", [35, 50])
def test_detect_single_scale_bit_exact(ctx, oracle, stream, cell, subpix):
    for img in (_corner_image(), stream.left(4)):
        pyr = fe.preprocess_image(ctx, img, use_clahe=False, nklt_pyr_lvl=0)
        cur = np.zeros((0, 2), np.float32)
        ex = fe.FeatureExtractor(ctx, nmaxdist=cell, dmaxquality=0.001)
        q = 0.001
        for _ in range(3):      # three keyframes in a row: existing keypoints mask cells, thresholds adapt
            got = ex.detectSingleScale(pyr, cur, subpix=subpix)
            want, q = oracle.detect_single_scale(img, cell, cur, q, subpix=subpix)
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            assert ex.dmaxquality_ == q
            cur = np.concatenate([cur, got[: len(got) // 2]])


def test_detect_single_scale_roi_and_clahe_frame(ctx, oracle, stream):
    img = stream.left(2)
    pyr = fe.preprocess_image(ctx, img, use_clahe=True)      # the reference detects on the CLAHE'd frame
    cl = oracle.clahe(img)
    roi = [40, 30, 600, 400]
    ex = fe.FeatureExtractor(ctx, nmaxdist=35, dmaxquality=0.001)
    got = ex.detectSingleScale(pyr, np.array([[100.2, 100.7], [400.0, 300.0]], np.float32), roi=roi)
    want, q = oracle.detect_single_scale(cl, 35, np.array([[100.2, 100.7], [400.0, 300.0]], np.float32), 0.001, roi=roi)
    assert len(want) > 50 and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and ex.dmaxquality_ == q


@pytest.mark.parametrize("th", [10, 25])
def test_detect_grid_fast_bit_exact(ctx, oracle, stream, th):
    for img in (stream.left(1), _corner_image(seed=9)):
        pyr = fe.preprocess_image(ctx, img, use_clahe=False, nklt_pyr_lvl=0)
        ex = fe.FeatureExtractor(ctx, nmaxdist=50, nfast_th=th)
        cur = np.zeros((0, 2), np.float32)
        t = th
        for _ in range(3):
            got = ex.detectGridFAST(pyr, cur)
            want, t = oracle.detect_grid_fast(img, 50, cur, t)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            assert ex.nfast_th_ == t
            cur = np.concatenate([cur, got])


def test_detect_edge_cases(ctx, oracle):
    flat = np.full((480, 752), 90, np.uint8)
    pyr = fe.preprocess_image(ctx, flat, use_clahe=False, nklt_pyr_lvl=0)
    ex = fe.FeatureExtractor(ctx, nmaxdist=35, dmaxquality=0.001, nfast_th=10)
    assert len(ex.detectSingleScale(pyr, np.zeros((0, 2), np.float32))) == 0 and ex.dmaxquality_ == 0.0005
    ex.nmaxdist_ = 50
    assert len(ex.detectGridFAST(pyr, np.zeros((0, 2), np.float32))) == 0 and ex.nfast_th_ == 6
    # image smaller than one cell: no cells, no output
    small = fe.preprocess_image(ctx, np.zeros((30, 30), np.uint8), use_clahe=False, nklt_pyr_lvl=0)
    ex.nmaxdist_ = 35
    assert len(ex.detectSingleScale(small, np.zeros((0, 2), np.float32))) == 0
    with pytest.raises(Exception):
        fe.FeatureExtractor(ctx, nmaxdist=4).detectSingleScale(pyr, np.zeros((0, 2), np.float32))


def test_detect_batched_equals_per_image(ctx, oracle, stream):
    B = 3
    raw = [stream.left(t) for t in (0, 3, 7)]
    imgs = fe.Images(ctx, B, 752, 480)
    for b in range(B):
        imgs.upload(b, raw[b])
    pyr = fe.preprocess_images(ctx, imgs, use_clahe=True)
    ctx.synchronize()
    cur = [np.zeros((0, 2), np.float32), np.array([[200.5, 100.25], [50.0, 400.0]], np.float32), np.zeros((0, 2), np.float32)]
    for mode, cell, th0 in ((1, 35, 0.001), (0, 50, 10.0)):
        th = np.full(B, th0, np.float64)
        got = fe.detect_grid_batch(ctx, pyr, cell, mode, th, cur)
        for b in range(B):
            cl = oracle.clahe(raw[b])
            if mode == 1:
                want, t = oracle.detect_single_scale(cl, cell, cur[b], th0)
            else:
                want, t = oracle.detect_grid_fast(cl, cell, cur[b], int(th0))
            assert np.array_equal(got[b].view(np.uint32), want.view(np.uint32)), (mode, b)
            assert th[b] == t
        # single-image call on image 1 of the batch gives the same points
        ex = fe.FeatureExtractor(ctx, nmaxdist=cell, dmaxquality=th0, nfast_th=int(th0))
        one = ex.detectSingleScale(pyr, cur[1], b=1) if mode == 1 else ex.detectGridFAST(pyr, cur[1], b=1)
        assert np.array_equal(one.view(np.uint32), got[1].view(np.uint32))


def test_detect_dev_variant_with_validity_mask(ctx, oracle, stream):
    """ov2_detect_grid_batch_dev: keypoints, thresholds, counts and corners device-resident; d_cur_valid selects which
    keypoints count (the tracking status in a real front-end).  Same result as the oracle fed the valid subset."""
    from ov2slam_amd import frontend as fe, synth
    B = 3
    ims = fe.Images(ctx, B, 752, 480)
    raws = [stream.left(4 * b) for b in range(B)]
    for b in range(B):
        ims.upload(b, raws[b])
    pyr = fe.preprocess_images(ctx, ims)
    rng = np.random.default_rng(3)
    kps = [synth.grid_keypoints(300, seed=50 + b) for b in range(B)]
    valid = [rng.uniform(size=len(k)) < 0.7 for k in kps]
    cell, cap = 35, 2 * (752 // 35) * (480 // 35)
    d_xy = ctx.to_device(np.concatenate(kps).astype(np.float32))
    d_img = ctx.to_device(np.concatenate([np.full(len(k), b, np.int32) for b, k in enumerate(kps)]))
    d_val = ctx.to_device(np.concatenate(valid).astype(np.uint8))
    d_th = ctx.to_device(np.full(B, 0.001))
    d_n, d_out = ctx.empty((B,), np.int32), ctx.empty((B, cap, 2), np.float32)
    fe.detect_grid_batch_dev(ctx, pyr, cell, 1, d_th, sum(len(k) for k in kps), d_xy, d_img, d_val, d_n, d_out, cap)
    ctx.synchronize()
    n, out, th = d_n.get(), d_out.get(), d_th.get()
    for b in range(B):
        e, eth = oracle.detect_single_scale(oracle.clahe(raws[b]), cell, kps[b][valid[b]], 0.001)
        assert n[b] == len(e)
        assert np.array_equal(out[b, :n[b]].view(np.uint32), e.view(np.uint32))
        assert th[b] == eth
