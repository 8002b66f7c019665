"""GPU parity tests (through the C ABI) of CLAHE + pyramid + forward-backward KLT against the CPU oracle.
Bar: bit-exact -- pyramid bytes, KLT status flags, and tracked positions (float32 bit patterns, which is
stronger than the integer-pixel bar north_star asks for)."""
import numpy as np
import pytest

from ov2slam_amd import frontend as fe, synth

pytestmark = pytest.mark.gpu


def _assert_pyr_equal(gp, op, b=0):
    assert gp.nlevels == op.nlevels
    for l in range(op.nlevels):
        gi, gg, w, h, p = gp.level(l, b)
        oi, og, ow, oh, opad = op.level(l)
        assert (w, h, p) == (ow, oh, opad)
        assert np.array_equal(gi, oi), f"level {l} image differs at {np.argwhere(gi != oi)[:5]}"
        assert np.array_equal(gg, og), f"level {l} gradient differs at {np.argwhere(gg != og)[:5]}"


@pytest.mark.parametrize("use_clahe", [False, True])
def test_pyramid_752x480_bit_exact(ctx, oracle, stream, use_clahe):
    img = stream.left(3)
    gp = fe.preprocess_image(ctx, img, use_clahe=use_clahe)
    src = oracle.clahe(img, 3.0, 15, 9) if use_clahe else img
    _assert_pyr_equal(gp, oracle.Pyramid(src, 9, 3))


@pytest.mark.parametrize("w,h", [(64, 48), (101, 67), (333, 257), (1241, 376), (20, 20), (19, 23)])
def test_pyramid_odd_sizes(ctx, oracle, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    gp = fe.preprocess_image(ctx, img, use_clahe=False)
    _assert_pyr_equal(gp, oracle.Pyramid(img, 9, 3))


@pytest.mark.parametrize("w,h,tiles,clip", [(752, 480, (15, 9), 3.0), (640, 480, (12, 9), 3.0), (1241, 376, (24, 7), 3.0),
                                            (320, 240, (8, 8), 2.0), (97, 61, (3, 2), 40.0), (128, 128, (4, 4), 0.5)])
def test_clahe_bit_exact(ctx, oracle, w, h, tiles, clip):
    rng = np.random.default_rng(w + h)
    # mix of smooth ramp and noise so that clipping and the residual redistribution are both exercised
    img = (np.linspace(0, 255, w)[None, :] * 0.6 + rng.integers(0, 100, size=(h, w))).clip(0, 255).astype(np.uint8)
    gp = fe.preprocess_image(ctx, img, use_clahe=True, fclahe_val=clip, tiles=tiles, nklt_pyr_lvl=0)
    gi, _, gw, gh, p = gp.level(0)
    ref = oracle.clahe(img, clip, tiles[0], tiles[1])
    assert np.array_equal(gi[p:p + gh, p:p + gw], ref)


@pytest.mark.parametrize("w,h,tiles", [(97, 61, (3, 2)), (333, 257, (6, 5)), (1241, 376, (24, 7)), (70, 34, (2, 2)),
                                       (66, 20, (1, 1)), (131, 18, (2, 1))])
def test_clahe_pyramid_odd_sizes(ctx, oracle, w, h, tiles):
    """the fused level-0 kernel (CLAHE -> plane + Scharr + pyrDown) on ragged tiles, padding and gradients included"""
    rng = np.random.default_rng(7 * w + h)
    img = (np.linspace(0, 255, w)[None, :] * 0.5 + rng.integers(0, 128, size=(h, w))).clip(0, 255).astype(np.uint8)
    for nl in (0, 3):
        gp = fe.preprocess_image(ctx, img, use_clahe=True, fclahe_val=3.0, tiles=tiles, nklt_pyr_lvl=nl)
        _assert_pyr_equal(gp, oracle.Pyramid(oracle.clahe(img, 3.0, tiles[0], tiles[1]), 9, nl))


def test_pyramid_batched_matches_single(ctx, oracle, stream):
    B = 3
    imgs = fe.Images(ctx, B, 752, 480)
    raw = [stream.left(t) for t in (0, 5, 9)]
    for b in range(B):
        imgs.upload(b, raw[b])
    gp = fe.preprocess_images(ctx, imgs)
    ctx.synchronize()
    assert gp.batch == B
    for b in range(B):
        _assert_pyr_equal(gp, oracle.Pyramid(oracle.clahe(raw[b]), 9, 3), b)


def _pyrs(ctx, oracle, stream, t0, t1, clahe=True):
    I0, I1 = stream.left(t0), stream.left(t1)
    g0, g1 = fe.preprocess_image(ctx, I0, use_clahe=clahe), fe.preprocess_image(ctx, I1, use_clahe=clahe)
    if clahe:
        I0, I1 = oracle.clahe(I0), oracle.clahe(I1)
    return g0, g1, oracle.Pyramid(I0), oracle.Pyramid(I1)


@pytest.mark.parametrize("nlevels", [0, 1, 3])
@pytest.mark.parametrize("t1", [1, 12])
def test_fb_klt_bit_exact(ctx, oracle, stream, nlevels, t1):
    g0, g1, o0, o1 = _pyrs(ctx, oracle, stream, 0, t1)
    kps = synth.grid_keypoints(1000)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    out, st = trk.fbKltTracking(g0, g1, 9, nlevels, 30.0, 0.5, kps, kps)
    eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, kps, 9, nlevels, 30.0, 0.5, 30, 0.01)
    assert np.array_equal(st, est.astype(bool))
    assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))
    assert np.array_equal(np.rint(out).astype(np.int32), np.rint(eout).astype(np.int32))  # north_star's bar
    if nlevels == 3:
        gt = stream.flow(0, t1, kps)
        assert st.mean() > 0.95
        assert np.median(np.linalg.norm(out[st] - gt[st], axis=1)) < 0.2


@pytest.mark.parametrize("win", [5, 7, 9, 11])
def test_fb_klt_random_sweep(ctx, oracle, stream, win):
    """keypoints anywhere in (and a little outside) the image with priors up to 6 px off, every pyramid depth, several
    frame pairs: the staged-window loads (16-byte rows from the dword below the window, padding on all sides) against the
    scalar oracle, bit for bit"""
    rng = np.random.default_rng(100 + win)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for t0, t1 in ((0, 2), (3, 9), (14, 5)):
        I0, I1 = stream.left(t0), stream.left(t1)
        g0 = fe.preprocess_image(ctx, I0, use_clahe=True, klt_win_size=11)
        g1 = fe.preprocess_image(ctx, I1, use_clahe=True, klt_win_size=11)
        o0 = oracle.Pyramid(oracle.clahe(I0, 3.0, 15, 9), 11, 3)
        o1 = oracle.Pyramid(oracle.clahe(I1, 3.0, 15, 9), 11, 3)
        n = 1500
        kps = np.stack([rng.uniform(-2, 754, n), rng.uniform(-2, 482, n)], 1).astype(np.float32)
        kps[:200, 0] = rng.choice([0.0, 0.5, 3.0, 748.2, 751.0], 200)      # columns next to the left / right padding
        kps[200:400, 1] = rng.choice([0.0, 0.7, 2.0, 477.4, 479.0], 200)   # rows next to the top / bottom padding
        pri = (kps + rng.uniform(-6, 6, kps.shape)).astype(np.float32)
        for nl in (0, 1, 2, 3):
            out, st = trk.fbKltTracking(g0, g1, win, nl, 30.0, 0.5, kps, pri)
            eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, pri, win, nl, 30.0, 0.5, 30, 0.01)
            assert np.array_equal(st, est.astype(bool)), (t0, t1, nl)
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), (t0, t1, nl)


def test_pyramids_consumed_by_another_context(ctx, oracle, stream):
    """the mapper thread's pattern (src/ov2slam.cpp:175-180, src/mapper.cpp:76-97): a second context tracks on pyramids
    the front-end context built and has already released on its side; the front-end keeps building into its pool in the
    meantime.  ov2_pyr_release_from marks the consumer's stream on the buffer, so no build overwrites what the
    consumer still reads -- results stay bit-identical to the oracle."""
    ctx2 = fe.Context(0)
    try:
        trk2 = fe.FeatureTracker(ctx2, 30, 0.01)
        kps = synth.grid_keypoints(1500)
        frames = [(0, 3), (5, 2), (9, 12), (4, 7)]
        want = {}
        for t0, t1 in frames:
            o0, o1 = oracle.Pyramid(oracle.clahe(stream.left(t0), 3.0, 15, 9), 9, 3), oracle.Pyramid(oracle.clahe(stream.left(t1), 3.0, 15, 9), 9, 3)
            want[(t0, t1)] = oracle.fb_klt_tracking(o0, o1, kps, kps, 9, 3, 30.0, 0.5, 30, 0.01)
        d_k = ctx2.to_device(kps)
        for rep in range(6):
            outs = []
            for t0, t1 in frames:
                p0, p1 = fe.preprocess_image(ctx, stream.left(t0)), fe.preprocess_image(ctx, stream.left(t1))
                k0, k1 = p0.retain(), p1.retain()        # the keyframe's share
                p0.release(); p1.release()               # the front-end moves on ...
                d_p, d_s = ctx2.to_device(kps), ctx2.empty((len(kps),), np.uint8)
                trk2.fbKltTracking_dev(k0, k1, 9, 3, 30.0, 0.5, d_k, d_p, d_s, len(kps))
                k0.release_from(ctx2); k1.release_from(ctx2)
                for t in (1, 6, 11):                      # ... and keeps its pyramid pool busy
                    fe.preprocess_image(ctx, stream.left(t)).release()
                outs.append((d_p, d_s))
            ctx2.synchronize()
            for (t0, t1), (d_p, d_s) in zip(frames, outs):
                eout, est, _ = want[(t0, t1)]
                assert np.array_equal(d_s.get().astype(bool), est.astype(bool)), (rep, t0, t1)
                assert np.array_equal(d_p.get().view(np.uint32), eout.view(np.uint32)), (rep, t0, t1)
    finally:
        ctx2.close()


def test_fb_klt_edge_cases(ctx, oracle, stream):
    """points on/over the border, in flat (min-eig reject) regions, priors far outside, empty input."""
    I0 = stream.left(0).copy()
    I1 = stream.left(4).copy()
    I0[100:200, 100:300] = 128   # textureless block -> minEig < 1e-4 -> status 0
    I1[100:200, 100:300] = 128
    g0, g1 = fe.preprocess_image(ctx, I0, use_clahe=False), fe.preprocess_image(ctx, I1, use_clahe=False)
    o0, o1 = oracle.Pyramid(I0), oracle.Pyramid(I1)
    kps = np.array([[0, 0], [751, 479], [0.4, 479.6], [-3.0, 10.0], [760.0, 100.0], [150.0, 150.0], [200.0, 150.0],
                    [1.0, 1.0], [750.0, 478.0], [375.5, 239.5], [5.2, 300.7], [746.9, 5.1]], np.float32)
    pri = kps.copy()
    pri[9] = [900.0, -50.0]   # prior far outside the image -> out-of-range break in the iteration
    pri[10] = [-8.0, 300.0]
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for nl in (0, 1, 3):
        out, st = trk.fbKltTracking(g0, g1, 9, nl, 30.0, 0.5, kps, pri)
        eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, pri, 9, nl, 30.0, 0.5, 30, 0.01)
        assert np.array_equal(st, est.astype(bool)), nl
        assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), nl
        assert not st[5]      # flat region
    out, st = trk.fbKltTracking(g0, g1, 9, 3, 30.0, 0.5, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)


@pytest.mark.parametrize("win", [9, 11])
def test_fb_klt_saturated_images(ctx, oracle, win):
    """range limits of the packed int16 arithmetic (dot2 taps, pk_sub, w11 = -1 weights) and of the int32 fast path of
    the b1/b2 row sums.  Left third: binary 0/255 block noise with a real 1-px shift.  Middle: unrelated block noise.
    Right third: 0/255 vertical stripes (+ a grey line every 12 rows so that the 2x2 system is regular) tracked into a
    constant 255 image: in a stripe-edge column every row has diff = +8160 and Ix = +4080, so one lane's partial sum is
    ~2.7e8 > 2^27 and the kernel must take the exact wide-sum path."""
    rng = np.random.default_rng(42 + win)
    def blocks(k):
        return np.kron(rng.integers(0, 2, (480 // k + 1, 752 // k + 1)), np.ones((k, k)))[:480, :752].astype(np.uint8) * 255
    I0, I1 = blocks(3), blocks(3)
    I1[:, :250] = np.roll(I0, 1, axis=1)[:, :250]
    xs = np.arange(752)
    stripes = np.where((xs // 3) % 2 == 1, 255, 0).astype(np.uint8)
    I0[:, 500:] = stripes[None, 500:]
    I0[::12, 500:] = 128
    I1[:, 500:] = 255
    g0 = fe.preprocess_image(ctx, I0, use_clahe=False, klt_win_size=11)
    g1 = fe.preprocess_image(ctx, I1, use_clahe=False, klt_win_size=11)
    o0, o1 = oracle.Pyramid(I0, 11, 3), oracle.Pyramid(I1, 11, 3)
    kps = synth.grid_keypoints(900, seed=9)
    kps = (kps + rng.uniform(-0.5, 0.5, kps.shape)).astype(np.float32)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for nl in (0, 3):
        out, st = trk.fbKltTracking(g0, g1, win, nl, 30.0, 0.5, kps, kps)
        eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, kps, win, nl, 30.0, 0.5, 30, 0.01)
        assert np.array_equal(st, est.astype(bool)), nl
        assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), nl
    assert st[kps[:, 0] < 230].mean() > 0.5


@pytest.mark.parametrize("win,max_iter,eps,fb", [(7, 30, 0.01, 0.5), (11, 10, 0.03, 1.0), (5, 3, 0.01, 0.25), (9, 0, 0.01, 0.5)])
def test_fb_klt_other_parameters(ctx, oracle, stream, win, max_iter, eps, fb):
    I0, I1 = stream.left(2), stream.left(9)
    g0 = fe.preprocess_image(ctx, I0, use_clahe=False, klt_win_size=11)
    g1 = fe.preprocess_image(ctx, I1, use_clahe=False, klt_win_size=11)
    o0, o1 = oracle.Pyramid(I0, 11, 3), oracle.Pyramid(I1, 11, 3)
    kps = synth.grid_keypoints(500, seed=5)
    trk = fe.FeatureTracker(ctx, max_iter, eps)
    out, st = trk.fbKltTracking(g0, g1, win, 2, 30.0, fb, kps, kps)
    eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, kps, win, 2, 30.0, fb, max_iter, eps)
    assert np.array_equal(st, est.astype(bool))
    assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))


def test_klt_tracking_frame_two_stage(ctx, oracle, stream):
    """VisualFrontEnd::kltTracking batching: priors on 2 levels, failures re-queued on the full pyramid."""
    g0, g1, o0, o1 = _pyrs(ctx, oracle, stream, 0, 10)
    kps = synth.grid_keypoints(2000)
    gt = stream.flow(0, 10, kps)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for sigma, expect_p3p in ((1.0, False), (25.0, True)):   # bad motion model -> <33 % good -> priors dropped
        pri, has = synth.make_priors(kps, gt, sigma=sigma)
        out, st, p3p = trk.kltTracking(g0, g1, 9, 3, 30.0, 0.5, kps, pri, has)
        eout, est, ep3p = oracle.klt_tracking_frame(o0, o1, kps, pri, has)
        assert p3p == ep3p == expect_p3p
        assert np.array_equal(st, est.astype(bool))
        assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))


def test_fb_klt_full_size_properties(ctx, stream):
    """config-4 size (4000 kps): size-independent properties -- tracks land on the analytic flow, identity
    pair returns the keypoints themselves, batched call == per-image calls."""
    I0, I1 = stream.left(0), stream.left(8)
    g0, g1 = fe.preprocess_image(ctx, I0), fe.preprocess_image(ctx, I1)
    kps = synth.grid_keypoints(4000)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    out, st = trk.fbKltTracking(g0, g1, 9, 3, 30.0, 0.5, kps, kps)
    gt = stream.flow(0, 8, kps)
    e = np.linalg.norm(out[st] - gt[st], axis=1)
    assert st.mean() > 0.95 and np.median(e) < 0.2 and np.percentile(e, 99) < 1.0
    same, st2 = trk.fbKltTracking(g0, g0, 9, 3, 30.0, 0.5, kps, kps)
    assert st2.all() and np.abs(same - kps).max() < 1e-3
    # idempotence of the call itself (deterministic reductions)
    out3, st3 = trk.fbKltTracking(g0, g1, 9, 3, 30.0, 0.5, kps, kps)
    assert np.array_equal(out3.view(np.uint32), out.view(np.uint32)) and np.array_equal(st3, st)


@pytest.mark.parametrize("win", [5, 7, 9, 11])
def test_klt_large_call_eight_lane_path(ctx, oracle, stream, win):
    """calls with >= 65536 keypoints switch the kernels to 8 lanes per keypoint (eight keypoints per wave; for
    win 11 the columns beyond the eighth are dealt out pixel by pixel; win 9 takes the three-lane kernels from 4096
    keypoints on, test_klt_every_lane_mapping_matches_the_oracle forces the others): bit parity with the oracle on 66k keypoints,
    through both entry points, incl. the saturated stripe images that force the wide b-sum path."""
    rng = np.random.default_rng(win)
    n = 66000
    kps = np.stack([rng.uniform(4, 748, n), rng.uniform(4, 476, n)], 1).astype(np.float32)
    # windows hanging into the padding on every side / corner, and keypoints outside the image (LDS row staging reads
    # 16 bytes per window row from the dword below the window's first column)
    kps[:12] = [[0, 0], [751, 479], [0.4, 479.6], [-3.0, 10.0], [760.0, 100.0], [751.0, 0.0], [0.0, 479.0], [1.0, 1.0],
                [750.0, 478.0], [3.3, 240.2], [5.2, 300.7], [746.9, 5.1]]
    I0, I1 = stream.left(0).copy(), stream.left(5).copy()
    xs = np.arange(752)
    I0[300:, 500:] = np.where((xs // 3) % 2 == 1, 255, 0).astype(np.uint8)[None, 500:]
    I0[300::12, 500:] = 128
    I1[300:, 500:] = 255
    g0 = fe.preprocess_image(ctx, I0, use_clahe=False, klt_win_size=11)
    g1 = fe.preprocess_image(ctx, I1, use_clahe=False, klt_win_size=11)
    o0, o1 = oracle.Pyramid(I0, 11, 3), oracle.Pyramid(I1, 11, 3)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    out, st = trk.fbKltTracking(g0, g1, win, 3, 30.0, 0.5, kps, kps)
    eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, kps, win, 3, 30.0, 0.5, 30, 0.01)
    assert np.array_equal(st, est.astype(bool))
    assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))
    assert st[12:][(kps[12:, 0] < 480) | (kps[12:, 1] < 280)].mean() > 0.9
    if win == 9:
        gt = stream.flow(0, 5, kps)
        pri, has = synth.make_priors(kps, gt, sigma=1.0)
        out, st, p3p = trk.kltTracking(g0, g1, win, 3, 30.0, 0.5, kps, pri, has)
        eout, est, ep3p = oracle.klt_tracking_frame(o0, o1, kps, pri, has, win)
        assert p3p == ep3p
        assert np.array_equal(st, est.astype(bool))
        assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))


@pytest.mark.parametrize("lanes", [3, 8, 16])
def test_klt_every_lane_mapping_matches_the_oracle(ctx, oracle, stream, lanes):
    """ov2_klt_set_lanes forces one wave mapping (3 lanes per keypoint = the 9x9 path that derives the Scharr values in
    the kernel and stages the search region once per level pass; 8 / 16 = the gradient-plane kernels): each is bit-equal
    to the oracle on (a) large flow on few levels (the window leaves the staged region: re-staging), (b) windows over every
    border / flat regions / priors far outside, (c) the two-stage batching, (d) the batched frame call with image indices."""
    ctx.set_klt_lanes(lanes)
    try:
        trk = fe.FeatureTracker(ctx, 30, 0.01)
        # (a) flow of several pixels with 1 and 2 levels only
        g0, g1, o0, o1 = _pyrs(ctx, oracle, stream, 0, 12)
        kps = synth.grid_keypoints(1500)
        moved = 0
        for nl in (0, 1, 3):
            out, st = trk.fbKltTracking(g0, g1, 9, nl, 30.0, 0.5, kps, kps)
            eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, kps, 9, nl, 30.0, 0.5, 30, 0.01)
            assert np.array_equal(st, est.astype(bool)), nl
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), nl
            moved = max(moved, float(np.abs(out - kps)[:, 1].max()))
        assert moved > 3.0   # windows did travel more rows than the region's margin
        # (b) borders, flat block, wild priors
        I0, I1 = stream.left(0).copy(), stream.left(4).copy()
        I0[100:200, 100:300] = 128
        I1[100:200, 100:300] = 128
        h0, h1 = fe.preprocess_image(ctx, I0, use_clahe=False), fe.preprocess_image(ctx, I1, use_clahe=False)
        p0, p1 = oracle.Pyramid(I0), oracle.Pyramid(I1)
        rng = np.random.default_rng(lanes)
        edge = np.array([[0, 0], [751, 479], [0.4, 479.6], [-3.0, 10.0], [760.0, 100.0], [150.0, 150.0], [200.0, 150.0], [1.0, 1.0],
                         [750.0, 478.0], [375.5, 239.5], [5.2, 300.7], [746.9, 5.1], [3.9, 3.9], [747.2, 475.1]], np.float32)
        ring = np.concatenate([np.stack([rng.uniform(-2, 12, 40), rng.uniform(-2, 481, 40)], 1),
                               np.stack([rng.uniform(740, 754, 40), rng.uniform(-2, 481, 40)], 1),
                               np.stack([rng.uniform(-2, 753, 40), rng.uniform(-2, 12, 40)], 1),
                               np.stack([rng.uniform(-2, 753, 40), rng.uniform(468, 482, 40)], 1)]).astype(np.float32)
        kps = np.concatenate([edge, ring])
        pri = kps + rng.normal(0, 1.5, kps.shape).astype(np.float32)
        pri[9] = [900.0, -50.0]
        pri[10] = [-8.0, 300.0]
        for nl in (0, 3):
            out, st = trk.fbKltTracking(h0, h1, 9, nl, 30.0, 0.5, kps, pri)
            eout, est, _ = oracle.fb_klt_tracking(p0, p1, kps, pri, 9, nl, 30.0, 0.5, 30, 0.01)
            assert np.array_equal(st, est.astype(bool)), nl
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), nl
        # (c) two-stage batching incl. the 33 % rule
        g0, g1, o0, o1 = _pyrs(ctx, oracle, stream, 0, 10)
        kps = synth.grid_keypoints(2000)
        gt = stream.flow(0, 10, kps)
        for sigma, expect_p3p in ((1.0, False), (25.0, True)):
            pri, has = synth.make_priors(kps, gt, sigma=sigma)
            out, st, p3p = trk.kltTracking(g0, g1, 9, 3, 30.0, 0.5, kps, pri, has)
            eout, est, ep3p = oracle.klt_tracking_frame(o0, o1, kps, pri, has)
            assert p3p == ep3p == expect_p3p
            assert np.array_equal(st, est.astype(bool))
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32))
    finally:
        ctx.set_klt_lanes(0)


@pytest.mark.parametrize("w,h", [(101, 67), (64, 40), (233, 121)])
def test_klt_three_lane_path_on_small_odd_images(ctx, oracle, w, h):
    """the three-lane kernels address the planes with clamped 48-byte region rows and 12-byte template rows: small and
    odd geometries (rows shorter than a region, levels smaller than a window) against the oracle, keypoints everywhere
    incl. outside the image"""
    rng = np.random.default_rng(w)
    tex = synth.base_texture(h + 40, w + 40, seed=w)
    I0 = np.ascontiguousarray(tex[10:10 + h, 10:10 + w]).astype(np.uint8)
    I1 = np.ascontiguousarray(tex[11:11 + h, 12:12 + w]).astype(np.uint8)
    g0, g1 = fe.preprocess_image(ctx, I0, use_clahe=False), fe.preprocess_image(ctx, I1, use_clahe=False)
    o0, o1 = oracle.Pyramid(I0), oracle.Pyramid(I1)
    n = 4500
    kps = np.stack([rng.uniform(-6, w + 6, n), rng.uniform(-6, h + 6, n)], 1).astype(np.float32)
    pri = kps + rng.normal(0, 1.5, kps.shape).astype(np.float32)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    ctx.set_klt_lanes(3)
    try:
        for nl in (0, 2, 3):
            out, st = trk.fbKltTracking(g0, g1, 9, nl, 30.0, 0.5, kps, pri)
            eout, est, _ = oracle.fb_klt_tracking(o0, o1, kps, pri, 9, nl, 30.0, 0.5, 30, 0.01)
            assert np.array_equal(st, est.astype(bool)), nl
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), nl
        assert st.mean() > 0.3
    finally:
        ctx.set_klt_lanes(0)


@pytest.mark.parametrize("after,groups,pickup", [(0, 0, 0), (1, 19, 0), (1, 20, 1), (2, 8, 3), (6, 5, 8), (30, 20, 1)])
def test_klt_yield_resume_is_bit_identical(ctx, oracle, stream, after, groups, pickup):
    """ov2_klt_set_yield: the stragglers of a level pass leave their wave and a second launch continues each one's iteration
    sequence (klt_rec in klt.hip).  Whatever the setting -- never, (1, 19/20) = nearly every keypoint yields at its second
    iteration of EVERY pass incl. the backward one, (6, 5) -- positions, statuses and the 33 % flag equal the
    oracle bit for bit, with and without waves of the first launch taking stragglers themselves (pickup): easy priors, priors 25 px off (33 % rule, failures re-queued), borders, and the stereo call."""
    ctx.set_klt_yield(after, groups, pickup)
    try:
        trk = fe.FeatureTracker(ctx, 30, 0.01)
        g0, g1, o0, o1 = _pyrs(ctx, oracle, stream, 0, 10)
        kps = synth.grid_keypoints(6000, seed=after + 3)
        gt = stream.flow(0, 10, kps)
        for sigma, expect_p3p in ((1.0, False), (4.0, False), (25.0, True)):
            pri, has = synth.make_priors(kps, gt, sigma=sigma, seed=groups)
            out, st, p3p = trk.kltTracking(g0, g1, 9, 3, 30.0, 0.5, kps, pri, has)
            eout, est, ep3p = oracle.klt_tracking_frame(o0, o1, kps, pri, has)
            assert p3p == ep3p == expect_p3p
            assert np.array_equal(st, est.astype(bool)), sigma
            assert np.array_equal(out.view(np.uint32), eout.view(np.uint32)), sigma
        # keypoints over every border, wild priors
        rng = np.random.default_rng(7)
        kb = np.stack([rng.uniform(-4, 756, 3000), rng.uniform(-4, 484, 3000)], 1).astype(np.float32)
        pb = kb + rng.normal(0, 3.0, kb.shape).astype(np.float32)
        hb = (rng.uniform(size=3000) < 0.6).astype(np.uint8)
        out, st, p3p = trk.kltTracking(g0, g1, 9, 3, 30.0, 0.5, kb, pb, hb)
        eout, est, ep3p = oracle.klt_tracking_frame(o0, o1, kb, pb, hb)
        assert p3p == ep3p and np.array_equal(st, est.astype(bool)) and np.array_equal(out.view(np.uint32), eout.view(np.uint32))
        # left -> right (no 33 % rule, failures re-queued with the updated prior)
        gl, gr = fe.preprocess_image(ctx, stream.left(2)), fe.preprocess_image(ctx, stream.right(2))
        ol, orr = oracle.Pyramid(oracle.clahe(stream.left(2))), oracle.Pyramid(oracle.clahe(stream.right(2)))
        sg = stream.stereo_gt(kps).astype(np.float32)
        spri, shas = synth.make_priors(kps, sg, sigma=3.0, seed=5)
        rxy, rst = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, spri, shas, rectified=True)
        exy, est2 = oracle.stereo_matching(ol, orr, kps, spri, shas, rectified=True)
        assert np.array_equal(rst, est2) and np.array_equal(rxy.view(np.uint32), exy.view(np.uint32))
    finally:
        ctx.set_klt_yield(0, 0)
