"""Host-side mirror of the reference's front-end operator surface, over the C ABI (libov2hip.so).

Names and argument order follow the reference so the parity tests read like its call sites:
  FeatureTracker.fbKltTracking      <- include/feature_tracker.hpp:45, src/feature_tracker.cpp:35-137
  preprocess_image / build pyramid  <- VisualFrontEnd::preprocessImage src/visual_front_end.cpp:1143-1177
  klt_tracking                      <- VisualFrontEnd::kltTracking     src/visual_front_end.cpp:132-275
Everything here is plumbing (ctypes + numpy); all arithmetic runs in the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _lib


def _check(ctx_handle, status):
    if status != 0:
        lib = _lib.load()
        msg = lib.ov2_last_error(ctx_handle).decode() if ctx_handle else ""
        raise _lib.Ov2Error(f"{lib.ov2_status_string(status).decode()} ({status}): {msg}")


class Context:
    """one device + one HIP stream (reference: one per calling thread, T1 front-end / T3 mapper)."""

    def __init__(self, device=0, high_priority=False):
        self.lib = _lib.load()
        h = C.c_void_p()
        st = self.lib.ov2_ctx_create_ex(device, int(bool(high_priority)), C.byref(h))
        if st != 0:
            raise _lib.Ov2Error(f"ov2_ctx_create(device={device}): {self.lib.ov2_status_string(st).decode()} -- "
                                "an MI355X/gfx950 device is required, there is no CPU fallback")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.ov2_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_klt_lanes(self, lanes):
        """ov2_klt_set_lanes: 0 = by call size, 3 / 8 / 16 = lanes per keypoint in the tracking kernels"""
        _check(self.h, self.lib.ov2_klt_set_lanes(self.h, int(lanes)))

    def set_klt_yield(self, after, groups, pickup=0):
        """ov2_klt_set_yield: stragglers of a level pass leave their wave after `after` iterations once <= `groups` run"""
        _check(self.h, self.lib.ov2_klt_set_yield(self.h, int(after), int(groups), int(pickup)))

    def synchronize(self):
        _check(self.h, self.lib.ov2_ctx_synchronize(self.h))

    def timer_start(self):
        _check(self.h, self.lib.ov2_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        _check(self.h, self.lib.ov2_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def kernel_timing(self, on):
        _check(self.h, self.lib.ov2_ktime_enable(self.h, int(bool(on))))

    def kernel_times(self):
        """{kernel name: (total_ms, launches)} since the last call (synchronises)."""
        names = (C.c_char_p * 48)()
        tot = (C.c_double * 48)()
        cnt = (C.c_longlong * 48)()
        n = C.c_int()
        _check(self.h, self.lib.ov2_ktime_report(self.h, 48, names, tot, cnt, C.byref(n)))
        return {names[i].decode(): (tot[i], cnt[i]) for i in range(n.value)}

    # -- device arrays ------------------------------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        _check(self.h, self.lib.ov2_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        if p:
            self.lib.ov2_dev_free(self.h, p)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        d = DeviceArray(self, arr.shape, arr.dtype)
        _check(self.h, self.lib.ov2_memcpy_h2d(self.h, d.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return d

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, np.dtype(dtype))


class DeviceArray:
    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.alloc(max(self.nbytes, 1))

    def get(self):
        out = np.empty(self.shape, self.dtype)
        _check(self.ctx.h, self.ctx.lib.ov2_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        _check(self.ctx.h, self.ctx.lib.ov2_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def copy_from(self, other):
        """asynchronous device-to-device copy on the ctx stream"""
        assert other.nbytes == self.nbytes
        _check(self.ctx.h, self.ctx.lib.ov2_memcpy_d2d(self.ctx.h, self.ptr, other.ptr, self.nbytes))

    def __del__(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "h", None):
            self.ctx.free(self.ptr)
            self.ptr = None


class Images:
    """batch of same-size u8 images resident in HBM."""

    def __init__(self, ctx, batch, w, h):
        self.ctx, self.batch, self.w, self.h = ctx, batch, w, h
        p = C.c_void_p()
        _check(ctx.h, ctx.lib.ov2_images_create(ctx.h, batch, w, h, C.byref(p)))
        self.h_ = p

    def upload(self, b, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.shape == (self.h, self.w)
        _check(self.ctx.h, self.ctx.lib.ov2_images_upload(self.ctx.h, self.h_, b, img.ctypes.data_as(C.c_void_p), self.w))

    def __del__(self):
        if getattr(self, "h_", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.ov2_images_destroy(self.h_)
            self.h_ = None


class Pyramid:
    """ref-counted device pyramid handle: the counterpart of the std::vector<cv::Mat> the reference swaps
    between prev_pyr_/cur_pyr_ and shares with the mapper thread."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        self.nlevels = ctx.lib.ov2_pyr_nlevels(handle)
        self.batch = ctx.lib.ov2_pyr_batch(handle)

    def release(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.ov2_pyr_release(self.h)
        self.h = None

    def retain(self):
        """a second owner (the Keyframe handed to the mapper thread, src/ov2slam.cpp:175-180): returns a new handle object"""
        self.ctx.lib.ov2_pyr_retain(self.h)
        return Pyramid(self.ctx, self.h)

    def release_from(self, user_ctx):
        """release a reference whose readers ran on another context's stream (ov2_pyr_release_from)"""
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.ov2_pyr_release_from(user_ctx.h, self.h)
        self.h = None

    def __del__(self):
        self.release()

    def level_size(self, l):
        w, h, p = C.c_int(), C.c_int(), C.c_int()
        _check(self.ctx.h, self.ctx.lib.ov2_pyr_level_size(self.h, l, C.byref(w), C.byref(h), C.byref(p)))
        return w.value, h.value, p.value

    def level(self, l, b=0):
        """(img (h+2p, w+2p) u8, grad (h+2p, w+2p, 2) i16, w, h, pad) downloaded to host."""
        w, h, p = self.level_size(l)
        img = np.empty((h + 2 * p, w + 2 * p), np.uint8)
        grad = np.empty((h + 2 * p, w + 2 * p, 2), np.int16)
        _check(self.ctx.h, self.ctx.lib.ov2_pyr_download_level(self.ctx.h, self.h, b, l, img.ctypes.data_as(C.c_void_p),
                                                             grad.ctypes.data_as(C.c_void_p)))
        return img, grad, w, h, p


def clahe_tiles(w, h, tilesize=50):
    """src/ov2slam.cpp:85-89"""
    return w // tilesize, h // tilesize


def preprocess_image(ctx, img_raw, use_clahe=True, fclahe_val=3.0, klt_win_size=9, nklt_pyr_lvl=3, tiles=None):
    """VisualFrontEnd::preprocessImage: CLAHE (optional) + buildOpticalFlowPyramid -> Pyramid."""
    img = np.ascontiguousarray(img_raw, dtype=np.uint8)
    h, w = img.shape
    tx, ty = tiles if tiles is not None else clahe_tiles(w, h)
    p = C.c_void_p()
    _check(ctx.h, ctx.lib.ov2_pyramid_build(ctx.h, img.ctypes.data_as(C.c_void_p), w, h, w, klt_win_size, nklt_pyr_lvl,
                                            int(bool(use_clahe)), float(fclahe_val), tx, ty, C.byref(p)))
    return Pyramid(ctx, p)


def preprocess_images(ctx, images, use_clahe=True, fclahe_val=3.0, klt_win_size=9, nklt_pyr_lvl=3, tiles=None):
    """batched, device-resident, asynchronous preprocessImage (one pyramid per image of `images`)."""
    tx, ty = tiles if tiles is not None else clahe_tiles(images.w, images.h)
    p = C.c_void_p()
    _check(ctx.h, ctx.lib.ov2_pyramid_build_images(ctx.h, images.h_, klt_win_size, nklt_pyr_lvl, int(bool(use_clahe)),
                                                   float(fclahe_val), tx, ty, C.byref(p)))
    return Pyramid(ctx, p)


class FeatureTracker:
    """mirror of the reference FeatureTracker (include/feature_tracker.hpp:32-56)."""

    def __init__(self, ctx, nmax_iter=30, fmax_px_precision=0.01):
        self.ctx = ctx
        self.nmax_iter = int(nmax_iter)              # klt_convg_crit_.maxCount
        self.fmax_px_precision = float(fmax_px_precision)  # klt_convg_crit_.epsilon

    def fbKltTracking(self, vprevpyr, vcurpyr, nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist, vkps, vpriorkps):
        """returns (vpriorkps_out (n,2) f32, vkpstatus (n,) bool).  Empty input returns empty outputs
        (src/feature_tracker.cpp:43-46)."""
        kps = np.ascontiguousarray(vkps, dtype=np.float32).reshape(-1, 2)
        pri = np.ascontiguousarray(vpriorkps, dtype=np.float32).reshape(-1, 2).copy()
        n = kps.shape[0]
        if pri.shape[0] != n:
            raise ValueError("vkps and vpriorkps differ in length")
        st = np.zeros(n, np.uint8)
        lib, ctx = self.ctx.lib, self.ctx
        _check(ctx.h, lib.ov2_klt_track_fb(ctx.h, vprevpyr.h, vcurpyr.h, nwinsize, nbpyrlvl, self.nmax_iter,
                                           self.fmax_px_precision, ferr, fmax_fbklt_dist, n,
                                           kps.ctypes.data_as(C.c_void_p), pri.ctypes.data_as(C.c_void_p),
                                           st.ctypes.data_as(C.c_void_p)))
        return pri, st.astype(bool)

    def fbKltTracking_dev(self, vprevpyr, vcurpyr, nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist, d_kps, d_priors, d_status,
                          n, d_img_idx=None, d_iters=None):
        """device-resident asynchronous form; arguments are DeviceArray (or raw pointers)."""
        ctx = self.ctx
        ptr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceArray) else a)
        _check(ctx.h, ctx.lib.ov2_klt_track_fb_dev(ctx.h, vprevpyr.h, vcurpyr.h, nwinsize, nbpyrlvl, self.nmax_iter,
                                                   self.fmax_px_precision, ferr, fmax_fbklt_dist, n, ptr(d_kps),
                                                   ptr(d_priors), ptr(d_status), ptr(d_img_idx), ptr(d_iters)))

    def kltTracking_dev(self, vprevpyr, vcurpyr, nwinsize, nklt_pyr_lvl, ferr, fmax_fbklt_dist, d_kps, d_prior,
                        d_has_prior, d_out_xy, d_out_status, n, d_img_idx=None, d_p3p_req=None, d_iters=None):
        """VisualFrontEnd::kltTracking's two stages fused on device (src/visual_front_end.cpp:132-275)."""
        ctx = self.ctx
        ptr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceArray) else a)
        _check(ctx.h, ctx.lib.ov2_klt_tracking_frame_dev(ctx.h, vprevpyr.h, vcurpyr.h, nwinsize, nklt_pyr_lvl,
                                                         self.nmax_iter, self.fmax_px_precision, ferr, fmax_fbklt_dist,
                                                         n, ptr(d_kps), ptr(d_prior), ptr(d_has_prior), ptr(d_img_idx),
                                                         ptr(d_out_xy), ptr(d_out_status), ptr(d_p3p_req), ptr(d_iters)))

    def kltTracking(self, vprevpyr, vcurpyr, nwinsize, nklt_pyr_lvl, ferr, fmax_fbklt_dist, vkps, vpriors, has_prior):
        """host-array convenience around kltTracking_dev: returns (out_xy, status(bool), p3p_req)."""
        ctx = self.ctx
        kps = np.ascontiguousarray(vkps, np.float32).reshape(-1, 2)
        n = kps.shape[0]
        if n == 0:
            return kps.copy(), np.zeros(0, bool), False
        d_k = ctx.to_device(kps)
        d_p = ctx.to_device(np.ascontiguousarray(vpriors, np.float32).reshape(-1, 2))
        d_h = ctx.to_device(np.ascontiguousarray(has_prior, np.uint8))
        d_o = ctx.empty((n, 2), np.float32)
        d_s = ctx.empty((n,), np.uint8)
        d_r = ctx.empty((max(vprevpyr.batch, 1),), np.int32)
        self.kltTracking_dev(vprevpyr, vcurpyr, nwinsize, nklt_pyr_lvl, ferr, fmax_fbklt_dist, d_k, d_p, d_h, d_o, d_s,
                             n, None, d_r, None)
        ctx.synchronize()
        return d_o.get(), d_s.get().astype(bool), bool(d_r.get()[0])


    def getLineMinSAD(self, vleftpyr, vrightpyr, level, pts_xy, nwinsize=7, bgoleft=True):
        """FeatureTracker::getLineMinSAD (src/feature_tracker.cpp:140-213) for n points of pyramid level `level`
        (coordinates of that level).  returns (xprior (n,) f32, l1err (n,) f32)."""
        ctx = self.ctx
        pts = np.ascontiguousarray(pts_xy, np.float32).reshape(-1, 2)
        n = len(pts)
        xp, er = np.full(n, -1.0, np.float32), np.zeros(n, np.float32)
        _check(ctx.h, ctx.lib.ov2_line_min_sad(ctx.h, vleftpyr.h, vrightpyr.h, int(level), int(nwinsize), int(bool(bgoleft)), n,
                                               pts.ctypes.data_as(C.c_void_p), xp.ctypes.data_as(C.c_void_p),
                                               er.ctypes.data_as(C.c_void_p)))
        return xp, er

    def stereoMatching(self, vleftpyr, vrightpyr, nwinsize, nklt_pyr_lvl, ferr, fmax_fbklt_dist, vkps, vpriors, has_prior,
                       lunpx=None, rectified=True, F_rl=None, right_cam=None):
        """tracking + epipolar gate of MapManager::stereoMatching (src/map_manager.cpp:493-604) on flat arrays.
        right_cam: ba_types.CamModelC of the right camera (its undistortImagePoint feeds the gate, :586) or None.
        returns (right pixels (n,2) f32, status (n,) bool)."""
        ctx = self.ctx
        kps = np.ascontiguousarray(vkps, np.float32).reshape(-1, 2)
        pri = np.ascontiguousarray(vpriors, np.float32).reshape(-1, 2)
        hp = np.ascontiguousarray(has_prior, np.uint8)
        n = len(kps)
        lu = None if lunpx is None else np.ascontiguousarray(lunpx, np.float32).reshape(-1, 2)
        F = None if F_rl is None else np.ascontiguousarray(F_rl, np.float64).reshape(9)
        out, st = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8)
        _check(ctx.h, ctx.lib.ov2_stereo_matching(ctx.h, vleftpyr.h, vrightpyr.h, nwinsize, nklt_pyr_lvl, self.nmax_iter,
                                                  self.fmax_px_precision, ferr, fmax_fbklt_dist, n,
                                                  kps.ctypes.data_as(C.c_void_p), pri.ctypes.data_as(C.c_void_p),
                                                  hp.ctypes.data_as(C.c_void_p),
                                                  None if lu is None else lu.ctypes.data_as(C.c_void_p), int(bool(rectified)),
                                                  None if F is None else F.ctypes.data_as(C.c_void_p),
                                                  None if right_cam is None else C.addressof(right_cam),
                                                  out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)))
        return out, st.astype(bool)

    def stereoMatching_dev(self, vleftpyr, vrightpyr, nwinsize, nklt_pyr_lvl, ferr, fmax_fbklt_dist, d_kps, d_prior,
                           d_has_prior, d_out_rxy, d_out_status, n, d_img_idx=None, d_lunpx=None, rectified=True, F_rl=None,
                           d_iters=None, right_cam=None):
        """device-resident asynchronous form (ov2_stereo_matching_dev)"""
        ctx = self.ctx
        ptr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceArray) else a)
        F = None if F_rl is None else np.ascontiguousarray(F_rl, np.float64).reshape(9)
        _check(ctx.h, ctx.lib.ov2_stereo_matching_dev(ctx.h, vleftpyr.h, vrightpyr.h, nwinsize, nklt_pyr_lvl, self.nmax_iter,
                                                      self.fmax_px_precision, ferr, fmax_fbklt_dist, n, ptr(d_kps),
                                                      ptr(d_prior), ptr(d_has_prior), ptr(d_img_idx), ptr(d_lunpx),
                                                      int(bool(rectified)),
                                                      None if F is None else F.ctypes.data_as(C.c_void_p),
                                                      None if right_cam is None else C.addressof(right_cam), ptr(d_out_rxy),
                                                      ptr(d_out_status), ptr(d_iters)))


class FeatureExtractor:
    """mirror of the reference FeatureExtractor for the two grid detectors (include/feature_extractor.hpp:37-46):
    keeps the adaptive thresholds dmaxquality_ / nfast_th_ across calls exactly as the reference object does."""

    def __init__(self, ctx, nmaxdist=35, dmaxquality=0.001, nfast_th=10):
        self.ctx, self.nmaxdist_ = ctx, int(nmaxdist)
        self.dmaxquality_ = float(dmaxquality)
        self.nfast_th_ = int(nfast_th)

    def _detect(self, mode, pyr, vcurkps, roi, b, subpix):
        cur = np.ascontiguousarray(vcurkps, np.float32).reshape(-1, 2)
        w, h, _ = pyr.level_size(0)
        cap = max(1, (w // self.nmaxdist_) * (h // self.nmaxdist_)) * 2 + 2
        out = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        th = C.c_double(self.dmaxquality_ if mode == 1 else float(self.nfast_th_))
        r = None if roi is None else np.ascontiguousarray(roi, np.int32)
        _check(self.ctx.h, self.ctx.lib.ov2_detect_grid(self.ctx.h, pyr.h, b, self.nmaxdist_, mode, C.byref(th), len(cur),
                                                       cur.ctypes.data_as(C.c_void_p),
                                                       None if r is None else r.ctypes.data_as(C.c_void_p), int(subpix),
                                                       C.byref(n), out.ctypes.data_as(C.c_void_p)))
        if mode == 1:
            self.dmaxquality_ = th.value
        else:
            self.nfast_th_ = int(th.value)
        return out[:n.value].copy()

    def detectSingleScale(self, pyr, vcurkps, roi=None, b=0, subpix=True):
        """src/feature_extractor.cpp:288-440 on level 0 of `pyr` (the CLAHE'd frame)."""
        return self._detect(1, pyr, vcurkps, roi, b, subpix)

    def detectGridFAST(self, pyr, vcurkps, roi=None, b=0, subpix=True):
        """src/feature_extractor.cpp:443-570."""
        return self._detect(0, pyr, vcurkps, roi, b, subpix)


def detect_grid_batch_dev(ctx, pyr, cell, mode, d_thresh, n_cur, d_cur_xy, d_cur_img, d_cur_valid, d_n_out, d_out_xy,
                          out_cap, roi=None, subpix=True):
    """device-resident, asynchronous keyframe detection on every image of `pyr` (ov2_detect_grid_batch_dev): thresholds,
    keypoints, counts and corners all stay in HBM; nothing is synchronised."""
    ptr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceArray) else a)
    r = None if roi is None else np.ascontiguousarray(roi, np.int32)
    _check(ctx.h, ctx.lib.ov2_detect_grid_batch_dev(ctx.h, pyr.h, cell, mode, ptr(d_thresh), int(n_cur), ptr(d_cur_xy),
                                                    ptr(d_cur_img), ptr(d_cur_valid),
                                                    None if r is None else r.ctypes.data_as(C.c_void_p), int(subpix),
                                                    ptr(d_n_out), ptr(d_out_xy), int(out_cap)))


def detect_grid_batch(ctx, pyr, cell, mode, thresh, cur_list, roi=None, subpix=True):
    """every image of `pyr` in one call.  thresh: float array [B] (updated in place), cur_list: list of (n_b,2) arrays.
    returns list of (n_b, 2) float32 arrays."""
    B = pyr.batch
    w, h, _ = pyr.level_size(0)
    cap = max(1, (w // cell) * (h // cell)) * 2
    n_cur = np.ascontiguousarray([len(c) for c in cur_list], np.int32)
    cur = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32).reshape(-1, 2) for c in cur_list]) if B else
                               np.zeros((0, 2)), np.float32)
    out = np.zeros((B, cap, 2), np.float32)
    n_out = np.zeros(B, np.int32)
    r = None if roi is None else np.ascontiguousarray(roi, np.int32)
    assert thresh.dtype == np.float64 and len(thresh) == B
    _check(ctx.h, ctx.lib.ov2_detect_grid_batch(ctx.h, pyr.h, cell, mode, thresh.ctypes.data_as(C.POINTER(C.c_double)),
                                                n_cur.ctypes.data_as(C.POINTER(C.c_int)), cur.ctypes.data_as(C.c_void_p),
                                                None if r is None else r.ctypes.data_as(C.c_void_p), int(subpix),
                                                n_out.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.c_void_p), cap))
    return [out[b, :n_out[b]].copy() for b in range(B)]
