/*
 * ov2slam_hip.h -- C ABI of libov2hip.so: the MI355X (gfx950) implementation of OV2SLAM's per-frame
 * front-end (CLAHE + optical-flow pyramid + forward-backward pyramidal KLT) and keyframe-rate local
 * bundle adjustment.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * The reference (chngdickson/ov2slam) has no FFI of its own: the path sits behind C++ member functions
 * with OpenCV/Eigen types.  Each entry point below names the reference call it replaces (file:line in
 * /root/reference); INTEGRATION.md shows the C++ adapter a maintainer adds on the reference side.
 *
 * Conventions
 *   - every call takes an ov2_ctx (device + one HIP stream + scratch).  One ctx per calling thread:
 *     the reference calls fbKltTracking concurrently from the front-end and the mapper threads
 *     (src/visual_front_end.cpp:196 and src/map_manager.cpp:510), so contexts are independent.
 *   - return value: OV2_OK (0) or a negative ov2_status; ov2_last_error(ctx) gives the message.
 *     No exceptions, no exit(); failures of individual keypoints are reported per keypoint in
 *     `status` exactly as the reference does (src/feature_tracker.cpp:79-131).
 *   - `_dev` variants take DEVICE pointers, enqueue on the ctx stream and return without
 *     synchronising; host-pointer variants copy in/out and synchronise before returning.
 *   - pixels: float32 (x,y) pairs, same as cv::Point2f.
 */
#ifndef OV2SLAM_HIP_H
#define OV2SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int ov2_status;
enum {
    OV2_OK = 0,
    OV2_ERR_INVALID = -1,   /* bad argument (null handle, size mismatch, unsupported window ...) */
    OV2_ERR_HIP = -2,       /* a HIP runtime call failed */
    OV2_ERR_NOMEM = -3,
    OV2_ERR_NODEVICE = -4,  /* no gfx950 device visible: the product path never falls back to CPU */
    OV2_ERR_UNSUPPORTED = -5
};

typedef struct ov2_ctx ov2_ctx;        /* device, stream, scratch, buffer pool */
typedef struct ov2_images ov2_images;  /* batch of B same-size u8 images resident in HBM */
typedef struct ov2_pyr ov2_pyr;        /* ref-counted device pyramid(s): B x levels x {u8 image, s16x2 gradient} */

/* ---- context --------------------------------------------------------------------------------- */
ov2_status ov2_ctx_create(int device, ov2_ctx **out);
/* high_priority != 0 creates the ctx stream at the highest HIP stream priority: for the latency-critical, small-grid
 * caller (the Estimator / localBA thread) that shares the GPU with the front-end's large grids. */
ov2_status ov2_ctx_create_ex(int device, int high_priority, ov2_ctx **out);
void ov2_ctx_destroy(ov2_ctx *ctx);
const char *ov2_last_error(const ov2_ctx *ctx);
const char *ov2_status_string(ov2_status s);
ov2_status ov2_ctx_synchronize(ov2_ctx *ctx);
/* hipEvent pair on the ctx stream (used by bench.py: torch.cuda.Event would only see torch's stream) */
ov2_status ov2_timer_start(ov2_ctx *ctx);
ov2_status ov2_timer_stop(ov2_ctx *ctx, float *elapsed_ms);   /* synchronises on the stop event */
/* optional per-kernel timing: when enabled every kernel launch of this ctx is bracketed by a hipEvent pair on the
 * ctx stream; ov2_ktime_report synchronises, returns per-kernel {name, total ms, launches} since the last report
 * and resets.  Used by bench.py for roofline.achieved; off by default (it perturbs back-to-back launches). */
ov2_status ov2_ktime_enable(ov2_ctx *ctx, int on);
ov2_status ov2_ktime_report(ov2_ctx *ctx, int max_kernels, const char **names, double *total_ms,
                            long long *launches, int *n_out);
/* raw device memory for callers that keep keypoints resident (C++ hosts without torch) */
ov2_status ov2_dev_alloc(ov2_ctx *ctx, size_t bytes, void **dptr);
ov2_status ov2_dev_free(ov2_ctx *ctx, void *dptr);
ov2_status ov2_memcpy_h2d(ov2_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);  /* sync */
ov2_status ov2_memcpy_d2h(ov2_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);  /* sync */
ov2_status ov2_memcpy_d2d(ov2_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);   /* async, ctx stream */

/* ---- images ---------------------------------------------------------------------------------- */
ov2_status ov2_images_create(ov2_ctx *ctx, int batch, int w, int h, ov2_images **out);
ov2_status ov2_images_upload(ov2_ctx *ctx, ov2_images *imgs, int b, const uint8_t *host, int stride);
void ov2_images_destroy(ov2_images *imgs);

/* ---- pyramid --------------------------------------------------------------------------------- */
/* Replaces VisualFrontEnd::preprocessImage's  pclahe_->apply(img_raw, cur_img_)  +
 * cv::buildOpticalFlowPyramid(cur_img_, cur_pyr_, Size(win,win), max_level)
 * (src/visual_front_end.cpp:1159,1172; right image: src/mapper.cpp:76,81).
 * use_clahe=0 skips CLAHE (`use_clahe: 0`).  tiles = (w/50, h/50) in the reference (src/ov2slam.cpp:85-89).
 * The returned pyramid holds levels 0..L (L <= max_level, same early stop as OpenCV), each level a u8 image
 * padded by `win` px of REFLECT_101 and an int16 (Ix,Iy) Scharr gradient padded by `win` px of zeros.
 * The gradient planes (withDerivatives = true in OpenCV: 4 of the 5 bytes per pixel) are written ON DEMAND, by the first
 * consumer that reads them (the tracking kernels for other window sizes than 9, or a forced 8 / 16-lane mapping,
 * ov2_pyr_download_level with a gradient pointer); the 9 x 9 path derives the same Scharr integers from
 * the image windows inside its kernel and a build that only feeds it never writes them.
 * Handles are ref-counted because the reference shares pyramids by value between threads
 * (src/ov2slam.cpp:175-180). */
ov2_status ov2_pyramid_build(ov2_ctx *ctx, const uint8_t *img, int w, int h, int stride, int win, int max_level,
                             int use_clahe, float clahe_clip, int tiles_x, int tiles_y, ov2_pyr **out);
/* batched, device-resident form: one pyramid per image of `imgs`, no host traffic, asynchronous. */
ov2_status ov2_pyramid_build_images(ov2_ctx *ctx, const ov2_images *imgs, int win, int max_level, int use_clahe,
                                    float clahe_clip, int tiles_x, int tiles_y, ov2_pyr **out);
void ov2_pyr_retain(ov2_pyr *p);
void ov2_pyr_release(ov2_pyr *p);
/* release a reference whose readers were enqueued on ANOTHER context's stream (the mapper thread's context: Keyframe
 * holds the pyramids for Mapper::run, include/mapper.hpp:39-85, src/mapper.cpp:76-97): the buffer is not reused before
 * those readers have run.  One foreign consumer context per pyramid. */
void ov2_pyr_release_from(ov2_ctx *user, ov2_pyr *p);
int ov2_pyr_batch(const ov2_pyr *p);
int ov2_pyr_nlevels(const ov2_pyr *p);
ov2_status ov2_pyr_level_size(const ov2_pyr *p, int level, int *w, int *h, int *pad);
/* copies level `level` of pyramid `b` to host as tight padded planes: img (h+2pad)x(w+2pad) u8,
 * grad (h+2pad)x(w+2pad)x2 int16.  Either pointer may be NULL.  For parity tests / debugging. */
ov2_status ov2_pyr_download_level(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img, int16_t *grad);

/* ---- KLT ------------------------------------------------------------------------------------- */
/* Replaces FeatureTracker::fbKltTracking(vprevpyr, vcurpyr, nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist,
 * vkps, vpriorkps, vkpstatus)  (include/feature_tracker.hpp:45, src/feature_tracker.cpp:35-137):
 * forward pyramidal LK prev->cur from the priors on levels nlevels..0 (flags USE_INITIAL_FLOW |
 * LK_GET_MIN_EIGENVALS), reject !status / err > err_th / not inBorder(1 px), backward LK cur->prev on level 0
 * started at the original keypoint, reject |kp - back| > fb_th.  max_iter/eps are the TermCriteria of
 * include/feature_tracker.hpp:40.  priors_xy is in/out and receives the forward result for every point,
 * status[i] in {0,1}.  n == 0 returns OV2_OK and touches nothing (src/feature_tracker.cpp:43-46);
 * nlevels is clamped to the pyramid (:50-52).  Supported window: odd, 3 <= win <= 11 (reference: 9). */
ov2_status ov2_klt_track_fb(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels,
                            int max_iter, float eps, float err_th, float fb_th, int n, const float *kps_xy,
                            float *priors_xy, uint8_t *status);
/* device-resident, asynchronous, batched form.  img_idx (n int32, may be NULL = all 0) selects which pyramid
 * of the batch each keypoint lives in.  d_iters
 * (n uint32, may be NULL) receives per keypoint a work word: low 16 bits = LK iterations executed (forward
 * levels + backward pass), high 16 bits = level passes started (template fetches) (drives the algorithmic-bytes model of bench.py). */
ov2_status ov2_klt_track_fb_dev(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels,
                                int max_iter, float eps, float err_th, float fb_th, int n, const float *d_kps_xy,
                                float *d_priors_xy, uint8_t *d_status, const int32_t *d_img_idx,
                                uint32_t *d_iters);

/* Replaces the two-stage batching of VisualFrontEnd::kltTracking (src/visual_front_end.cpp:132-275) with no
 * host round trip between the stages: keypoints with has_prior[i] are tracked on 2 levels from prior_xy,
 * their failures are re-queued on the full pyramid (starting from the failed forward result, :217-219), and
 * if fewer than 33 % of them succeed every re-queued prior is reset to the keypoint (:228-233) and
 * *d_p3p_req (per image) is set.  out_xy/out_status per keypoint.  Device pointers, asynchronous. */
ov2_status ov2_klt_tracking_frame_dev(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur, int win,
                                      int nlevels_full, int max_iter, float eps, float err_th, float fb_th, int n,
                                      const float *d_kps_xy, const float *d_prior_xy, const uint8_t *d_has_prior,
                                      const int32_t *d_img_idx, float *d_out_xy, uint8_t *d_out_status,
                                      int32_t *d_p3p_req /* [batch] */, uint32_t *d_iters /* 2n work words: [0,n) stage 1, [n,2n) stage 2; may be NULL */);

/* Tuning / test knob: lanes of a wavefront that share one keypoint in the tracking kernels.  0 (default) picks by call
 * size: 9 x 9 windows (nklt_win_size of every parameter file of the reference) take THREE lanes per keypoint at every
 * call size (20 keypoints per wave, Scharr derivatives formed in the kernel, no gradient planes read);
 * other windows 8 lanes from 65 536 keypoints on, 16 below.  All mappings give bit-identical results. */
ov2_status ov2_klt_set_lanes(ov2_ctx *ctx, int lanes);
/* Tuning / test knob of the two-stage tracking calls with the three-lane mapping (ov2_klt_tracking_frame_dev, ov2_stereo_
 * matching*): a level pass that has made `after` LK iterations and is down to <= `groups` (of 20) running keypoints hands
 * those stragglers to a second, small launch that continues each one's iteration sequence from its stored state -- the
 * other keypoints of the wave no longer wait for them; pickup = k > 0: every k-th wave of the first launch also takes one
 * batch of stragglers itself when it has finished (0: none).  Per-keypoint arithmetic is unchanged: results are bit-identical
 * with any setting.  OFF by default = (0, 0): measured slower than letting the wave wait (numbers in csrc/klt.hip at klt_rec);
 * OV2_KLT_YIELD=after,groups (and OV2_KLT_PICKUP=k) in the environment switch it on for a process. */
ov2_status ov2_klt_set_yield(ov2_ctx *ctx, int after, int groups, int pickup);

/* ---- stereo matching (keyframe rate) -------------------------------------------------------------- */
/* Replaces FeatureTracker::getLineMinSAD(iml, imr, pt, nwinsize, xprior, l1err, bgoleft) (include/feature_tracker.hpp:50,
 * src/feature_tracker.cpp:140-213) as MapManager::stereoMatching calls it for rectified rigs (src/map_manager.cpp:425-435):
 * iml / imr = level `level` of the two pyramids (the reference passes vleftpyr.at(2 * nklt_pyr_lvl)), pts_xy = the
 * keypoints scaled to that level (kp.px_ * 2^-level), nwinsize = 7, go_left = 1.  Per point: cv::getRectSubPix patches
 * (8U, 16-bit fixed-point bilinear, replicated border) of the left point and of every integer step along the row of the
 * right image, xprior = the step with the smallest mean absolute difference (first minimum; -1 if the window does not fit
 * or no step beats 255), l1err = that mean (may be NULL).  Points outside the level image give xprior = -1.
 * nwinsize odd, <= 15.  Host pointers; the _dev form takes device pointers (+ optional image index per point), is
 * asynchronous and synchronises nothing. */
ov2_status ov2_line_min_sad(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize, int go_left,
                            int n, const float *pts_xy, float *xprior, float *l1err);
ov2_status ov2_line_min_sad_dev(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize,
                                int go_left, int n, const float *d_pts_xy, const int32_t *d_img_idx, float *d_xprior,
                                float *d_l1err);

/* Lens model of a camera (CameraCalibration, src/camera_calibration.cpp): intrinsics + the distortion coefficients Dcv_.
 * model 0: none (undistortImagePoint is the identity, :317-319); 1: radial-tangential k1 k2 p1 p2 [k3] (`pinhole`:
 * cv::undistortPoints with P = K, five fixed-point sweeps / cv::projectPoints); 2: fisheye k1..k4 (cv::fisheye::
 * undistortPoints, Newton on theta / cv::fisheye::distortPoints).  OpenCV is not vendored by the reference: restated from
 * its published algorithms, the same restatement as the C++ host mirror's CameraCalibration (parity unpinned). */
typedef struct ov2_cam_model {
    double K[4];        /* fx fy cx cy */
    int32_t model;      /* 0 none, 1 radial-tangential, 2 fisheye */
    int32_t n_coeffs;   /* coefficients given in D (the others are 0) */
    double D[5];
} ov2_cam_model;

/* Replaces the tracking + gating part of MapManager::stereoMatching(frame, vleftpyr, vrightpyr) (include/map_manager.hpp:96,
 * src/map_manager.cpp:493-604): keypoints with has_prior[i] are tracked left -> right on 2 levels from prior_xy[i]
 * (:497-541; priors from the 3D point or the neighbours' depth), their failures are re-queued on the full pyramid with the
 * UPDATED prior (:533-537, no 33 % rule here), the others on the full pyramid from prior_xy[i] (:544-580: kp.px_ or the
 * SAD prior); then the epipolar gate (:583-604): rectified != 0: |lunpx.y - r.y| <= 2 and the right point is moved onto
 * the left row (:592); otherwise MultiViewGeometry::computeSampsonDistance(F_rl, lunpx, runpx) <= 2
 * (src/multi_view_geometry.cpp:798-821; F_rl row-major = Frame::Frl_, src/frame.cpp:62).  The gate works on UNDISTORTED
 * pixels as the reference does (:586): lunpx_xy = the left undistorted pixels (NULL = kps_xy), and the tracked right pixel
 * goes through right_cam's undistortImagePoint first (right_cam NULL or model 0: identity -- rectified / undistorted
 * streams; the EuRoC stereo files of the reference run with bdo_stereo_rect 0 and radial-tangential lenses, i.e. with a
 * model here).  out_rxy stays the RAW right pixel (Frame::updateKeypointStereo undistorts it itself).
 * out_status[i] = 1 where the reference reaches Frame::updateKeypointStereo(id, out_rxy[i]).
 * Host pointers; the _dev form takes device pointers (F_rl stays a host pointer), is asynchronous, d_iters as in
 * ov2_klt_tracking_frame_dev. */
ov2_status ov2_stereo_matching(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int win, int nlevels_full,
                               int max_iter, float eps, float err_th, float fb_th, int n, const float *kps_xy,
                               const float *prior_xy, const uint8_t *has_prior, const float *lunpx_xy, int rectified,
                               const double *F_rl, const ov2_cam_model *right_cam, float *out_rxy, uint8_t *out_status);
ov2_status ov2_stereo_matching_dev(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int win, int nlevels_full,
                                   int max_iter, float eps, float err_th, float fb_th, int n, const float *d_kps_xy,
                                   const float *d_prior_xy, const uint8_t *d_has_prior, const int32_t *d_img_idx,
                                   const float *d_lunpx_xy, int rectified, const double *F_rl, const ov2_cam_model *right_cam,
                                   float *d_out_rxy, uint8_t *d_out_status, uint32_t *d_iters);

/* ---- detectors (keyframe rate) -------------------------------------------------------------------- */
enum { OV2_DETECT_FAST = 0, OV2_DETECT_MINEIG = 1 };
/* Replaces FeatureExtractor::detectGridFAST (mode OV2_DETECT_FAST, `use_fast: 1` configs, src/feature_extractor.cpp:443-570)
 * and FeatureExtractor::detectSingleScale (mode OV2_DETECT_MINEIG, `use_singlescale_detector: 1`, :288-440), both
 * including their final cv::cornerSubPix(3x3, 30 it, 0.01) when do_subpix != 0; called from
 * MapManager::extractKeypoints (src/map_manager.cpp:312-319) on the CLAHE'd frame = level 0 of pyramid `b` of `pyr`.
 * cell = nmaxdist; cur_xy = the keypoints already in the frame (their cells are skipped and a disc of radius cell/4 is
 * masked around each); roi = {x, y, w, h} (NULL = whole image; detectGridFAST ignores it, as in the reference).
 * *thresh is the detector state the reference adapts from call to call: dmaxquality_ (MINEIG) or nfast_th_ (FAST,
 * integer valued), updated in place.  out_xy must hold 2 * (w/cell)*(h/cell) points; *n_out receives the count.
 * The cells of the reference's racy parallel_for_ are visited in the 2x2-colouring order. */
ov2_status ov2_detect_grid(ov2_ctx *ctx, const ov2_pyr *pyr, int b, int cell, int mode, double *thresh, int n_cur,
                           const float *cur_xy, const int *roi, int do_subpix, int *n_out, float *out_xy);
/* every image of the pyramid batch in one call (one synchronisation for all of them): thresh[B] in/out, n_cur[B],
 * cur_xy = the keypoints of image 0, then image 1, ... ; out_xy[B][out_cap][2], n_out[B]; out_cap >= 2 * cells. */
ov2_status ov2_detect_grid_batch(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, int mode, double *thresh,
                                 const int *n_cur, const float *cur_xy, const int *roi, int do_subpix, int *n_out,
                                 float *out_xy, int out_cap);
/* Device-resident, fully asynchronous form of the same call (no host synchronisation: a multi-sequence front-end can
 * enqueue the next frames while keyframe detection runs).  All d_* pointers are DEVICE pointers:
 *   d_thresh   B doubles, in/out (dmaxquality_ / nfast_th_ per image, adapted on the device)
 *   n_cur keypoints of all images: d_cur_xy (n_cur x 2 floats), d_cur_img (image index per keypoint),
 *   d_cur_valid (optional, n_cur bytes: only keypoints with a non-zero byte count, e.g. the tracking status)
 *   d_n_out    B ints, d_out_xy  B x out_cap x 2 floats (out_cap >= 2 * cells)
 * roi is a host pointer (4 ints or NULL). */
ov2_status ov2_detect_grid_batch_dev(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, int mode, double *d_thresh, int n_cur,
                                     const float *d_cur_xy, const int32_t *d_cur_img, const uint8_t *d_cur_valid,
                                     const int *roi, int do_subpix, int32_t *d_n_out, float *d_out_xy, int out_cap);

/* ---- local bundle adjustment ------------------------------------------------------------------- */
/* Flat, POD restatement of the ceres::Problem that Optimizer::localBA assembles (src/optimizer.cpp:76-392).
 * The host adapter (ov2slam_amd/host/local_ba_adapter.*) fills it from Frame/MapPoint graphs and applies the
 * result back with the semantics of src/optimizer.cpp:741-882.
 * Parameter memory layout follows the reference PODs (include/ceres_parametrization/.../se3_param_block.hpp):
 *   pose = Twc as [tx ty tz qx qy qz qw]  (Eigen quaternion coefficient order x,y,z,w)
 *   landmark = world XYZ (inv_depth = 0) or inverse depth in its anchor keyframe (inv_depth = 1)
 * Residual types = the cost functors of src/ceres_parametrization.cpp: */
enum {
    OV2_BA_L_XYZ = 0,      /* ReprojectionErrorKSE3XYZ               :107  {K_l, T_k, X}            */
    OV2_BA_R_XYZ = 1,      /* ReprojectionErrorRightCamKSE3XYZ       :198  {K_r, T_k, T_rl, X}      */
    OV2_BA_L_INV = 2,      /* ReprojectionErrorKSE3AnchInvDepth      :361  {K_l, T_anch, T_k, rho}  */
    OV2_BA_R_INV = 3,      /* ReprojectionErrorRightCamKSE3AnchInvDepth :579 {K_l,K_r,T_anch,T_k,T_rl,rho} */
    OV2_BA_RANCH_INV = 4   /* ReprojectionErrorRightAnchCamKSE3AnchInvDepth :476 {K_l,K_r,T_rl,rho} */
};

typedef struct ov2_ba_problem {
    double calib_l[4], calib_r[4];   /* fx fy cx cy (constant blocks, src/optimizer.cpp:94-113) */
    double T_rl[7];                  /* right<-left extrinsic, constant (:116-125) */
    int32_t inv_depth;               /* buse_inv_depth */
    int32_t n_pose;
    double *pose;                    /* n_pose x 7, in/out */
    const uint8_t *pose_const;       /* n_pose: 1 = SetParameterBlockConstant (:181-185, :229-246, :397-407) */
    int32_t n_lm;
    double *lm;                      /* n_lm x (inv_depth ? 1 : 3), in/out */
    const int32_t *lm_anchor_pose;   /* n_lm, inv_depth only: pose index of the anchor keyframe (:258-267) */
    const double *lm_anchor_uv;      /* n_lm x 2, inv_depth only: undistorted anchor pixel */
    int32_t n_res;
    const uint8_t *res_type;         /* n_res: OV2_BA_* */
    const int32_t *res_pose;         /* n_res: observing keyframe (ignored for RANCH_INV) */
    const int32_t *res_lm;           /* n_res */
    const double *res_uv;            /* n_res x 2: unpx_ (left types) or runpx_ (right types) */
    const double *res_sigma;         /* n_res: sqrt_info = I / sigma, sigma = 2^scale (NULL = all 1) */
} ov2_ba_problem;

typedef struct ov2_ba_options {
    double huber_delta;      /* a of ceres::HuberLoss = sqrtf(robust_mono_th) (:49); <= 0 disables the loss */
    double chi2_th;          /* robust_mono_th: residual flagged when chi2 > chi2_th or depth <= 0 (:500-592) */
    int32_t max_iters;       /* 5  (:462) */
    int32_t l2_refine;       /* apply_l2_after_robust: re-solve without the flagged residuals (:603-627) */
    int32_t l2_max_iters;    /* 10 (:610) */
    double function_tolerance;     /* 1e-3 (:463) */
    /* ceres::Solver::Options defaults the reference leaves untouched (include/ceres/solver.h) */
    double initial_radius;         /* 1e4  */
    double max_radius;             /* 1e16 */
    double min_radius;             /* 1e-32 */
    double min_lm_diagonal;        /* 1e-6 */
    double max_lm_diagonal;        /* 1e32 */
    double min_relative_decrease;  /* 1e-3 */
    double parameter_tolerance;    /* 1e-8 */
    double gradient_tolerance;     /* 1e-10 */
    int32_t jacobi_scaling;        /* 1 */
    int32_t max_consecutive_invalid_steps; /* 5 */
} ov2_ba_options;

typedef struct ov2_ba_iter {
    double cost, cost_change, radius, relative_decrease, model_cost_change;
    int32_t step_is_valid, step_is_successful;
} ov2_ba_iter;

enum { OV2_BA_TERM_MAX_ITER = 0, OV2_BA_TERM_FTOL = 1, OV2_BA_TERM_PTOL = 2, OV2_BA_TERM_GTOL = 3,
       OV2_BA_TERM_MIN_RADIUS = 4, OV2_BA_TERM_FAILURE = 5, OV2_BA_TERM_SKIPPED = 6 };

#define OV2_BA_MAX_LOG 40
typedef struct ov2_ba_result {
    double *chi2;            /* n_res (caller allocated, may be NULL): ||r||^2 at the state the flag was taken */
    uint8_t *depth_positive; /* n_res (may be NULL) */
    uint8_t *outlier;        /* n_res (may be NULL): 0 kept, 1 flagged after the robust solve, 2 after the L2 re-solve */
    double initial_cost, final_cost;   /* of the robust solve */
    double l2_initial_cost, l2_final_cost;
    int32_t n_log;           /* iteration log of both solves (robust first), iteration 0 included */
    int32_t n_log_robust;
    int32_t termination, l2_termination, l2_done;
    int32_t n_outliers_pass1, n_outliers_pass2;
    ov2_ba_iter log[OV2_BA_MAX_LOG];
} ov2_ba_result;

/* fills `o` with the values Optimizer::localBA uses (robust_mono_th from the YAML, default 5.9915) */
void ov2_ba_default_options(ov2_ba_options *o, float robust_mono_th);

/* Replaces the ceres::Solve + chi2 flagging + L2 re-solve of Optimizer::localBA(Frame&, bool)
 * (include/optimizer.hpp:42, src/optimizer.cpp:439-735): Levenberg-Marquardt with Jacobi scaling, Huber loss via the
 * Ceres corrector, landmark Schur complement, reduced camera system solved on device; time caps are not applied
 * (the reference's 0.2 s wall-clock truncation makes it non-deterministic).
 * p->pose / p->lm are HOST pointers, updated in place for the non-constant blocks.
 * r->chi2 / depth_positive / outlier follow the reference's post-solve reads of the cost functors' cached fields
 * (src/optimizer.cpp:500-592): they are the values of the LAST evaluation Ceres made -- the final state after an accepted
 * last step, the rejected / tolerance-terminating candidate otherwise (trust_region_minimizer.cc:108-131). */
ov2_status ov2_ba_solve(ov2_ctx *ctx, const ov2_ba_problem *p, const ov2_ba_options *o, ov2_ba_result *r);

/* B independent local-BA windows in one call: p[B], r[B], one set of options (ov2_ba_solve is the B = 1 case and runs the
 * same code).  This is how one GPU serves many SLAM instances: the reference runs one Estimator thread per instance
 * (src/estimator.cpp:32-98), each calling Optimizer::localBA on its own keyframe window; here the windows that are pending
 * at the same time are laid end to end, every O(residuals) / O(landmarks) kernel covers all of them in one launch, every
 * window gets its own reduced camera system, its own Cholesky workgroup and its own device-side copy of Ceres'
 * trust-region state machine (radius, accept / reject, tolerances), and the fixed chain of max_iters LM rounds is enqueued
 * without a host synchronisation.  A window's result does not depend on the batch it is solved in (bitwise).
 * All windows of a batch share one landmark parametrisation (inv_depth); calibrations / extrinsics may differ. */
ov2_status ov2_ba_solve_batch(ov2_ctx *ctx, int B, const ov2_ba_problem *p, const ov2_ba_options *o, ov2_ba_result *r);

/* The same solve for DEVICE-RESIDENT windows: every array pointer inside p[w] (pose, pose_const, lm, lm_anchor_pose,
 * lm_anchor_uv, res_type, res_pose, res_lm, res_uv, res_sigma) and the per-residual outputs of r[w] (chi2, depth_positive,
 * outlier; NULL = not wanted) are device pointers on ctx's device (ov2_dev_alloc, or arrays a device-side producer such as
 * the map mirror left there); p[] / r[] themselves and the scalar fields / iteration logs of r[] are host memory.  One kernel
 * lays the windows end to end, one hands the solved states and the outputs back; what crosses PCIe is a table of B
 * pointers in and the window records out (a batch of 64 50-keyframe windows is 205 MB of arrays in the host form: ~5 ms
 * of link time plus the host staging).  Replaces the same reference code as ov2_ba_solve_batch
 * (src/optimizer.cpp:439-735 per window, src/estimator.cpp:32-98 per instance); results are bitwise those of the host form.
 * The arrays must not be written by other streams during the call; it returns after one synchronisation. */
ov2_status ov2_ba_solve_batch_dev(ov2_ctx *ctx, int B, const ov2_ba_problem *p, const ov2_ba_options *o, ov2_ba_result *r);

/* ---------------------------------------------------------------------------------------------------
 * Motion-only BA (pose refinement on fixed 3D points).
 * Replaces MultiViewGeometry::ceresPnP(vunkps, vwpts, vscales, Twc, nmaxiter, chi2th, buse_robust,
 * bapply_l2_after_robust, fx, fy, cx, cy, voutliersidx)  include/multi_view_geometry.hpp:104,
 * src/multi_view_geometry.cpp:492-586, called per frame by VisualFrontEnd::computePose
 * (src/visual_front_end.cpp:791) and by the relocalisation / loop-closure checks.
 * B independent frames per call (B = 1 reproduces the reference call), frame b owns n_pts[b] consecutive
 * entries of the point arrays.  All pointers are HOST pointers.
 *   unpx   sum(n) x 2   undistorted pixel observations        wpts   sum(n) x 3  world points
 *   scales sum(n)       pyramid octave of each keypoint (sigma = 2^scale), NULL = all 0
 *   K      B x 4        fx fy cx cy                            Twc    B x 7  [t, qx qy qz qw], updated in place
 *   outlier sum(n)      1 where chi2 > chi2th or depth <= 0 after the robust solve (voutliersidx as a mask)
 *   success B           the reference's bool return: 0 if the solver failed or every point was flagged
 *                       (in the second case Twc[b] is left untouched, :573-575)
 *   iters  B x 2        (may be NULL) LM iterations of the robust solve and of the L2 re-solve
 * The reference's 5 ms wall-clock cap (:533-535) is not reproduced (non-deterministic), nmaxiter is. */
ov2_status ov2_pnp_solve_batch(ov2_ctx *ctx, int B, const int *n_pts, const double *unpx, const double *wpts,
                               const int *scales, const double *K, double *Twc, int max_iters, float chi2th,
                               int use_robust, int l2_after_robust, uint8_t *outlier, int *success, int *iters);
/* device-resident, asynchronous form: d_off = B + 1 prefix offsets of the frames' points, d_removed = scratch of
 * sum(n) bytes; every other array as above but in HBM.  Nothing is synchronised. */
ov2_status ov2_pnp_solve_batch_dev(ov2_ctx *ctx, int B, const int32_t *d_off, const double *d_unpx, const double *d_wpts,
                                   const int32_t *d_scales, const double *d_K, double *d_Twc, int max_iters, float chi2th,
                                   int use_robust, int l2_after_robust, uint8_t *d_outlier, uint8_t *d_removed,
                                   int32_t *d_success, int32_t *d_iters);

/* ---------------------------------------------------------------------------------------------------
 * Pose graphs: Optimizer::localPoseGraph (src/optimizer.cpp:2346-2592, the loop closer's chain of
 * keyframes loop .. new + the loop edge) and Optimizer::fullPoseGraph (:2783-2870, the chain of all frames between
 * constant keyframes at the end of a run) = LeftSE3RelativePoseError (src/ceres_parametrization.cpp:30-102,
 * se3left_parametrization.hpp:76-99) on SE3LeftParameterization blocks, LEVENBERG_MARQUARDT, no loss function.
 *   residual of edge (i, j):  log( Twc_j^-1 * Twc_i * T_ij ),  T_ij = the measured pose of camera j in camera i
 *   (parameters[0] = pose i, parameters[1] = pose j; sigma = 1), jacobians as the reference writes them
 *   ((I + J_c / 2) Adj, "adapted from Strasdat").
 * Structure: both graphs are CHAINS -- every edge joins two free poses that are neighbours in the order of the free
 * poses, or has a constant end -- so the normal equations are block tridiagonal per run of free poses; other graphs
 * return OV2_ERR_UNSUPPORTED.  The whole minimisation is one launch (one workgroup; runs of free poses in parallel).
 * Options: the trust-region fields of ov2_ba_options (max_iters, function_tolerance, radii, diagonal clamps, tolerances,
 * jacobi_scaling); localPoseGraph: 10 iterations, 1e-4; fullPoseGraph: 100, 1e-6. */
typedef struct ov2_pg_problem {
    int32_t n_pose;
    double *pose;                 /* n_pose x 7 Twc (tx ty tz qx qy qz qw), in / out */
    const uint8_t *pose_const;    /* n_pose */
    int32_t n_edge;
    const int32_t *edge_i, *edge_j;
    const double *T_ij;           /* n_edge x 7 */
} ov2_pg_problem;

typedef struct ov2_pg_result {
    double initial_cost, final_cost;
    int32_t termination;          /* OV2_BA_TERM_* */
    int32_t n_log;
    ov2_ba_iter log[OV2_BA_MAX_LOG];
} ov2_pg_result;

ov2_status ov2_pose_graph_solve(ov2_ctx *ctx, const ov2_pg_problem *p, const ov2_ba_options *o, ov2_pg_result *r);

/* ---------------------------------------------------------------------------------------------------
 * Flat device-resident map mirror: keyframe poses, landmark states and the observation table
 * (keyframe, landmark, unpx, runpx, scale, stereo flag) as SoA arrays in HBM, kept in step with the reference's
 * MapManager by the hooks below, so that the set-up stage of Optimizer::localBA (src/optimizer.cpp:43-430: the walk
 * over Frame::map_covkfs_, Frame::mapkps_, MapPoint::set_kfids_ and the two hash maps of the MapManager) becomes a
 * handful of linear scans over the observation table.  kfid / lmid index the tables directly (both are small dense
 * counters in the reference: src/map_manager.cpp:621-690); the capacities given at creation are initial sizes, the
 * tables grow (x1.5, device copies) when an id or the observation count passes them.
 *
 *   hook                          reference mutation it mirrors
 *   ov2_map_add_keyframe          MapManager::addKeyframe + addMapPointKfObs   src/map_manager.cpp:621-634,769-799
 *   ov2_map_set_landmarks         MapManager::addMapPoint / updateMapPoint / setMapPointObs  :636-689,715-767,1053
 *   ov2_map_set_poses             Frame::setTwc after localBA / pose-graph      src/optimizer.cpp:767-786
 *   ov2_map_remove_obs            MapManager::removeMapPointObs                 :970-1019
 *   ov2_map_set_obs_stereo        Frame::removeStereoKeypointById / stereoMatching results
 *   ov2_map_remove_landmarks      MapManager::removeMapPoint                    :922-968
 *   ov2_map_remove_keyframe       MapManager::removeKeyframe                    :885-920
 * All array arguments are HOST pointers, staged through the ctx's pinned block; a hook returns once its copy has
 * been consumed (one stream synchronisation). */
typedef struct ov2_map ov2_map;

#define OV2_LM_ALIVE 1   /* MapManager::map_plms_ holds it */
#define OV2_LM_3D    2   /* MapPoint::is3d_ */
#define OV2_LM_OBS   4   /* MapPoint::isobs_ (seen by the current frame) */
#define OV2_LM_KP3D  8   /* its keypoints carry Keypoint::is3d_ (Frame::turnKeypoint3d ran): MapPoint::isBad() clears
                            is3d_ but leaves the keypoints 3D, and the set-up selects by the keypoint flag (:176-180) */

ov2_status ov2_map_create(ov2_ctx *ctx, int max_kf, int max_lm, int max_obs, ov2_map **out);
void ov2_map_destroy(ov2_map *m);
/* one keyframe with its n keypoints: lmid, unpx (n x 2), runpx (n x 2, read where is_stereo), is_stereo, scale */
ov2_status ov2_map_add_keyframe(ov2_map *m, int kfid, const double *Twc, int n, const int32_t *lmid, const double *unpx,
                                const double *runpx, const uint8_t *is_stereo, const int32_t *scale);
/* state = OR of OV2_LM_*; xyz (n x 3) may be NULL to change the states only */
ov2_status ov2_map_set_landmarks(ov2_map *m, int n, const int32_t *lmid, const double *xyz, const uint8_t *state);
ov2_status ov2_map_set_poses(ov2_map *m, int n, const int32_t *kfid, const double *Twc);
ov2_status ov2_map_remove_obs(ov2_map *m, int n, const int32_t *kfid, const int32_t *lmid);
ov2_status ov2_map_set_obs_stereo(ov2_map *m, int n, const int32_t *kfid, const int32_t *lmid, const uint8_t *is_stereo,
                                  const double *runpx);
ov2_status ov2_map_remove_landmarks(ov2_map *m, int n, const int32_t *lmid);
ov2_status ov2_map_remove_keyframe(ov2_map *m, int kfid);
/* Removals leave dead rows in the observation table (the reference erases the entries from its hash maps,
 * src/map_manager.cpp:885-1019).  ov2_map_local_ba_setup squeezes them out (stable: the order of the live rows, and so
 * every result, is unchanged) when fewer than half of >= 4096 rows are live, so the table stays within 2x the live
 * observations over any sequence length; ov2_map_compact does it on request.  kfid / lmid are never reused by the
 * reference (nkfid_ / nlmid_ only grow), which is what makes a row of a removed keyframe or landmark dead for good. */
ov2_status ov2_map_compact(ov2_map *m, int *rows_before, int *rows_after);
ov2_status ov2_map_obs_rows(const ov2_map *m, int *rows, int *capacity, int *compactions);

/* The flat problem of one local BA, in pinned host memory owned by the map (valid until the next set-up call):
 * exactly the arrays ov2_ba_problem wants plus the reference ids behind the indices. */
typedef struct ov2_local_ba_setup {
    int32_t aborted;                 /* nb3dkps < nmin_covscore  (src/optimizer.cpp:61-63) */
    int32_t n_pose, n_lm, n_res, n_bad;
    const int32_t *pose_kfid;        /* n_pose, ascending kfid */
    const uint8_t *pose_const;       /* n_pose */
    double *pose;                    /* n_pose x 7 (Twc) */
    const int32_t *lm_lmid;          /* n_lm, ascending lmid */
    double *lm;                      /* n_lm x (inv_depth ? 1 : 3) */
    const int32_t *lm_anchor_pose;   /* n_lm (inv_depth) */
    const double *lm_anchor_uv;      /* n_lm x 2 (inv_depth) */
    const uint8_t *res_type;         /* n_res */
    const int32_t *res_pose, *res_lm;
    const double *res_uv, *res_sigma;
    const int32_t *bad_lmid;         /* n_bad: MapPoint::isBad() landmarks met on the way (set_badlmids, :204-207) */
    uint8_t *res_outlier;            /* device views only: n_res bytes of the map's block, zeroed by the set-up, for
                                        ov2_ba_result.outlier of the solve (the update stage reads the flags there); NULL in the host form */
} ov2_local_ba_setup;

/* Replaces the set-up stage of Optimizer::localBA (src/optimizer.cpp:43-430) for the keyframe newkf:
 *  - covisibility scores of newkf (Frame::map_covkfs_, src/map_manager.cpp:117-193) recounted from the table,
 *  - keyframes newest -> oldest: optimised while score >= nmin_covscore and kfid > 0, constant from the first that
 *    fails (:150-190); their 3D keypoints' landmarks are the local landmarks (isBad() ones are listed, not used),
 *  - every alive observation of a local landmark by a keyframe <= the newest covisible one becomes one or two
 *    residual blocks (:193-392); observers outside the window enter as constant poses (:229-246); with inv_depth
 *    the first observer is the anchor (:251-287),
 *  - at least nmin_cst_kfs constant keyframes, smallest kfids first (:394-407).
 * One synchronisation for the sizes, one for the arrays. */
ov2_status ov2_map_local_ba_setup(ov2_map *m, int newkf, int nmin_covscore, int nmin_cst_kfs, int inv_depth,
                                  const double *calib_l /* fx fy cx cy: anchor depth needs no intrinsics; reserved */,
                                  ov2_local_ba_setup *out);

/* The set-up stage for B maps of one context at once (one SLAM instance each: the Estimator threads of B sequences that
 * have a keyframe pending, src/estimator.cpp:32-98).  Same result per map as ov2_map_local_ba_setup, but
 *  - every kernel of the chain serves all B maps (map = blockIdx.y), the sizes a kernel needs come from device memory,
 *    and NOTHING is synchronised until the B headers (16 ints per map) have been gathered: one synchronisation per call,
 *  - the flat problems stay ON THE DEVICE: dev[b] holds DEVICE pointers (what ov2_ba_solve_batch_dev takes), valid until
 *    the map's next set-up; dev[b].res_outlier is a zeroed n_res-byte array for the solve's outlier flags,
 *  - a mostly dead observation table is squeezed BEFORE the chain (from the live count of the map's previous set-up), never
 *    after it: the update stage needs this set-up's row indices.
 * calib_l: B x 4 left intrinsics (fx fy cx cy) -- needed by the update stage of the inverse-depth form -- or NULL.
 * newkf[b] = the new keyframe of map b.  A map may appear only once per call. */
ov2_status ov2_map_local_ba_setup_batch(ov2_ctx *ctx, int B, ov2_map *const *maps, const int32_t *newkf, int nmin_covscore,
                                        int nmin_cst_kfs, int inv_depth, const double *calib_l, ov2_local_ba_setup *dev);

/* Replaces the update stage of Optimizer::localBA (src/optimizer.cpp:741-882) on the tables of B maps, from the flat
 * problems of their last set-up (whose pose / lm arrays now hold the solved states) and the solve's per-block outlier
 * flags d_outlier[b] (n_res device bytes, e.g. the set-up's res_outlier; NULL = nothing flagged):
 *  - a flagged right-camera block demotes the observation to mono (Frame::removeStereoKeypointById, :743-751), a flagged
 *    left block removes it (MapManager::removeMapPointObs(lmid, kfid), :753-764; for an observation of keyframe
 *    cur_kfid[b] -- the current frame's, -1 = none -- MapPoint::isobs_ is cleared too, removeObsFromCurFrameById),
 *  - the solved poses of the non-constant keyframes are stored (:767-786),
 *  - per local landmark: MapPoint::isBad() / fewer than 3 observers, older than newkf - 3 and not in the current frame /
 *    non-positive anchor depth -> MapManager::removeMapPoint; otherwise its new world point (inverse depth: Twc_anchor *
 *    (K^-1 [u v 1] / rho) with the UPDATED anchor pose) via MapManager::updateMapPoint (:789-853),
 *  - the second culling pass over set_badlmids (:856-882).
 * MapPoint::kfid_ is taken to be the landmark's oldest observer (how MapManager::addMapPoint creates it and
 * MapPoint::removeKfObs maintains it; mergeMapPoints is out of scope).
 * out == NULL: fully asynchronous.  out != NULL: one synchronisation; out[b] lists what the host must replay on its own
 * Frame / MapPoint objects (DEVICE arrays inside the map's block, valid until its next set-up; order arbitrary).
 */
typedef struct ov2_local_ba_update {
    int32_t n_removed_lm, n_removed_obs, n_stereo_off;
    const int32_t *removed_lmid;     /* MapManager::removeMapPoint(lmid) */
    const int32_t *removed_obs;      /* pairs (kfid, lmid): MapManager::removeMapPointObs(lmid, kfid) */
    const int32_t *stereo_off;       /* pairs (kfid, lmid): Frame::removeStereoKeypointById(lmid) on keyframe kfid */
} ov2_local_ba_update;
ov2_status ov2_map_local_ba_update_batch(ov2_ctx *ctx, int B, ov2_map *const *maps, const uint8_t *const *d_outlier,
                                         const int32_t *cur_kfid, ov2_local_ba_update *out);

/* Test / bench support.  ov2_map_save_state keeps a device copy of the mutable state of the tables (poses, landmark points
 * and states, observation flags); ov2_map_restore_state_batch rewinds B maps to it in one launch, asynchronously (every
 * bench job starts from the same noisy map, as a new keyframe of a live sequence would bring it).  ov2_map_download copies
 * the tables to host arrays sized by the returned capacities (any pointer may be NULL; call once with NULLs for sizes). */
ov2_status ov2_map_save_state(ov2_map *m);
ov2_status ov2_map_restore_state_batch(ov2_ctx *ctx, int B, ov2_map *const *maps);
ov2_status ov2_map_download(ov2_map *m, int *n_kf, int *n_lm, int *n_obs, double *kf_pose, uint8_t *kf_state, double *lm_xyz,
                            uint8_t *lm_state, int32_t *obs_kf, int32_t *obs_lm, uint8_t *obs_flag, double *obs_uv, double *obs_ruv);

/* The flat problem of the last set-up where the kernels left it ON THE DEVICE: `dev` = `host` with every array pointer
 * translated (same validity: until the next set-up call of this map).  These are the pointers ov2_ba_solve_batch_dev
 * takes, so set-up -> solve runs without the measurement arrays crossing PCIe in either direction; the solved pose / lm
 * arrays are then the device copies (ov2_memcpy_d2h what the host-side update stage, src/optimizer.cpp:741-882, needs). */
ov2_status ov2_map_setup_device_view(const ov2_map *m, const ov2_local_ba_setup *host, ov2_local_ba_setup *dev);

/* ---------------------------------------------------------------------------------------------------
 * Two-view triangulation of keypoint pairs + the mapper's acceptance gates (second half).
 * Replaces the per-keypoint bodies of Mapper::triangulateStereo (src/mapper.cpp:346-461) and
 * Mapper::triangulateTemporal (:191-344): MultiViewGeometry::triangulate(Tlr, bvl, bvr)
 * (include/multi_view_geometry.hpp:46, src/multi_view_geometry.cpp:53-61 -> opengvTriangulate2 :85-99, the mid-point
 * method) or the rectified disparity form (:411-422), the depth gate (z < 0.1 in either view, :428 / :316), the
 * reprojection gate against max_reproj_err in both views (:433-446 / :322-336), Frame::projCamToWorld (:449 / :339) and,
 * for the temporal case, the rotation-compensated parallax (:301-302).
 *   view a = left camera / older keyframe, view b = right camera / new keyframe
 *   T_ab    G x 7   pose of b in a [t, qx qy qz qw]: Tlr (stereo) or Tcicj (temporal, one per source keyframe)
 *   Twc_a   G x 7   (may be NULL) world pose of view a -> wpt
 *   grp     n       (may be NULL = all 0) which of the G pose pairs a keypoint pair uses
 *   bv_a, bv_b  n x 3 bearing vectors (Keypoint::bv_ / rbv_)      unpx_a, unpx_b  n x 2 undistorted pixels (float)
 *   K_a, K_b    fx fy cx cy of the two views
 *   pt_a    n x 3   the point in view a (left_pt)                  wpt  n x 3 (may be NULL)
 *   parallax n      (may be NULL) |unpx_a - proj_b(R_ab bv_b)|
 *   status  n       OV2_TRI_OK / _BEHIND / _REPROJ / _NEG_DISP (the reference removes the observation / skips the point)
 * Host pointers; ov2_triangulate_pairs_dev takes the same arrays in HBM and synchronises nothing. */
#define OV2_TRI_MIDPOINT  0
#define OV2_TRI_RECTIFIED 1
#define OV2_TRI_OK        0
#define OV2_TRI_BEHIND    1
#define OV2_TRI_REPROJ    2
#define OV2_TRI_NEG_DISP  3
ov2_status ov2_triangulate_pairs(ov2_ctx *ctx, int n, int method, int G, const double *T_ab, const double *Twc_a,
                                 const int32_t *grp, const double *bv_a, const double *bv_b, const float *unpx_a,
                                 const float *unpx_b, const double *K_a, const double *K_b, float max_reproj_err,
                                 double *pt_a, double *wpt, double *parallax, uint8_t *status);
ov2_status ov2_triangulate_pairs_dev(ov2_ctx *ctx, int n, int method, int G, const double *d_T_ab, const double *d_Twc_a,
                                     const int32_t *d_grp, const double *d_bv_a, const double *d_bv_b,
                                     const float *d_unpx_a, const float *d_unpx_b, const double *K_a, const double *K_b,
                                     float max_reproj_err, double *d_pt_a, double *d_wpt, double *d_parallax,
                                     uint8_t *d_status);

/* ---------------------------------------------------------------------------------------------------
 * Keyframe descriptors and map matching (first half).
 *
 * ov2_describe_brief replaces FeatureExtractor::describeBRIEF(im, vpts) (include/feature_extractor.hpp:48,
 * src/feature_extractor.cpp:224-285) = cv::xfeatures2d::BriefDescriptorExtractor::create()->compute (32 bytes, patch 48,
 * 9 x 9 box sums at the ROUNDED keypoint, bit k of the descriptor = box(y1, x1) < box(y2, x2) of test k, first test in the
 * most significant bit of byte 0; keypoints closer than 28 px to the border get no descriptor: valid[i] = 0, the reference
 * returns an empty cv::Mat for them).  im = level 0 of pyramid `b` of `pyr` (the CLAHE'd frame).  The 256 test pairs of
 * opencv_contrib (generated_32.i) are not in the reference tree: `pattern` (256 x {y1, x1, y2, x2}, int8, |v| <= 24) is
 * supplied by the caller.  Host pointers; the _dev form takes device pointers (pattern too) + an image index per point. */
ov2_status ov2_describe_brief(ov2_ctx *ctx, const ov2_pyr *pyr, int b, int n, const float *pts_xy, const int8_t *pattern,
                              uint8_t *desc /* n x 32 */, uint8_t *valid /* n */);
ov2_status ov2_describe_brief_dev(ov2_ctx *ctx, const ov2_pyr *pyr, int n, const float *d_pts_xy, const int32_t *d_img_idx,
                                  const int8_t *d_pattern, uint8_t *d_desc, uint8_t *d_valid);

/* Flat inputs of Mapper::matchToMap(frame, fmaxprojerr, fdistratio, set_local_lmids) (include/mapper.hpp:66,
 * src/mapper.cpp:576-774).  Keypoints = the frame's keypoints that carry a map point (kp.lmid_ >= 0); candidates = the
 * local map points to be matched, in the order the caller iterates set_local_lmids, WITHOUT those the frame already
 * observes (Frame::isObservingKp, :613) or that are not 3-D (:622).  Descriptor sets = MapPoint::map_kf_desc_ (32 bytes
 * each; an empty set = desc_.empty()).  Keyframe lists = MapPoint::set_kfids_, ascending; kp_kf_px = the pixel of the
 * keypoint's map point in each of those keyframes (Frame::getKeypointById(lmid).px_); kf_Twc is indexed by kfid.
 * grid = Frame::vgridkps_ restricted to the listed keypoints, cells row-major (nbwcells = ceil(img_w / cell)), ids in
 * their vector order (they decide ties).  cam: the lens model of Frame::projWorldToImageDist (:631, :706; NULL or model 0 =
 * the pinhole projection with K). */
typedef struct ov2_match_input {
    double Twc[7], K[4];
    int32_t img_w, img_h, cell, nb3dkps;
    int32_t n_kp;
    const float *kp_px;            /* n_kp x 2 */
    const int32_t *kp_desc_ptr;    /* n_kp + 1 */
    const uint8_t *kp_descs;
    const int32_t *kp_kf_ptr;      /* n_kp + 1 */
    const int32_t *kp_kfids;
    const float *kp_kf_px;         /* per kp_kfids entry x 2 */
    const int32_t *grid_ptr;       /* cells + 1 */
    const int32_t *grid_kp;        /* keypoint indices */
    int32_t n_cand;
    const double *cand_wpt;        /* n_cand x 3 */
    const int32_t *cand_desc_ptr;  /* n_cand + 1 */
    const uint8_t *cand_descs;
    const int32_t *cand_kf_ptr;    /* n_cand + 1 */
    const int32_t *cand_kfids;
    int32_t n_kf;
    const double *kf_Twc;          /* n_kf x 7 */
    const ov2_cam_model *cam;      /* lens model of the projections (host pointer), NULL = pinhole K */
} ov2_match_input;

/* match_cand[k] = index of the candidate matched to keypoint k or -1 (the reference's map_previd_newid: keypoint lmid ->
 * map point id), match_dist[k] = its descriptor distance.  Per candidate: projection + field-of-view gates (:626-645), the
 * keypoints of the 2 x 2 grid cells around the projection (Frame::getSurroundingKeypoints, src/frame.cpp:624-650) within
 * dmaxpxdist, never co-observed (:686-695), mean reprojection distance in the keypoint's keyframes <= dmaxpxdist
 * (:700-719), MapPoint::computeMinDescDist (Hamming), best / second best with the 0.9 ratio (:723-740); per keypoint the
 * candidate with the smallest distance, the later one on ties (:754-771).  Host pointers, synchronous. */
ov2_status ov2_match_to_map(ov2_ctx *ctx, const ov2_match_input *in, float fmaxprojerr, float fdistratio,
                            int32_t *match_cand /* n_kp */, float *match_dist /* n_kp */);

/* ---- diagnostics -----------------------------------------------------------------------------------
 * Reads a device buffer of nrows x stride_bytes exactly once with the access pattern of the KLT window staging (each lane of
 * a wave loads 16 bytes from a different row): a known byte count to calibrate rocprofv3's FETCH_SIZE on that pattern
 * (scripts/fetch_calib.sh).  d_out: nrows words.  Asynchronous. */
ov2_status ov2_dbg_rowload16(ov2_ctx *ctx, const void *d_buf, size_t stride_bytes, size_t nrows, uint32_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* OV2SLAM_HIP_H */
