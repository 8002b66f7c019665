/*
 * ov2_oracle.h -- CPU restatement ("oracle") of the OV2SLAM front-end + localBA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path is the HIP library
 * (ov2slam_amd/csrc -> libov2hip.so) and it never links or calls any of this.
 *
 * PARITY STATUS
 *   - pyramid / Scharr / CLAHE / pyramidal LK / forward-backward wrapper: the arithmetic lives in
 *     OpenCV (modules/video/src/lkpyramid.cpp, modules/imgproc/src/{pyramids,clahe}.cpp), which is
 *     NOT vendored in /root/reference and whose version is unpinned (CMakeLists.txt:76-80).  The
 *     reference holds no test or fixture for it  =>  **parity unpinned** for the KLT arithmetic;
 *     the restatement follows OpenCV's published algorithm and the
 *     reference's own call sites (src/feature_tracker.cpp:35-137, src/visual_front_end.cpp:1143-1177).
 *   - localBA linear algebra: pinned by the known-answer vectors the vendored Ceres 2.0.0 tests hold
 *     as text (Thirdparty/ceres-solver/internal/ceres/linear_least_squares_problems.cc:66-180, ...),
 *     see tests/golden/.  The reference's own localBA outputs: **parity unpinned** (no fixtures,
 *     reference not buildable here: needs OpenCV, Eigen, ROS, PCL, SuiteSparse).
 */
#ifndef OV2_ORACLE_H
#define OV2_ORACLE_H

#include <stdint.h>

#include "../include/ov2slam_hip.h"   /* shared PODs (ov2_ba_problem, ov2_match_input) */

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ front-end ---- */

typedef struct ov2o_level {
    int w, h;        /* level size (unpadded) */
    int pad;         /* border on every side (= winSize, 9 in every shipped config) */
    int stride;      /* elements per padded row = w + 2*pad */
    uint8_t *img;    /* (h+2pad) x stride, REFLECT_101 padded */
    int16_t *grad;   /* (h+2pad) x stride x 2 (Ix,Iy interleaved), zero padded */
} ov2o_level;

typedef struct ov2o_pyr {
    int nlevels;
    ov2o_level lv[8];
} ov2o_pyr;

/* cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), max_level) with the defaults the reference
 * uses (src/visual_front_end.cpp:1172, src/mapper.cpp:81). */
ov2o_pyr *ov2o_pyramid_build(const uint8_t *img, int w, int h, int stride, int win, int max_level);
void ov2o_pyramid_free(ov2o_pyr *p);
int ov2o_pyr_nlevels(const ov2o_pyr *p);
void ov2o_pyr_level_info(const ov2o_pyr *p, int l, int *w, int *h, int *pad, int *stride);
const uint8_t *ov2o_pyr_image(const ov2o_pyr *p, int l);
const int16_t *ov2o_pyr_grad(const ov2o_pyr *p, int l);

/* cv::CLAHE::apply for CV_8UC1 (src/visual_front_end.cpp:1159; tiles from src/ov2slam.cpp:85-89). */
void ov2o_clahe(const uint8_t *src, int w, int h, int stride, float clip, int tiles_x, int tiles_y,
                uint8_t *dst, int dst_stride);

/* cv::calcOpticalFlowPyrLK on prebuilt pyramids, flags = USE_INITIAL_FLOW | LK_GET_MIN_EIGENVALS
 * (src/feature_tracker.cpp:66-69,113-116).  next_xy is in/out.  iters (optional, n*(max_level+1))
 * receives the executed iteration count per point per level (for the algorithmic-bytes model). */
void ov2o_calc_optical_flow_pyr_lk(const ov2o_pyr *prev, const ov2o_pyr *next, int n,
                                   const float *prev_xy, float *next_xy, uint8_t *status, float *err,
                                   int win, int max_level, int max_iter, float eps,
                                   float min_eig_thr, int *iters);

/* worker threads of the call above (cv::parallel_for_ over the points of one calcOpticalFlowPyrLK call); default 1.
 * Results do not depend on it.  Used by bench.py for the N-thread CPU baseline. */
void ov2o_set_num_threads(int n);
int ov2o_get_num_threads(void);

/* FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35-137). priors_xy in/out, status out. */
void ov2o_fb_klt_tracking(const ov2o_pyr *prev, const ov2o_pyr *cur, int win, int nlevels,
                          float err_th, float fb_th, int max_iter, float eps, int n,
                          const float *kps_xy, float *priors_xy, uint8_t *status,
                          int64_t *iter_count /* optional: total LK iterations executed */);

/* VisualFrontEnd::kltTracking batching logic (src/visual_front_end.cpp:132-275) on flat arrays:
 * stage 1 tracks the has_prior kps on 2 levels, failures are re-queued on the full pyramid,
 * <33 % good drops all priors.  out_xy (n*2) = tracked position, out_status (n). */
void ov2o_klt_tracking_frame(const ov2o_pyr *prev, const ov2o_pyr *cur, int win, int nlevels_full,
                             float err_th, float fb_th, int max_iter, float eps, int n,
                             const float *kps_xy, const float *prior_xy, const uint8_t *has_prior,
                             float *out_xy, uint8_t *out_status, int *p3p_req);

/* ------------------------------------------------------------------ detectors (keyframe rate) ---- */

/* FeatureExtractor::detectSingleScale (src/feature_extractor.cpp:288-440): per empty grid cell GaussianBlur 3x3 +
 * cornerMinEigenVal(3,3) + masked arg-max (two candidates per cell), threshold adaptation, cornerSubPix(3x3,30,0.01).
 * The reference runs the cells in a racy parallel_for_ over a SHARED mask: any cell order is a
 * valid execution.  The canonical order used here and on the GPU is the 2x2 colouring of the grid -- colour
 * (r&1)*2+(c&1) ascending, cells of one colour never interact (mask discs have radius cell/4).
 * roi = {x, y, w, h}; dmaxquality in/out; out_xy capacity = number of cells * 2. */
void ov2o_detect_single_scale(const uint8_t *img, int w, int h, int stride, int cell, int n_cur, const float *cur_xy,
                              const int roi[4], double *dmaxquality, int do_subpix, int *n_out, float *out_xy);
/* FeatureExtractor::detectGridFAST (src/feature_extractor.cpp:443-570): per empty cell FAST-9/16 (threshold
 * nfast_th, non-max suppression, mask), best response >= 20, threshold adaptation, cornerSubPix. */
void ov2o_detect_grid_fast(const uint8_t *img, int w, int h, int stride, int cell, int n_cur, const float *cur_xy,
                           const int roi[4], int *nfast_th, int do_subpix, int *n_out, float *out_xy);
/* cv::cornerSubPix(img, pts, Size(3,3), Size(-1,-1), TermCriteria(EPS+MAX_ITER, max_iter, eps)) */
void ov2o_corner_subpix(const uint8_t *img, int w, int h, int stride, int n, float *xy, int half_win, int max_iter,
                        double eps);
/* pieces exposed for unit tests */
void ov2o_draw_disc_u8(uint8_t *mask, int w, int h, int cx, int cy, int radius, uint8_t value);   /* cv::circle filled */
void ov2o_min_eig_cell(const uint8_t *img, int w, int h, int stride, int x0, int y0, int cell, float *hmap);
int ov2o_fast_score(const uint8_t *img, int stride, int x, int y, int threshold);   /* cornerScore<16>, 0 if not a corner */


/* ov2_oracle_tri.c: two-view triangulation + the mapper's acceptance gates (src/mapper.cpp:191-461); returns 0, -1 on a bad group index */
int ov2o_triangulate_pairs(int n, int method, int G, const double *T_ab, const double *Twc_a, const int *grp,
                           const double *bv_a, const double *bv_b, const float *unpx_a, const float *unpx_b,
                           const double *K_a, const double *K_b, float max_reproj_err, double *pt_a, double *wpt,
                           double *parallax, unsigned char *status);

/* ---- stereo matching pieces (ov2_oracle_stereo.c) ------------------------------------------------- */
/* cv::getRectSubPix 8U -> 8U (imgproc/samplers.cpp), dst = ww x wh tight */
void ov2o_get_rect_sub_pix_u8(const uint8_t *src, int w, int h, int stride, int ww, int wh, float cx, float cy, uint8_t *dst);
/* FeatureTracker::getLineMinSAD (src/feature_tracker.cpp:140-213) on two same-size images / on a pyramid level */
void ov2o_line_min_sad_img(const uint8_t *iml, const uint8_t *imr, int w, int h, int stride, float x, float y, int nwinsize,
                           int go_left, float *xprior, float *l1err);
void ov2o_line_min_sad(const ov2o_pyr *left, const ov2o_pyr *right, int level, int nwinsize, int go_left, int n,
                       const float *pts_xy, float *xprior, float *l1err);
/* MultiViewGeometry::computeSampsonDistance (src/multi_view_geometry.cpp:798-821), F row-major */
float ov2o_sampson_distance(const double F[9], float lx, float ly, float rx, float ry);
/* tracking + epipolar gate of MapManager::stereoMatching on flat arrays (src/map_manager.cpp:493-604) */
void ov2o_stereo_matching(const ov2o_pyr *left, const ov2o_pyr *right, int win, int nlevels_full, float err_th, float fb_th,
                          int max_iter, float eps, int n, const float *kps_xy, const float *prior_xy,
                          const uint8_t *has_prior, const float *lunpx_xy, int rectified, const double F_rl[9], const ov2_cam_model *right_cam,
                          float *out_rxy, uint8_t *out_status);

/* ---- keyframe descriptors and map matching (ov2_oracle_match.c) ------------------------------------ */
/* FeatureExtractor::describeBRIEF (src/feature_extractor.cpp:224-285) with a caller-supplied 256 x 4 test table */
void ov2o_describe_brief(const uint8_t *img, int w, int h, int stride, int n, const float *pts_xy, const int8_t *pattern,
                         uint8_t *desc, uint8_t *valid);
/* CameraCalibration::undistortImagePoint / projectCamToImageDist (src/camera_calibration.cpp:254-332) = OpenCV's
 * undistortPoints (5 sweeps) / projectPoints / fisheye::undistortPoints / fisheye::distortPoints restated from their published
 * definitions (OpenCV is not vendored: parity unpinned).  cam NULL or model 0: the pinhole maps. */
void ov2o_cam_undistort(const ov2_cam_model *cam, float u, float v, float *ou, float *ov);
void ov2o_cam_project_dist(const ov2_cam_model *cam, const double K[4], const double pc[3], float *px, float *py);

/* Mapper::matchToMap (src/mapper.cpp:576-774) on the flat inputs of ov2_match_input (include/ov2slam_hip.h) */
void ov2o_match_to_map(const ov2_match_input *in, float fmaxprojerr, float fdistratio, int32_t *match_cand, float *match_dist);

#ifdef __cplusplus
}
#endif
#endif
