// ov2_host_capi.cpp -- flat C hooks around the C++ host mirror so that pytest can build a Frame/MapPoint graph,
// run Optimizer::setupLocalBA (CPU only) or the whole Estimator::applyLocalBA (GPU) and read the map back.
// Test/bring-up surface only; a real integration uses the C++ classes of ov2_host.hpp directly.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "ov2_host.hpp"
#include "ov2_slam.hpp"

using namespace ov2;

namespace {
struct HostMap {
    std::shared_ptr<SlamParams> st = std::make_shared<SlamParams>();
    std::shared_ptr<MapManager> map = std::make_shared<MapManager>();
    std::shared_ptr<CameraCalibration> cl = std::make_shared<CameraCalibration>(), cr = std::make_shared<CameraCalibration>();
    LocalBAProblem pb;
};
}  // namespace

extern "C" {

void *ov2h_map_create(int stereo, int inv_depth, const double *Kl, const double *Kr, const double *T_lr7, int w, int h,
                      int nmin_covscore)
{
    HostMap *m = new HostMap();
    m->st->stereo_ = stereo != 0; m->st->mono_ = !stereo; m->st->buse_inv_depth_ = inv_depth != 0;
    m->st->nmin_covscore_ = nmin_covscore;
    m->cl->fx_ = Kl[0]; m->cl->fy_ = Kl[1]; m->cl->cx_ = Kl[2]; m->cl->cy_ = Kl[3]; m->cl->img_w_ = w; m->cl->img_h_ = h;
    m->cr->fx_ = Kr[0]; m->cr->fy_ = Kr[1]; m->cr->cx_ = Kr[2]; m->cr->cy_ = Kr[3]; m->cr->img_w_ = w; m->cr->img_h_ = h;
    for (int i = 0; i < 7; ++i) m->cr->Tc0ci_.v[i] = T_lr7[i];
    return m;
}

void ov2h_map_destroy(void *p) { delete (HostMap *)p; }

int ov2h_map_add_keyframe(void *p, int kfid, const double *Twc7)
{
    HostMap *m = (HostMap *)p;
    auto f = std::make_shared<Frame>();
    f->id_ = kfid; f->kfid_ = kfid;
    f->pcalib_leftcam_ = m->cl; f->pcalib_rightcam_ = m->cr;
    SE3 T;
    for (int i = 0; i < 7; ++i) T.v[i] = Twc7[i];
    f->setTwc(T);
    m->map->map_pkfs_[kfid] = f;
    return 0;
}

int ov2h_map_add_landmark(void *p, int lmid, const double *xyz, int anchor_kfid)
{
    HostMap *m = (HostMap *)p;
    auto lm = std::make_shared<MapPoint>(lmid, anchor_kfid, true);
    lm->set_kfids_.clear();
    lm->setPoint(Vec3{xyz[0], xyz[1], xyz[2]});
    m->map->map_plms_[lmid] = lm;
    return 0;
}

int ov2h_map_add_obs(void *p, int kfid, int lmid, float ux, float uy, int is_stereo, float rux, float ruy)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(kfid);
    auto lm = m->map->getMapPoint(lmid);
    if (!f || !lm) return -1;
    Keypoint kp;
    kp.lmid_ = lmid; kp.px_ = {ux, uy}; kp.unpx_ = {ux, uy}; kp.is3d_ = true;
    kp.is_stereo_ = is_stereo != 0; kp.rpx_ = {rux, ruy}; kp.runpx_ = {rux, ruy};
    f->addKeypoint(kp);
    lm->addKfObs(kfid);
    return 0;
}

// a keypoint of keyframe kfid that is 2D or 3D (is3d); the map point is created on first use (2D ones carry no position)
int ov2h_map_add_kp(void *p, int kfid, int lmid, float ux, float uy, int is3d, const double *xyz)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(kfid);
    if (!f) return -1;
    auto lm = m->map->getMapPoint(lmid);
    if (!lm) {
        lm = std::make_shared<MapPoint>(lmid, kfid, true);
        if (is3d && xyz) lm->setPoint(Vec3{xyz[0], xyz[1], xyz[2]});
        m->map->map_plms_[lmid] = lm;
    }
    Keypoint kp;
    kp.lmid_ = lmid;
    f->computeKeypoint(Point2f{ux, uy}, kp);
    kp.is3d_ = is3d != 0;
    f->addKeypoint(kp);
    lm->addKfObs(kfid);
    return 0;
}

int ov2h_frame_init_grid(void *p, int kfid, int ncellsize)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    f->initGrid((size_t)ncellsize);
    return 0;
}

// drops a map point but leaves the keypoints that reference it (the "plm == nullptr" branch of stereoMatching, :414)
int ov2h_map_forget_landmark(void *p, int lmid) { ((HostMap *)p)->map->map_plms_.erase(lmid); return 0; }

int ov2h_set_params(void *p, int klt_use_prior, int stereo_rect, int nklt_pyr_lvl, int nklt_win_size)
{
    HostMap *m = (HostMap *)p;
    m->st->klt_use_prior_ = klt_use_prior != 0; m->st->bdo_stereo_rect_ = stereo_rect != 0;
    m->st->nklt_pyr_lvl_ = nklt_pyr_lvl; m->st->nklt_win_size_ = nklt_win_size;
    return 0;
}

// MapManager::stereoMatching(frame, vleftpyr, vrightpyr) on keyframe kfid; the two pyramids are built here the way
// Mapper::run does (src/mapper.cpp:76-81: CLAHE + buildOpticalFlowPyramid of the right image; the left one is the frame's)
int ov2h_stereo_matching(void *p, void *ctx, int kfid, const uint8_t *img_left, const uint8_t *img_right, int w, int h)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(kfid);
    if (!f) return -1;
    ov2_ctx *c = (ov2_ctx *)ctx;
    ov2_pyr *pl = nullptr, *pr = nullptr;
    const SlamParams &st = *m->st;
    ov2_status s = ov2_pyramid_build(c, img_left, w, h, w, st.nklt_win_size_, st.nklt_pyr_lvl_, st.use_clahe_ ? 1 : 0, st.fclahe_val_,
                                     w / 50, h / 50, &pl);
    if (s != OV2_OK) return s;
    Pyramid L(pl);
    s = ov2_pyramid_build(c, img_right, w, h, w, st.nklt_win_size_, st.nklt_pyr_lvl_, st.use_clahe_ ? 1 : 0, st.fclahe_val_, w / 50,
                          h / 50, &pr);
    if (s != OV2_OK) return s;
    Pyramid R(pr);
    FeatureTracker tracker(c, st.nmax_iter_, st.fmax_px_precision_);
    return m->map->stereoMatching(*f, L, R, tracker, st);
}

// VisualFrontEnd::preprocessImage (twice: previous image, current image) + kltTracking with keyframe kfid as the
// current frame (its keypoints sit at their previous-image positions, as at src/visual_front_end.cpp:132)
int ov2h_klt_tracking(void *p, void *ctx, int kfid, const uint8_t *img_prev, const uint8_t *img_cur, int w, int h, int *p3p_req)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(kfid);
    if (!f) return -1;
    ov2_ctx *c = (ov2_ctx *)ctx;
    m->map->pcurframe_ = f;
    auto tracker = std::make_shared<FeatureTracker>(c, m->st->nmax_iter_, m->st->fmax_px_precision_);
    VisualFrontEnd fe(c, m->st, f, m->map, tracker);
    ov2_status s = fe.preprocessImage(img_prev, w, h, w);
    if (s != OV2_OK) return s;
    if ((s = fe.preprocessImage(img_cur, w, h, w)) != OV2_OK) return s;
    s = fe.kltTracking();
    if (p3p_req) *p3p_req = fe.bp3preq_ ? 1 : 0;
    return s;
}

// keypoints of keyframe kfid: lmid, px, is3d, is_stereo, rpx (arrays of capacity cap); returns the count
int ov2h_get_keypoints(void *p, int kfid, int cap, int *lmid, float *px, uint8_t *is3d, uint8_t *is_stereo, float *rpx)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    int n = 0;
    for (const auto &kv : f->mapkps_) {
        if (n < cap) {
            const Keypoint &k = kv.second;
            lmid[n] = k.lmid_; px[2 * n] = k.px_.x; px[2 * n + 1] = k.px_.y; is3d[n] = k.is3d_; is_stereo[n] = k.is_stereo_;
            rpx[2 * n] = k.rpx_.x; rpx[2 * n + 1] = k.rpx_.y;
        }
        ++n;
    }
    return n;
}

// Frame::projWorldToRightImageDist / projWorldToImage of keyframe kfid (test hook)
int ov2h_project(void *p, int kfid, const double *xyz, int right, float *px)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    const Point2f q = right ? f->projWorldToRightImageDist(Vec3{xyz[0], xyz[1], xyz[2]}) : f->projWorldToImage(Vec3{xyz[0], xyz[1], xyz[2]});
    px[0] = q.x; px[1] = q.y;
    return 0;
}

// CameraCalibration::Dcv_ / model_ of the left (cam = 0) or right camera: model 0 = pinhole (k1 k2 p1 p2 [k3]), 1 = fisheye (k1..k4)
int ov2h_set_distortion(void *p, int cam, int model, int n, const double *coeffs)
{
    HostMap *m = (HostMap *)p;
    auto &c = cam ? m->cr : m->cl;
    if (!c || n < 0 || n > 5) return -1;
    c->model_ = model ? CameraCalibration::Fisheye : CameraCalibration::Pinhole;
    c->D_.assign(coeffs, coeffs + n);
    return 0;
}

// CameraCalibration::undistortImagePoint / projectCamToImageDist (test hooks)
int ov2h_undistort(void *p, int cam, float x, float y, float *out)
{
    HostMap *m = (HostMap *)p;
    const auto &c = cam ? m->cr : m->cl;
    if (!c) return -1;
    const Point2f q = c->undistortImagePoint(Point2f{x, y});
    out[0] = q.x; out[1] = q.y;
    return 0;
}

int ov2h_project_dist(void *p, int cam, const double *pc, float *out)
{
    HostMap *m = (HostMap *)p;
    const auto &c = cam ? m->cr : m->cl;
    if (!c) return -1;
    const Point2f q = c->projectCamToImageDist(Vec3{pc[0], pc[1], pc[2]});
    out[0] = q.x; out[1] = q.y;
    return 0;
}

int ov2h_get_frl(void *p, int kfid, double *F9)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    for (int i = 0; i < 9; ++i) F9[i] = f->Frl_[i];
    return 0;
}

int ov2h_map_finalize(void *p, int newkf)
{
    HostMap *m = (HostMap *)p;
    for (auto &kv : m->map->map_pkfs_) m->map->updateFrameCovisibility(*kv.second);
    m->map->pcurframe_ = m->map->getKeyframe(newkf);
    return m->map->pcurframe_ ? 0 : -1;
}

int ov2h_local_ba_setup(void *p, int newkf, int *n_pose, int *n_lm, int *n_res)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(newkf);
    if (!f) return -1;
    Optimizer opt(nullptr, m->st, m->map);
    m->pb = LocalBAProblem();
    opt.setupLocalBA(*f, m->pb);
    *n_pose = (int)m->pb.pose_const.size(); *n_lm = (int)m->pb.lm_lmid.size(); *n_res = (int)m->pb.res_type.size();
    return m->pb.aborted ? 1 : 0;
}

// device map mirror: walk the host map once into an ov2_map; later set-ups (and ov2h_apply_local_ba) use it
int ov2h_map_attach_device(void *p, void *ctx, int max_kf, int max_lm, int max_obs)
{
    HostMap *m = (HostMap *)p;
    return (int)m->map->attachDevice((ov2_ctx *)ctx, max_kf, max_lm, max_obs);
}

// the ov2_map behind the host map (tests compare its tables with a map updated on the device)
void *ov2h_map_device_handle(void *p) { return ((HostMap *)p)->map->dev_; }

// MapManager::flushDevice: push the queued edits of the host objects to the device mirror
int ov2h_map_flush_device(void *p) { return (int)((HostMap *)p)->map->flushDevice(); }

// rows / capacity / compactions of the device mirror's observation table (ov2_map_obs_rows)
int ov2h_map_device_rows(void *p, int *rows, int *capacity, int *compactions)
{
    HostMap *m = (HostMap *)p;
    if (!m->map->dev_) return -1;
    return (int)ov2_map_obs_rows(m->map->dev_, rows, capacity, compactions);
}

int ov2h_local_ba_setup_dev(void *p, int newkf, int *n_pose, int *n_lm, int *n_res)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(newkf);
    if (!f || !m->map->dev_) return -1;
    Optimizer opt(nullptr, m->st, m->map);
    m->pb = LocalBAProblem();
    if (opt.setupLocalBADevice(*f, m->pb) != OV2_OK) return -2;
    *n_pose = (int)m->pb.pose_const.size(); *n_lm = (int)m->pb.lm_lmid.size(); *n_res = (int)m->pb.res_type.size();
    return m->pb.aborted ? 1 : 0;
}

// test hooks for the incremental path: mutate the host map through the MapManager (which queues the device edits)
int ov2h_map_remove_obs(void *p, int kfid, int lmid) { ((HostMap *)p)->map->removeMapPointObs(lmid, kfid); return 0; }
int ov2h_map_remove_landmark(void *p, int lmid) { ((HostMap *)p)->map->removeMapPoint(lmid); return 0; }
int ov2h_map_set_isobs(void *p, int lmid, int isobs)
{
    HostMap *m = (HostMap *)p;
    auto plm = m->map->getMapPoint(lmid);
    if (!plm) return -1;
    plm->isobs_ = isobs != 0;
    m->map->touchMapPoint(lmid);
    return 0;
}
int ov2h_map_bad_lmids(void *p, int *out, int cap)
{
    HostMap *m = (HostMap *)p;
    int n = 0;
    for (int l : m->pb.set_badlmids) { if (n < cap) out[n] = l; ++n; }
    return n;
}

int ov2h_local_ba_get(void *p, int *pose_kfid, uint8_t *pose_const, double *pose, int *lm_lmid, double *lm,
                      int *lm_anchor_kfid, double *lm_anchor_uv, uint8_t *res_type, int *res_kfid, int *res_lmid,
                      double *res_uv)
{
    HostMap *m = (HostMap *)p;
    const LocalBAProblem &pb = m->pb;
    const int e = m->st->buse_inv_depth_ ? 1 : 3;
    for (size_t i = 0; i < pb.pose_const.size(); ++i) { pose_kfid[i] = pb.pose_kfid[i]; pose_const[i] = pb.pose_const[i]; }
    std::memcpy(pose, pb.pose.data(), pb.pose.size() * sizeof(double));
    for (size_t i = 0; i < pb.lm_lmid.size(); ++i) {
        lm_lmid[i] = pb.lm_lmid[i];
        lm_anchor_kfid[i] = pb.lm_anchor_pose[i] >= 0 ? pb.pose_kfid[pb.lm_anchor_pose[i]] : -1;
        lm_anchor_uv[2 * i] = pb.lm_anchor_uv[2 * i]; lm_anchor_uv[2 * i + 1] = pb.lm_anchor_uv[2 * i + 1];
        for (int c = 0; c < e; ++c) lm[i * e + c] = pb.lm[i * e + c];
    }
    for (size_t i = 0; i < pb.res_type.size(); ++i) {
        res_type[i] = pb.res_type[i]; res_kfid[i] = pb.pose_kfid[pb.res_pose[i]]; res_lmid[i] = pb.lm_lmid[pb.res_lm[i]];
        res_uv[2 * i] = pb.res_uv[2 * i]; res_uv[2 * i + 1] = pb.res_uv[2 * i + 1];
    }
    return 0;
}

// Estimator::applyLocalBA on the GPU context `ctx` (an ov2_ctx*)
int ov2h_apply_local_ba(void *p, void *ctx, int newkf, int *n_outliers1, int *n_outliers2, double *final_cost)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(newkf);
    if (!f) return -1;
    auto opt = std::make_shared<Optimizer>((ov2_ctx *)ctx, m->st, m->map);
    Estimator est(m->st, m->map, opt);
    est.pnewkf_ = f;
    const ov2_status s = est.applyLocalBA();
    if (n_outliers1) *n_outliers1 = opt->last_result_.n_outliers_pass1;
    if (n_outliers2) *n_outliers2 = opt->last_result_.n_outliers_pass2;
    if (final_cost) *final_cost = opt->last_result_.l2_done ? opt->last_result_.l2_final_cost : opt->last_result_.final_cost;
    return s;
}

// Optimizer::structureOnlyBA(vlm2optids) on the GPU context `ctx`
int ov2h_structure_only_ba(void *p, void *ctx, int n, const int *lmids, double *final_cost, int *n_iters)
{
    HostMap *m = (HostMap *)p;
    Optimizer opt((ov2_ctx *)ctx, m->st, m->map);
    const ov2_status s = opt.structureOnlyBA(std::vector<int>(lmids, lmids + n));
    if (final_cost) *final_cost = opt.last_result_.final_cost;
    if (n_iters) *n_iters = opt.last_result_.n_log - 1;
    return s;
}

// set-up stage shared by Optimizer::fullBA / looseBA (keyframes kf_lo .. kf_hi, observers above kf_obs_max ignored,
// landmarks with fewer than min_obs observers set aside); read back with ov2h_local_ba_get
int ov2h_range_ba_setup(void *p, int kf_lo, int kf_hi, int kf_obs_max, int min_obs, int *n_pose, int *n_lm, int *n_res)
{
    HostMap *m = (HostMap *)p;
    Optimizer opt(nullptr, m->st, m->map);
    m->pb = LocalBAProblem();
    opt.setupRangeBA(kf_lo, kf_hi, kf_obs_max, (size_t)min_obs, m->pb);
    *n_pose = (int)m->pb.pose_const.size(); *n_lm = (int)m->pb.lm_lmid.size(); *n_res = (int)m->pb.res_type.size();
    return 0;
}

// Optimizer::fullBA(buse_robust_cost) on the GPU context `ctx`
int ov2h_full_ba(void *p, void *ctx, int robust, int *n_out1, int *n_out2, double *final_cost, int *n_iters)
{
    HostMap *m = (HostMap *)p;
    Optimizer opt((ov2_ctx *)ctx, m->st, m->map);
    const ov2_status s = opt.fullBA(robust != 0);
    const ov2_ba_result &r = opt.last_result_;
    if (n_out1) *n_out1 = r.n_outliers_pass1;
    if (n_out2) *n_out2 = r.n_outliers_pass2;
    if (final_cost) *final_cost = r.l2_done ? r.l2_final_cost : r.final_cost;
    if (n_iters) *n_iters = r.n_log;
    return s;
}

// Optimizer::looseBA(inikfid, nkfid, buse_robust_cost)
int ov2h_loose_ba(void *p, void *ctx, int inikfid, int nkfid, int robust, int *n_out1, double *final_cost)
{
    HostMap *m = (HostMap *)p;
    Optimizer opt((ov2_ctx *)ctx, m->st, m->map);
    const ov2_status s = opt.looseBA(inikfid, nkfid, robust != 0);
    if (n_out1) *n_out1 = opt.last_result_.n_outliers_pass1;
    if (final_cost) *final_cost = opt.last_result_.final_cost;
    return s;
}

// Optimizer::localPoseGraph(newframe = keyframe newkf, kfloop_id, newTwc). returns 1 accepted, 0 rejected, < 0 error
int ov2h_local_pose_graph(void *p, void *ctx, int newkf, int kfloop_id, const double *newTwc7, double *final_cost, int *n_log)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(newkf);
    if (!f) return -1;
    SE3 T;
    for (int k = 0; k < 7; ++k) T.v[k] = newTwc7[k];
    Optimizer opt((ov2_ctx *)ctx, m->st, m->map);
    ov2_status s = OV2_OK;
    const bool ok = opt.localPoseGraph(*f, kfloop_id, T, &s);
    if (final_cost) *final_cost = opt.last_pg_.final_cost;
    if (n_log) *n_log = opt.last_pg_.n_log;
    return s != OV2_OK ? -2 : (ok ? 1 : 0);
}

// Optimizer::fullPoseGraph(vTwc, vTpc, viskf) on flat arrays (n x 7, n x 7, n); vTwc is overwritten
int ov2h_full_pose_graph(void *p, void *ctx, int n, double *Twc7, const double *Tpc7, const unsigned char *iskf, double *final_cost)
{
    HostMap *m = (HostMap *)p;
    std::vector<SE3> vTwc((size_t)n), vTpc((size_t)n);
    std::vector<bool> viskf((size_t)n);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 7; ++k) { vTwc[i].v[k] = Twc7[7 * (size_t)i + k]; vTpc[i].v[k] = Tpc7[7 * (size_t)i + k]; }
        viskf[i] = iskf[i] != 0;
    }
    Optimizer opt((ov2_ctx *)ctx, m->st, m->map);
    ov2_status s = OV2_OK;
    const bool ok = opt.fullPoseGraph(vTwc, vTpc, viskf, &s);
    if (final_cost) *final_cost = opt.last_pg_.final_cost;
    if (ok)
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 7; ++k) Twc7[7 * (size_t)i + k] = vTwc[i].v[k];
    return s != OV2_OK ? -2 : (ok ? 1 : 0);
}

// VisualFrontEnd::computePose on keyframe `kfid` taken as the current frame, starting from pose Twc7_init
int ov2h_compute_pose(void *p, void *ctx, int kfid, const double *Twc7_init, int *p3p_req)
{
    HostMap *m = (HostMap *)p;
    auto f = m->map->getKeyframe(kfid);
    if (!f) return -1;
    SE3 T0;
    for (int i = 0; i < 7; ++i) T0.v[i] = Twc7_init[i];
    f->setTwc(T0);
    m->map->pcurframe_ = f;
    VisualFrontEnd fe((ov2_ctx *)ctx, m->st, f, m->map, nullptr);
    const ov2_status s = fe.computePose();
    if (p3p_req) *p3p_req = fe.bp3preq_ ? 1 : 0;
    return s;
}

int ov2h_get_pose(void *p, int kfid, double *Twc7)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    for (int i = 0; i < 7; ++i) Twc7[i] = f->Twc_.v[i];
    return 0;
}

int ov2h_get_landmark(void *p, int lmid, double *xyz, int *n_obs)
{
    auto lm = ((HostMap *)p)->map->getMapPoint(lmid);
    if (!lm) return -1;
    xyz[0] = lm->ptxyz_.x; xyz[1] = lm->ptxyz_.y; xyz[2] = lm->ptxyz_.z;
    if (n_obs) *n_obs = (int)lm->set_kfids_.size();
    return 0;
}

int ov2h_count_keypoints(void *p, int kfid, int *nbkps, int *nb3d, int *nbstereo)
{
    auto f = ((HostMap *)p)->map->getKeyframe(kfid);
    if (!f) return -1;
    *nbkps = (int)f->nbkps_; *nb3d = (int)f->nb3dkps_; *nbstereo = (int)f->nb_stereo_kps_;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// ov2::SlamManager (ov2_slam.hpp): one stereo frame per call through visualTracking and, on keyframes, Mapper::run +
// Estimator::applyLocalBA -- the closed loop in C++.  policy = {kf_every, ba_window, ba_fixed, compose_motion, midpoint_stereo,
// pose_from_kf} (all zero = the reference's own heuristics); rect != 0: rectified rig (bdo_stereo_rect_).
void *ov2h_slam_create(void *ctx, const double *K4, double baseline, int w, int h, int cell, int rect, const int *policy, int use_device_map)
{
    auto st = std::make_shared<SlamParams>();
    st->stereo_ = true; st->mono_ = false; st->bdo_stereo_rect_ = rect != 0;
    st->nmaxdist_ = cell;
    st->nbmaxkps_ = (int)(std::ceil((float)w / cell) * std::ceil((float)h / cell));   // src/slam_params.cpp:107-110
    auto cl = std::make_shared<CameraCalibration>(), cr = std::make_shared<CameraCalibration>();
    cl->fx_ = cr->fx_ = K4[0]; cl->fy_ = cr->fy_ = K4[1]; cl->cx_ = cr->cx_ = K4[2]; cl->cy_ = cr->cy_ = K4[3];
    cl->img_w_ = cr->img_w_ = w; cl->img_h_ = cr->img_h_ = h;
    cr->Tc0ci_.v = {baseline, 0, 0, 0, 0, 0, 1};
    LoopPolicy pol;
    if (policy) {
        pol.kf_every = policy[0]; pol.ba_window = policy[1]; pol.ba_fixed = policy[2]; pol.compose_motion = policy[3] != 0;
        pol.midpoint_stereo = policy[4] != 0; pol.pose_from_kf = policy[5] != 0;
    }
    SlamManager *S = new SlamManager((ov2_ctx *)ctx, st, cl, cr, pol);
    if (use_device_map && S->pmap_->attachDevice((ov2_ctx *)ctx, 64, 4096, 16384) != OV2_OK) { delete S; return nullptr; }
    return S;
}

int ov2h_slam_add_stereo(void *p, double time, const uint8_t *left, const uint8_t *right, int w, int h)
{
    return (int)((SlamManager *)p)->addNewStereoImages(time, left, right, w, h, w);
}

void ov2h_slam_pose(void *p, double *Twc7)
{
    const SE3 T = ((SlamManager *)p)->pose();
    for (int i = 0; i < 7; ++i) Twc7[i] = T.v[i];
}

// out[16]: frame, keypoints, 3D keypoints, keyframe?, new keypoints, stereo keypoints, 3D landmarks, BA ran?, residual blocks,
// robust iterations, L2 iterations, outliers, cost before, cost after, keyframes so far, landmarks so far
void ov2h_slam_stats(void *p, double *out)
{
    SlamManager *S = (SlamManager *)p;
    const SlamStats &s = S->last_;
    const double v[16] = {(double)s.frame, (double)s.tracked, (double)s.n3d, (double)s.is_kf, (double)s.n_new, (double)s.n_stereo,
                          (double)s.n_lm3d, (double)s.ba_done, (double)s.ba_res, (double)s.ba_it_robust, (double)s.ba_it_l2,
                          (double)s.ba_outliers, s.ba_cost0, s.ba_cost1, (double)S->pmap_->map_pkfs_.size(), (double)S->pmap_->map_plms_.size()};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
}

// ids + world points of the 3D landmarks (capacity cap); returns the count
int ov2h_slam_landmarks(void *p, int cap, int *lmid, double *xyz)
{
    SlamManager *S = (SlamManager *)p;
    std::vector<int> ids;
    for (const auto &kv : S->pmap_->map_plms_) if (kv.second->is3d_) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    int n = 0;
    for (int id : ids) {
        if (n < cap) { const Vec3 q = S->pmap_->getMapPoint(id)->getPoint(); lmid[n] = id; xyz[3 * n] = q.x; xyz[3 * n + 1] = q.y; xyz[3 * n + 2] = q.z; }
        ++n;
    }
    return n;
}

void ov2h_slam_destroy(void *p) { delete (SlamManager *)p; }

// keyframe descriptors + Mapper::matchingToLocalMap: pattern = the 256 x 4 int8 BRIEF test table (NULL keeps the current one)
void ov2h_slam_set_brief(void *p, const int8_t *pattern, int use_brief, int track_localmap, float fmax_desc_dist, float fmax_proj_pxdist)
{
    SlamManager *S = (SlamManager *)p;
    if (pattern) S->setBriefPattern(pattern);
    S->pslamstate_->use_brief_ = use_brief != 0;
    S->pslamstate_->bdo_track_localmap_ = track_localmap != 0;
    if (fmax_desc_dist > 0.f) S->pslamstate_->fmax_desc_dist_ = fmax_desc_dist;
    if (fmax_proj_pxdist > 0.f) S->pslamstate_->fmax_proj_pxdist_ = fmax_proj_pxdist;
}

// out[3] of the last frame (keyframes only): keypoints described, local map points offered to matchToMap, merges
void ov2h_slam_kf_stats(void *p, double *out)
{
    const SlamStats &s = ((SlamManager *)p)->last_;
    out[0] = s.n_described; out[1] = s.n_local; out[2] = s.n_matched;
}

void *ov2h_slam_device_handle(void *p) { return ((SlamManager *)p)->pmap_->dev_; }
int ov2h_slam_flush_device(void *p) { return (int)((SlamManager *)p)->pmap_->flushDevice(); }

// Invariants of the host map after any sequence of frames.  out[6]: keypoints of keyframes whose map point is missing,
// keypoints whose map point does not list the keyframe, observers a map point lists that do not hold it (or are gone),
// covisibility entries that differ from the count of co-observed map points, 3D map points without a descriptor although
// use_brief_ is on, map points with descriptors from keyframes that do not observe them.  returns their sum
int ov2h_slam_check_map(void *p, int *out)
{
    SlamManager *S = (SlamManager *)p;
    MapManager &M = *S->pmap_;
    int v[6] = {0, 0, 0, 0, 0, 0};
    for (const auto &kf : M.map_pkfs_) {
        std::map<int, int> cov;
        for (const auto &kv : kf.second->mapkps_) {
            auto plm = M.getMapPoint(kv.first);
            if (!plm) { ++v[0]; continue; }
            if (!plm->set_kfids_.count(kf.first)) ++v[1];
            for (int o : plm->set_kfids_) if (o != kf.first && M.getKeyframe(o)) cov[o]++;
        }
        for (const auto &c : cov) { auto it = kf.second->map_covkfs_.find(c.first); if (it == kf.second->map_covkfs_.end() || it->second != c.second) ++v[3]; }
        for (const auto &c : kf.second->map_covkfs_) if (!cov.count(c.first) && M.getKeyframe(c.first)) ++v[3];
    }
    for (const auto &lm : M.map_plms_) {
        for (int kfid : lm.second->set_kfids_) {
            auto pkf = M.getKeyframe(kfid);
            if (!pkf || !pkf->mapkps_.count(lm.first)) ++v[2];
        }
        if (S->pslamstate_->use_brief_ && lm.second->is3d_ && !lm.second->has_desc_) ++v[4];
        for (const auto &d : lm.second->map_kf_desc_) if (!lm.second->set_kfids_.count(d.first)) ++v[5];
    }
    int tot = 0;
    for (int i = 0; i < 6; ++i) { if (out) out[i] = v[i]; tot += v[i]; }
    return tot;
}

// the host map as flat arrays (capacity caps; counts come back in n[3] = keyframes, landmarks, observations):
// keyframes (id, Twc), landmarks (id, xyz, state bits: 1 alive | 2 is3d | 4 isobs), observations (kfid, lmid, stereo)
void ov2h_slam_export_map(void *p, int cap_kf, int cap_lm, int cap_obs, int *n, int *kf_id, double *kf_pose, int *lm_id, double *lm_xyz,
                          uint8_t *lm_state, int *obs_kf, int *obs_lm, uint8_t *obs_stereo)
{
    MapManager &M = *((SlamManager *)p)->pmap_;
    int nk = 0, nl = 0, no = 0;
    std::map<int, std::shared_ptr<Frame>> kfs(M.map_pkfs_.begin(), M.map_pkfs_.end());
    for (const auto &kf : kfs) {
        if (nk < cap_kf) { kf_id[nk] = kf.first; const SE3 T = kf.second->getTwc(); for (int i = 0; i < 7; ++i) kf_pose[7 * nk + i] = T.v[i]; }
        ++nk;
        std::map<int, Keypoint> kps(kf.second->mapkps_.begin(), kf.second->mapkps_.end());
        for (const auto &kv : kps) {
            if (no < cap_obs) { obs_kf[no] = kf.first; obs_lm[no] = kv.first; obs_stereo[no] = kv.second.is_stereo_ ? 1 : 0; }
            ++no;
        }
    }
    std::map<int, std::shared_ptr<MapPoint>> lms(M.map_plms_.begin(), M.map_plms_.end());
    for (const auto &lm : lms) {
        if (nl < cap_lm) {
            lm_id[nl] = lm.first; const Vec3 q = lm.second->getPoint(); lm_xyz[3 * nl] = q.x; lm_xyz[3 * nl + 1] = q.y; lm_xyz[3 * nl + 2] = q.z;
            lm_state[nl] = (uint8_t)(1 | (lm.second->is3d_ ? 2 : 0) | (lm.second->isobs_ ? 4 : 0));
        }
        ++nl;
    }
    n[0] = nk; n[1] = nl; n[2] = no;
}

// ---- MapPoint descriptor bookkeeping alone (tests: against a restatement of src/map_point.cpp:106-211) ----
void *ov2h_mp_new(int lmid, int kfid, const uint8_t *desc)
{
    if (!desc) return new MapPoint(lmid, kfid, true);
    Desc d; std::copy(desc, desc + 32, d.begin());
    return new MapPoint(lmid, kfid, d, true);
}
void ov2h_mp_add_obs(void *p, int kfid) { ((MapPoint *)p)->addKfObs(kfid); }
void ov2h_mp_add_desc(void *p, int kfid, const uint8_t *desc) { Desc d; std::copy(desc, desc + 32, d.begin()); ((MapPoint *)p)->addDesc(kfid, d); }
void ov2h_mp_remove_obs(void *p, int kfid) { ((MapPoint *)p)->removeKfObs(kfid); }
// out_i[4] = has_desc, anchor kfid, observers, descriptors; desc[32] = the representative one; per descriptor (cap): kfid + summed distance
int ov2h_mp_state(void *p, int *out_i, uint8_t *desc, int cap, int *kfids, float *dists)
{
    MapPoint &m = *(MapPoint *)p;
    out_i[0] = m.has_desc_ ? 1 : 0; out_i[1] = m.kfid_; out_i[2] = (int)m.set_kfids_.size(); out_i[3] = (int)m.map_kf_desc_.size();
    std::copy(m.desc_.begin(), m.desc_.end(), desc);
    std::map<int, float> d(m.map_desc_dist_.begin(), m.map_desc_dist_.end());
    int n = 0;
    for (const auto &kv : d) { if (n < cap) { kfids[n] = kv.first; dists[n] = kv.second; } ++n; }
    return n;
}
void ov2h_mp_free(void *p) { delete (MapPoint *)p; }

// ---------------------------------------------------------------------------------------------------------------
// Estimator threads of `nseq` SLAM instances (reference src/estimator.cpp:32-98 run(): wait for a keyframe ->
// applyLocalBA -> next), served by ONE native thread with its own HIP context: it takes every sequence that has a
// keyframe pending and solves their windows in one ov2_ba_solve_batch call (at most max_batch windows per call).  A
// keyframe that arrives while its sequence still has one pending replaces it (the reference keeps only the newest,
// src/estimator.cpp:185-210).  Every job solves a fresh copy of the window it was created with.  Native so that it runs
// beside a Python front-end loop without sharing the interpreter lock.
struct BaWorkerNative {
    ov2_ctx *ctx = nullptr;
    std::vector<double> pose0, lm0, lm_auv, res_uv, res_sigma;
    std::vector<uint8_t> pose_const, res_type;
    std::vector<int32_t> lm_anchor, res_pose, res_lm;
    ov2_ba_problem P{};
    ov2_ba_options opt{};
    int max_batch = 64;
    std::vector<uint8_t> pending;
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
    bool stop = false, counting = false;
    long long solves = 0, iters = 0, dropped = 0, submitted = 0, batches = 0;
    double busy_s = 0.0;
    int last_status = 0;

    // device-resident windows (ov2_ba_solve_batch_dev): the window's arrays are uploaded once, every job solves its own
    // device copy of the state -- what a device-side producer of the problems (the map mirror) hands over
    bool device_resident = false;
    ov2_ba_problem Pd{};                       // P with device pointers
    std::vector<void *> pose_d, lm_d, owned_d;

    bool dev_setup()
    {
        auto up = [&](const void *h, size_t bytes) -> void * {
            if (!h || !bytes) return nullptr;
            void *d = nullptr;
            if (ov2_dev_alloc(ctx, bytes, &d) != OV2_OK) return nullptr;
            owned_d.push_back(d);
            if (ov2_memcpy_h2d(ctx, d, h, bytes) != OV2_OK) return nullptr;
            return d;
        };
        const size_t e = P.inv_depth ? 1 : 3, np = (size_t)P.n_pose, nl = (size_t)P.n_lm, nr = (size_t)P.n_res;
        Pd = P;
        Pd.pose_const = (const uint8_t *)up(P.pose_const, np);
        Pd.lm_anchor_pose = (const int32_t *)up(P.lm_anchor_pose, nl * 4);
        Pd.lm_anchor_uv = (const double *)up(P.lm_anchor_uv, 2 * nl * 8);
        Pd.res_type = (const uint8_t *)up(P.res_type, nr);
        Pd.res_pose = (const int32_t *)up(P.res_pose, nr * 4);
        Pd.res_lm = (const int32_t *)up(P.res_lm, nr * 4);
        Pd.res_uv = (const double *)up(P.res_uv, 2 * nr * 8);
        Pd.res_sigma = (const double *)up(P.res_sigma, nr * 8);
        (void)e;
        return (Pd.res_uv || !nr) && (Pd.pose_const || !np) && (Pd.res_type || !nr);
    }

    // a fresh copy of the window per job: the states of all jobs of a batch live in two contiguous device buffers that
    // are re-filled from pre-replicated templates with two copies per batch (one copy per job and array was 128 small
    // transfers in front of every batch of 64)
    void *tmpl_pose_d = nullptr, *tmpl_lm_d = nullptr, *state_pose_d = nullptr, *state_lm_d = nullptr;
    size_t state_cap = 0;

    bool dev_job_state(size_t nb)
    {
        const size_t e = P.inv_depth ? 1 : 3, pb = 7 * (size_t)P.n_pose * 8, lb = e * (size_t)P.n_lm * 8;
        if (nb > state_cap) {
            const size_t cap = std::max<size_t>(nb, (size_t)max_batch);
            std::vector<double> rp((size_t)7 * P.n_pose * cap), rl(e * (size_t)P.n_lm * cap);
            for (size_t k = 0; k < cap; ++k) {
                std::copy(pose0.begin(), pose0.end(), rp.begin() + k * pose0.size());
                std::copy(lm0.begin(), lm0.end(), rl.begin() + k * lm0.size());
            }
            void *bufs[4] = {nullptr, nullptr, nullptr, nullptr};
            const size_t bytes[4] = {pb * cap, lb * cap, pb * cap, lb * cap};
            for (int i = 0; i < 4; ++i) {
                if (ov2_dev_alloc(ctx, bytes[i] ? bytes[i] : 8, &bufs[i]) != OV2_OK) return false;
                owned_d.push_back(bufs[i]);
            }
            if (pb && ov2_memcpy_h2d(ctx, bufs[0], rp.data(), pb * cap) != OV2_OK) return false;
            if (lb && ov2_memcpy_h2d(ctx, bufs[1], rl.data(), lb * cap) != OV2_OK) return false;
            tmpl_pose_d = bufs[0]; tmpl_lm_d = bufs[1]; state_pose_d = bufs[2]; state_lm_d = bufs[3];
            state_cap = cap;
            pose_d.assign(cap, nullptr); lm_d.assign(cap, nullptr);
            for (size_t k = 0; k < cap; ++k) { pose_d[k] = (char *)state_pose_d + k * pb; lm_d[k] = (char *)state_lm_d + k * lb; }
        }
        if (pb && ov2_memcpy_d2d(ctx, state_pose_d, tmpl_pose_d, pb * nb) != OV2_OK) return false;   // on the solver's stream
        if (lb && ov2_memcpy_d2d(ctx, state_lm_d, tmpl_lm_d, lb * nb) != OV2_OK) return false;
        return true;
    }

    void loop()
    {
        size_t rr = 0;
        bool dev_ready = false, dev_failed = false, use_dev = false;
        std::vector<int> jobs;
        std::vector<std::vector<double>> poses, lms;   // per job: the window's own state (solved in place)
        std::vector<ov2_ba_problem> Ps;
        std::vector<ov2_ba_result> Rs;
        for (;;) {
            jobs.clear();
            bool counted = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (stop) return;
                    for (size_t k = 0; k < pending.size() && (int)jobs.size() < max_batch; ++k) {
                        const size_t b = (rr + k) % pending.size();
                        if (pending[b]) { pending[b] = 0; jobs.push_back((int)b); }
                    }
                    if (!jobs.empty()) { rr = ((size_t)jobs.back() + 1) % pending.size(); break; }
                    cv.wait_for(lk, std::chrono::milliseconds(2));
                }
                counted = counting;   // a batch counts only if it started inside the counted region
                use_dev = device_resident;   // latched under the mutex ov2h_ba_worker_set_device_resident takes
            }
            const size_t nb = jobs.size();
            if (poses.size() < nb) { poses.resize(nb); lms.resize(nb); }
            Ps.assign(nb, P);
            Rs.resize(nb);
            for (size_t k = 0; k < nb; ++k) {
                if (!use_dev) {
                    poses[k] = pose0; lms[k] = lm0;
                    Ps[k].pose = poses[k].data(); Ps[k].lm = lms[k].data();
                }
                std::memset(&Rs[k], 0, sizeof(ov2_ba_result));
            }
            if (use_dev && !dev_ready && !dev_failed) { dev_ready = dev_setup(); dev_failed = !dev_ready; }   // a failed upload is not retried
            const auto t0 = std::chrono::steady_clock::now();
            ov2_status s;
            if (use_dev) {
                s = OV2_ERR_NOMEM;
                if (dev_ready && dev_job_state(nb)) {
                    Ps.assign(nb, Pd);
                    for (size_t k = 0; k < nb; ++k) { Ps[k].pose = (double *)pose_d[k]; Ps[k].lm = (double *)lm_d[k]; }
                    s = ov2_ba_solve_batch_dev(ctx, (int)nb, Ps.data(), &opt, Rs.data());
                }
            } else {
                s = ov2_ba_solve_batch(ctx, (int)nb, Ps.data(), &opt, Rs.data());
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> lk(mu);
            if (last_status == 0) last_status = s;   // sticky: a later good batch must not hide a failed one from stats()
            if (counted && counting && s == OV2_OK) {   // ... and finished inside it
                ++batches;
                for (size_t k = 0; k < nb; ++k) {
                    ++solves;
                    const int it = Rs[k].n_log - 1 - (Rs[k].l2_done ? 1 : 0);   // the log holds one iteration-0 record per solve
                    iters += it > 0 ? it : 0;
                }
                busy_s += dt;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// The same Estimator threads with the WHOLE of Optimizer::localBA per keyframe job, on device-resident maps: every
// sequence owns an ov2_map (its own keyframes / landmarks / observations, its own window size and outlier rate); a job is
//     [the mapper's edits: here the map is rewound to its saved state, ov2_map_restore_state_batch]
//     set-up   ov2_map_local_ba_setup_batch    src/optimizer.cpp:43-430   (one synchronisation for all pending maps)
//     solve    ov2_ba_solve_batch_dev          :439-735                   (on the flat problems the set-up left in HBM)
//     update   ov2_map_local_ba_update_batch   :741-882                   (on the tables, asynchronous)
// and a batch ends with one synchronisation of the worker's context, so that the counted time closes over all three stages.
// The context and the maps are created by the caller (on that context) and belong to this thread until it is destroyed.
struct BaPipelineNative {
    ov2_ctx *ctx = nullptr;
    std::vector<ov2_map *> maps;
    std::vector<int32_t> newkf;
    std::vector<double> calib_l;   // nseq x 4
    ov2_ba_problem proto{};
    ov2_ba_options opt{};
    int inv_depth = 1, max_batch = 64;
    std::vector<uint8_t> pending;
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
    bool stop = false, counting = false;
    long long solves = 0, iters = 0, dropped = 0, submitted = 0, batches = 0, slowest_sum = 0, res_blocks = 0, aborted = 0, iter_blocks = 0;
    long long hist_robust[8] = {0}, hist_l2[16] = {0};
    double busy_s = 0.0, t_setup = 0.0, t_solve = 0.0, t_update = 0.0;
    int last_status = 0;

    void loop()
    {
        size_t rr = 0;
        std::vector<int> jobs;
        std::vector<ov2_map *> M;
        std::vector<int32_t> nk;
        std::vector<double> K;
        std::vector<ov2_local_ba_setup> V;
        std::vector<ov2_ba_problem> Ps;
        std::vector<ov2_ba_result> Rs;
        std::vector<const uint8_t *> outl;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double>(b - a).count();
        };
        for (;;) {
            jobs.clear();
            bool counted = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (stop) return;
                    for (size_t k = 0; k < pending.size() && (int)jobs.size() < max_batch; ++k) {
                        const size_t b = (rr + k) % pending.size();
                        if (pending[b]) { pending[b] = 0; jobs.push_back((int)b); }
                    }
                    if (!jobs.empty()) { rr = ((size_t)jobs.back() + 1) % pending.size(); break; }
                    cv.wait_for(lk, std::chrono::milliseconds(2));
                }
                counted = counting;   // a batch counts only if it started inside the counted region
            }
            const int nb = (int)jobs.size();
            M.resize(nb); nk.resize(nb); K.resize(4 * (size_t)nb); V.resize(nb); Ps.resize(nb); Rs.resize(nb); outl.resize(nb);
            for (int k = 0; k < nb; ++k) {
                M[k] = maps[jobs[k]]; nk[k] = newkf[jobs[k]];
                for (int q = 0; q < 4; ++q) K[4 * (size_t)k + q] = calib_l[4 * (size_t)jobs[k] + q];
            }
            const auto t0 = now();
            ov2_status s = ov2_map_restore_state_batch(ctx, nb, M.data());
            if (s == OV2_OK) s = ov2_map_local_ba_setup_batch(ctx, nb, M.data(), nk.data(), 25, 1, inv_depth, K.data(), V.data());
            const auto t1 = now();
            long long blocks = 0, nab = 0;
            if (s == OV2_OK) {
                for (int k = 0; k < nb; ++k) {
                    const ov2_local_ba_setup &v = V[k];
                    ov2_ba_problem &P = Ps[k];
                    P = proto;
                    for (int q = 0; q < 4; ++q) P.calib_l[q] = K[4 * (size_t)k + q];
                    P.inv_depth = inv_depth;
                    if (v.aborted) { P.n_pose = P.n_lm = P.n_res = 0; ++nab; }
                    else { P.n_pose = v.n_pose; P.n_lm = v.n_lm; P.n_res = v.n_res; blocks += v.n_res; }
                    P.pose = v.pose; P.pose_const = v.pose_const; P.lm = v.lm; P.lm_anchor_pose = v.lm_anchor_pose; P.lm_anchor_uv = v.lm_anchor_uv;
                    P.res_type = v.res_type; P.res_pose = v.res_pose; P.res_lm = v.res_lm; P.res_uv = v.res_uv; P.res_sigma = v.res_sigma;
                    std::memset(&Rs[k], 0, sizeof(ov2_ba_result));
                    Rs[k].outlier = v.aborted ? nullptr : v.res_outlier;
                    outl[k] = Rs[k].outlier;
                }
                s = ov2_ba_solve_batch_dev(ctx, nb, Ps.data(), &opt, Rs.data());
            }
            const auto t2 = now();
            if (s == OV2_OK) s = ov2_map_local_ba_update_batch(ctx, nb, M.data(), outl.data(), nk.data(), nullptr);
            if (s == OV2_OK) s = ov2_ctx_synchronize(ctx);
            const auto t3 = now();
            std::lock_guard<std::mutex> lk(mu);
            if (last_status == 0) last_status = s;   // sticky: a later good batch must not hide a failed one from stats()
            if (counted && counting && s == OV2_OK) {   // ... and finished inside it
                ++batches;
                int slowest = 0;
                for (int k = 0; k < nb; ++k) {
                    ++solves;
                    const int n1 = Rs[k].n_log_robust > 0 ? Rs[k].n_log_robust - 1 : 0;
                    const int n2 = Rs[k].l2_done ? Rs[k].n_log - Rs[k].n_log_robust - 1 : 0;
                    iters += n1 + (n2 > 0 ? n2 : 0);
                    hist_robust[n1 < 7 ? n1 : 7]++;
                    hist_l2[n2 < 0 ? 0 : (n2 < 15 ? n2 : 15)]++;
                    slowest = std::max(slowest, n1 + (n2 > 0 ? n2 : 0));
                    iter_blocks += (long long)(n1 + (n2 > 0 ? n2 : 0)) * Ps[k].n_res;
                }
                slowest_sum += slowest; res_blocks += blocks; aborted += nab;
                busy_s += sec(t0, t3); t_setup += sec(t0, t1); t_solve += sec(t1, t2); t_update += sec(t2, t3);
            }
        }
    }
};

void *ov2h_ba_pipeline_create(void *ctx, int nseq, void *const *maps, const int *newkf, const double *calib_l, const ov2_ba_problem *proto,
                              float robust_mono_th, int inv_depth, int max_batch)
{
    if (!ctx || nseq <= 0 || !maps || !newkf || !calib_l || !proto) return nullptr;
    BaPipelineNative *w = new BaPipelineNative();
    w->ctx = (ov2_ctx *)ctx;
    w->maps.assign((ov2_map *const *)maps, (ov2_map *const *)maps + nseq);
    w->newkf.assign(newkf, newkf + nseq);
    w->calib_l.assign(calib_l, calib_l + 4 * (size_t)nseq);
    w->proto = *proto;   // calibrations + extrinsic; the array pointers are replaced per job
    w->inv_depth = inv_depth ? 1 : 0;
    w->max_batch = max_batch > 0 ? max_batch : 1;
    ov2_ba_default_options(&w->opt, robust_mono_th);
    w->pending.assign((size_t)nseq, 0);
    w->th = std::thread([w] { w->loop(); });
    return w;
}

void ov2h_ba_pipeline_submit_all(void *p)
{
    BaPipelineNative *w = (BaPipelineNative *)p;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        for (auto &f : w->pending) {
            if (f && w->counting) ++w->dropped;
            f = 1;
            if (w->counting) ++w->submitted;
        }
    }
    w->cv.notify_one();
}

void ov2h_ba_pipeline_set_counting(void *p, int on)
{
    BaPipelineNative *w = (BaPipelineNative *)p;
    std::lock_guard<std::mutex> lk(w->mu);
    w->counting = on != 0;
}

// out[40]: 0 solves, 1 LM iterations, 2 jobs replaced by a newer keyframe, 3 jobs submitted, 4 busy seconds, 5 last status,
// 6 batches, 7 set-up s, 8 solve s, 9 update s, 10 sum over batches of the slowest window's iterations, 11 residual blocks solved,
// 12 aborted set-ups, 13 sum over windows of LM iterations x residual blocks, 16..23 windows by robust iterations (0..7+), 24..39 windows by L2 iterations (0..15+)
void ov2h_ba_pipeline_stats(void *p, double *out)
{
    BaPipelineNative *w = (BaPipelineNative *)p;
    std::lock_guard<std::mutex> lk(w->mu);
    for (int i = 0; i < 40; ++i) out[i] = 0.0;
    out[0] = (double)w->solves; out[1] = (double)w->iters; out[2] = (double)w->dropped; out[3] = (double)w->submitted;
    out[4] = w->busy_s; out[5] = (double)w->last_status; out[6] = (double)w->batches;
    out[7] = w->t_setup; out[8] = w->t_solve; out[9] = w->t_update; out[10] = (double)w->slowest_sum; out[11] = (double)w->res_blocks;
    out[12] = (double)w->aborted; out[13] = (double)w->iter_blocks;
    for (int i = 0; i < 8; ++i) out[16 + i] = (double)w->hist_robust[i];
    for (int i = 0; i < 16; ++i) out[24 + i] = (double)w->hist_l2[i];
}

// stops the thread; the context and the maps stay with the caller
void ov2h_ba_pipeline_destroy(void *p)
{
    BaPipelineNative *w = (BaPipelineNative *)p;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
}

void *ov2h_ba_worker_create(int device, const ov2_ba_problem *P, float robust_mono_th, int nseq, int max_batch, int high_priority)
{
    BaWorkerNative *w = new BaWorkerNative();
    w->max_batch = max_batch > 0 ? max_batch : 1;
    if (ov2_ctx_create_ex(device, high_priority ? 1 : 0, &w->ctx) != OV2_OK) { delete w; return nullptr; }
    const int e = P->inv_depth ? 1 : 3;
    w->pose0.assign(P->pose, P->pose + 7 * (size_t)P->n_pose);
    w->lm0.assign(P->lm, P->lm + (size_t)e * P->n_lm);
    w->pose_const.assign(P->pose_const, P->pose_const + P->n_pose);
    if (P->inv_depth) {
        w->lm_anchor.assign(P->lm_anchor_pose, P->lm_anchor_pose + P->n_lm);
        w->lm_auv.assign(P->lm_anchor_uv, P->lm_anchor_uv + 2 * (size_t)P->n_lm);
    }
    w->res_type.assign(P->res_type, P->res_type + P->n_res);
    w->res_pose.assign(P->res_pose, P->res_pose + P->n_res);
    w->res_lm.assign(P->res_lm, P->res_lm + P->n_res);
    w->res_uv.assign(P->res_uv, P->res_uv + 2 * (size_t)P->n_res);
    if (P->res_sigma) w->res_sigma.assign(P->res_sigma, P->res_sigma + P->n_res);
    w->P = *P;
    w->P.pose_const = w->pose_const.data();
    w->P.lm_anchor_pose = P->inv_depth ? w->lm_anchor.data() : nullptr;
    w->P.lm_anchor_uv = P->inv_depth ? w->lm_auv.data() : nullptr;
    w->P.res_type = w->res_type.data(); w->P.res_pose = w->res_pose.data(); w->P.res_lm = w->res_lm.data();
    w->P.res_uv = w->res_uv.data(); w->P.res_sigma = P->res_sigma ? w->res_sigma.data() : nullptr;
    ov2_ba_default_options(&w->opt, robust_mono_th);
    w->pending.assign((size_t)(nseq > 0 ? nseq : 1), 0);
    w->th = std::thread([w] { w->loop(); });
    return w;
}

// before the first submission: the worker keeps its windows in device memory and solves them with ov2_ba_solve_batch_dev
void ov2h_ba_worker_set_device_resident(void *p, int on)
{
    BaWorkerNative *w = (BaWorkerNative *)p;
    std::lock_guard<std::mutex> lk(w->mu);
    w->device_resident = on != 0;
}

// Mapper::run -> Estimator::addNewKf for every sequence of the batch
void ov2h_ba_worker_submit_all(void *p)
{
    BaWorkerNative *w = (BaWorkerNative *)p;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        for (auto &f : w->pending) {
            if (f && w->counting) ++w->dropped;
            f = 1;
            if (w->counting) ++w->submitted;
        }
    }
    w->cv.notify_one();
}

void ov2h_ba_worker_set_counting(void *p, int on)
{
    BaWorkerNative *w = (BaWorkerNative *)p;
    std::lock_guard<std::mutex> lk(w->mu);
    w->counting = on != 0;
}

// out[7] = solves, LM iterations, jobs replaced by a newer keyframe, jobs submitted, busy seconds, last status, batches
void ov2h_ba_worker_stats(void *p, double *out)
{
    BaWorkerNative *w = (BaWorkerNative *)p;
    std::lock_guard<std::mutex> lk(w->mu);
    out[0] = (double)w->solves; out[1] = (double)w->iters; out[2] = (double)w->dropped; out[3] = (double)w->submitted;
    out[4] = w->busy_s; out[5] = (double)w->last_status; out[6] = (double)w->batches;
}

void ov2h_ba_worker_destroy(void *p)
{
    BaWorkerNative *w = (BaWorkerNative *)p;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    for (void *d : w->owned_d) ov2_dev_free(w->ctx, d);
    ov2_ctx_destroy(w->ctx);
    delete w;
}


// ---------------------------------------------------------------------------------------------------------------------
// Native per-frame driver of the bench streams: what SlamManager::run's loop does per image for B lock-step sequences
// (preprocessImage -> kltTracking -> ceresPnP; on a keyframe: right pyramid -> stereoMatching -> detector, then the
// keyframe goes to the Estimator workers) over device-resident inputs, enqueued from C++ so that small streams (one
// 308-keypoint sequence) are not bound by the interpreter's ~15 us per ctypes call.  Every pointer is a device pointer
// owned by the caller; nothing is synchronised here.
struct ov2h_feloop_cfg {
    int32_t L, B, n, win, nlvl, use_clahe, tiles_x, tiles_y;
    float clahe_clip, err_th, fb_th, eps;
    int32_t max_iter, detect, det_cell, det_ncur, det_cap, pnp;
    void *const *left, *const *right;                   // L batches of images (ov2_images*)
    const float *const *kps, *const *pri, *const *st_pri;   // per cycle position: n x 2
    const uint8_t *const *has, *const *st_has;
    const int32_t *img_idx;
    float *out_xy; uint8_t *out_st; int32_t *p3p;
    // per-frame pose refinement (one problem per sequence, re-solved from T0 every frame)
    const int32_t *pnp_off; const double *pnp_unpx, *pnp_wpts, *pnp_K, *pnp_T0; double *pnp_T;
    uint8_t *pnp_outl, *pnp_rem; int32_t *pnp_ok; uint64_t pnp_T_bytes;
    // keyframe detector
    double *det_thresh; const float *det_cur; const int32_t *det_img; int32_t *det_nout; float *det_out;
};

struct FeLoopNative {
    ov2_ctx *ctx = nullptr;
    ov2h_feloop_cfg c{};
    std::vector<void *> left, right;
    std::vector<const float *> kps, pri, st_pri;
    std::vector<const uint8_t *> has, st_has;
    ov2_pyr *prev = nullptr;
    long step_no = 0;
};

void *ov2h_feloop_create(void *ctx, const ov2h_feloop_cfg *cfg)
{
    if (!ctx || !cfg || cfg->L <= 0) return nullptr;
    FeLoopNative *f = new FeLoopNative();
    f->ctx = (ov2_ctx *)ctx; f->c = *cfg;
    const int L = cfg->L;
    f->left.assign(cfg->left, cfg->left + L); f->right.assign(cfg->right, cfg->right + L);
    f->kps.assign(cfg->kps, cfg->kps + L); f->pri.assign(cfg->pri, cfg->pri + L); f->st_pri.assign(cfg->st_pri, cfg->st_pri + L);
    f->has.assign(cfg->has, cfg->has + L); f->st_has.assign(cfg->st_has, cfg->st_has + L);
    return f;
}

// runs `steps` frames; a keyframe every kf_every-th frame submits a job to each of the n_pipes Estimator pipelines.
// returns 0 or the first failing status; *n_kf = keyframes in this call
int ov2h_feloop_run(void *h, int steps, int kf_every, void *const *pipes, int n_pipes, int *n_kf)
{
    FeLoopNative *f = (FeLoopNative *)h;
    const ov2h_feloop_cfg &c = f->c;
    ov2_ctx *ctx = f->ctx;
    int kfs = 0;
    for (int it = 0; it < steps; ++it) {
        const int p = (int)(f->step_no % c.L);
        ov2_pyr *cur = nullptr;
        ov2_status s = ov2_pyramid_build_images(ctx, (const ov2_images *)f->left[p], c.win, c.nlvl, c.use_clahe, c.clahe_clip, c.tiles_x,
                                                c.tiles_y, &cur);
        if (s != OV2_OK) return (int)s;
        if (f->prev) {
            s = ov2_klt_tracking_frame_dev(ctx, f->prev, cur, c.win, c.nlvl, c.max_iter, c.eps, c.err_th, c.fb_th, c.n, f->kps[p], f->pri[p],
                                           f->has[p], c.img_idx, c.out_xy, c.out_st, c.p3p, nullptr);
            ov2_pyr_release(f->prev);
            f->prev = nullptr;
            if (s != OV2_OK) { ov2_pyr_release(cur); return (int)s; }
            if (c.pnp) {
                if ((s = ov2_memcpy_d2d(ctx, c.pnp_T, c.pnp_T0, (size_t)c.pnp_T_bytes)) != OV2_OK ||
                    (s = ov2_pnp_solve_batch_dev(ctx, c.B, c.pnp_off, c.pnp_unpx, c.pnp_wpts, nullptr, c.pnp_K, c.pnp_T, 5, 5.9915f, 1, 1,
                                                 c.pnp_outl, c.pnp_rem, c.pnp_ok, nullptr)) != OV2_OK) { ov2_pyr_release(cur); return (int)s; }
            }
        }
        f->prev = cur;
        const bool is_kf = kf_every > 0 && (f->step_no % kf_every) == 0;
        if (is_kf) {
            ov2_pyr *rp = nullptr;
            if ((s = ov2_pyramid_build_images(ctx, (const ov2_images *)f->right[p], c.win, c.nlvl, c.use_clahe, c.clahe_clip, c.tiles_x,
                                              c.tiles_y, &rp)) != OV2_OK) return (int)s;
            s = ov2_stereo_matching_dev(ctx, cur, rp, c.win, c.nlvl, c.max_iter, c.eps, c.err_th, c.fb_th, c.n, f->kps[p], f->st_pri[p],
                                        f->st_has[p], c.img_idx, nullptr, 1, nullptr, nullptr, c.out_xy, c.out_st, nullptr);
            ov2_pyr_release(rp);
            if (s != OV2_OK) return (int)s;
            if (c.detect && (s = ov2_detect_grid_batch_dev(ctx, cur, c.det_cell, OV2_DETECT_MINEIG, c.det_thresh, c.det_ncur, c.det_cur, c.det_img,
                                                           nullptr, nullptr, 1, c.det_nout, c.det_out, c.det_cap)) != OV2_OK) return (int)s;
            for (int k = 0; k < n_pipes; ++k) ov2h_ba_pipeline_submit_all(pipes[k]);
            ++kfs;
        }
        ++f->step_no;
    }
    if (n_kf) *n_kf = kfs;
    return 0;
}

void ov2h_feloop_destroy(void *h)
{
    FeLoopNative *f = (FeLoopNative *)h;
    if (!f) return;
    if (f->prev) ov2_pyr_release(f->prev);
    delete f;
}

}  // extern "C"
