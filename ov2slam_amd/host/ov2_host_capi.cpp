// ov2_host_capi.cpp -- flat C hooks around the C++ host mirror so that pytest can build a Frame/MapPoint graph,
// run Optimizer::setupLocalBA (CPU only) or the whole Estimator::applyLocalBA (GPU) and read the map back.
// Test/bring-up surface only; a real integration uses the C++ classes of ov2_host.hpp directly.
#include <cstring>

#include "ov2_host.hpp"

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

}  // extern "C"
