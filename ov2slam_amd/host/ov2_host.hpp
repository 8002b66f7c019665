// ov2_host.hpp -- C++ host side above the C ABI (include/ov2slam_hip.h): the reference's data model and call surface
// for the hot path, without OpenCV / Eigen / Sophus / Ceres, so that the HIP path drops into a SlamManager-like
// pipeline.  Names, members and control flow mirror the reference (file:line in /root/reference):
//   Keypoint            include/frame.hpp:46-76          Frame        include/frame.hpp:78-237, src/frame.cpp
//   MapPoint            include/map_point.hpp:37-97      MapManager   include/map_manager.hpp:41-129 (subset)
//   FeatureTracker      include/feature_tracker.hpp:32-56, src/feature_tracker.cpp:35-137
//   VisualFrontEnd      src/visual_front_end.cpp:132-275 (kltTracking), :657-830 (computePose), :1143-1177 (preprocessImage)
//   MultiViewGeometry::ceresPnP  src/multi_view_geometry.cpp:492-586
//   MapManager::stereoMatching  src/map_manager.cpp:367-611 (KLT part + epipolar gate for rectified / row check)
//   Optimizer::localBA  src/optimizer.cpp:34-897         Estimator::applyLocalBA  src/estimator.cpp:67-98
// All arithmetic of the path runs behind the ABI on the GPU; this file is graph walking and bookkeeping.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/ov2slam_hip.h"

namespace ov2 {

struct Point2f {
    float x = 0.f, y = 0.f;
};

struct Vec3 {
    double x = 0, y = 0, z = 0;
};

// rigid transform stored like the reference's PoseParametersBlock: [tx ty tz qx qy qz qw]
struct SE3 {
    std::array<double, 7> v{{0, 0, 0, 0, 0, 0, 1}};
    void rotation(double R[9]) const;
    Vec3 operator*(const Vec3 &p) const;
    SE3 inverse() const;
    SE3 operator*(const SE3 &o) const;
    static SE3 fromRt(const double R[9], const double t[3]);
};

struct CameraCalibration {   // the part of src/camera_calibration.cpp that the path needs
    enum Model { Pinhole, Fisheye };
    double fx_ = 1, fy_ = 1, cx_ = 0, cy_ = 0;
    int img_w_ = 0, img_h_ = 0;
    SE3 Tc0ci_;               // extrinsic: this camera in the left-camera frame (getExtrinsic(), T_left_right)
    Model model_ = Pinhole;
    std::vector<double> D_;   // Dcv_: empty = no distortion; Pinhole: k1 k2 p1 p2 [k3]; Fisheye: k1 k2 k3 k4
    Vec3 projectCamToImage(const Vec3 &pc) const
    {   // src/camera_calibration.cpp:243-252: invz first, then fx * x + cx
        const double invz = 1. / pc.z, x = pc.x * invz, y = pc.y * invz;
        return {fx_ * x + cx_, fy_ * y + cy_, pc.z};
    }
    // :313-332 -> cv::undistortPoints(px, K, D, noArray(), K) (five fixed-point sweeps) / cv::fisheye::undistortPoints
    // (Newton on theta, at most ten steps); OpenCV is not vendored by the reference: restated from its published algorithm
    Point2f undistortImagePoint(const Point2f &pt) const;
    // :254-282 -> cv::projectPoints with zero rvec / tvec (radial-tangential) / cv::fisheye::distortPoints
    Point2f projectCamToImageDist(const Vec3 &pc) const;
    bool fillCamModel(ov2_cam_model *m) const   // the lens model as the kernels take it; false = no distortion
    {
        *m = ov2_cam_model();
        m->K[0] = fx_; m->K[1] = fy_; m->K[2] = cx_; m->K[3] = cy_;
        m->model = D_.empty() ? 0 : (model_ == Fisheye ? 2 : 1);
        m->n_coeffs = (int32_t)(D_.size() < 5 ? D_.size() : 5);
        for (int k = 0; k < m->n_coeffs; ++k) m->D[k] = D_[k];
        return m->model != 0;
    }
};

typedef std::array<uint8_t, 32> Desc;   // one BRIEF-32 descriptor (the 1 x 32 CV_8U cv::Mat of the reference)

struct Keypoint {   // include/frame.hpp:46-76
    int lmid_ = -1;
    Desc desc_{};
    bool has_desc_ = false;   // !desc_.empty()
    Point2f px_, unpx_;
    int scale_ = 0;
    float angle_ = -1.f;
    bool is3d_ = false;
    bool is_stereo_ = false;
    Point2f rpx_, runpx_;
    bool is_retracked_ = false;
    Vec3 bv_, rbv_;           // bearing vectors (left / right), include/frame.hpp:60-66
};

class Frame {   // include/frame.hpp:78-237 (members the path touches)
public:
    int id_ = -1, kfid_ = 0;
    double img_time_ = 0.;
    std::unordered_map<int, Keypoint> mapkps_;
    size_t nbkps_ = 0, nb2dkps_ = 0, nb3dkps_ = 0, nb_stereo_kps_ = 0;
    SE3 Twc_, Tcw_;
    std::shared_ptr<CameraCalibration> pcalib_leftcam_, pcalib_rightcam_;
    std::map<int, int> map_covkfs_;
    std::unordered_set<int> set_local_mapids_;   // include/frame.hpp:209: map points of the covisible keyframes this frame does not observe

    // keypoint grid (include/frame.hpp:214-221, src/frame.cpp:40-45): cells of ncellsize_ px, ids per cell in insertion
    // order.  initGrid() plays the part of the reference constructors; a Frame without it keeps no grid.
    size_t ncellsize_ = 0, nbwcells_ = 0, nbhcells_ = 0, noccupcells_ = 0;
    std::vector<std::vector<int>> vgridkps_;
    double Frl_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // F_rl = K_r^-T [t]x R K_l^-1 (src/frame.cpp:53-62), row-major
    void initGrid(size_t ncellsize);                  // + Frl_ when a right camera is set
    int getKeypointCellIdx(const Point2f &pt) const;
    std::vector<Keypoint> getKeypoints() const;
    std::vector<Keypoint> getSurroundingKeypoints(const Keypoint &kp) const;   // src/frame.cpp:594-622
    void computeKeypoint(const Point2f &pt, Keypoint &kp) const;              // :246-254 (no distortion: unpx = px)
    void updateKeypointStereo(int lmid, const Point2f &pt);                    // :405-433
    Vec3 projWorldToCam(const Vec3 &wpt) const { return Tcw_ * wpt; }
    Point2f projWorldToRightImageDist(const Vec3 &wpt) const;                  // :835-838
    Point2f projCamToRightImageDist(const Vec3 &pt) const;                     // :796-799
    bool isInRightImage(const Point2f &pt) const;                              // :845-848
    SE3 getTwc() const { return Twc_; }
    SE3 getTcw() const { return Tcw_; }
    void setTwc(const SE3 &Twc) { Twc_ = Twc; Tcw_ = Twc.inverse(); }
    std::vector<Keypoint> getKeypoints3d() const;
    Keypoint getKeypointById(int lmid) const;   // returns Keypoint with lmid_ = -1 if absent (src/frame.cpp)
    std::map<int, int> getCovisibleKfMap() const { return map_covkfs_; }
    void removeCovisibleKf(int kfid) { map_covkfs_.erase(kfid); }
    void addCovisibleKf(int kfid) { if (kfid != kfid_) map_covkfs_[kfid] += 1; }   // src/frame.cpp:664-677
    bool isObservingKp(int lmid) const { return mapkps_.count(lmid) != 0; }       // :502-506
    void updateKeypointDesc(int lmid, const Desc &d);                              // :356-366
    bool updateKeypointId(int prevlmid, int newlmid, bool is3d);                   // :380-402
    void decreaseCovisibleKf(int kfid)   // src/frame.cpp:689-705
    {
        if (kfid == kfid_) return;
        auto it = map_covkfs_.find(kfid);
        if (it != map_covkfs_.end() && it->second != 0 && --it->second == 0) map_covkfs_.erase(it);
    }
    void addKeypoint(const Keypoint &kp);
    void updateKeypoint(int lmid, const Point2f &pt);   // src/frame.cpp: px_ / unpx_ update (pinhole: unpx = px)
    void removeKeypointById(int lmid);
    void removeStereoKeypointById(int lmid);
    void turnKeypoint3d(int lmid);
    bool isInImage(const Point2f &pt) const;
    Point2f projWorldToImage(const Vec3 &wpt) const;
};

class MapPoint {   // include/map_point.hpp:37-97
public:
    MapPoint(int lmid, int kfid, bool bobs = true) : lmid_(lmid), isobs_(bobs), kfid_(kfid) { set_kfids_.insert(kfid); }
    MapPoint(int lmid, int kfid, const Desc &desc, bool bobs = true) : lmid_(lmid), isobs_(bobs), kfid_(kfid)   // src/map_point.cpp:40-51
    {
        set_kfids_.insert(kfid);
        map_kf_desc_.emplace(kfid, desc); map_desc_dist_.emplace(kfid, 0.f);
        desc_ = desc; has_desc_ = true;
    }
    // descriptors (include/map_point.hpp:84-88): one per observing keyframe, the sum of its Hamming distances to the others,
    // and the representative one (desc_).  Same containers as the reference: their iteration order decides ties.
    Desc desc_{};
    bool has_desc_ = false;
    std::unordered_map<int, Desc> map_kf_desc_;
    std::unordered_map<int, float> map_desc_dist_;
    void addDesc(int kfid, const Desc &d);   // src/map_point.cpp:162-211
    int lmid_;
    bool isobs_;
    bool is3d_ = false;
    std::set<int> set_kfids_;
    Vec3 ptxyz_;
    int kfid_;
    double invdepth_ = -1.;
    void setPoint(const Vec3 &p, double kfanch_invdepth = -1.) { ptxyz_ = p; is3d_ = true; if (kfanch_invdepth >= 0.) invdepth_ = kfanch_invdepth; }
    Vec3 getPoint() const { return ptxyz_; }
    std::set<int> getKfObsSet() const { return set_kfids_; }
    void addKfObs(int kfid) { set_kfids_.insert(kfid); }
    void removeKfObs(int kfid);   // src/map_point.cpp:106-160: the anchor moves to the oldest observer left, the descriptor of kfid goes
    bool isBad();   // src/map_point.cpp:215-234
};

class MapManager {   // subset of include/map_manager.hpp:41-129 used by localBA / kltTracking
public:
    std::shared_ptr<Frame> pcurframe_;
    std::unordered_map<int, std::shared_ptr<Frame>> map_pkfs_;
    std::unordered_map<int, std::shared_ptr<MapPoint>> map_plms_;
    std::shared_ptr<Frame> getKeyframe(int kfid) const;
    std::shared_ptr<MapPoint> getMapPoint(int lmid) const;
    void updateMapPoint(int lmid, const Vec3 &wpt, double kfanch_invdepth = -1.);   // src/map_manager.cpp
    void removeMapPointObs(int lmid, int kfid);
    void removeMapPoint(int lmid);
    void removeObsFromCurFrameById(int lmid);
    void updateFrameCovisibility(Frame &frame);   // src/map_manager.cpp:117-192: co-observation counts + the frame's local map
    void mergeMapPoints(int prevlmid, int newlmid);   // :801-882
    void setMapPointObs(int lmid);                    // :1053-1090 (the isobs_ flag; the point cloud colour is not mirrored)
    // src/map_manager.cpp:367-611, statement by statement: priors (3D point reprojected into the right camera :398-413;
    // rectified rigs: getLineMinSAD on the coarsest level :419-436 = ov2_line_min_sad; otherwise the inverse-distance
    // weighted depth of the 3D neighbours :438-483), 3D keypoints whose map point is gone lose their observation (:414),
    // then ov2_stereo_matching: 2-level KLT on the priors, failures re-queued with the updated prior, full pyramid for the
    // rest, epipolar gate (row check + snap | Sampson with Frame::Frl_) -> Frame::updateKeypointStereo.  Cameras
    // without distortion (Dcv_.empty(): undistortImagePoint / projectCamToImageDist are the pinhole maps).
    ov2_status stereoMatching(Frame &frame, const struct Pyramid &vleftpyr, const struct Pyramid &vrightpyr,
                              const class FeatureTracker &tracker, const struct SlamParams &st);

    // ---- flat device mirror (include/ov2slam_hip.h "ov2_map"): the hash maps above stay the host's source of truth,
    // every mutation that the local-BA set-up can see is queued here and pushed in batches (flushDevice) before the
    // next device set-up.  attachDevice walks the whole map once.
    ~MapManager();
    ov2_status attachDevice(ov2_ctx *ctx, int max_kf, int max_lm, int max_obs);
    ov2_status addKeyframeToDevice(const Frame &kf);        // MapManager::addKeyframe hook
    ov2_status flushDevice();
    void touchMapPoint(int lmid) { if (dev_) dev_lm_dirty_.push_back(lmid); }
    void touchPose(int kfid) { if (dev_) dev_pose_dirty_.push_back(kfid); }
    void touchStereoOff(int kfid, int lmid) { if (dev_) { dev_st_kf_.push_back(kfid); dev_st_lm_.push_back(lmid); } }
    ov2_map *dev_ = nullptr;
    std::vector<int32_t> dev_lm_dirty_, dev_pose_dirty_, dev_rm_kf_, dev_rm_lm_, dev_st_kf_, dev_st_lm_;
    std::vector<std::pair<int, int>> dev_add_obs_;   // (kfid, lmid): observations a merge gave to keyframes the mirror already holds
    std::set<int> dev_kfs_;                          // keyframes pushed to the mirror
};

struct SlamParams {   // the subset of include/slam_params.hpp the path reads (YAML keys of the same name)
    bool stereo_ = true, mono_ = false, buse_inv_depth_ = true, apply_l2_after_robust_ = true, klt_use_prior_ = true;
    float robust_mono_th_ = 5.9915f;
    int nmin_covscore_ = 25;
    int nklt_win_size_ = 9, nklt_pyr_lvl_ = 3, nmax_iter_ = 30;
    float fmax_px_precision_ = 0.01f, fmax_fbklt_dist_ = 0.5f, nklt_err_ = 30.f;
    bool use_clahe_ = true;
    float fclahe_val_ = 3.f;
    bool blocalba_is_on_ = false, bforce_realtime_ = true;
    bool dop3p_ = false;
    bool bdo_stereo_rect_ = false;   // parameters_files/*/euroc_stereo.yaml: 0
    // keyframe creation / selection (src/slam_params.cpp:95-125, YAML keys of the same name)
    int nmaxdist_ = 35, nbmaxkps_ = 308;            // nbmaxkps_ = ceil(w / nmaxdist) * ceil(h / nmaxdist) (:107-110)
    bool use_fast_ = false, use_singlescale_detector_ = true, use_brief_ = false, doepipolar_ = false;
    bool bdo_track_localmap_ = true;                               // src/slam_params.cpp:133
    float fmax_desc_dist_ = 0.2f, fmax_proj_pxdist_ = 2.f;         // :135-136
    double dmaxquality_ = 0.001;
    int nfast_th_ = 10;
    float finit_parallax_ = 20.f, fmax_reproj_err_ = 3.f;
};

struct Vec2 {
    double x = 0, y = 0;
};

class MultiViewGeometry {   // include/multi_view_geometry.hpp:104, src/multi_view_geometry.cpp:492-586
public:
    // motion-only BA on the GPU (ov2_pnp_solve_batch, B = 1); same arguments and return value as the reference
    static bool ceresPnP(ov2_ctx *ctx, const std::vector<Vec2> &vunkps, const std::vector<Vec3> &vwpts,
                         const std::vector<int> &vscales, SE3 &Twc, int nmaxiter, float chi2th, bool buse_robust,
                         bool bapply_l2_after_robust, float fx, float fy, float cx, float cy,
                         std::vector<int> &voutliersidx);
};

// owning handle of an ov2_pyr (the std::vector<cv::Mat> pyramid of the reference)
struct Pyramid {
    ov2_pyr *h = nullptr;
    Pyramid() = default;
    explicit Pyramid(ov2_pyr *p) : h(p) {}
    Pyramid(const Pyramid &o) : h(o.h) { if (h) ov2_pyr_retain(h); }
    Pyramid &operator=(const Pyramid &o) { if (o.h) ov2_pyr_retain(o.h); if (h) ov2_pyr_release(h); h = o.h; return *this; }
    ~Pyramid() { if (h) ov2_pyr_release(h); }
    bool empty() const { return h == nullptr; }
    void swap(Pyramid &o) { std::swap(h, o.h); }
};

class FeatureTracker {   // include/feature_tracker.hpp:32-56
public:
    FeatureTracker(ov2_ctx *ctx, int nmax_iter, float fmax_px_precision) : ctx_(ctx), nmax_iter_(nmax_iter), fmax_px_precision_(fmax_px_precision) {}
    // src/feature_tracker.cpp:35-137; returns the ov2_status of the call (the reference has no error channel)
    ov2_status fbKltTracking(const Pyramid &vprevpyr, const Pyramid &vcurpyr, int nwinsize, int nbpyrlvl, float ferr,
                             float fmax_fbklt_dist, std::vector<Point2f> &vkps, std::vector<Point2f> &vpriorkps,
                             std::vector<bool> &vkpstatus) const;
    bool inBorder(const Point2f &pt, int cols, int rows) const;   // :216-221
    ov2_ctx *ctx_;
    int nmax_iter_;
    float fmax_px_precision_;
};

class FeatureExtractor {   // include/feature_extractor.hpp:37-55 (grid detectors + their adaptive thresholds)
public:
    FeatureExtractor(ov2_ctx *ctx, size_t nmaxpts, size_t nmaxdist, double dmaxquality, int nfast_th)
        : ctx_(ctx), nmaxpts_(nmaxpts), nmaxdist_(nmaxdist), dmaxquality_(dmaxquality), nfast_th_(nfast_th) {}
    // both run on level 0 of `pyr` (the CLAHE'd frame the reference passes as `im`); roi = {x,y,w,h}
    std::vector<Point2f> detectSingleScale(const Pyramid &pyr, int ncellsize, const std::vector<Point2f> &vcurkps,
                                           const int roi[4]);   // src/feature_extractor.cpp:288-440
    std::vector<Point2f> detectGridFAST(const Pyramid &pyr, int ncellsize, const std::vector<Point2f> &vcurkps,
                                        const int roi[4]);      // :443-570
    ov2_status last_status_ = OV2_OK;
    ov2_ctx *ctx_;
    size_t nmaxpts_, nmaxdist_;
    double dmaxquality_;
    int nfast_th_;
private:
    std::vector<Point2f> detect(const Pyramid &pyr, int ncellsize, int mode, const std::vector<Point2f> &vcurkps, const int roi[4]);
};

class VisualFrontEnd {   // src/visual_front_end.cpp (preprocessImage + kltTracking)
public:
    VisualFrontEnd(ov2_ctx *ctx, std::shared_ptr<SlamParams> pstate, std::shared_ptr<Frame> pframe,
                   std::shared_ptr<MapManager> pmap, std::shared_ptr<FeatureTracker> ptracker)
        : ctx_(ctx), pslamstate_(pstate), pcurframe_(pframe), pmap_(pmap), ptracker_(ptracker) {}
    ov2_status preprocessImage(const uint8_t *img_raw, int w, int h, int stride);   // :1143-1177
    ov2_status kltTracking();                                                        // :132-275
    // :657-830 without the P3P-RANSAC branch (OpenGV, out of scope): when P3P is required (bp3preq_ or dop3p_) the
    // call returns OV2_ERR_UNSUPPORTED and leaves the frame untouched.
    ov2_status computePose();
    bool bp3preq_ = false;
    Pyramid prev_pyr_, cur_pyr_;
    ov2_ctx *ctx_;
    std::shared_ptr<SlamParams> pslamstate_;
    std::shared_ptr<Frame> pcurframe_;
    std::shared_ptr<MapManager> pmap_;
    std::shared_ptr<FeatureTracker> ptracker_;
};

// flat problem assembled by Optimizer::localBA's set-up stage, kept for the update stage
struct LocalBAProblem {
    std::vector<double> pose, lm, lm_anchor_uv, res_uv, res_sigma;
    std::vector<uint8_t> pose_const, res_type;
    std::vector<int32_t> lm_anchor_pose, res_pose, res_lm;
    std::vector<int> pose_kfid, lm_lmid;                 // block index -> reference ids
    std::unordered_map<int, int> kfid_to_pose, lmid_to_lm;
    std::unordered_set<int> set_cstkfids, set_badlmids;
    std::unordered_map<int, std::shared_ptr<Frame>> map_local_pkfs;
    std::unordered_map<int, std::shared_ptr<MapPoint>> map_local_plms;
    size_t nbmono = 0, nbstereo = 0;
    bool aborted = false;                                // early return of :61-63
    ov2_local_ba_setup dev_view{};   // set by setupLocalBADevice: the same arrays where the set-up kernels left them (has_dev)
    bool has_dev = false;
    ov2_ba_problem view(const SlamParams &st, const Frame &newframe);
};

class Optimizer {   // include/optimizer.hpp:42, src/optimizer.cpp:34-897
public:
    Optimizer(ov2_ctx *ctx, std::shared_ptr<SlamParams> pstate, std::shared_ptr<MapManager> pmap)
        : ctx_(ctx), pslamstate_(pstate), pmap_(pmap) {}
    ov2_status localBA(Frame &newframe, const bool buse_robust_cost);
    // src/optimizer.cpp:2594-2781: points-only refinement (all poses constant, XYZ, Huber, 10 iterations) through ov2_ba_solve
    ov2_status structureOnlyBA(const std::vector<int> &vlm2optids);
    // src/optimizer.cpp:1674-2332 (Mapper::runFullBA, src/mapper.cpp:780): every keyframe (the first nmincstkfs constant), every
    // 3D landmark with >= 3 observers, 100 iterations, flags / L2 refinement / update as in localBA -- through ov2_ba_solve
    ov2_status fullBA(const bool buse_robust_cost);
    // src/optimizer.cpp:900-1670 (LoopCloser, src/loop_closer.cpp:368): keyframes inikfid .. nkfid (the first nmincstkfs
    // constant), one robust solve of 5 iterations (function_tolerance 1e-4), then the corrections are propagated to the
    // younger keyframes, their own landmarks and the current frame
    ov2_status looseBA(int inikfid, const int nkfid, const bool buse_robust_cost);
    // src/optimizer.cpp:2346-2592 (LoopCloser, src/loop_closer.cpp:320): keyframes kfloop_id .. newframe joined by their
    // relative poses + the loop edge kfloop_id -> newframe from the P3P pose newTwc, loop keyframe constant; solved by
    // ov2_pose_graph_solve (10 iterations, 1e-4); rejected when the optimised new pose lands > 0.3 m from newTwc (stereo);
    // then keyframes, the landmarks they anchor, the younger keyframes and the current frame are moved
    bool localPoseGraph(Frame &newframe, int kfloop_id, const SE3 &newTwc, ov2_status *status = nullptr);
    // :2783-2870 (SlamManager::writeFullTrajectoryLC, src/ov2slam.cpp:702): every frame a pose (keyframes constant),
    // consecutive frames joined by vTpc; vTwc receives the optimised trajectory (the reference writes it to a file)
    bool fullPoseGraph(std::vector<SE3> &vTwc, const std::vector<SE3> &vTpc, const std::vector<bool> &viskf, ov2_status *status = nullptr);
    ov2_pg_result last_pg_{};
    // problem assembly shared by the three: keyframes kf_lo .. kf_hi, observers above kf_obs_max ignored, landmarks with
    // fewer than min_obs observers set aside as bad
    void setupRangeBA(int kf_lo, int kf_hi, int kf_obs_max, size_t min_obs, LocalBAProblem &pb);
    // the three stages, exposed for tests
    void setupLocalBA(Frame &newframe, LocalBAProblem &pb);                                   // :43-430
    // the same stage from the device map mirror (ov2_map_local_ba_setup): linear scans instead of the hash-map walk
    ov2_status setupLocalBADevice(Frame &newframe, LocalBAProblem &pb);
    void updateAfterLocalBA(Frame &newframe, LocalBAProblem &pb, const ov2_ba_result &res, bool cur_frame_obs = true);   // :741-882
    bool stopLocalBA() const { return bstop_localba_; }
    void signalStopLocalBA() { bstop_localba_ = true; }
    ov2_ba_result last_result_{};
    void *dev_out_ = nullptr;        // device scratch of the per-residual outputs when the solve runs on the set-up's device arrays
    size_t dev_out_cap_ = 0;
    ~Optimizer() { if (dev_out_) ov2_dev_free(ctx_, dev_out_); }
    ov2_ctx *ctx_;
    std::shared_ptr<SlamParams> pslamstate_;
    std::shared_ptr<MapManager> pmap_;
    bool bstop_localba_ = false;
};

class Estimator {   // src/estimator.cpp:67-98
public:
    Estimator(std::shared_ptr<SlamParams> pstate, std::shared_ptr<MapManager> pmap, std::shared_ptr<Optimizer> popt)
        : pslamstate_(pstate), pmap_(pmap), poptimizer_(popt) {}
    ov2_status applyLocalBA();
    std::shared_ptr<Frame> pnewkf_;
    std::shared_ptr<SlamParams> pslamstate_;
    std::shared_ptr<MapManager> pmap_;
    std::shared_ptr<Optimizer> poptimizer_;
};

}  // namespace ov2
