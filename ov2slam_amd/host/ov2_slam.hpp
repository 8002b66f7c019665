// ov2_slam.hpp -- the per-frame / per-keyframe drivers of the reference on top of the host mirror (ov2_host.hpp), so that
// a closed loop runs through C++ and the C ABI without Python in the path:
//   MotionModel                      include/visual_front_end.hpp:38-90
//   VisualFrontEnd::visualTracking   src/visual_front_end.cpp:40-62        trackMono :66-130      checkNewKfReq :985-1064
//                   computeParallax  :1069-1142
//   MapManager::createKeyframe       src/map_manager.cpp:43-60  prepareFrame :64-115  extractKeypoints :286-340
//               addKeypointsToFrame  :196-211   addKeyframe :621-634   addMapPoint :636-659
//   Mapper::run (one keyframe)       src/mapper.cpp:38-189      triangulateStereo :346-461
//   SlamManager::run (one image)     src/ov2slam.cpp:152-205    Estimator::applyLocalBA  src/estimator.cpp:67-98
// The reference runs front-end, mapper and estimator on three threads; here one call processes one stereo frame to the
// end (keyframe work included), i.e. the reference with bforce_realtime = 0 and an idle back-end: deterministic.
// Out of scope and refused loudly where reached: P3P / 5-point RANSAC (OpenGV), loop closing, BRIEF map matching.
#pragma once
#include "ov2_host.hpp"

namespace ov2 {

// Knobs that replace three heuristics of the reference by the fixed stand-ins of ov2slam_amd/slam_loop.py, so that this
// driver can be compared pose by pose with that loop (and through it with the CPU oracle); all zero = the reference.
struct LoopPolicy {
    int kf_every = 0;             // > 0: a keyframe every kf_every-th frame instead of checkNewKfReq
    int ba_window = 0;            // > 0: local BA over the last ba_window keyframes (the oldest ba_fixed constant) instead of the covisibility walk
    int ba_fixed = 2;
    bool compose_motion = false;  // prediction = Twc * (Twc_prev^-1 * Twc) instead of exp(log(.) / dt * dt) (equal up to rounding)
    bool midpoint_stereo = false; // stereo triangulation by the mid-point method also for rectified rigs; a failed triangulation
                                  // keeps the stereo observation (the reference demotes it, src/mapper.cpp:412,428,440)
    bool pose_from_kf = false;    // after a local BA the current frame takes the refined pose of its keyframe
};

class MotionModel {   // include/visual_front_end.hpp:38-90: constant velocity in se3
public:
    void applyMotionModel(SE3 &Twc, double time);
    void updateMotionModel(const SE3 &Twc, double time);
    void reset() { prev_time_ = -1.; for (double &v : log_relT_) v = 0.; }
    double prev_time_ = -1.;
    SE3 prevTwc_;
    double log_relT_[6] = {0, 0, 0, 0, 0, 0};   // [upsilon, omega] per second
};

void se3_log(const SE3 &T, double out[6]);   // Sophus::SE3::log  (se3.hpp), tangent order [upsilon, omega]
SE3 se3_exp(const double a[6]);              // Sophus::SE3::exp  (se3.hpp:763-784)

struct Keyframe {   // include/mapper.hpp:39-85: what the front-end hands to the mapper
    int kfid_ = -1;
    Pyramid vpyr_imleft_;
    const uint8_t *imrightraw_ = nullptr;
    int w = 0, h = 0, stride = 0;
};

struct SlamStats {   // per frame, for tests / profiles
    int frame = 0, tracked = 0, n3d = 0, is_kf = 0, n_new = 0, n_stereo = 0, n_lm3d = 0;
    int ba_done = 0, ba_res = 0, ba_it_robust = 0, ba_it_l2 = 0, ba_outliers = 0;
    int n_described = 0, n_local = 0, n_matched = 0;   // keyframe: keypoints with a descriptor, local map points offered, merges
    double ba_cost0 = 0., ba_cost1 = 0.;
};

class SlamManager {   // src/ov2slam.cpp (the stereo branch of run())
public:
    SlamManager(ov2_ctx *ctx, std::shared_ptr<SlamParams> pstate, std::shared_ptr<CameraCalibration> cl,
                std::shared_ptr<CameraCalibration> cr, const LoopPolicy &policy);
    // one stereo frame through visualTracking and, when it asks for a keyframe, Mapper::run + Estimator::applyLocalBA
    ov2_status addNewStereoImages(double time, const uint8_t *im0, const uint8_t *im1, int w, int h, int stride);
    SE3 pose() const { return pcurframe_->getTwc(); }
    // the 256 test pairs of BRIEF-32 (y1, x1, y2, x2 per test, int8: opencv_contrib's generated_32.i, absent from the reference
    // tree, so the caller supplies it) -- with use_brief_ the keyframe path describes its keypoints and runs
    // Mapper::matchingToLocalMap (src/mapper.cpp:469-554)
    void setBriefPattern(const int8_t *pattern256x4) { brief_pattern_.assign(pattern256x4, pattern256x4 + 1024); }
    std::vector<int8_t> brief_pattern_;

    ov2_ctx *ctx_;
    std::shared_ptr<SlamParams> pslamstate_;
    std::shared_ptr<Frame> pcurframe_;
    std::shared_ptr<MapManager> pmap_;
    std::shared_ptr<FeatureTracker> ptracker_;
    std::shared_ptr<FeatureExtractor> pfeatextract_;
    std::shared_ptr<VisualFrontEnd> pvisualfrontend_;
    std::shared_ptr<Optimizer> poptimizer_;
    std::shared_ptr<Estimator> pestimator_;
    LoopPolicy policy_;
    MotionModel motion_model_;
    int frame_id_ = -1;
    SlamStats last_;
    std::vector<SlamStats> stats_;
    std::vector<SE3> traj_;

private:
    bool visualTracking(const uint8_t *iml, int w, int h, int stride, double time, ov2_status *st);   // :40-62
    bool trackMono(const uint8_t *im, int w, int h, int stride, double time, ov2_status *st);          // :66-130
    bool checkNewKfReq();                                                                              // :985-1064
    float computeParallax(int kfid, bool do_unrot, bool bmedian, bool b2donly);                        // :1069-1142
    ov2_status createKeyframe();                                                                       // src/map_manager.cpp:43-60
    void prepareFrame();
    ov2_status extractKeypoints();
    void addKeypointsToFrame(const std::vector<Point2f> &vpts, Frame &frame);
    void addKeypointsToFrame(const std::vector<Point2f> &vpts, const std::vector<Desc> &vdescs, const std::vector<uint8_t> &valid, Frame &frame);
    ov2_status describeBRIEF(const std::vector<Point2f> &vpts, std::vector<Desc> &vdescs, std::vector<uint8_t> &valid);   // src/feature_extractor.cpp:224-285
    ov2_status describeKeypoints(const std::vector<Keypoint> &vkps, const std::vector<Point2f> &vpts);                   // src/map_manager.cpp:343-362
    ov2_status matchingToLocalMap(Frame &frame);                                                        // src/mapper.cpp:469-554
    ov2_status matchToMap(const Frame &frame, float fmaxprojerr, float fdistratio, std::unordered_set<int> &set_local_lmids,
                          std::map<int, int> &map_previd_newid);                                        // :576-774 -> ov2_match_to_map
    Pyramid raw_pyr_;        // level 0 of the raw left image (describeBRIEF works on imraw, src/map_manager.cpp:300, 325)
    const uint8_t *imraw_ = nullptr; int imw_ = 0, imh_ = 0, imstride_ = 0;
    void addKeyframe();
    ov2_status mapperRun(const Keyframe &kf);                                                          // src/mapper.cpp:38-189
    ov2_status triangulateStereo(Frame &frame);                                                        // :346-461
    ov2_status fixedWindowBA();                                                                        // LoopPolicy::ba_window
    int nkfid_ = 0, nlmid_ = 0;
    SE3 Twc_prev_;           // compose_motion: the pose of the frame before the last
    bool have_prev_ = false;
};

}  // namespace ov2
