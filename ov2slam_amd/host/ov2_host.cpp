// ov2_host.cpp -- see ov2_host.hpp.  Graph walking / bookkeeping of the reference's hot-path callers, re-expressed on
// the flat C ABI.  Reference line numbers (in /root/reference) are cited at each block.
#include "ov2_host.hpp"

#include <climits>
#include <map>
#include <set>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ov2 {

// ---------------------------------------------------------------------------------------------- SE3
void SE3::rotation(double R[9]) const
{
    double x = v[3], y = v[4], z = v[5], w = v[6];
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

Vec3 SE3::operator*(const Vec3 &p) const
{
    double R[9];
    rotation(R);
    return {R[0] * p.x + R[1] * p.y + R[2] * p.z + v[0], R[3] * p.x + R[4] * p.y + R[5] * p.z + v[1],
            R[6] * p.x + R[7] * p.y + R[8] * p.z + v[2]};
}

static void rot_to_quat(const double R[9], double q[4])
{
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        const double s = std::sqrt(t + 1.0) * 2;
        q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
        const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
}

SE3 SE3::fromRt(const double R[9], const double t[3])
{
    SE3 T;
    double q[4];
    rot_to_quat(R, q);
    T.v = {t[0], t[1], t[2], q[0], q[1], q[2], q[3]};
    return T;
}

SE3 SE3::inverse() const
{
    double R[9], Rt[9];
    rotation(R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
    const double t[3] = {-(Rt[0] * v[0] + Rt[1] * v[1] + Rt[2] * v[2]), -(Rt[3] * v[0] + Rt[4] * v[1] + Rt[5] * v[2]),
                         -(Rt[6] * v[0] + Rt[7] * v[1] + Rt[8] * v[2])};
    SE3 T;
    const double n = std::sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5] + v[6] * v[6]);
    T.v = {t[0], t[1], t[2], -v[3] / n, -v[4] / n, -v[5] / n, v[6] / n};
    return T;
}

SE3 SE3::operator*(const SE3 &o) const
{
    double Ra[9], Rb[9], R[9];
    rotation(Ra);
    o.rotation(Rb);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = Ra[3 * i] * Rb[j] + Ra[3 * i + 1] * Rb[3 + j] + Ra[3 * i + 2] * Rb[6 + j];
    const Vec3 t = (*this) * Vec3{o.v[0], o.v[1], o.v[2]};
    const double tt[3] = {t.x, t.y, t.z};
    return fromRt(R, tt);
}

// ---------------------------------------------------------------------------------------------- Frame
std::vector<Keypoint> Frame::getKeypoints3d() const
{
    std::vector<Keypoint> v;
    v.reserve(nb3dkps_);
    for (const auto &kv : mapkps_) if (kv.second.is3d_) v.push_back(kv.second);
    return v;
}

Keypoint Frame::getKeypointById(int lmid) const
{
    auto it = mapkps_.find(lmid);
    return it == mapkps_.end() ? Keypoint() : it->second;
}

void Frame::addKeypoint(const Keypoint &kp)
{
    if (mapkps_.count(kp.lmid_)) return;
    mapkps_.emplace(kp.lmid_, kp);
    if (ncellsize_) {   // addKeypointToGrid (src/frame.cpp:508-519)
        const int idx = getKeypointCellIdx(kp.px_);
        if (idx >= 0 && idx < (int)vgridkps_.size()) { if (vgridkps_[idx].empty()) noccupcells_++; vgridkps_[idx].push_back(kp.lmid_); }
    }
    ++nbkps_;
    if (kp.is3d_) ++nb3dkps_; else ++nb2dkps_;
    if (kp.is_stereo_) ++nb_stereo_kps_;
}

void Frame::updateKeypoint(int lmid, const Point2f &pt)
{
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end()) return;
    if (ncellsize_) {   // updateKeypointInGrid (src/frame.cpp:543-575): move the id when the cell changes
        const int idx = getKeypointCellIdx(it->second.px_), nidx = getKeypointCellIdx(pt);
        if (idx != nidx) {
            if (idx >= 0 && idx < (int)vgridkps_.size()) {
                auto &cell = vgridkps_[idx];
                for (size_t k = 0; k < cell.size(); ++k)
                    if (cell[k] == lmid) { cell.erase(cell.begin() + k); if (cell.empty()) noccupcells_--; break; }
            }
            if (nidx >= 0 && nidx < (int)vgridkps_.size()) { if (vgridkps_[nidx].empty()) noccupcells_++; vgridkps_[nidx].push_back(lmid); }
        }
    }
    computeKeypoint(pt, it->second);   // px_, unpx_ (no distortion), bv_
}

void Frame::removeKeypointById(int lmid)
{
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end()) return;
    if (it->second.is3d_) --nb3dkps_; else --nb2dkps_;
    if (it->second.is_stereo_) --nb_stereo_kps_;
    --nbkps_;
    if (ncellsize_) {   // removeKeypointFromGrid (src/frame.cpp:521-541)
        const int idx = getKeypointCellIdx(it->second.px_);
        if (idx >= 0 && idx < (int)vgridkps_.size()) {
            auto &cell = vgridkps_[idx];
            for (size_t k = 0; k < cell.size(); ++k)
                if (cell[k] == lmid) { cell.erase(cell.begin() + k); if (cell.empty()) noccupcells_--; break; }
        }
    }
    mapkps_.erase(it);
}

void Frame::updateKeypointDesc(int lmid, const Desc &d)
{   // src/frame.cpp:356-366
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end()) return;
    it->second.desc_ = d; it->second.has_desc_ = true;
}

bool Frame::updateKeypointId(int prevlmid, int newlmid, bool is3d)
{   // src/frame.cpp:380-402
    if (mapkps_.count(newlmid)) return false;
    auto it = mapkps_.find(prevlmid);
    if (it == mapkps_.end()) return false;
    Keypoint upkp = it->second;
    upkp.lmid_ = newlmid;
    upkp.is_retracked_ = true;
    upkp.is3d_ = is3d;
    removeKeypointById(prevlmid);
    addKeypoint(upkp);
    return true;
}

void Frame::removeStereoKeypointById(int lmid)
{
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end() || !it->second.is_stereo_) return;
    it->second.is_stereo_ = false;
    --nb_stereo_kps_;
}

void Frame::turnKeypoint3d(int lmid)
{
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end() || it->second.is3d_) return;
    it->second.is3d_ = true;
    ++nb3dkps_;
    --nb2dkps_;
}

bool Frame::isInImage(const Point2f &pt) const
{
    return pt.x >= 0 && pt.y >= 0 && pt.x < pcalib_leftcam_->img_w_ && pt.y < pcalib_leftcam_->img_h_;
}

Point2f Frame::projWorldToImage(const Vec3 &wpt) const
{
    const Vec3 pc = Tcw_ * wpt;
    const Vec3 px = pcalib_leftcam_->projectCamToImage(pc);
    return {(float)px.x, (float)px.y};
}

void Frame::initGrid(size_t ncellsize)
{   // src/frame.cpp:35-72
    ncellsize_ = ncellsize;
    nbwcells_ = (size_t)std::ceil((float)pcalib_leftcam_->img_w_ / (float)ncellsize_);
    nbhcells_ = (size_t)std::ceil((float)pcalib_leftcam_->img_h_ / (float)ncellsize_);
    noccupcells_ = 0;
    vgridkps_.assign(nbwcells_ * nbhcells_, {});
    for (const auto &kv : mapkps_) {
        const int idx = getKeypointCellIdx(kv.second.px_);
        if (idx >= 0 && idx < (int)vgridkps_.size()) vgridkps_[idx].push_back(kv.first);
    }
    if (pcalib_rightcam_) {   // Frl_ = K_r^-T [t]x R K_l^-1 with T = Tcic0 (:53-62)
        const SE3 Tcic0 = pcalib_rightcam_->Tc0ci_.inverse();
        double R[9];
        Tcic0.rotation(R);
        const double t[3] = {Tcic0.v[0], Tcic0.v[1], Tcic0.v[2]};
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        const CameraCalibration &cl = *pcalib_leftcam_, &cr = *pcalib_rightcam_;
        const double Kli[9] = {1 / cl.fx_, 0, -cl.cx_ / cl.fx_, 0, 1 / cl.fy_, -cl.cy_ / cl.fy_, 0, 0, 1};
        const double Krit[9] = {1 / cr.fx_, 0, 0, 0, 1 / cr.fy_, 0, -cr.cx_ / cr.fx_, -cr.cy_ / cr.fy_, 1};
        double A[9], B[9];
        auto mul = [](const double *X, const double *Y, double *Z) {
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Z[3 * i + j] = X[3 * i] * Y[j] + X[3 * i + 1] * Y[3 + j] + X[3 * i + 2] * Y[6 + j];
        };
        mul(Krit, tx, A); mul(A, R, B); mul(B, Kli, Frl_);
    }
}

int Frame::getKeypointCellIdx(const Point2f &pt) const
{   // src/frame.cpp: r = floor(y / cell), c = floor(x / cell)
    if (!ncellsize_) return -1;
    const int r = (int)std::floor(pt.y / (float)ncellsize_), c = (int)std::floor(pt.x / (float)ncellsize_);
    return r * (int)nbwcells_ + c;
}

std::vector<Keypoint> Frame::getKeypoints() const
{
    std::vector<Keypoint> v;
    v.reserve(nbkps_);
    for (const auto &kv : mapkps_) v.push_back(kv.second);
    return v;
}

std::vector<Keypoint> Frame::getSurroundingKeypoints(const Keypoint &kp) const
{   // src/frame.cpp:594-622 (note the reference's half-open loops: the cell itself and its upper / left neighbours)
    std::vector<Keypoint> vkps;
    if (!ncellsize_) return vkps;
    const int rkp = (int)std::floor(kp.px_.y / (float)ncellsize_), ckp = (int)std::floor(kp.px_.x / (float)ncellsize_);
    for (int r = rkp - 1; r < rkp + 1; r++)
        for (int c = ckp - 1; c < ckp + 1; c++) {
            const int idx = r * (int)nbwcells_ + c;
            if (r < 0 || c < 0 || idx > (int)vgridkps_.size()) continue;
            if (idx == (int)vgridkps_.size()) continue;   // the reference's `idx > size` lets idx == size through to .at(): out of range there
            for (const int id : vgridkps_[idx])
                if (id != kp.lmid_) {
                    auto it = mapkps_.find(id);
                    if (it != mapkps_.end()) vkps.push_back(it->second);
                }
        }
    return vkps;
}

Point2f CameraCalibration::undistortImagePoint(const Point2f &pt) const
{
    if (D_.empty()) return pt;
    const double u = pt.x, v = pt.y;
    if (model_ == Pinhole) {
        const double k1 = D_[0], k2 = D_.size() > 1 ? D_[1] : 0., p1 = D_.size() > 2 ? D_[2] : 0., p2 = D_.size() > 3 ? D_[3] : 0.,
                     k3 = D_.size() > 4 ? D_[4] : 0.;
        double x = (u - cx_) * (1. / fx_), y = (v - cy_) * (1. / fy_);
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {   // TermCriteria(COUNT, 5): no epsilon test
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1. + ((k3 * r2 + k2) * r2 + k1) * r2);
            if (icdist < 0) { x = (u - cx_) * (1. / fx_); y = (v - cy_) * (1. / fy_); break; }
            const double dX = 2. * p1 * x * y + p2 * (r2 + 2. * x * x), dY = p1 * (r2 + 2. * y * y) + 2. * p2 * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        return Point2f{(float)(fx_ * x + cx_), (float)(fy_ * y + cy_)};
    }
    const double k[4] = {D_[0], D_.size() > 1 ? D_[1] : 0., D_.size() > 2 ? D_[2] : 0., D_.size() > 3 ? D_[3] : 0.};
    const double pwx = (u - cx_) / fx_, pwy = (v - cy_) / fy_;
    double theta_d = std::sqrt(pwx * pwx + pwy * pwy);
    theta_d = std::min(std::max(-M_PI / 2., theta_d), M_PI / 2.);
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double k0t2 = k[0] * t2, k1t4 = k[1] * t4, k2t6 = k[2] * t6, k3t8 = k[3] * t8;
            const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
            theta -= fix;
            if (std::fabs(fix) < 1e-8) break;
        }
        scale = std::tan(theta) / theta_d;
    }
    return Point2f{(float)(fx_ * (pwx * scale) + cx_), (float)(fy_ * (pwy * scale) + cy_)};
}

Point2f CameraCalibration::projectCamToImageDist(const Vec3 &pc) const
{
    const double invz = 1. / pc.z, x = pc.x * invz, y = pc.y * invz;
    if (D_.empty()) return Point2f{(float)(fx_ * x + cx_), (float)(fy_ * y + cy_)};
    if (model_ == Pinhole) {
        // the reference hands cv::projectPoints a Point3f(x, y, 1): the normalised coordinates pass through float
        const double xf = (double)(float)x, yf = (double)(float)y;
        const double k1 = D_[0], k2 = D_.size() > 1 ? D_[1] : 0., p1 = D_.size() > 2 ? D_[2] : 0., p2 = D_.size() > 3 ? D_[3] : 0.,
                     k3 = D_.size() > 4 ? D_[4] : 0.;
        const double r2 = xf * xf + yf * yf, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * xf * yf, a2 = r2 + 2 * xf * xf, a3 = r2 + 2 * yf * yf;
        const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        const double xd = xf * cdist + p1 * a1 + p2 * a2, yd = yf * cdist + p1 * a3 + p2 * a1;
        return Point2f{(float)(xd * fx_ + cx_), (float)(yd * fy_ + cy_)};
    }
    const double xf = (double)(float)x, yf = (double)(float)y;   // Point2f(x, y)
    const double r = std::sqrt(xf * xf + yf * yf), theta = std::atan(r);
    const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double theta_d = theta * (1 + D_[0] * t2 + (D_.size() > 1 ? D_[1] : 0.) * t4 + (D_.size() > 2 ? D_[2] : 0.) * t6 + (D_.size() > 3 ? D_[3] : 0.) * t8);
    const double inv_r = r > 1e-8 ? 1.0 / r : 1.0, cdist = r > 1e-8 ? theta_d * inv_r : 1.0;
    return Point2f{(float)(fx_ * (xf * cdist) + cx_), (float)(fy_ * (yf * cdist) + cy_)};
}

void Frame::computeKeypoint(const Point2f &pt, Keypoint &kp) const
{   // src/frame.cpp:246-254
    kp.px_ = pt;
    kp.unpx_ = pcalib_leftcam_ ? pcalib_leftcam_->undistortImagePoint(pt) : pt;
    if (!pcalib_leftcam_) return;
    const CameraCalibration &cl = *pcalib_leftcam_;
    const double hx = (double)kp.unpx_.x, hy = (double)kp.unpx_.y;
    Vec3 bv{(hx - cl.cx_) / cl.fx_, (hy - cl.cy_) / cl.fy_, 1.0};   // iK * [unpx, 1]
    const double nrm = std::sqrt(bv.x * bv.x + bv.y * bv.y + bv.z * bv.z);
    kp.bv_ = Vec3{bv.x / nrm, bv.y / nrm, bv.z / nrm};
}

void Frame::updateKeypointStereo(int lmid, const Point2f &pt)
{   // src/frame.cpp:405-433
    auto it = mapkps_.find(lmid);
    if (it == mapkps_.end()) return;
    Keypoint &kp = it->second;
    kp.rpx_ = pt;
    const CameraCalibration &cr = *pcalib_rightcam_;
    kp.runpx_ = cr.undistortImagePoint(pt);
    Vec3 bv{((double)kp.runpx_.x - cr.cx_) / cr.fx_, ((double)kp.runpx_.y - cr.cy_) / cr.fy_, 1.0};
    const double nrm = std::sqrt(bv.x * bv.x + bv.y * bv.y + bv.z * bv.z);
    kp.rbv_ = Vec3{bv.x / nrm, bv.y / nrm, bv.z / nrm};
    if (!kp.is_stereo_) { kp.is_stereo_ = true; nb_stereo_kps_++; }
}

Point2f Frame::projCamToRightImageDist(const Vec3 &pt) const
{   // src/frame.cpp:796-799 -> CameraCalibration::projectCamToImageDist (src/camera_calibration.cpp:254-282)
    const CameraCalibration &cr = *pcalib_rightcam_;
    return cr.projectCamToImageDist(cr.Tc0ci_.inverse() * pt);
}

Point2f Frame::projWorldToRightImageDist(const Vec3 &wpt) const { return projCamToRightImageDist(projWorldToCam(wpt)); }

bool Frame::isInRightImage(const Point2f &pt) const
{
    return pt.x >= 0 && pt.y >= 0 && pt.x < pcalib_rightcam_->img_w_ && pt.y < pcalib_rightcam_->img_h_;
}

// ---------------------------------------------------------------------------------------------- MapPoint / MapManager
static inline float hamming32(const Desc &a, const Desc &b)   // cv::norm(a, b, cv::NORM_HAMMING) of two 1 x 32 CV_8U rows
{
    int d = 0;
    for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return (float)d;
}

void MapPoint::removeKfObs(int kfid)
{   // src/map_point.cpp:106-160
    if (!set_kfids_.count(kfid)) return;
    set_kfids_.erase(kfid);
    if (set_kfids_.empty()) {
        has_desc_ = false; map_kf_desc_.clear(); map_desc_dist_.clear();
        return;
    }
    if (kfid == kfid_) kfid_ = *set_kfids_.begin();   // the anchor moves to the oldest observer left
    // the most representative descriptor among those left (:129-159)
    float mindist = (has_desc_ ? 32 : 0) * 8.f;      // desc_.cols * 8
    int minid = -1;
    auto itdesc = map_kf_desc_.find(kfid);
    if (itdesc != map_kf_desc_.end()) {
        for (const auto &kf_d : map_kf_desc_) {
            if (kf_d.first == kfid) continue;
            const float dist = hamming32(itdesc->second, kf_d.second);
            float &descdist = map_desc_dist_.find(kf_d.first)->second;
            descdist -= dist;
            if (descdist < mindist) { mindist = descdist; minid = kf_d.first; }
        }
        map_kf_desc_.erase(kfid);
        map_desc_dist_.erase(kfid);
        if (minid > 0) desc_ = map_kf_desc_.at(minid);   // (`> 0`, not `>= 0`: keyframe 0 never becomes the representative here -- as in the reference)
    }
}

void MapPoint::addDesc(int kfid, const Desc &d)
{   // src/map_point.cpp:162-211
    if (map_kf_desc_.count(kfid)) return;
    map_kf_desc_.emplace(kfid, d);
    map_desc_dist_.emplace(kfid, 0.f);
    float &newdescdist = map_desc_dist_.find(kfid)->second;
    if (map_kf_desc_.size() == 1) { desc_ = d; has_desc_ = true; return; }
    float mindist = (has_desc_ ? 32 : 0) * 8.f;
    int minid = -1;
    for (const auto &kf_d : map_kf_desc_) {   // includes the new one (distance 0 to itself), as the reference's loop does
        const float dist = hamming32(d, kf_d.second);
        map_desc_dist_.at(kf_d.first) += dist;
        if (dist < mindist) { mindist = dist; minid = kf_d.first; }
        newdescdist += dist;
    }
    if (newdescdist < mindist) minid = kfid;
    desc_ = map_kf_desc_.at(minid);   // (minid == -1 only when desc_ was empty with descriptors present: the reference throws there too)
    has_desc_ = true;
}

bool MapPoint::isBad()
{   // src/map_point.cpp:215-234
    if (set_kfids_.size() < 2) {
        if (!isobs_ && is3d_) { is3d_ = false; return true; }
    }
    if (set_kfids_.size() == 0 && !isobs_) { is3d_ = false; return true; }
    return false;
}

std::shared_ptr<Frame> MapManager::getKeyframe(int kfid) const
{
    auto it = map_pkfs_.find(kfid);
    return it == map_pkfs_.end() ? nullptr : it->second;
}

std::shared_ptr<MapPoint> MapManager::getMapPoint(int lmid) const
{
    auto it = map_plms_.find(lmid);
    return it == map_plms_.end() ? nullptr : it->second;
}

void MapManager::updateMapPoint(int lmid, const Vec3 &wpt, double kfanch_invdepth)
{
    auto plm = getMapPoint(lmid);
    if (!plm) return;
    if (!plm->is3d_) {   // turn the observations 3D (src/map_manager.cpp updateMapPoint)
        for (int kfid : plm->getKfObsSet()) {
            auto pkf = getKeyframe(kfid);
            if (pkf) pkf->turnKeypoint3d(lmid);
        }
        if (plm->isobs_ && pcurframe_) pcurframe_->turnKeypoint3d(lmid);
    }
    plm->setPoint(wpt, kfanch_invdepth);
    touchMapPoint(lmid);
}

void MapManager::removeMapPointObs(int lmid, int kfid)
{
    // src/map_manager.cpp:970-1003
    auto pkf = getKeyframe(kfid);
    if (pkf) pkf->removeKeypointById(lmid);
    if (dev_) { dev_rm_kf_.push_back(kfid); dev_rm_lm_.push_back(lmid); }
    auto plm = getMapPoint(lmid);
    if (!plm) return;
    plm->removeKfObs(kfid);
    if (pkf)
        for (int cokfid : plm->getKfObsSet()) {
            auto pcokf = getKeyframe(cokfid);
            if (pcokf) { pkf->decreaseCovisibleKf(cokfid); pcokf->decreaseCovisibleKf(kfid); }
        }
}

void MapManager::removeMapPoint(int lmid)
{
    auto plm = getMapPoint(lmid);
    if (!plm) return;
    for (int kfid : plm->getKfObsSet()) {   // src/map_manager.cpp:930-944
        auto pkf = getKeyframe(kfid);
        if (!pkf) continue;
        pkf->removeKeypointById(lmid);
        for (int cokfid : plm->getKfObsSet())
            if (cokfid != kfid) pkf->decreaseCovisibleKf(cokfid);
    }
    if (plm->isobs_ && pcurframe_) pcurframe_->removeKeypointById(lmid);
    map_plms_.erase(lmid);
    touchMapPoint(lmid);
}

void MapManager::removeObsFromCurFrameById(int lmid)
{
    if (pcurframe_) pcurframe_->removeKeypointById(lmid);
    auto plm = getMapPoint(lmid);
    if (plm) plm->isobs_ = false;
    touchMapPoint(lmid);
}

void MapManager::updateFrameCovisibility(Frame &frame)
{   // src/map_manager.cpp:117-192: co-observation counts per keyframe + the local map (3D points of the covisible keyframes
    // that this frame does not observe) for Mapper::matchingToLocalMap
    std::map<int, int> map_covkfs;
    std::unordered_set<int> set_local_mapids;
    for (const auto &kp : frame.getKeypoints()) {
        auto plm = getMapPoint(kp.lmid_);
        if (!plm) {
            removeMapPointObs(kp.lmid_, frame.kfid_);
            removeObsFromCurFrameById(kp.lmid_);
            continue;
        }
        for (int kfid : plm->getKfObsSet())
            if (kfid != frame.kfid_) map_covkfs[kfid] += 1;
    }
    std::set<int> set_badkfids;
    for (const auto &kfid_cov : map_covkfs) {
        auto pkf = getKeyframe(kfid_cov.first);
        if (pkf) {
            pkf->map_covkfs_[frame.kfid_] = kfid_cov.second;
            for (const auto &kp : pkf->getKeypoints3d())
                if (!frame.isObservingKp(kp.lmid_)) set_local_mapids.insert(kp.lmid_);
        } else set_badkfids.insert(kfid_cov.first);
    }
    for (int kfid : set_badkfids) map_covkfs.erase(kfid);
    frame.map_covkfs_.swap(map_covkfs);
    if (set_local_mapids.size() > 0.5 * frame.set_local_mapids_.size()) frame.set_local_mapids_.swap(set_local_mapids);
    else frame.set_local_mapids_.insert(set_local_mapids.begin(), set_local_mapids.end());
}

void MapManager::setMapPointObs(int lmid)
{   // src/map_manager.cpp:1053-1090
    auto plm = getMapPoint(lmid);
    if (!plm) return;
    plm->isobs_ = true;
    touchMapPoint(lmid);
}

void MapManager::mergeMapPoints(int prevlmid, int newlmid)
{   // src/map_manager.cpp:801-882: the observations and descriptors of prevlmid go to newlmid, prevlmid leaves the map
    auto pprev = getMapPoint(prevlmid), pnew = getMapPoint(newlmid);
    if (!pprev || !pnew || !pnew->is3d_) return;
    const std::set<int> setnewkfids = pnew->getKfObsSet(), setprevkfids = pprev->getKfObsSet();
    const std::unordered_map<int, Desc> map_prev_kf_desc = pprev->map_kf_desc_;
    for (int pkfid : setprevkfids) {
        auto pkf = getKeyframe(pkfid);
        if (!pkf) continue;
        if (pkf->updateKeypointId(prevlmid, newlmid, pnew->is3d_)) {
            pnew->addKfObs(pkfid);
            for (int nkfid : setnewkfids) {
                auto pcokf = getKeyframe(nkfid);
                if (pcokf) { pkf->addCovisibleKf(nkfid); pcokf->addCovisibleKf(pkfid); }
            }
            if (dev_ && dev_kfs_.count(pkfid)) {   // the mirror holds this keyframe: its row (pkfid, prevlmid) becomes (pkfid, newlmid)
                dev_rm_kf_.push_back(pkfid); dev_rm_lm_.push_back(prevlmid);
                dev_add_obs_.emplace_back(pkfid, newlmid);
            }
        }
    }
    for (const auto &kfid_desc : map_prev_kf_desc) pnew->addDesc(kfid_desc.first, kfid_desc.second);
    if (pcurframe_ && pcurframe_->isObservingKp(prevlmid) && pcurframe_->updateKeypointId(prevlmid, newlmid, pnew->is3d_)) setMapPointObs(newlmid);
    map_plms_.erase(prevlmid);
    touchMapPoint(prevlmid);   // gone: the mirror's row dies with the next flush
    touchMapPoint(newlmid);
}

// ---------------------------------------------------------------------------------------------- device map mirror
MapManager::~MapManager() { if (dev_) ov2_map_destroy(dev_); }

static uint8_t lm_state_of(const MapManager &map, const MapPoint &lm)
{
    // Keypoint::is3d_ of its observations (all turned together by Frame::turnKeypoint3d): read it off the first observer
    bool kp3d = lm.is3d_;
    if (!kp3d && !lm.set_kfids_.empty()) {
        auto pkf = map.getKeyframe(*lm.set_kfids_.begin());
        if (pkf) { const Keypoint kp = pkf->getKeypointById(lm.lmid_); kp3d = kp.lmid_ == lm.lmid_ && kp.is3d_; }
    }
    return (uint8_t)(OV2_LM_ALIVE | (lm.is3d_ ? OV2_LM_3D : 0) | (lm.isobs_ ? OV2_LM_OBS : 0) | (kp3d ? OV2_LM_KP3D : 0));
}

ov2_status MapManager::addKeyframeToDevice(const Frame &kf)
{
    if (!dev_) return OV2_OK;
    const size_t n = kf.mapkps_.size();
    std::vector<int32_t> lmid, scale; std::vector<double> un, run; std::vector<uint8_t> st;
    lmid.reserve(n); scale.reserve(n); un.reserve(2 * n); run.reserve(2 * n); st.reserve(n);
    for (const auto &kv : kf.mapkps_) {
        const Keypoint &kp = kv.second;
        lmid.push_back(kp.lmid_); scale.push_back(kp.scale_);
        un.push_back(kp.unpx_.x); un.push_back(kp.unpx_.y);
        run.push_back(kp.runpx_.x); run.push_back(kp.runpx_.y);
        st.push_back(kp.is_stereo_ ? 1 : 0);
    }
    const SE3 T = kf.getTwc();
    dev_kfs_.insert(kf.kfid_);
    return ov2_map_add_keyframe(dev_, kf.kfid_, T.v.data(), (int)n, lmid.data(), un.data(), run.data(), st.data(), scale.data());
}

ov2_status MapManager::attachDevice(ov2_ctx *ctx, int max_kf, int max_lm, int max_obs)
{
    if (dev_) { ov2_map_destroy(dev_); dev_ = nullptr; }
    dev_kfs_.clear(); dev_add_obs_.clear();
    ov2_status s = ov2_map_create(ctx, max_kf, max_lm, max_obs, &dev_);
    if (s != OV2_OK) return s;
    std::vector<int32_t> lmid; std::vector<double> xyz; std::vector<uint8_t> st;
    for (const auto &kv : map_plms_) {
        const Vec3 p = kv.second->getPoint();
        lmid.push_back(kv.first); xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z);
        st.push_back(lm_state_of(*this, *kv.second));
    }
    if ((s = ov2_map_set_landmarks(dev_, (int)lmid.size(), lmid.data(), xyz.data(), st.data())) != OV2_OK) return s;
    for (const auto &kv : map_pkfs_)
        if ((s = addKeyframeToDevice(*kv.second)) != OV2_OK) return s;
    dev_lm_dirty_.clear(); dev_pose_dirty_.clear(); dev_rm_kf_.clear(); dev_rm_lm_.clear(); dev_st_kf_.clear(); dev_st_lm_.clear();
    return OV2_OK;
}

ov2_status MapManager::flushDevice()
{
    if (!dev_) return OV2_OK;
    ov2_status s;
    if (!dev_rm_kf_.empty()) {
        if ((s = ov2_map_remove_obs(dev_, (int)dev_rm_kf_.size(), dev_rm_kf_.data(), dev_rm_lm_.data())) != OV2_OK) return s;
        dev_rm_kf_.clear(); dev_rm_lm_.clear();
    }
    if (!dev_add_obs_.empty()) {   // after the removals: a merged observation (kf, prev) -> (kf, new) is a dead row + an appended one
        std::map<int, std::vector<int>> by_kf;
        for (const auto &e : dev_add_obs_) by_kf[e.first].push_back(e.second);
        dev_add_obs_.clear();
        for (const auto &kv : by_kf) {
            auto pkf = getKeyframe(kv.first);
            if (!pkf) continue;
            std::vector<int32_t> lmid, scale; std::vector<double> un, run; std::vector<uint8_t> st;
            for (int l : kv.second) {
                const Keypoint kp = pkf->getKeypointById(l);
                if (kp.lmid_ != l) continue;   // removed again since
                lmid.push_back(l); scale.push_back(kp.scale_);
                un.push_back(kp.unpx_.x); un.push_back(kp.unpx_.y); run.push_back(kp.runpx_.x); run.push_back(kp.runpx_.y);
                st.push_back(kp.is_stereo_ ? 1 : 0);
            }
            const SE3 T = pkf->getTwc();
            if (!lmid.empty() && (s = ov2_map_add_keyframe(dev_, kv.first, T.v.data(), (int)lmid.size(), lmid.data(), un.data(), run.data(), st.data(),
                                                           scale.data())) != OV2_OK) return s;
        }
    }
    if (!dev_st_kf_.empty()) {
        std::vector<uint8_t> off(dev_st_kf_.size(), 0);
        std::vector<double> zero(2 * dev_st_kf_.size(), 0.0);
        if ((s = ov2_map_set_obs_stereo(dev_, (int)dev_st_kf_.size(), dev_st_kf_.data(), dev_st_lm_.data(), off.data(), zero.data())) != OV2_OK)
            return s;
        dev_st_kf_.clear(); dev_st_lm_.clear();
    }
    if (!dev_lm_dirty_.empty()) {
        std::sort(dev_lm_dirty_.begin(), dev_lm_dirty_.end());
        dev_lm_dirty_.erase(std::unique(dev_lm_dirty_.begin(), dev_lm_dirty_.end()), dev_lm_dirty_.end());
        std::vector<double> xyz; std::vector<uint8_t> st;
        for (int lmid : dev_lm_dirty_) {
            auto plm = getMapPoint(lmid);
            const Vec3 p = plm ? plm->getPoint() : Vec3{0, 0, 0};
            xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z);
            st.push_back(plm ? lm_state_of(*this, *plm) : 0);
        }
        if ((s = ov2_map_set_landmarks(dev_, (int)dev_lm_dirty_.size(), dev_lm_dirty_.data(), xyz.data(), st.data())) != OV2_OK) return s;
        dev_lm_dirty_.clear();
    }
    if (!dev_pose_dirty_.empty()) {
        std::sort(dev_pose_dirty_.begin(), dev_pose_dirty_.end());
        dev_pose_dirty_.erase(std::unique(dev_pose_dirty_.begin(), dev_pose_dirty_.end()), dev_pose_dirty_.end());
        std::vector<int32_t> ids; std::vector<double> T;
        for (int kfid : dev_pose_dirty_) {
            auto pkf = getKeyframe(kfid);
            if (!pkf) continue;
            const SE3 t = pkf->getTwc();
            ids.push_back(kfid); T.insert(T.end(), t.v.begin(), t.v.end());
        }
        if ((s = ov2_map_set_poses(dev_, (int)ids.size(), ids.data(), T.data())) != OV2_OK) return s;
        dev_pose_dirty_.clear();
    }
    return OV2_OK;
}

// ---------------------------------------------------------------------------------------------- FeatureTracker
ov2_status FeatureTracker::fbKltTracking(const Pyramid &vprevpyr, const Pyramid &vcurpyr, int nwinsize, int nbpyrlvl,
                                         float ferr, float fmax_fbklt_dist, std::vector<Point2f> &vkps,
                                         std::vector<Point2f> &vpriorkps, std::vector<bool> &vkpstatus) const
{
    if (vkps.empty()) return OV2_OK;   // src/feature_tracker.cpp:43-46
    const int n = (int)vkps.size();
    if ((int)vpriorkps.size() != n) return OV2_ERR_INVALID;
    std::vector<uint8_t> st(n);
    static_assert(sizeof(Point2f) == 2 * sizeof(float), "Point2f must be two packed floats");
    ov2_status s = ov2_klt_track_fb(ctx_, vprevpyr.h, vcurpyr.h, nwinsize, nbpyrlvl, nmax_iter_, fmax_px_precision_, ferr,
                                    fmax_fbklt_dist, n, &vkps[0].x, &vpriorkps[0].x, st.data());
    if (s != OV2_OK) return s;
    vkpstatus.resize(n);
    for (int i = 0; i < n; ++i) vkpstatus[i] = st[i] != 0;
    return OV2_OK;
}

bool FeatureTracker::inBorder(const Point2f &pt, int cols, int rows) const
{
    const float B = 1.f;
    return B <= pt.x && pt.x < cols - B && B <= pt.y && pt.y < rows - B;
}

// ---------------------------------------------------------------------------------------------- FeatureExtractor
std::vector<Point2f> FeatureExtractor::detect(const Pyramid &pyr, int ncellsize, int mode, const std::vector<Point2f> &vcurkps,
                                              const int roi[4])
{
    std::vector<Point2f> out;
    int w = 0, h = 0;
    if (pyr.empty() || ov2_pyr_level_size(pyr.h, 0, &w, &h, nullptr) != OV2_OK || ncellsize <= 0) { last_status_ = OV2_ERR_INVALID; return out; }
    const int cap = 2 * (w / ncellsize) * (h / ncellsize) + 2;
    out.resize(cap);
    double th = mode == OV2_DETECT_MINEIG ? dmaxquality_ : (double)nfast_th_;
    int n = 0;
    last_status_ = ov2_detect_grid(ctx_, pyr.h, 0, ncellsize, mode, &th, (int)vcurkps.size(),
                                   vcurkps.empty() ? nullptr : &vcurkps[0].x, roi, 1, &n, &out[0].x);
    if (last_status_ != OV2_OK) n = 0;
    else if (mode == OV2_DETECT_MINEIG) dmaxquality_ = th;   // :418-423
    else nfast_th_ = (int)th;                                 // :546-552
    out.resize(n);
    return out;
}

std::vector<Point2f> FeatureExtractor::detectSingleScale(const Pyramid &pyr, int ncellsize, const std::vector<Point2f> &vcurkps,
                                                         const int roi[4])
{
    return detect(pyr, ncellsize, OV2_DETECT_MINEIG, vcurkps, roi);
}

std::vector<Point2f> FeatureExtractor::detectGridFAST(const Pyramid &pyr, int ncellsize, const std::vector<Point2f> &vcurkps,
                                                      const int roi[4])
{
    return detect(pyr, ncellsize, OV2_DETECT_FAST, vcurkps, roi);
}

// ---------------------------------------------------------------------------------------------- stereoMatching
ov2_status MapManager::stereoMatching(Frame &frame, const Pyramid &vleftpyr, const Pyramid &vrightpyr,
                                      const FeatureTracker &tracker, const SlamParams &st)
{   // src/map_manager.cpp:367-611
    const std::vector<Keypoint> vleftkps = frame.getKeypoints();
    const size_t nbkps = vleftkps.size();
    // ZNCC parameters (:376-381)
    const int nmaxpyrlvl = st.nklt_pyr_lvl_;   // vleftpyr.at(nklt_pyr_lvl_ * 2) = the image of that level
    const int winsize = 7;
    const float uppyrcoef = std::pow(2.f, (float)st.nklt_pyr_lvl_);
    const float downpyrcoef = (float)(1. / uppyrcoef);

    // one flat batch for ov2_stereo_matching: has_prior = 1 <-> the reference's v3dkps / v3dpriors lists
    std::vector<int> vids;
    std::vector<Point2f> vkps, vpriors, vlunpx;
    std::vector<uint8_t> vhas;
    std::vector<size_t> vsad;            // entries whose prior comes from the SAD line search
    std::vector<Point2f> vsadpts;
    vids.reserve(nbkps); vkps.reserve(nbkps); vpriors.reserve(nbkps); vlunpx.reserve(nbkps); vhas.reserve(nbkps);
    auto push = [&](const Keypoint &kp, const Point2f &prior, bool has) {
        vids.push_back(kp.lmid_); vkps.push_back(kp.px_); vpriors.push_back(prior); vlunpx.push_back(kp.unpx_);
        vhas.push_back(has ? 1 : 0);
    };
    for (size_t i = 0; i < nbkps; ++i) {
        const Keypoint &kp = vleftkps[i];
        if (kp.is3d_) {                                                     // :398-417
            auto plm = getMapPoint(kp.lmid_);
            if (plm != nullptr) {
                const Point2f projpt = frame.projWorldToRightImageDist(plm->getPoint());
                if (frame.isInRightImage(projpt)) { push(kp, projpt, true); continue; }
            } else {
                removeMapPointObs(kp.lmid_, frame.kfid_);
                continue;
            }
        }
        if (st.bdo_stereo_rect_) {                                          // :419-436: prior from SAD (batched below)
            vsad.push_back(vids.size());
            vsadpts.push_back(Point2f{kp.px_.x * downpyrcoef, kp.px_.y * downpyrcoef});
        } else {                                                            // :438-483: prior from the 3D neighbours
            const size_t nbmin3dcokps = 1;
            const std::vector<Keypoint> vnearkps = frame.getSurroundingKeypoints(kp);
            if (vnearkps.size() >= nbmin3dcokps) {
                std::vector<Keypoint> vnear3dkps;
                for (const auto &cokp : vnearkps)
                    if (cokp.is3d_) vnear3dkps.push_back(cokp);
                if (vnear3dkps.size() >= nbmin3dcokps) {
                    size_t nb3dkp = 0;
                    double mean_z = 0., weights = 0.;
                    for (const auto &cokp : vnear3dkps) {
                        auto plm = getMapPoint(cokp.lmid_);
                        if (plm != nullptr) {
                            nb3dkp++;
                            const float dx = cokp.unpx_.x - kp.unpx_.x, dy = cokp.unpx_.y - kp.unpx_.y;   // Point2f difference
                            const double coef = 1. / std::sqrt((double)dx * dx + (double)dy * dy);       // cv::norm(Point2f)
                            weights += coef;
                            mean_z += coef * frame.projWorldToCam(plm->getPoint()).z;
                        }
                    }
                    if (nb3dkp >= nbmin3dcokps) {
                        mean_z /= weights;
                        const Vec3 predcampt{mean_z * (kp.bv_.x / kp.bv_.z), mean_z * (kp.bv_.y / kp.bv_.z),
                                             mean_z * (kp.bv_.z / kp.bv_.z)};
                        const Point2f projpt = frame.projCamToRightImageDist(predcampt);
                        if (frame.isInRightImage(projpt)) { push(kp, projpt, true); continue; }
                    }
                }
            }
        }
        push(kp, kp.px_, false);                                            // :486-488 (priorpt = kp.px_ so far)
    }
    if (!vsad.empty()) {   // FeatureTracker::getLineMinSAD for every rectified 2D keypoint in one launch
        std::vector<float> xprior(vsad.size()), l1err(vsad.size());
        const ov2_status s = ov2_line_min_sad(tracker.ctx_, vleftpyr.h, vrightpyr.h, nmaxpyrlvl, winsize, 1, (int)vsad.size(),
                                              &vsadpts[0].x, xprior.data(), l1err.data());
        if (s != OV2_OK) return s;
        for (size_t k = 0; k < vsad.size(); ++k) {
            float xp = xprior[k];
            xp *= uppyrcoef;                                                // :431
            if (xp >= 0 && xp <= vkps[vsad[k]].x) vpriors[vsad[k]].x = xp;  // :433-435
        }
    }
    const int n = (int)vids.size();
    if (n == 0) return OV2_OK;
    static_assert(sizeof(Point2f) == 8, "packed floats");
    std::vector<Point2f> vout((size_t)n);
    std::vector<uint8_t> vstatus((size_t)n);
    // the gate undistorts the tracked right pixel with the right camera's lens model (:586)
    const CameraCalibration &cr = *frame.pcalib_rightcam_;
    ov2_cam_model rcam;
    std::memset(&rcam, 0, sizeof(rcam));
    rcam.K[0] = cr.fx_; rcam.K[1] = cr.fy_; rcam.K[2] = cr.cx_; rcam.K[3] = cr.cy_;
    rcam.model = cr.D_.empty() ? 0 : (cr.model_ == CameraCalibration::Fisheye ? 2 : 1);
    rcam.n_coeffs = (int32_t)std::min<size_t>(cr.D_.size(), 5);
    for (int k = 0; k < rcam.n_coeffs; ++k) rcam.D[k] = cr.D_[k];
    const ov2_status s = ov2_stereo_matching(tracker.ctx_, vleftpyr.h, vrightpyr.h, st.nklt_win_size_, st.nklt_pyr_lvl_,
                                             tracker.nmax_iter_, tracker.fmax_px_precision_, st.nklt_err_, st.fmax_fbklt_dist_, n,
                                             &vkps[0].x, &vpriors[0].x, vhas.data(), &vlunpx[0].x, st.bdo_stereo_rect_ ? 1 : 0,
                                             frame.Frl_, &rcam, &vout[0].x, vstatus.data());
    if (s != OV2_OK) return s;
    for (int i = 0; i < n; ++i)                                             // :597-601
        if (vstatus[i]) frame.updateKeypointStereo(vids[i], vout[i]);
    return OV2_OK;
}

// ---------------------------------------------------------------------------------------------- VisualFrontEnd
ov2_status VisualFrontEnd::preprocessImage(const uint8_t *img_raw, int w, int h, int stride)
{   // src/visual_front_end.cpp:1143-1177: swap pyramids, CLAHE (tiles w/50 x h/50, src/ov2slam.cpp:85-89), pyramid
    if (!cur_pyr_.empty()) prev_pyr_.swap(cur_pyr_);
    ov2_pyr *p = nullptr;
    ov2_status s = ov2_pyramid_build(ctx_, img_raw, w, h, stride, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_,
                                     pslamstate_->use_clahe_ ? 1 : 0, pslamstate_->fclahe_val_, w / 50, h / 50, &p);
    if (s != OV2_OK) return s;
    cur_pyr_ = Pyramid(p);
    return OV2_OK;
}

ov2_status VisualFrontEnd::kltTracking()
{   // src/visual_front_end.cpp:132-275
    std::vector<int> v3dkpids, vkpids;
    std::vector<Point2f> v3dkps, v3dpriors, vkps, vpriors;
    for (const auto &it : pcurframe_->mapkps_) {   // :155-184
        const Keypoint &kp = it.second;
        if (pslamstate_->klt_use_prior_ && kp.is3d_) {
            auto plm = pmap_->getMapPoint(kp.lmid_);
            if (plm) {
                const Point2f projpx = pcurframe_->projWorldToImage(plm->getPoint());
                if (pcurframe_->isInImage(projpx)) {
                    v3dkps.push_back(kp.px_); v3dpriors.push_back(projpx); v3dkpids.push_back(kp.lmid_);
                    continue;
                }
            }
        }
        vkpids.push_back(kp.lmid_); vkps.push_back(kp.px_); vpriors.push_back(kp.px_);
    }
    if (pslamstate_->klt_use_prior_ && !v3dpriors.empty()) {   // :187-234
        std::vector<bool> vkpstatus;
        ov2_status s = ptracker_->fbKltTracking(prev_pyr_, cur_pyr_, pslamstate_->nklt_win_size_, 1, pslamstate_->nklt_err_,
                                                pslamstate_->fmax_fbklt_dist_, v3dkps, v3dpriors, vkpstatus);
        if (s != OV2_OK) return s;
        size_t nbgood = 0;
        const size_t nbkps = v3dkps.size();
        for (size_t i = 0; i < nbkps; ++i) {
            if (vkpstatus[i]) { pcurframe_->updateKeypoint(v3dkpids[i], v3dpriors[i]); ++nbgood; }
            else { vkpids.push_back(v3dkpids[i]); vkps.push_back(v3dkps[i]); vpriors.push_back(v3dpriors[i]); }
        }
        if (nbgood < 0.33 * nbkps) { bp3preq_ = true; vpriors = vkps; }   // :228-233
    }
    if (!vkps.empty()) {   // :237-270
        std::vector<bool> vkpstatus;
        ov2_status s = ptracker_->fbKltTracking(prev_pyr_, cur_pyr_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_,
                                                pslamstate_->nklt_err_, pslamstate_->fmax_fbklt_dist_, vkps, vpriors, vkpstatus);
        if (s != OV2_OK) return s;
        for (size_t i = 0; i < vkps.size(); ++i) {
            if (vkpstatus[i]) pcurframe_->updateKeypoint(vkpids[i], vpriors[i]);
            else pmap_->removeObsFromCurFrameById(vkpids[i]);
        }
    }
    return OV2_OK;
}

// ---------------------------------------------------------------------------------------------- pose refinement
bool MultiViewGeometry::ceresPnP(ov2_ctx *ctx, const std::vector<Vec2> &vunkps, const std::vector<Vec3> &vwpts,
                                 const std::vector<int> &vscales, SE3 &Twc, int nmaxiter, float chi2th,
                                 bool buse_robust, bool bapply_l2_after_robust, float fx, float fy, float cx, float cy,
                                 std::vector<int> &voutliersidx)
{   // src/multi_view_geometry.cpp:492-586; the solve, the chi2 flags and the L2 re-solve are one kernel launch
    if (vunkps.size() != vwpts.size() || (!vscales.empty() && vscales.size() != vunkps.size())) return false;   // :500
    const int n = (int)vunkps.size();
    static_assert(sizeof(Vec2) == 16 && sizeof(Vec3) == 24, "packed doubles");
    const double K[4] = {fx, fy, cx, cy};
    std::vector<uint8_t> out((size_t)n + 1);
    int ok = 0;
    const ov2_status s = ov2_pnp_solve_batch(ctx, 1, &n, n ? &vunkps[0].x : nullptr, n ? &vwpts[0].x : nullptr,
                                             vscales.empty() ? nullptr : vscales.data(), K, Twc.v.data(), nmaxiter,
                                             chi2th, buse_robust, bapply_l2_after_robust, out.data(), &ok, nullptr);
    if (s != OV2_OK) return false;
    for (int i = 0; i < n; ++i)
        if (out[i]) voutliersidx.push_back(i);
    return ok != 0;
}

ov2_status VisualFrontEnd::computePose()
{   // src/visual_front_end.cpp:657-830
    const size_t nb3dkps = pcurframe_->nb3dkps_;
    if (nb3dkps < 4) return OV2_OK;                                        // :665-669
    if (bp3preq_ || pslamstate_->dop3p_) return OV2_ERR_UNSUPPORTED;       // P3P-RANSAC branch :722-785 (OpenGV)
    std::vector<Vec2> vkps;
    std::vector<Vec3> vwpts;
    std::vector<int> vkpids, voutliersidx, vscales;
    std::vector<int> order;                                               // the reference walks its hash map; ascending ids make the
    for (const auto &it : pcurframe_->mapkps_) order.push_back(it.first);  // summation order of the solve independent of the hash
    std::sort(order.begin(), order.end());
    for (const int id : order) {                                           // :688-708
        const Keypoint &kp = pcurframe_->mapkps_.at(id);
        if (!kp.is3d_) continue;
        auto plm = pmap_->getMapPoint(kp.lmid_);
        if (!plm) continue;
        vkps.push_back({kp.unpx_.x, kp.unpx_.y});
        vwpts.push_back(plm->getPoint());
        vscales.push_back(kp.scale_);
        vkpids.push_back(kp.lmid_);
    }
    SE3 Twc = pcurframe_->getTwc();
    const CameraCalibration &c = *pcurframe_->pcalib_leftcam_;
    const bool success = MultiViewGeometry::ceresPnP(ctx_, vkps, vwpts, vscales, Twc, 5, pslamstate_->robust_mono_th_, true,
                                                     pslamstate_->apply_l2_after_robust_, (float)c.fx_, (float)c.fy_,
                                                     (float)c.cx_, (float)c.cy_, voutliersidx);   // :788-803
    const size_t nbinliers = vwpts.size() - voutliersidx.size();
    bool bad_t = false;
    for (int i = 0; i < 3; ++i) bad_t = bad_t || !std::isfinite(Twc.v[i]);
    if (!success || nbinliers < 5 || voutliersidx.size() > 0.5 * vwpts.size() || bad_t) {   // :806-826
        bp3preq_ = true;   // "weird results, skipping here and applying p3p next"
        return OV2_OK;
    }
    pcurframe_->setTwc(Twc);                                              // :831
    bp3preq_ = false;
    for (const int idx : voutliersidx) pmap_->removeObsFromCurFrameById(vkpids.at(idx));   // :838-841
    return OV2_OK;
}

// ---------------------------------------------------------------------------------------------- Optimizer::localBA
ov2_ba_problem LocalBAProblem::view(const SlamParams &st, const Frame &newframe)
{
    ov2_ba_problem p;
    std::memset(&p, 0, sizeof(p));
    const CameraCalibration &cl = *newframe.pcalib_leftcam_;
    p.calib_l[0] = cl.fx_; p.calib_l[1] = cl.fy_; p.calib_l[2] = cl.cx_; p.calib_l[3] = cl.cy_;
    p.T_rl[6] = 1.0;
    if (st.stereo_ && newframe.pcalib_rightcam_) {
        const CameraCalibration &cr = *newframe.pcalib_rightcam_;
        p.calib_r[0] = cr.fx_; p.calib_r[1] = cr.fy_; p.calib_r[2] = cr.cx_; p.calib_r[3] = cr.cy_;
        const SE3 Trl = cr.Tc0ci_.inverse();   // Tlr = getExtrinsic(); Trl = Tlr.inverse()  (src/optimizer.cpp:116-118)
        for (int i = 0; i < 7; ++i) p.T_rl[i] = Trl.v[i];
    }
    p.inv_depth = st.buse_inv_depth_ ? 1 : 0;
    p.n_pose = (int)pose_const.size(); p.pose = pose.data(); p.pose_const = pose_const.data();
    p.n_lm = (int)lm_lmid.size(); p.lm = lm.data();
    p.lm_anchor_pose = lm_anchor_pose.data(); p.lm_anchor_uv = lm_anchor_uv.data();
    p.n_res = (int)res_type.size(); p.res_type = res_type.data(); p.res_pose = res_pose.data(); p.res_lm = res_lm.data();
    p.res_uv = res_uv.data(); p.res_sigma = res_sigma.data();
    return p;
}

void Optimizer::setupLocalBA(Frame &newframe, LocalBAProblem &pb)
{   // src/optimizer.cpp:43-430
    const int nmincovscore = pslamstate_->nmin_covscore_;
    if ((int)newframe.nb3dkps_ < nmincovscore) { pb.aborted = true; return; }   // :61-63
    size_t nmincstkfs = pslamstate_->stereo_ ? 1 : 2;                            // :65-68
    const bool inv = pslamstate_->buse_inv_depth_;

    auto add_pose = [&](int kfid, const std::shared_ptr<Frame> &pkf, bool cst) {
        const int idx = (int)pb.pose_const.size();
        pb.kfid_to_pose.emplace(kfid, idx);
        pb.pose_kfid.push_back(kfid);
        const SE3 T = pkf->getTwc();
        pb.pose.insert(pb.pose.end(), T.v.begin(), T.v.end());
        pb.pose_const.push_back(cst ? 1 : 0);
        pb.map_local_pkfs.emplace(kfid, pkf);
        if (cst) pb.set_cstkfids.insert(kfid);
        return idx;
    };

    std::map<int, int> map_covkfs = newframe.getCovisibleKfMap();   // :128-131
    map_covkfs.emplace(newframe.kfid_, (int)newframe.nb3dkps_);
    std::vector<int> lmids2opt;                                     // insertion-ordered set_lmids2opt
    std::unordered_set<int> set_lmids2opt;
    bool all_cst = false;
    const int nmaxkfid = map_covkfs.rbegin()->first;
    for (auto it = map_covkfs.rbegin(); it != map_covkfs.rend(); ++it) {   // :150-190, newest -> oldest
        const int kfid = it->first;
        int covscore = it->second;
        if (kfid > newframe.kfid_) covscore = (int)newframe.nbkps_;
        auto pkf = pmap_->getKeyframe(kfid);
        if (!pkf) { newframe.removeCovisibleKf(kfid); continue; }
        if (covscore >= nmincovscore && !all_cst && kfid > 0) {
            add_pose(kfid, pkf, false);
            for (const auto &kp : pkf->getKeypoints3d())
                if (set_lmids2opt.insert(kp.lmid_).second) lmids2opt.push_back(kp.lmid_);
        } else {
            add_pose(kfid, pkf, true);
            all_cst = true;
        }
    }
    std::sort(lmids2opt.begin(), lmids2opt.end());   // the reference iterates an unordered_set; any order is valid

    for (int lmid : lmids2opt) {   // :193-392
        auto plm = pmap_->getMapPoint(lmid);
        if (!plm) continue;
        if (plm->isBad()) { pb.set_badlmids.insert(lmid); continue; }
        pb.map_local_plms.emplace(lmid, plm);
        int lmidx = -1;
        if (!inv) {
            lmidx = (int)pb.lm_lmid.size();
            pb.lmid_to_lm.emplace(lmid, lmidx);
            pb.lm_lmid.push_back(lmid);
            const Vec3 p = plm->getPoint();
            pb.lm.push_back(p.x); pb.lm.push_back(p.y); pb.lm.push_back(p.z);
            pb.lm_anchor_pose.push_back(-1); pb.lm_anchor_uv.push_back(0); pb.lm_anchor_uv.push_back(0);
        }
        int kfanchid = -1;
        for (int kfid : plm->getKfObsSet()) {   // ascending kfid (std::set)
            if (kfid > nmaxkfid) continue;
            std::shared_ptr<Frame> pkf;
            auto pkfit = pb.map_local_pkfs.find(kfid);
            if (pkfit == pb.map_local_pkfs.end()) {   // :229-246: observers outside the window enter as constants
                pkf = pmap_->getKeyframe(kfid);
                if (!pkf) { pmap_->removeMapPointObs(lmid, kfid); continue; }
                add_pose(kfid, pkf, true);
            } else {
                pkf = pkfit->second;
            }
            const Keypoint kp = pkf->getKeypointById(lmid);
            if (kp.lmid_ != lmid) { pmap_->removeMapPointObs(lmid, kfid); continue; }
            const double sigma = std::pow(2., kp.scale_);
            auto add_res = [&](int type, const Point2f &uv) {
                pb.res_type.push_back((uint8_t)type);
                pb.res_pose.push_back(pb.kfid_to_pose.at(kfid));
                pb.res_lm.push_back(lmidx);
                pb.res_uv.push_back(uv.x); pb.res_uv.push_back(uv.y);
                pb.res_sigma.push_back(sigma);
            };
            if (inv && kfanchid < 0) {   // :251-287 anchor = first valid observer
                kfanchid = kfid;
                const double zanch = (pkf->getTcw() * plm->getPoint()).z;
                lmidx = (int)pb.lm_lmid.size();
                pb.lmid_to_lm.emplace(lmid, lmidx);
                pb.lm_lmid.push_back(lmid);
                pb.lm.push_back(1. / zanch);
                pb.lm_anchor_pose.push_back(pb.kfid_to_pose.at(kfid));
                pb.lm_anchor_uv.push_back(kp.unpx_.x); pb.lm_anchor_uv.push_back(kp.unpx_.y);
                if (kp.is_stereo_) { add_res(OV2_BA_RANCH_INV, kp.runpx_); pb.nbstereo++; }
                else pb.nbmono++;
                continue;
            }
            if (kp.is_stereo_) {   // :293-361
                add_res(inv ? OV2_BA_L_INV : OV2_BA_L_XYZ, kp.unpx_);
                add_res(inv ? OV2_BA_R_INV : OV2_BA_R_XYZ, kp.runpx_);
                pb.nbstereo++;
            } else {               // :363-391
                add_res(inv ? OV2_BA_L_INV : OV2_BA_L_XYZ, kp.unpx_);
                pb.nbmono++;
            }
        }
    }
    // gauge: at least nmincstkfs constant keyframes (:394-407; the reference walks an unordered_map, we take the
    // smallest kfids first)
    size_t nbcstkfs = pb.set_cstkfids.size();
    if (nbcstkfs < nmincstkfs) {
        std::vector<int> ids = pb.pose_kfid;
        std::sort(ids.begin(), ids.end());
        for (int kfid : ids) {
            if (nbcstkfs >= nmincstkfs) break;
            if (pb.set_cstkfids.count(kfid)) continue;
            pb.pose_const[pb.kfid_to_pose.at(kfid)] = 1;
            pb.set_cstkfids.insert(kfid);
            ++nbcstkfs;
        }
    }
}

ov2_status Optimizer::setupLocalBADevice(Frame &newframe, LocalBAProblem &pb)
{   // src/optimizer.cpp:43-430 through ov2_map_local_ba_setup; fills the same LocalBAProblem as setupLocalBA
    ov2_status s = pmap_->flushDevice();
    if (s != OV2_OK) return s;
    const bool inv = pslamstate_->buse_inv_depth_;
    const CameraCalibration &cl = *newframe.pcalib_leftcam_;
    const double Kl[4] = {cl.fx_, cl.fy_, cl.cx_, cl.cy_};
    ov2_local_ba_setup f;
    s = ov2_map_local_ba_setup(pmap_->dev_, newframe.kfid_, pslamstate_->nmin_covscore_, pslamstate_->stereo_ ? 1 : 2, inv ? 1 : 0, Kl, &f);
    if (s != OV2_OK) return s;
    if (f.aborted) { pb.aborted = true; return OV2_OK; }
    const size_t P = f.n_pose, L = f.n_lm, R = f.n_res, e = inv ? 1 : 3;
    pb.pose_kfid.assign(f.pose_kfid, f.pose_kfid + P);
    pb.pose_const.assign(f.pose_const, f.pose_const + P);
    pb.pose.assign(f.pose, f.pose + 7 * P);
    pb.lm_lmid.assign(f.lm_lmid, f.lm_lmid + L);
    pb.lm.assign(f.lm, f.lm + e * L);
    pb.lm_anchor_pose.assign(f.lm_anchor_pose, f.lm_anchor_pose + L);
    pb.lm_anchor_uv.assign(f.lm_anchor_uv, f.lm_anchor_uv + 2 * L);
    pb.res_type.assign(f.res_type, f.res_type + R);
    pb.res_pose.assign(f.res_pose, f.res_pose + R);
    pb.res_lm.assign(f.res_lm, f.res_lm + R);
    pb.res_uv.assign(f.res_uv, f.res_uv + 2 * R);
    pb.res_sigma.assign(f.res_sigma, f.res_sigma + R);
    // the id maps the update stage walks (:741-882)
    pb.kfid_to_pose.reserve(P); pb.map_local_pkfs.reserve(P);
    for (size_t i = 0; i < P; ++i) {
        const int kfid = pb.pose_kfid[i];
        pb.kfid_to_pose.emplace(kfid, (int)i);
        pb.map_local_pkfs.emplace(kfid, pmap_->getKeyframe(kfid));
        if (pb.pose_const[i]) pb.set_cstkfids.insert(kfid);
    }
    pb.lmid_to_lm.reserve(L); pb.map_local_plms.reserve(L);
    for (size_t i = 0; i < L; ++i) {
        const int lmid = pb.lm_lmid[i];
        pb.lmid_to_lm.emplace(lmid, (int)i);
        pb.map_local_plms.emplace(lmid, pmap_->getMapPoint(lmid));
    }
    pb.has_dev = ov2_map_setup_device_view(pmap_->dev_, &f, &pb.dev_view) == OV2_OK;
    for (int i = 0; i < f.n_bad; ++i) {   // MapPoint::isBad() also clears is3d_ (src/map_point.cpp:219,227)
        pb.set_badlmids.insert(f.bad_lmid[i]);
        auto plm = pmap_->getMapPoint(f.bad_lmid[i]);
        if (plm) { plm->isBad(); pmap_->touchMapPoint(f.bad_lmid[i]); }
    }
    return OV2_OK;
}

void Optimizer::updateAfterLocalBA(Frame &newframe, LocalBAProblem &pb, const ov2_ba_result &res, bool cur_frame_obs)
{   // flags :500-592 / :637-735, update :741-882
    const bool inv = pslamstate_->buse_inv_depth_;
    std::vector<std::pair<int, int>> vbadkflmids, vbadstereokflmids;
    for (size_t i = 0; i < pb.res_type.size(); ++i) {
        if (!res.outlier || !res.outlier[i]) continue;
        const int kfid = pb.pose_kfid[pb.res_pose[i]], lmid = pb.lm_lmid[pb.res_lm[i]];
        const int t = pb.res_type[i];
        if (t == OV2_BA_L_XYZ || t == OV2_BA_L_INV) vbadkflmids.emplace_back(kfid, lmid);
        else vbadstereokflmids.emplace_back(kfid, lmid);
        pb.set_badlmids.insert(lmid);
    }
    for (const auto &b : vbadstereokflmids) {   // :743-751
        auto it = pb.map_local_pkfs.find(b.first);
        if (it != pb.map_local_pkfs.end()) { it->second->removeStereoKeypointById(b.second); pmap_->touchStereoOff(b.first, b.second); }
        pb.set_badlmids.insert(b.second);
    }
    for (const auto &b : vbadkflmids) {         // :753-764
        auto it = pb.map_local_pkfs.find(b.first);
        if (it != pb.map_local_pkfs.end()) pmap_->removeMapPointObs(b.second, b.first);
        if (cur_frame_obs && pmap_->pcurframe_ && b.first == pmap_->pcurframe_->kfid_) pmap_->removeObsFromCurFrameById(b.second);
        pb.set_badlmids.insert(b.second);
    }
    for (const auto &kv : pb.map_local_pkfs) {   // :767-786 poses of the non-constant keyframes
        if (pb.set_cstkfids.count(kv.first) || !kv.second) continue;
        const int idx = pb.kfid_to_pose.at(kv.first);
        SE3 T;
        for (int k = 0; k < 7; ++k) T.v[k] = pb.pose[7 * idx + k];
        kv.second->setTwc(T);
        pmap_->touchPose(kv.first);
    }
    for (const auto &kv : pb.map_local_plms) {   // :789-853 landmarks
        const int lmid = kv.first;
        auto plm = kv.second;
        if (!plm) { pb.set_badlmids.erase(lmid); continue; }
        if (plm->isBad()) { pmap_->removeMapPoint(lmid); pb.set_badlmids.erase(lmid); continue; }
        if (plm->getKfObsSet().size() < 3) {
            if (plm->kfid_ < newframe.kfid_ - 3 && !plm->isobs_) { pmap_->removeMapPoint(lmid); pb.set_badlmids.erase(lmid); continue; }
        }
        auto lit = pb.lmid_to_lm.find(lmid);
        if (lit == pb.lmid_to_lm.end()) { pb.set_badlmids.insert(lmid); continue; }
        if (inv) {
            const double rho = pb.lm[lit->second];
            const double zanch = 1. / rho;
            if (zanch <= 0.) { pmap_->removeMapPoint(lmid); pb.set_badlmids.erase(lmid); continue; }
            auto it = pb.map_local_pkfs.find(plm->kfid_);   // the MapPoint's own anchor keyframe (:822)
            if (it == pb.map_local_pkfs.end() || !it->second) { pb.set_badlmids.insert(lmid); continue; }
            auto pkfanch = it->second;
            const Keypoint kp = pkfanch->getKeypointById(lmid);
            const CameraCalibration &c = *pkfanch->pcalib_leftcam_;
            const Vec3 cam{zanch * (kp.unpx_.x - c.cx_) / c.fx_, zanch * (kp.unpx_.y - c.cy_) / c.fy_, zanch};
            pmap_->updateMapPoint(lmid, pkfanch->getTwc() * cam, rho);
        } else {
            pmap_->updateMapPoint(lmid, Vec3{pb.lm[3 * lit->second], pb.lm[3 * lit->second + 1], pb.lm[3 * lit->second + 2]});
        }
    }
    for (int lmid : pb.set_badlmids) {   // :856-882 culling
        std::shared_ptr<MapPoint> plm;
        auto it = pb.map_local_plms.find(lmid);
        plm = (it == pb.map_local_plms.end()) ? pmap_->getMapPoint(lmid) : it->second;
        if (!plm) continue;
        if (plm->isBad()) pmap_->removeMapPoint(lmid);
        else if (plm->getKfObsSet().size() < 3 && plm->kfid_ < newframe.kfid_ - 3 && !plm->isobs_) pmap_->removeMapPoint(lmid);
    }
    bstop_localba_ = false;   // :896
}

ov2_status Optimizer::localBA(Frame &newframe, const bool buse_robust_cost)
{
    LocalBAProblem pb;
    if (pmap_->dev_) {
        const ov2_status ss = setupLocalBADevice(newframe, pb);
        if (ss != OV2_OK) return ss;
    } else {
        setupLocalBA(newframe, pb);
    }
    if (pb.aborted || pb.res_type.empty()) { bstop_localba_ = false; return OV2_OK; }
    ov2_ba_problem p = pb.view(*pslamstate_, newframe);
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    if (!buse_robust_cost) o.huber_delta = 0.0;                       // loss_function->Reset(nullptr) :51-53
    o.l2_refine = (pslamstate_->apply_l2_after_robust_ && !stopLocalBA()) ? 1 : 0;   // :603-604
    std::vector<double> chi2(p.n_res);
    std::vector<uint8_t> depth(p.n_res), outlier(p.n_res);
    std::memset(&last_result_, 0, sizeof(last_result_));
    last_result_.chi2 = chi2.data(); last_result_.depth_positive = depth.data(); last_result_.outlier = outlier.data();
    ov2_status s;
    if (pb.has_dev) {
        // the set-up kernels left the flat problem in HBM: solve it there (ov2_ba_solve_batch_dev); only the solved states
        // and the flags the update stage walks come back
        const size_t R = (size_t)p.n_res, e = p.inv_depth ? 1 : 3;
        const size_t need = R * 8 + 2 * ((R + 15) & ~(size_t)15) + 64;
        if (need > dev_out_cap_) {
            if (dev_out_) ov2_dev_free(ctx_, dev_out_);
            dev_out_ = nullptr; dev_out_cap_ = 0;
            if ((s = ov2_dev_alloc(ctx_, need + need / 2, &dev_out_)) != OV2_OK) return s;
            dev_out_cap_ = need + need / 2;
        }
        unsigned char *d = (unsigned char *)dev_out_;
        ov2_ba_problem pd = p;
        const ov2_local_ba_setup &v = pb.dev_view;
        pd.pose = v.pose; pd.pose_const = v.pose_const; pd.lm = v.lm; pd.lm_anchor_pose = v.lm_anchor_pose; pd.lm_anchor_uv = v.lm_anchor_uv;
        pd.res_type = v.res_type; pd.res_pose = v.res_pose; pd.res_lm = v.res_lm; pd.res_uv = v.res_uv; pd.res_sigma = v.res_sigma;
        ov2_ba_result rd = last_result_;
        rd.chi2 = (double *)d; rd.depth_positive = d + R * 8; rd.outlier = d + R * 8 + ((R + 15) & ~(size_t)15);
        s = ov2_ba_solve_batch_dev(ctx_, 1, &pd, &o, &rd);
        if (s == OV2_OK) s = ov2_memcpy_d2h(ctx_, pb.pose.data(), v.pose, 7 * (size_t)p.n_pose * 8);
        if (s == OV2_OK) s = ov2_memcpy_d2h(ctx_, pb.lm.data(), v.lm, e * (size_t)p.n_lm * 8);
        if (s == OV2_OK && R) s = ov2_memcpy_d2h(ctx_, chi2.data(), rd.chi2, R * 8);
        if (s == OV2_OK && R) s = ov2_memcpy_d2h(ctx_, depth.data(), rd.depth_positive, R);
        if (s == OV2_OK && R) s = ov2_memcpy_d2h(ctx_, outlier.data(), rd.outlier, R);
        rd.chi2 = chi2.data(); rd.depth_positive = depth.data(); rd.outlier = outlier.data();
        last_result_ = rd;
    } else {
        s = ov2_ba_solve(ctx_, &p, &o, &last_result_);
    }
    if (s == OV2_OK) updateAfterLocalBA(newframe, pb, last_result_);
    last_result_.chi2 = nullptr; last_result_.depth_positive = nullptr; last_result_.outlier = nullptr;
    return s;
}

// ---- fullBA / looseBA: the same functors and solver over another selection of keyframes ----------------------------
void Optimizer::setupRangeBA(int kf_lo, int kf_hi, int kf_obs_max, size_t min_obs, LocalBAProblem &pb)
{   // src/optimizer.cpp:985-1270 (looseBA) / :1768-2026 (fullBA)
    const size_t nmincstkfs = pslamstate_->stereo_ ? 1 : 2;
    const bool inv = pslamstate_->buse_inv_depth_;
    auto add_pose = [&](int kfid, const std::shared_ptr<Frame> &pkf, bool cst) {
        const int idx = (int)pb.pose_const.size();
        pb.kfid_to_pose.emplace(kfid, idx);
        pb.pose_kfid.push_back(kfid);
        const SE3 T = pkf->getTwc();
        pb.pose.insert(pb.pose.end(), T.v.begin(), T.v.end());
        pb.pose_const.push_back(cst ? 1 : 0);
        pb.map_local_pkfs.emplace(kfid, pkf);
        if (cst) pb.set_cstkfids.insert(kfid);
    };
    std::set<int> set_lmids2opt;   // the reference's std::set: ascending lmid
    for (int kfid = kf_lo; kfid <= kf_hi; ++kfid) {
        auto pkf = pmap_->getKeyframe(kfid);
        if (!pkf) continue;
        if (pb.set_cstkfids.size() < nmincstkfs) add_pose(kfid, pkf, true);
        else {
            add_pose(kfid, pkf, false);
            for (const auto &kp : pkf->getKeypoints3d()) set_lmids2opt.insert(kp.lmid_);
        }
    }
    for (int lmid : set_lmids2opt) {
        auto plm = pmap_->getMapPoint(lmid);
        if (!plm || plm->isBad() || plm->getKfObsSet().size() < min_obs) { pb.set_badlmids.insert(lmid); continue; }
        pb.map_local_plms.emplace(lmid, plm);
        int lmidx = -1;
        if (!inv) {
            lmidx = (int)pb.lm_lmid.size();
            pb.lmid_to_lm.emplace(lmid, lmidx);
            pb.lm_lmid.push_back(lmid);
            const Vec3 p = plm->getPoint();
            pb.lm.push_back(p.x); pb.lm.push_back(p.y); pb.lm.push_back(p.z);
            pb.lm_anchor_pose.push_back(-1); pb.lm_anchor_uv.push_back(0); pb.lm_anchor_uv.push_back(0);
        }
        int kfanchid = -1;
        for (int kfid : plm->getKfObsSet()) {
            if (kfid > kf_obs_max) continue;
            std::shared_ptr<Frame> pkf;
            auto pkfit = pb.map_local_pkfs.find(kfid);
            if (pkfit == pb.map_local_pkfs.end()) {   // observers outside the range enter as constants
                pkf = pmap_->getKeyframe(kfid);
                if (!pkf) { pmap_->removeMapPointObs(lmid, kfid); continue; }
                add_pose(kfid, pkf, true);
            } else {
                pkf = pkfit->second;
            }
            const Keypoint kp = pkf->getKeypointById(lmid);
            if (kp.lmid_ != lmid) { pmap_->removeMapPointObs(lmid, kfid); continue; }
            const double sigma = std::pow(2., kp.scale_);
            auto add_res = [&](int type, const Point2f &uv) {
                pb.res_type.push_back((uint8_t)type);
                pb.res_pose.push_back(pb.kfid_to_pose.at(kfid));
                pb.res_lm.push_back(lmidx);
                pb.res_uv.push_back(uv.x); pb.res_uv.push_back(uv.y);
                pb.res_sigma.push_back(sigma);
            };
            if (inv && kfanchid < 0) {
                kfanchid = kfid;
                const double zanch = (pkf->getTcw() * plm->getPoint()).z;
                lmidx = (int)pb.lm_lmid.size();
                pb.lmid_to_lm.emplace(lmid, lmidx);
                pb.lm_lmid.push_back(lmid);
                pb.lm.push_back(1. / zanch);
                pb.lm_anchor_pose.push_back(pb.kfid_to_pose.at(kfid));
                pb.lm_anchor_uv.push_back(kp.unpx_.x); pb.lm_anchor_uv.push_back(kp.unpx_.y);
                if (kp.is_stereo_) { add_res(OV2_BA_RANCH_INV, kp.runpx_); pb.nbstereo++; }
                else pb.nbmono++;
                continue;
            }
            add_res(inv ? OV2_BA_L_INV : OV2_BA_L_XYZ, kp.unpx_);
            if (kp.is_stereo_) { add_res(inv ? OV2_BA_R_INV : OV2_BA_R_XYZ, kp.runpx_); pb.nbstereo++; }
            else pb.nbmono++;
        }
    }
}

ov2_status Optimizer::fullBA(const bool buse_robust_cost)
{
    auto pkf0 = pmap_->getKeyframe(0);   // Frame &newframe = *pmap_->getKeyframe(0)  (:1680)
    if (!pkf0) return OV2_ERR_INVALID;
    int nkfid = -1;
    for (const auto &kv : pmap_->map_pkfs_) nkfid = std::max(nkfid, kv.first);
    LocalBAProblem pb;
    setupRangeBA(0, nkfid, INT_MAX, 3, pb);   // no observer filter (:1830), landmarks need 3 observers (:1813)
    if (pb.res_type.empty()) return OV2_OK;
    ov2_ba_problem p = pb.view(*pslamstate_, *pkf0);
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    if (!buse_robust_cost) o.huber_delta = 0.0;
    o.max_iters = 100; o.l2_max_iters = 100;     // options.max_num_iterations = 100, reused by the refinement (:2056, :2150)
    o.function_tolerance = 1e-6;                 // Ceres default: fullBA sets none
    o.l2_refine = pslamstate_->apply_l2_after_robust_ ? 1 : 0;
    std::vector<double> chi2(p.n_res);
    std::vector<uint8_t> depth(p.n_res), outlier(p.n_res);
    std::memset(&last_result_, 0, sizeof(last_result_));
    last_result_.chi2 = chi2.data(); last_result_.depth_positive = depth.data(); last_result_.outlier = outlier.data();
    const ov2_status s = ov2_ba_solve(ctx_, &p, &o, &last_result_);
    if (s == OV2_OK) updateAfterLocalBA(*pkf0, pb, last_result_, false);   // same update stage (:2232-2330); kfid_ - 3 < 0: no age culling
    last_result_.chi2 = nullptr; last_result_.depth_positive = nullptr; last_result_.outlier = nullptr;
    return s;
}

ov2_status Optimizer::looseBA(int inikfid, const int nkfid, const bool buse_robust_cost)
{
    auto pnew = pmap_->getKeyframe(nkfid);
    if (!pnew) return OV2_ERR_INVALID;
    Frame &newframe = *pnew;
    LocalBAProblem pb;
    setupRangeBA(inikfid, nkfid, newframe.kfid_, 0, pb);   // observers younger than the loop keyframe are left out (:1056)
    if (pb.res_type.empty() || !pb.kfid_to_pose.count(newframe.kfid_)) return OV2_OK;
    ov2_ba_problem p = pb.view(*pslamstate_, newframe);
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    if (!buse_robust_cost) o.huber_delta = 0.0;
    o.max_iters = 5; o.function_tolerance = 1e-4;   // :1298-1299
    o.l2_refine = 0;                                // one solve; the flags only pick what is removed
    std::vector<double> chi2(p.n_res);
    std::vector<uint8_t> depth(p.n_res), outlier(p.n_res);
    std::memset(&last_result_, 0, sizeof(last_result_));
    last_result_.chi2 = chi2.data(); last_result_.depth_positive = depth.data(); last_result_.outlier = outlier.data();
    const ov2_status s = ov2_ba_solve(ctx_, &p, &o, &last_result_);
    if (s != OV2_OK) { last_result_.chi2 = nullptr; last_result_.depth_positive = nullptr; last_result_.outlier = nullptr; return s; }
    const bool inv = pslamstate_->buse_inv_depth_;
    // flags (:1330-1426)
    std::vector<std::pair<int, int>> vbadkflmids, vbadstereokflmids;
    for (size_t i = 0; i < pb.res_type.size(); ++i) {
        if (!outlier[i]) continue;
        const int kfid = pb.pose_kfid[pb.res_pose[i]], lmid = pb.lm_lmid[pb.res_lm[i]];
        const int t = pb.res_type[i];
        if (t == OV2_BA_L_XYZ || t == OV2_BA_L_INV) vbadkflmids.emplace_back(kfid, lmid);
        else vbadstereokflmids.emplace_back(kfid, lmid);
        pb.set_badlmids.insert(lmid);
    }
    last_result_.chi2 = nullptr; last_result_.depth_positive = nullptr; last_result_.outlier = nullptr;
    // new states (:1432-1526)
    const SE3 iniTnewkfw = newframe.getTcw();
    SE3 optTwnewkf;
    { const int idx = pb.kfid_to_pose.at(newframe.kfid_); for (int k = 0; k < 7; ++k) optTwnewkf.v[k] = pb.pose[7 * idx + k]; }
    std::vector<std::pair<int, Vec3>> vlm;
    for (const auto &kv : pb.map_local_plms) {
        const int lmid = kv.first;
        auto plm = kv.second;
        if (!plm) continue;
        if (plm->isBad()) { pb.set_badlmids.insert(lmid); continue; }
        auto lit = pb.lmid_to_lm.find(lmid);
        if (lit == pb.lmid_to_lm.end()) { pb.set_badlmids.insert(lmid); continue; }
        if (inv) {
            const double zanch = 1. / pb.lm[lit->second];
            if (zanch <= 0.) { pb.set_badlmids.insert(lmid); continue; }
            auto it = pb.map_local_pkfs.find(plm->kfid_);
            if (it == pb.map_local_pkfs.end() || !it->second) { pb.set_badlmids.insert(lmid); continue; }
            auto pkfanch = it->second;
            const Keypoint kp = pkfanch->getKeypointById(lmid);
            const CameraCalibration &c = *pkfanch->pcalib_leftcam_;
            const Vec3 cam{zanch * (kp.unpx_.x - c.cx_) / c.fx_, zanch * (kp.unpx_.y - c.cy_) / c.fy_, zanch};
            vlm.emplace_back(lmid, pkfanch->getTwc() * cam);   // with the keyframe's pose BEFORE the update, as the reference does
        } else {
            vlm.emplace_back(lmid, Vec3{pb.lm[3 * lit->second], pb.lm[3 * lit->second + 1], pb.lm[3 * lit->second + 2]});
        }
    }
    // update (:1533-1548)
    for (const auto &e : vlm) pmap_->updateMapPoint(e.first, e.second);
    for (const auto &kv : pb.map_local_pkfs) {
        if (pb.set_cstkfids.count(kv.first) || !kv.second) continue;
        const int idx = pb.kfid_to_pose.at(kv.first);
        SE3 T;
        for (int k = 0; k < 7; ++k) T.v[k] = pb.pose[7 * idx + k];
        kv.second->setTwc(T);
        pmap_->touchPose(kv.first);
    }
    // propagate the correction of the loop keyframe to the younger keyframes and the landmarks they anchor (:1552-1593)
    int nkfid_max = -1;
    for (const auto &kv : pmap_->map_pkfs_) nkfid_max = std::max(nkfid_max, kv.first);
    std::unordered_set<int> uplmid_set;
    for (int kfid = newframe.kfid_ + 1; kfid <= nkfid_max; ++kfid) {
        if (pb.map_local_pkfs.count(kfid)) continue;
        auto pkf = pmap_->getKeyframe(kfid);
        if (!pkf) continue;
        const SE3 updTwkf = optTwnewkf * (iniTnewkfw * pkf->getTwc());
        for (const auto &kp : pkf->getKeypoints3d()) {
            if (uplmid_set.count(kp.lmid_) || pb.map_local_plms.count(kp.lmid_)) continue;
            auto plm = pmap_->getMapPoint(kp.lmid_);
            if (!plm) { pmap_->removeMapPointObs(kp.lmid_, kfid); continue; }
            if (plm->kfid_ == kfid) {
                pmap_->updateMapPoint(kp.lmid_, updTwkf * pkf->projWorldToCam(plm->getPoint()));
                uplmid_set.insert(plm->lmid_);
            }
        }
        pkf->setTwc(updTwkf);
        pmap_->touchPose(kfid);
    }
    // bad observations (:1596-1620), culling (:1623-1650)
    for (const auto &b : vbadstereokflmids) {
        auto it = pb.map_local_pkfs.find(b.first);
        if (it != pb.map_local_pkfs.end()) { it->second->removeStereoKeypointById(b.second); pmap_->touchStereoOff(b.first, b.second); }
    }
    for (const auto &b : vbadkflmids) {
        if (pb.map_local_pkfs.count(b.first)) pmap_->removeMapPointObs(b.second, b.first);
        if (pmap_->pcurframe_ && b.first == pmap_->pcurframe_->kfid_) pmap_->removeObsFromCurFrameById(b.second);
    }
    for (int lmid : pb.set_badlmids) {
        auto it = pb.map_local_plms.find(lmid);
        std::shared_ptr<MapPoint> plm = (it == pb.map_local_plms.end()) ? pmap_->getMapPoint(lmid) : it->second;
        if (!plm) continue;
        if (plm->isBad()) pmap_->removeMapPoint(lmid);
        else if (plm->getKfObsSet().size() < 3 && plm->kfid_ < newframe.kfid_ - 3 && !plm->isobs_) pmap_->removeMapPoint(lmid);
    }
    if (pmap_->pcurframe_) {   // :1653-1656
        auto pc = pmap_->pcurframe_;
        pc->setTwc(optTwnewkf * (iniTnewkfw * pc->getTwc()));
    }
    return OV2_OK;
}

// ---- pose graphs ------------------------------------------------------------------------------------------------
bool Optimizer::localPoseGraph(Frame &newframe, int kfloop_id, const SE3 &newTwc, ov2_status *status)
{
    if (status) *status = OV2_OK;
    const SE3 iniTcw = newframe.getTcw();
    int nkfid_max = -1;
    for (const auto &kv : pmap_->map_pkfs_) nkfid_max = std::max(nkfid_max, kv.first);
    auto ploopkf = pmap_->getKeyframe(kfloop_id);
    while (!ploopkf && kfloop_id < nkfid_max) ploopkf = pmap_->getKeyframe(++kfloop_id);   // :2368-2371
    if (!ploopkf) return false;
    std::vector<double> pose, Tij;
    std::vector<uint8_t> cst;
    std::vector<int32_t> ei, ej;
    std::vector<int> kfids;
    std::map<int, std::shared_ptr<Frame>> map_pkfs;
    auto push_pose = [&](int kfid, const SE3 &T, bool c) {
        kfids.push_back(kfid);
        pose.insert(pose.end(), T.v.begin(), T.v.end());
        cst.push_back(c ? 1 : 0);
        return (int)kfids.size() - 1;
    };
    auto push_edge = [&](int a, int b, const SE3 &T) { ei.push_back(a); ej.push_back(b); Tij.insert(Tij.end(), T.v.begin(), T.v.end()); };
    push_pose(kfloop_id, ploopkf->getTwc(), true);
    SE3 Tciw = ploopkf->getTcw();
    int ci = 0;
    const SE3 Tloop_new = Tciw * newTwc;                         // :2387
    for (int kfid = kfloop_id + 1; kfid <= newframe.kfid_; ++kfid) {   // :2389-2420
        auto pkf = pmap_->getKeyframe(kfid);
        if (!pkf) { if (kfid == newframe.kfid_) return false; continue; }
        map_pkfs.emplace(kfid, pkf);
        const SE3 Twcj = pkf->getTwc();
        const int j = push_pose(kfid, Twcj, false);
        push_edge(ci, j, Tciw * Twcj);
        Tciw = Twcj.inverse();
        ci = j;
    }
    if (kfids.size() < 2 || kfids.back() != newframe.kfid_) return false;
    push_edge(0, (int)kfids.size() - 1, Tloop_new);              // :2422-2425
    ov2_pg_problem P;
    P.n_pose = (int)kfids.size(); P.pose = pose.data(); P.pose_const = cst.data();
    P.n_edge = (int)ei.size(); P.edge_i = ei.data(); P.edge_j = ej.data(); P.T_ij = Tij.data();
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    o.max_iters = 10; o.function_tolerance = 1e-4;               // :2445-2446
    const ov2_status s = ov2_pose_graph_solve(ctx_, &P, &o, &last_pg_);
    if (status) *status = s;
    if (s != OV2_OK) return false;
    auto pose_of = [&](int idx) { SE3 T; for (int k = 0; k < 7; ++k) T.v[k] = pose[7 * (size_t)idx + k]; return T; };
    const SE3 newoptTwc = pose_of((int)kfids.size() - 1);
    {
        const double dx = newTwc.v[0] - newoptTwc.v[0], dy = newTwc.v[1] - newoptTwc.v[1], dz = newTwc.v[2] - newoptTwc.v[2];
        if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.3 && pslamstate_->stereo_) return false;   // :2468-2474: degenerate
    }
    // updated keyframes and the landmarks they anchor (:2487-2520)
    std::unordered_set<int> processed_lmids;
    std::vector<std::pair<int, Vec3>> vlm;
    std::vector<std::pair<int, SE3>> vkf;
    for (size_t i = 1; i < kfids.size(); ++i) {
        const int kfid = kfids[i];
        auto pkf = map_pkfs.at(kfid);
        const SE3 T = pose_of((int)i);
        vkf.emplace_back(kfid, T);
        for (const auto &kp : pkf->getKeypoints3d()) {
            const int lmid = kp.lmid_;
            if (processed_lmids.count(lmid)) continue;
            auto plm = pmap_->getMapPoint(lmid);
            if (!plm) { pmap_->removeMapPointObs(lmid, kfid); continue; }
            if (plm->kfid_ == kfid) {
                vlm.emplace_back(lmid, T * pkf->projWorldToCam(plm->getPoint()));
                processed_lmids.insert(lmid);
            }
        }
    }
    // the correction of the new keyframe carried to the younger keyframes and their landmarks (:2527-2560)
    for (int kfid = newframe.kfid_ + 1; kfid <= nkfid_max; ++kfid) {
        auto pkf = pmap_->getKeyframe(kfid);
        if (!pkf) continue;
        const SE3 updTwkf = newoptTwc * (iniTcw * pkf->getTwc());
        for (const auto &kp : pkf->getKeypoints3d()) {
            const int lmid = kp.lmid_;
            if (processed_lmids.count(lmid)) continue;
            auto plm = pmap_->getMapPoint(lmid);
            if (!plm) { pmap_->removeMapPointObs(lmid, kfid); continue; }
            if (plm->kfid_ == kfid) {
                pmap_->updateMapPoint(lmid, updTwkf * pkf->projWorldToCam(plm->getPoint()));
                processed_lmids.insert(lmid);
            }
        }
        pkf->setTwc(updTwkf);
        pmap_->touchPose(kfid);
    }
    for (const auto &e : vlm) pmap_->updateMapPoint(e.first, e.second);   // :2563-2577
    for (const auto &e : vkf) {
        auto pkf = pmap_->getKeyframe(e.first);
        if (pkf) { pkf->setTwc(e.second); pmap_->touchPose(e.first); }
    }
    if (pmap_->pcurframe_) {                                               // :2580-2583
        auto pc = pmap_->pcurframe_;
        pc->setTwc(newoptTwc * (iniTcw * pc->getTwc()));
    }
    return true;
}

bool Optimizer::fullPoseGraph(std::vector<SE3> &vTwc, const std::vector<SE3> &vTpc, const std::vector<bool> &viskf, ov2_status *status)
{
    if (status) *status = OV2_OK;
    const size_t n = vTwc.size();
    if (n == 0 || vTpc.size() != n || viskf.size() != n) return false;
    std::vector<double> pose(7 * n), Tij;
    std::vector<uint8_t> cst(n);
    std::vector<int32_t> ei, ej;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 7; ++k) pose[7 * i + k] = vTwc[i].v[k];
        cst[i] = viskf[i] ? 1 : 0;
        if (i == 0) continue;
        ei.push_back((int32_t)i - 1); ej.push_back((int32_t)i);            // :2806-2809
        Tij.insert(Tij.end(), vTpc[i].v.begin(), vTpc[i].v.end());
    }
    ov2_pg_problem P;
    P.n_pose = (int)n; P.pose = pose.data(); P.pose_const = cst.data();
    P.n_edge = (int)ei.size(); P.edge_i = ei.data(); P.edge_j = ej.data(); P.T_ij = Tij.data();
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    o.max_iters = 100; o.function_tolerance = 1e-6;                          // :2821-2824
    const ov2_status s = ov2_pose_graph_solve(ctx_, &P, &o, &last_pg_);
    if (status) *status = s;
    if (s != OV2_OK) return false;
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 7; ++k) vTwc[i].v[k] = pose[7 * i + k];
    return true;
}

ov2_status Optimizer::structureOnlyBA(const std::vector<int> &vlm2optids)
{   // src/optimizer.cpp:2594-2781: every keyframe pose constant, the listed map points free (XYZ), Huber loss, 10 LM
    // iterations, no chi2 flags / no L2 pass; the wall-clock cap (:2749) is not applied
    auto pkf0 = pmap_->getKeyframe(0);
    if (!pkf0) return OV2_OK;
    LocalBAProblem pb;
    for (const int lmid : vlm2optids) {
        auto plm = pmap_->getMapPoint(lmid);
        if (plm == nullptr) continue;
        const int il = (int)pb.lm_lmid.size();
        pb.lm_lmid.push_back(lmid);
        const Vec3 pt = plm->getPoint();
        pb.lm.push_back(pt.x); pb.lm.push_back(pt.y); pb.lm.push_back(pt.z);
        for (const int kfid : plm->getKfObsSet()) {
            auto pkf = pmap_->getKeyframe(kfid);
            if (pkf == nullptr) continue;
            const Keypoint kp = pkf->getKeypointById(lmid);
            if (kp.lmid_ != lmid) continue;
            auto it = pb.kfid_to_pose.find(kfid);
            if (it == pb.kfid_to_pose.end()) {
                it = pb.kfid_to_pose.emplace(kfid, (int)pb.pose_kfid.size()).first;
                pb.pose_kfid.push_back(kfid);
                const SE3 T = pkf->getTwc();
                pb.pose.insert(pb.pose.end(), T.v.begin(), T.v.end());
                pb.pose_const.push_back(1);                                            // SetParameterBlockConstant :2677
            }
            const double sigma = std::pow(2., kp.scale_);
            pb.res_type.push_back(OV2_BA_L_XYZ); pb.res_pose.push_back(it->second); pb.res_lm.push_back(il);
            pb.res_uv.push_back(kp.unpx_.x); pb.res_uv.push_back(kp.unpx_.y); pb.res_sigma.push_back(sigma);
            if (kp.is_stereo_) {                                                       // :2682-2703
                pb.res_type.push_back(OV2_BA_R_XYZ); pb.res_pose.push_back(it->second); pb.res_lm.push_back(il);
                pb.res_uv.push_back(kp.runpx_.x); pb.res_uv.push_back(kp.runpx_.y); pb.res_sigma.push_back(sigma);
            }
        }
    }
    if (pb.res_type.empty()) return OV2_OK;
    SlamParams st = *pslamstate_;
    st.buse_inv_depth_ = false;
    ov2_ba_problem p = pb.view(st, *pkf0);
    p.lm_anchor_pose = nullptr; p.lm_anchor_uv = nullptr;
    ov2_ba_options o;
    ov2_ba_default_options(&o, pslamstate_->robust_mono_th_);
    o.max_iters = 10;                                                                  // :2747
    o.l2_refine = 0;
    std::memset(&last_result_, 0, sizeof(last_result_));
    const ov2_status s = ov2_ba_solve(ctx_, &p, &o, &last_result_);
    if (s != OV2_OK) return s;
    for (size_t i = 0; i < pb.lm_lmid.size(); ++i)                                     // :2769-2777
        pmap_->updateMapPoint(pb.lm_lmid[i], Vec3{pb.lm[3 * i], pb.lm[3 * i + 1], pb.lm[3 * i + 2]});
    return OV2_OK;
}

ov2_status Estimator::applyLocalBA()
{   // src/estimator.cpp:67-98
    const int nmincstkfs = pslamstate_->mono_ ? 2 : 1;
    if (!pnewkf_ || pnewkf_->kfid_ < nmincstkfs) return OV2_OK;
    if (pnewkf_->nb3dkps_ == 0) return OV2_OK;
    pslamstate_->blocalba_is_on_ = true;
    const ov2_status s = poptimizer_->localBA(*pnewkf_, true);
    pslamstate_->blocalba_is_on_ = false;
    return s;
}

}  // namespace ov2
