// Flat device-resident map mirror + the set-up stage of Optimizer::localBA as linear scans.
//
// The reference keeps the map as hash maps of shared_ptr<Frame> / shared_ptr<MapPoint> (include/map_manager.hpp:41-129,
// include/frame.hpp mapkps_/map_covkfs_, include/map_point.hpp set_kfids_) and assembles every local BA by walking
// them (src/optimizer.cpp:43-430): ~4.4 ms for 40 keyframes / 3 k landmarks / 38 k residual blocks in the C++ mirror
// of that walk (ov2slam_amd/host), the same order as the solve itself.  Here the map is three SoA tables in HBM
// (keyframes, landmarks, observations; kfid and lmid index them directly; the tables grow on demand) and the set-up is a dozen scans of the
// observation table -- HBM-bound integer work: at 8 TB/s a million observations (40 B each) are a 5 us read, so no
// per-landmark adjacency needs to be maintained incrementally, the table itself is the adjacency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "ov2_internal.h"

struct ov2_map {
    ov2_ctx *c;
    int max_kf, max_lm, max_obs, n_obs;
    int n_compactions;                                              // times the observation table was squeezed
    // tables
    double *kf_pose; unsigned char *kf_state;                       // [max_kf]
    double *lm_xyz; unsigned char *lm_state;                        // [max_lm]
    int *obs_kf, *obs_lm, *obs_scale; double *obs_uv, *obs_ruv; unsigned char *obs_flag;   // [max_obs]
    // set-up scratch
    int *hdr;                                                       // MH_N ints
    int *cov, *kf_role, *kf_idx;                                    // [max_kf]
    int *lm_nobs, *lm_sel, *lm_anchor, *lm_flag;                    // [max_lm]; lm_sel: 0 none, 1 local, 2 bad
    unsigned long long *lm_pack, *lm_pidx;                          // [max_lm] (flag | bad << 32) and its exclusive scan: (lm index | bad index << 32)
    unsigned char *lm_new;                                          // [max_lm] observed by the new keyframe
    int *obs_cnt, *obs_off;                                         // [max_obs]
    int *blk;                                                       // block sums of the scans
    unsigned char *zero_blk; size_t zero_bytes;                     // hdr | cov | kf_role | lm_nobs | lm_sel | lm_anchor | lm_new: one memset per set-up
    // outputs: device image + pinned host image of the flat problem
    unsigned char *out_dev, *out_host;
    size_t out_cap;
    int *hdr_host;                                                  // pinned
};

namespace {

enum { MH_NBKPS = 0, MH_NB3D, MH_ABORT, MH_NMAXKF, MH_NPOSE, MH_NRES, MH_NLM, MH_NBAD, MH_NLIVE, MH_N = 16 };   // NLM|NBAD: one 64-bit scan total
#define ANCH_TOP 0x40000000   // anchor keyframe stored as ANCH_TOP - kfid under atomicMax: 0 = none, so the array lives in the zeroed block
enum { OBS_ALIVE = 1, OBS_STEREO = 2 };
enum { MAP_COMPACT_MIN_ROWS = 4096 };   // below this the scans cost nothing worth a reallocation

struct map_view {
    int max_kf, max_lm, n_obs;
    const double *kf_pose; const unsigned char *kf_state;
    const double *lm_xyz; const unsigned char *lm_state;
    const int *obs_kf, *obs_lm, *obs_scale; const double *obs_uv, *obs_ruv; const unsigned char *obs_flag;
};

// an observation counts when it, its keyframe and its landmark are alive (MapPoint::set_kfids_ / Frame::mapkps_ agree)
__device__ __forceinline__ bool obs_live(const map_view &M, int i, int &kf, int &lm)
{
    if (!(M.obs_flag[i] & OBS_ALIVE)) return false;
    kf = M.obs_kf[i]; lm = M.obs_lm[i];
    return M.kf_state[kf] && (M.lm_state[lm] & OV2_LM_ALIVE);
}

// ---- hooks ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void map_set_lm_kernel(int n, const int *__restrict__ lmid, const double *__restrict__ xyz,
                                                         const unsigned char *__restrict__ state, double *__restrict__ lm_xyz,
                                                         unsigned char *__restrict__ lm_state)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = lmid[i];
    lm_state[l] = state[i];
    if (xyz) { lm_xyz[3 * l] = xyz[3 * i]; lm_xyz[3 * l + 1] = xyz[3 * i + 1]; lm_xyz[3 * l + 2] = xyz[3 * i + 2]; }
}

__global__ __launch_bounds__(256) void map_set_pose_kernel(int n, const int *__restrict__ kfid, const double *__restrict__ T,
                                                           double *__restrict__ kf_pose)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 7) return;
    kf_pose[7 * kfid[i / 7] + i % 7] = T[i];
}

// (kfid, lmid) pairs are few per call: every observation checks the list (list in LDS by chunks of 256)
// mode 0: kill the observation; mode 1: set / clear the stereo flag (+ runpx)
__global__ __launch_bounds__(256) void map_edit_obs_kernel(int n_obs, const int *__restrict__ obs_kf, const int *__restrict__ obs_lm,
                                                           unsigned char *__restrict__ obs_flag, double *__restrict__ obs_ruv,
                                                           int n, const int *__restrict__ kfid, const int *__restrict__ lmid,
                                                           int mode, const unsigned char *__restrict__ st,
                                                           const double *__restrict__ ruv)
{
    __shared__ int sk[256], sl[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_obs && (obs_flag[i] & OBS_ALIVE);
    const int k = live ? obs_kf[i] : -1, l = live ? obs_lm[i] : -1;
    for (int base = 0; base < n; base += 256) {
        const int m = min(256, n - base);
        __syncthreads();
        if ((int)threadIdx.x < m) { sk[threadIdx.x] = kfid[base + threadIdx.x]; sl[threadIdx.x] = lmid[base + threadIdx.x]; }
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < m; ++j)
            if (sl[j] == l && (sk[j] == k || sk[j] < 0)) {   // kfid < 0: every keyframe (removeMapPoint)
                if (mode == 0) obs_flag[i] = 0;
                else {
                    if (st[base + j]) {
                        obs_flag[i] |= OBS_STEREO;
                        obs_ruv[2 * i] = ruv[2 * (base + j)]; obs_ruv[2 * i + 1] = ruv[2 * (base + j) + 1];
                    } else obs_flag[i] &= ~OBS_STEREO;
                }
            }
    }
}

// ---- set-up scans -----------------------------------------------------------------------------------
// observers per landmark, landmarks of the new keyframe, its keypoint counts (Frame::nbkps_, nb3dkps_)
__global__ __launch_bounds__(256) void ms_count_kernel(map_view M, int newkf, int *__restrict__ lm_nobs,
                                                       unsigned char *__restrict__ lm_new, int *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf = 0, lm = 0;
    const bool live = i < M.n_obs && obs_live(M, i, kf, lm);
    // live rows of the table (one atomic per wave): the host compacts the table when most rows are dead
    const unsigned long long bal = __ballot(live);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&hdr[MH_NLIVE], __popcll(bal));
    if (!live) return;
    atomicAdd(&lm_nobs[lm], 1);
    if (kf == newkf) {
        lm_new[lm] = 1;
        atomicAdd(&hdr[MH_NBKPS], 1);
        if (M.lm_state[lm] & OV2_LM_KP3D) atomicAdd(&hdr[MH_NB3D], 1);
    }
}

// ---- compaction of the observation table ------------------------------------------------------------
// Observations are appended and killed in place (tombstones), so a long sequence leaves mostly dead rows behind
// (the reference erases them from its hash maps: src/map_manager.cpp:885-1019).  Dead rows (flag cleared, or keyframe /
// landmark gone -- neither id is ever reused) are squeezed out by a stable scatter: the row order, and so every set-up
// result, stays what it was.
__global__ __launch_bounds__(256) void mc_mark_kernel(map_view M, int *__restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.n_obs) return;
    int kf, lm;
    keep[i] = obs_live(M, i, kf, lm) ? 1 : 0;
}

struct obs_cols { int *kf, *lm, *scale; double *uv, *ruv; unsigned char *flag; };

__global__ __launch_bounds__(256) void mc_scatter_kernel(map_view M, const int *__restrict__ keep, const int *__restrict__ pos, obs_cols O)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.n_obs || !keep[i]) return;
    const int j = pos[i];
    O.kf[j] = M.obs_kf[i]; O.lm[j] = M.obs_lm[i]; O.scale[j] = M.obs_scale[i]; O.flag[j] = M.obs_flag[i];
    reinterpret_cast<double2 *>(O.uv)[j] = reinterpret_cast<const double2 *>(M.obs_uv)[i];
    reinterpret_cast<double2 *>(O.ruv)[j] = reinterpret_cast<const double2 *>(M.obs_ruv)[i];
}

// MapManager::updateFrameCovisibility (src/map_manager.cpp:117-193): co-observed landmarks per other keyframe
__global__ __launch_bounds__(256) void ms_cov_kernel(map_view M, int newkf, const unsigned char *__restrict__ lm_new,
                                                     int *__restrict__ cov)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    if (kf != newkf && lm_new[lm]) atomicAdd(&cov[kf], 1);
}

// src/optimizer.cpp:61-63,128-190: one wave walks the covisible keyframes newest -> oldest, 64 per step.
// kf_role: 0 = not in the problem, 1 = optimised, 2 = constant
__global__ __launch_bounds__(64) void ms_select_kernel(map_view M, int newkf, int nmin_cov, const int *__restrict__ cov,
                                                       int *__restrict__ kf_role, int *__restrict__ hdr)
{
    const int lane = threadIdx.x;
    const int nb3d = hdr[MH_NB3D], nbkps = hdr[MH_NBKPS];
    if (nb3d < nmin_cov) { if (lane == 0) hdr[MH_ABORT] = 1; return; }
    bool all_cst = false;
    int nmax = -1;
    for (int hi = M.max_kf - 1; hi >= 0; hi -= 64) {
        const int kf = hi - lane;
        bool in = false, good = false;
        if (kf >= 0 && M.kf_state[kf]) {
            int score = (kf == newkf) ? nb3d : cov[kf];
            in = (kf == newkf) || score > 0;
            if (kf > newkf) score = nbkps;
            good = score >= nmin_cov && kf > 0;
        }
        const unsigned long long min_ = __ballot(in), bad_ = __ballot(in && !good);
        if (nmax < 0 && min_) nmax = hi - (__ffsll((long long)min_) - 1);   // lane 0 holds the largest kfid of the step
        // lanes after (older than) the first failing one are constant too
        bool cst = all_cst;
        if (bad_) {
            const int first_bad = __ffsll((long long)bad_) - 1;
            if (lane >= first_bad) cst = true;
            all_cst = true;
        }
        if (in) kf_role[kf] = cst ? 2 : 1;
    }
    if (lane == 0) hdr[MH_NMAXKF] = nmax;
}

// landmarks of the optimised keyframes' 3D keypoints (:176-180) and MapPoint::isBad (src/map_point.cpp:215-234)
__global__ __launch_bounds__(256) void ms_local_lm_kernel(map_view M, const int *__restrict__ kf_role,
                                                          const int *__restrict__ lm_nobs, int *__restrict__ lm_sel,
                                                          const int *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (hdr[MH_ABORT] || i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    const int st = M.lm_state[lm];
    if (kf_role[kf] != 1 || !(st & OV2_LM_KP3D)) return;
    const int nobs = lm_nobs[lm];
    const bool isobs = st & OV2_LM_OBS;
    const bool bad = (nobs < 2 && !isobs && (st & OV2_LM_3D)) || (nobs == 0 && !isobs);
    lm_sel[lm] = bad ? 2 : 1;
}

// observers of the local landmarks: outside keyframes become constant poses (:229-246), the first observer anchors
// the landmark (:251-287)
__global__ __launch_bounds__(256) void ms_observers_kernel(map_view M, const int *__restrict__ lm_sel, int *__restrict__ kf_role,
                                                           int *__restrict__ lm_anchor, const int *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (hdr[MH_ABORT] || i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    if (lm_sel[lm] != 1 || kf > hdr[MH_NMAXKF]) return;
    if (kf_role[kf] == 0) kf_role[kf] = 2;     // every writer stores the same value
    atomicMax(&lm_anchor[lm], ANCH_TOP - kf);   // the smallest kfid wins
}

// gauge (:394-407) + dense pose numbering in ascending kfid.  One workgroup.
__global__ __launch_bounds__(1024) void ms_poses_kernel(int max_kf, int nmin_cst, int *__restrict__ kf_role, int *__restrict__ kf_idx,
                                                        int *__restrict__ hdr)
{
    __shared__ int wsum[16], carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (hdr[MH_ABORT]) return;
    // constants so far
    int ncst = 0;
    for (int k = tid; k < max_kf; k += 1024) ncst += kf_role[k] == 2;
    for (int o = 32; o; o >>= 1) ncst += __shfl_xor(ncst, o);
    if (lane == 0) wsum[wv] = ncst;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        // not enough constants: the smallest optimised kfids are fixed (serial: at most nmin_cst <= 2 hits)
        for (int k = 0; k < max_kf && t < nmin_cst; ++k)
            if (kf_role[k] == 1) { kf_role[k] = 2; ++t; }
        carry_s = 0;
    }
    __syncthreads();
    for (int base = 0; base < max_kf; base += 1024) {
        const int k = base + tid;
        const int f = (k < max_kf && kf_role[k] != 0) ? 1 : 0;
        int x = f;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (k < max_kf) kf_idx[k] = f ? off + x - 1 : -1;
        __syncthreads();
        if (tid == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (tid == 0) hdr[MH_NPOSE] = carry_s;
}

// flags of the landmark numbering: local and (with inverse depth) anchored; bad ones go to their own list.  Both flags
// ride one 64-bit word so that one scan numbers both lists.
__global__ __launch_bounds__(256) void ms_lm_flags_kernel(int max_lm, int inv, const int *__restrict__ lm_sel,
                                                          const int *__restrict__ lm_anchor, int *__restrict__ lm_flag,
                                                          unsigned long long *__restrict__ lm_pack)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= max_lm) return;
    const int s = lm_sel[l];
    const int f = (s == 1 && (!inv || lm_anchor[l] != 0)) ? 1 : 0;
    lm_flag[l] = f;
    lm_pack[l] = (unsigned long long)f | ((unsigned long long)(s == 2) << 32);
}

// residual blocks per observation (:251-391): anchor observation 1 if stereo (right-anchor block) else 0; any other
// observation 2 if stereo (left + right) else 1
__global__ __launch_bounds__(256) void ms_res_count_kernel(map_view M, int inv, const int *__restrict__ lm_flag,
                                                           const int *__restrict__ lm_anchor, int *__restrict__ obs_cnt,
                                                           const int *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.n_obs) return;
    int kf, lm, c = 0;
    if (!hdr[MH_ABORT] && obs_live(M, i, kf, lm) && lm_flag[lm] && kf <= hdr[MH_NMAXKF]) {
        const bool stereo = M.obs_flag[i] & OBS_STEREO;
        if (inv && lm_anchor[lm] == ANCH_TOP - kf) c = stereo ? 1 : 0;
        else c = stereo ? 2 : 1;
    }
    obs_cnt[i] = c;
}

// ---- exclusive scan of an array (int, or two counters packed in 64 bits) in three launches:
// block sums -> their scan -> block offsets
template <typename T>
__device__ __forceinline__ T shfl_up_t(T v, int o)
{
    if constexpr (sizeof(T) == 8) {
        const unsigned lo = (unsigned)__shfl_up((int)(v & 0xffffffffull), o), hi = (unsigned)__shfl_up((int)(v >> 32), o);
        return ((T)hi << 32) | lo;
    } else {
        return __shfl_up(v, o);
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_block_kernel(const T *__restrict__ in, int n, T *__restrict__ out, T *__restrict__ blk)
{
    __shared__ T wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const T v = i < n ? in[i] : (T)0;
    T x = v;
    for (int o = 1; o < 64; o <<= 1) { const T y = shfl_up_t(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    T off = 0;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (i < n) out[i] = off + x - v;
    if (tid == 1023) blk[blockIdx.x] = off + x;
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_top_kernel(T *__restrict__ blk, int nb, T *__restrict__ total)
{
    __shared__ T wsum[16], carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + tid;
        const T v = i < nb ? blk[i] : (T)0;
        T x = v;
        for (int o = 1; o < 64; o <<= 1) { const T y = shfl_up_t(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        T off = carry_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (i < nb) blk[i] = off + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_add_kernel(T *__restrict__ out, int n, const T *__restrict__ blk)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += blk[blockIdx.x];
}

// ---- emission into the flat problem -------------------------------------------------------------------
struct flat_out {
    int *pose_kfid; unsigned char *pose_const; double *pose;
    int *lm_lmid; double *lm; int *lm_anchor_pose; double *lm_anchor_uv;
    unsigned char *res_type; int *res_pose, *res_lm; double *res_uv, *res_sigma;
    int *bad_lmid;
};

__device__ __forceinline__ void me_poses(const map_view &M, int k, const int *__restrict__ kf_role, const int *__restrict__ kf_idx,
                                         const flat_out &O)
{
    if (k >= M.max_kf || kf_role[k] == 0) return;
    const int j = kf_idx[k];
    O.pose_kfid[j] = k;
    O.pose_const[j] = kf_role[k] == 2;
    for (int t = 0; t < 7; ++t) O.pose[7 * j + t] = M.kf_pose[7 * k + t];
}

__device__ __forceinline__ void me_lms(const map_view &M, int l, int inv, const int *__restrict__ lm_sel,
                                       const int *__restrict__ lm_flag, const unsigned long long *__restrict__ lm_pidx, const flat_out &O)
{
    if (l >= M.max_lm) return;
    const unsigned long long pi = lm_pidx[l];
    if (lm_sel[l] == 2) O.bad_lmid[(int)(pi >> 32)] = l;
    if (!lm_flag[l]) return;
    const int j = (int)(pi & 0xffffffffull);
    O.lm_lmid[j] = l;
    if (!inv) {
        for (int t = 0; t < 3; ++t) O.lm[3 * j + t] = M.lm_xyz[3 * l + t];
        O.lm_anchor_pose[j] = -1;
        O.lm_anchor_uv[2 * j] = O.lm_anchor_uv[2 * j + 1] = 0.0;
    }
}

// z of (Twc^-1 * p): third row of R' times (p - t); Twc = [t, qx qy qz qw] (Sophus, include/frame.hpp getTcw)
__device__ __forceinline__ double depth_in_kf(const double *T, const double *p)
{
    const double x = T[3], y = T[4], z = T[5], w = T[6];
    const double dx = p[0] - T[0], dy = p[1] - T[1], dz = p[2] - T[2];
    // third column of R(q) = third row of R'
    const double r02 = 2.0 * (x * z + w * y), r12 = 2.0 * (y * z - w * x), r22 = 1.0 - 2.0 * (x * x + y * y);
    return r02 * dx + r12 * dy + r22 * dz;
}

__device__ __forceinline__ void me_res(const map_view &M, int i, int inv, const int *__restrict__ lm_flag,
                                       const unsigned long long *__restrict__ lm_pidx, const int *__restrict__ lm_anchor,
                                       const int *__restrict__ kf_idx, const int *__restrict__ obs_off, const int *__restrict__ hdr,
                                       const flat_out &O)
{
    if (i >= M.n_obs) return;
    int kf, lm;
    if (hdr[MH_ABORT] || !obs_live(M, i, kf, lm) || !lm_flag[lm] || kf > hdr[MH_NMAXKF]) return;
    const int j = (int)(lm_pidx[lm] & 0xffffffffull), pj = kf_idx[kf];
    const bool stereo = M.obs_flag[i] & OBS_STEREO;
    const double sigma = (double)(1 << M.obs_scale[i]);   // std::pow(2., kp.scale_)
    int o = obs_off[i];
    auto put = [&](int type, const double *uv) {
        O.res_type[o] = (unsigned char)type; O.res_pose[o] = pj; O.res_lm[o] = j;
        O.res_uv[2 * o] = uv[0]; O.res_uv[2 * o + 1] = uv[1]; O.res_sigma[o] = sigma;
        ++o;
    };
    if (inv && lm_anchor[lm] == ANCH_TOP - kf) {
        O.lm[j] = 1.0 / depth_in_kf(M.kf_pose + 7 * kf, M.lm_xyz + 3 * lm);
        O.lm_anchor_pose[j] = pj;
        O.lm_anchor_uv[2 * j] = M.obs_uv[2 * i]; O.lm_anchor_uv[2 * j + 1] = M.obs_uv[2 * i + 1];
        if (stereo) put(OV2_BA_RANCH_INV, M.obs_ruv + 2 * i);
        return;
    }
    put(inv ? OV2_BA_L_INV : OV2_BA_L_XYZ, M.obs_uv + 2 * i);
    if (stereo) put(inv ? OV2_BA_R_INV : OV2_BA_R_XYZ, M.obs_ruv + 2 * i);
}

// one launch: blocks [0, gK) write the poses, [gK, gK + gL) the landmarks and the bad list, the rest the residual blocks
__global__ __launch_bounds__(256) void me_all_kernel(map_view M, int gK, int gL, int inv, const int *__restrict__ kf_role,
                                                     const int *__restrict__ kf_idx, const int *__restrict__ lm_sel,
                                                     const int *__restrict__ lm_flag, const unsigned long long *__restrict__ lm_pidx,
                                                     const int *__restrict__ lm_anchor, const int *__restrict__ obs_off,
                                                     const int *__restrict__ hdr, flat_out O)
{
    const int blk = blockIdx.x, t = threadIdx.x;
    if (blk < gK) me_poses(M, blk * 256 + t, kf_role, kf_idx, O);
    else if (blk < gK + gL) me_lms(M, (blk - gK) * 256 + t, inv, lm_sel, lm_flag, lm_pidx, O);
    else me_res(M, (blk - gK - gL) * 256 + t, inv, lm_flag, lm_pidx, lm_anchor, kf_idx, obs_off, hdr, O);
}

map_view view_of(const ov2_map *m)
{
    map_view v;
    v.max_kf = m->max_kf; v.max_lm = m->max_lm; v.n_obs = m->n_obs;
    v.kf_pose = m->kf_pose; v.kf_state = m->kf_state; v.lm_xyz = m->lm_xyz; v.lm_state = m->lm_state;
    v.obs_kf = m->obs_kf; v.obs_lm = m->obs_lm; v.obs_scale = m->obs_scale; v.obs_uv = m->obs_uv; v.obs_ruv = m->obs_ruv;
    v.obs_flag = m->obs_flag;
    return v;
}

template <typename T>
ov2_status exclusive_scan(ov2_map *m, const T *in, int n, T *out, T *total)
{
    ov2_ctx *c = m->c;
    const int nb = (n + 1023) / 1024;
    if (n <= 0) { OV2_HIP(c, hipMemsetAsync(total, 0, sizeof(T), c->stream)); return OV2_OK; }
    T *blk = reinterpret_cast<T *>(m->blk);
    OV2_LAUNCH(c, OV2_K_MAP, scan_block_kernel<T>, dim3(nb), dim3(1024), 0, c->stream, in, n, out, blk);
    OV2_LAUNCH(c, OV2_K_MAP, scan_top_kernel<T>, dim3(1), dim3(1024), 0, c->stream, blk, nb, total);
    OV2_LAUNCH(c, OV2_K_MAP, scan_add_kernel<T>, dim3(nb), dim3(1024), 0, c->stream, out, n, (const T *)blk);
    return OV2_OK;
}

template <typename T>
ov2_status dmalloc(ov2_ctx *c, T **p, size_t count)
{
    hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "map table hipMalloc(%zu): %s", count * sizeof(T), hipGetErrorString(e));
    return OV2_OK;
}

// stage `count` elements of each host array behind each other in the ctx staging block; returns device pointers
struct stager {
    ov2_ctx *c; unsigned char *h = nullptr, *d = nullptr; size_t off = 0, cap = 0;
    ov2_status begin(size_t bytes)
    {
        void *hh, *dd;
        ov2_status s = ov2_staging(c, bytes + 256, &hh, &dd);
        if (s != OV2_OK) return s;
        h = (unsigned char *)hh; d = (unsigned char *)dd; cap = bytes + 256; off = 0;
        return OV2_OK;
    }
    template <typename T> const T *put(const T *src, size_t count)
    {
        if (!src) return nullptr;
        off = (off + 15) & ~(size_t)15;
        memcpy(h + off, src, count * sizeof(T));
        const T *dp = reinterpret_cast<const T *>(d + off);
        off += count * sizeof(T);
        return dp;
    }
    ov2_status flush() { OV2_HIP(c, hipMemcpyAsync(d, h, off, hipMemcpyHostToDevice, c->stream)); return OV2_OK; }
};

}  // namespace

// every device array whose size follows the capacities (tables + set-up scratch); pointers must be null on entry
static ov2_status alloc_tables(ov2_map *m)
{
    ov2_ctx *c = m->c;
    ov2_status s = OV2_OK;
    const size_t K = m->max_kf, L = m->max_lm, N = m->max_obs;
#define A(p, n) if (s == OV2_OK) s = dmalloc(c, &m->p, (n))
    A(kf_pose, 7 * K); A(kf_state, K); A(lm_xyz, 3 * L); A(lm_state, L);
    A(obs_kf, N); A(obs_lm, N); A(obs_scale, N); A(obs_uv, 2 * N); A(obs_ruv, 2 * N); A(obs_flag, N);
    A(kf_idx, K); A(lm_flag, L); A(lm_pack, L); A(lm_pidx, L);
    m->zero_bytes = sizeof(int) * (MH_N + 2 * K + 3 * L) + L;
    A(zero_blk, m->zero_bytes);
    if (s == OV2_OK) {
        int *z = reinterpret_cast<int *>(m->zero_blk);
        m->hdr = z; m->cov = z + MH_N; m->kf_role = m->cov + K; m->lm_nobs = m->kf_role + K; m->lm_sel = m->lm_nobs + L;
        m->lm_anchor = m->lm_sel + L;
        m->lm_new = reinterpret_cast<unsigned char *>(m->lm_anchor + L);
    }
    A(obs_cnt, N); A(obs_off, N); A(blk, 2 * ((std::max(N, L) + 1023) / 1024 + 1));   // block sums, up to 64-bit
#undef A
    return s;
}

static void free_capacity_arrays(ov2_map *m)
{
    void *dev[] = {m->kf_pose, m->kf_state, m->lm_xyz, m->lm_state, m->obs_kf, m->obs_lm, m->obs_scale, m->obs_uv, m->obs_ruv,
                   m->obs_flag, m->zero_blk, m->kf_idx, m->lm_flag, m->lm_pack, m->lm_pidx, m->obs_cnt, m->obs_off, m->blk};
    for (void *p : dev) if (p) (void)hipFree(p);
}

// The tables grow by doubling when an id or the observation count passes the capacity: fresh arrays, device copies of
// the live prefixes, old arrays freed (a rare event; one synchronisation).
static ov2_status ensure_capacity(ov2_map *m, int need_kf, int need_lm, int need_obs)
{
    if (need_kf <= m->max_kf && need_lm <= m->max_lm && need_obs <= m->max_obs) return OV2_OK;
    ov2_ctx *c = m->c;
    ov2_map old = *m;
    auto grown = [](int have, int need) { return need <= have ? have : std::max(need, have + have / 2 + 16); };
    m->max_kf = grown(old.max_kf, need_kf); m->max_lm = grown(old.max_lm, need_lm); m->max_obs = grown(old.max_obs, need_obs);
    m->kf_pose = nullptr; m->kf_state = nullptr; m->lm_xyz = nullptr; m->lm_state = nullptr;
    m->obs_kf = m->obs_lm = m->obs_scale = nullptr; m->obs_uv = m->obs_ruv = nullptr; m->obs_flag = nullptr;
    m->zero_blk = nullptr; m->kf_idx = m->lm_flag = m->obs_cnt = m->obs_off = m->blk = nullptr; m->lm_pack = m->lm_pidx = nullptr;
    ov2_status s = alloc_tables(m);
    if (s != OV2_OK) {
        free_capacity_arrays(m);
        *m = old;
        return s;
    }
    hipStream_t st = c->stream;
    const size_t K = old.max_kf, L = old.max_lm, N = old.n_obs;
    OV2_HIP(c, hipMemsetAsync(m->kf_state, 0, (size_t)m->max_kf, st));
    OV2_HIP(c, hipMemsetAsync(m->lm_state, 0, (size_t)m->max_lm, st));
    OV2_HIP(c, hipMemsetAsync(m->obs_flag, 0, (size_t)m->max_obs, st));
#define CP(p, n) if ((n) > 0) OV2_HIP(c, hipMemcpyAsync(m->p, old.p, (n) * sizeof(*m->p), hipMemcpyDeviceToDevice, st))
    CP(kf_pose, 7 * K); CP(kf_state, K); CP(lm_xyz, 3 * L); CP(lm_state, L);
    CP(obs_kf, N); CP(obs_lm, N); CP(obs_scale, N); CP(obs_uv, 2 * N); CP(obs_ruv, 2 * N); CP(obs_flag, N);
#undef CP
    OV2_HIP(c, hipStreamSynchronize(st));
    free_capacity_arrays(&old);
    return OV2_OK;
}

// squeezes the dead rows out of the observation table (stable); one synchronisation, fresh column arrays of the same capacity
static ov2_status compact_obs(ov2_map *m)
{
    ov2_ctx *c = m->c;
    hipStream_t st = c->stream;
    const int N = m->n_obs;
    if (N <= 0) return OV2_OK;
    obs_cols O = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t cap = (size_t)m->max_obs;
    ov2_status s = OV2_OK;
#define A(p, n) if (s == OV2_OK) s = dmalloc(c, &O.p, (n))
    A(kf, cap); A(lm, cap); A(scale, cap); A(uv, 2 * cap); A(ruv, 2 * cap); A(flag, cap);
#undef A
    void *fresh[] = {O.kf, O.lm, O.scale, O.uv, O.ruv, O.flag};
    if (s != OV2_OK) {
        for (void *p : fresh) if (p) (void)hipFree(p);
        return s;
    }
    const map_view M = view_of(m);
    const dim3 g((N + 255) / 256), b(256);
    OV2_HIP(c, hipMemsetAsync(O.flag, 0, cap, st));
    OV2_LAUNCH(c, OV2_K_MAP, mc_mark_kernel, g, b, 0, st, M, m->obs_cnt);
    if ((s = exclusive_scan<int>(m, m->obs_cnt, N, m->obs_off, m->hdr + MH_NLIVE)) != OV2_OK) {
        for (void *p : fresh) (void)hipFree(p);
        return s;
    }
    OV2_LAUNCH(c, OV2_K_MAP, mc_scatter_kernel, g, b, 0, st, M, (const int *)m->obs_cnt, (const int *)m->obs_off, O);
    OV2_HIP(c, hipMemcpyAsync(m->hdr_host, m->hdr, MH_N * sizeof(int), hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    void *old[] = {m->obs_kf, m->obs_lm, m->obs_scale, m->obs_uv, m->obs_ruv, m->obs_flag};
    for (void *p : old) (void)hipFree(p);
    m->obs_kf = O.kf; m->obs_lm = O.lm; m->obs_scale = O.scale; m->obs_uv = O.uv; m->obs_ruv = O.ruv; m->obs_flag = O.flag;
    m->n_obs = m->hdr_host[MH_NLIVE];
    m->n_compactions++;
    return OV2_OK;
}

extern "C" ov2_status ov2_map_compact(ov2_map *m, int *rows_before, int *rows_after)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    OV2_HIP(m->c, hipSetDevice(m->c->device));
    if (rows_before) *rows_before = m->n_obs;
    const ov2_status s = compact_obs(m);
    if (rows_after) *rows_after = m->n_obs;
    return s;
}

extern "C" ov2_status ov2_map_obs_rows(const ov2_map *m, int *rows, int *capacity, int *compactions)
{
    if (!m) return OV2_ERR_INVALID;
    if (rows) *rows = m->n_obs;
    if (capacity) *capacity = m->max_obs;
    if (compactions) *compactions = m->n_compactions;
    return OV2_OK;
}

extern "C" ov2_status ov2_map_create(ov2_ctx *c, int max_kf, int max_lm, int max_obs, ov2_map **out)
{
    if (!c || !out || max_kf <= 0 || max_lm <= 0 || max_obs <= 0) return OV2_ERR_INVALID;
    *out = nullptr;
    OV2_HIP(c, hipSetDevice(c->device));
    ov2_map *m = new (std::nothrow) ov2_map();
    if (!m) return OV2_ERR_NOMEM;
    memset(m, 0, sizeof(*m));
    m->c = c; m->max_kf = max_kf; m->max_lm = max_lm; m->max_obs = max_obs;
    ov2_status s = alloc_tables(m);
    if (s == OV2_OK && hipHostMalloc((void **)&m->hdr_host, MH_N * sizeof(int), hipHostMallocDefault) != hipSuccess)
        s = ov2_set_err(c, OV2_ERR_NOMEM, "map header hipHostMalloc");
    if (s != OV2_OK) { ov2_map_destroy(m); return s; }
    OV2_HIP(c, hipMemsetAsync(m->kf_state, 0, (size_t)max_kf, c->stream));
    OV2_HIP(c, hipMemsetAsync(m->lm_state, 0, (size_t)max_lm, c->stream));
    OV2_HIP(c, hipMemsetAsync(m->obs_flag, 0, (size_t)max_obs, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    { std::lock_guard<std::mutex> g(c->mu); c->maps.push_back(m); }
    *out = m;
    return OV2_OK;
}

static void free_tables(ov2_map *m)
{
    free_capacity_arrays(m);
    if (m->out_dev) (void)hipFree(m->out_dev);
    if (m->out_host) (void)hipHostFree(m->out_host);
    if (m->hdr_host) (void)hipHostFree(m->hdr_host);
}

void ov2_map_orphan(ov2_map *m)
{
    free_tables(m);
    ov2_ctx *none = nullptr;
    ov2_map blank;
    memset(&blank, 0, sizeof(blank));
    *m = blank;
    m->c = none;
}

extern "C" void ov2_map_destroy(ov2_map *m)
{
    if (!m) return;
    if (m->c) {   // c == nullptr: the ctx went first and already released the tables
        ov2_ctx *c = m->c;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        free_tables(m);
        std::lock_guard<std::mutex> g(c->mu);
        c->maps.erase(std::remove(c->maps.begin(), c->maps.end(), m), c->maps.end());
    }
    delete m;
}

extern "C" ov2_status ov2_map_add_keyframe(ov2_map *m, int kfid, const double *Twc, int n, const int32_t *lmid,
                                           const double *unpx, const double *runpx, const uint8_t *is_stereo,
                                           const int32_t *scale)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (!Twc || n < 0 || (n && (!lmid || !unpx))) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_add_keyframe: null argument");
    if (kfid < 0) return ov2_set_err(c, OV2_ERR_INVALID, "negative kfid %d", kfid);
    int top_lm = -1;
    for (int i = 0; i < n; ++i) {
        if (lmid[i] < 0) return ov2_set_err(c, OV2_ERR_INVALID, "negative lmid %d", lmid[i]);
        top_lm = std::max(top_lm, lmid[i]);
    }
    OV2_HIP(c, hipSetDevice(c->device));
    {
        const ov2_status gs = ensure_capacity(m, kfid + 1, top_lm + 1, m->n_obs + n);
        if (gs != OV2_OK) return gs;
    }
    // the rows are appended as they are: assemble them in the staging block, then plain copies into the tables
    stager S{c};
    ov2_status s = S.begin((size_t)n * (3 * sizeof(int) + 4 * sizeof(double) + 1) + 7 * sizeof(double) + 256);
    if (s != OV2_OK) return s;
    const double *dT = S.put(Twc, 7);
    const int *dl = S.put(lmid, n);
    const double *du = S.put(unpx, 2 * (size_t)n);
    // derived columns are built in place in the pinned block
    S.off = (S.off + 15) & ~(size_t)15;
    int *hk = reinterpret_cast<int *>(S.h + S.off); const int *dk = reinterpret_cast<const int *>(S.d + S.off); S.off += sizeof(int) * n;
    S.off = (S.off + 15) & ~(size_t)15;
    int *hs = reinterpret_cast<int *>(S.h + S.off); const int *ds = reinterpret_cast<const int *>(S.d + S.off); S.off += sizeof(int) * n;
    S.off = (S.off + 15) & ~(size_t)15;
    double *hr = reinterpret_cast<double *>(S.h + S.off); const double *dr = reinterpret_cast<const double *>(S.d + S.off); S.off += sizeof(double) * 2 * n;
    S.off = (S.off + 15) & ~(size_t)15;
    unsigned char *hf = S.h + S.off; const unsigned char *df = S.d + S.off; S.off += n;
    for (int i = 0; i < n; ++i) {
        hk[i] = kfid;
        hs[i] = scale ? scale[i] : 0;
        const bool st = is_stereo && is_stereo[i] && runpx;
        hr[2 * i] = st ? runpx[2 * i] : 0.0; hr[2 * i + 1] = st ? runpx[2 * i + 1] : 0.0;
        hf[i] = OBS_ALIVE | (st ? OBS_STEREO : 0);
    }
    if ((s = S.flush()) != OV2_OK) return s;
    const size_t o = m->n_obs;
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemcpyAsync(m->kf_pose + 7 * (size_t)kfid, dT, 7 * sizeof(double), hipMemcpyDeviceToDevice, st));
    OV2_HIP(c, hipMemsetAsync(m->kf_state + kfid, 1, 1, st));
    if (n) {
        OV2_HIP(c, hipMemcpyAsync(m->obs_lm + o, dl, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(m->obs_uv + 2 * o, du, sizeof(double) * 2 * n, hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(m->obs_kf + o, dk, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(m->obs_scale + o, ds, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(m->obs_ruv + 2 * o, dr, sizeof(double) * 2 * n, hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(m->obs_flag + o, df, n, hipMemcpyDeviceToDevice, st));
    }
    m->n_obs += n;
    OV2_HIP(c, hipStreamSynchronize(st));   // the staging block is free again
    return OV2_OK;
}

extern "C" ov2_status ov2_map_set_landmarks(ov2_map *m, int n, const int32_t *lmid, const double *xyz, const uint8_t *state)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (n < 0 || (n && (!lmid || !state))) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_set_landmarks: null argument");
    if (!n) return OV2_OK;
    int top_lm = -1;
    for (int i = 0; i < n; ++i) {
        if (lmid[i] < 0) return ov2_set_err(c, OV2_ERR_INVALID, "negative lmid %d", lmid[i]);
        top_lm = std::max(top_lm, lmid[i]);
    }
    OV2_HIP(c, hipSetDevice(c->device));
    {
        const ov2_status gs = ensure_capacity(m, 0, top_lm + 1, 0);
        if (gs != OV2_OK) return gs;
    }
    stager S{c};
    ov2_status s = S.begin((size_t)n * (sizeof(int) + 3 * sizeof(double) + 1) + 256);
    if (s != OV2_OK) return s;
    const int *dl = S.put(lmid, n);
    const double *dx = S.put(xyz, 3 * (size_t)n);
    const unsigned char *dst = S.put(state, n);
    if ((s = S.flush()) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, map_set_lm_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dl, dx, dst, m->lm_xyz, m->lm_state);
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_map_set_poses(ov2_map *m, int n, const int32_t *kfid, const double *Twc)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (n < 0 || (n && (!kfid || !Twc))) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_set_poses: null argument");
    if (!n) return OV2_OK;
    for (int i = 0; i < n; ++i)
        if (kfid[i] < 0 || kfid[i] >= m->max_kf) return ov2_set_err(c, OV2_ERR_INVALID, "kfid %d outside the map capacity", kfid[i]);
    OV2_HIP(c, hipSetDevice(c->device));
    stager S{c};
    ov2_status s = S.begin((size_t)n * (sizeof(int) + 7 * sizeof(double)) + 256);
    if (s != OV2_OK) return s;
    const int *dk = S.put(kfid, n);
    const double *dT = S.put(Twc, 7 * (size_t)n);
    if ((s = S.flush()) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, map_set_pose_kernel, dim3((7 * n + 255) / 256), dim3(256), 0, c->stream, n, dk, dT, m->kf_pose);
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

static ov2_status edit_obs(ov2_map *m, int n, const int32_t *kfid, const int32_t *lmid, int mode, const uint8_t *st, const double *ruv)
{
    ov2_ctx *c = m->c;
    if (!n || !m->n_obs) return OV2_OK;
    OV2_HIP(c, hipSetDevice(c->device));
    stager S{c};
    ov2_status s = S.begin((size_t)n * (2 * sizeof(int) + 2 * sizeof(double) + 1) + 256);
    if (s != OV2_OK) return s;
    const int *dk = S.put(kfid, n);
    const int *dl = S.put(lmid, n);
    const unsigned char *ds = S.put(st, n);
    const double *dr = S.put(ruv, 2 * (size_t)n);
    if ((s = S.flush()) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, map_edit_obs_kernel, dim3((m->n_obs + 255) / 256), dim3(256), 0, c->stream, m->n_obs, (const int *)m->obs_kf,
               (const int *)m->obs_lm, m->obs_flag, m->obs_ruv, n, dk, dl, mode, ds, dr);
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_map_remove_obs(ov2_map *m, int n, const int32_t *kfid, const int32_t *lmid)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    if (n < 0 || (n && (!kfid || !lmid))) return ov2_set_err(m->c, OV2_ERR_INVALID, "ov2_map_remove_obs: null argument");
    return edit_obs(m, n, kfid, lmid, 0, nullptr, nullptr);
}

extern "C" ov2_status ov2_map_set_obs_stereo(ov2_map *m, int n, const int32_t *kfid, const int32_t *lmid, const uint8_t *is_stereo,
                                             const double *runpx)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    if (n < 0 || (n && (!kfid || !lmid || !is_stereo || !runpx)))
        return ov2_set_err(m->c, OV2_ERR_INVALID, "ov2_map_set_obs_stereo: null argument");
    return edit_obs(m, n, kfid, lmid, 1, is_stereo, runpx);
}

extern "C" ov2_status ov2_map_remove_landmarks(ov2_map *m, int n, const int32_t *lmid)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (n < 0 || (n && !lmid)) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_remove_landmarks: null argument");
    if (!n) return OV2_OK;
    // the landmark row dies; its observations stop counting through obs_live (their rows stay as tombstones)
    std::vector<uint8_t> zero((size_t)n, 0);
    return ov2_map_set_landmarks(m, n, lmid, nullptr, zero.data());
}

extern "C" ov2_status ov2_map_remove_keyframe(ov2_map *m, int kfid)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (kfid < 0 || kfid >= m->max_kf) return ov2_set_err(c, OV2_ERR_INVALID, "kfid %d outside the map capacity", kfid);
    OV2_HIP(c, hipSetDevice(c->device));
    OV2_HIP(c, hipMemsetAsync(m->kf_state + kfid, 0, 1, c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_map_local_ba_setup(ov2_map *m, int newkf, int nmin_covscore, int nmin_cst_kfs, int inv_depth,
                                             const double *calib_l, ov2_local_ba_setup *out)
{
    (void)calib_l;
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (!out) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_local_ba_setup: null output");
    memset(out, 0, sizeof(*out));
    if (newkf < 0 || newkf >= m->max_kf) return ov2_set_err(c, OV2_ERR_INVALID, "kfid %d outside the map capacity", newkf);
    OV2_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int N = m->n_obs, K = m->max_kf, L = m->max_lm, inv = inv_depth ? 1 : 0;
    const map_view M = view_of(m);
    const dim3 gN((std::max(N, 1) + 255) / 256), gL((L + 255) / 256), gK((K + 255) / 256), b(256);
    OV2_HIP(c, hipMemsetAsync(m->zero_blk, 0, m->zero_bytes, st));
    OV2_LAUNCH(c, OV2_K_MAP, ms_count_kernel, gN, b, 0, st, M, newkf, m->lm_nobs, m->lm_new, m->hdr);
    OV2_LAUNCH(c, OV2_K_MAP, ms_cov_kernel, gN, b, 0, st, M, newkf, (const unsigned char *)m->lm_new, m->cov);
    OV2_LAUNCH(c, OV2_K_MAP, ms_select_kernel, dim3(1), dim3(64), 0, st, M, newkf, nmin_covscore, (const int *)m->cov, m->kf_role, m->hdr);
    OV2_LAUNCH(c, OV2_K_MAP, ms_local_lm_kernel, gN, b, 0, st, M, (const int *)m->kf_role, (const int *)m->lm_nobs, m->lm_sel,
               (const int *)m->hdr);
    OV2_LAUNCH(c, OV2_K_MAP, ms_observers_kernel, gN, b, 0, st, M, (const int *)m->lm_sel, m->kf_role, m->lm_anchor, (const int *)m->hdr);
    OV2_LAUNCH(c, OV2_K_MAP, ms_poses_kernel, dim3(1), dim3(1024), 0, st, K, nmin_cst_kfs, m->kf_role, m->kf_idx, m->hdr);
    OV2_LAUNCH(c, OV2_K_MAP, ms_lm_flags_kernel, gL, b, 0, st, L, inv, (const int *)m->lm_sel, (const int *)m->lm_anchor, m->lm_flag, m->lm_pack);
    ov2_status s;
    // one 64-bit scan numbers the landmarks (low word) and the bad list (high word); its total lands on NLM | NBAD
    if ((s = exclusive_scan<unsigned long long>(m, m->lm_pack, L, m->lm_pidx, reinterpret_cast<unsigned long long *>(m->hdr + MH_NLM))) != OV2_OK)
        return s;
    OV2_LAUNCH(c, OV2_K_MAP, ms_res_count_kernel, gN, b, 0, st, M, inv, (const int *)m->lm_flag, (const int *)m->lm_anchor, m->obs_cnt,
               (const int *)m->hdr);
    if ((s = exclusive_scan<int>(m, m->obs_cnt, N, m->obs_off, m->hdr + MH_NRES)) != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(m->hdr_host, m->hdr, MH_N * sizeof(int), hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    const int *H = m->hdr_host;
    out->aborted = H[MH_ABORT];
    // mostly tombstones left: squeeze the table once this set-up has read it (amortised like a growth)
    const bool squeeze = N >= MAP_COMPACT_MIN_ROWS && 2 * (long long)H[MH_NLIVE] < N;
    if (out->aborted) return squeeze ? compact_obs(m) : OV2_OK;
    const size_t P = H[MH_NPOSE], NL = H[MH_NLM], R = H[MH_NRES], NB = H[MH_NBAD];
    const int e = inv ? 1 : 3;
    // carve the flat problem (same layout on the device and in the pinned mirror)
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; };
    const size_t o_pk = carve(P * 4), o_pc = carve(P), o_po = carve(P * 56), o_ll = carve(NL * 4), o_lm = carve(NL * 8 * e),
                 o_la = carve(NL * 4), o_lu = carve(NL * 16), o_rt = carve(R), o_rp = carve(R * 4), o_rl = carve(R * 4),
                 o_ru = carve(R * 16), o_rs = carve(R * 8), o_bl = carve(NB * 4);
    if (off > m->out_cap) {
        if (m->out_dev) OV2_HIP(c, hipFree(m->out_dev));
        if (m->out_host) OV2_HIP(c, hipHostFree(m->out_host));
        m->out_dev = m->out_host = nullptr; m->out_cap = 0;
        const size_t want = off + off / 2 + 4096;
        if (hipMalloc((void **)&m->out_dev, want) != hipSuccess || hipHostMalloc((void **)&m->out_host, want, hipHostMallocDefault) != hipSuccess)
            return ov2_set_err(c, OV2_ERR_NOMEM, "flat problem buffers of %zu bytes", want);
        m->out_cap = want;
    }
    auto at = [&](unsigned char *base, size_t o) { return base + o; };
    flat_out O;
    unsigned char *D = m->out_dev;
    O.pose_kfid = (int *)at(D, o_pk); O.pose_const = at(D, o_pc); O.pose = (double *)at(D, o_po);
    O.lm_lmid = (int *)at(D, o_ll); O.lm = (double *)at(D, o_lm); O.lm_anchor_pose = (int *)at(D, o_la); O.lm_anchor_uv = (double *)at(D, o_lu);
    O.res_type = at(D, o_rt); O.res_pose = (int *)at(D, o_rp); O.res_lm = (int *)at(D, o_rl); O.res_uv = (double *)at(D, o_ru);
    O.res_sigma = (double *)at(D, o_rs); O.bad_lmid = (int *)at(D, o_bl);
    OV2_LAUNCH(c, OV2_K_MAP, me_all_kernel, dim3(gK.x + gL.x + gN.x), b, 0, st, M, (int)gK.x, (int)gL.x, inv, (const int *)m->kf_role,
               (const int *)m->kf_idx, (const int *)m->lm_sel, (const int *)m->lm_flag, (const unsigned long long *)m->lm_pidx,
               (const int *)m->lm_anchor, (const int *)m->obs_off, (const int *)m->hdr, O);
    OV2_HIP(c, hipMemcpyAsync(m->out_host, m->out_dev, off, hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    unsigned char *Hh = m->out_host;
    out->n_pose = (int)P; out->n_lm = (int)NL; out->n_res = (int)R; out->n_bad = (int)NB;
    out->pose_kfid = (const int32_t *)at(Hh, o_pk); out->pose_const = at(Hh, o_pc); out->pose = (double *)at(Hh, o_po);
    out->lm_lmid = (const int32_t *)at(Hh, o_ll); out->lm = (double *)at(Hh, o_lm);
    out->lm_anchor_pose = (const int32_t *)at(Hh, o_la); out->lm_anchor_uv = (const double *)at(Hh, o_lu);
    out->res_type = at(Hh, o_rt); out->res_pose = (const int32_t *)at(Hh, o_rp); out->res_lm = (const int32_t *)at(Hh, o_rl);
    out->res_uv = (const double *)at(Hh, o_ru); out->res_sigma = (const double *)at(Hh, o_rs);
    out->bad_lmid = (const int32_t *)at(Hh, o_bl);
    return squeeze ? compact_obs(m) : OV2_OK;
}

// The same flat problem, where the set-up left it on the device (ov2_ba_solve_batch_dev takes these pointers as they are)
extern "C" ov2_status ov2_map_setup_device_view(const ov2_map *m, const ov2_local_ba_setup *host, ov2_local_ba_setup *dev)
{
    if (!m || !host || !dev) return OV2_ERR_INVALID;
    *dev = *host;
    if (host->aborted) return OV2_OK;
    const unsigned char *H = m->out_host, *D = m->out_dev;
    auto tr = [&](const void *p) -> const unsigned char * {
        const unsigned char *q = (const unsigned char *)p;
        return (q && H && q >= H && q < H + m->out_cap) ? D + (q - H) : nullptr;
    };
    if (host->n_pose && !tr(host->pose)) return OV2_ERR_INVALID;   // not the arrays of this map's last set-up
    dev->pose_kfid = (const int32_t *)tr(host->pose_kfid); dev->pose_const = tr(host->pose_const); dev->pose = (double *)tr(host->pose);
    dev->lm_lmid = (const int32_t *)tr(host->lm_lmid); dev->lm = (double *)tr(host->lm);
    dev->lm_anchor_pose = (const int32_t *)tr(host->lm_anchor_pose); dev->lm_anchor_uv = (const double *)tr(host->lm_anchor_uv);
    dev->res_type = tr(host->res_type); dev->res_pose = (const int32_t *)tr(host->res_pose); dev->res_lm = (const int32_t *)tr(host->res_lm);
    dev->res_uv = (const double *)tr(host->res_uv); dev->res_sigma = (const double *)tr(host->res_sigma);
    dev->bad_lmid = (const int32_t *)tr(host->bad_lmid);
    return OV2_OK;
}
