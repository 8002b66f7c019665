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
    size_t out_cap, out_host_cap;
    int *hdr_host;                                                  // pinned
    // the last set-up (the update stage reads its scratch arrays and its flat problem where they were left)
    int last_hdr[16];                                               // host copy of its header
    int last_newkf, last_inv, last_valid;
    int live_rows, live_known;                                      // live rows of the table as the last set-up counted them
    double last_K[4]; int have_K;                                   // left intrinsics (anchored inverse depth -> world point)
    // ov2_map_save_state / ov2_map_restore_state_batch
    double *snap_kf_pose, *snap_lm_xyz; unsigned char *snap_kf_state, *snap_lm_state, *snap_obs_flag;
    int snap_kf, snap_lm, snap_obs;
};

namespace {

enum { MH_NBKPS = 0, MH_NB3D, MH_ABORT, MH_NMAXKF, MH_NPOSE, MH_NRES, MH_NLM, MH_NBAD, MH_NLIVE,   // NLM|NBAD: one 64-bit scan total
       MH_NEED16,                       // bytes / 16 the flat problem needs (written with the gathered header)
       MH_OVER,                         // ... and 1 when that exceeds the map's output block: nothing was emitted, the host grows it and re-runs
       MH_NRM_LM, MH_NRM_OBS, MH_NST_OFF,   // update stage: landmarks removed, observations removed, stereo observations demoted
       MH_N = 16 };
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

// ---- batched set-up / update ---------------------------------------------------------------------------
// Every kernel below serves B maps in one launch: the map is blockIdx.y, its tables and scratch arrays come from a
// device table of map_job records (one H2D copy per call).  Sizes that a later kernel needs (poses, landmarks, residual
// blocks of the flat problem) are read from the map's device-side header, never from the host: the whole chain is
// enqueued without a synchronisation.
struct map_job {
    map_view M;
    int newkf, cur_kfid;
    int *hdr, *cov, *kf_role, *kf_idx, *lm_nobs, *lm_sel, *lm_anchor, *lm_flag;
    unsigned long long *lm_pack, *lm_pidx;
    unsigned char *lm_new;
    int *obs_cnt, *obs_off;
    void *blk;                                   // block sums of the scans
    unsigned char *zero_blk; unsigned zero_vec16; // the zeroed block, in 16-byte units
    unsigned char *out; unsigned long long out_cap;
    int *hdr_out;                                // this map's slot of the gathered headers (one D2H for all maps)
    // writable twins of the tables (update stage, restore)
    double *kf_pose_w, *lm_xyz_w; unsigned char *kf_state_w, *lm_state_w, *obs_flag_w;
    const unsigned char *outlier;                // update stage: per residual block flags of the solve (may be null)
    double K[4];                                 // left intrinsics (update stage, inverse depth)
    const double *snap_kf_pose, *snap_lm_xyz; const unsigned char *snap_kf_state, *snap_lm_state, *snap_obs_flag;
    int snap_kf, snap_lm, snap_obs;
};

// the flat problem of one map inside its output block: the same carve on the device (emitters, update stage) and on the
// host (pointer translation), from the four counts of the header
enum { FO_POSE_KFID = 0, FO_POSE_CONST, FO_POSE, FO_LM_LMID, FO_LM, FO_LM_ANCH, FO_LM_AUV, FO_RES_TYPE, FO_RES_POSE, FO_RES_LM,
       FO_RES_UV, FO_RES_SIGMA, FO_BAD, FO_HOST_END,   // [0, FO_HOST_END): what the host form copies to pinned memory
       FO_RES_OUT = FO_HOST_END, FO_RM_LM, FO_RM_OBS, FO_ST_OFF, FO_N };

__host__ __device__ inline size_t flat_layout(size_t P, size_t NL, size_t R, size_t NB, int e, size_t *off)
{
    size_t o = 0;
#define CARVE(k, bytes) do { off[k] = o; o = (o + (size_t)(bytes) + 15) & ~(size_t)15; } while (0)
    CARVE(FO_POSE_KFID, P * 4); CARVE(FO_POSE_CONST, P); CARVE(FO_POSE, P * 56); CARVE(FO_LM_LMID, NL * 4); CARVE(FO_LM, NL * 8 * e);
    CARVE(FO_LM_ANCH, NL * 4); CARVE(FO_LM_AUV, NL * 16); CARVE(FO_RES_TYPE, R); CARVE(FO_RES_POSE, R * 4); CARVE(FO_RES_LM, R * 4);
    CARVE(FO_RES_UV, R * 16); CARVE(FO_RES_SIGMA, R * 8); CARVE(FO_BAD, NB * 4);
    // device only: the solver's per-block outlier flags, and what the update stage reports back
    CARVE(FO_RES_OUT, R); CARVE(FO_RM_LM, (NL + NB) * 4); CARVE(FO_RM_OBS, R * 8); CARVE(FO_ST_OFF, R * 8);
#undef CARVE
    return o;
}

struct flat_out {
    int *pose_kfid; unsigned char *pose_const; double *pose;
    int *lm_lmid; double *lm; int *lm_anchor_pose; double *lm_anchor_uv;
    unsigned char *res_type; int *res_pose, *res_lm; double *res_uv, *res_sigma;
    int *bad_lmid;
    unsigned char *res_out; int *rm_lm; int2 *rm_obs, *st_off;
};

__host__ __device__ inline flat_out flat_ptrs(unsigned char *D, const size_t *off)
{
    flat_out O;
    O.pose_kfid = (int *)(D + off[FO_POSE_KFID]); O.pose_const = D + off[FO_POSE_CONST]; O.pose = (double *)(D + off[FO_POSE]);
    O.lm_lmid = (int *)(D + off[FO_LM_LMID]); O.lm = (double *)(D + off[FO_LM]); O.lm_anchor_pose = (int *)(D + off[FO_LM_ANCH]);
    O.lm_anchor_uv = (double *)(D + off[FO_LM_AUV]); O.res_type = D + off[FO_RES_TYPE]; O.res_pose = (int *)(D + off[FO_RES_POSE]);
    O.res_lm = (int *)(D + off[FO_RES_LM]); O.res_uv = (double *)(D + off[FO_RES_UV]); O.res_sigma = (double *)(D + off[FO_RES_SIGMA]);
    O.bad_lmid = (int *)(D + off[FO_BAD]); O.res_out = D + off[FO_RES_OUT]; O.rm_lm = (int *)(D + off[FO_RM_LM]);
    O.rm_obs = (int2 *)(D + off[FO_RM_OBS]); O.st_off = (int2 *)(D + off[FO_ST_OFF]);
    return O;
}

__device__ __forceinline__ flat_out flat_of(const map_job &j, int inv, bool *fits = nullptr)
{
    const int *H = j.hdr;
    size_t off[FO_N];
    const size_t need = flat_layout((size_t)H[MH_NPOSE], (size_t)H[MH_NLM], (size_t)H[MH_NRES], (size_t)H[MH_NBAD], inv ? 1 : 3, off);
    if (fits) *fits = need <= j.out_cap;
    return flat_ptrs(j.out, off);
}

__global__ __launch_bounds__(256) void mb_zero_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    uint4 *z = reinterpret_cast<uint4 *>(j.zero_blk);
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < j.zero_vec16; i += gridDim.x * 256) z[i] = zero;
}

// ---- set-up scans -----------------------------------------------------------------------------------
// observers per landmark, landmarks of the new keyframe, its keypoint counts (Frame::nbkps_, nb3dkps_)
__global__ __launch_bounds__(256) void ms_count_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf = 0, lm = 0;
    const bool live = i < M.n_obs && obs_live(M, i, kf, lm);
    // live rows of the table (one atomic per wave): the host compacts the table when most rows are dead
    const unsigned long long bal = __ballot(live);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&j.hdr[MH_NLIVE], __popcll(bal));
    if (!live) return;
    atomicAdd(&j.lm_nobs[lm], 1);
    if (kf == j.newkf) {
        j.lm_new[lm] = 1;
        atomicAdd(&j.hdr[MH_NBKPS], 1);
        if (M.lm_state[lm] & OV2_LM_KP3D) atomicAdd(&j.hdr[MH_NB3D], 1);
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
__global__ __launch_bounds__(256) void ms_cov_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    if (kf != j.newkf && j.lm_new[lm]) atomicAdd(&j.cov[kf], 1);
}

// src/optimizer.cpp:61-63,128-190: one wave walks the covisible keyframes newest -> oldest, 64 per step.
// kf_role: 0 = not in the problem, 1 = optimised, 2 = constant
__global__ __launch_bounds__(64) void ms_select_kernel(const map_job *__restrict__ J, int nmin_cov)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    int *hdr = j.hdr;
    const int newkf = j.newkf;
    const int lane = threadIdx.x;
    const int nb3d = hdr[MH_NB3D], nbkps = hdr[MH_NBKPS];
    if (nb3d < nmin_cov) { if (lane == 0) hdr[MH_ABORT] = 1; return; }
    bool all_cst = false;
    int nmax = -1;
    for (int hi = M.max_kf - 1; hi >= 0; hi -= 64) {
        const int kf = hi - lane;
        bool in = false, good = false;
        if (kf >= 0 && M.kf_state[kf]) {
            int score = (kf == newkf) ? nb3d : j.cov[kf];
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
        if (in) j.kf_role[kf] = cst ? 2 : 1;
    }
    if (lane == 0) hdr[MH_NMAXKF] = nmax;
}

// landmarks of the optimised keyframes' 3D keypoints (:176-180) and MapPoint::isBad (src/map_point.cpp:215-234)
__global__ __launch_bounds__(256) void ms_local_lm_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (j.hdr[MH_ABORT] || i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    const int st = M.lm_state[lm];
    if (j.kf_role[kf] != 1 || !(st & OV2_LM_KP3D)) return;
    const int nobs = j.lm_nobs[lm];
    const bool isobs = st & OV2_LM_OBS;
    const bool bad = (nobs < 2 && !isobs && (st & OV2_LM_3D)) || (nobs == 0 && !isobs);
    j.lm_sel[lm] = bad ? 2 : 1;
}

// observers of the local landmarks: outside keyframes become constant poses (:229-246), the first observer anchors
// the landmark (:251-287).  The isBad() landmarks keep their oldest observer too (MapPoint::kfid_, which the culling
// rule of the update stage reads, :810 / :871).
__global__ __launch_bounds__(256) void ms_observers_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (j.hdr[MH_ABORT] || i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    const int sel = j.lm_sel[lm];
    if (sel == 2) { atomicMax(&j.lm_anchor[lm], ANCH_TOP - kf); return; }
    if (sel != 1 || kf > j.hdr[MH_NMAXKF]) return;
    if (j.kf_role[kf] == 0) j.kf_role[kf] = 2;     // every writer stores the same value
    atomicMax(&j.lm_anchor[lm], ANCH_TOP - kf);     // the smallest kfid wins
}

// gauge (:394-407) + dense pose numbering in ascending kfid.  One workgroup per map.
__global__ __launch_bounds__(1024) void ms_poses_kernel(const map_job *__restrict__ J, int nmin_cst)
{
    const map_job &j = J[blockIdx.y];
    const int max_kf = j.M.max_kf;
    int *kf_role = j.kf_role, *kf_idx = j.kf_idx, *hdr = j.hdr;
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
__global__ __launch_bounds__(256) void ms_lm_flags_kernel(const map_job *__restrict__ J, int inv)
{
    const map_job &j = J[blockIdx.y];
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= j.M.max_lm) return;
    const int s = j.lm_sel[l];
    const int f = (s == 1 && (!inv || j.lm_anchor[l] != 0)) ? 1 : 0;
    j.lm_flag[l] = f;
    j.lm_pack[l] = (unsigned long long)f | ((unsigned long long)(s == 2) << 32);
}

// residual blocks per observation (:251-391): anchor observation 1 if stereo (right-anchor block) else 0; any other
// observation 2 if stereo (left + right) else 1
__global__ __launch_bounds__(256) void ms_res_count_kernel(const map_job *__restrict__ J, int inv)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.n_obs) return;
    int kf, lm, c = 0;
    if (!j.hdr[MH_ABORT] && obs_live(M, i, kf, lm) && j.lm_flag[lm] && kf <= j.hdr[MH_NMAXKF]) {
        const bool stereo = M.obs_flag[i] & OBS_STEREO;
        if (inv && j.lm_anchor[lm] == ANCH_TOP - kf) c = stereo ? 1 : 0;
        else c = stereo ? 2 : 1;
    }
    j.obs_cnt[i] = c;
}

// ---- exclusive scan of an array (int, or two counters packed in 64 bits) in three launches:
// block sums -> their scan -> block offsets.  `which` selects the array of the map: SC_LM the landmark flags (64-bit,
// total -> NLM | NBAD), SC_RES the residual counts (total -> NRES), SC_LIVE the keep flags of a compaction (total -> NLIVE)
enum { SC_LM = 0, SC_RES, SC_LIVE };

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
__device__ __forceinline__ void scan_args(const map_job &j, int which, const T *&in, T *&out, T *&blk, T *&total, int &n)
{
    blk = reinterpret_cast<T *>(j.blk);
    if (which == SC_LM) {
        in = reinterpret_cast<const T *>(j.lm_pack); out = reinterpret_cast<T *>(j.lm_pidx); n = j.M.max_lm;
        total = reinterpret_cast<T *>(j.hdr + MH_NLM);
    } else {
        in = reinterpret_cast<const T *>(j.obs_cnt); out = reinterpret_cast<T *>(j.obs_off); n = j.M.n_obs;
        total = reinterpret_cast<T *>(j.hdr + (which == SC_RES ? MH_NRES : MH_NLIVE));
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_block_kernel(const map_job *__restrict__ J, int which)
{
    const T *in; T *out, *blk, *total; int n;
    scan_args<T>(J[blockIdx.y], which, in, out, blk, total, n);
    if ((int)blockIdx.x * 1024 >= n) return;
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
__global__ __launch_bounds__(1024) void scan_top_kernel(const map_job *__restrict__ J, int which)
{
    const T *in; T *out, *blk, *total; int n;
    scan_args<T>(J[blockIdx.y], which, in, out, blk, total, n);
    const int nb = (n + 1023) / 1024;
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
__global__ __launch_bounds__(1024) void scan_add_kernel(const map_job *__restrict__ J, int which)
{
    const T *in; T *out, *blk, *total; int n;
    scan_args<T>(J[blockIdx.y], which, in, out, blk, total, n);
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += blk[blockIdx.x];
}

// ---- emission into the flat problem -------------------------------------------------------------------
__device__ __forceinline__ void me_poses(const map_job &j, int k, const flat_out &O)
{
    const map_view &M = j.M;
    if (k >= M.max_kf || j.kf_role[k] == 0) return;
    const int p = j.kf_idx[k];
    O.pose_kfid[p] = k;
    O.pose_const[p] = j.kf_role[k] == 2;
    for (int t = 0; t < 7; ++t) O.pose[7 * p + t] = M.kf_pose[7 * k + t];
}

__device__ __forceinline__ void me_lms(const map_job &j, int l, int inv, const flat_out &O)
{
    const map_view &M = j.M;
    if (l >= M.max_lm) return;
    const unsigned long long pi = j.lm_pidx[l];
    if (j.lm_sel[l] == 2) {
        O.bad_lmid[(int)(pi >> 32)] = l;
        j.lm_state_w[l] = M.lm_state[l] & ~OV2_LM_3D;   // MapPoint::isBad() clears is3d_ when it answers true (src/map_point.cpp:219,227)
    }
    if (!j.lm_flag[l]) return;
    const int q = (int)(pi & 0xffffffffull);
    O.lm_lmid[q] = l;
    if (!inv) {
        for (int t = 0; t < 3; ++t) O.lm[3 * q + t] = M.lm_xyz[3 * l + t];
        O.lm_anchor_pose[q] = -1;
        O.lm_anchor_uv[2 * q] = O.lm_anchor_uv[2 * q + 1] = 0.0;
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

__device__ __forceinline__ void me_res(const map_job &j, int i, int inv, const flat_out &O)
{
    const map_view &M = j.M;
    if (i >= M.n_obs) return;
    int kf, lm;
    if (!obs_live(M, i, kf, lm) || !j.lm_flag[lm] || kf > j.hdr[MH_NMAXKF]) return;
    const int q = (int)(j.lm_pidx[lm] & 0xffffffffull), pj = j.kf_idx[kf];
    const bool stereo = M.obs_flag[i] & OBS_STEREO;
    const double sigma = (double)(1 << M.obs_scale[i]);   // std::pow(2., kp.scale_)
    int o = j.obs_off[i];
    auto put = [&](int type, const double *uv) {
        O.res_type[o] = (unsigned char)type; O.res_pose[o] = pj; O.res_lm[o] = q;
        O.res_uv[2 * o] = uv[0]; O.res_uv[2 * o + 1] = uv[1]; O.res_sigma[o] = sigma;
        O.res_out[o] = 0;
        ++o;
    };
    if (inv && j.lm_anchor[lm] == ANCH_TOP - kf) {
        O.lm[q] = 1.0 / depth_in_kf(M.kf_pose + 7 * kf, M.lm_xyz + 3 * lm);
        O.lm_anchor_pose[q] = pj;
        O.lm_anchor_uv[2 * q] = M.obs_uv[2 * i]; O.lm_anchor_uv[2 * q + 1] = M.obs_uv[2 * i + 1];
        if (stereo) put(OV2_BA_RANCH_INV, M.obs_ruv + 2 * i);
        return;
    }
    put(inv ? OV2_BA_L_INV : OV2_BA_L_XYZ, M.obs_uv + 2 * i);
    if (stereo) put(inv ? OV2_BA_R_INV : OV2_BA_R_XYZ, M.obs_ruv + 2 * i);
}

// one launch: blocks [0, gK) write the poses, [gK, gK + gL) the landmarks and the bad list, the rest the residual
// blocks; block 0 of every map also hands its header to the gathered copy the host reads.  A map whose flat problem does
// not fit its output block emits nothing (the host grows the block and runs the set-up again).
__global__ __launch_bounds__(256) void me_all_kernel(const map_job *__restrict__ J, int gK, int gL, int inv)
{
    const map_job &j = J[blockIdx.y];
    const int blk = blockIdx.x, t = threadIdx.x;
    const int *H = j.hdr;
    size_t off[FO_N];
    const bool aborted = H[MH_ABORT] != 0;
    const size_t need = aborted ? 0 : flat_layout((size_t)H[MH_NPOSE], (size_t)H[MH_NLM], (size_t)H[MH_NRES], (size_t)H[MH_NBAD], inv ? 1 : 3, off);
    const bool fits = need <= j.out_cap;
    if (blk == 0 && t < MH_N) {
        int v = H[t];
        if (t == MH_NEED16) v = (int)(need >> 4);
        if (t == MH_OVER) v = fits ? 0 : 1;
        j.hdr_out[t] = v;
    }
    if (aborted || !fits) return;
    const flat_out O = flat_ptrs(j.out, off);
    if (blk < gK) me_poses(j, blk * 256 + t, O);
    else if (blk < gK + gL) me_lms(j, (blk - gK) * 256 + t, inv, O);
    else me_res(j, (blk - gK - gL) * 256 + t, inv, O);
}

// ---- update stage of Optimizer::localBA (src/optimizer.cpp:741-882) on the tables ---------------------------------
// Inputs: the flat problem where the set-up left it (poses / landmarks now hold the solved states), the per residual block
// outlier flags of the solve, and the set-up's scratch arrays (residual blocks per observation row, observers per
// landmark, oldest observer per landmark), which stay valid until the map's next set-up.
//
// pass 1, one thread per observation row: a flagged LEFT block removes the observation (MapManager::removeMapPointObs,
// :753-764; an observation of the current frame also clears MapPoint::isobs_, removeObsFromCurFrameById), a flagged RIGHT
// block demotes it to mono (Frame::removeStereoKeypointById, :743-751); both put the landmark into set_badlmids.
__global__ __launch_bounds__(256) void mu_obs_kernel(const map_job *__restrict__ J, int inv)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    if (!j.outlier || j.hdr[MH_ABORT] || (int)blockIdx.x * 256 >= M.n_obs) return;   // uniform per workgroup
    __shared__ int s_cnt[2], s_base[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const flat_out O = flat_of(j, inv);
    bool left = false, right = false;
    const int c = i < M.n_obs ? j.obs_cnt[i] : 0;
    if (c) {
        const int o = j.obs_off[i];
        for (int k = 0; k < c; ++k)
            if (j.outlier[o + k]) {
                const int t = O.res_type[o + k];
                if (t == OV2_BA_L_XYZ || t == OV2_BA_L_INV) left = true; else right = true;
            }
    }
    // the two report lists: ranks inside the workgroup from LDS counters, ONE global atomic per workgroup and list
    // (one per flagged row on the map's header line serialised the whole launch: 1.66 ms for 64 maps of 65 k rows)
    const int rl = left ? atomicAdd(&s_cnt[0], 1) : 0, rr = right ? atomicAdd(&s_cnt[1], 1) : 0;
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&j.hdr[threadIdx.x == 0 ? MH_NRM_OBS : MH_NST_OFF], s_cnt[threadIdx.x]);
    __syncthreads();
    if (!left && !right) return;
    const int kf = M.obs_kf[i], lm = M.obs_lm[i];
    atomicOr(&j.lm_sel[lm], 4);
    if (right) {
        j.obs_flag_w[i] = M.obs_flag[i] & ~OBS_STEREO;
        O.st_off[s_base[1] + rr] = make_int2(kf, lm);
    }
    if (left) {
        j.obs_flag_w[i] = 0;
        atomicSub(&j.lm_nobs[lm], 1);
        O.rm_obs[s_base[0] + rl] = make_int2(kf, lm);
        if (kf == j.cur_kfid) j.lm_state_w[lm] = M.lm_state[lm] & ~OV2_LM_OBS;   // one observation per (keyframe, landmark): one writer
        // the oldest observer is gone (XYZ only: an anchor observation carries no left block): MapPoint::removeKfObs moves
        // kfid_ to the next one (src/map_point.cpp:124-126); pass 1b recounts it
        if (j.lm_anchor[lm] == ANCH_TOP - kf) { j.lm_anchor[lm] = 0; atomicOr(&j.lm_sel[lm], 8); }
    }
}

__global__ __launch_bounds__(256) void mu_reanchor_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int kf, lm;
    if (i >= M.n_obs || !obs_live(M, i, kf, lm)) return;
    if (j.lm_sel[lm] & 8) atomicMax(&j.lm_anchor[lm], ANCH_TOP - kf);
}

// rotation of the unit quaternion (x, y, z, w) applied to v, plus t: Twc * v
__device__ __forceinline__ void se3_apply(const double *T, const double *v, double *out)
{
    const double x = T[3], y = T[4], z = T[5], w = T[6];
    const double r00 = 1.0 - 2.0 * (y * y + z * z), r01 = 2.0 * (x * y - z * w), r02 = 2.0 * (x * z + y * w);
    const double r10 = 2.0 * (x * y + z * w), r11 = 1.0 - 2.0 * (x * x + z * z), r12 = 2.0 * (y * z - x * w);
    const double r20 = 2.0 * (x * z - y * w), r21 = 2.0 * (y * z + x * w), r22 = 1.0 - 2.0 * (x * x + y * y);
    out[0] = r00 * v[0] + r01 * v[1] + r02 * v[2] + T[0];
    out[1] = r10 * v[0] + r11 * v[1] + r12 * v[2] + T[1];
    out[2] = r20 * v[0] + r21 * v[1] + r22 * v[2] + T[2];
}

// pass 2: blocks [0, gP) write the solved poses of the non-constant keyframes (:767-786); the others take one landmark
// each: the local landmarks (:789-853: isBad / culling / positive depth / new world point) and then, for the members of
// set_badlmids -- the isBad() landmarks of the set-up and every landmark that lost an observation -- the second
// culling pass (:856-882).
__global__ __launch_bounds__(256) void mu_apply_kernel(const map_job *__restrict__ J, int gP, int inv)
{
    const map_job &j = J[blockIdx.y];
    const map_view &M = j.M;
    const int *H = j.hdr;
    if (H[MH_ABORT]) return;
    const flat_out O = flat_of(j, inv);
    const int blk = blockIdx.x, t = threadIdx.x;
    if (blk < gP) {
        const int p = blk * 256 + t;
        if (p >= H[MH_NPOSE] || O.pose_const[p]) return;
        const int kf = O.pose_kfid[p];
        if (!M.kf_state[kf]) return;
        for (int k = 0; k < 7; ++k) j.kf_pose_w[7 * kf + k] = O.pose[7 * p + k];
        return;
    }
    const int idx = (blk - gP) * 256 + t, NL = H[MH_NLM], NB = H[MH_NBAD];
    if (idx >= NL + NB) return;
    const bool local = idx < NL;
    const int l = local ? O.lm_lmid[idx] : O.bad_lmid[idx - NL];
    int st = M.lm_state[l];
    if (!(st & OV2_LM_ALIVE)) return;   // MapManager::getMapPoint returned nullptr
    const int nobs = j.lm_nobs[l];
    const bool isobs = st & OV2_LM_OBS;
    const int anch = j.lm_anchor[l];
    const int lm_kfid = anch ? ANCH_TOP - anch : -1;   // MapPoint::kfid_ = its oldest observer
    auto remove = [&]() { j.lm_state_w[l] = 0; O.rm_lm[atomicAdd(&j.hdr[MH_NRM_LM], 1)] = l; };
    auto is_bad = [&]() {   // MapPoint::isBad (src/map_point.cpp:215-234)
        if ((nobs < 2 && !isobs && (st & OV2_LM_3D)) || (nobs == 0 && !isobs)) { st &= ~OV2_LM_3D; return true; }
        return false;
    };
    auto cull = [&]() { return nobs < 3 && lm_kfid < j.newkf - 3 && !isobs; };   // :808-813, :868-874
    bool in_bad = !local || (j.lm_sel[l] & 4);
    if (local) {
        if (is_bad() || cull()) { remove(); return; }
        double wpt[3];
        bool have = true;
        if (inv) {
            const double rho = O.lm[idx], zanch = 1.0 / rho;
            if (zanch <= 0.0) { remove(); return; }
            const int a = O.lm_anchor_pose[idx];
            if (!M.kf_state[O.pose_kfid[a]]) { in_bad = true; have = false; }   // pkfanch == nullptr (:846-848)
            else {
                const double u = O.lm_anchor_uv[2 * idx], v = O.lm_anchor_uv[2 * idx + 1];
                const double cam[3] = {zanch * (u - j.K[2]) / j.K[0], zanch * (v - j.K[3]) / j.K[1], zanch};
                se3_apply(O.pose + 7 * a, cam, wpt);
            }
        } else {
            wpt[0] = O.lm[3 * idx]; wpt[1] = O.lm[3 * idx + 1]; wpt[2] = O.lm[3 * idx + 2];
        }
        if (have) {   // MapManager::updateMapPoint: a 2D point turns 3D with its keypoints, then MapPoint::setPoint
            j.lm_xyz_w[3 * l] = wpt[0]; j.lm_xyz_w[3 * l + 1] = wpt[1]; j.lm_xyz_w[3 * l + 2] = wpt[2];
            st |= OV2_LM_3D | OV2_LM_KP3D;
        }
    }
    if (in_bad && (is_bad() || cull())) { remove(); return; }
    if (st != (int)M.lm_state[l]) j.lm_state_w[l] = (unsigned char)st;
}

// gathers the headers of all maps (update counts) for one D2H
__global__ __launch_bounds__(64) void mb_hdr_gather_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    if (threadIdx.x < MH_N) j.hdr_out[threadIdx.x] = j.hdr[threadIdx.x];
}

// ov2_map_restore_state_batch: the tables of every map back to their saved state, one launch
__global__ __launch_bounds__(256) void mb_restore_kernel(const map_job *__restrict__ J)
{
    const map_job &j = J[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < 7 * (size_t)j.snap_kf; i += nth) j.kf_pose_w[i] = j.snap_kf_pose[i];
    for (size_t i = tid; i < 3 * (size_t)j.snap_lm; i += nth) j.lm_xyz_w[i] = j.snap_lm_xyz[i];
    for (size_t i = tid; i < (size_t)j.snap_kf; i += nth) j.kf_state_w[i] = j.snap_kf_state[i];
    for (size_t i = tid; i < (size_t)j.snap_lm; i += nth) j.lm_state_w[i] = j.snap_lm_state[i];
    for (size_t i = tid; i < (size_t)j.snap_obs; i += nth) j.obs_flag_w[i] = j.snap_obs_flag[i];
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

// one record of the job table
map_job job_of(const ov2_map *m, int newkf, int cur_kfid)
{
    map_job j;
    memset(&j, 0, sizeof(j));
    j.M = view_of(m);
    j.newkf = newkf; j.cur_kfid = cur_kfid;
    j.hdr = m->hdr; j.cov = m->cov; j.kf_role = m->kf_role; j.kf_idx = m->kf_idx; j.lm_nobs = m->lm_nobs; j.lm_sel = m->lm_sel;
    j.lm_anchor = m->lm_anchor; j.lm_flag = m->lm_flag; j.lm_pack = m->lm_pack; j.lm_pidx = m->lm_pidx; j.lm_new = m->lm_new;
    j.obs_cnt = m->obs_cnt; j.obs_off = m->obs_off; j.blk = m->blk;
    j.zero_blk = m->zero_blk; j.zero_vec16 = (unsigned)(m->zero_bytes / 16);
    j.out = m->out_dev; j.out_cap = m->out_cap;
    j.kf_pose_w = m->kf_pose; j.lm_xyz_w = m->lm_xyz; j.kf_state_w = m->kf_state; j.lm_state_w = m->lm_state; j.obs_flag_w = m->obs_flag;
    for (int k = 0; k < 4; ++k) j.K[k] = m->last_K[k];
    j.snap_kf_pose = m->snap_kf_pose; j.snap_lm_xyz = m->snap_lm_xyz; j.snap_kf_state = m->snap_kf_state;
    j.snap_lm_state = m->snap_lm_state; j.snap_obs_flag = m->snap_obs_flag;
    j.snap_kf = m->snap_kf; j.snap_lm = m->snap_lm; j.snap_obs = m->snap_obs;
    return j;
}

// The job table of a call lives in the ctx staging block: [B records | B gathered headers].  The records go up with one
// copy; the headers come back with one copy (fetch_headers) into the pinned twin.
struct job_table {
    ov2_ctx *c = nullptr; int B = 0;
    map_job *host = nullptr; const map_job *dev = nullptr;
    int *hdr_host = nullptr, *hdr_dev = nullptr;
    ov2_status begin(ov2_ctx *ctx, int nb)
    {
        c = ctx; B = nb;
        void *h, *d;
        const size_t tab = ((size_t)B * sizeof(map_job) + 255) & ~(size_t)255;
        const ov2_status s = ov2_staging(c, tab + (size_t)B * MH_N * sizeof(int) + 256, &h, &d);
        if (s != OV2_OK) return s;
        host = (map_job *)h; dev = (const map_job *)d;
        hdr_host = (int *)((unsigned char *)h + tab); hdr_dev = (int *)((unsigned char *)d + tab);
        return OV2_OK;
    }
    void set(int b, const map_job &j) { host[b] = j; host[b].hdr_out = hdr_dev + (size_t)b * MH_N; }
    ov2_status upload()
    {
        OV2_HIP(c, hipMemcpyAsync((void *)dev, host, (size_t)B * sizeof(map_job), hipMemcpyHostToDevice, c->stream));
        // calls that return without synchronising leave this copy in flight: the next user of the staging block waits for it
        OV2_HIP(c, hipEventRecord(c->stage_ev, c->stream));
        c->stage_ev_pending = true;
        return OV2_OK;
    }
    ov2_status fetch_headers()   // D2H of the gathered headers + the call's one synchronisation
    {
        OV2_HIP(c, hipMemcpyAsync(hdr_host, hdr_dev, (size_t)B * MH_N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        return OV2_OK;
    }
};

template <typename T>
ov2_status exclusive_scan(ov2_ctx *c, const job_table &T_, int which, int n_max)
{
    if (n_max <= 0) return OV2_OK;   // the totals already hold the zeros of the cleared header
    const dim3 g((n_max + 1023) / 1024, T_.B);
    OV2_LAUNCH(c, OV2_K_MAP, scan_block_kernel<T>, g, dim3(1024), 0, c->stream, T_.dev, which);
    OV2_LAUNCH(c, OV2_K_MAP, scan_top_kernel<T>, dim3(1, T_.B), dim3(1024), 0, c->stream, T_.dev, which);
    OV2_LAUNCH(c, OV2_K_MAP, scan_add_kernel<T>, g, dim3(1024), 0, c->stream, T_.dev, which);
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
    m->zero_bytes = (sizeof(int) * (MH_N + 2 * K + 3 * L) + L + 15) & ~(size_t)15;   // cleared 16 bytes at a time
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
    // the scratch arrays of the last set-up and a saved state belong to the old capacities
    m->last_valid = 0;
    void *snap[] = {m->snap_kf_pose, m->snap_lm_xyz, m->snap_kf_state, m->snap_lm_state, m->snap_obs_flag};
    for (void *p : snap) if (p) (void)hipFree(p);
    m->snap_kf_pose = m->snap_lm_xyz = nullptr; m->snap_kf_state = m->snap_lm_state = m->snap_obs_flag = nullptr;
    m->snap_kf = m->snap_lm = m->snap_obs = 0;
    return OV2_OK;
}

// squeezes the dead rows out of the observation table (stable); one synchronisation, fresh column arrays of the same capacity
static ov2_status compact_obs(ov2_map *m)
{
    ov2_ctx *c = m->c;
    hipStream_t st = c->stream;
    const int N = m->n_obs;
    if (N <= 0) return OV2_OK;
    // the fresh column arrays are released on every early return, kept once they have replaced the old ones
    struct fresh_cols {
        obs_cols O = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool committed = false;
        ~fresh_cols()
        {
            if (committed) return;
            void *p[] = {O.kf, O.lm, O.scale, O.uv, O.ruv, O.flag};
            for (void *q : p) if (q) (void)hipFree(q);
        }
    } F;
    obs_cols &O = F.O;
    const size_t cap = (size_t)m->max_obs;
    ov2_status s = OV2_OK;
#define A(p, n) if (s == OV2_OK) s = dmalloc(c, &O.p, (n))
    A(kf, cap); A(lm, cap); A(scale, cap); A(uv, 2 * cap); A(ruv, 2 * cap); A(flag, cap);
#undef A
    if (s != OV2_OK) return s;
    const map_view M = view_of(m);
    const dim3 g((N + 255) / 256), b(256);
    job_table JT;
    if ((s = JT.begin(c, 1)) != OV2_OK) return s;
    JT.set(0, job_of(m, -1, -1));
    if ((s = JT.upload()) != OV2_OK) return s;
    OV2_HIP(c, hipMemsetAsync(O.flag, 0, cap, st));
    OV2_LAUNCH(c, OV2_K_MAP, mc_mark_kernel, g, b, 0, st, M, m->obs_cnt);
    if ((s = exclusive_scan<int>(c, JT, SC_LIVE, N)) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, mc_scatter_kernel, g, b, 0, st, M, (const int *)m->obs_cnt, (const int *)m->obs_off, O);
    OV2_HIP(c, hipMemcpyAsync(m->hdr_host, m->hdr, MH_N * sizeof(int), hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    void *old[] = {m->obs_kf, m->obs_lm, m->obs_scale, m->obs_uv, m->obs_ruv, m->obs_flag};
    for (void *p : old) (void)hipFree(p);
    m->obs_kf = O.kf; m->obs_lm = O.lm; m->obs_scale = O.scale; m->obs_uv = O.uv; m->obs_ruv = O.ruv; m->obs_flag = O.flag;
    F.committed = true;
    m->n_obs = m->hdr_host[MH_NLIVE];
    m->n_compactions++;
    m->last_valid = 0;   // the row indices of the last set-up's scratch arrays are gone
    m->live_rows = m->n_obs; m->live_known = 1;
    if (m->snap_obs_flag) { (void)hipFree(m->snap_obs_flag); m->snap_obs_flag = nullptr; m->snap_obs = -1; }   // a saved state of the old rows
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
    void *snap[] = {m->snap_kf_pose, m->snap_lm_xyz, m->snap_kf_state, m->snap_lm_state, m->snap_obs_flag};
    for (void *p : snap) if (p) (void)hipFree(p);
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

// ---- set-up of B maps: the whole chain of launches without a synchronisation, then one for the gathered headers ----
namespace {

ov2_status grow_out(ov2_map *m, size_t need)
{
    ov2_ctx *c = m->c;
    if (need <= m->out_cap) return OV2_OK;
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    if (m->out_dev) OV2_HIP(c, hipFree(m->out_dev));
    m->out_dev = nullptr; m->out_cap = 0;
    const size_t want = need + need / 2 + 4096;
    if (hipMalloc((void **)&m->out_dev, want) != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "flat problem block of %zu bytes", want);
    m->out_cap = want;
    return OV2_OK;
}

void fill_view(const ov2_map *m, unsigned char *base, ov2_local_ba_setup *out)
{
    const int *H = m->last_hdr;
    memset(out, 0, sizeof(*out));
    out->aborted = H[MH_ABORT];
    if (out->aborted) return;
    size_t off[FO_N];
    flat_layout((size_t)H[MH_NPOSE], (size_t)H[MH_NLM], (size_t)H[MH_NRES], (size_t)H[MH_NBAD], m->last_inv ? 1 : 3, off);
    const flat_out O = flat_ptrs(base, off);
    out->n_pose = H[MH_NPOSE]; out->n_lm = H[MH_NLM]; out->n_res = H[MH_NRES]; out->n_bad = H[MH_NBAD];
    out->pose_kfid = O.pose_kfid; out->pose_const = O.pose_const; out->pose = O.pose;
    out->lm_lmid = O.lm_lmid; out->lm = O.lm; out->lm_anchor_pose = O.lm_anchor_pose; out->lm_anchor_uv = O.lm_anchor_uv;
    out->res_type = O.res_type; out->res_pose = O.res_pose; out->res_lm = O.res_lm; out->res_uv = O.res_uv; out->res_sigma = O.res_sigma;
    out->bad_lmid = O.bad_lmid;
    out->res_outlier = base == m->out_dev ? O.res_out : nullptr;   // device form only
}

// enqueues the set-up chain for the maps of `sel` (indices into maps[]), fetches their headers (one synchronisation)
ov2_status setup_pass(ov2_ctx *c, const std::vector<int> &sel, ov2_map *const *maps, const int32_t *newkf, int nmin_cov, int nmin_cst,
                      int inv)
{
    const int B = (int)sel.size();
    hipStream_t st = c->stream;
    job_table JT;
    ov2_status s = JT.begin(c, B);
    if (s != OV2_OK) return s;
    int nmax = 0, lmax = 0, kmax = 0; unsigned zmax = 0;
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[sel[b]];
        if (!m->out_dev && (s = grow_out(m, 1u << 20)) != OV2_OK) return s;
        JT.set(b, job_of(m, newkf[sel[b]], -1));
        nmax = std::max(nmax, m->n_obs); lmax = std::max(lmax, m->max_lm); kmax = std::max(kmax, m->max_kf);
        zmax = std::max(zmax, (unsigned)(m->zero_bytes / 16));
    }
    if ((s = JT.upload()) != OV2_OK) return s;
    const dim3 gN((std::max(nmax, 1) + 255) / 256, B), gL((lmax + 255) / 256, B), gK((kmax + 255) / 256, B), one(1, B), b(256);
    const map_job *J = JT.dev;
    OV2_LAUNCH(c, OV2_K_MAP, mb_zero_kernel, dim3(std::min(64u, (zmax + 255) / 256), B), b, 0, st, J);
    OV2_LAUNCH(c, OV2_K_MAP, ms_count_kernel, gN, b, 0, st, J);
    OV2_LAUNCH(c, OV2_K_MAP, ms_cov_kernel, gN, b, 0, st, J);
    OV2_LAUNCH(c, OV2_K_MAP, ms_select_kernel, one, dim3(64), 0, st, J, nmin_cov);
    OV2_LAUNCH(c, OV2_K_MAP, ms_local_lm_kernel, gN, b, 0, st, J);
    OV2_LAUNCH(c, OV2_K_MAP, ms_observers_kernel, gN, b, 0, st, J);
    OV2_LAUNCH(c, OV2_K_MAP, ms_poses_kernel, one, dim3(1024), 0, st, J, nmin_cst);
    OV2_LAUNCH(c, OV2_K_MAP, ms_lm_flags_kernel, gL, b, 0, st, J, inv);
    // one 64-bit scan numbers the landmarks (low word) and the bad list (high word); its total lands on NLM | NBAD
    if ((s = exclusive_scan<unsigned long long>(c, JT, SC_LM, lmax)) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, ms_res_count_kernel, gN, b, 0, st, J, inv);
    if ((s = exclusive_scan<int>(c, JT, SC_RES, nmax)) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, me_all_kernel, dim3(gK.x + gL.x + gN.x, B), b, 0, st, J, (int)gK.x, (int)gL.x, inv);
    if ((s = JT.fetch_headers()) != OV2_OK) return s;
    for (int b2 = 0; b2 < B; ++b2) {
        ov2_map *m = maps[sel[b2]];
        memcpy(m->last_hdr, JT.hdr_host + (size_t)b2 * MH_N, MH_N * sizeof(int));
        m->last_newkf = newkf[sel[b2]]; m->last_inv = inv; m->last_valid = 1;
        m->live_rows = m->last_hdr[MH_NLIVE]; m->live_known = 1;
    }
    return OV2_OK;
}

ov2_status setup_batch_impl(ov2_ctx *c, int B, ov2_map *const *maps, const int32_t *newkf, int nmin_cov, int nmin_cst, int inv,
                            const double *calib_l, bool squeeze_first)
{
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[b];
        if (!m || m->c != c) return ov2_set_err(c, OV2_ERR_INVALID, "map %d of the batch is null or belongs to another context", b);
        if (newkf[b] < 0 || newkf[b] >= m->max_kf) return ov2_set_err(c, OV2_ERR_INVALID, "kfid %d outside the capacity of map %d", newkf[b], b);
        for (int q = 0; q < b; ++q)
            if (maps[q] == m) return ov2_set_err(c, OV2_ERR_INVALID, "map %d appears twice in the batch", b);
        m->have_K = calib_l != nullptr;
        for (int k = 0; k < 4; ++k) m->last_K[k] = calib_l ? calib_l[4 * b + k] : 0.0;
    }
    OV2_HIP(c, hipSetDevice(c->device));
    if (squeeze_first)   // the batched form squeezes a mostly dead table BEFORE the chain: its update stage needs the row indices after it
        for (int b = 0; b < B; ++b) {
            ov2_map *m = maps[b];
            if (m->live_known && m->n_obs >= MAP_COMPACT_MIN_ROWS && 2 * (long long)m->live_rows < m->n_obs) {
                const ov2_status cs = compact_obs(m);
                if (cs != OV2_OK) return cs;
            }
        }
    std::vector<int> sel((size_t)B);
    for (int b = 0; b < B; ++b) sel[b] = b;
    for (int pass = 0; pass < 3 && !sel.empty(); ++pass) {
        ov2_status s = setup_pass(c, sel, maps, newkf, nmin_cov, nmin_cst, inv);
        if (s != OV2_OK) return s;
        // a flat problem that did not fit its block: grow the block, run that map again (first calls / growing windows only)
        std::vector<int> again;
        for (int b : sel) {
            ov2_map *m = maps[b];
            if (!m->last_hdr[MH_OVER]) continue;
            if ((s = grow_out(m, (size_t)m->last_hdr[MH_NEED16] << 4)) != OV2_OK) return s;
            again.push_back(b);
        }
        sel.swap(again);
    }
    if (!sel.empty()) return ov2_set_err(c, OV2_ERR_NOMEM, "flat problem blocks kept overflowing");
    return OV2_OK;
}

}  // namespace

extern "C" ov2_status ov2_map_local_ba_setup_batch(ov2_ctx *c, int B, ov2_map *const *maps, const int32_t *newkf, int nmin_covscore,
                                                   int nmin_cst_kfs, int inv_depth, const double *calib_l, ov2_local_ba_setup *dev)
{
    if (!c) return OV2_ERR_INVALID;
    if (B == 0) return OV2_OK;
    if (B < 0 || !maps || !newkf || !dev) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_local_ba_setup_batch: null argument");
    const ov2_status s = setup_batch_impl(c, B, maps, newkf, nmin_covscore, nmin_cst_kfs, inv_depth ? 1 : 0, calib_l, true);
    if (s != OV2_OK) return s;
    for (int b = 0; b < B; ++b) fill_view(maps[b], maps[b]->out_dev, &dev[b]);
    return OV2_OK;
}

extern "C" ov2_status ov2_map_local_ba_setup(ov2_map *m, int newkf, int nmin_covscore, int nmin_cst_kfs, int inv_depth,
                                             const double *calib_l, ov2_local_ba_setup *out)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    if (!out) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_local_ba_setup: null output");
    memset(out, 0, sizeof(*out));
    const int32_t nk = newkf;
    ov2_status s = setup_batch_impl(c, 1, &m, &nk, nmin_covscore, nmin_cst_kfs, inv_depth ? 1 : 0, calib_l, false);
    if (s != OV2_OK) return s;
    const int *H = m->last_hdr;
    const int N = m->n_obs;
    // mostly tombstones left: squeeze the table once this set-up has read it (amortised like a growth)
    const bool squeeze = N >= MAP_COMPACT_MIN_ROWS && 2 * (long long)H[MH_NLIVE] < N;
    if (H[MH_ABORT]) { out->aborted = 1; return squeeze ? compact_obs(m) : OV2_OK; }
    // the host form: the arrays of the flat problem into the pinned mirror (second synchronisation)
    size_t off[FO_N];
    flat_layout((size_t)H[MH_NPOSE], (size_t)H[MH_NLM], (size_t)H[MH_NRES], (size_t)H[MH_NBAD], inv_depth ? 1 : 3, off);
    const size_t bytes = off[FO_HOST_END];
    if (bytes > m->out_host_cap) {
        if (m->out_host) OV2_HIP(c, hipHostFree(m->out_host));
        m->out_host = nullptr; m->out_host_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void **)&m->out_host, want, hipHostMallocDefault) != hipSuccess)
            return ov2_set_err(c, OV2_ERR_NOMEM, "pinned flat problem of %zu bytes", want);
        m->out_host_cap = want;
    }
    if (bytes) OV2_HIP(c, hipMemcpyAsync(m->out_host, m->out_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    fill_view(m, m->out_host, out);
    return squeeze ? compact_obs(m) : OV2_OK;
}

// The same flat problem, where the set-up left it on the device (ov2_ba_solve_batch_dev takes these pointers as they are)
extern "C" ov2_status ov2_map_setup_device_view(const ov2_map *m, const ov2_local_ba_setup *host, ov2_local_ba_setup *dev)
{
    if (!m || !host || !dev) return OV2_ERR_INVALID;
    if (host->aborted) { *dev = *host; return OV2_OK; }
    const unsigned char *q = (const unsigned char *)host->pose_kfid;
    if (!m->out_dev || !(m->out_host && q >= m->out_host && q < m->out_host + m->out_host_cap)) return OV2_ERR_INVALID;   // not the arrays of this map's last set-up
    if (host->n_pose != m->last_hdr[MH_NPOSE] || host->n_lm != m->last_hdr[MH_NLM] || host->n_res != m->last_hdr[MH_NRES])
        return OV2_ERR_INVALID;
    fill_view(m, m->out_dev, dev);
    return OV2_OK;
}

// ---- update stage of B maps --------------------------------------------------------------------------------------
extern "C" ov2_status ov2_map_local_ba_update_batch(ov2_ctx *c, int B, ov2_map *const *maps, const uint8_t *const *d_outlier,
                                                    const int32_t *cur_kfid, ov2_local_ba_update *out)
{
    if (!c) return OV2_ERR_INVALID;
    if (B == 0) return OV2_OK;
    if (B < 0 || !maps) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_local_ba_update_batch: null argument");
    int inv = -1;
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[b];
        if (!m || m->c != c) return ov2_set_err(c, OV2_ERR_INVALID, "map %d of the batch is null or belongs to another context", b);
        if (!m->last_valid)
            return ov2_set_err(c, OV2_ERR_INVALID, "map %d: no set-up to update from (the table changed since, or none ran)", b);
        if (inv < 0) inv = m->last_inv;
        if (m->last_inv != inv) return ov2_set_err(c, OV2_ERR_INVALID, "the maps of a batch must share one landmark parametrisation (map %d)", b);
        if (inv && !m->have_K && !m->last_hdr[MH_ABORT])
            return ov2_set_err(c, OV2_ERR_INVALID, "map %d: the inverse-depth update needs the left intrinsics (calib_l of the set-up)", b);
    }
    OV2_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    job_table JT;
    ov2_status s = JT.begin(c, B);
    if (s != OV2_OK) return s;
    int nmax = 0, pmax = 0, lmax = 0;
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[b];
        map_job j = job_of(m, m->last_newkf, cur_kfid ? cur_kfid[b] : -1);
        j.outlier = d_outlier ? d_outlier[b] : nullptr;
        JT.set(b, j);
        if (m->last_hdr[MH_ABORT]) continue;
        nmax = std::max(nmax, m->n_obs); pmax = std::max(pmax, m->last_hdr[MH_NPOSE]);
        lmax = std::max(lmax, m->last_hdr[MH_NLM] + m->last_hdr[MH_NBAD]);
    }
    if ((s = JT.upload()) != OV2_OK) return s;
    const dim3 gN((std::max(nmax, 1) + 255) / 256, B), b256(256);
    const int gP = (pmax + 255) / 256, gLM = (lmax + 255) / 256;
    OV2_LAUNCH(c, OV2_K_MAP, mu_obs_kernel, gN, b256, 0, st, JT.dev, inv);
    if (!inv) OV2_LAUNCH(c, OV2_K_MAP, mu_reanchor_kernel, gN, b256, 0, st, JT.dev);
    if (gP + gLM > 0) OV2_LAUNCH(c, OV2_K_MAP, mu_apply_kernel, dim3(gP + gLM, B), b256, 0, st, JT.dev, gP, inv);
    if (!out) return OV2_OK;   // asynchronous: the caller did not ask for what to replay
    OV2_LAUNCH(c, OV2_K_MAP, mb_hdr_gather_kernel, dim3(1, B), dim3(64), 0, st, JT.dev);
    if ((s = JT.fetch_headers()) != OV2_OK) return s;
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[b];
        const int *H = JT.hdr_host + (size_t)b * MH_N;
        memset(&out[b], 0, sizeof(out[b]));
        if (m->last_hdr[MH_ABORT]) continue;
        size_t off[FO_N];
        flat_layout((size_t)m->last_hdr[MH_NPOSE], (size_t)m->last_hdr[MH_NLM], (size_t)m->last_hdr[MH_NRES], (size_t)m->last_hdr[MH_NBAD],
                    inv ? 1 : 3, off);
        const flat_out O = flat_ptrs(m->out_dev, off);
        out[b].n_removed_lm = H[MH_NRM_LM]; out[b].n_removed_obs = H[MH_NRM_OBS]; out[b].n_stereo_off = H[MH_NST_OFF];
        out[b].removed_lmid = O.rm_lm; out[b].removed_obs = (const int32_t *)O.rm_obs; out[b].stereo_off = (const int32_t *)O.st_off;
    }
    return OV2_OK;
}

// ---- saved state (bench / tests: every job starts from the same map) ----------------------------------------------
extern "C" ov2_status ov2_map_save_state(ov2_map *m)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    OV2_HIP(c, hipSetDevice(c->device));
    void *old[] = {m->snap_kf_pose, m->snap_lm_xyz, m->snap_kf_state, m->snap_lm_state, m->snap_obs_flag};
    for (void *p : old) if (p) (void)hipFree(p);
    m->snap_kf_pose = m->snap_lm_xyz = nullptr; m->snap_kf_state = m->snap_lm_state = m->snap_obs_flag = nullptr;
    const size_t K = m->max_kf, L = m->max_lm, N = m->n_obs;
    ov2_status s = OV2_OK;
    if (s == OV2_OK) s = dmalloc(c, &m->snap_kf_pose, 7 * K);
    if (s == OV2_OK) s = dmalloc(c, &m->snap_lm_xyz, 3 * L);
    if (s == OV2_OK) s = dmalloc(c, &m->snap_kf_state, K);
    if (s == OV2_OK) s = dmalloc(c, &m->snap_lm_state, L);
    if (s == OV2_OK) s = dmalloc(c, &m->snap_obs_flag, N);
    if (s != OV2_OK) return s;
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemcpyAsync(m->snap_kf_pose, m->kf_pose, 7 * K * 8, hipMemcpyDeviceToDevice, st));
    OV2_HIP(c, hipMemcpyAsync(m->snap_lm_xyz, m->lm_xyz, 3 * L * 8, hipMemcpyDeviceToDevice, st));
    OV2_HIP(c, hipMemcpyAsync(m->snap_kf_state, m->kf_state, K, hipMemcpyDeviceToDevice, st));
    OV2_HIP(c, hipMemcpyAsync(m->snap_lm_state, m->lm_state, L, hipMemcpyDeviceToDevice, st));
    if (N) OV2_HIP(c, hipMemcpyAsync(m->snap_obs_flag, m->obs_flag, N, hipMemcpyDeviceToDevice, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    m->snap_kf = (int)K; m->snap_lm = (int)L; m->snap_obs = (int)N;
    return OV2_OK;
}

extern "C" ov2_status ov2_map_restore_state_batch(ov2_ctx *c, int B, ov2_map *const *maps)
{
    if (!c) return OV2_ERR_INVALID;
    if (B == 0) return OV2_OK;
    if (B < 0 || !maps) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_map_restore_state_batch: null argument");
    for (int b = 0; b < B; ++b) {
        ov2_map *m = maps[b];
        if (!m || m->c != c) return ov2_set_err(c, OV2_ERR_INVALID, "map %d of the batch is null or belongs to another context", b);
        if (!m->snap_kf_pose || m->snap_kf != m->max_kf || m->snap_lm != m->max_lm || m->snap_obs != m->n_obs)
            return ov2_set_err(c, OV2_ERR_INVALID, "map %d: no saved state, or the tables changed shape since it was saved", b);
    }
    OV2_HIP(c, hipSetDevice(c->device));
    job_table JT;
    ov2_status s = JT.begin(c, B);
    if (s != OV2_OK) return s;
    for (int b = 0; b < B; ++b) { JT.set(b, job_of(maps[b], -1, -1)); maps[b]->last_valid = 0; }
    if ((s = JT.upload()) != OV2_OK) return s;
    OV2_LAUNCH(c, OV2_K_MAP, mb_restore_kernel, dim3(64, B), dim3(256), 0, c->stream, JT.dev);
    // the job table must have been read before the staging block is reused: the next call that stages waits for this
    // stream anyway (same stream, in order), and a staging reallocation synchronises first
    return OV2_OK;
}

// ---- the tables to host memory (tests) ----------------------------------------------------------------------------
extern "C" ov2_status ov2_map_download(ov2_map *m, int *n_kf, int *n_lm, int *n_obs, double *kf_pose, uint8_t *kf_state, double *lm_xyz,
                                       uint8_t *lm_state, int32_t *obs_kf, int32_t *obs_lm, uint8_t *obs_flag, double *obs_uv, double *obs_ruv)
{
    if (!m || !m->c) return OV2_ERR_INVALID;
    ov2_ctx *c = m->c;
    OV2_HIP(c, hipSetDevice(c->device));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    if (n_kf) *n_kf = m->max_kf;
    if (n_lm) *n_lm = m->max_lm;
    if (n_obs) *n_obs = m->n_obs;
    const size_t K = m->max_kf, L = m->max_lm, N = m->n_obs;
#define DL(dst, src, bytes) if ((dst) && (bytes)) OV2_HIP(c, hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToHost))
    DL(kf_pose, m->kf_pose, 7 * K * 8); DL(kf_state, m->kf_state, K); DL(lm_xyz, m->lm_xyz, 3 * L * 8); DL(lm_state, m->lm_state, L);
    DL(obs_kf, m->obs_kf, N * 4); DL(obs_lm, m->obs_lm, N * 4); DL(obs_flag, m->obs_flag, N); DL(obs_uv, m->obs_uv, 2 * N * 8);
    DL(obs_ruv, m->obs_ruv, 2 * N * 8);
#undef DL
    return OV2_OK;
}
