// vs_ba_internal.h -- types, small math and kernel prototypes shared by the bundle-adjustment, PnP and tracking-session
// translation units of libvslam_hip.so (gfx950 only).
#pragma once
#include "vs_internal.h"

#include <math.h>

namespace vsba {

constexpr int kCamStride = 19;  // t[3] q[4] w2n[12]
constexpr int kMoThreads = 512;  // threads per camera workgroup of the motion-only step (one observation per thread at ~500 matches)
constexpr int kPnpFinish = 256;
constexpr int kPnpModel = 32;        // doubles per hypothesis model row
constexpr int kPnpHypIters = 5;      // LM iterations of a RANSAC hypothesis at most (specification, oracle alike: vo_pnp_ransac)
constexpr double kPnpStep2 = 1e-14;  // the PnP LM stops on a step with |x|^2 below this (specification, oracle alike)

struct lm_state {
  double lambda, ni, current_chi, temp_chi, rho, scale_pose, chi0;
  int cur;       // index of the state buffer holding the accepted estimate
  int it;        // outer iterations finished
  int trials, qmax, not_pd;
  int need_lin;  // 1: the next slot starts with a linearisation
  int done, terminated, solve_ok;
  int accepted;  // steps accepted so far in this solve
};

struct ba_dev {
  int n_poses, n_points, n_obs, n_scale, nfp, nfl, np, n_act;  // n_act: points with >= 1 active observation
  int ns, nb_pt, mmax, has_info, dups, max_it, lds_slab, pad0;
  double fx, fy, cx, cy, huber, dcs;
  const int *pose_slot, *pt_slot, *act_pt, *pt_start, *o_cam, *o_pt, *cam_start, *cam_obs;
  const int* cam_pt;  // [cam_obs entries] point index of the observation
  const int *o_hpl, *fp_start, *fp_slot, *slot_pose;  // slot_pose[free camera slot] = pose index  // Hpl block index of an observation (-1: none); per free point: its blocks
  const double *o_uv, *o_info;
  const int *sc_parent, *sc_child;
  const double* sc_meas;
  double* cam[2];
  double* pts[2];
  double *Hpp, *bp, *Hll, *bl, *Hpl, *Dinv, *slab, *S, *bs, *xp;
  const unsigned long long* fp_mask;  // per free point: bit t set iff it is observed by a camera of tile row t
  double* Dbl;     // [nfl][3] (Hll + lambda I)^-1 bl (tiled Schur)
  int ntile;       // tiles of kTileCams cameras per side (tiled Schur), 0 otherwise
  int small;       // 1: ba_schur_small produced the slabs (single tile, lower block triangle only)
  int win, win_per, win_n;  // 1: ba_schur_window (banded window); points per slab; contributing free points
  const int *win_order, *win_w0, *win_len;  // their order by lowest camera slot; per slab: first camera slot, cameras
  const int* win_first;  // [nfp + 1] first slab whose window starts at this camera slot or later
  int band;  // > 0: S[r][c] = 0 for r - c >= band (banded windows without scale edges): ba_chol_band instead of the panel launches
  int cam_split;   // workgroups that share one camera in the linearisation's camera role
  double* cam_part;       // [nfp][cam_split][27] their partial sums
  unsigned* cam_ticket;   // [nfp] arrival counters
  int spec;        // 1: two linearisations, indexed like the state buffers (ba_point_trial linearises the trial state)
  double *Hpp1, *bp1, *Hll1, *bl1, *Hpl1;  // the second linearisation (spec)
  double* rinv;    // [np] reciprocal Cholesky pivots (large systems)
  int* chol_fail;  // set by a panel kernel that met a non-positive pivot
  double *part_chi, *part_scale, *part_maxd;  // per point-block partials; part_maxd has nb_pt + nfp entries
  double *chi_trace, *lambda_trace;
  double* trial_trace;  // [trial_cap][4] per-trial rows (lambda, trial chi2, rho, solve ok) or nullptr (test aid)
  int trial_cap;
  int max_rank;         // duplicate observations: highest repeat count of one camera within one point (0: none)
  const int* fp_rank;   // per Hpl block: how many earlier blocks of the same point belong to the same camera
  lm_state* st;
  unsigned* trial_ticket;  // arrival counter of ba_point_trial's workgroups (the last one takes the LM decision)
  // motion-only kernel: camera-major observation copy cut into chunks of 64
  const double *mo_X, *mo_uv, *mo_info;  // [n_obs_free_cam][3|2|3] in camera-major (cam_start) order
  double *mo_part, *mo_H;                // [2][nfp][4] per-camera partials by step parity; [nfp][42] H upper + b
  unsigned long long* mo_box;            // ba_motion_persistent: [2][kMoPersistCameras][8] tagged mailbox words
  unsigned mo_epoch;                     // ... tag of this solve (20 bits), so that words of an earlier solve never match
  unsigned long long* mo_stamps;         // diagnostic (vs_mo_profile): [64 steps][8] shader-clock stamps of camera 0's workgroup, or nullptr
  unsigned* mo_done;                     // chained tracking: [0] arrival counter, [64] tag published when every camera workgroup is through (nullptr: none)
  unsigned mo_done_tag;
};

// motion-only LM state (double-buffered by launch parity, see vs_ba.hip)
struct mo_state {
  double lambda, ni, current_chi, chi0;
  int cur, it, trials, qmax, not_pd, need_lin, done, terminated;
  int stage;  // 0: nothing yet, 1: the previous launch only linearised (iteration 0), 2: it ran a trial
  int seq;
};

// Reciprocal / reciprocal square root from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26) + two Newton steps: <= 1-2
// ulp, a third of the instructions of the IEEE divide / sqrt sequences.  For the serial stretches of latency-bound
// kernels (6x6 pivots, quaternion normalisation); inputs are positive normal numbers there.
__device__ __forceinline__ double vs_fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double vs_fast_rsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
  r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
  return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence also waits
// for every global load and store the wave has in flight (s_waitcnt vmcnt(0)): a kernel that streams results to HBM and
// requests rows ahead while its waves meet through LDS would pay a memory round trip at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Xor-butterfly sum (or maximum) over aligned groups of 2^STEPS lanes, every lane of a group ending with the same bits -- the
// bits `for (d = 1; d < 2^STEPS; d <<= 1) v += __shfl_xor(v, d)` gives, without its two ds_bpermute round trips per step:
// partners lane^1 and lane^2 by quad permutes, lane^4 by the half-row mirror and lane^8 by the row mirror (lane 7 - i / 15 - i
// of the group holds what lane i^4 / i^8 holds once the smaller groups are uniform); across the four rows of 16 the row values
// are read out and combined as the butterfly combines them, (r0 + r1) + (r2 + r3) -- addition and fmax commute.
template <int STEPS, bool MAX = false>
__device__ __forceinline__ double vs_group_reduce(double x) {
  static_assert(STEPS >= 1 && STEPS <= 6 && STEPS != 5, "groups of 2, 4, 8, 16 lanes or the whole wave");
#define VS_DPP_STEP(ctrl)                                                                                      \
  {                                                                                                            \
    const double o_ = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, false), \
                                       __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, false)); \
    x = MAX ? fmax(x, o_) : x + o_;                                                                            \
  }
  if (STEPS >= 1) VS_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
  if (STEPS >= 2) VS_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
  if (STEPS >= 3) VS_DPP_STEP(0x141)  // row_half_mirror
  if (STEPS >= 4) VS_DPP_STEP(0x140)  // row_mirror
#undef VS_DPP_STEP
  if (STEPS == 6) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    x = MAX ? fmax(fmax(r0, r1), fmax(r2, r3)) : (r0 + r1) + (r2 + r3);
  }
  return x;
}

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __host__ inline void quat_to_w2n(const double* t, const double* q, double* w /*[12]*/) {
  const double x = q[0], y = q[1], z = q[2], ww = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * ww, twy = ty * ww, twz = tz * ww, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  double R[3][3];
  R[0][0] = 1 - (tyy + tzz);
  R[0][1] = txy - twz;
  R[0][2] = txz + twy;
  R[1][0] = txy + twz;
  R[1][1] = 1 - (txx + tzz);
  R[1][2] = tyz - twx;
  R[2][0] = txz - twy;
  R[2][1] = tyz + twx;
  R[2][2] = 1 - (txx + tyy);
  for (int i = 0; i < 3; ++i) {
    w[4 * i + 0] = R[0][i];
    w[4 * i + 1] = R[1][i];
    w[4 * i + 2] = R[2][i];
    w[4 * i + 3] = -(w[4 * i + 0] * t[0] + w[4 * i + 1] * t[1] + w[4 * i + 2] * t[2]);
  }
}

// Eigen's matrix -> quaternion, then SE3Quat::normalizeRotation (w >= 0, unit norm)   [setup only]
__device__ __host__ inline void quat_from_pose(const double* m, double* q) {
#define M(r, c) m[(r)*4 + (c)]
  const double tr = M(0, 0) + M(1, 1) + M(2, 2);
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (M(2, 1) - M(1, 2)) * s;
    q[1] = (M(0, 2) - M(2, 0)) * s;
    q[2] = (M(1, 0) - M(0, 1)) * s;
  } else {
    // largest diagonal element i, j = (i + 1) % 3, k = (j + 1) % 3 -- written out per i: with run-time indices the matrix
    // and the quaternion live in scratch memory on the device (same expressions, same order of operations)
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if ((i == 0 && M(2, 2) > M(0, 0)) || (i == 1 && M(2, 2) > M(1, 1))) i = 2;
    if (i == 0) {
      double s = sqrt(M(0, 0) - M(1, 1) - M(2, 2) + 1.0);
      q[0] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (M(2, 1) - M(1, 2)) * s;
      q[1] = (M(1, 0) + M(0, 1)) * s;
      q[2] = (M(2, 0) + M(0, 2)) * s;
    } else if (i == 1) {
      double s = sqrt(M(1, 1) - M(2, 2) - M(0, 0) + 1.0);
      q[1] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (M(0, 2) - M(2, 0)) * s;
      q[2] = (M(2, 1) + M(1, 2)) * s;
      q[0] = (M(0, 1) + M(1, 0)) * s;
    } else {
      double s = sqrt(M(2, 2) - M(0, 0) - M(1, 1) + 1.0);
      q[2] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (M(1, 0) - M(0, 1)) * s;
      q[0] = (M(0, 2) + M(2, 0)) * s;
      q[1] = (M(1, 2) + M(2, 1)) * s;
    }
  }
#undef M
  if (q[3] < 0.0) {
    q[0] = -q[0];
    q[1] = -q[1];
    q[2] = -q[2];
    q[3] = -q[3];
  }
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= nrm;
  q[1] /= nrm;
  q[2] /= nrm;
  q[3] /= nrm;
}

// arguments of the PnP-RANSAC kernels (vs_pnp.hip); also launched by the tracking session (vs_track.hip)
struct pnp_args {
  const double* obj;   // [n][3]
  const double* img;   // [n][2]
  int n, iters_lm, iterations;
  int obj_f32;         // 1: object points are rounded to float32 when read (the reference passes objectPoints.astype(np.float32))
  double fx, fy, cx, cy, thr2, confidence;
  unsigned long long seed;
  double cam0[kCamStride];
  double* model_out;   // [H][kPnpModel]: rows of [R|t] (camera-to-world, 12), the record re-derived from it (19), pad; 256-byte rows
  unsigned long long* tag;  // [H] (call epoch << 32) | inlier count of hypothesis h, published when its model is complete
  unsigned epoch;      // this call's epoch (never 0; words of earlier calls never carry it)
  double* host_result;  // optional pinned mirror of `result` [20] (class-API period), or nullptr
  int* host_inl;        // ... and of inl_out
  unsigned* host_tag_word;  // ... and the word (pinned) that receives host_tag once both are complete: the host polls it
  unsigned host_tag;        //     instead of waiting for an event behind the launch
  unsigned long long* stamps;  // diagnostic (vs_pnp_profile): wall-clock stamps [H + 1][8] of the roles' phases, or nullptr
  double* result;      // [20]: pose 4x4, found, inliers, best hypothesis, hypotheses used
  int* inl_out;        // [n]
  const int* n_dev;    // tracking session: the number of correspondences lives on the device (nullptr: use n)
  double* rec_out[2];  // tracking session: camera record of the result (the guess if nothing was found), or nullptr
  mo_state* lm_init;   // tracking session: the two motion-only LM records to reset for the solve that follows, or nullptr
  int lm_cur;          // ... their state-buffer index
  // chained tracking (vs_track_frame_pipelined): what the host would know only after the previous frame's results are read
  // on the device instead, so that this launch can be enqueued while the previous back half is still running
  const int* off_dev;        // row offset of this frame's correspondences in obj / img, or nullptr (obj / img point at them)
  const double* guess_dev[2];  // the guess: the previous frame's camera record in either state buffer (the one `cur` names), or nullptr (cam0)
  const mo_state* cur_dev;   // record whose `cur` is the state-buffer index, or nullptr (lm_cur)
  const unsigned* front_tag_dev;  // word the frame's front half (another stream) sets to front_tag when its append is complete:
  unsigned front_tag;             // ... every workgroup waits for it here instead of a stream-level event wait in front of the launch
  const unsigned* back_tag_dev;   // likewise the previous frame's motion-only solve, which runs on ANOTHER stream: this launch is
  unsigned back_tag;              // resident and has sampled when that solve ends (nullptr: stream order already says so)
  // ... and the PREVIOUS frame's read-back rides along: the finishing workgroup, which has nothing to do until the hypotheses
  // report, first copies that frame's result block to pinned host memory and tags it (what track_publish_kernel does)
  const uint4* pub_src;
  uint4* pub_dst;
  int pub_n16;
  unsigned* pub_tag_word;
  unsigned pub_tag;
};

__global__ __launch_bounds__(kMoThreads) void ba_motion_step(ba_dev D, int step);
template <bool OVF>
__global__ __launch_bounds__(kMoThreads) void ba_motion_persistent(ba_dev D, int max_steps);
bool mo_persistent_ok(vs_ctx* ctx, int cameras, int max_steps);  // vs_ba.hip: may the one-launch form run on this device?
constexpr int kMoPersistCameras = 64;          // limit of the one-launch form: workgroups that have to be resident together
constexpr int kMoPersistObs = 2 * kMoThreads;  // observations per camera its threads keep in registers (more: the <true> instantiation)
__global__ __launch_bounds__(kPnpFinish) void pnp_ransac_kernel(pnp_args P);
int pnp_grid(int iterations);  // workgroups of pnp_ransac_kernel: four hypotheses each + the finishing one
int pnp_tags(vs_ctx* ctx, int iterations, hipStream_t s, unsigned long long** tag, unsigned* epoch);  // vs_pnp.hip
int pnp_stamps(vs_ctx* ctx, int iterations, hipStream_t s, unsigned long long** stamps);

// ---- vs_ba_build.hip: the structure arrays of a large problem produced on the device from the device copy of the caller's
// observation list (what vs_ba_solve's host passes produce otherwise; same arrays, bit for bit)
constexpr int kBuildMaxKeys = 1022;      // free cameras + 1 that the stable counting sort takes (its histogram lives in LDS)
constexpr int kBuildMaxPerPoint = 512;   // observations per point up to which the duplicate-camera scan runs on the device
// words of ba_build::info (zeroed by ba_build_enqueue)
enum { kBuildBad = 0, kBuildUngrouped, kBuildInactive, kBuildDups, kBuildHpl, kBuildMmax, kBuildCamMax, kBuildWinN, kBuildWinPer, kBuildWinSlabs,
       kBuildWinCams, kBuildInfoInts = 16 };
struct ba_build {
  // in (device)
  const int *o_cam, *o_pt, *pose_slot, *pt_slot;
  int n_obs, F, P, nfp, nfl, tile_cams;
  int win_target, win_per_tune, win_per_max, ns_cap;  // window plan: slabs wanted, vs_tune_ba's slab size (0: none), kWinPerMax, capacity of win_w0 / win_len
  // out (device)
  int *pt_start, *act_pt, *o_hpl, *fp_start, *fp_slot, *cam_start, *cam_obs, *cam_pt;
  unsigned long long* fp_mask;                         // or nullptr
  int *win_order, *win_w0, *win_len, *win_first;       // win_order == nullptr: no window plan
  int* info;                                           // [kBuildInfoInts]
  // temporaries (device)
  int *ckey, *wkey, *wlo, *whi, *hist_o, *tot_o, *hist_p, *tot_p, *win_start;
};
size_t ba_build_temp_ints(int n_obs, int nfp, int nfl);
int ba_build_enqueue(vs_ctx* ctx, hipStream_t s, const ba_build& B);

}  // namespace vsba
