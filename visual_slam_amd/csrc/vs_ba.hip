// vs_ba.hip -- local / motion-only bundle adjustment: Levenberg-Marquardt over SE3 cameras and 3-D points with the
// points marginalised by a Schur complement, FP64 throughout (gfx950).
//
// Replaces what the reference delegates to g2o (src/v2/LocalBA.py:20-94,115-131,39-42): SBACam/VertexCam poses,
// VertexSBAPointXYZ points, EdgeProjectP2MC + Huber, EdgeSBAScale + DCS, BlockSolverSE3 + Cholesky,
// OptimizationAlgorithmLevenberg.  The arithmetic follows oracle/vs_oracle.c operation for operation where the
// parallel decomposition allows it (-ffp-contract=off), so poses agree with the oracle to ~1e-10, far inside the
// 1e-4 contract.
//
// Data flow (everything stays in HBM/L2 for the whole solve; the host only polls a `done` flag per batch of launches):
//  general problem (free points => Schur complement), one "slot" per LM trial, every kernel predicated on the
//  device-resident LM state:
//   ba_linearize      grid = point blocks + one block per free camera (one launch, two roles)
//       point block : 8 lanes share a point; each takes every 8th observation (CSR by point), Hll / bl are summed over
//                     the lanes with xor-shuffles, the 6x3 blocks Hpl are stored contiguously per free point, robust
//                     chi2 partial per block
//       camera block: workgroup = free camera; 512 threads stride over the camera's observation list, 27 register
//                     accumulators (Hpp upper 21 + bp 6), fixed-order LDS tree reduction; also the EdgeSBAScale terms
//   ba_lambda_init    first slot only: chi2_0 and lambda_0 = 1e-5 * max diag(H)
//   ba_dinv           (Hll + lambda I)^-1 and its product with bl, thread = free point (windows of several tiles only;
//                     with a single tile ba_schur_tile computes it for its own points)
//   ba_schur_tile     workgroup = (slab of points, tile of 10 x 10 camera blocks); the tile is accumulated in registers:
//                     wave w owns the row cameras w, w+4, w+8, lane (block row, column camera) owns 6 elements per row
//                     camera -- fixed ownership, no atomics, no LDS read-modify-write; Hpl blocks staged 8 points per
//                     barrier (ba_schur<LDS_SLAB> remains for duplicate observations of one camera: LDS atomics there)
//   ba_schur_window   windows of several tiles whose points are seen from neighbouring cameras only (a banded system): the
//                     points ordered by lowest camera, workgroup = slab of that order, the slab's dense 96 x 96 window of S
//                     accumulated on the FP64 matrix cores from two K x 96 LDS panels; ba_reduce_window sums the slabs
//   ba_reduce         S = Hpp + lambda I - sum of slabs, rhs likewise (four slab groups per element, fixed order)
//   dense solve       n <= 126: ba_solve_block, one workgroup, [S | rhs] in LDS, blocked by the 6x6 camera blocks,
//                     blocked back-substitution, trial camera states;  beyond: ba_chol_panel + ba_chol_update per block
//                     column in HBM -- banded systems: ba_chol_band, the same arithmetic in one launch -- then
//                     ba_chol_finish;  ba_solve<false> (element-wise, one workgroup) is the last
//                     resort when even a 6-column panel does not fit in LDS
//   ba_point_trial    8 lanes per point: back-substitution x_l = Dinv (bl - sum Hpl^T x_p) with the sum split over the
//                     lanes, trial point, robust chi2 of the trial state, fixed-order block reduction
//   (ba_decide)       inside ba_point_trial, by the workgroup that publishes its partials last: one wave sums the block
//                     partials, one thread: gain ratio, accept (flip the state buffer index) or reject, lambda
//                     update, stop rules
//  motion-only problem (no free points, no scale edges => block diagonal): ba_motion_step, see below.
// Trial states are written to the OTHER of two state buffers, so a rejected step needs no restore.
#include "vs_ba_internal.h"

#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <type_traits>
#include <vector>

using namespace vsba;

namespace {

constexpr int kPtThreads = 256; // threads per point block
#ifndef VS_PT_LANES
#define VS_PT_LANES 8
#endif
constexpr int kPtLanes = VS_PT_LANES;  // lanes that share one point's observations (linearisation and trial)
constexpr int kPtLaneSteps = kPtLanes == 16 ? 4 : 3;
static_assert(kPtLanes == 8 || kPtLanes == 16, "vs_group_reduce over the lanes of a point");
constexpr int kPtPerBlock = kPtThreads / kPtLanes;
constexpr int kCamThreads = 512;  // ba_linearize: a camera's observations are spread over this many threads
constexpr int kSchurThreads = 256;
constexpr int kSolveThreads = 512;
constexpr int kMaxLdsN = 126;   // reduced systems up to 126 x 126 (21 free cameras) are factorised in LDS by one workgroup
constexpr int kMaxLdsPacked = 198;  // ... up to 198 x 198 (33 free cameras) with the lower triangle packed (160 KB of LDS)
constexpr int kMaxSlabN = 90;   // Schur slabs up to 90 x 90 (15 free cameras) live in LDS



// ------------------------------------------------------------------------------------------------ small math



__device__ inline void huber_rho(double delta, double e2, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e2 <= dsqr) {
    rho0 = e2;
    rho1 = 1.0;
  } else {
    const double sqrte = sqrt(e2);
    rho0 = 2 * sqrte * delta - dsqr;
    rho1 = delta / sqrte;
  }
}

__device__ inline void dcs_rho(double phi, double e2, double& rho0, double& rho1) {
  const double scale = (2.0 * phi) / (phi + e2);
  if (scale >= 1.0) {
    rho0 = e2;
    rho1 = 1.0;
  } else {
    rho0 = scale * e2 * scale;
    rho1 = scale * scale;
  }
}

struct edge_t {
  double e[2], W[3], rho0, rho1;
  double Ji[2][3], Jj[2][6];
};

// EdgeProjectP2MC::computeError (+ optionally linearizeOplus) for camera record c (kCamStride doubles) and point X
template <bool JAC>
__device__ inline void eval_edge(const ba_dev& D, const double* __restrict__ c, const double* X, const double* uv,
                                 const double* info, edge_t& E) {
  const double* w = c + 7;
  double pc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = w[4 * i] * X[0] + w[4 * i + 1] * X[1] + w[4 * i + 2] * X[2] + w[4 * i + 3];
  const double u = D.fx * pc[0] + D.cx * pc[2], v = D.fy * pc[1] + D.cy * pc[2], wz = pc[2];
  E.e[0] = u / wz - uv[0];
  E.e[1] = v / wz - uv[1];
  if (info) {
    E.W[0] = info[0];
    E.W[1] = info[1];
    E.W[2] = info[2];
  } else {
    E.W[0] = 1;
    E.W[1] = 0;
    E.W[2] = 1;
  }
  const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
  const double e2 = E.e[0] * We0 + E.e[1] * We1;
  E.rho0 = e2;
  E.rho1 = 1.0;
  if (D.huber > 0) huber_rho(D.huber, e2, E.rho0, E.rho1);
  if (JAC) {
    const double px = pc[0], py = pc[1], pz = pc[2];
    const double ipz2 = 1.0 / (pz * pz);
    const double ipz2fx = ipz2 * D.fx, ipz2fy = ipz2 * D.fy;
    const double p0 = X[0] - c[0], p1 = X[1] - c[1], p2 = X[2] - c[2];
    double r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = w[4 * k] * p0 + w[4 * k + 1] * p1 + w[4 * k + 2] * p2;
    // dRd{x,y,z} = dRid{x,y,z} * R^T has rows {0, +-2 * row of R^T}: dp = dRd * (X - t)
    const double dpx[3] = {0.0, 2 * r[2], -2 * r[1]};
    const double dpy[3] = {-2 * r[2], 0.0, 2 * r[0]};
    const double dpz[3] = {2 * r[1], -2 * r[0], 0.0};
    E.Jj[0][3] = (pz * dpx[0] - px * dpx[2]) * ipz2fx;
    E.Jj[1][3] = (pz * dpx[1] - py * dpx[2]) * ipz2fy;
    E.Jj[0][4] = (pz * dpy[0] - px * dpy[2]) * ipz2fx;
    E.Jj[1][4] = (pz * dpy[1] - py * dpy[2]) * ipz2fy;
    E.Jj[0][5] = (pz * dpz[0] - px * dpz[2]) * ipz2fx;
    E.Jj[1][5] = (pz * dpz[1] - py * dpz[2]) * ipz2fy;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double d0 = w[a], d1 = w[4 + a], d2 = w[8 + a];
      E.Ji[0][a] = (pz * d0 - px * d2) * ipz2fx;
      E.Ji[1][a] = (pz * d1 - py * d2) * ipz2fy;
      E.Jj[0][a] = -E.Ji[0][a];
      E.Jj[1][a] = -E.Ji[1][a];
    }
  }
}

__device__ inline double scale_err(const double* t1, const double* t2, double meas) {
  const double dx = t2[0] - t1[0], dy = t2[1] - t1[1], dz = t2[2] - t1[2];
  return meas - sqrt(dx * dx + dy * dy + dz * dz);
}

// fixed-order workgroup reduction of two doubles per thread (sum, and sum or max): xor-butterfly inside each wave (every
// lane ends with the wave's value), the wave values through LDS, then added in wave order -- one barrier instead of a
// barrier per tree level.  s_red holds 2 * (T / 64) doubles; every thread returns with both results.
template <int T, bool SECOND_IS_MAX>
__device__ inline void block_reduce2(double& a, double& b, double* s_red) {
  constexpr int kW = T / 64;
  a = vs_group_reduce<6>(a);
  b = vs_group_reduce<6, SECOND_IS_MAX>(b);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[wv] = a;
    s_red[kW + wv] = b;
  }
  __syncthreads();
  a = s_red[0];
  b = s_red[kW];
#pragma unroll
  for (int w = 1; w < kW; ++w) {
    a += s_red[w];
    b = SECOND_IS_MAX ? fmax(b, s_red[kW + w]) : b + s_red[kW + w];
  }
}

// sum / max of the per-block partials by one wave: lane l takes entries l, l + 64, ... in order, then an xor-butterfly
// (the loads of eight entries are issued together, the additions keep their order: with thousands of point blocks -- the scaled
// run has 6 250 -- one load per round trip made this the tail of every trial)
__device__ inline double wave_sum_partials(const double* v, int n) {
  double a = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * 64 < n; i += 8 * 64) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = v[i + 64 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) a += x[k];
  }
  for (; i < n; i += 64) a += v[i];
  return vs_group_reduce<6>(a);
}
__device__ inline double wave_max_partials(const double* v, int n) {
  double a = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * 64 < n; i += 8 * 64) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = v[i + 64 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) a = fmax(a, x[k]);
  }
  for (; i < n; i += 64) a = fmax(a, v[i]);
  return vs_group_reduce<6, true>(a);
}

// ------------------------------------------------------------------------------------------------ linearize
// The linearisation (Hpp, bp, Hll, bl, Hpl) of a state.  Single-tile windows keep two of them, indexed like the state
// buffers: ba_point_trial linearises the TRIAL state's points while it evaluates the trial (speculatively -- the values are
// simply overwritten if the step is rejected), so an accepted step needs no linearisation launch of its own.
struct lin_view {
  double *Hpp, *bp, *Hll, *bl, *Hpl;
};
__device__ inline lin_view lin_of(const ba_dev& D, int buf) {
  lin_view L;
  const bool second = D.spec && buf;
  L.Hpp = second ? D.Hpp1 : D.Hpp;
  L.bp = second ? D.bp1 : D.bp;
  L.Hll = second ? D.Hll1 : D.Hll;
  L.bl = second ? D.bl1 : D.bl;
  L.Hpl = second ? D.Hpl1 : D.Hpl;
  return L;
}

// Point role of the linearisation for the kPtLanes lanes that share active point `a` (lane `sub` takes its observations
// sub, sub + kPtLanes, ...): Hll / bl summed over the lanes with xor-shuffles (every lane ends with the same bits), the
// Hpl blocks written per observation.  Returns this lane's share of the robust chi2; maxd = largest |diagonal| of Hll.
__device__ inline double linearize_point(const ba_dev& D, const lin_view& L, const double* cams, const double* X, int a, int ls,
                                         int sub, double& maxd) {
  double chi = 0.0;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int o = D.pt_start[a] + sub; o < D.pt_start[a + 1]; o += kPtLanes) {
    const int ci = D.o_cam[o];
    const int cs = D.pose_slot[ci];
    edge_t E;
    eval_edge<true>(D, cams + (size_t)ci * kCamStride, X, D.o_uv + 2 * (size_t)o, D.has_info ? D.o_info + 3 * (size_t)o : nullptr, E);
    chi += E.rho0;
    const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
    const double r0 = -We0 * E.rho1, r1 = -We1 * E.rho1;
    const double w0 = E.rho1 * E.W[0], w1 = E.rho1 * E.W[1], w2 = E.rho1 * E.W[2];
    double WJi[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      WJi[0][k] = w0 * E.Ji[0][k] + w1 * E.Ji[1][k];
      WJi[1][k] = w1 * E.Ji[0][k] + w2 * E.Ji[1][k];
    }
    if (ls >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b[k] += E.Ji[0][k] * r0 + E.Ji[1][k] * r1;
#pragma unroll
        for (int l = 0; l < 3; ++l) H[3 * k + l] += E.Ji[0][k] * WJi[0][l] + E.Ji[1][k] * WJi[1][l];
      }
      if (cs >= 0) {
        double* B = L.Hpl + 18 * (size_t)D.o_hpl[o];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int l = 0; l < 3; ++l) B[3 * k + l] = E.Jj[0][k] * WJi[0][l] + E.Jj[1][k] * WJi[1][l];
      }
    }
  }
  if (ls >= 0) {  // uniform inside the lane group
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = vs_group_reduce<kPtLaneSteps>(H[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = vs_group_reduce<kPtLaneSteps>(b[k]);
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) L.Hll[9 * (size_t)ls + k] = H[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) L.bl[3 * (size_t)ls + k] = b[k];
      maxd = fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8])));
    }
  }
  return chi;
}

// Camera role: block row c of Hpp and bp from free camera slot c's observations.  Evaluating an edge costs some 400 FP64
// instructions, so one workgroup (= one CU) per camera is compute-bound on that CU: D.cam_split workgroups of kCamThreads
// threads share a camera (workgroup `part` takes the observations part * kCamThreads + tid + j * cam_split * kCamThreads),
// publish their 27 sums, and the one that arrives last adds the parts in part order and assembles the block row.
// publish_only: the workgroup leaves its 27 sums in D.cam_part and is done -- ba_reduce adds the parts itself (windows
// without scale edges whose Hpp is nothing but these diagonal blocks); otherwise the block row of Hpp / bp is assembled here.
__device__ inline void linearize_camera(const ba_dev& D, int c, int part, double (*s_all)[27], double (*s_grp)[27],
                                        bool publish_only = false) {
  const int tid = threadIdx.x, split = D.cam_split, vthreads = split * kCamThreads;
  // four observations per thread and pass.  The chain camera list -> (observation, point) -> position is dependent round
  // trips, so the loads of all four are issued level by level -- and the first two levels of the first pass before the LM
  // state is read (their addresses do not depend on it; caches are cold after every kernel boundary)
  constexpr int kCamUnroll = 4;
  const int i_beg = D.cam_start[c] + part * kCamThreads + tid, i_end = D.cam_start[c + 1];
  int ou[kCamUnroll], pu[kCamUnroll];
  double Xu[kCamUnroll][3], uvu[kCamUnroll][2], infou[kCamUnroll][3];
  auto load_indices = [&](int ib) {
#pragma unroll
    for (int u = 0; u < kCamUnroll; ++u) {
      const bool have = ib + u * vthreads < i_end;
      ou[u] = have ? D.cam_obs[ib + u * vthreads] : -1;
      pu[u] = have ? D.cam_pt[ib + u * vthreads] : 0;
    }
#pragma unroll
    for (int u = 0; u < kCamUnroll; ++u)
      if (ou[u] >= 0) {
        uvu[u][0] = D.o_uv[2 * (size_t)ou[u]];
        uvu[u][1] = D.o_uv[2 * (size_t)ou[u] + 1];
        if (D.has_info) {
#pragma unroll
          for (int k = 0; k < 3; ++k) infou[u][k] = D.o_info[3 * (size_t)ou[u] + k];
        }
      }
  };
  load_indices(i_beg);
  const lm_state st = *D.st;
  if (st.done || !st.need_lin) return;  // uniform
  const lin_view L = lin_of(D, st.cur);
  const double* cams = D.cam[st.cur];
  const double* pts = D.pts[st.cur];
  const int pose = D.slot_pose[c];
  const double* cam = cams + (size_t)pose * kCamStride;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  for (int ib = i_beg; ib < i_end; ib += kCamUnroll * vthreads) {
    if (ib != i_beg) load_indices(ib);
#pragma unroll
    for (int u = 0; u < kCamUnroll; ++u)
      if (ou[u] >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Xu[u][k] = pts[3 * (size_t)pu[u] + k];
      }
#pragma unroll
    for (int u = 0; u < kCamUnroll; ++u) {
      if (ou[u] < 0) continue;
      const double* X = Xu[u];
      edge_t E;
      eval_edge<true>(D, cam, X, uvu[u], D.has_info ? infou[u] : nullptr, E);
      const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
      const double r0 = -We0 * E.rho1, r1 = -We1 * E.rho1;
      const double w0 = E.rho1 * E.W[0], w1 = E.rho1 * E.W[1], w2 = E.rho1 * E.W[2];
      double WJj[2][6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        WJj[0][k] = w0 * E.Jj[0][k] + w1 * E.Jj[1][k];
        WJj[1][k] = w1 * E.Jj[0][k] + w2 * E.Jj[1][k];
      }
      int n = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = k; l < 6; ++l) acc[n++] += E.Jj[0][k] * WJj[0][l] + E.Jj[1][k] * WJj[1][l];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[21 + k] += E.Jj[0][k] * r0 + E.Jj[1][k] * r1;
    }
  }
  // fixed-order reduction of the 27 sums: the upper half of the threads hands its values to the lower half through
  // LDS, then the 256 rows are added by (value, group-of-32) threads in row order and the 8 group sums in group order
  constexpr int kHalf = kCamThreads / 2, kGroups = kHalf / 32;
  if (tid >= kHalf) {
#pragma unroll
    for (int k = 0; k < 27; ++k) s_all[tid - kHalf][k] = acc[k];
  }
  __syncthreads();
  if (tid < kHalf) {
#pragma unroll
    for (int k = 0; k < 27; ++k) s_all[tid][k] += acc[k];
  }
  __syncthreads();
  {
    const int lane = tid & 63, wv = tid >> 6;  // waves 0..kGroups-1: one group each, lanes 0..26 = the 27 values
    if (wv < kGroups && lane < 27) {
      double a2 = 0.0;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) a2 += s_all[wv * 32 + j][lane];
      s_grp[wv][lane] = a2;
    }
  }
  __syncthreads();
  if (tid < 27) {
    double a2 = s_grp[0][tid];
#pragma unroll
    for (int g = 1; g < kGroups; ++g) a2 += s_grp[g][tid];
    s_grp[0][tid] = a2;
  }
  __syncthreads();
  if (publish_only) {
    if (tid < 27) D.cam_part[((size_t)c * split + part) * 27 + tid] = s_grp[0][tid];
    return;
  }
  if (split > 1) {
    // hand-off as in ba_point_trial: write-through stores, drain, one relaxed agent-scope ticket add; the last arriver
    // acquires, resets the ticket for the next linearisation and adds the parts in part order
    __shared__ int s_last;
    double* mine = D.cam_part + ((size_t)c * split + part) * 27;
    if (tid < 27)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + tid), (unsigned long long)__double_as_longlong(s_grp[0][tid]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(D.cam_ticket + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)split - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(D.cam_ticket + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 27) {
      const double* parts = D.cam_part + (size_t)c * split * 27;
      double a2 = parts[tid];
      for (int k = 1; k < split; ++k) a2 += parts[k * 27 + tid];
      s_grp[0][tid] = a2;
    }
    __syncthreads();
  }
  const double* s_acc = s_grp[0];
  // block row c of Hpp: zero, diagonal block, then the EdgeSBAScale terms of every scale edge touching this camera
  const int np = D.np;
  for (int i = tid; i < 6 * np; i += kCamThreads) L.Hpp[(size_t)(6 * c) * np + i] = 0.0;
  __syncthreads();
  if (tid == 0) {
    double* Hrow = L.Hpp + (size_t)(6 * c) * np;
    int n = 0;
    for (int k = 0; k < 6; ++k)
      for (int l = k; l < 6; ++l) {
        Hrow[(size_t)k * np + 6 * c + l] = s_acc[n];
        Hrow[(size_t)l * np + 6 * c + k] = s_acc[n];
        ++n;
      }
    double b[6];
    for (int k = 0; k < 6; ++k) b[k] = s_acc[21 + k];
    for (int e = 0; e < D.n_scale; ++e) {
      const int v0 = D.sc_parent[e], v1 = D.sc_child[e];
      const int s0 = D.pose_slot[v0], s1 = D.pose_slot[v1];
      if (s0 != c && s1 != c) continue;
      const double* t0 = cams + (size_t)v0 * kCamStride;
      const double* t1 = cams + (size_t)v1 * kCamStride;
      const double m = D.sc_meas[e];
      const double err = scale_err(t0, t1, m);
      // numeric Jacobian of BaseBinaryEdge (central differences, delta 1e-9); rotation columns are exactly zero
      double J[2][3] = {{0, 0, 0}, {0, 0, 0}};
      const double delta = 1e-9, scalar = 1.0 / (2 * delta);
      for (int side = 0; side < 2; ++side) {
        if ((side == 0 ? s0 : s1) < 0) continue;
        for (int d = 0; d < 3; ++d) {
          double tp[3], tm[3];
          const double* ts = side == 0 ? t0 : t1;
          for (int k = 0; k < 3; ++k) tp[k] = tm[k] = ts[k];
          tp[d] += delta;
          tm[d] += -delta;
          const double ep = side == 0 ? scale_err(tp, t1, m) : scale_err(t0, tp, m);
          const double em = side == 0 ? scale_err(tm, t1, m) : scale_err(t0, tm, m);
          J[side][d] = scalar * (ep - em);
        }
      }
      double rho0, rho1;
      dcs_rho(D.dcs, err * err, rho0, rho1);
      const double r = -err * rho1, wgt = rho1;
      for (int si = 0; si < 2; ++si) {
        if ((si == 0 ? s0 : s1) != c) continue;  // this workgroup owns block row c only
        for (int k = 0; k < 3; ++k) {
          b[k] += J[si][k] * r;
          for (int sj = 0; sj < 2; ++sj) {
            const int cj = sj == 0 ? s0 : s1;
            if (cj < 0) continue;
            for (int l = 0; l < 3; ++l) Hrow[(size_t)k * np + 6 * cj + l] += J[si][k] * wgt * J[sj][l];
          }
        }
      }
    }
    for (int k = 0; k < 6; ++k) L.bp[6 * c + k] = b[k];
    // largest |diagonal| of the block row (lambda_0); with scale edges the diagonal has just been updated in memory
    double mx = 0.0;
    n = 0;
    for (int k = 0; k < 6; ++k) {
      mx = fmax(mx, fabs(D.n_scale ? Hrow[(size_t)k * np + 6 * c + k] : s_acc[n]));
      n += 6 - k;
    }
    D.part_maxd[D.nb_pt + c] = mx;
  }
}

__global__ __launch_bounds__(kCamThreads) void ba_linearize(ba_dev D) {
  __shared__ double s_red[2 * kCamThreads / 64];
  __shared__ double s_all[kCamThreads / 2][27];  // camera role: upper half of the first level lives in registers
  __shared__ double s_grp[kCamThreads / 64][27];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= D.nb_pt) {
    const int cw = blockIdx.x - D.nb_pt;
    linearize_camera(D, cw / D.cam_split, cw % D.cam_split, s_all, s_grp);
    return;
  }
  // ---- point role: the first kPtThreads threads of the workgroup, kPtLanes lanes per point
  const int a = blockIdx.x * kPtPerBlock + tid / kPtLanes, sub = tid % kPtLanes;
  const bool mine = tid < kPtThreads && a < D.n_act;
  const int p = mine ? D.act_pt[a] : 0;  // requested before the LM state: the address does not depend on it
  const lm_state st = *D.st;
  if (st.done || !st.need_lin) return;
  const double* cams = D.cam[st.cur];
  const double* pts = D.pts[st.cur];
  const lin_view L = lin_of(D, st.cur);
  double chi = 0.0, maxd = 0.0;
  if (mine) {
    const double X[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
    chi = linearize_point(D, L, cams, X, a, D.pt_slot[p], sub, maxd);
  }
  double csum = chi, cmax = maxd;
  block_reduce2<kCamThreads, true>(csum, cmax, s_red);
  if (tid == 0) {
    D.part_chi[blockIdx.x] = csum;
    D.part_maxd[blockIdx.x] = cmax;
  }
}

// The two roles as launches of their own, for large problems.  In the combined kernel a point block is a 512-thread workgroup
// of which 256 threads work, with the camera role's register count (204): one workgroup per CU, ONE working wave per SIMD -- and
// a lone wave issues an instruction only every ~9 cycles (tools/f64_probe.hip).  At 6 250 point blocks (the scaled run) that
// was 24 rounds of 11 us; as 256-thread workgroups four of them share a CU.  (Small problems keep the single launch: there
// the launch boundary is what counts.)
__global__ __launch_bounds__(kPtThreads, 3) void ba_linearize_points(ba_dev D) {
  __shared__ double s_red[2 * kPtThreads / 64];
  const int tid = threadIdx.x;
  const int a = blockIdx.x * kPtPerBlock + tid / kPtLanes, sub = tid % kPtLanes;
  const bool mine = a < D.n_act;
  const int p = mine ? D.act_pt[a] : 0;
  const lm_state st = *D.st;
  if (st.done || !st.need_lin) return;
  const double* cams = D.cam[st.cur];
  const double* pts = D.pts[st.cur];
  const lin_view L = lin_of(D, st.cur);
  double chi = 0.0, maxd = 0.0;
  if (mine) {
    const double X[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
    chi = linearize_point(D, L, cams, X, a, D.pt_slot[p], sub, maxd);
  }
  double csum = chi, cmax = maxd;
  block_reduce2<kPtThreads, true>(csum, cmax, s_red);  // (four wave values instead of eight: added in wave order either way)
  if (tid == 0) {
    D.part_chi[blockIdx.x] = csum;
    D.part_maxd[blockIdx.x] = cmax;
  }
}
__global__ __launch_bounds__(kCamThreads) void ba_linearize_cameras(ba_dev D) {
  __shared__ double s_all[kCamThreads / 2][27];
  __shared__ double s_grp[kCamThreads / 64][27];
  linearize_camera(D, blockIdx.x / D.cam_split, blockIdx.x % D.cam_split, s_all, s_grp);
}

// chi2 of the scale edges for state buffer `buf` (few edges: one thread)
__device__ inline double scale_edges_chi(const ba_dev& D, const double* cams) {
  double chi = 0.0;
  for (int e = 0; e < D.n_scale; ++e) {
    const int v0 = D.sc_parent[e], v1 = D.sc_child[e];
    if (D.pose_slot[v0] < 0 && D.pose_slot[v1] < 0) continue;
    const double err = scale_err(cams + (size_t)v0 * kCamStride, cams + (size_t)v1 * kCamStride, D.sc_meas[e]);
    double rho0, rho1;
    dcs_rho(D.dcs, err * err, rho0, rho1);
    chi += rho0;
  }
  return chi;
}

__global__ __launch_bounds__(64) void ba_lambda_init(ba_dev D) {
  lm_state* st = D.st;
  if (blockIdx.x != 0) return;
  double chi = wave_sum_partials(D.part_chi, D.nb_pt);
  const double mx = wave_max_partials(D.part_maxd, D.nb_pt + D.nfp);
  if (threadIdx.x != 0) return;
  chi += scale_edges_chi(D, D.cam[st->cur]);
  st->current_chi = chi;
  st->chi0 = chi;
  st->lambda = 1e-5 * mx;
  st->ni = 2.0;
}

// ------------------------------------------------------------------------------------------------ Schur complement
__device__ inline void inv3(const double* Dm, double* inv) {
  const double c00 = Dm[4] * Dm[8] - Dm[5] * Dm[7], c01 = Dm[5] * Dm[6] - Dm[3] * Dm[8],
               c02 = Dm[3] * Dm[7] - Dm[4] * Dm[6];
  const double det = Dm[0] * c00 + Dm[1] * c01 + Dm[2] * c02;
  const double id = 1.0 / det;
  inv[0] = c00 * id;
  inv[1] = (Dm[2] * Dm[7] - Dm[1] * Dm[8]) * id;
  inv[2] = (Dm[1] * Dm[5] - Dm[2] * Dm[4]) * id;
  inv[3] = c01 * id;
  inv[4] = (Dm[0] * Dm[8] - Dm[2] * Dm[6]) * id;
  inv[5] = (Dm[2] * Dm[3] - Dm[0] * Dm[5]) * id;
  inv[6] = c02 * id;
  inv[7] = (Dm[1] * Dm[6] - Dm[0] * Dm[7]) * id;
  inv[8] = (Dm[0] * Dm[4] - Dm[1] * Dm[3]) * id;
}

// dynamic LDS: [slab np*np + np doubles when LDS_SLAB] [Dinv+db: per*12] [Y mmax*18] [B mmax*18] [slots mmax ints]
// The Hpl blocks of a free point are contiguous (fp_start), so staging is one coalesced read without index chasing;
// (Hll + lambda I)^-1 of all the slab's points is computed up front by one thread per point.
template <bool LDS_SLAB>
__global__ __launch_bounds__(kSchurThreads) void ba_schur(ba_dev D) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  const lm_state st = *D.st;
  if (st.done) return;
  const int np = D.np, tid = threadIdx.x;
  const int slab_elems = np * np + np;
  const int per = (D.nfl + D.ns - 1) / D.ns;
  double* slab = LDS_SLAB ? s_mem : D.slab + (size_t)blockIdx.x * slab_elems;
  double* sDi = s_mem + (LDS_SLAB ? slab_elems : 0);  // [per][12]: Dinv (9) + Dinv*bl (3)
  double* sY = sDi + (size_t)per * 12;
  double* sB = sY + D.mmax * 18;
  int* sSlot = reinterpret_cast<int*>(sB + D.mmax * 18);
  int* sRank = sSlot + D.mmax;  // duplicate observations only
  const int maxr = D.max_rank;
  for (int i = tid; i < slab_elems; i += kSchurThreads) slab[i] = 0.0;
  const double lambda = st.lambda;
  const int l0 = blockIdx.x * per, l1 = min(l0 + per, D.nfl);
  for (int l = l0 + tid; l < l1; l += kSchurThreads) {
    double Dm[9], inv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Dm[k] = D.Hll[9 * (size_t)l + k];
    Dm[0] += lambda;
    Dm[4] += lambda;
    Dm[8] += lambda;
    inv3(Dm, inv);
    const double b0 = D.bl[3 * (size_t)l], b1 = D.bl[3 * (size_t)l + 1], b2 = D.bl[3 * (size_t)l + 2];
    double* d = sDi + (size_t)(l - l0) * 12;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      d[k] = inv[k];
      D.Dinv[9 * (size_t)l + k] = inv[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) d[9 + k] = inv[3 * k] * b0 + inv[3 * k + 1] * b1 + inv[3 * k + 2] * b2;
  }
  __syncthreads();
  for (int l = l0; l < l1; ++l) {
    const int base = D.fp_start[l], m = D.fp_start[l + 1] - base;
    const double* sD = sDi + (size_t)(l - l0) * 12;
    for (int i = tid; i < m * 18; i += kSchurThreads) {
      const int obs = i / 18, k = i - obs * 18;
      const int arow = k / 3, bcol = k - arow * 3;
      const double* B = D.Hpl + 18 * (size_t)(base + obs);
      sB[i] = B[k];
      sY[i] = B[3 * arow] * sD[bcol] + B[3 * arow + 1] * sD[3 + bcol] + B[3 * arow + 2] * sD[6 + bcol];
    }
    for (int i = tid; i < m; i += kSchurThreads) {
      sSlot[i] = D.fp_slot[base + i];
      sRank[i] = maxr ? D.fp_rank[base + i] : 0;
    }
    __syncthreads();
    // Distinct cameras -> distinct slab elements.  A camera that observes the point more than once (never produced by
    // the reference's dict-of-frames, but legal at the ABI) would make several (obs, obs) pairs hit one element: those
    // are processed in rounds by repeat rank -- inside a round every element has one writer, and the rounds run in the
    // sequential algorithm's order (i ascending, then j), so the sums are deterministic and ordered.  No atomics.
    // rhs: slab_b[ci] += B_i * (Dinv bl)   (one thread per (obs, row))
    for (int ri = 0; ri <= maxr; ++ri) {
      for (int i = tid; i < m * 6; i += kSchurThreads) {
        const int obs = i / 6, arow = i - obs * 6;
        if (sRank[obs] != ri) continue;
        const double* B = sB + obs * 18 + 3 * arow;
        const double v = B[0] * sD[9] + B[1] * sD[10] + B[2] * sD[11];
        slab[np * np + 6 * sSlot[obs] + arow] += v;
      }
      if (maxr) __syncthreads();
    }
    // matrix: slab[ci][cj] += Y_i * B_j^T
    const int nel = m * m * 36;
    for (int ri = 0; ri <= maxr; ++ri)
      for (int rj = 0; rj <= maxr; ++rj) {
        for (int i = tid; i < nel; i += kSchurThreads) {
          const int pair = i / 36, k = i - pair * 36;
          const int oi = pair / m, oj = pair - oi * m;
          if (sRank[oi] != ri || sRank[oj] != rj) continue;
          const int arow = k / 6, bcol = k - arow * 6;
          const double* Y = sY + oi * 18 + 3 * arow;
          const double* B = sB + oj * 18 + 3 * bcol;
          const double v = Y[0] * B[0] + Y[1] * B[1] + Y[2] * B[2];
          slab[(size_t)(6 * sSlot[oi] + arow) * np + 6 * sSlot[oj] + bcol] += v;
        }
        if (maxr) __syncthreads();
      }
    __syncthreads();
  }
  if (LDS_SLAB) {
    double* out = D.slab + (size_t)blockIdx.x * slab_elems;
    for (int i = tid; i < slab_elems; i += kSchurThreads) out[i] = slab[i];
  }
}

// Tiled variant for windows beyond 15 free cameras (the slab no longer fits in LDS).  The slab is cut into tiles of
// kTileCams x kTileCams camera blocks; workgroup (s, tr, tc) walks slab s's points, keeps those that are seen from both
// camera ranges (one 64-bit mask per point, written by the host) and accumulates its tile IN REGISTERS: wave w owns the
// local row cameras w, w+4, w+8 and lane (block row a, column camera c) owns the 6 elements [a][0..5] of block (row, c)
// for each of them - fixed ownership, so no atomics, no LDS read-modify-write and no barrier between points; per
// element the points arrive in the same order as in the kernel above, so the sums are identical.  Hpl blocks are staged
// for kTileBatch points per barrier, all loads issued before the first use, and the camera maps of batch b+1 are
// scattered while batch b is staged (three map buffers: read / being filled / being cleared).
constexpr int kTileCams = 10, kTileN = 6 * kTileCams, kTileBatch = 8, kTileChunk = 512, kTileRows = (kTileCams + 3) / 4;

__global__ __launch_bounds__(256) void ba_dinv(ba_dev D) {
  const lm_state st = *D.st;
  if (st.done) return;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= D.nfl) return;
  const lin_view L = lin_of(D, st.cur);
  double Dm[9], inv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Dm[k] = L.Hll[9 * (size_t)l + k];
  Dm[0] += st.lambda;
  Dm[4] += st.lambda;
  Dm[8] += st.lambda;
  inv3(Dm, inv);
  const double b0 = L.bl[3 * (size_t)l], b1 = L.bl[3 * (size_t)l + 1], b2 = L.bl[3 * (size_t)l + 2];
#pragma unroll
  for (int k = 0; k < 9; ++k) D.Dinv[9 * (size_t)l + k] = inv[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) D.Dbl[3 * (size_t)l + k] = inv[3 * k] * b0 + inv[3 * k + 1] * b1 + inv[3 * k + 2] * b2;
}

__global__ __launch_bounds__(256) void ba_schur_tile(ba_dev D) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  if (D.st->done) return;
  const double lambda = D.st->lambda;
  const int np = D.np, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // blockIdx.y enumerates the tiles of the LOWER triangle (tr >= tc), row by row: the Cholesky reads nothing else
  int tr = 0;
  while ((tr + 1) * (tr + 2) / 2 <= (int)blockIdx.y) ++tr;
  const int tc = (int)blockIdx.y - tr * (tr + 1) / 2;
  const int per = (D.nfl + D.ns - 1) / D.ns;
  const int l0 = blockIdx.x * per, l1 = min(l0 + per, D.nfl);
  double* sD = s_mem;                                 // [2][batch][12]   Dinv, Dinv bl (double-buffered)
  double* sY = sD + 2 * kTileBatch * 12;              // [batch][cams][18]
  double* sB = sY + kTileBatch * kTileCams * 18;      // [batch][cams][18]
  int* sMap = reinterpret_cast<int*>(sB + kTileBatch * kTileCams * 18);  // [3][batch][2][16]: Hpl block of a local camera, -1
  int* sList = sMap + 3 * kTileBatch * 32;            // [kTileChunk][3]: point, first Hpl block, blocks
  int* sCnt = sList + 3 * kTileChunk;                 // [8]: wave counts
  const int arow = lane / kTileCams, ccam = lane - arow * kTileCams;  // lanes 60..63 (arow == 6) idle in the products
  double acc[kTileRows][6], racc[kTileRows];
#pragma unroll
  for (int r = 0; r < kTileRows; ++r) {
    racc[r] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[r][c] = 0.0;
  }
  const unsigned long long want = (1ull << tr) | (1ull << tc);
  constexpr int kPerThread = kTileChunk / 256, kScat = 256 / kTileBatch;
  auto scatter = [&](int b0, int nb, int mbuf, int dbuf) {
    const int pb = tid / kScat, i0 = tid - pb * kScat;  // kScat threads per point of the batch
    if (pb < nb) {
      const int base = sList[3 * (b0 + pb) + 1], m = sList[3 * (b0 + pb) + 2];
      for (int i = i0; i < m; i += kScat) {
        const int slot = D.fp_slot[base + i];
        const int t = slot / kTileCams, ls = slot - t * kTileCams;
        if (t == tr) sMap[(mbuf * kTileBatch + pb) * 32 + ls] = base + i;
        if (t == tc) sMap[(mbuf * kTileBatch + pb) * 32 + 16 + ls] = base + i;
      }
    }
    // (Hll + lambda I)^-1 and its product with bl of the batch's points: the last lane of every point's group
    // (single-tile windows: nobody else computes it, so it is also stored for ba_point_trial; with several tiles
    // ba_dinv has run before and the values are only re-read)
    if (pb < nb && i0 == kScat - 1) {
      const int l = sList[3 * (b0 + pb)];
      double* d = sD + (dbuf * kTileBatch + pb) * 12;
      if (D.ntile == 1) {
        double Dm[9], inv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Dm[k] = D.Hll[9 * (size_t)l + k];
        Dm[0] += lambda;
        Dm[4] += lambda;
        Dm[8] += lambda;
        inv3(Dm, inv);
        const double b0_ = D.bl[3 * (size_t)l], b1_ = D.bl[3 * (size_t)l + 1], b2_ = D.bl[3 * (size_t)l + 2];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          d[k] = inv[k];
          D.Dinv[9 * (size_t)l + k] = inv[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) d[9 + k] = inv[3 * k] * b0_ + inv[3 * k + 1] * b1_ + inv[3 * k + 2] * b2_;
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = D.Dinv[9 * (size_t)l + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) d[9 + k] = D.Dbl[3 * (size_t)l + k];
      }
    }
  };
  for (int c0 = l0; c0 < l1; c0 += kTileChunk) {
    // ---- ordered list of this chunk's points that touch both camera ranges (thread = kPerThread consecutive points)
    const int cn = min(kTileChunk, l1 - c0);
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int i = tid * kPerThread + k;
      // single tile: every free point is listed, also one seen by fixed cameras only -- its Dinv is computed here
      if (i < cn && (D.ntile == 1 || (D.fp_mask[c0 + i] & want) == want)) keep |= 1u << k;
    }
    const int cnt = __popc(keep);
    int scan = cnt;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(scan, d);
      if (lane >= d) scan += v;
    }
    __syncthreads();  // the previous chunk's readers of sList / sCnt / sMap are done
    if (lane == 63) sCnt[wv] = scan;
    for (int i = tid; i < 3 * kTileBatch * 32; i += 256) sMap[i] = -1;
    __syncthreads();
    int off = scan - cnt;
    for (int w = 0; w < wv; ++w) off += sCnt[w];
    const int total = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];
    for (int k = 0; k < kPerThread; ++k)
      if (keep >> k & 1) {
        const int l = c0 + tid * kPerThread + k;
        const int base = D.fp_start[l];
        sList[3 * off] = l;
        sList[3 * off + 1] = base;
        sList[3 * off + 2] = D.fp_start[l + 1] - base;
        ++off;
      }
    __syncthreads();
    if (total == 0) continue;  // uniform
    scatter(0, min(kTileBatch, total), 0, 0);
    __syncthreads();
    int bi = 0;
    for (int b0 = 0; b0 < total; b0 += kTileBatch, ++bi) {
      const int nb = min(kTileBatch, total - b0);
      const int mb = bi % 3, db = bi & 1;
      // P1: stage batch b from its maps; scatter batch b+1
      {
        // item = (point, side, local camera, block row): 3 contiguous values of the Hpl block
        constexpr int kItems = kTileBatch * 2 * kTileCams * 6, kRounds = (kItems + 255) / 256;
        double v[kRounds][3];
        int dst[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
          const int i = tid + 256 * r;
          const int ar = i % 6, q = i / 6;
          const int ls = q % kTileCams, side = (q / kTileCams) & 1, pb = q / (2 * kTileCams);
          dst[r] = -1;
          if (pb < nb) {
            const int blk = sMap[(mb * kTileBatch + pb) * 32 + side * 16 + ls];
            if (blk >= 0) {
              const double* B = D.Hpl + 18 * (size_t)blk + 3 * ar;
              v[r][0] = B[0];
              v[r][1] = B[1];
              v[r][2] = B[2];
              dst[r] = ((pb * kTileCams + ls) * 18 + 3 * ar) * 2 + side;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
          if (dst[r] < 0) continue;
          const int o = dst[r] >> 1;
          if (dst[r] & 1) {
            sB[o] = v[r][0];
            sB[o + 1] = v[r][1];
            sB[o + 2] = v[r][2];
          } else {
            const double* sd = sD + (db * kTileBatch + o / (kTileCams * 18)) * 12;
#pragma unroll
            for (int c = 0; c < 3; ++c) sY[o + c] = v[r][0] * sd[c] + v[r][1] * sd[3 + c] + v[r][2] * sd[6 + c];
          }
        }
      }
      if (b0 + kTileBatch < total) scatter(b0 + kTileBatch, min(kTileBatch, total - b0 - kTileBatch), (bi + 1) % 3, db ^ 1);
      __syncthreads();
      // P2: products of batch b into the register accumulators; clear the map buffer of batch b+2
      for (int i = tid; i < kTileBatch * 32; i += 256) sMap[((bi + 2) % 3) * kTileBatch * 32 + i] = -1;
      for (int pb = 0; pb < nb; ++pb) {
        const int* map = sMap + (mb * kTileBatch + pb) * 32;
        bool any = false;
#pragma unroll
        for (int r = 0; r < kTileRows; ++r) any |= wv + 4 * r < kTileCams && map[wv + 4 * r] >= 0;
        if (!any) continue;  // wave-uniform
        const bool have = arow < 6 && map[16 + ccam] >= 0;
        double Bv[18];
        if (have) {
          const double* B = sB + (pb * kTileCams + ccam) * 18;
#pragma unroll
          for (int k = 0; k < 18; ++k) Bv[k] = B[k];
        }
        const double* sd = sD + (db * kTileBatch + pb) * 12;
#pragma unroll
        for (int r = 0; r < kTileRows; ++r) {
          const int ls = wv + 4 * r;
          if (ls >= kTileCams || map[ls] < 0) continue;  // wave-uniform
          if (have) {
            const double* Y = sY + (pb * kTileCams + ls) * 18 + 3 * arow;
            const double y0 = Y[0], y1 = Y[1], y2 = Y[2];
#pragma unroll
            for (int bc = 0; bc < 6; ++bc) acc[r][bc] += y0 * Bv[3 * bc] + y1 * Bv[3 * bc + 1] + y2 * Bv[3 * bc + 2];
          }
          if (tr == tc && arow < 6 && ccam == 0) {
            const double* B = sB + (pb * kTileCams + ls) * 18 + 3 * arow;
            racc[r] += B[0] * sd[9] + B[1] * sd[10] + B[2] * sd[11];
          }
        }
      }
      __syncthreads();
    }
  }
  double* out = D.slab + (size_t)blockIdx.x * ((size_t)np * np + np);
  if (arow < 6) {
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const int ls = wv + 4 * r;
      if (ls >= kTileCams) continue;
      const int row = tr * kTileN + 6 * ls + arow, col = tc * kTileN + 6 * ccam;
      if (row < np && col < np) {
#pragma unroll
        for (int bc = 0; bc < 6; ++bc) out[(size_t)row * np + col + bc] = acc[r][bc];
      }
      if (tr == tc && ccam == 0 && row < np) out[(size_t)np * np + row] = racc[r];
    }
  }
}

// Banded windows beyond one tile (a sliding window of key frames: every point is seen from a few NEIGHBOURING cameras).
// The host orders the contributing free points by their lowest camera slot and cuts that order into slabs; when the
// cameras of every slab span at most kWinCams slots, the slab's whole contribution is a dense kWinN x kWinN block of S
// that starts at camera w0 - and the sum over its points is one matrix product
//     S_win -= [Y_1 Y_2 ...] [H_1 H_2 ...]^T,   Y_l = H_l Dinv_l  (6 wlen x 3 per point, zero rows for absent cameras)
// with K = 3 per point.  That is what the FP64 matrix cores are for: v_mfma_f64_16x16x4_f64 over the lower 16 x 16
// tiles of the window (21 for 16 cameras, dealt round-robin to the four waves, 8 accumulator VGPRs per tile), four k per
// instruction = 4/3 points.  A batch of kWinBatch points is staged as two zero-padded K x kWinN panels in LDS (row stride
// kWinStride: the four k rows an operand read touches fall on disjoint banks); VALU work is only the staging (Y = H Dinv,
// nine FMAs per camera row) and the right-hand side.  Per element the points arrive in the sorted order and the slabs are
// summed in slab order by ba_reduce_window: deterministic, no atomics.  The window slab is 74 KB instead of np^2 doubles.
constexpr int kWinCams = 16, kWinN = 6 * kWinCams, kWinBatch = 8, kWinK = 3 * kWinBatch, kWinStride = kWinN + 16;
constexpr int kWinSlabElems = kWinN * kWinN + kWinN, kWinPerMax = 512, kWinTilesPerWave = 6;
typedef double win_d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void ba_schur_window(ba_dev D) {
  __shared__ __attribute__((aligned(16))) double sY[kWinK * kWinStride];
  __shared__ __attribute__((aligned(16))) double sH[kWinK * kWinStride];
  __shared__ double sD[kWinBatch][12];
  __shared__ int sRec[kWinPerMax][3];  // point slot, first Hpl block, blocks
  __shared__ unsigned sMask[kWinBatch];  // window cameras that see the point
  if (D.st->done) return;
  const double* const Hpl = lin_of(D, D.st->cur).Hpl;  // (large problems keep two linearisations, too: see `spec` in vs_ba_solve)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the tile tests below are branches, not masks
  const int s = blockIdx.x, i0 = s * D.win_per, n = min(D.win_per, D.win_n - i0);
  const int w0 = D.win_w0[s], wlen = D.win_len[s], ncol = 6 * wlen, ntr = (ncol + 15) >> 4;
  const int ntiles = ntr * (ntr + 1) / 2;
  for (int i = tid; i < n; i += 256) {
    const int l = D.win_order[i0 + i], b = D.fp_start[l];
    sRec[i][0] = l;
    sRec[i][1] = b;
    sRec[i][2] = D.fp_start[l + 1] - b;
  }
  for (int i = tid; i < kWinK * kWinStride; i += 256) sY[i] = sH[i] = 0.0;  // the padding columns stay zero
  if (tid < kWinBatch) sMask[tid] = 0u;
  // this wave's tiles: t = wv, wv + 4, ... of the lower triangle, enumerated row by row
  int toff_a[kWinTilesPerWave], toff_b[kWinTilesPerWave];
  win_d4 acc[kWinTilesPerWave];
#pragma unroll
  for (int q = 0; q < kWinTilesPerWave; ++q) {
    const int t = 4 * q + wv;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    toff_a[q] = __builtin_amdgcn_readfirstlane(16 * ti);
    toff_b[q] = __builtin_amdgcn_readfirstlane(16 * tj);
    acc[q] = win_d4{0.0, 0.0, 0.0, 0.0};
  }
  const int nq = __builtin_amdgcn_readfirstlane(wv < ntiles ? (ntiles - 1 - wv) / 4 + 1 : 0);  // this wave's tiles: q < nq
  double racc = 0.0;
  const int krow = lane >> 4, kcol = lane & 15;
  // Staging tasks (point of the batch, Hpl block of the point, block row): kWinBatch * kWinCams * 6 = 3 per thread.  Their
  // global loads -- camera slot and the three values of the row -- are issued one batch ahead and land while the matrix
  // cores work on the current one; the same goes for Dinv and Dinv bl (thread = one of the twelve values of a point).
  constexpr int kTasks = kWinBatch * kWinCams * 6 / 256;
  static_assert(kTasks * 256 == kWinBatch * kWinCams * 6, "three staging tasks per thread");
  int pslot[kTasks];     // camera slot of the task's block: the raw loaded value -- arithmetic on it here would wait for the load
  unsigned pvalid = 0u;  // bit k: task k has a block
  double ph[kTasks][3];  // the row of the Hpl block
  double pd = 0.0;
  __syncthreads();
  auto prefetch = [&](int b0) {
    const int nb = min(kWinBatch, n - b0);
#pragma unroll
    for (int k = 0; k < kTasks; ++k) {
      const int t = tid + 256 * k;
      const int pb = t / (kWinCams * 6), rem = t - pb * (kWinCams * 6), i = rem / 6, ar = rem - 6 * i;
      pvalid &= ~(1u << k);
      if (pb < nb && i < sRec[b0 + pb][2]) {
        const int blk = sRec[b0 + pb][1] + i;
        const double* B = Hpl + 18 * (size_t)blk + 3 * ar;
        pvalid |= 1u << k;
        pslot[k] = D.fp_slot[blk];
        ph[k][0] = B[0];
        ph[k][1] = B[1];
        ph[k][2] = B[2];
      }
    }
    if (tid < kWinBatch * 12) {
      const int pb = tid / 12, k = tid - 12 * pb;
      if (pb < nb) {
        const int l = sRec[b0 + pb][0];
        pd = k < 9 ? D.Dinv[9 * (size_t)l + k] : D.Dbl[3 * (size_t)l + k - 9];
      }
    }
  };
  prefetch(0);
#ifdef VS_WIN_STAMPS
  long long tsA = 0, tsB = 0, tsP = 0, tsC = 0, ts0 = __builtin_readcyclecounter(), tsStart = ts0;
#define VS_WIN_LAP(x) { const long long t_ = __builtin_readcyclecounter(); x += t_ - ts0; ts0 = t_; }
#else
#define VS_WIN_LAP(x)
#endif
  for (int b0 = 0; b0 < n; b0 += kWinBatch) {
    const int nb = min(kWinBatch, n - b0);
    // A: which window cameras see each point; Dinv, Dinv bl
#pragma unroll
    for (int k = 0; k < kTasks; ++k) {
      const int t = tid + 256 * k;
      if ((pvalid >> k & 1u) && t % 6 == 0) atomicOr(&sMask[t / (kWinCams * 6)], 1u << (pslot[k] - w0));
    }
    if (tid < kWinBatch * 12) sD[tid / 12][tid % 12] = pd;
    __syncthreads();
    VS_WIN_LAP(tsA)
    // B: the panels.  Present (point, camera) rows from the prefetched values (three k of H and of Y = H Dinv), absent
    // ones -- and the points beyond the end of the slab -- zeroed.
#pragma unroll
    for (int k = 0; k < kTasks; ++k) {
      const int t = tid + 256 * k;
      const int pb = t / (kWinCams * 6);
      if (pvalid >> k & 1u) {
        const double* sd = sD[pb];
        const double h0 = ph[k][0], h1 = ph[k][1], h2 = ph[k][2];
        const int o = 3 * pb * kWinStride + 6 * (pslot[k] - w0) + t % 6;
        sH[o] = h0;
        sH[o + kWinStride] = h1;
        sH[o + 2 * kWinStride] = h2;
        sY[o] = h0 * sd[0] + h1 * sd[3] + h2 * sd[6];
        sY[o + kWinStride] = h0 * sd[1] + h1 * sd[4] + h2 * sd[7];
        sY[o + 2 * kWinStride] = h0 * sd[2] + h1 * sd[5] + h2 * sd[8];
      }
    }
    for (int t = tid; t < kWinBatch * ncol; t += 256) {
      const int pb = t / ncol, col = t - pb * ncol;
      if (pb < nb && (sMask[pb] >> (col / 6) & 1u)) continue;
      const int o = 3 * pb * kWinStride + col;
      sH[o] = sH[o + kWinStride] = sH[o + 2 * kWinStride] = 0.0;
      sY[o] = sY[o + kWinStride] = sY[o + 2 * kWinStride] = 0.0;
    }
    VS_WIN_LAP(tsB)
    if (b0 + kWinBatch < n) prefetch(b0 + kWinBatch);
    __syncthreads();
    VS_WIN_LAP(tsP)
    // C: matrix cores; the right-hand side on the first threads; the masks cleared for the next batch
    const int nk = (3 * nb + 3) >> 2;
    // (one copy of the loop per tile count: with the count tested inside, the compiler issues every operand read right in
    // front of its own MFMA and waits for it there)
    auto mma = [&](auto nq_) {
      constexpr int NQ = decltype(nq_)::value;
      for (int ks = 0; ks < nk; ++ks) {
        const int o = (4 * ks + krow) * kWinStride + kcol;
        double av[NQ], bv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          av[q] = sY[o + toff_a[q]];
          bv[q] = sH[o + toff_b[q]];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc[q], 0, 0, 0);
      }
    };
    switch (nq) {
      case 1: mma(std::integral_constant<int, 1>{}); break;
      case 2: mma(std::integral_constant<int, 2>{}); break;
      case 3: mma(std::integral_constant<int, 3>{}); break;
      case 4: mma(std::integral_constant<int, 4>{}); break;
      case 5: mma(std::integral_constant<int, 5>{}); break;
      case 6: mma(std::integral_constant<int, 6>{}); break;
      default: break;
    }
    if (tid < ncol) {
      for (int pb = 0; pb < nb; ++pb) {
        const int o = 3 * pb * kWinStride + tid;
        racc += sH[o] * sD[pb][9] + sH[o + kWinStride] * sD[pb][10] + sH[o + 2 * kWinStride] * sD[pb][11];
      }
    } else if (tid >= 128 && tid < 128 + kWinBatch) {
      sMask[tid - 128] = 0u;
    }
    __syncthreads();
    VS_WIN_LAP(tsC)
  }
#ifdef VS_WIN_STAMPS
  if (tid == 0 && (s == 0 || s == 100 || s == 400))
    printf("slab %d: %d points, window %d; cycles A(wait loads + masks) %lld, B(panels) %lld, prefetch issue + barrier %lld, C(mfma) %lld, total %lld\n", s, n, wlen,
           tsA, tsB, tsP, tsC, (long long)__builtin_readcyclecounter() - tsStart);
#endif
  double* out = D.slab + (size_t)s * kWinSlabElems;
#pragma unroll
  for (int q = 0; q < kWinTilesPerWave; ++q) {
    if (q >= nq) break;
#pragma unroll
    for (int v = 0; v < 4; ++v) out[(size_t)(toff_a[q] + krow + 4 * v) * kWinN + toff_b[q] + kcol] = acc[q][v];
  }
  if (tid < ncol) out[kWinN * kWinN + tid] = racc;
}

// Windows of at most kTileCams free cameras (one tile; BASELINE cfg4, the driver's local BA): the tile kernel's camera
// maps, masks and chunk lists are not needed, and its chain of dependent global loads (mask -> list -> slots -> blocks)
// is what a workgroup with 8 points spends its time on.  Here the Hpl blocks of a point are read where they lie
// (contiguous from fp_start[l]), so a batch costs two round trips: [LM state, block ranges, Hll / bl] and then
// [camera slots, Hpl blocks]; (Hll + lambda I)^-1 is inverted while the second is in flight.  Accumulator ownership, the
// product expressions and the order in which a workgroup's points arrive are those of ba_schur_tile, so the sums are
// bit-identical; only the blocks of the lower block triangle are produced (the Cholesky reads nothing else).
constexpr int kSmallPts = 8, kSmallThreads = kCamThreads;
// the product phase is bound by LDS reads (every wave re-reads the B blocks): four waves with three row cameras each move
// 44 % less than eight with two; the other four waves only help with loading and staging
constexpr int kProdWaves = 4, kSmallRows = (kTileCams + kProdWaves - 1) / kProdWaves;

// The last ns workgroups produce the slabs; with two linearisations (D.spec) the first nfp * cam_split workgroups run the
// camera role of the linearisation of a freshly accepted state next to them (its point role ran inside ba_point_trial), so that
// ba_reduce finds Hpp / bp without a linearisation launch in between.  lin_cameras = 0 in the first slot of a solve,
// whose state ba_linearize has linearised completely.
__global__ __launch_bounds__(kSmallThreads) void ba_schur_small(ba_dev D, int lin_cameras) {
  constexpr int kStage = kSmallPts * kTileCams * 18;
  constexpr int kSchurDoubles = kSmallPts * 12 + 2 * kStage;
  constexpr int kCamDoubles = (kCamThreads / 2) * 27 + (kCamThreads / 64) * 27;
  __shared__ double s_raw[kSchurDoubles > kCamDoubles ? kSchurDoubles : kCamDoubles];
  // The camera workgroups come FIRST in the grid.  ns + ncam exceeds the 256 CUs by a few dozen workgroups; the
  // dispatcher hands them out in order, so the surplus (the last Schur workgroups) shares a CU with a camera workgroup --
  // short and light on LDS -- instead of with another Schur workgroup, whose LDS-bound product phase would then take
  // twice as long (measured: 15.0 us per launch with the cameras last, i.e. two Schur workgroups on 30 CUs).
  const int ncam = (int)gridDim.x - D.ns;
  if ((int)blockIdx.x < ncam) {
    if (lin_cameras)
      linearize_camera(D, blockIdx.x / D.cam_split, blockIdx.x % D.cam_split, reinterpret_cast<double(*)[27]>(s_raw),
                       reinterpret_cast<double(*)[27]>(s_raw + (kCamThreads / 2) * 27), D.n_scale == 0);
    return;
  }
  const int sb = (int)blockIdx.x - ncam;  // slab of this workgroup
  double(*sD)[12] = reinterpret_cast<double(*)[12]>(s_raw);
  double* sY = s_raw + kSmallPts * 12;
  double* sB = sY + kStage;
  const int np = D.np, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (D.nfl + D.ns - 1) / D.ns;
  const int l0 = sb * per, l1 = min(l0 + per, D.nfl);
  // item = (point of the batch, block of the point, block row): 3 contiguous values of an Hpl block
  constexpr int kItems = kSmallPts * kTileCams * 6, kRounds = (kItems + kSmallThreads - 1) / kSmallThreads;
  int blk[kRounds];
  auto block_ranges = [&](int b0, int nb) {  // round trip 1 of a batch: which Hpl block each item reads
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int q = (tid + kSmallThreads * r) / 6, pb = q / kTileCams, j = q - pb * kTileCams;
      blk[r] = -1;
      if (pb < nb) {
        const int f0 = D.fp_start[b0 + pb], f1 = D.fp_start[b0 + pb + 1];
        if (j < f1 - f0) blk[r] = f0 + j;
      }
    }
  };
  block_ranges(l0, min(kSmallPts, l1 - l0));  // requested before the LM state: the addresses do not depend on it
  // A camera that does not see a point contributes zero blocks: the staging is cleared (while the loads above are in
  // flight) and only the present blocks are written, so the product phase needs no per-camera branches -- its LDS loads
  // of a point are all issued up front.  Adding the exact zeros does not change a sum.
  for (int k = tid; k < 2 * kStage; k += kSmallThreads) sY[k] = 0.0;  // sY and sB are adjacent
  const lm_state st = *D.st;
  if (st.done) return;
  const lin_view L = lin_of(D, st.cur);
  const double lambda = st.lambda;
  const int arow = lane / kTileCams, ccam = lane - arow * kTileCams;  // lanes 60..63 (arow == 6) idle in the products
  double acc[kSmallRows][6], racc[kSmallRows];
#pragma unroll
  for (int r = 0; r < kSmallRows; ++r) {
    racc[r] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[r][c] = 0.0;
  }
  const int dpb = tid >> 5;  // thread 32 pb + 31 inverts point pb's Hll (one per half wave: the inversions run side by side)
  for (int b0 = l0; b0 < l1; b0 += kSmallPts) {
    const int nb = min(kSmallPts, l1 - b0);
    const bool dinv_thread = (tid & 31) == 31 && dpb < nb;
    // ---- round trip 1: block ranges (the first batch's are on their way already), Hll / bl
    if (b0 != l0) block_ranges(b0, nb);
    double Dm[9], bl[3];
    if (dinv_thread) {
      const size_t l = (size_t)(b0 + dpb);
#pragma unroll
      for (int k = 0; k < 9; ++k) Dm[k] = L.Hll[9 * l + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) bl[k] = L.bl[3 * l + k];
    }
    // ---- round trip 2: camera slots and block rows
    double v[kRounds][3];
    int slot[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
      if (blk[r] >= 0) {
        const int ar = (tid + kSmallThreads * r) % 6;
        slot[r] = D.fp_slot[blk[r]];
        const double* B = L.Hpl + 18 * (size_t)blk[r] + 3 * ar;
        v[r][0] = B[0];
        v[r][1] = B[1];
        v[r][2] = B[2];
      }
    if (dinv_thread) {
      double inv[9];
      Dm[0] += lambda;
      Dm[4] += lambda;
      Dm[8] += lambda;
      inv3(Dm, inv);
      const size_t l = (size_t)(b0 + dpb);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        sD[dpb][k] = inv[k];
        D.Dinv[9 * l + k] = inv[k];  // ba_point_trial's back-substitution reads it
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) sD[dpb][9 + k] = inv[3 * k] * bl[0] + inv[3 * k + 1] * bl[1] + inv[3 * k + 2] * bl[2];
    }
    __syncthreads();
    // ---- stage B = Hpl and Y = Hpl (Hll + lambda I)^-1 by camera slot
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
      if (blk[r] >= 0) {
        const int i = tid + kSmallThreads * r, ar = i % 6, pb = i / (6 * kTileCams);
        const int o = (pb * kTileCams + slot[r]) * 18 + 3 * ar;
        const double* sd = sD[pb];
        sB[o] = v[r][0];
        sB[o + 1] = v[r][1];
        sB[o + 2] = v[r][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) sY[o + c] = v[r][0] * sd[c] + v[r][1] * sd[3 + c] + v[r][2] * sd[6 + c];
      }
    __syncthreads();
    // ---- products into the register accumulators: wave w < kProdWaves owns the row cameras w, w + 4, w + 8; branch-free
    const int ar6 = arow < 6 ? arow : 0;  // lanes 60..63 compute on row 0's operands and never store
    for (int pb = 0; pb < (wv < kProdWaves ? nb : 0); ++pb) {
      double Bv[18];
      const double* B = sB + (pb * kTileCams + ccam) * 18;
#pragma unroll
      for (int k = 0; k < 18; ++k) Bv[k] = B[k];
      const double* sd = sD[pb];
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r) {
        const int ls = min(wv + kProdWaves * r, kTileCams - 1);  // a row beyond the tile repeats the last one, never stored
        const double* Y = sY + (pb * kTileCams + ls) * 18 + 3 * ar6;
        const double y0 = Y[0], y1 = Y[1], y2 = Y[2];
#pragma unroll
        for (int bc = 0; bc < 6; ++bc) acc[r][bc] += y0 * Bv[3 * bc] + y1 * Bv[3 * bc + 1] + y2 * Bv[3 * bc + 2];
        const double* Br = sB + (pb * kTileCams + ls) * 18 + 3 * ar6;
        racc[r] += Br[0] * sd[9] + Br[1] * sd[10] + Br[2] * sd[11];
      }
    }
    if (b0 + kSmallPts < l1) {  // the next batch overwrites the staging: clear it again first
      __syncthreads();
      for (int k = tid; k < 2 * kStage; k += kSmallThreads) sY[k] = 0.0;
    }
  }
  double* out = D.slab + (size_t)sb * ((size_t)np * np + np);
  if (arow < 6 && wv < kProdWaves) {
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) {
      const int ls = wv + kProdWaves * r;
      if (ls >= kTileCams) continue;
      const int row = 6 * ls + arow, col = 6 * ccam;
      if (row < np && col < np && ccam <= ls) {
#pragma unroll
        for (int bc = 0; bc < 6; ++bc) out[(size_t)row * np + col + bc] = acc[r][bc];
      }
      if (ccam == 0 && row < np) out[(size_t)np * np + row] = racc[r];
    }
  }
}

// S = Hpp + lambda I - sum_s slab_s ; bs = bp - sum_s bslab_s.  Workgroup = 64 elements x kRedSplit slab groups: group g
// sums the slabs g, g + kRedSplit, ... with all of its loads in flight at once (up to 16 per pass), the group sums are
// added in group order - fixed order, deterministic.  The Hpp / bp element is requested before the slabs, not after.
constexpr int kRedSplit = 16;
__global__ __launch_bounds__(64 * kRedSplit) void ba_reduce(ba_dev D) {
  __shared__ double s_part[kRedSplit][64];
  const lm_state st = *D.st;
  if (st.done) return;
  const int np = D.np;
  const int slab_elems = np * np + np;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  const lin_view L = lin_of(D, st.cur);
  // Two linearisations without scale edges: once a step has been accepted, the camera workgroups next to ba_schur_small
  // left the 27 sums of every camera in D.cam_part (parts of up to 8 workgroups, added here in part order) instead of
  // assembling Hpp / bp; bp is written out for the gain denominator of the dense solve.
  const bool from_parts = D.spec && D.n_scale == 0 && st.accepted > 0;
  double base = 0.0;
  if (g == 0 && i < slab_elems) {
    if (!from_parts) {
      base = i < np * np ? L.Hpp[i] : L.bp[i - np * np];
    } else {
      int cam, n = -1;
      if (i < np * np) {
        const int r = i / np, c = i - r * np;
        cam = r / 6;
        if (c / 6 == cam) {
          const int a = r - 6 * cam, b = c - 6 * cam, lo = min(a, b), hi = max(a, b);
          n = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);  // packed upper triangle, row-major
        }
      } else {
        cam = (i - np * np) / 6;
        n = 21 + (i - np * np) - 6 * cam;
      }
      if (n >= 0) {
        const double* parts = D.cam_part + (size_t)cam * D.cam_split * 27 + n;
        base = parts[0];
        for (int k = 1; k < D.cam_split; ++k) base += parts[k * 27];
        if (i >= np * np) L.bp[i - np * np] = base;
      }
    }
  }
  double acc = 0.0;
  // the tiled Schur kernels produce the lower triangle only (tiles, or 6x6 blocks for ba_schur_small) - the Cholesky
  // reads nothing else: elements above it have no slab contribution to fetch
  bool have = i < slab_elems;
  if (have && D.ntile >= 1 && i < np * np) {
    const int r = i / np, c = i - r * np;
    have = D.small ? r / 6 >= c / 6 : r / kTileN >= c / kTileN;
  }
  if (have) {
    for (int s = g; s < D.ns; s += 16 * kRedSplit) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = D.slab[(size_t)min(s + kRedSplit * u, D.ns - 1) * slab_elems + i];  // unconditional loads
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += s + kRedSplit * u < D.ns ? v[u] : 0.0;  // a missing slab contributes an exact zero
    }
  }
  s_part[g][e] = acc;
  __syncthreads();
  if (g != 0 || i >= slab_elems) return;
  acc = s_part[0][e];
#pragma unroll
  for (int k = 1; k < kRedSplit; ++k) acc += s_part[k][e];
  if (i < np * np) {
    const int r = i / np, c = i - r * np;
    if (r == c) base += st.lambda;
    D.S[i] = base - acc;
  } else {
    D.bs[i - np * np] = base - acc;
  }
}

// The window slabs (ba_schur_window) summed into S and bs: thread = element of the lower block triangle (a diagonal 6 x 6
// block takes its upper half from the mirrored element: the tiles above the diagonal are not produced).  The slabs are
// ordered by their first camera, so those whose window can hold both cameras of an element are the range
// [win_first[row camera - kWinCams + 1], win_first[column camera + 1]) of the host's table (win_first[c] = first slab
// that starts at camera c or later); they are added in slab order.
constexpr int kRedWinGroups = 4, kRedWinElems = 256 / kRedWinGroups, kRedWinInFlight = 8;
__global__ __launch_bounds__(256) void ba_reduce_window(ba_dev D) {
  extern __shared__ int s_win[];  // [ns][2]: first camera, cameras (one level of dependent loads less per slab)
  __shared__ double s_part[kRedWinGroups][kRedWinElems];
  const lm_state st = *D.st;
  if (st.done) return;
  const int np = D.np, ns = D.ns;
  for (int k = threadIdx.x; k < ns; k += 256) {
    s_win[2 * k] = D.win_w0[k];
    s_win[2 * k + 1] = D.win_len[k];
  }
  __syncthreads();
  // thread = (element e of this workgroup, group g): the element's slab range is cut into chunks of kRedWinInFlight slabs,
  // group g adds the chunks g, g + 4, ... in order with all loads of a chunk in flight; the groups are added in group
  // order -- a fixed order, and 32 independent loads per element instead of a chain of ~60 round trips
  const int e = threadIdx.x % kRedWinElems, g = threadIdx.x / kRedWinElems;
  const int i = blockIdx.x * kRedWinElems + e;
  const bool live = i < np * np + np;
  const bool mat = i < np * np;
  int r = 0, c = 0;
  bool upper = false;
  if (live) {
    if (mat) {
      r = i / np;
      c = i - r * np;
      upper = r / 6 < c / 6;  // above the block diagonal: never read, no slab contribution
      if (r < c) {
        const int t = r;
        r = c;
        c = t;
      }
    } else {
      r = c = i - np * np;
    }
  }
  double acc = 0.0;
  if (live && !upper) {
    const int cr = r / 6, cc = c / 6;
    const int s0 = D.win_first[max(cr - kWinCams + 1, 0)], s1 = D.win_first[cc + 1];
    for (int sb = s0 + g * kRedWinInFlight; sb < s1; sb += kRedWinGroups * kRedWinInFlight) {
      double v[kRedWinInFlight];
      bool use[kRedWinInFlight];
#pragma unroll
      for (int u = 0; u < kRedWinInFlight; ++u) {
        const int s = min(sb + u, s1 - 1);
        const int w0 = s_win[2 * s];
        use[u] = sb + u < s1 && cr < w0 + s_win[2 * s + 1];
        const double* slab = D.slab + (size_t)s * kWinSlabElems;
        // (an unused slab still gets a valid address: its own first element)
        const size_t off = !use[u] ? 0 : mat ? (size_t)(r - 6 * w0) * kWinN + (c - 6 * w0) : (size_t)kWinN * kWinN + (r - 6 * w0);
        v[u] = slab[off];
      }
#pragma unroll
      for (int u = 0; u < kRedWinInFlight; ++u)
        if (use[u]) acc += v[u];
    }
  }
  s_part[g][e] = acc;
  __syncthreads();
  if (g != 0 || !live) return;
  acc = s_part[0][e];
#pragma unroll
  for (int k = 1; k < kRedWinGroups; ++k) acc += s_part[k][e];
  const lin_view L = lin_of(D, st.cur);
  if (mat) {
    double base = L.Hpp[i];
    if (i / np == i % np) base += st.lambda;
    D.S[i] = base - acc;
  } else {
    D.bs[i - np * np] = L.bp[i - np * np] - acc;
  }
}

// ------------------------------------------------------------------------------------------------ dense solve
// One workgroup.  Right-looking Cholesky: the subtraction order per element equals the oracle's dot-product order.
template <bool IN_LDS>
__global__ __launch_bounds__(kSolveThreads) void ba_solve(ba_dev D) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  lm_state* st = D.st;
  if (st->done) return;
  const int n = D.np, tid = threadIdx.x;
  int& s_fail = *reinterpret_cast<int*>(s_mem);  // first 16 bytes of the dynamic region
  double* A = IN_LDS ? s_mem + 2 : D.S;
  double* x = IN_LDS ? s_mem + 2 + n * n : D.bs;
  if (IN_LDS) {
    for (int i = tid; i < n * n; i += kSolveThreads) A[i] = D.S[i];
    for (int i = tid; i < n; i += kSolveThreads) x[i] = D.bs[i];
  }
  if (tid == 0) s_fail = 0;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      const double d = A[(size_t)j * n + j];
      if (!(d > 0.0)) s_fail = 1;
      else A[(size_t)j * n + j] = sqrt(d);
    }
    __syncthreads();
    if (s_fail) break;
    const double ljj = A[(size_t)j * n + j];
    for (int i = j + 1 + tid; i < n; i += kSolveThreads) A[(size_t)i * n + j] = A[(size_t)i * n + j] / ljj;
    __syncthreads();
    // trailing update of the lower triangle: A[i][k] -= L[i][j] * L[k][j], j < k <= i
    const int rem = n - j - 1;
    for (int e = tid; e < rem * rem; e += kSolveThreads) {
      const int i = j + 1 + e / rem, k = j + 1 + e % rem;
      if (k <= i) A[(size_t)i * n + k] -= A[(size_t)i * n + j] * A[(size_t)k * n + j];
    }
    __syncthreads();
  }
  const int ok = !s_fail;
  if (ok) {
    // forward L y = b (column oriented: same subtraction order as the oracle), then L^T x = y
    for (int k = 0; k < n; ++k) {
      if (tid == 0) x[k] = x[k] / A[(size_t)k * n + k];
      __syncthreads();
      const double xk = x[k];
      for (int i = k + 1 + tid; i < n; i += kSolveThreads) x[i] -= A[(size_t)i * n + k] * xk;
      __syncthreads();
    }
    for (int k = n - 1; k >= 0; --k) {
      if (tid == 0) x[k] = x[k] / A[(size_t)k * n + k];
      __syncthreads();
      const double xk = x[k];
      for (int i = tid; i < k; i += kSolveThreads) x[i] -= A[(size_t)k * n + i] * xk;
      __syncthreads();
    }
    for (int i = tid; i < n; i += kSolveThreads) D.xp[i] = x[i];
  }
  __syncthreads();
  // trial camera states into the other buffer (SBACam::update), fixed cameras copied
  const int cur = st->cur;
  const double* c0 = D.cam[cur];
  double* c1 = D.cam[cur ^ 1];
  for (int p = tid; p < D.n_poses; p += kSolveThreads) {
    const double* src = c0 + (size_t)p * kCamStride;
    double* dst = c1 + (size_t)p * kCamStride;
    const int cs = D.pose_slot[p];
    if (cs < 0 || !ok) {
      for (int k = 0; k < kCamStride; ++k) dst[k] = src[k];
    } else {
      const double* d = x + 6 * cs;
      double t[3] = {src[0] + d[0], src[1] + d[1], src[2] + d[2]};
      const double bx = d[3], by = d[4], bz = d[5];
      const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));  // NaN for an oversized step -> trial rejected
      const double ax = src[3], ay = src[4], az = src[5], aw = src[6];
      const double w = aw * bw - ax * bx - ay * by - az * bz;
      const double xx = aw * bx + ax * bw + ay * bz - az * by;
      const double yy = aw * by + ay * bw + az * bx - ax * bz;
      const double zz = aw * bz + az * bw + ax * by - ay * bx;
      const double nrm = sqrt(xx * xx + yy * yy + zz * zz + w * w);
      double q[4] = {xx / nrm, yy / nrm, zz / nrm, w / nrm};
      for (int k = 0; k < 3; ++k) dst[k] = t[k];
      for (int k = 0; k < 4; ++k) dst[3 + k] = q[k];
      quat_to_w2n(t, q, dst + 7);
    }
  }
  if (tid == 0) {
    double sc = 0.0;
    if (ok)
      for (int j = 0; j < n; ++j) sc += x[j] * (st->lambda * x[j] + lin_of(D, st->cur).bp[j]);
    st->scale_pose = sc;
    st->solve_ok = ok;
    st->trials += 1;
    if (!ok) st->not_pd += 1;
  }
}

// Small systems (n <= 126): one 256-thread workgroup factorises in LDS.
//   * the right-hand side is appended as row n of the matrix, so the forward substitution L y = b happens inside the
//     factorisation (row n receives exactly the updates of a matrix row) -- no separate sequential pass
//   * blocked by the 6x6 camera blocks (n = 6 nb): per block column one thread factorises the diagonal block in
//     registers, every row below solves its 6 entries against it, then the trailing update is a 6-term dot product
//     per element tiled 16 x 16 over the threads -- nb dependent steps instead of n
//   * 1/L[j][j] from v_rsq_f64 + two Newton steps (one correction for L[j][j] itself): no IEEE sqrt/divide sequence
//     in the dependent chain; the results differ from the oracle's sqrt/divide by <= 2 ulp
//   * row stride is odd so the 16 row-owners of a tile hit distinct banks
//   * the backward substitution L^T x = y runs on wave 0 alone, wave-synchronously

constexpr int kSolveTile = 32;  // threads as rows of kSolveTile over matrix tiles
#ifdef VS_SOLVE_STAMPS  // developer build: cycles of thread 0 per phase of a step, printed for the third trial of a solve
#define VS_SOLVE_LAP(x) { const long long t_ = __builtin_readcyclecounter(); x += t_ - sts0; sts0 = t_; }
#else
#define VS_SOLVE_LAP(x)
#endif

// (256 / 512 threads for small systems were measured in round 5 and are no faster: a step is the chain on wave 0, not the barriers --
// profiles/r05_solve_stamps.txt, profiles/tried_and_dropped.md)
constexpr int kSolveBlock = 1024;
// PACKED: the rows of the lower triangle one behind the other (row r at r (r + 1) / 2, the right-hand side as row n) instead of a
// square: 160 KB of LDS then hold 198 unknowns (33 free cameras) instead of 126 (21) -- systems that otherwise take two launches
// per 24 columns through HBM (ba_chol_panel / ba_chol_update).  Same operations in the same order: the same bits.
template <bool PACKED>
__global__ __launch_bounds__(kSolveBlock) void ba_solve_block(ba_dev D) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  lm_state* st = D.st;
  // the state, the system and everything the epilogue needs are requested together (one round trip); a finished solve
  // pays for the unused loads, which is rare and cheap
  const int done = st->done, cur = st->cur;
  const double lambda = st->lambda;
  const int n = D.np, tid = threadIdx.x, tx = tid % kSolveTile, ty = tid / kSolveTile;
#ifndef VS_SOLVE_NO_PRIO
  if (tid < 64) __builtin_amdgcn_s_setprio(3);  // wave 0 factorises the diagonal blocks: the critical path of every step
#endif
  const int ld = (n + 1) | 1;
  auto IX = [ld](int r, int c) { return PACKED ? ((r * (r + 1)) >> 1) + c : r * ld + c; };
  double* A = s_mem;                        // (n + 1) rows (of ld, or packed): rows 0..n-1 matrix, row n = rhs
  double* rinv = s_mem + (PACKED ? (size_t)(n + 1) * (n + 2) / 2 : (size_t)(n + 1) * ld);  // [n]
  double* s_bp = rinv + n + 1;              // [n] right-hand side before the Schur complement (gain denominator)
  int* s_flag = reinterpret_cast<int*>(s_bp + n + 1);
  for (int r = ty; r < n; r += kSolveBlock / kSolveTile)  // the lower triangle: nothing else is read
    for (int c = tx; c <= r; c += kSolveTile) A[IX(r, c)] = D.S[(size_t)r * n + c];
  const double* g_bp = lin_of(D, cur).bp;
  for (int c = tid; c < n; c += kSolveBlock) {
    A[IX(n, c)] = D.bs[c];
    s_bp[c] = g_bp[c];
  }
  if (tid == 0) *s_flag = 0;
  if (done) return;  // uniform
  __syncthreads();
  // ---- blocked right-looking Cholesky on the 6x6 camera blocks (n = 6 * nb); row n carries the right-hand side.
  // Per block column: (1) thread 0 factorises the diagonal block in registers, (2) every row below solves its 6 entries
  // against it, (3) the trailing update.  (1) is a long dependent chain on one lane, so it is taken off the critical
  // path: in step J's trailing update wave 0 updates the NEXT diagonal block first and factorises it while the other
  // waves update the rest of the trailing matrix -- two barriers per block column, and the chain of (1) overlaps (3).
  int ok = 1;
  const int nb = n / 6;
  auto factor_diag = [&](int j0) {  // thread 0: L_JJ in registers, reciprocal pivots to rinv
    double L[6][6], ri[6];
    int good = 1;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) L[r][c] = A[IX(j0 + r, j0 + c)];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double d = L[c][c];
      if (!(d > 0.0)) good = 0;
      double r = __builtin_amdgcn_rsq(d);
      r = r * (1.5 - 0.5 * d * r * r);
      r = r * (1.5 - 0.5 * d * r * r);
      double l = d * r;
      l = l + 0.5 * r * (d - l * l);
      L[c][c] = l;
      ri[c] = r;
#pragma unroll
      for (int i = c + 1; i < 6; ++i) L[i][c] = L[i][c] * r;
#pragma unroll
      for (int i = c + 1; i < 6; ++i)
#pragma unroll
        for (int k = c + 1; k <= i; ++k) L[i][k] -= L[i][c] * L[k][c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) A[IX(j0 + r, j0 + c)] = L[r][c];
      rinv[j0 + r] = ri[r];
    }
    if (!good) *s_flag = 1;
  };
#ifdef VS_SOLVE_STAMPS
  long long stsP = 0, stsB1 = 0, stsU = 0, stsF = 0, stsB2 = 0, sts0 = __builtin_readcyclecounter();
  const long long stsBegin = sts0;
#endif
  if (tid == 0 && nb > 0) factor_diag(0);
  __syncthreads();
  VS_SOLVE_LAP(stsF)
  for (int J = 0; J < nb; ++J) {
    const int j0 = 6 * J;
    if (*s_flag) {  // uniform (written before the last barrier)
      ok = 0;
      break;
    }
    // (2) panel: every row below the block (and the rhs row) solves  x * L_JJ^T = a  (forward substitution, 6 steps)
    for (int r = j0 + 6 + tid; r <= n; r += kSolveBlock) {
      double a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) a[c] = A[IX(r, j0 + c)];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= a[k] * A[IX(j0 + c, j0 + k)];
        a[c] = v * rinv[j0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) A[IX(r, j0 + c)] = a[c];
    }
    VS_SOLVE_LAP(stsP)
    __syncthreads();
    VS_SOLVE_LAP(stsB1)
    // (3) trailing update: A[r][c] -= sum_k A[r][j0+k] * A[c][j0+k]  for r > j0+5 (incl. rhs row), j0+5 < c <= min(r, n-1).
    // Wave 0 owns the next diagonal block (rows/columns j0+6 .. j0+11): 21 lanes update it, then lane 0 factorises it;
    // the other waves take the rest on a (kSolveTile - 2) x kSolveTile grid of their own.
    const bool next_diag = J + 1 < nb;
    if (tid < 64) {
      if (next_diag) {
        if (tid < 21) {
          int r = 0, c = tid;
          while (c > r) {  // lane -> (r, c) of the lower triangle, row-major
            c -= r + 1;
            ++r;
          }
          const int rr = j0 + 6 + r, cc = j0 + 6 + c;
          double acc = A[IX(rr, cc)];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc -= A[IX(rr, j0 + k)] * A[IX(cc, j0 + k)];
          A[IX(rr, cc)] = acc;
        }
        wave_lds_sync();
        VS_SOLVE_LAP(stsU)
        if (tid == 0) factor_diag(j0 + 6);
        VS_SOLVE_LAP(stsF)
      }
    } else if ((tid >> 6) & 3) {
      // (waves 4, 8, 12 share wave 0's SIMD: they stay out of the update, the factorisation has the SIMD's issue slots to itself)
      const int wv_ = tid >> 6, t2 = (wv_ - 1 - (wv_ >> 2)) * 64 + (tid & 63), ux = t2 % kSolveTile, uy = t2 / kSolveTile;
      constexpr int kRowsPer = (kSolveBlock / 64 - kSolveBlock / 256) * 64 / kSolveTile;  // twelve waves' threads as rows of kSolveTile
      for (int r = j0 + (next_diag ? 12 : 6) + uy; r <= n; r += kRowsPer) {  // rows j0+6..j0+11 lie inside wave 0's block
        double ar[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) ar[k] = A[IX(r, j0 + k)];
        const int cmax = r < n ? r : n - 1;
        for (int c = j0 + 6 + ux; c <= cmax; c += kSolveTile) {
          double acc = A[IX(r, c)];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc -= ar[k] * A[IX(c, j0 + k)];
          A[IX(r, c)] = acc;
        }
      }
    }
    __syncthreads();
    VS_SOLVE_LAP(stsB2)
  }
#ifdef VS_SOLVE_STAMPS
  if (tid == 0 && st->trials == 2)
    printf("ba_solve_block<%d> n=%d: %d steps; cycles of thread 0: panel %lld, barrier after the panel %lld, next diagonal block updated %lld, "
           "factorised %lld, barrier after the update %lld; factorisation loop %lld of which first block + load wait %lld (per step %lld)\n",
           kSolveBlock, n, nb, stsP, stsB1, stsU, stsF, stsB2, (long long)__builtin_readcyclecounter() - stsBegin, 0LL,
           nb ? ((long long)__builtin_readcyclecounter() - stsBegin) / nb : 0LL);
#endif
  if (ok && *s_flag) ok = 0;  // the last look-ahead factorisation failed
  // backward substitution L^T x = y on wave 0, by 6x6 blocks from the bottom: lane 0 solves the block's triangular
  // system in registers (descending k, as the element-wise recurrence does), then the lanes subtract the block's
  // contribution from the rows above -- nb dependent steps instead of n.  x lives in row n.
  double* x = A + IX(n, 0);
  if (ok && tid < 64) {
    for (int J = nb - 1; J >= 0; --J) {
      const int j0 = 6 * J;
      if (tid == 0) {
        double xb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xb[k] = x[j0 + k];
#pragma unroll
        for (int k = 5; k >= 0; --k) {
          xb[k] = xb[k] * rinv[j0 + k];
#pragma unroll
          for (int i = 0; i < k; ++i) xb[i] -= A[IX(j0 + k, j0 + i)] * xb[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) x[j0 + k] = xb[k];
      }
      wave_lds_sync();
      for (int i = tid; i < j0; i += 64) {
        double v = x[i];
#pragma unroll
        for (int k = 5; k >= 0; --k) v -= A[IX(j0 + k, i)] * x[j0 + k];
        x[i] = v;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  if (ok)
    for (int i = tid; i < n; i += kSolveBlock) D.xp[i] = x[i];
  // trial camera states into the other buffer (SBACam::update), fixed cameras copied
  const double* c0 = D.cam[cur];
  double* c1 = D.cam[cur ^ 1];
  for (int p = tid; p < D.n_poses; p += kSolveBlock) {
    const double* src = c0 + (size_t)p * kCamStride;
    double* dst = c1 + (size_t)p * kCamStride;
    const int cs = D.pose_slot[p];
    if (cs < 0 || !ok) {
      for (int k = 0; k < kCamStride; ++k) dst[k] = src[k];
    } else {
      const double* d = x + 6 * cs;
      double t[3] = {src[0] + d[0], src[1] + d[1], src[2] + d[2]};
      const double bx = d[3], by = d[4], bz = d[5];
      const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));  // NaN for an oversized step -> trial rejected
      const double ax = src[3], ay = src[4], az = src[5], aw = src[6];
      const double w = aw * bw - ax * bx - ay * by - az * bz;
      const double xx = aw * bx + ax * bw + ay * bz - az * by;
      const double yy = aw * by + ay * bw + az * bx - ax * bz;
      const double zz = aw * bz + az * bw + ax * by - ay * bx;
      const double inrm = vs_fast_rsq(xx * xx + yy * yy + zz * zz + w * w);  // as ba_motion_step does
      double q[4] = {xx * inrm, yy * inrm, zz * inrm, w * inrm};
      for (int k = 0; k < 3; ++k) dst[k] = t[k];
      for (int k = 0; k < 4; ++k) dst[3 + k] = q[k];
      quat_to_w2n(t, q, dst + 7);
    }
  }
  // gain denominator x^T (lambda x + b): the last wave sums it (lane l takes the terms l, l+64; butterfly over the wave)
  if (tid >= kSolveBlock - 64) {
    const int lane = tid & 63;
    double sc = 0.0;
    if (ok)
      for (int j = lane; j < n; j += 64) sc += x[j] * (lambda * x[j] + s_bp[j]);
    sc = vs_group_reduce<6>(sc);
    if (lane == 0) {
      st->scale_pose = sc;
      st->solve_ok = ok;
      st->trials += 1;
      if (!ok) st->not_pd += 1;
    }
  }
}
template __global__ void ba_solve_block<false>(ba_dev D);
template __global__ void ba_solve_block<true>(ba_dev D);

// Large systems (n > 126): blocked right-looking Cholesky in HBM, two launches per block column.
//   ba_chol_panel(j0, nbw): one workgroup takes columns [j0, j0+nbw) of rows j0..n (row n = rhs, as above) into LDS,
//     factorises them with the same 6x6 scheme as ba_solve_block and writes them back;
//   ba_chol_update(j0, nbw): 32x32 tiles of the trailing lower triangle (and the rhs row) subtract the panel's
//     contribution, every element in ascending column order -- the sequential algorithm's order;
//   ba_chol_finish: blocked backward substitution by one workgroup, then the camera update epilogue.
constexpr int kPanelThreads = 512;
constexpr int kUpdTile = 32;

__device__ inline double* chol_row(const ba_dev& D, int r) { return r < D.np ? D.S + (size_t)r * D.np : D.bs; }

__global__ __launch_bounds__(kPanelThreads) void ba_chol_panel(ba_dev D, int j0, int nbw) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  if (D.st->done) return;
  const int n = D.np, tid = threadIdx.x;
  if (j0 == 0) {
    if (tid == 0) *D.chol_fail = 0;
  } else if (*D.chol_fail) {
    return;
  }
  const int rows = n + 1 - j0, ld = nbw | 1;
  double* A = s_mem;                       // [rows][ld]
  double* rinv = s_mem + (size_t)rows * ld;  // [nbw]
  int* s_flag = reinterpret_cast<int*>(rinv + nbw);
  for (int e = tid; e < rows * nbw; e += kPanelThreads) {
    const int r = e / nbw, c = e - r * nbw;
    A[r * ld + c] = chol_row(D, j0 + r)[j0 + c];
  }
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  // Per 6-column step: (1) the diagonal block on one lane, (2) every row below solves its six entries against it, (3) the panel's
  // remaining columns are updated.  (1) is a chain of ~160 instructions on ONE lane; since round 5 it is taken off the critical
  // path as ba_solve_block does: in step b's update wave 0 first updates the NEXT diagonal block (21 lanes), then factorises it
  // (lane 0), while the other waves update the rest -- two barriers per step instead of three, and the chain runs beside the
  // update.  (A factorisation of 306 unknowns -- 52 key frames of a growing global bundle adjustment -- is 13 launches of this
  // kernel per LM trial: tools/ba_growth.py.)  Every element receives the same subtractions in the same order as before.
  auto factor_diag = [&](int b0) {  // thread 0: the 6 x 6 block at (b0, b0) in registers, reciprocal pivots to rinv
    double L[6][6], ri[6];
    int good = 1;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) L[r][c] = A[(b0 + r) * ld + b0 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double d = L[c][c];
      if (!(d > 0.0)) good = 0;
      double r = __builtin_amdgcn_rsq(d);
      r = r * (1.5 - 0.5 * d * r * r);
      r = r * (1.5 - 0.5 * d * r * r);
      double l = d * r;
      l = l + 0.5 * r * (d - l * l);
      L[c][c] = l;
      ri[c] = r;
#pragma unroll
      for (int i = c + 1; i < 6; ++i) L[i][c] = L[i][c] * r;
#pragma unroll
      for (int i = c + 1; i < 6; ++i)
#pragma unroll
        for (int k = c + 1; k <= i; ++k) L[i][k] -= L[i][c] * L[k][c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) A[(b0 + r) * ld + b0 + c] = L[r][c];
      rinv[b0 + r] = ri[r];
    }
    if (!good) *s_flag = 1;
  };
  if (tid == 0) factor_diag(0);
  __syncthreads();
  for (int b0 = 0; b0 < nbw; b0 += 6) {
    if (*s_flag) break;  // uniform (written before the last barrier)
    // (2) every row below the block (and the rhs row)
    for (int r = b0 + 6 + tid; r < rows; r += kPanelThreads) {
      double a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) a[c] = A[r * ld + b0 + c];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= a[k] * A[(b0 + c) * ld + b0 + k];
        a[c] = v * rinv[b0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r * ld + b0 + c] = a[c];
    }
    __syncthreads();
    // (3) update of the panel's remaining columns b0+6 .. nbw-1.  Wave 0 owns the next diagonal block (rows and columns b0+6 ..
    // b0+11): 21 lanes update its lower triangle, lane 0 factorises it; the other waves (not the one that shares wave 0's SIMD)
    // take the rows from b0+12 on: thread = (row, column), 6-term dot product.
    const int wc = nbw - b0 - 6;
    if (wc > 0) {
      if (tid < 64) {
        if (tid < 21) {
          int r = 0, c = tid;
          while (c > r) {  // lane -> (r, c) of the lower triangle, row-major
            c -= r + 1;
            ++r;
          }
          const int rr = b0 + 6 + r, cc = b0 + 6 + c;
          double acc = A[rr * ld + cc];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc -= A[rr * ld + b0 + k] * A[cc * ld + b0 + k];
          A[rr * ld + cc] = acc;
        }
        wave_lds_sync();
        if (tid == 0) factor_diag(b0 + 6);
      } else if ((tid >> 6) & 3) {
        constexpr int kWorkers = (kPanelThreads / 64 - kPanelThreads / 256) * 64;  // the waves that share no SIMD with wave 0
        const int wv_ = tid >> 6, t2 = (wv_ - 1 - (wv_ >> 2)) * 64 + (tid & 63);
        for (int e = t2; e < (rows - b0 - 12) * wc; e += kWorkers) {
          const int r = b0 + 12 + e / wc, c = b0 + 6 + e % wc;
          if (c > r && r < rows - 1) continue;  // strictly upper part of the matrix rows (the rhs row keeps all columns)
          double acc = A[r * ld + c];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc -= A[r * ld + b0 + k] * A[c * ld + b0 + k];
          A[r * ld + c] = acc;
        }
      }
      __syncthreads();
    }
  }
  if (*s_flag) {
    if (tid == 0) *D.chol_fail = 1;
    return;
  }
  for (int e = tid; e < rows * nbw; e += kPanelThreads) {
    const int r = e / nbw, c = e - r * nbw;
    if (c <= r || r == rows - 1) chol_row(D, j0 + r)[j0 + c] = A[r * ld + c];
  }
  for (int c = tid; c < nbw; c += kPanelThreads) D.rinv[j0 + c] = rinv[c];
}

// grid (T, T) over the trailing block starting at c0 = j0 + nbw; tiles above the diagonal exit
__global__ __launch_bounds__(256) void ba_chol_update(ba_dev D, int j0, int nbw) {
  __shared__ double sR[kUpdTile][25], sC[kUpdTile][25];
  if (D.st->done || *D.chol_fail) return;
  if (blockIdx.x > blockIdx.y) return;
  const int n = D.np, c0 = j0 + nbw, tid = threadIdx.x;
  const int r_base = c0 + blockIdx.y * kUpdTile, c_base = c0 + blockIdx.x * kUpdTile;
  const int tx = tid & 15, ty = tid >> 4;
  // the thread's 2 x 2 elements stay in registers while the panel's columns pass through LDS 24 at a time (panels of 48 columns
  // since round 5: half the launches of a factorisation; every element still receives its subtractions in ascending column order)
  double acc[2][2];
  bool live[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = r_base + ty + 16 * a, c = c_base + tx + 16 * b;
      live[a][b] = !(r > n || c >= n || (c > r));
      acc[a][b] = live[a][b] ? chol_row(D, r)[c] : 0.0;
    }
  for (int kh = 0; kh < nbw; kh += 24) {
    const int kw = min(24, nbw - kh);
    if (kh) __syncthreads();  // the previous slice has been consumed
    for (int e = tid; e < kUpdTile * kw; e += 256) {
      const int r = e / kw, k = e - r * kw;
      sR[r][k] = r_base + r <= n ? chol_row(D, r_base + r)[j0 + kh + k] : 0.0;
      sC[r][k] = c_base + r < n ? D.S[(size_t)(c_base + r) * n + j0 + kh + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int lr = ty + 16 * a, lc = tx + 16 * b;
        double v = acc[a][b];
        for (int k = 0; k < kw; ++k) v -= sR[lr][k] * sC[lc][k];
        acc[a][b] = v;
      }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (live[a][b]) chol_row(D, r_base + ty + 16 * a)[c_base + tx + 16 * b] = acc[a][b];
}

// Banded systems (ba_schur_window's windows: S[r][c] = 0 for r - c >= band): the whole factorisation in ONE launch, one
// 6 x 6 block column per step.  The rows the current block column reaches (band + 5 of them) live in LDS as a RING in both
// directions -- element (i, c) at [i mod WR][c mod WR], WR = band + 12 -- so nothing is ever moved: the six rows that enter
// the window for the next step are written into slots no current row uses, straight from the registers they were fetched
// into a step earlier.  A step is
//     diagonal 6 x 6 block (one lane; a chain of six dependent reciprocal square roots, ~0.5 us: THE critical path)
//  X  column solve of the rows below and of the right-hand side (thread = row; result to LDS and to HBM from registers)
//  Y  rank-6 update of the trailing lower triangle (2 x 2 register tiles) -- wave 0 takes the NEXT diagonal block's 21
//     elements first and factorises it at once, while the other waves update the rest and bring the new rows in
// so the diagonal chain of step b + 1 runs beside the update of step b (look-ahead) and a step costs two barriers.  Per
// element the subtractions are those of ba_chol_panel / ba_chol_update in their order (ascending column, one product at a
// time), so L is bit-identical to the dense path's; what lies outside the band is an exact zero there.
// Round 3's form (24-column panels, the window slid through registers after each) spent 27 % of its 420 us at 594 unknowns
// with 511 threads waiting for the diagonal lane, 30 % moving the window and writing panels, 22 % in the trailing update.
constexpr int kBandMax = 6 * kWinCams, kBandRing = kBandMax + 12, kBandLd = kBandRing | 1, kBandPld = kBandMax + 8;
constexpr size_t kBandLds = sizeof(double) * ((size_t)kBandRing * kBandLd + kBandRing + 6 * kBandPld + 128) + 64;
constexpr int kBandThreads = 1024, kBandLoaders = kBandThreads - 64;  // sixteen waves; fifteen of them update while wave 0 factorises the next diagonal block
constexpr int kBandNew = (6 * (kBandMax + 6) + kBandLoaders - 1) / kBandLoaders;

__global__ __launch_bounds__(kBandThreads) void ba_chol_band(ba_dev D, int band) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  if (D.st->done) return;
  const int n = D.np, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int WR = band + 12;
  if (wv == 0) __builtin_amdgcn_s_setprio(3);  // the wave that factorises the diagonal blocks is the critical path: it goes first on its SIMD
  constexpr int ld = kBandLd, pld = kBandPld;
  double* A = s_mem;                     // [WR][ld] ring
  double* Y = A + kBandRing * ld;        // [WR] right-hand side entries by column slot
  double* P = Y + kBandRing;             // [6][pld] the solved block column, transposed: P[k][t] = L[c0 + 6 + t][c0 + k]
  double* Ld = P + 6 * pld;              // [6][6] the factorised diagonal block
  double* blk = Ld + 36;                 // [21] the next diagonal block before its factorisation
  double* rv = blk + 24;                 // [6] reciprocal pivots of the current block column
  double* yb = rv + 8;                   // [6] solved right-hand side entries of the current block column
  int* s_flag = reinterpret_cast<int*>(yb + 8);
  {
    const int R0 = min(n, band + 5);
    for (int e = tid; e < R0 * R0; e += kBandThreads) {
      const int r = e / R0, c = e - r * R0;
      if (c <= r) A[r * ld + c] = D.S[(size_t)r * n + c];
    }
    for (int c = tid; c < R0; c += kBandThreads) Y[c] = D.bs[c];
  }
  if (tid == 0) {
    *s_flag = 0;
    *D.chol_fail = 0;
  }
  __syncthreads();
#ifdef VS_BAND_STAMPS
  long long td[5] = {0, 0, 0, 0, 0};
#define VS_DIAG_LAP(k) { const long long t_ = __builtin_readcyclecounter(); td[k] += t_ - td0; td0 = t_; }
#else
#define VS_DIAG_LAP(k)
#endif
  // the diagonal block whose first column is global column g0 (ring slot sg): lanes e < 21 of wave 0 bring its elements --
  // after the rank-6 update by the block column in P when `upd` -- to `blk`, lane 0 factorises (ba_chol_panel's
  // arithmetic), stores L in Ld and in HBM and the reciprocal pivots in rv / D.rinv
  auto diag = [&](int g0, int sg, bool upd) {
#ifdef VS_BAND_STAMPS
    long long td0 = __builtin_readcyclecounter();
#endif
    if (lane < 21) {
      int r = 0, e = lane;
      while (e > r) e -= ++r;  // (r, c): lane = r (r + 1) / 2 + c
      const int c = e;
      const int sr = sg + r >= WR ? sg + r - WR : sg + r, sc = sg + c >= WR ? sg + c - WR : sg + c;
      double acc = A[sr * ld + sc];
      if (upd) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc -= P[k * pld + r] * P[k * pld + c];
      }
      blk[lane] = acc;
    }
    wave_lds_sync();
    VS_DIAG_LAP(0)
    if (lane == 0) {
      double L[6][6], ri[6];
      int good = 1;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) L[r][c] = blk[r * (r + 1) / 2 + c];
#ifdef VS_BAND_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      VS_DIAG_LAP(1)
#endif
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double d = L[c][c];
        if (!(d > 0.0)) good = 0;
        double r = __builtin_amdgcn_rsq(d);
        r = r * (1.5 - 0.5 * d * r * r);
        r = r * (1.5 - 0.5 * d * r * r);
        double l = d * r;
        l = l + 0.5 * r * (d - l * l);
        L[c][c] = l;
        ri[c] = r;
#pragma unroll
        for (int i = c + 1; i < 6; ++i) L[i][c] = L[i][c] * r;
#pragma unroll
        for (int i = c + 1; i < 6; ++i)
#pragma unroll
          for (int k = c + 1; k <= i; ++k) L[i][k] -= L[i][c] * L[k][c];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          Ld[6 * r + c] = L[r][c];
          D.S[(size_t)(g0 + r) * n + g0 + c] = L[r][c];
        }
        rv[r] = ri[r];
        D.rinv[g0 + r] = ri[r];
      }
      if (!good) *s_flag = 1;
      VS_DIAG_LAP(2)
    }
  };
  if (wv == 0) diag(0, 0, false);
  int s0 = 0;  // ring slot of column c0
#ifdef VS_BAND_STAMPS
  long long tb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tb0 = __builtin_readcyclecounter(), tbS = tb0;
#define VS_BAND_LAP(k) { const long long t_ = __builtin_readcyclecounter(); tb[k] += t_ - tb0; tb0 = t_; }
#else
#define VS_BAND_LAP(k)
#endif
  for (int c0 = 0; c0 < n; c0 += 6) {
    const int m = min(band + 5, n - c0) - 6;  // trailing rows of this block column: c0 + 6 .. c0 + 6 + m - 1
    auto slot = [&](int d) { return s0 + d >= WR ? s0 + d - WR : s0 + d; };  // of global index c0 + d, 0 <= d < WR
    // the rows that enter the window for the next step, requested now: g = c0 + band + 5 + rr, columns c0 + 6 .. g
    const int g_first = c0 + band + 5;
    double nv[kBandNew];
    if (wv > 0) {
#pragma unroll
      for (int q = 0; q < kBandNew; ++q) {
        const int e = tid - 64 + kBandLoaders * q, rr = e / (band + 6), cc = e - rr * (band + 6), g = g_first + rr;
        nv[q] = 0.0;
        if (rr < 6 && g < n) {
          if (cc < band + rr) nv[q] = D.S[(size_t)g * n + c0 + 6 + cc];
          else if (cc == band + 5) nv[q] = D.bs[g];
        }
      }
    }
    VS_BAND_LAP(0)  // prefetch issue
    lds_barrier();  // X: the diagonal block of this column is factorised; the previous step's update is complete
    VS_BAND_LAP(1)  // wait at X
    if (*s_flag) {
      if (tid == 0) *D.chol_fail = 1;
      return;
    }
    // ---- column solve: thread t < m takes row c0 + 6 + t, thread m the right-hand side
    if (tid <= m) {
      const double* row = tid < m ? A + slot(6 + tid) * ld : Y;
      double a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) a[c] = row[slot(c)];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= a[k] * Ld[6 * c + k];
        a[c] = v * rv[c];
      }
      double* out = tid < m ? D.S + (size_t)(c0 + 6 + tid) * n + c0 : D.bs + c0;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (tid < m) P[c * pld + tid] = a[c];
        else yb[c] = a[c];
        out[c] = a[c];
      }
    }
    VS_BAND_LAP(2)  // solve
    lds_barrier();  // Y
    VS_BAND_LAP(3)  // wait at Y
    if (m <= 0) break;
    if (wv == 0) {
      diag(c0 + 6, slot(6), true);
    } else {
      const int t = tid - 64;
      // 2 x 2 tiles of the trailing lower triangle, rows 2 tr, 2 tr + 1 and columns 2 tc, 2 tc + 1 relative to c0 + 6, without
      // the next diagonal block (tr < 3), which wave 0 owns.  One tile per thread and many waves per SIMD: a lone wave issues
      // an FP64 multiply or add only every ~16 cycles (stamps: a 4 x 4 tile per thread on three waves took 4 000 cycles).
      const int nt = (m + 1) >> 1, total = nt * (nt + 1) / 2;
      // (the waves that share wave 0's SIMD -- 4, 8, 12 -- take no tiles: they would take issue slots from the factorisation)
      const int tw = (wv & 3) ? (wv - 1 - (wv >> 2)) * 64 + lane : total;
      for (int e = tw + 6; e < total; e += 12 * 64) {
        int tr = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (tr * (tr + 1) / 2 > e) --tr;
        while ((tr + 1) * (tr + 2) / 2 <= e) ++tr;
        const int tc = e - tr * (tr + 1) / 2;
        const int i0 = 2 * tr, j0 = 2 * tc;
        // (the ring length and every block start are even: a column pair never straddles the wrap)
        const int sj = slot(6 + j0), si0 = slot(6 + i0) * ld, si1 = slot(6 + min(i0 + 1, m - 1)) * ld;
        double acc00 = A[si0 + sj], acc01 = A[si0 + sj + 1], acc10 = A[si1 + sj], acc11 = A[si1 + sj + 1];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double r0 = P[k * pld + i0], r1 = P[k * pld + i0 + 1], q0 = P[k * pld + j0], q1 = P[k * pld + j0 + 1];  // (beyond m: padding, never stored)
          acc00 -= r0 * q0;
          acc01 -= r0 * q1;
          acc10 -= r1 * q0;
          acc11 -= r1 * q1;
        }
        A[si0 + sj] = acc00;
        if (j0 + 1 <= i0) A[si0 + sj + 1] = acc01;
        if (i0 + 1 < m) {
          A[si1 + sj] = acc10;
          A[si1 + sj + 1] = acc11;  // j0 + 1 <= i0 + 1
        }
      }
      VS_BAND_LAP(5)
      // the right-hand side entries of the trailing columns
      for (int j = t; j < m; j += kBandLoaders) {
        const int sj = slot(6 + j);
        double acc = Y[sj];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc -= yb[k] * P[k * pld + j];
        Y[sj] = acc;
      }
      VS_BAND_LAP(6)
      // the new rows
#pragma unroll
      for (int q = 0; q < kBandNew; ++q) {
        const int e = t + kBandLoaders * q, rr = e / (band + 6), cc = e - rr * (band + 6), g = g_first + rr;
        if (rr < 6 && g < n) {
          if (cc < band + rr) A[slot(band + 5 + rr) * ld + slot(6 + cc)] = nv[q];
          else if (cc == band + 5) Y[slot(band + 5 + rr)] = nv[q];
        }
      }
    }
    s0 = s0 + 6 >= WR ? s0 + 6 - WR : s0 + 6;
    VS_BAND_LAP(4)  // diagonal block (wave 0) / update + new rows (the others)
  }
#ifdef VS_BAND_STAMPS
  if (tid == 0) printf("  diagonal block: update + sync %lld, load %lld, chain + stores %lld\n", td[0], td[1], td[2]);
  if (lane == 0 && (wv < 2 || wv == 9 || wv == 15))
    printf("ba_chol_band n %d band %d wave %d: cycles prefetch issue %lld, wait X %lld, solve %lld, wait Y %lld, %s %lld (tiles %lld, rhs %lld), total %lld\n", n, band, wv, tb[0], tb[1], tb[2],
           tb[3], wv ? "new rows" : "diagonal", tb[4], tb[5], tb[6], (long long)__builtin_readcyclecounter() - tbS);
#endif
}

__global__ __launch_bounds__(kPanelThreads) void ba_chol_finish(ba_dev D, int nbw) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  lm_state* st = D.st;
  if (st->done) return;
  const int n = D.np, tid = threadIdx.x;
  double* x = s_mem;  // [n]
  const int ok = !*D.chol_fail;
  if (ok) {
    for (int i = tid; i < n; i += kPanelThreads) x[i] = D.bs[i];
    __syncthreads();
    if (D.band > 0 && D.band <= 128) {
      // Banded L: one wave, x in registers.  Lane l holds x[64 m + l] of the chunk being finished (xa) and of the two chunks
      // below it (xb, xc: a band of at most 128 reaches no further); step k -- from the last unknown down -- broadcasts
      // x_k = x[k] / L[k][k] out of its lane and subtracts L[k][i] x_k from every x_i the band reaches: each x_i receives
      // its subtractions in descending k, one product at a time, exactly as the blocked form below orders them (same
      // bits).  Row k of L is three coalesced loads that do not depend on x: eight rows are requested ahead.  A step is a
      // dozen instructions of one wave (~110 cycles), against two LDS round trips and two barriers per unknown in the
      // blocked form.
      if (tid < 64) {
        const int nch = (n + 63) >> 6;
        auto ld_y = [&](int ch) { const int i = 64 * ch + tid; return ch >= 0 && i < n ? x[i] : 0.0; };
        double xa = ld_y(nch - 1), xb = ld_y(nch - 2), xc = ld_y(nch - 3);
        constexpr int G = 8;
        double la0[G], lb0[G], lc0[G], la1[G], lb1[G], lc1[G];  // two buffers, named apart: a run-time buffer index would put them in scratch
        // rows k_hi, k_hi - 1, ... of chunk m, columns 64 (m - j) + lane.  The loads are unconditional (addresses clamped into
        // the matrix) so that the compiler can count them: a conditional load makes every later wait a wait for all of them.
        // What lies on or above the diagonal, beyond the band or outside the matrix is masked when the value is used.
        auto fetch = [&](double* la, double* lb, double* lc, int m, int k_hi) {
          const int ca = 64 * m + tid, a_ = min(ca, n - 1), b_ = max(ca - 64, 0), c_ = max(ca - 128, 0);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const double* row = D.S + (size_t)max(k_hi - g, 64 * m) * n;
            la[g] = row[a_];
            lb[g] = row[b_];
            lc[g] = row[c_];
          }
        };
        auto lane_read = [](double v, int l) {  // v of lane l (uniform): two v_readlane_b32, not a trip through the LDS crossbar
          const long long b = __double_as_longlong(v);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
          return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        for (int m = nch - 1; m >= 0; --m) {
          const int k_top = min(n, 64 * m + 64) - 1;
          const double ri = 64 * m + tid < n ? D.rinv[64 * m + tid] : 0.0;
          const int ca = 64 * m + tid;
          // A lone wave issues one instruction every ~9 cycles whatever it is (tools/f64_probe.hip), so a step is priced by its
          // instruction count: the whole chunk is scaled by the reciprocal pivots in one instruction (lane k's product is
          // x_k: its value no longer changes once step k is through, because the mask below keeps columns >= k out), x_k goes
          // to scalar registers and is the scalar operand of the three multiplications.  Only the chunk on the diagonal needs a
          // mask (above the diagonal S holds other data); outside the band L is an exact zero in memory, and lanes in front of
          // column 0 (chunks m - 1, m - 2 of the first chunks) carry values that are never stored.
          auto steps = [&](const double* la, const double* lb, const double* lc, int k_hi) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int k = k_hi - g;
              if (k < 64 * m) break;  // uniform
              const double xk = lane_read(xa * ri, __builtin_amdgcn_readfirstlane(k & 63));
              xa -= (ca < k ? la[g] : 0.0) * xk;
              xb -= lb[g] * xk;
              xc -= lc[g] * xk;
            }
          };
          fetch(la0, lb0, lc0, m, k_top);
          for (int k_hi = k_top; k_hi >= 64 * m; k_hi -= 2 * G) {
            fetch(la1, lb1, lc1, m, k_hi - G);
            steps(la0, lb0, lc0, k_hi);
            fetch(la0, lb0, lc0, m, k_hi - 2 * G);
            steps(la1, lb1, lc1, k_hi - G);
          }
          if (64 * m + tid < n) x[64 * m + tid] = xa * ri;
          xa = xb;
          xb = xc;
          xc = ld_y(m - 3);
        }
      }
      __syncthreads();
    } else {
    // Dense L: blocks of kFinW rows from the bottom.  Per block (a) the triangular solve inside the block and (b) one matvec for
    // the rows above.  Round 5 (the growing global bundle adjustment of real sequences: 52 key frames = 306 unknowns spent 89 us
    // per LM trial here, tools/ba_growth.py): (a) used to run on wave 0 through LDS, two wave synchronisations per unknown; now the
    // block's 24 x 24 triangle sits in wave 0's REGISTERS (lane = column, one register per row), x of the block in one register
    // per lane, and step k is a cross-lane read of x_k and one predicated multiply-subtract -- the banded branch's scheme.  (b)'s
    // operands S[k][i] do not depend on x: every thread requests them BEFORE the serial part and holds them in registers, so the
    // global round trip runs beside (a).  Every x_i still receives its subtractions one product at a time in descending k -- the
    // order of the form this replaces and of the banded branch: the same bits.
    constexpr int kFinW = 24;  // block width; the first kPanelThreads rows above a block have their operands requested ahead
    auto lane_read = [](double v, int l) {
      const long long b = __double_as_longlong(v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    // Software pipeline over the blocks: the operands of block J - 1 -- its triangle and pivots (wave 0) and the matvec rows
    // S[k][i] (every thread) -- are requested while block J is solved; none of them depends on x.  The two barriers of a block
    // order LDS traffic only (lds_barrier: __syncthreads would also wait for the loads just requested).
    auto bounds = [&](int k1, int* k0, int* bw, int* i_lo) {
      *k0 = k1 - ((k1 % kFinW) ? (k1 % kFinW) : kFinW);
      *bw = k1 - *k0;
      *i_lo = D.band > 0 ? max(*k0 - D.band, 0) : 0;
    };
    double tl[kFinW], mv[kFinW], ri = 0.0;
    auto request = [&](int k0, int bw, int i_lo, double* tl_, double* mv_, double* ri_) {
      const int i = min(i_lo + tid, n - 1), c = min(k0 + tid, n - 1);
#pragma unroll
      for (int k = 0; k < kFinW; ++k) mv_[k] = D.S[(size_t)min(k0 + k, n - 1) * n + i];
      if (tid < 64) {
#pragma unroll
        for (int k = 0; k < kFinW; ++k) tl_[k] = D.S[(size_t)min(k0 + k, n - 1) * n + c];
        *ri_ = D.rinv[min(k0 + tid, n - 1)];
      }
    };
    int k1 = n, k0, bw, i_lo;
    bounds(k1, &k0, &bw, &i_lo);
    request(k0, bw, i_lo, tl, mv, &ri);
    while (k1 > 0) {
      // (a) wave 0: the block's right-hand side into a register, then the serial steps
      if (tid < 64) {
        double xl = tid < bw ? x[k0 + tid] : 0.0;
#pragma unroll
        for (int k = kFinW - 1; k >= 0; --k) {
          if (k >= bw) continue;  // uniform (the bottom block may be narrower)
          const double xk = lane_read(xl * ri, k);  // lane k's value is final once the steps above it are through
          if (tid < k) xl -= tl[k] * xk;
        }
        if (tid < bw) x[k0 + tid] = xl * ri;
      }
      // the next block's operands: requested now, used after the two barriers below
      int k0n = 0, bwn = 0, i_lon = 0;
      double tln[kFinW], mvn[kFinW], rin = 0.0;
      if (k0 > 0) {  // uniform
        bounds(k0, &k0n, &bwn, &i_lon);
        request(k0n, bwn, i_lon, tln, mvn, &rin);
      }
      lds_barrier();
      // (b) the rows above the block: the first kPanelThreads of them from the registers, the rest the plain way
      {
        const int i = i_lo + tid;
        if (i < k0) {
          double v = x[i];
#pragma unroll
          for (int k = kFinW - 1; k >= 0; --k)
            if (k < bw) v -= mv[k] * x[k0 + k];
          x[i] = v;
        }
      }
      for (int i = i_lo + tid + kPanelThreads; i < k0; i += kPanelThreads) {
        double v = x[i];
        for (int k = k1 - 1; k >= k0; --k) v -= D.S[(size_t)k * n + i] * x[k];
        x[i] = v;
      }
      lds_barrier();
      k1 = k0;
      if (k1 > 0) {
        k0 = k0n;
        bw = bwn;
        i_lo = i_lon;
        ri = rin;
#pragma unroll
        for (int k = 0; k < kFinW; ++k) {
          tl[k] = tln[k];
          mv[k] = mvn[k];
        }
      }
    }
    }
    for (int i = tid; i < n; i += kPanelThreads) D.xp[i] = x[i];
  }
  __syncthreads();
  const int cur = st->cur;
  const double* c0 = D.cam[cur];
  double* c1 = D.cam[cur ^ 1];
  for (int p = tid; p < D.n_poses; p += kPanelThreads) {
    const double* src = c0 + (size_t)p * kCamStride;
    double* dst = c1 + (size_t)p * kCamStride;
    const int cs = D.pose_slot[p];
    if (cs < 0 || !ok) {
      for (int k = 0; k < kCamStride; ++k) dst[k] = src[k];
    } else {
      const double* d = x + 6 * cs;
      double t[3] = {src[0] + d[0], src[1] + d[1], src[2] + d[2]};
      const double bx = d[3], by = d[4], bz = d[5];
      const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));
      const double ax = src[3], ay = src[4], az = src[5], aw = src[6];
      const double w = aw * bw - ax * bx - ay * by - az * bz;
      const double xx = aw * bx + ax * bw + ay * bz - az * by;
      const double yy = aw * by + ay * bw + az * bx - ax * bz;
      const double zz = aw * bz + az * bw + ax * by - ay * bx;
      const double nrm = sqrt(xx * xx + yy * yy + zz * zz + w * w);
      double q[4] = {xx / nrm, yy / nrm, zz / nrm, w / nrm};
      for (int k = 0; k < 3; ++k) dst[k] = t[k];
      for (int k = 0; k < 4; ++k) dst[3 + k] = q[k];
      quat_to_w2n(t, q, dst + 7);
    }
  }
  // gain denominator x^T (lambda x + b): one wave sums it (lane l takes the terms l, l + 64, ...; butterfly over the wave), as
  // ba_solve_block does -- on one lane the 594 terms of the scaled run were 13 us of dependent instructions
  if (tid < 64) {
    const double lambda = st->lambda;
    const double* const bp = lin_of(D, cur).bp;
    double sc = 0.0;
    if (ok)
      for (int j = tid; j < n; j += 64) sc += x[j] * (lambda * x[j] + bp[j]);
    sc = vs_group_reduce<6>(sc);
    if (tid == 0) {
      st->scale_pose = sc;
      st->solve_ok = ok;
      st->trials += 1;
      if (!ok) st->not_pd += 1;
    }
  }
}

// OptimizationAlgorithmLevenberg::solve's accept/reject logic and SparseOptimizer::optimize's loop control
// Runs on ONE wave (threadIdx.x < 64) of the workgroup of ba_point_trial that finished last: no launch of its own.
__device__ inline void ba_decide(const ba_dev& D) {
  // the record is read once, updated in registers and written back once: field-by-field read-modify-writes of global
  // memory were a chain of dependent round trips at the very end of every trial (2.5 us by in-kernel stamps)
  lm_state s = *D.st;
  if (s.done) return;
  const double psum = wave_sum_partials(D.part_chi, D.nb_pt), ssum = wave_sum_partials(D.part_scale, D.nb_pt);
  if (threadIdx.x != 0) return;
  double temp = 0.0, scale = s.scale_pose;
  if (s.solve_ok) {
    temp = psum;
    scale += ssum;
    if (D.n_scale) temp += scale_edges_chi(D, D.cam[s.cur ^ 1]);
  } else {
    temp = 1.7976931348623157e308;
  }
  s.temp_chi = temp;
  double rho = s.current_chi - temp;
  scale += 1e-3;
  rho /= scale;
  s.rho = rho;
  if (D.trial_trace && s.trials >= 1 && s.trials <= D.trial_cap) {
    double* row = D.trial_trace + 4 * (size_t)(s.trials - 1);
    row[0] = s.lambda;
    row[1] = temp;
    row[2] = rho;
    row[3] = s.solve_ok ? 1.0 : 0.0;
  }
  int stop = 0;
  if (rho > 0 && isfinite(temp)) {
    const double g = 2 * rho - 1;
    double alpha = 1.0 - g * g * g;  // as mo_decide
    alpha = fmin(alpha, 2.0 / 3.0);
    const double f = fmax(1.0 / 3.0, alpha);
    s.lambda *= f;
    s.ni = 2.0;
    s.current_chi = temp;
    s.cur ^= 1;  // accept: the trial buffer becomes the estimate
    s.accepted += 1;
  } else {
    s.lambda *= s.ni;
    s.ni *= 2;
    if (!isfinite(s.lambda)) stop = 1;
  }
  s.qmax += 1;
  if (!stop && rho < 0 && s.qmax < 10) {
    s.need_lin = 0;  // retry with the same linearisation
  } else {
    // the outer iteration is over
    if (D.chi_trace) D.chi_trace[s.it] = s.current_chi;
    if (D.lambda_trace) D.lambda_trace[s.it] = s.lambda;
    s.it += 1;
    if (s.qmax == 10 || rho == 0 || stop) {
      s.done = 1;
      s.terminated = 1;
    } else if (s.it >= D.max_it) {
      s.done = 1;
    } else {
      s.need_lin = 1;
      s.qmax = 0;
    }
  }
  *D.st = s;
}

// ------------------------------------------------------------------------------------------------ trial + chi2
__global__ __launch_bounds__(kPtThreads) void ba_point_trial(ba_dev D) {
  __shared__ double s_red[2 * kPtThreads / 64];
  const int tid = threadIdx.x;
  const int a = blockIdx.x * kPtPerBlock + tid / kPtLanes, sub = tid % kPtLanes;  // kPtLanes lanes per point, as above
  // the point's index records are requested before the LM state: their addresses do not depend on it
  int p = 0, ls = -1, o0 = 0, o1 = 0;
  if (a < D.n_act) {
    p = D.act_pt[a];
    o0 = D.pt_start[a];
    o1 = D.pt_start[a + 1];
    ls = D.pt_slot[p];
  }
  const lm_state st = *D.st;
  if (st.done) return;
  double chi = 0.0, sc = 0.0;
  if (st.solve_ok && a < D.n_act) {
    const double* cams1 = D.cam[st.cur ^ 1];
    const double* pts0 = D.pts[st.cur];
    double* pts1 = D.pts[st.cur ^ 1];
    const lin_view L = lin_of(D, st.cur);
    double X[3] = {pts0[3 * (size_t)p], pts0[3 * (size_t)p + 1], pts0[3 * (size_t)p + 2]};
    if (ls >= 0) {
      // back substitution: cl = bl - sum_i Hpl_i^T dx_cam(i); the sum is split over the lanes and shuffle-reduced
      double part[3] = {0.0, 0.0, 0.0};
      for (int o = o0 + sub; o < o1; o += kPtLanes) {
        const int cs = D.pose_slot[D.o_cam[o]];
        if (cs < 0) continue;
        const double* B = L.Hpl + 18 * (size_t)D.o_hpl[o];
        const double* xc = D.xp + 6 * cs;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int k = 0; k < 6; ++k) part[b] += B[3 * k + b] * xc[k];
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) part[b] = vs_group_reduce<kPtLaneSteps>(part[b]);
      const double cl[3] = {L.bl[3 * (size_t)ls] - part[0], L.bl[3 * (size_t)ls + 1] - part[1], L.bl[3 * (size_t)ls + 2] - part[2]};
      const double* Di = D.Dinv + 9 * (size_t)ls;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double xl = Di[3 * k] * cl[0] + Di[3 * k + 1] * cl[1] + Di[3 * k + 2] * cl[2];
        if (sub == 0) sc += xl * (st.lambda * xl + L.bl[3 * (size_t)ls + k]);
        X[k] += xl;
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) pts1[3 * (size_t)p + k] = X[k];
    }
    if (D.spec && st.it + 1 < D.max_it) {
      // the trial state's point role of the linearisation comes with its chi2 (same code, lanes and operands as
      // ba_linearize would use after an accepted step): written to the other linearisation, dropped if the step is rejected.
      // Not in the last outer iteration: accepted or not, nothing is linearised after it (the chi2 below is the same sum).
      double maxd = 0.0;
      chi = linearize_point(D, lin_of(D, st.cur ^ 1), cams1, X, a, ls, sub, maxd);
    } else {
      for (int o = o0 + sub; o < o1; o += kPtLanes) {
        edge_t E;
        eval_edge<false>(D, cams1 + (size_t)D.o_cam[o] * kCamStride, X, D.o_uv + 2 * (size_t)o, D.has_info ? D.o_info + 3 * (size_t)o : nullptr, E);
        chi += E.rho0;
      }
    }
  }
  double csum = chi, ssum = sc;
  block_reduce2<kPtThreads, false>(csum, ssum, s_red);
  // The workgroup that publishes its partials last takes the LM decision (ba_decide) in the same launch.  Hand-off as in
  // vs_match.hip: write-through stores (agent-scope relaxed atomic stores), drain, one relaxed agent-scope ticket add;
  // the last arriver acquires (its CU's L1 must not serve older partials) and resets the ticket for the next slot.
  __shared__ int s_last;
  if (tid == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(D.part_chi + blockIdx.x), (unsigned long long)__double_as_longlong(csum),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(D.part_scale + blockIdx.x), (unsigned long long)__double_as_longlong(ssum),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(D.trial_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(D.trial_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last && tid < 64) ba_decide(D);
}

// ------------------------------------------------------------------------------------------------ result export
// Gathers what the host reads after a batch of slots -- LM state, the accepted cameras and points, the traces -- into one
// contiguous block, so that the poll for `done` and the read-back are ONE device-to-host copy and one synchronisation.
// Layout in doubles: [state 32 | cameras F x kCamStride | points 3 P | chi2 trace | lambda trace | trial rows 4 x cap].
__global__ __launch_bounds__(256) void ba_export(ba_dev D, double* out) {
  const lm_state st = *D.st;
  const size_t nc = (size_t)D.n_poses * kCamStride, npt = 3 * (size_t)D.n_points, ntr = (size_t)D.max_it, ntt = 4 * (size_t)D.trial_cap;
  const size_t total = 32 + nc + npt + 2 * ntr + ntt;
  const double* cam = D.cam[st.cur];
  const double* pts = D.pts[st.cur];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    double v;
    if (i < 32) {
      v = i < sizeof(lm_state) / sizeof(double) ? reinterpret_cast<const double*>(D.st)[i] : 0.0;
    } else if (i < 32 + nc) {
      v = cam[i - 32];
    } else if (i < 32 + nc + npt) {
      v = pts[i - 32 - nc];
    } else if (i < 32 + nc + npt + ntr) {
      v = D.chi_trace[i - 32 - nc - npt];
    } else if (i < 32 + nc + npt + 2 * ntr) {
      v = D.lambda_trace[i - 32 - nc - npt - ntr];
    } else {
      v = D.trial_trace[i - 32 - nc - npt - 2 * ntr];
    }
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ motion-only BA
// motionOnlyBundleAdjustement (reference LocalBA.py:195-229) has no free points and no scale edges: the normal
// equations are block diagonal, so linearisation, the 6x6 solve, the trial state and the trial chi2 of one camera need
// nothing from any other camera.  The only global coupling of g2o's LM is the accept/reject decision (sums of chi2 and
// of the gain denominator over all cameras) and lambda_0 (max diagonal).
// => ONE kernel launch per LM trial, one workgroup per free camera, spread over the CUs, and the decision on the
// previous trial is recomputed redundantly in the prologue of the next launch from the per-camera partials (the same
// inputs and code in every workgroup give the same answer; workgroup 0 records it).  No grid barrier, no host round
// trip: the host enqueues 1 + max_iterations launches and polls `done` once.
//   step 0           : linearise only (lambda_0 needs the maximum over ALL cameras)
//   step s >= 1      : decide(step s-1) -> [re-linearise if a step was accepted] -> solve -> trial state -> trial chi2
// State and partials are double buffered by step parity; observations are read from a camera-major copy
// (mo_X / mo_uv / mo_info) so every load is coalesced and index-free.


// Wave-wide sum (or maximum of non-negative values) in a fixed order, result in every lane: four row_shr steps inside each row
// of 16 lanes, row_bcast:15 / row_bcast:31 across the rows (data-parallel-primitive moves: ~20 cycles per step; the same
// reduction as an xor-butterfly of __shfl_xor is twelve ds_bpermute round trips, ~0.3 us on a lone wave), readlane 63.
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ double mo_dpp_acc(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);  // lanes without a source add 0.0
  const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
  const double o = __hiloint2double(ohi, olo);
  return MAX ? fmax(v, o) : v + o;
}
template <bool MAX>
__device__ __forceinline__ double mo_wave_reduce(double v) {
  v = mo_dpp_acc<0x111, 0xF, MAX>(v);  // row_shr:1
  v = mo_dpp_acc<0x112, 0xF, MAX>(v);  // row_shr:2
  v = mo_dpp_acc<0x114, 0xF, MAX>(v);  // row_shr:4
  v = mo_dpp_acc<0x118, 0xF, MAX>(v);  // row_shr:8   -> lane 15 of every row holds the row's value
  v = mo_dpp_acc<0x142, 0xA, MAX>(v);  // row_bcast:15 into rows 1 and 3
  v = mo_dpp_acc<0x143, 0xC, MAX>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's value
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// The LM bookkeeping of the motion-only path, run on identical inputs by EVERY thread of every camera's workgroup (in registers:
// no broadcast of the outcome, no barrier): st is
// the state after the launch / step that produced the sums (chi2; max diagonal or gain denominator; cameras whose 6x6
// system was not positive definite).  Camera 0's workgroup records the traces.
__device__ inline void mo_decide(const ba_dev& D, mo_state& st, double chi, double second, double bad, bool writer) {
  if (st.stage == 1) {  // iteration 0 has just been linearised: computeLambdaInit + the first chi2
    st.lambda = 1e-5 * second;
    st.ni = 2.0;
    st.current_chi = chi;
    st.chi0 = chi;
    st.need_lin = 0;
  } else {  // a trial has just been evaluated
    double temp = chi, scale = second;
    st.trials += 1;
    if (bad > 0.0) {
      temp = 1.7976931348623157e308;
      st.not_pd += 1;
    }
    double rho = st.current_chi - temp;
    scale += 1e-3;
    rho /= scale;
    if (writer && D.trial_trace && st.trials <= D.trial_cap) {
      double* row = D.trial_trace + 4 * (size_t)(st.trials - 1);
      row[0] = st.lambda;
      row[1] = temp;
      row[2] = rho;
      row[3] = bad > 0.0 ? 0.0 : 1.0;
    }
    int stop = 0;
    if (rho > 0 && isfinite(temp)) {
      const double g = 2 * rho - 1;
      double alpha = 1.0 - g * g * g;  // the cube by two multiplications: the library pow() was 0.85 us of every accepted step
      alpha = fmin(alpha, 2.0 / 3.0);
      st.lambda *= fmax(1.0 / 3.0, alpha);
      st.ni = 2.0;
      st.current_chi = temp;
      st.cur ^= 1;  // accept: the trial buffer becomes the estimate
    } else {
      st.lambda *= st.ni;
      st.ni *= 2;
      if (!isfinite(st.lambda)) stop = 1;
    }
    st.qmax += 1;
    if (!stop && rho < 0 && st.qmax < 10) {
      st.need_lin = 0;  // retry with the same linearisation
    } else {
      if (writer) {
        if (D.chi_trace) D.chi_trace[st.it] = st.current_chi;
        if (D.lambda_trace) D.lambda_trace[st.it] = st.lambda;
      }
      st.it += 1;
      if (st.qmax == 10 || rho == 0 || stop) {
        st.done = 1;
        st.terminated = 1;
      } else if (st.it >= D.max_it) {
        st.done = 1;
      } else {
        st.need_lin = 1;
        st.qmax = 0;
      }
    }
  }
}

// sums 28 doubles per thread over the workgroup in a fixed order; result in s_out[0..27].  The upper half of the threads
// hands its values to the lower half through LDS; the kMoRows rows are then transposed-reduced: (value k, group g)
// threads add the 32 rows of their group in row order and 28 threads add the group sums in group order -- four barriers
// instead of the ten of a binary tree over 28-vectors.
constexpr int kMoRows = kMoThreads / 2, kRedGroups = kMoRows / 32;
__device__ inline void block_reduce28(double (&acc)[28], double (*s_all)[29], double (*s_grp)[28], double* s_out, int tid) {
  if (tid >= kMoRows) {
#pragma unroll
    for (int k = 0; k < 28; ++k) s_all[tid - kMoRows][k] = acc[k];
  }
  __syncthreads();
  if (tid < kMoRows) {
#pragma unroll
    for (int k = 0; k < 28; ++k) s_all[tid][k] += acc[k];
  }
  __syncthreads();
  {
    // wave w < kRedGroups / 2 handles the groups 2w and 2w+1 with its lanes 0..55 (value k = lane % 28)
    const int lane = tid & 63, wv = tid >> 6;
    if (wv < kRedGroups / 2 && lane < 56) {
      const int k = lane % 28, g = 2 * wv + lane / 28;
      double a = 0.0;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) a += s_all[g * 32 + j][k];
      s_grp[g][k] = a;
    }
  }
  __syncthreads();
  if (tid < 28) {
    double a = s_grp[0][tid];
#pragma unroll
    for (int g = 1; g < kRedGroups; ++g) a += s_grp[g][tid];
    s_out[tid] = a;
  }
  __syncthreads();
}

}  // namespace

namespace vsba {
__global__ __launch_bounds__(kMoThreads) void ba_motion_step(ba_dev D, int step) {
  // this launch is a chain of short dependent phases (latency-bound): fused multiply-adds in its own accumulations and
  // in the 6x6 solve shorten the chain; the edge evaluation (eval_edge) keeps the file's operation-for-operation rounding
#pragma clang fp contract(fast)
  __shared__ double s_all[kMoRows][29];
  __shared__ double s_grp[kRedGroups][28];
  __shared__ double s_sum[28];
  __shared__ double s_part[1][kMoThreads / 64];  // the waves' chi2 totals
  __shared__ mo_state s_st;
  __shared__ double s_x[6];
  __shared__ int s_ok;
  const int tid = threadIdx.x, c = blockIdx.x, nfp = D.nfp;
  mo_state* g_state = reinterpret_cast<mo_state*>(D.st);
  const double* prev_part = D.mo_part + (size_t)((step + 1) & 1) * 4 * nfp;
  double* my_part = D.mo_part + (size_t)(step & 1) * 4 * nfp;

  // ---- everything whose ADDRESS does not depend on the decision is requested first, so that the launch pays one
  // HBM/L2 round trip instead of four dependent ones (caches are cold after every kernel boundary): this thread's
  // first observation, the stored normal equations, both state buffers' record of this camera
  const int pose = D.slot_pose[c];
  const int o0 = D.cam_start[c], o1 = D.cam_start[c + 1];
  double* gH = D.mo_H + (size_t)c * 42;
  double preX[3] = {0, 0, 0}, preUV[2] = {0, 0}, preW[3] = {1, 0, 1};
  const int i_first = o0 + tid;
  if (i_first < o1) {
    preX[0] = D.mo_X[3 * (size_t)i_first];
    preX[1] = D.mo_X[3 * (size_t)i_first + 1];
    preX[2] = D.mo_X[3 * (size_t)i_first + 2];
    preUV[0] = D.mo_uv[2 * (size_t)i_first];
    preUV[1] = D.mo_uv[2 * (size_t)i_first + 1];
    if (D.has_info) {
      preW[0] = D.mo_info[3 * (size_t)i_first];
      preW[1] = D.mo_info[3 * (size_t)i_first + 1];
      preW[2] = D.mo_info[3 * (size_t)i_first + 2];
    }
  }
  const double pre_gh = tid < 27 ? gH[tid] : 0.0;
  const double pre_cam0 = tid < kCamStride ? D.cam[0][(size_t)pose * kCamStride + tid] : 0.0;
  const double pre_cam1 = tid < kCamStride ? D.cam[1][(size_t)pose * kCamStride + tid] : 0.0;
  // ... and the previous launch's per-camera partials (lane l of every wave: camera l), which the decision below sums
  const int lane = tid & 63;
  double pre_part[4] = {0.0, 0.0, 0.0, 0.0};
  if (lane < nfp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pre_part[k] = prev_part[4 * (size_t)lane + k];
  }

  // ---- prologue: every THREAD derives the current LM state from the previous launch's state + partials: lane l of every
  // wave sums cameras l, l + 64, ... in that order, the wave reduces in a fixed order (mo_wave_reduce), every lane runs the
  // decision on the same sums.  (One thread summing all cameras and taking the decision for the others cost two barriers
  // and, in the one-launch form where this block sits on the critical path of every step, 0.9 - 1.7 us.)
  if (tid == 0) s_st = g_state[(step + 1) & 1];
  __syncthreads();
  mo_state st = s_st;
  const int done_on_entry = st.done;
  if (st.stage != 0 && !st.done) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int k = lane; k < nfp; k += 64) {
      const bool pre = k == lane;  // the first pass was requested at the top of the launch
      a0 += pre ? pre_part[0] : prev_part[4 * k];
      if (st.stage == 1) a1 = fmax(a1, pre ? pre_part[3] : prev_part[4 * k + 3]);
      else a1 += pre ? pre_part[1] : prev_part[4 * k + 1];
      a2 += pre ? pre_part[2] : prev_part[4 * k + 2];  // number of cameras whose 6x6 system was not positive definite
    }
    const double chi = mo_wave_reduce<false>(a0);
    const double second = st.stage == 1 ? mo_wave_reduce<true>(a1) : mo_wave_reduce<false>(a1);
    const double bad = mo_wave_reduce<false>(a2);
    mo_decide(D, st, chi, second, bad, c == 0 && tid == 0);
  }
  if (!done_on_entry) st.seq = step;  // ends as the index of the launch whose prologue found the solve finished
  if (st.done) {
    if (c == 0 && tid == 0) g_state[step & 1] = st;
    return;
  }

  // ---- this camera (its current record goes to LDS from the preloaded registers)
  __shared__ double s_cam[kCamStride];
  if (tid < kCamStride) s_cam[tid] = st.cur ? pre_cam1 : pre_cam0;
  __syncthreads();
  const double* cam = s_cam;
  const double* preInfo = D.has_info ? preW : nullptr;
  const bool lin_only = st.need_lin && st.it == 0 && st.stage == 0;
  if (st.need_lin) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for (int i = i_first; i < o1; i += kMoThreads) {
      edge_t E;
      if (i == i_first) {
        eval_edge<true>(D, cam, preX, preUV, preInfo, E);
      } else {
        const double X[3] = {D.mo_X[3 * (size_t)i], D.mo_X[3 * (size_t)i + 1], D.mo_X[3 * (size_t)i + 2]};
        eval_edge<true>(D, cam, X, D.mo_uv + 2 * (size_t)i, D.has_info ? D.mo_info + 3 * (size_t)i : nullptr, E);
      }
      const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
      const double r0 = -We0 * E.rho1, r1 = -We1 * E.rho1;
      const double w0 = E.rho1 * E.W[0], w1 = E.rho1 * E.W[1], w2 = E.rho1 * E.W[2];
      double WJ[2][6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        WJ[0][k] = w0 * E.Jj[0][k] + w1 * E.Jj[1][k];
        WJ[1][k] = w1 * E.Jj[0][k] + w2 * E.Jj[1][k];
      }
      int n = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = k; l < 6; ++l) acc[n++] += E.Jj[0][k] * WJ[0][l] + E.Jj[1][k] * WJ[1][l];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[21 + k] += E.Jj[0][k] * r0 + E.Jj[1][k] * r1;
      acc[27] += E.rho0;
    }
    block_reduce28(acc, s_all, s_grp, s_sum, tid);
    if (tid < 27) gH[tid] = s_sum[tid];  // upper triangle (21) + b (6), re-read by retries of this linearisation
    if (lin_only) {
      if (tid == 0) {
        double mx = 0.0;
        int n = 0;
        for (int k = 0; k < 6; ++k) {
          mx = fmax(mx, fabs(s_sum[n]));
          n += 6 - k;
        }
        my_part[4 * c] = s_sum[27];
        my_part[4 * c + 1] = 0.0;
        my_part[4 * c + 2] = 0.0;
        my_part[4 * c + 3] = mx;
        if (c == 0) {
          st.stage = 1;
          g_state[step & 1] = st;
        }
      }
      return;
    }
  } else {
    if (tid < 27) s_sum[tid] = pre_gh;
    __syncthreads();
  }

  // ---- 6x6 solve (H + lambda I) x = b and SBACam::update into the trial buffer: one thread
  double* trial = D.cam[st.cur ^ 1] + (size_t)pose * kCamStride;
  if (tid == 0) {
    double A[6][6], x[6];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int l = k; l < 6; ++l) {
        A[k][l] = s_sum[n];
        A[l][k] = s_sum[n];
        ++n;
      }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      A[k][k] += st.lambda;
      x[k] = s_sum[21 + k];
    }
    int ok = 1;
    double rinv[6];  // 1 / L[j][j] from v_rsq_f64 + Newton (<= 2 ulp from sqrt / divide): no IEEE sequences in the chain
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double d = A[j][j];
      if (!(d > 0.0)) ok = 0;
      const double ri = vs_fast_rsq(d);
      A[j][j] = d * ri;
      rinv[j] = ri;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) A[i][j] = A[i][j] * rinv[j];
#pragma unroll
      for (int i = j + 1; i < 6; ++i)
#pragma unroll
        for (int k = j + 1; k <= i; ++k) A[i][k] -= A[i][j] * A[k][j];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      x[k] = x[k] * rinv[k];
#pragma unroll
      for (int i = k + 1; i < 6; ++i) x[i] -= A[i][k] * x[k];
    }
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      x[k] = x[k] * rinv[k];
#pragma unroll
      for (int i = 0; i < k; ++i) x[i] -= A[k][i] * x[k];
    }
    double sc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) sc += x[k] * (st.lambda * x[k] + s_sum[21 + k]);
    double t[3] = {cam[0] + x[0], cam[1] + x[1], cam[2] + x[2]};
    const double bx = x[3], by = x[4], bz = x[5];
    const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));
    const double ax = cam[3], ay = cam[4], az = cam[5], aw = cam[6];
    const double w = aw * bw - ax * bx - ay * by - az * bz;
    const double xx = aw * bx + ax * bw + ay * bz - az * by;
    const double yy = aw * by + ay * bw + az * bx - ax * bz;
    const double zz = aw * bz + az * bw + ax * by - ay * bx;
    const double inrm = vs_fast_rsq(xx * xx + yy * yy + zz * zz + w * w);
    double q[4] = {xx * inrm, yy * inrm, zz * inrm, w * inrm};
    double rec[kCamStride];
    for (int k = 0; k < 3; ++k) rec[k] = t[k];
    for (int k = 0; k < 4; ++k) rec[3 + k] = q[k];
    quat_to_w2n(t, q, rec + 7);
    for (int k = 0; k < kCamStride; ++k) {
      trial[k] = rec[k];
      s_all[0][k] = rec[k];  // the workgroup evaluates the trial from LDS
    }
    s_x[0] = sc;
    s_ok = ok;
  }
  __syncthreads();
  // ---- robust chi2 of this camera's trial state
  double tcam[kCamStride];
#pragma unroll
  for (int k = 0; k < kCamStride; ++k) tcam[k] = s_all[0][k];
  const double scl = s_x[0];
  const int ok = s_ok;
  __syncthreads();
  double chi = 0.0;
  for (int i = i_first; i < o1; i += kMoThreads) {
    edge_t E;
    if (i == i_first) {
      eval_edge<false>(D, tcam, preX, preUV, preInfo, E);
    } else {
      const double X[3] = {D.mo_X[3 * (size_t)i], D.mo_X[3 * (size_t)i + 1], D.mo_X[3 * (size_t)i + 2]};
      eval_edge<false>(D, tcam, X, D.mo_uv + 2 * (size_t)i, D.has_info ? D.mo_info + 3 * (size_t)i : nullptr, E);
    }
    chi += E.rho0;
  }
  // workgroup sum of chi: fixed-order reduction inside each wave (mo_wave_reduce), then the wave totals in wave order
  chi = mo_wave_reduce<false>(chi);
  if ((tid & 63) == 0) s_part[0][tid >> 6] = chi;
  __syncthreads();
  if (tid == 0) {
    double tot = s_part[0][0];
    for (int wv = 1; wv < kMoThreads / 64; ++wv) tot += s_part[0][wv];
    my_part[4 * c] = tot;
    my_part[4 * c + 1] = scl;
    my_part[4 * c + 2] = ok ? 0.0 : 1.0;
    my_part[4 * c + 3] = 0.0;
    if (c == 0) {
      st.stage = 2;
      g_state[step & 1] = st;
    }
  }
}


// ---- the same solve as ONE launch.  Between two launches of ba_motion_step lie a kernel boundary (~1.7 us) and a round
// trip to HBM for everything the next step needs (caches are cold after a boundary: ~2.5 us before the first operand
// arrives) -- about 4 of the 7.4 us of a step.  For windows of up to kMoMaxPersist cameras (all workgroups co-resident
// with room to spare) the steps run inside one launch: a thread keeps its observations in registers, the camera record
// and the normal equations stay in LDS, and the workgroups exchange their per-step partials through tagged mailboxes
// (post_partials / the rendezvous at the top of the step loop: write-through stores on one side, agent-scope polling
// loads on the other, no ticket).  Mailboxes are double buffered by step parity, as the partials of the multi-launch form:
// a workgroup can be at most one step ahead of the slowest.  The arithmetic -- operands, order,
// reductions, the decision (mo_decide) -- is that of ba_motion_step, the results are bit-identical (tested).
// Every wait is bounded: a workgroup that does not see the others within 2^16 polls (~0.1 s) ends the solve with terminated = 3.
constexpr int kMoObsRegs = 2;      // observations a thread keeps in registers (kMoThreads * kMoObsRegs per camera)
static_assert(kMoObsRegs * kMoThreads == kMoPersistObs, "register capacity of ba_motion_persistent");

// OVF: cameras with more observations than the registers hold re-read the rest from memory in every step (same order of
// accumulation); a separate instantiation, so that the common one keeps its schedule (folding the extra loops into it
// cost the ICL-NUIM tracking path 6 %).
template <bool OVF>
__global__ __launch_bounds__(kMoThreads) void ba_motion_persistent(ba_dev D, int max_steps) {
#pragma clang fp contract(fast)
  __builtin_amdgcn_s_setprio(3);  // the critical path of pipelined tracking; the other stream's kernels are throughput work
  __shared__ double s_all[kMoRows][29];
  __shared__ double s_grp[kRedGroups][28];
  __shared__ double s_sum[28];
  __shared__ double s_part[1][kMoThreads / 64];  // the waves' chi2 totals
  __shared__ mo_state s_st;
  __shared__ double s_x[6];
  __shared__ double s_cam[kCamStride], s_trial[kCamStride];
  __shared__ int s_ok, s_abort;
  __shared__ unsigned s_box[kMoPersistCameras][8];  // payload halves of every camera's four partials
  __shared__ double s_pv[4];
  const int tid = threadIdx.x, c = blockIdx.x, nfp = D.nfp;
  // diagnostic (vs_mo_profile): thread 0 of camera 0's workgroup stamps the phases of every step with the shader clock
  // [0] step entered, [1] everybody's partials arrived, [2] decision taken, [3] linearised + reduced (0: linearisation kept),
  // [4] 6x6 system solved + trial record written, [5] trial chi2 evaluated and summed, [6] partials posted, [7] wall clock
  unsigned long long* const stamp_row = (D.mo_stamps && c == 0 && tid == 0) ? D.mo_stamps : nullptr;
#define VS_MO_STAMP(col) do { if (stamp_row && step < 64) stamp_row[(size_t)step * 8 + (col)] = (col) == 7 ? wall_clock64() : (unsigned long long)__builtin_readcyclecounter(); } while (0)
  mo_state* g_state = reinterpret_cast<mo_state*>(D.st);
  const int pose = D.slot_pose[c];
  const int o0 = D.cam_start[c], o1 = D.cam_start[c + 1];
  // this thread's observations, kept for the whole solve
  double oX[kMoObsRegs][3], oUV[kMoObsRegs][2], oW[kMoObsRegs][3];
#pragma unroll
  for (int j = 0; j < kMoObsRegs; ++j) {
    const int i = o0 + tid + j * kMoThreads;
    oX[j][0] = oX[j][1] = oX[j][2] = oUV[j][0] = oUV[j][1] = 0.0;
    oW[j][0] = oW[j][2] = 1.0;
    oW[j][1] = 0.0;
    if (i < o1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) oX[j][k] = D.mo_X[3 * (size_t)i + k];
      oUV[j][0] = D.mo_uv[2 * (size_t)i];
      oUV[j][1] = D.mo_uv[2 * (size_t)i + 1];
      if (D.has_info) {
#pragma unroll
        for (int k = 0; k < 3; ++k) oW[j][k] = D.mo_info[3 * (size_t)i + k];
      }
    }
  }
  if (tid == 0) {
    s_st = g_state[1];  // the initial record (need_lin = 1, ni = 2, cur)
    s_abort = 0;
  }
  __syncthreads();
  mo_state st = s_st;
  if (tid < kCamStride) s_cam[tid] = D.cam[st.cur][(size_t)pose * kCamStride + tid];
  __syncthreads();
  // This camera's four partials of a step go out as eight 64-bit words, each carrying 32 payload bits and the tag
  // (solve epoch, step + 1) in its low half: every word validates itself, so the readers need neither a ticket nor an
  // acknowledged store -- one write-through store instruction of eight lanes, no wait.  Wave 0 only; s_pv from thread 0.
  auto post_partials = [&](int at_step) {
    if (tid < 64) {
      wave_lds_sync();
      if (tid < 8) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s_pv[tid >> 1]);
        const unsigned half = (tid & 1) ? (unsigned)(bits & 0xFFFFFFFFull) : (unsigned)(bits >> 32);
        const unsigned tag = (D.mo_epoch << 12) | (unsigned)(at_step + 1);
        __hip_atomic_store(D.mo_box + ((size_t)(at_step & 1) * kMoPersistCameras + c) * 8 + tid, ((unsigned long long)half << 32) | tag,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  // the normal equations of this camera at the record `camp`: upper triangle (21) + b (6) + chi2 into out[28]
  auto linearise = [&](const double* camp, double* out) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
#pragma unroll
    for (int j = 0; j < kMoObsRegs; ++j) {
      if (o0 + tid + j * kMoThreads >= o1) continue;
      edge_t E;
      eval_edge<true>(D, camp, oX[j], oUV[j], D.has_info ? oW[j] : nullptr, E);
      const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
      const double r0 = -We0 * E.rho1, r1 = -We1 * E.rho1;
      const double w0 = E.rho1 * E.W[0], w1 = E.rho1 * E.W[1], w2 = E.rho1 * E.W[2];
      double WJ[2][6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        WJ[0][k] = w0 * E.Jj[0][k] + w1 * E.Jj[1][k];
        WJ[1][k] = w1 * E.Jj[0][k] + w2 * E.Jj[1][k];
      }
      int n = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = k; l < 6; ++l) acc[n++] += E.Jj[0][k] * WJ[0][l] + E.Jj[1][k] * WJ[1][l];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[21 + k] += E.Jj[0][k] * r0 + E.Jj[1][k] * r1;
      acc[27] += E.rho0;
    }
    if constexpr (OVF) {
      for (int i = o0 + tid + kMoObsRegs * kMoThreads; i < o1; i += kMoThreads) {
        const double X[3] = {D.mo_X[3 * (size_t)i], D.mo_X[3 * (size_t)i + 1], D.mo_X[3 * (size_t)i + 2]};
        edge_t E;
        eval_edge<true>(D, camp, X, D.mo_uv + 2 * (size_t)i, D.has_info ? D.mo_info + 3 * (size_t)i : nullptr, E);
        const double We0 = E.W[0] * E.e[0] + E.W[1] * E.e[1], We1 = E.W[1] * E.e[0] + E.W[2] * E.e[1];
        const double r0 = -We0 * E.rho1, r1 = -We1 * E.rho1;
        const double w0 = E.rho1 * E.W[0], w1 = E.rho1 * E.W[1], w2 = E.rho1 * E.W[2];
        double WJ[2][6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          WJ[0][k] = w0 * E.Jj[0][k] + w1 * E.Jj[1][k];
          WJ[1][k] = w1 * E.Jj[0][k] + w2 * E.Jj[1][k];
        }
        int n = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int l = k; l < 6; ++l) acc[n++] += E.Jj[0][k] * WJ[0][l] + E.Jj[1][k] * WJ[1][l];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[21 + k] += E.Jj[0][k] * r0 + E.Jj[1][k] * r1;
        acc[27] += E.rho0;
      }
    }
    block_reduce28(acc, s_all, s_grp, out, tid);
  };
  // Speculation (round 4): while the other cameras' partials of a trial are on their way (the rendezvous at the top of the
  // next step: ~1.2 us of waiting), this workgroup already linearises at its TRIAL record.  If the trial is accepted that IS the
  // next linearisation (same record, same code: same bits) and the step goes straight to the 6x6 solve; if it is rejected the
  // sums are dropped.  Done only while trials are being accepted (the first trial, and any trial after an accepted one): the
  // rejected trials at the end of a solve -- LM retries up to ten times before it gives up -- cost nothing extra.
  __shared__ double s_spec[28];
  bool speculate = true, have_spec = false, used_spec = false;
  int step = 0;
  for (;; ++step) {
    VS_MO_STAMP(0);
    VS_MO_STAMP(7);
    if (step > 0) {
      // ---- rendezvous: everybody's partials of step - 1, then the decision (every workgroup, identical inputs).
      // Eight lanes per camera poll the eight tagged words of its mailbox (see the end of the step) until every word of
      // every camera carries this step's tag; the payload halves go through LDS.
      {
        static_assert(kMoThreads / 8 == kMoPersistCameras, "eight polling lanes per camera mailbox: thread tid >> 3 polls camera tid >> 3");
        const int g = tid >> 3, jw = tid & 7;
        const unsigned tag = (D.mo_epoch << 12) | (unsigned)step;
        const unsigned long long* word = D.mo_box + ((size_t)((step + 1) & 1) * kMoPersistCameras + g) * 8 + jw;
        int polls = 0;
        for (;;) {
          unsigned long long w = 0ull;
          bool ready = true;
          if (g < nfp) {
            w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ready = (unsigned)(w & 0xFFFFFFFFull) == tag;
            s_box[g][jw] = (unsigned)(w >> 32);
          }
          if (__syncthreads_and(ready)) break;
          if (++polls > (1 << 16)) {  // (~0.1 s; a rendezvous takes ~1 us) uniform: every thread counts the same rounds
            s_abort = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      VS_MO_STAMP(1);
      if (s_abort) {
        st.done = 1;
        st.terminated = 3;
        break;
      }
      auto part_at = [&](int cam_k, int which) {
        return __longlong_as_double((long long)(((unsigned long long)s_box[cam_k][2 * which] << 32) | (unsigned long long)s_box[cam_k][2 * which + 1]));
      };
      // the decision, by every thread on the same sums (lane l of every wave: camera l; at most 64 cameras in this form): the
      // same order of operations as ba_motion_step's prologue
      const int lane = tid & 63;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      if (lane < nfp) {
        a0 = part_at(lane, 0);
        a1 = st.stage == 1 ? fmax(0.0, part_at(lane, 3)) : part_at(lane, 1);
        a2 = part_at(lane, 2);
      }
      const double chi = mo_wave_reduce<false>(a0);
      const double second = st.stage == 1 ? mo_wave_reduce<true>(a1) : mo_wave_reduce<false>(a1);
      const double bad = mo_wave_reduce<false>(a2);
      const int cur_before = st.cur;
      mo_decide(D, st, chi, second, bad, c == 0 && tid == 0);
      st.seq = step;
      const bool accepted = st.cur != cur_before;
      used_spec = accepted && have_spec && st.need_lin;
      if (accepted) {  // the trial record is the estimate now -- and, if it was linearised ahead, its sums the normal equations
        if (tid < kCamStride) s_cam[tid] = s_trial[tid];
        if (used_spec && tid < 28) s_sum[tid] = s_spec[tid];
        __syncthreads();
      }
      speculate = accepted || st.stage == 1;  // (stage 1: the decision behind the first linearisation -- the first trial follows)
      have_spec = false;
      VS_MO_STAMP(2);
    }
    if (step == 0) used_spec = false;
    if (st.done || step > max_steps) break;
    const double* cam = s_cam;
    const bool lin_only = st.need_lin && st.it == 0 && st.stage == 0;
    if (st.need_lin && !used_spec) {
      linearise(s_cam, s_sum);  // s_sum is kept across retries
      VS_MO_STAMP(3);
      if (lin_only) {
        if (tid == 0) {
          double mx = 0.0;
          int n = 0;
          for (int k = 0; k < 6; ++k) {
            mx = fmax(mx, fabs(s_sum[n]));
            n += 6 - k;
          }
          s_pv[0] = s_sum[27];
          s_pv[1] = 0.0;
          s_pv[2] = 0.0;
          s_pv[3] = mx;
        }
        post_partials(step);
        VS_MO_STAMP(6);
        st.stage = 1;
        continue;
      }
    }
    // ---- 6x6 solve (H + lambda I) x = b and SBACam::update into the trial buffer: one thread
    double* trial = D.cam[st.cur ^ 1] + (size_t)pose * kCamStride;
    if (tid == 0) {
      double A[6][6], x[6];
      int n = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = k; l < 6; ++l) {
          A[k][l] = s_sum[n];
          A[l][k] = s_sum[n];
          ++n;
        }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        A[k][k] += st.lambda;
        x[k] = s_sum[21 + k];
      }
      int ok = 1;
      double rinv[6];
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        const double d = A[jj][jj];
        if (!(d > 0.0)) ok = 0;
        const double ri = vs_fast_rsq(d);
        A[jj][jj] = d * ri;
        rinv[jj] = ri;
#pragma unroll
        for (int i = jj + 1; i < 6; ++i) A[i][jj] = A[i][jj] * rinv[jj];
#pragma unroll
        for (int i = jj + 1; i < 6; ++i)
#pragma unroll
          for (int k = jj + 1; k <= i; ++k) A[i][k] -= A[i][jj] * A[k][jj];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        x[k] = x[k] * rinv[k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) x[i] -= A[i][k] * x[k];
      }
#pragma unroll
      for (int k = 5; k >= 0; --k) {
        x[k] = x[k] * rinv[k];
#pragma unroll
        for (int i = 0; i < k; ++i) x[i] -= A[k][i] * x[k];
      }
      double sc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) sc += x[k] * (st.lambda * x[k] + s_sum[21 + k]);
      double t[3] = {cam[0] + x[0], cam[1] + x[1], cam[2] + x[2]};
      const double bx = x[3], by = x[4], bz = x[5];
      const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));
      const double ax = cam[3], ay = cam[4], az = cam[5], aw = cam[6];
      const double w = aw * bw - ax * bx - ay * by - az * bz;
      const double xx = aw * bx + ax * bw + ay * bz - az * by;
      const double yy = aw * by + ay * bw + az * bx - ax * bz;
      const double zz = aw * bz + az * bw + ax * by - ay * bx;
      const double inrm = vs_fast_rsq(xx * xx + yy * yy + zz * zz + w * w);
      double q[4] = {xx * inrm, yy * inrm, zz * inrm, w * inrm};
      double rec[kCamStride];
      for (int k = 0; k < 3; ++k) rec[k] = t[k];
      for (int k = 0; k < 4; ++k) rec[3 + k] = q[k];
      quat_to_w2n(t, q, rec + 7);
      for (int k = 0; k < kCamStride; ++k) {
        trial[k] = rec[k];
        s_trial[k] = rec[k];  // the workgroup evaluates the trial from LDS
      }
      s_x[0] = sc;
      s_ok = ok;
    }
    VS_MO_STAMP(4);
    __syncthreads();
    // ---- robust chi2 of this camera's trial state
    double tcam[kCamStride];
#pragma unroll
    for (int k = 0; k < kCamStride; ++k) tcam[k] = s_trial[k];
    const double scl = s_x[0];
    const int ok = s_ok;
    double chi = 0.0;
#pragma unroll
    for (int j = 0; j < kMoObsRegs; ++j) {
      if (o0 + tid + j * kMoThreads >= o1) continue;
      edge_t E;
      eval_edge<false>(D, tcam, oX[j], oUV[j], D.has_info ? oW[j] : nullptr, E);
      chi += E.rho0;
    }
    if constexpr (OVF) {
      for (int i = o0 + tid + kMoObsRegs * kMoThreads; i < o1; i += kMoThreads) {
        const double X[3] = {D.mo_X[3 * (size_t)i], D.mo_X[3 * (size_t)i + 1], D.mo_X[3 * (size_t)i + 2]};
        edge_t E;
        eval_edge<false>(D, tcam, X, D.mo_uv + 2 * (size_t)i, D.has_info ? D.mo_info + 3 * (size_t)i : nullptr, E);
        chi += E.rho0;
      }
    }
    // workgroup sum of chi: fixed-order reduction inside each wave (mo_wave_reduce), then the wave totals in wave order
    chi = mo_wave_reduce<false>(chi);
    if ((tid & 63) == 0) s_part[0][tid >> 6] = chi;
    __syncthreads();
    if (tid == 0) {
      double tot = s_part[0][0];
      for (int wv = 1; wv < kMoThreads / 64; ++wv) tot += s_part[0][wv];
      s_pv[0] = tot;
      s_pv[1] = scl;
      s_pv[2] = ok ? 0.0 : 1.0;
      s_pv[3] = 0.0;
    }
    VS_MO_STAMP(5);
    post_partials(step);
    VS_MO_STAMP(6);
    st.stage = 2;
    if (speculate && st.it + 1 < D.max_it && step < max_steps) {  // (an accepted trial that ends the solve needs no linearisation)
      linearise(s_trial, s_spec);
      have_spec = true;
    }
  }
#undef VS_MO_STAMP
  if (c == 0 && tid == 0) {
    g_state[0] = st;
    g_state[1] = st;
  }
  // chained tracking: "every camera's record and the LM state are final" as a tagged word -- the next frame's PnP launch
  // (another stream) is already resident and waits for it.  Thread 0 is the only writer of this workgroup's global results.
  if (D.mo_done && tid == 0) {
    __threadfence();
    const unsigned prev = __hip_atomic_fetch_add(D.mo_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)nfp - 1u) {
      __hip_atomic_store(D.mo_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(D.mo_done + 64, D.mo_done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template __global__ void ba_motion_persistent<false>(ba_dev D, int max_steps);
template __global__ void ba_motion_persistent<true>(ba_dev D, int max_steps);

}  // namespace vsba

namespace {
// ------------------------------------------------------------------------------------------------ host
struct arena {
  uint8_t* base = nullptr;  // device
  uint8_t* host = nullptr;  // pinned mirror for the upload part
  size_t off = 0;
  template <class T>
  T* take(size_t count, T** host_ptr = nullptr) {
    off = (off + 255) & ~(size_t)255;
    T* d = reinterpret_cast<T*>(base + off);
    if (host_ptr) *host_ptr = reinterpret_cast<T*>(host + off);
    off += sizeof(T) * (count ? count : 1);
    return d;
  }
};

// dense solver of the reduced camera system: which kernels a system of np unknowns takes, and their launch
struct solve_plan {
  bool lds = false;        // np <= kMaxLdsPacked: ba_solve_block, one workgroup, [S | rhs] in LDS
  bool packed = false;     // ... as a packed lower triangle (np > kMaxLdsN)
  size_t lds_bytes = 0;
  int nbw = 0;             // otherwise: panel width of the blocked HBM factorisation (0: element-wise last resort)
  size_t panel_lds = 0;
  bool band_ok = false;    // ba_chol_band + ba_chol_finish fit (any np up to ~18 000 unknowns): taken when the system is banded
};

int plan_solve(vs_ctx* ctx, int np, solve_plan* P) {
  const int mode = ctx->tune.ba_solve_packed;  // vs_tune_ba_solve: tests hold the three forms to the same bits
  P->lds = np <= (mode == 1 ? kMaxLdsN : kMaxLdsPacked);
  P->packed = P->lds && (np > kMaxLdsN || mode == 2);
  P->lds_bytes = 32 + (!P->lds ? 0 : sizeof(double) * ((P->packed ? (size_t)(np + 1) * (np + 2) / 2 : (size_t)(np + 1) * ((np + 1) | 1)) + 2 * (size_t)np + 4));
  P->nbw = 0;
  P->panel_lds = 0;
  P->band_ok = false;
  if (!P->lds) {
    // widest panel (24 / 12 / 6 columns) whose rows j0..n fit in LDS.  (48-column panels were measured in round 5 and are slower:
    // the update of a panel's own remaining columns is one workgroup's work and grows with the square of the width -- 7 launches of
    // 33 us against 13 of 14.5 at 306 unknowns; ba_chol_update takes any width in slices of 24 all the same.)
    for (int w : {24, 12, 6}) {
      const size_t b = sizeof(double) * ((size_t)(np + 1) * (w | 1) + w) + 64;
      if (b <= 150 * 1024 && sizeof(double) * ((size_t)np + 25 * 24 + 24) + 64 <= 150 * 1024) {
        P->nbw = w;
        P->panel_lds = b;
        break;
      }
    }
    if (P->nbw) VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_chol_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->panel_lds));
    // the banded factorisation keeps a window of the band in LDS whatever np is; the back substitution keeps x (np doubles)
    P->band_ok = sizeof(double) * ((size_t)np + 25 * 24 + 24) + 64 <= 150 * 1024;
    if (P->nbw || P->band_ok) {
      VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_chol_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * ((size_t)np + 25 * 24 + 24) + 64)));
      VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_chol_band, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLds));
    }
  }
  if (P->lds_bytes > 64 * 1024) {
    if (P->packed) VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_solve_block<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_bytes));
    else VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_solve_block<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_bytes));
  }
  return VS_OK;
}

int launch_solve(vs_ctx* ctx, hipStream_t s, const ba_dev& D, const solve_plan& P) {
  const int np = D.np;
  if (P.lds && P.packed) {
    hipLaunchKernelGGL(ba_solve_block<true>, dim3(1), dim3(kSolveBlock), P.lds_bytes, s, D);
  } else if (P.lds) {
    hipLaunchKernelGGL(ba_solve_block<false>, dim3(1), dim3(kSolveBlock), P.lds_bytes, s, D);
  } else if (P.band_ok && D.band > 0 && D.band <= kBandMax) {
    hipLaunchKernelGGL(ba_chol_band, dim3(1), dim3(kBandThreads), kBandLds, s, D, std::max(D.band, 12));  // (the look-ahead wants the next diagonal block inside the window)
    hipLaunchKernelGGL(ba_chol_finish, dim3(1), dim3(kPanelThreads), sizeof(double) * ((size_t)np + 25 * 24 + 24) + 64, s, D, 24);
  } else if (P.nbw > 0) {
    for (int j0 = 0; j0 < np; j0 += P.nbw) {
      const int w = std::min(P.nbw, np - j0);
      hipLaunchKernelGGL(ba_chol_panel, dim3(1), dim3(kPanelThreads), P.panel_lds, s, D, j0, w);
      const int rem = np + 1 - (j0 + w);  // trailing rows incl. the rhs row
      if (rem > 0) {
        const unsigned T = (unsigned)((rem + kUpdTile - 1) / kUpdTile);
        hipLaunchKernelGGL(ba_chol_update, dim3(T, T), dim3(256), 0, s, D, j0, w);
      }
    }
    hipLaunchKernelGGL(ba_chol_finish, dim3(1), dim3(kPanelThreads), sizeof(double) * ((size_t)np + 25 * 24 + 24) + 64, s, D, P.nbw);
  } else {
    hipLaunchKernelGGL(ba_solve<false>, dim3(1), dim3(kSolveThreads), P.lds_bytes, s, D);
  }
  VS_LAUNCH_CHECK(ctx, "ba_solve");
  return VS_OK;
}

}  // namespace

// Test hook: the dense solver alone (see include/vslam_hip.h).  A minimal ba_dev without cameras or points: the
// kernels read S, bs, bp and the LM state and write xp and solve_ok.
// tuning / test hooks (not part of the stable ABI; per context, vs_tuning in vs_internal.h): Schur kernel of single-tile
// windows (0 automatic = ba_schur_small with the linearisation of accepted states folded into the trial, 1 = the general
// tile kernel, 2 = ba_schur_small with a linearisation launch per iteration), points per workgroup and the cap on the
// number of slabs of ba_schur_small (from 64 points up the value sets the slab size of ba_schur_window instead; 3 = banded
// windows of several tiles on the tile kernel, not on ba_schur_window); motion-only form (0 = one launch where it applies, 1 = one launch per LM step)
VS_API int vs_tune_ba(vs_ctx* ctx, int schur_variant, int points_per_workgroup, int max_slabs, int motion_variant) {
  if (!ctx) return VS_EINVAL;
  if (schur_variant >= 0 && schur_variant <= 3) ctx->tune.schur_variant = schur_variant;
  if (points_per_workgroup >= 64) ctx->tune.win_per = points_per_workgroup;  // ba_schur_window's slabs are never that small
  else if (points_per_workgroup > 0) ctx->tune.small_per = points_per_workgroup;
  if (max_slabs > 0) ctx->tune.small_ns_cap = max_slabs;
  if (motion_variant == 0 || motion_variant == 1) ctx->tune.motion_variant = motion_variant;
  return VS_OK;
}

// developer hook (include/vslam_hip_dev.h): where the structure of a large problem is built (tests compare both)
VS_API int vs_tune_ba_structure(vs_ctx* ctx, int on_host) {
  if (!ctx) return VS_EINVAL;
  ctx->tune.ba_host_structure = on_host != 0;
  return VS_OK;
}

VS_API int vs_ba_structure_on_device(vs_ctx* ctx) { return ctx ? ctx->ba_structure_dev : 0; }

// developer hook (include/vslam_hip_dev.h): experiment -- every batch of LM slots replayed as one captured hipGraph
VS_API int vs_tune_ba_graph(vs_ctx* ctx, int on, double* last_batch_us) {
  if (!ctx) return VS_EINVAL;
  if (on >= 0 && on <= 2) ctx->tune.ba_graph = on;
  if (last_batch_us) *last_batch_us = ctx->ba_batch_us;
  return VS_OK;
}

// developer hook (include/vslam_hip_dev.h): storage of the reduced system in ba_solve_block
VS_API int vs_tune_ba_solve(vs_ctx* ctx, int packed_mode) {
  if (!ctx) return VS_EINVAL;
  if (packed_mode >= 0 && packed_mode <= 2) ctx->tune.ba_solve_packed = packed_mode;
  return VS_OK;
}

// developer hook (include/vslam_hip_dev.h): the kernels the newest vs_ba_solve of this context took
VS_API int vs_ba_last_path(vs_ctx* ctx, int* out6) {
  if (!ctx || !out6) return VS_EINVAL;
  for (int i = 0; i < 6; ++i) out6[i] = ctx->ba_path[i];
  return VS_OK;
}

// developer hook (include/vslam_hip_dev.h): fill every device buffer allocated from now on with `byte` (-1: off)
VS_API int vs_debug_poison_alloc(vs_ctx* ctx, int byte) {
  if (!ctx || byte < -1 || byte > 255) return VS_EINVAL;
  ctx->tune.poison_alloc = byte;
  return VS_OK;
}

namespace vsba {
// The one-launch motion-only solve needs all its camera workgroups resident at the same time (they rendezvous through
// mailboxes).  The bound is what THIS device can hold -- compute units x workgroups of this kernel per unit, as the runtime
// reports it -- not a constant: a partitioned or smaller device takes the launch-per-step form instead of spinning.
bool mo_persistent_ok(vs_ctx* ctx, int cameras, int max_steps) {
  if (ctx->tune.motion_variant != 0 || cameras > kMoPersistCameras || max_steps >= 4000) return false;
  if (ctx->mo_persist_cap < 0) {
    int per_cu = 0, cap = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)ba_motion_persistent<true>, kMoThreads, 0) == hipSuccess && per_cu > 0) {
      int per_cu2 = per_cu;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, (const void*)ba_motion_persistent<false>, kMoThreads, 0) != hipSuccess || per_cu2 <= 0) per_cu2 = per_cu;
      cap = std::min(per_cu, per_cu2) * ctx->prop.multiProcessorCount;
    } else {
      (void)hipGetLastError();
    }
    ctx->mo_persist_cap = cap;
  }
  return cameras <= ctx->mo_persist_cap;
}
}  // namespace vsba

VS_API int vs_ba_debug_cholesky(vs_ctx* ctx, const double* S, int n, const double* b, double* x, int* ok) {
  if (!ctx) return VS_EINVAL;
  if (!S || !b || !x || !ok || n <= 0 || n % 6 != 0)
    return vs_fail(ctx, VS_EINVAL, "%s: need S, b, x, ok and n a positive multiple of 6", "vs_ba_debug_cholesky");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nn = (size_t)n * n;
  const size_t bytes = sizeof(double) * (nn + 4 * (size_t)n + 64) + sizeof(lm_state) + 1024;
  VS_TRY(vs_reserve(ctx, &ctx->d_ba, bytes));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, sizeof(lm_state) + 64));
  VS_HIP(ctx, hipStreamSynchronize(s));
  ba_dev D;
  memset(&D, 0, sizeof D);
  D.np = n;
  D.nfp = n / 6;
  double* base = (double*)ctx->d_ba.p;
  D.S = base;
  D.bs = D.S + nn;
  D.bp = D.bs + n;
  D.xp = D.bp + n;
  D.rinv = D.xp + n;
  D.chol_fail = reinterpret_cast<int*>(D.rinv + n);
  D.st = reinterpret_cast<lm_state*>(D.rinv + n + 8);
  D.cam[0] = D.cam[1] = base;  // n_poses = 0: never dereferenced
  lm_state* hst = (lm_state*)ctx->h_pin.p;
  memset(hst, 0, sizeof(lm_state));
  VS_HIP(ctx, hipMemcpyAsync(D.st, hst, sizeof(lm_state), hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(D.S, S, sizeof(double) * nn, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(D.bs, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(D.bp, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemsetAsync(D.chol_fail, 0, 16, s));
  if (ctx->tune.schur_variant != 3) {  // a banded matrix takes the banded factorisation, as vs_ba_solve's banded windows do
    int band = 0;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c <= r - band; ++c)
        if (S[(size_t)r * n + c] != 0.0) {
          band = r - c + 1;
          break;
        }
    band = (band + 5) / 6 * 6;
    if (band <= kBandMax) D.band = band;
  }
  solve_plan P;
  VS_TRY(plan_solve(ctx, n, &P));
  VS_TRY(launch_solve(ctx, s, D, P));
  VS_HIP(ctx, hipMemcpyAsync(hst, D.st, sizeof(lm_state), hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  *ok = hst->solve_ok;
  if (hst->solve_ok) {
    VS_HIP(ctx, hipMemcpyAsync(x, D.xp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  return VS_OK;
}

VS_API int vs_ba_solve(vs_ctx* ctx, const vs_ba_problem* p, vs_ba_result* res) {
  if (!ctx) return VS_EINVAL;
  if (!p || !res) return vs_fail(ctx, VS_EINVAL, "%s: null problem/result", "vs_ba_solve");
  if (p->n_poses < 0 || p->n_points < 0 || p->n_obs < 0 || p->n_scale < 0 || p->max_iterations < 0 ||
      (p->n_poses && (!p->poses || !p->pose_fixed)) || (p->n_points && (!p->points || !p->point_fixed)) ||
      (p->n_obs && (!p->obs_pose || !p->obs_point || !p->obs_uv)) ||
      (p->n_scale && (!p->scale_parent || !p->scale_child || !p->scale_meas)))
    return vs_fail(ctx, VS_EINVAL, "%s: inconsistent sizes / null arrays", "vs_ba_solve");
  for (int k = 0; k < p->n_scale; ++k)
    if (p->scale_parent[k] < 0 || p->scale_parent[k] >= p->n_poses || p->scale_child[k] < 0 ||
        p->scale_child[k] >= p->n_poses)
      return vs_fail(ctx, VS_EINVAL, "%s: scale edge index out of range", "vs_ba_solve");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;

  // developer aid (VS_BA_TIMING=1): host phase times on stderr
  static const bool timing = getenv("VS_BA_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  const auto t_begin = now();
  double lap_us[8] = {0};  // VS_BA_TIMING: stretches of the arena fill
  auto t_lap = t_begin;
  auto lap = [&](int k) {
    if (!timing) return;
    const auto t = now();
    lap_us[k] += us(t_lap, t);
    t_lap = t;
  };
  // ---- structure (host).  Pass 1 over the observations: index validation, active observations per point and per free
  // camera, number of Hpl blocks -- every array size is known after it, so pass 2 writes the device arrays straight into
  // the pinned arena.  Scratch vectors are kept per host thread (no allocation in the steady state).
  const int F = p->n_poses, P = p->n_points;
  static thread_local struct {
    std::vector<int> pose_slot, pt_slot, cnt, fill, order, cam_start, cfill, seen_by, seen_cnt, wlo, whi, wcnt;
  } W;
  W.pose_slot.resize(F ? F : 1);
  W.pt_slot.resize(P ? P : 1);
  int* pose_slot = W.pose_slot.data();
  int* pt_slot = W.pt_slot.data();
  int nfp = 0, nfl = 0;
  for (int i = 0; i < F; ++i) pose_slot[i] = p->pose_fixed[i] ? -1 : nfp++;
  for (int j = 0; j < P; ++j) pt_slot[j] = p->point_fixed[j] ? -1 : nfl++;
  const int np = 6 * nfp;
  // Large problems whose observation arrays lie in pinned memory: the arrays are DMA-ed from where they lie and the structure
  // below is built ON THE DEVICE from that copy (vs_ba_build.hip) -- no host pass reads the observations at all.  Whatever that
  // path does not cover (an index out of range, a list not grouped by point, inactive observations or points, a camera twice
  // in one point) comes back as a flag, and the host passes run after all (`goto host_passes`, once).
  static const bool kHostStructure = getenv("VS_BA_HOST_STRUCTURE") != nullptr;  // developer aid: never build on the device
  bool dev = !kHostStructure && !ctx->tune.ba_host_structure && p->n_obs >= 400000 && !p->obs_info && nfp > 0 && nfl > 0 && nfp + 1 <= kBuildMaxKeys &&
             p->max_iterations > 0 && vs_is_pinned(p->obs_uv) && vs_is_pinned(p->obs_pose) && vs_is_pinned(p->obs_point);
host_passes:
  W.cnt.assign(dev ? 1 : (size_t)P + 1, 0);
  W.cam_start.assign((size_t)nfp + 1, 0);
  int* cnt = W.cnt.data();
  int* cam_start = W.cam_start.data();
  int n_obs = 0, n_hpl = 0, prev_pt = -1;
  bool grouped = true;  // observations arrive grouped by point in ascending order (as the reference adds its edges)
  // large problems (the scaled run: 2 000 000 observations) spread both structure passes over a few host threads
  static const int kThreadsEnv = getenv("VS_BA_THREADS") ? atoi(getenv("VS_BA_THREADS")) : 0;  // developer aid
  int T = p->n_obs >= 400000 ? std::max(1, std::min(kThreadsEnv > 0 ? kThreadsEnv : 12, (int)std::thread::hardware_concurrency())) : 1;
  auto par_for = [ctx](int nt, auto&& body) {  // body(t, nt) on nt threads of the context's pool, the caller's included
    using B = std::remove_reference_t<decltype(body)>;
    ctx->pool.run(nt, [](void* a, int t, int n) { (*static_cast<B*>(a))(t, n); }, (void*)&body);
  };
  // (the loops are written out, not shared through a lambda: the closure's indirections cost 10 us at 20 000 observations)
  if (dev) {
    n_obs = n_hpl = p->n_obs;  // bounds until the device reports (every observation active is a condition of that path)
  } else if (T == 1) {
    // locals whose address is never taken: the threaded variant below captures the function's own variables by reference,
    // which would make every store here a possible alias of them
    int* const cnt_ = cnt;
    int* const cam_ = cam_start;
    const int* const pose_slot_ = pose_slot;
    const int* const pt_slot_ = pt_slot;
    const int* const op = p->obs_pose;
    const int* const oq = p->obs_point;
    const int n_all = p->n_obs;
    int n_obs_ = 0, n_hpl_ = 0, prev_ = -1;
    bool grouped_ = true;
    for (int o = 0; o < n_all; ++o) {
      const int ci = op[o], pj = oq[o];
      if (ci < 0 || ci >= F || pj < 0 || pj >= P) return vs_fail(ctx, VS_EINVAL, "%s: observation index out of range", "vs_ba_solve");
      grouped_ &= pj >= prev_;
      prev_ = pj;
      const int cs = pose_slot_[ci], ls = pt_slot_[pj];
      if (cs < 0 && ls < 0) continue;  // fixed camera and fixed point: not part of the problem
      cnt_[pj + 1]++;
      ++n_obs_;
      if (cs >= 0) {
        cam_[cs + 1]++;
        n_hpl_ += ls >= 0;
      }
    }
    n_obs = n_obs_;
    n_hpl = n_hpl_;
    prev_pt = prev_;
    grouped = grouped_;
  } else {
    // Threads take ranges of the observation list that begin and end where the point changes, so -- when the list is grouped
    // by point, which large problems are -- no two threads count for the same point and no atomics are needed.  A list
    // that turns out not to be grouped is counted again, sequentially.
    struct part1 {
      std::vector<int> cam;
      int n_obs = 0, n_hpl = 0, first = -1, last = -1;
      bool grouped = true, bad = false;
    };
    // (more ranges than threads, handed out through a counter: a thread that loses its core for a while does not hold
    // up the pass; inside a range everything lives in locals -- the closure's references would be re-read after each store)
    const int nchunk = 4 * T;
    std::vector<part1> parts((size_t)nchunk);
    std::atomic<int> next1{0};
    par_for(T, [&](int, int) {
      const int* const op = p->obs_pose;
      const int* const oq = p->obs_point;
      const int* const pose_slot_ = pose_slot;
      const int* const pt_slot_ = pt_slot;
      int* const cnt_ = cnt;
      const int n_all = p->n_obs, F_ = F, P_ = P, nfp_ = nfp;
      for (int ch = next1.fetch_add(1); ch < nchunk; ch = next1.fetch_add(1)) {
        part1& R = parts[(size_t)ch];
        R.cam.assign((size_t)nfp_ + 1, 0);
        int* const rcam = R.cam.data();
        int o0 = (int)((long long)n_all * ch / nchunk), o1 = (int)((long long)n_all * (ch + 1) / nchunk);
        // move both ends forward to the next change of point (the previous range finishes the point it is in)
        auto pt_ok = [=](int o) { return oq[o] >= 0 && oq[o] < P_; };
        while (o0 > 0 && o0 < n_all && pt_ok(o0) && pt_ok(o0 - 1) && oq[o0] == oq[o0 - 1]) ++o0;
        while (o1 > 0 && o1 < n_all && pt_ok(o1) && pt_ok(o1 - 1) && oq[o1] == oq[o1 - 1]) ++o1;
        int prev = -1, first = -1, n_o = 0, n_h = 0;
        bool grp = true, bad = false;
        for (int o = o0; o < o1; ++o) {
          const int ci = op[o], pj = oq[o];
          if (ci < 0 || ci >= F_ || pj < 0 || pj >= P_) {
            bad = true;
            break;
          }
          if (first < 0) first = pj;
          grp &= pj >= prev;
          prev = pj;
          const int cs = pose_slot_[ci], ls = pt_slot_[pj];
          if (cs < 0 && ls < 0) continue;
          cnt_[pj + 1]++;
          ++n_o;
          if (cs >= 0) {
            rcam[cs + 1]++;
            n_h += ls >= 0;
          }
        }
        R.first = first;
        R.last = prev;
        R.n_obs = n_o;
        R.n_hpl = n_h;
        R.grouped = grp;
        R.bad = bad;
      }
    });
    for (const part1& R : parts) {
      if (R.bad) return vs_fail(ctx, VS_EINVAL, "%s: observation index out of range", "vs_ba_solve");
      // strictly greater across ranges: a point that shows up in two ranges was counted by two threads at once
      grouped &= R.grouped && (R.first < 0 || R.first > prev_pt);
      if (R.last >= 0) prev_pt = R.last;
      n_obs += R.n_obs;
      n_hpl += R.n_hpl;
      for (int c = 0; c <= nfp; ++c) cam_start[c] += R.cam[(size_t)c];
    }
    if (!grouped) {  // not grouped by point: the ranges may have raced on a point's counter
      W.cnt.assign((size_t)P + 1, 0);
      W.cam_start.assign((size_t)nfp + 1, 0);
      cnt = W.cnt.data();
      cam_start = W.cam_start.data();
      n_obs = n_hpl = 0;
      prev_pt = -1;
      grouped = true;
    for (int o = 0; o < p->n_obs; ++o) {
        const int ci = p->obs_pose[o], pj = p->obs_point[o];
        if (ci < 0 || ci >= F || pj < 0 || pj >= P) return vs_fail(ctx, VS_EINVAL, "%s: observation index out of range", "vs_ba_solve");
        grouped &= pj >= prev_pt;
        prev_pt = pj;
        const int cs = pose_slot[ci], ls = pt_slot[pj];
        if (cs < 0 && ls < 0) continue;  // fixed camera and fixed point: not part of the problem
        cnt[pj + 1]++;
        ++n_obs;
        if (cs >= 0) {
          cam_start[cs + 1]++;
          n_hpl += ls >= 0;
        }
      }
      T = 1;
    }
  }
  // active points = free points (even without observations: they still receive the lambda damping) + fixed points
  // that are observed by a free camera.  Skipped points own no active observation, so the sorted observation ranges
  // of consecutive active points are adjacent: pt_start[a] = cnt[act_pt[a]], pt_start[n_act] = n_obs.
  int n_act = dev ? P : 0;  // (device path: every point active is a condition, too)
  if (!dev)
    for (int j = 0; j < P; ++j) {
      n_act += cnt[j + 1] > 0 || pt_slot[j] >= 0;
      cnt[j + 1] += cnt[j];
    }
  for (int c = 0; c < nfp; ++c) cam_start[c + 1] += cam_start[c];
  const int n_cam_obs = dev ? n_obs : cam_start[nfp];
  // stable order of the active observations by point; the identity when they arrive grouped and all are active
  const bool identity = dev || (grouped && n_obs == p->n_obs);
  const int* order = nullptr;
  if (!identity) {
    W.order.resize(n_obs ? n_obs : 1);
    W.fill.assign(cnt, cnt + P);
    int* fill = W.fill.data();
    for (int o = 0; o < p->n_obs; ++o)
      if (pose_slot[p->obs_pose[o]] >= 0 || pt_slot[p->obs_point[o]] >= 0) W.order[fill[p->obs_point[o]]++] = o;
    order = W.order.data();
  }
  const int ntile = (nfp + kTileCams - 1) / kTileCams;
  // tiled Schur (measured faster than the LDS-slab kernel at every window size) unless a point is seen twice from one
  // camera, which pass 2 finds out
  const bool tiled_possible = np > 0 && ntile <= 64 && nfl > 0;

  // ---- motion-only fast path (block-diagonal problem): one launch per LM trial, one workgroup per free camera
  const bool motion_only = nfl == 0 && p->n_scale == 0 && nfp > 0 && p->max_iterations > 0;

  // ---- result defaults
  res->iterations = res->trials = res->not_pd = res->terminated = 0;
  res->chi2_initial = res->chi2_final = res->lambda_final = 0.0;

  // ---- launch geometry (the slab count of the duplicate-observation fallback is the only thing pass 2 can change)
  const int nb_pt = std::max(1, (n_act + kPtPerBlock - 1) / kPtPerBlock);
  const bool lds_slab = np <= kMaxSlabN;
  const int ntile_pairs = ntile * (ntile + 1) / 2;  // tiles of the lower triangle
  const size_t slab_elems = (size_t)np * np + np;
  const int small_per = ctx->tune.small_per > 0 ? ctx->tune.small_per : kSmallPts;
  auto slabs_for = [&](bool tiled_, bool small_) {
    int n = nfl > 0 && nfp > 0 ? std::min(256, (nfl + 7) / 8) : 0;
    if (tiled_) {
      n = std::max(4, std::min(256, (16384 + ntile_pairs - 1) / ntile_pairs));  // >= 16k workgroups: most tiles are empty
      // ... but not more slabs than the points can fill: at least 8 free points per slab (round 5: the real-sequence problems and
      // the growing global BA have ~1 200 points -- 256 slabs of 5 points each made ba_schur_tile and ba_reduce 6 - 7 % of a
      // trial slower than 150 slabs; 16 / 32 / 64 / 128 points per slab measured 424 / 428 / 438 / 487 us per trial at 52 key
      // frames against 431 at 8 and 457 before; developer aid VS_BA_TILE_PER overrides)
      static const int kTilePer = getenv("VS_BA_TILE_PER") ? atoi(getenv("VS_BA_TILE_PER")) : 8;
      if (kTilePer > 0 && !small_ && ntile > 1) n = std::max(4, std::min(n, (nfl + kTilePer - 1) / kTilePer));  // (single tile: ba_schur_small's territory; the tile kernel only runs there when a test forces it)
    }
    if (small_) n = std::max(1, std::min(ctx->tune.small_ns_cap, (nfl + small_per - 1) / small_per));
    if (!lds_slab && n > 0) n = std::min(n, std::max(1, (int)((512u << 20) / (sizeof(double) * slab_elems))));
    return n;
  };
  const bool small_possible = tiled_possible && ntile == 1 && ctx->tune.schur_variant != 1;  // one tile: ba_schur_small
  const int ns_bound = std::max(slabs_for(false, false), slabs_for(tiled_possible, small_possible));
  // ... and of the banded-window path (ba_schur_window), whose slabs are kWinSlabElems doubles whatever np is: with few free
  // cameras np * np + np is much smaller than that, and a window plan with hundreds of slabs outgrew the reservation made for
  // the tile path (round-3 advisor: 13 poses x 20 000 points needed 58 MB against 39 MB).  The plan below takes at most
  // `target` slabs (win_per >= ceil(win_n / target)) unless vs_tune_ba fixes the slab size.
  const bool win_possible = tiled_possible && ntile > 1 && ctx->tune.schur_variant != 3;
  const int win_target = 2 * std::max(ctx->prop.multiProcessorCount, 64);
  const int ns_win_bound = !win_possible ? 0
                           : ctx->tune.win_per > 0 ? (nfl + std::min(kWinPerMax, ctx->tune.win_per) - 1) / std::min(kWinPerMax, ctx->tune.win_per)
                                                   : std::max(std::min(win_target, (nfl + 31) / 32), (nfl + kWinPerMax - 1) / kWinPerMax);
  const size_t slab_doubles = std::max((size_t)(ns_bound ? ns_bound : 1) * slab_elems, (size_t)ns_win_bound * kWinSlabElems);
  // workgroups per camera of the linearisation's camera role: about one observation per thread, at most 8
  int cam_split = 1;
  for (int c = 0; c < nfp; ++c) cam_split = std::max(cam_split, (cam_start[c + 1] - cam_start[c] + kCamThreads - 1) / kCamThreads);
  cam_split = dev ? 8 : std::min(cam_split, 8);  // (device path: the bound, for the arena; the count comes back with the structure)

  const auto t_struct = now();
  t_lap = t_struct;
  // ---- arena: [uploaded constants | state | system]
  vs_ba_problem const& q = *p;
  size_t need = (1u << 20) + sizeof(int) * ((size_t)F + P + 3 * (size_t)n_act + 4 * (size_t)n_obs + 2 * (size_t)nfp +
                                            2 * (size_t)n_cam_obs + 2 * (size_t)q.n_scale + nfl + 64) +
                sizeof(double) * (5 * (size_t)n_obs + (size_t)q.n_scale + 2 * (size_t)F * kCamStride + 6 * (size_t)P +
                                  2 * (size_t)np * np + 8 * (size_t)np + 12 * (size_t)nfl + 18 * (size_t)n_obs +
                                  9 * (size_t)nfl + (tiled_possible ? 5 * (size_t)nfl + nfp + 4096 : 0) + slab_doubles + 3 * (size_t)nb_pt + nfp +
                                  2 * (size_t)q.max_iterations + 64) +
                256 * 64 + sizeof(int) * (3 * (size_t)n_obs + nfl + 16 + 2 * (size_t)ns_win_bound + nfp + 64) + 4 * 256 + (sizeof(double) * 27 * 8 + 8) * (size_t)nfp + 1024 +
                sizeof(double) * (64 + (size_t)F * kCamStride + 3 * (size_t)P + 2 * (size_t)q.max_iterations) + 512 +
                (small_possible || (win_possible && nb_pt >= 1024) ? sizeof(double) * ((size_t)np * np + np + 12 * (size_t)nfl + 18 * (size_t)n_obs) + 5 * 256 : 0) + sizeof(double) * 8 * (size_t)(res->trial_trace ? std::max(res->trial_trace_cap, 0) : 0) + (motion_only ? sizeof(double) * (8 * (size_t)n_cam_obs + 50 * (size_t)nfp + 64) : 0);
  if (dev) need += sizeof(int) * (ba_build_temp_ints(n_obs, nfp, nfl) + (size_t)nfl + 2 * (size_t)ns_win_bound + nfp + 64) + 4096;
  VS_TRY(vs_reserve(ctx, &ctx->d_ba, need));
  // the pinned mirror covers what is uploaded from it or read back into it.  Host passes: the structure arrays, i.e. nearly
  // everything; device-built structure: the slot tables, the start state, the read-back block and a few flag words -- a few MB behind
  // the (device-only) observation arrays at the front of the arena, not the whole gigabyte (pinning it cost 0.23 s on a context's
  // first large solve)
  const int trial_cap_early = res->trial_trace && res->trial_trace_cap > 0 ? res->trial_trace_cap : 0;
  const size_t out_elems_early = 32 + (size_t)F * kCamStride + 3 * (size_t)P + 2 * (size_t)p->max_iterations + 4 * (size_t)trial_cap_early;
  const size_t pin_need = !dev ? need
                               : (size_t)24 * n_obs + sizeof(int) * ((size_t)F + P + 3 * (size_t)nfp + 2 * (size_t)p->n_scale + 256) +
                                     sizeof(double) * ((size_t)p->n_scale + (size_t)F * kCamStride + 3 * (size_t)P + out_elems_early + 256) + 256 * 40 + (1u << 16);
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, pin_need));
  VS_HIP(ctx, hipStreamSynchronize(s));
  if (ctx->ba_aux_copy_pending) {  // an earlier call failed between its copy on the auxiliary stream and the wait for it
    VS_HIP(ctx, hipStreamSynchronize(ctx->aux_stream[0]));
    ctx->ba_aux_copy_pending = false;
  }
  arena A;
  A.base = (uint8_t*)ctx->d_ba.p;
  A.host = (uint8_t*)ctx->h_pin_big.p;
  ba_dev D;
  memset(&D, 0, sizeof D);
  // Observation arrays that the caller keeps in pinned memory (Context.pinned_empty / vs_host_alloc) and that need no
  // reordering are DMA-ed from where they lie, started NOW so that the transfer runs beside the host passes below; they
  // sit in front of the uploaded part of the arena.  (Large problems only: the three pointer queries cost microseconds.)
  const bool direct_obs = dev || (identity && T > 1 && !q.obs_info && vs_is_pinned(q.obs_uv) && vs_is_pinned(q.obs_pose) && vs_is_pinned(q.obs_point));
  size_t upload_begin = 0;
  if (direct_obs) {
    D.o_cam = A.take<int>(n_obs);
    D.o_pt = A.take<int>(n_obs);
    D.o_uv = A.take<double>(2 * (size_t)n_obs);
    A.off = (A.off + 255) & ~(size_t)255;
    upload_begin = A.off;
    VS_HIP(ctx, hipMemcpyAsync((void*)D.o_cam, q.obs_pose, sizeof(int) * (size_t)n_obs, hipMemcpyHostToDevice, s));
    VS_HIP(ctx, hipMemcpyAsync((void*)D.o_pt, q.obs_point, sizeof(int) * (size_t)n_obs, hipMemcpyHostToDevice, s));
    // (device-built structure: the image points -- two thirds of the bytes, and nothing the structure kernels read -- go up later,
    // on an auxiliary stream, BEHIND the small tables, so that the structure is built while they are on the bus)
    if (!dev) VS_HIP(ctx, hipMemcpyAsync((void*)D.o_uv, q.obs_uv, sizeof(double) * 2 * (size_t)n_obs, hipMemcpyHostToDevice, s));
  }
  D.n_poses = F;
  D.n_points = P;
  D.n_obs = n_obs;
  D.n_scale = q.n_scale;
  D.nfp = nfp;
  D.nfl = nfl;
  D.np = np;
  D.n_act = n_act;
  D.nb_pt = nb_pt;
  D.has_info = q.obs_info != nullptr;
  D.max_it = q.max_iterations;
  D.lds_slab = lds_slab;
  D.fx = q.fx;
  D.fy = q.fy;
  D.cx = q.cx;
  D.cy = q.cy;
  D.huber = q.huber_delta;
  D.dcs = q.dcs_phi;
  int *h_pose_slot, *h_pt_slot, *h_act = nullptr, *h_ptstart = nullptr, *h_ocam = nullptr, *h_opt = nullptr, *h_cstart = nullptr, *h_cobs = nullptr, *h_scp, *h_scc;
  double *h_uv = nullptr, *h_info = nullptr, *h_scm, *h_cam0, *h_pts0;
  D.pose_slot = A.take<int>(F, &h_pose_slot);
  D.pt_slot = A.take<int>(P, &h_pt_slot);
  // (device path: the structure arrays are carved behind the uploaded part, see below)
  if (!dev) {
    D.act_pt = A.take<int>(n_act, &h_act);
    D.pt_start = A.take<int>(n_act + 1, &h_ptstart);
  }
  if (!direct_obs) {
    D.o_cam = A.take<int>(n_obs, &h_ocam);
    D.o_pt = A.take<int>(n_obs, &h_opt);
  }
  int* h_cpt = nullptr;
  if (!dev) {
    D.cam_start = A.take<int>(nfp + 1, &h_cstart);
    D.cam_obs = A.take<int>(n_cam_obs, &h_cobs);
    D.cam_pt = A.take<int>(n_cam_obs, &h_cpt);  // point index of the same entry (saves the camera role one dependent load)
  }
  D.sc_parent = A.take<int>(q.n_scale, &h_scp);
  D.sc_child = A.take<int>(q.n_scale, &h_scc);
  int* h_sp;
  D.slot_pose = A.take<int>(nfp, &h_sp);
  for (int i = 0; i < F; ++i)
    if (pose_slot[i] >= 0) h_sp[pose_slot[i]] = i;
  int *h_ohpl = nullptr, *h_fps = nullptr, *h_fpl = nullptr;
  unsigned long long* h_mask = nullptr;
  if (!dev) {
    D.o_hpl = A.take<int>(n_obs, &h_ohpl);
    D.fp_start = A.take<int>(nfl + 1, &h_fps);
    D.fp_slot = A.take<int>(n_hpl, &h_fpl);
    if (tiled_possible) D.fp_mask = A.take<unsigned long long>(nfl, &h_mask);
  }
  if (!direct_obs) D.o_uv = A.take<double>(2 * (size_t)n_obs, &h_uv);
  if (D.has_info) D.o_info = A.take<double>(3 * (size_t)n_obs, &h_info);
  D.sc_meas = A.take<double>(q.n_scale, &h_scm);
  double *h_mx = nullptr, *h_muv = nullptr, *h_minfo = nullptr;
  mo_state* h_mst = nullptr;
  mo_state* d_mst = nullptr;
  if (motion_only) {
    D.mo_X = A.take<double>(3 * (size_t)n_cam_obs, &h_mx);
    D.mo_uv = A.take<double>(2 * (size_t)n_cam_obs, &h_muv);
    if (D.has_info) D.mo_info = A.take<double>(3 * (size_t)n_cam_obs, &h_minfo);
    unsigned char* h_raw;
    d_mst = reinterpret_cast<mo_state*>(A.take<unsigned char>(256, &h_raw));  // the two records
    h_mst = reinterpret_cast<mo_state*>(h_raw);
    memset(h_raw, 0, 256);
    unsigned long long* h_box;
    D.mo_box = A.take<unsigned long long>(2 * (size_t)kMoPersistCameras * 8, &h_box);  // uploaded as zeros: no word carries a tag
    memset(h_box, 0, sizeof(unsigned long long) * 2 * kMoPersistCameras * 8);
    D.mo_epoch = 1;
  }
  D.cam[0] = A.take<double>((size_t)F * kCamStride, &h_cam0);
  D.pts[0] = A.take<double>(3 * (size_t)P, &h_pts0);
  lm_state* h_st;
  D.st = A.take<lm_state>(1, &h_st);
  unsigned* h_ticket;
  D.trial_ticket = A.take<unsigned>(4, &h_ticket);
  memset(h_ticket, 0, 4 * sizeof(unsigned));
  unsigned* h_cticket;
  D.cam_ticket = A.take<unsigned>(nfp, &h_cticket);
  memset(h_cticket, 0, sizeof(unsigned) * (size_t)(nfp ? nfp : 1));
  D.cam_split = cam_split;
  if (A.off > ctx->d_ba.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error", "vs_ba_solve");

  lap(0);  // reserve + carve
  // scale edges, start state (both state buffers), LM record
  auto fill_states = [&]() {
    for (int k = 0; k < q.n_scale; ++k) {
      h_scp[k] = q.scale_parent[k];
      h_scc[k] = q.scale_child[k];
      h_scm[k] = q.scale_meas[k];
    }
    for (int i = 0; i < F; ++i) {
      const double* m = q.poses + 16 * (size_t)i;
      double* c = h_cam0 + (size_t)i * kCamStride;
      c[0] = m[3];
      c[1] = m[7];
      c[2] = m[11];
      quat_from_pose(m, c + 3);
      quat_to_w2n(c, c + 3, c + 7);
    }
    memcpy(h_pts0, q.points, sizeof(double) * 3 * (size_t)P);
    memset(h_st, 0, sizeof(lm_state));
    h_st->need_lin = 1;
    h_st->ni = 2.0;
  };
  // ---- pass 2 over the observations in point order, straight into the pinned arena: the active points, the sorted
  // observation records, the Hpl blocks (observations whose point AND camera are free, stored contiguously per free
  // point), the per-camera lists, the camera-tile mask of every free point (tiled Schur) and duplicate cameras per point
  memcpy(h_pose_slot, pose_slot, sizeof(int) * F);
  memcpy(h_pt_slot, pt_slot, sizeof(int) * P);
  if (!dev) memcpy(h_cstart, cam_start, sizeof(int) * ((size_t)nfp + 1));
  // the observation records of the identity case are plain copies (48 MB at 2 000 000 observations): split over the threads
  // ... unless the caller keeps them in pinned memory (Context.pinned_empty / vs_host_alloc): then they are DMA-ed from
  // where they lie, behind the arena upload (large problems only: the three pointer queries cost a few microseconds)
  if (identity && !direct_obs) {
    par_for(T, [&](int t, int nt) {
      const size_t a = (size_t)((long long)n_obs * t / nt), b = (size_t)((long long)n_obs * (t + 1) / nt);
      memcpy(h_ocam + a, q.obs_pose + a, sizeof(int) * (b - a));
      memcpy(h_opt + a, q.obs_point + a, sizeof(int) * (b - a));
      memcpy(h_uv + 2 * a, q.obs_uv + 2 * a, sizeof(double) * 2 * (b - a));
      if (h_info) memcpy(h_info + 3 * a, q.obs_info + 3 * a, sizeof(double) * 3 * (b - a));
    });
  }
  lap(1);  // slot tables, observation records
  int mmax = 1, dups = 0, a_idx = 0, k_hpl = 0, max_rank = 0;
  double *d_out_pre = nullptr, *h_out_pre = nullptr;  // device-built structure: the read-back block, carved early (see there)
  auto t_win0 = now();
  bool win = false;
  int win_n = 0, win_per = 0, ns_win = 0, win_cams = 0;
  size_t upload_bytes = 0;
  if (dev) {
    // ---- the structure on the device: the small tables and the start state go up now, the structure arrays are carved behind
    // them and produced from the device copy of the observation list (vs_ba_build.hip); one 64-byte read-back tells what came out
    fill_states();
    upload_bytes = A.off;
    VS_HIP(ctx, hipMemcpyAsync(A.base + upload_begin, A.host + upload_begin, upload_bytes - upload_begin, hipMemcpyHostToDevice, s));
    if (!ctx->ev_after) VS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_after, hipEventDisableTiming));
    VS_HIP(ctx, hipMemcpyAsync((void*)D.o_uv, q.obs_uv, sizeof(double) * 2 * (size_t)n_obs, hipMemcpyHostToDevice, ctx->aux_stream[0]));
    VS_HIP(ctx, hipEventRecord(ctx->ev_after, ctx->aux_stream[0]));
    ctx->ba_aux_copy_pending = true;
    ba_build B;
    memset(&B, 0, sizeof B);
    // what comes BACK into the pinned mirror sits right behind the uploaded part: the read-back block of ba_export and the flag words
    d_out_pre = A.take<double>(out_elems_early, &h_out_pre);
    int* h_binfo;
    B.info = A.take<int>(kBuildInfoInts, &h_binfo);
    if (A.off > ctx->h_pin_big.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error (pinned mirror)", "vs_ba_solve");
    B.o_cam = D.o_cam;
    B.o_pt = D.o_pt;
    B.pose_slot = D.pose_slot;
    B.pt_slot = D.pt_slot;
    B.n_obs = n_obs;
    B.F = F;
    B.P = P;
    B.nfp = nfp;
    B.nfl = nfl;
    B.tile_cams = kTileCams;
    B.win_target = win_target;
    B.win_per_tune = ctx->tune.win_per;
    B.win_per_max = kWinPerMax;
    B.ns_cap = ns_win_bound;
    B.act_pt = A.take<int>(P);
    B.pt_start = A.take<int>((size_t)P + 1);
    B.cam_start = A.take<int>((size_t)nfp + 1);
    B.cam_obs = A.take<int>(n_obs);
    B.cam_pt = A.take<int>(n_obs);
    B.o_hpl = A.take<int>(n_obs);
    B.fp_start = A.take<int>((size_t)nfl + 1);
    B.fp_slot = A.take<int>(n_obs);
    if (tiled_possible) B.fp_mask = A.take<unsigned long long>(nfl);
    if (win_possible && ns_win_bound > 0) {
      B.win_order = A.take<int>(nfl);
      B.win_w0 = A.take<int>(ns_win_bound);
      B.win_len = A.take<int>(ns_win_bound);
      B.win_first = A.take<int>((size_t)nfp + 1);
    }
    const size_t nblk_o = ((size_t)n_obs + 1023) / 1024, nblk_p = ((size_t)nfl + 1023) / 1024, ncols = (size_t)nfp + 2;
    B.ckey = A.take<int>(n_obs);
    B.wkey = A.take<int>(nfl);
    B.wlo = A.take<int>(nfl);
    B.whi = A.take<int>(nfl);
    B.hist_o = A.take<int>(nblk_o * ncols);
    B.tot_o = A.take<int>(ncols);
    B.hist_p = A.take<int>(nblk_p * ncols);
    B.tot_p = A.take<int>(ncols);
    B.win_start = A.take<int>(ncols);
    if (A.off > ctx->d_ba.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error", "vs_ba_solve");
    VS_TRY(ba_build_enqueue(ctx, s, B));
    VS_HIP(ctx, hipMemcpyAsync(h_binfo, B.info, sizeof(int) * kBuildInfoInts, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
    VS_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_after, 0));  // everything enqueued behind this point may read the image points
    ctx->ba_aux_copy_pending = false;
    lap(2);
    if (h_binfo[kBuildBad] | h_binfo[kBuildUngrouped] | h_binfo[kBuildInactive] | h_binfo[kBuildDups]) {
      dev = false;  // not this path's case: the host passes take it (and report what is wrong with it, if anything)
      goto host_passes;
    }
    D.act_pt = B.act_pt;
    D.pt_start = B.pt_start;
    D.cam_start = B.cam_start;
    D.cam_obs = B.cam_obs;
    D.cam_pt = B.cam_pt;
    D.o_hpl = B.o_hpl;
    D.fp_start = B.fp_start;
    D.fp_slot = B.fp_slot;
    D.fp_mask = B.fp_mask;
    n_hpl = h_binfo[kBuildHpl];
    mmax = std::max(1, h_binfo[kBuildMmax]);
    cam_split = std::min(8, std::max(1, (h_binfo[kBuildCamMax] + kCamThreads - 1) / kCamThreads));
    D.cam_split = cam_split;
    if (B.win_order) {
      win_n = h_binfo[kBuildWinN];
      win_per = h_binfo[kBuildWinPer];
      ns_win = h_binfo[kBuildWinSlabs];
      win_cams = h_binfo[kBuildWinCams];
      if (ns_win > ns_win_bound || sizeof(int) * 2 * (size_t)ns_win > 48 * 1024) ns_win = 0;  // as below
      win = win_n > 0 && ns_win > 0 && win_cams <= kWinCams;
      D.win_order = B.win_order;
      D.win_w0 = B.win_w0;
      D.win_len = B.win_len;
      D.win_first = B.win_first;
    }
  }
  if (!dev) {
  // one point: its active-point record, its sorted observation records, its Hpl blocks and the per-camera lists.  The
  // running positions (active index, Hpl block index, per-camera fill positions) are the caller's: sequential for small
  // problems; for large ones every thread takes a range of points whose starting positions a counting pass fixed.
  // (captures BY VALUE: with the arena pointers captured by reference every store through an int* could alias them and they
  // were re-read from the closure after each one -- arena fill 43 -> 69 us at cfg4)
  // lowest / highest camera slot of every free point (banded-window plan below)
  int *wlo = nullptr, *whi = nullptr;
  if (tiled_possible && ntile > 1 && ctx->tune.schur_variant != 3) {
    W.wlo.resize((size_t)nfl);
    W.whi.resize((size_t)nfl);
    wlo = W.wlo.data();
    whi = W.whi.data();
  }
  auto do_point = [=](int j, int& a_i, int& k_h, int* cf, int* seen, int& mmax_, int& dups_) {
    const int i0 = cnt[j], i1 = cnt[j + 1], ls = pt_slot[j];
    if (i0 == i1 && ls < 0) return;
    h_act[a_i] = j;
    h_ptstart[a_i] = i0;
    ++a_i;
    int mf = 0, lo = INT_MAX, hi = -1;
    unsigned long long mask = 0ull;
    if (ls >= 0) h_fps[ls] = k_h;
    for (int i = i0; i < i1; ++i) {
      const int o = identity ? i : order[i];
      const int cam = q.obs_pose[o];
      const int cs = pose_slot[cam];
      if (!identity) {
        h_ocam[i] = cam;
        h_opt[i] = j;
        h_uv[2 * i] = q.obs_uv[2 * (size_t)o];
        h_uv[2 * i + 1] = q.obs_uv[2 * (size_t)o + 1];
        if (h_info) {
          h_info[3 * i] = q.obs_info[3 * (size_t)o];
          h_info[3 * i + 1] = q.obs_info[3 * (size_t)o + 1];
          h_info[3 * i + 2] = q.obs_info[3 * (size_t)o + 2];
        }
      }
      int blk = -1;
      if (cs >= 0) {
        h_cpt[cf[cs]] = j;
        h_cobs[cf[cs]++] = i;
        if (ls >= 0) {
          ++mf;
          dups_ |= seen[cam] == j;  // same camera twice: the ordered-rounds path of ba_schur
          seen[cam] = j;
          blk = k_h;
          h_fpl[k_h++] = cs;
          mask |= 1ull << ((cs / kTileCams) & 63);
          lo = std::min(lo, cs);
          hi = std::max(hi, cs);
        }
      }
      h_ohpl[i] = blk;
    }
    if (ls >= 0) {
      mmax_ = std::max(mmax_, mf);
      if (h_mask) h_mask[ls] = mask;
      if (wlo) {
        wlo[ls] = lo;  // INT_MAX / -1: seen from fixed cameras only, contributes nothing
        whi[ls] = hi;
      }
    }
  };
  if (T == 1) {
    W.cfill.assign(cam_start, cam_start + nfp + 1);
    W.seen_by.assign(F ? F : 1, -1);  // seen_by[camera] = last free point with an observation from that free camera
    for (int j = 0; j < P; ++j) do_point(j, a_idx, k_hpl, W.cfill.data(), W.seen_by.data(), mmax, dups);
  } else {
    // counting pass per range of points: active points, Hpl blocks, observations per free camera
    struct part2 {
      std::vector<int> cam, seen;
      int act = 0, hpl = 0, mmax = 1, dups = 0;
    };
    const int nchunk = 4 * T;  // handed out through a counter, as in pass 1
    std::vector<part2> parts((size_t)nchunk);
    std::atomic<int> next2{0};
    par_for(T, [&](int, int) {
      const int* const cnt_ = cnt;
      const int* const pt_slot_ = pt_slot;
      const int* const pose_slot_ = pose_slot;
      const int* const op = q.obs_pose;
      const int* const order_ = order;
      const bool identity_ = identity;
      const int P_ = P, nfp_ = nfp;
      for (int ch = next2.fetch_add(1); ch < nchunk; ch = next2.fetch_add(1)) {
        part2& R = parts[(size_t)ch];
        R.cam.assign((size_t)nfp_ + 1, 0);
        int* const rcam = R.cam.data();
        const int j0 = (int)((long long)P_ * ch / nchunk), j1 = (int)((long long)P_ * (ch + 1) / nchunk);
        int act = 0, hpl = 0;
        for (int j = j0; j < j1; ++j) {
          const int i0 = cnt_[j], i1 = cnt_[j + 1], ls = pt_slot_[j];
          if (i0 == i1 && ls < 0) continue;
          ++act;
          for (int i = i0; i < i1; ++i) {
            const int cs = pose_slot_[op[identity_ ? i : order_[i]]];
            if (cs >= 0) {
              rcam[cs]++;
              hpl += ls >= 0;
            }
          }
        }
        R.act = act;
        R.hpl = hpl;
      }
    });
    // starting positions of every range (exclusive prefix over the ranges; per camera on top of the camera's own start)
    int act0 = 0, hpl0 = 0;
    std::vector<int> cam_run(cam_start, cam_start + nfp + 1);
    for (part2& R : parts) {
      const int a = R.act, h = R.hpl;
      R.act = act0;
      R.hpl = hpl0;
      act0 += a;
      hpl0 += h;
      for (int c = 0; c < nfp; ++c) {
        const int k = R.cam[(size_t)c];
        R.cam[(size_t)c] = cam_run[(size_t)c];
        cam_run[(size_t)c] += k;
      }
    }
    std::atomic<int> next3{0};
    par_for(T, [&](int, int) {
      const int P_ = P, F_ = F;
      for (int ch = next3.fetch_add(1); ch < nchunk; ch = next3.fetch_add(1)) {
        part2& R = parts[(size_t)ch];
        R.seen.assign(F_ ? F_ : 1, -1);
        const int j0 = (int)((long long)P_ * ch / nchunk), j1 = (int)((long long)P_ * (ch + 1) / nchunk);
        int a_i = R.act, k_h = R.hpl, mm = 1, dd = 0;
        int* const cf = R.cam.data();
        int* const seen = R.seen.data();
        for (int j = j0; j < j1; ++j) do_point(j, a_i, k_h, cf, seen, mm, dd);
        R.mmax = mm;
        R.dups = dd;
      }
    });
    a_idx = act0;
    k_hpl = hpl0;
    for (const part2& R : parts) {
      mmax = std::max(mmax, R.mmax);
      dups |= R.dups;
    }
  }
  lap(2);  // per-point pass
  h_ptstart[n_act] = n_obs;
  h_fps[nfl] = k_hpl;
  if (dups) {  // rare: per Hpl block, how many earlier blocks of the same point belong to the same camera
    int* h_fpr;
    D.fp_rank = A.take<int>(n_hpl, &h_fpr);
    if (A.off > ctx->d_ba.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error", "vs_ba_solve");
    W.seen_by.assign(F ? F : 1, -1);
    W.seen_cnt.assign(F ? F : 1, 0);
    int* seen_by = W.seen_by.data();
    int* seen_cnt = W.seen_cnt.data();
    for (int a = 0; a < n_act; ++a)
      for (int i = h_ptstart[a]; i < h_ptstart[a + 1]; ++i) {
        if (h_ohpl[i] < 0) continue;
        const int cam = identity ? q.obs_pose[i] : h_ocam[i];
        seen_cnt[cam] = seen_by[cam] == h_act[a] ? seen_cnt[cam] + 1 : 0;
        seen_by[cam] = h_act[a];
        h_fpr[h_ohpl[i]] = seen_cnt[cam];
        max_rank = std::max(max_rank, seen_cnt[cam]);
      }
  }
  // ---- banded window?  (several tiles, no duplicates): the contributing free points ordered by their lowest camera slot
  // and cut into slabs; taken when the cameras of every slab span at most kWinCams slots (ba_schur_window)
  lap(3);
  t_win0 = now();
  if (tiled_possible && !dups && ntile > 1 && ctx->tune.schur_variant != 3) {
    W.wcnt.assign((size_t)nfp + 1, 0);
    int* const wcnt = W.wcnt.data();
    for (int l = 0; l < nfl; ++l)
      if (whi[l] >= 0) ++wcnt[wlo[l] + 1];
    for (int c = 0; c < nfp; ++c) wcnt[c + 1] += wcnt[c];
    win_n = wcnt[nfp];
    if (win_n > 0) {
      int* h_word;
      D.win_order = A.take<int>(nfl, &h_word);
      for (int l = 0; l < nfl; ++l)
        if (whi[l] >= 0) h_word[wcnt[wlo[l]]++] = l;  // stable: equal keys stay in point order
      // two workgroups per CU (registers), all resident at once; at least 32 points each
      const int target = win_target;
      win_per = std::min(kWinPerMax, std::max(32, (win_n + target - 1) / target));
      if (ctx->tune.win_per > 0) win_per = std::min(kWinPerMax, ctx->tune.win_per);
      ns_win = (win_n + win_per - 1) / win_per;
      if (sizeof(int) * 2 * (size_t)ns_win > 48 * 1024) ns_win = 0;  // ba_reduce_window keeps the slab windows in LDS: beyond ~3 million points the tile path
      int *h_w0, *h_wl, *h_wf;
      if (ns_win == 0) win_n = 0;
      D.win_w0 = A.take<int>(ns_win, &h_w0);
      D.win_len = A.take<int>(ns_win, &h_wl);
      D.win_first = A.take<int>(nfp + 1, &h_wf);
      if (A.off > ctx->d_ba.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error", "vs_ba_solve");
      win = ns_win > 0;
      for (int sl = 0; sl < ns_win && win; ++sl) {
        const int a = sl * win_per, b = std::min(a + win_per, win_n);
        const int lo = wlo[h_word[a]];  // sorted by it
        int hi = lo;
        for (int i = a; i < b; ++i) hi = std::max(hi, whi[h_word[i]]);
        h_w0[sl] = lo;
        h_wl[sl] = hi - lo + 1;
        win = hi - lo + 1 <= kWinCams;
        win_cams = std::max(win_cams, hi - lo + 1);
      }
      for (int c = 0, sl = 0; c <= nfp; ++c) {  // first slab that starts at camera c or later
        while (sl < ns_win && h_w0[sl] < c) ++sl;
        h_wf[c] = sl;
      }
    }
  }
  lap(4);  // window plan
  upload_bytes = A.off;
  }  // host passes
  const double win_plan_us = dev ? 0.0 : us(t_win0, now());
  ctx->ba_structure_dev = dev;
  const bool tiled = tiled_possible && !dups;
  const bool small = tiled && small_possible;
  // two linearisations, indexed like the state buffers: the trial kernel linearises the trial state's points, so an accepted step
  // needs no point linearisation of its own.  Single-tile windows (cfg4), and since round 4 large banded windows as well -- there
  // the point linearisation is a launch of 0.15 ms per accepted step (the scaled run), the second set of blocks is 300 MB of 288 GB
  const bool spec_win = win && nb_pt >= 1024 && ctx->tune.schur_variant == 0;
  const bool spec = (small && ctx->tune.schur_variant != 2) || spec_win;
  int ns = slabs_for(tiled, small);
  // ba_schur_small with camera workgroups beside it: keep the whole grid within one workgroup per CU.  Two Schur
  // workgroups on one CU take turns at its LDS in the product phase (measured 14-15 us per launch at 286 workgroups on
  // 256 CUs against 10 us alone); a workgroup with a few more points costs less than that.
  if (spec && !spec_win) ns = std::max(1, std::min(ns, std::max(ctx->prop.multiProcessorCount, 64) - nfp * cam_split));
  if (win) {
    ns = ns_win;
    D.win = 1;
    D.win_per = win_per;
    D.win_n = win_n;
    if (q.n_scale == 0) D.band = 6 * win_cams;  // scale edges couple arbitrary camera pairs in Hpp: dense factorisation then
  }
  D.ns = ns;
  D.mmax = mmax;
  D.dups = dups;
  D.max_rank = max_rank;
  if (tiled) {
    D.ntile = ntile;
    D.small = small;
  }
  // not uploaded
  D.Dinv = A.take<double>(9 * (size_t)nfl);
  D.cam[1] = A.take<double>((size_t)F * kCamStride);
  D.pts[1] = A.take<double>(3 * (size_t)P);
  D.Hpp = A.take<double>((size_t)np * np);
  D.bp = A.take<double>(np);
  D.Hll = A.take<double>(9 * (size_t)nfl);
  D.bl = A.take<double>(3 * (size_t)nfl);
  D.Hpl = A.take<double>(18 * (size_t)n_hpl);
  D.cam_part = A.take<double>(27 * (size_t)nfp * cam_split);
  if (spec) {
    D.spec = 1;
    D.Hpp1 = A.take<double>((size_t)np * np);
    D.bp1 = A.take<double>(np);
    D.Hll1 = A.take<double>(9 * (size_t)nfl);
    D.bl1 = A.take<double>(3 * (size_t)nfl);
    D.Hpl1 = A.take<double>(18 * (size_t)n_hpl);
  }
  D.slab = win ? A.take<double>((size_t)ns * kWinSlabElems) : A.take<double>((size_t)(ns ? ns : 1) * slab_elems);
  D.S = A.take<double>((size_t)np * np);
  D.bs = A.take<double>(np);
  D.xp = A.take<double>(np);
  D.rinv = A.take<double>(np);
  if (tiled) D.Dbl = A.take<double>(3 * (size_t)nfl);
  D.chol_fail = A.take<int>(4);
  D.part_chi = A.take<double>(nb_pt);
  D.part_scale = A.take<double>(nb_pt);
  D.part_maxd = A.take<double>((size_t)nb_pt + nfp);
  D.chi_trace = A.take<double>(q.max_iterations);
  D.lambda_trace = A.take<double>(q.max_iterations);
  const int trial_cap = res->trial_trace && res->trial_trace_cap > 0 ? res->trial_trace_cap : 0;
  if (trial_cap) {
    D.trial_trace = A.take<double>(4 * (size_t)trial_cap);
    D.trial_cap = trial_cap;
  }
  if (motion_only) {
    D.mo_part = A.take<double>(8 * (size_t)nfp);
    D.mo_H = A.take<double>(42 * (size_t)nfp);
  }
  const size_t out_elems = 32 + (size_t)F * kCamStride + 3 * (size_t)P + 2 * (size_t)q.max_iterations + 4 * (size_t)trial_cap;
  double* h_out = h_out_pre;
  double* d_out = d_out_pre ? d_out_pre : A.take<double>(out_elems, &h_out);  // ba_export's block and its pinned landing place
  if (A.off > ctx->d_ba.cap) return vs_fail(ctx, VS_ENOMEM, "%s: internal arena sizing error", "vs_ba_solve");

  if (motion_only) {
    memset(h_mst, 0, 2 * sizeof(mo_state));
    h_mst[1].need_lin = 1;  // step 0 reads the state of parity 1
    h_mst[1].ni = 2.0;
    for (int i = 0; i < n_cam_obs; ++i) {
      const int o = identity ? h_cobs[i] : order[h_cobs[i]];
      const double* X = q.points + 3 * (size_t)q.obs_point[o];
      h_mx[3 * i] = X[0];
      h_mx[3 * i + 1] = X[1];
      h_mx[3 * i + 2] = X[2];
      h_muv[2 * i] = q.obs_uv[2 * (size_t)o];
      h_muv[2 * i + 1] = q.obs_uv[2 * (size_t)o + 1];
      if (h_minfo) {
        h_minfo[3 * i] = q.obs_info[3 * (size_t)o];
        h_minfo[3 * i + 1] = q.obs_info[3 * (size_t)o + 1];
        h_minfo[3 * i + 2] = q.obs_info[3 * (size_t)o + 2];
      }
    }
  }
  if (!dev) fill_states();
  const bool nothing = (np + 3 * nfl == 0) || q.max_iterations == 0;
  lap(5);  // states
  const auto t_filled = now();
  if (!dev) VS_HIP(ctx, hipMemcpyAsync(A.base + upload_begin, A.host + upload_begin, upload_bytes - upload_begin, hipMemcpyHostToDevice, s));
  // both state buffers start identical (fixed cameras / points are never rewritten in the trial buffer's points)
  VS_HIP(ctx, hipMemcpyAsync(D.cam[1], D.cam[0], sizeof(double) * (size_t)F * kCamStride, hipMemcpyDeviceToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(D.pts[1], D.pts[0], sizeof(double) * 3 * (size_t)P, hipMemcpyDeviceToDevice, s));

  // ---- kernels
  const int schur_per = ns > 0 ? (nfl + ns - 1) / ns : 0;
  const size_t schur_lds = sizeof(double) * ((lds_slab ? slab_elems : 0) + 12 * (size_t)schur_per + 36 * (size_t)mmax) + sizeof(int) * (2 * (size_t)mmax + 2) + 16;
  const size_t tile_lds = sizeof(double) * (2 * kTileBatch * 12 + 2 * kTileBatch * kTileCams * 18) +
                          sizeof(int) * (3 * kTileBatch * 32 + 3 * kTileChunk + 8) + 64;
  if (tiled) VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_schur_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
  if (!tiled && schur_lds > 64 * 1024) {
    if (lds_slab) VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_schur<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_lds));
    else VS_HIP(ctx, hipFuncSetAttribute((const void*)ba_schur<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_lds));
  }
  solve_plan splan;
  VS_TRY(plan_solve(ctx, np, &splan));
  // which kernels this solve takes (vs_ba_last_path: tests of real-data problems assert the path they exercise)
  ctx->ba_path[0] = motion_only ? 4 : win ? 3 : small ? 2 : tiled ? 1 : 0;
  ctx->ba_path[1] = motion_only ? 4 : splan.lds ? 0 : (splan.band_ok && D.band > 0 && D.band <= kBandMax) ? 1 : splan.nbw > 0 ? 2 : 3;
  ctx->ba_path[2] = np;
  ctx->ba_path[3] = D.band;
  ctx->ba_path[4] = ntile;
  ctx->ba_path[5] = win ? win_cams : 0;
  if (!tiled && schur_lds > 160 * 1024) return vs_fail(ctx, VS_EINVAL, "%s: a point is observed by too many free cameras for the LDS staging", "vs_ba_solve");

  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, sizeof(lm_state) + sizeof(mo_state) + 128));
  lm_state* hst = reinterpret_cast<lm_state*>((uint8_t*)ctx->h_pin.p + 128);
  // one copy + one synchronisation bring back the LM state and, if the solve is over, everything else the host wants
  bool exported = false;
  auto enqueue_export = [&]() -> int {
    const unsigned blocks = (unsigned)std::min<size_t>((out_elems + 255) / 256, 1024);
    hipLaunchKernelGGL(ba_export, dim3(blocks), dim3(256), 0, s, D, d_out);
    VS_LAUNCH_CHECK(ctx, "ba_export");
    VS_HIP(ctx, hipMemcpyAsync(h_out, d_out, sizeof(double) * out_elems, hipMemcpyDeviceToHost, s));
    return VS_OK;
  };
  auto export_and_fetch = [&]() -> int {
    VS_TRY(enqueue_export());
    VS_HIP(ctx, hipStreamSynchronize(s));
    memcpy(hst, h_out, sizeof(lm_state));
    exported = true;
    return VS_OK;
  };
  auto launch_slot = [&](bool first) -> int {
    if (first || !spec) {  // spec: later states are linearised inside ba_point_trial (points) and ba_schur_small (cameras)
      if (nb_pt >= 1024) {  // large problems: the two roles as launches of their own (see ba_linearize_points)
        hipLaunchKernelGGL(ba_linearize_points, dim3(nb_pt), dim3(kPtThreads), 0, s, D);
        VS_LAUNCH_CHECK(ctx, "ba_linearize_points");
        if (nfp > 0) hipLaunchKernelGGL(ba_linearize_cameras, dim3(nfp * cam_split), dim3(kCamThreads), 0, s, D);
        VS_LAUNCH_CHECK(ctx, "ba_linearize_cameras");
      } else {
        hipLaunchKernelGGL(ba_linearize, dim3(nb_pt + nfp * cam_split), dim3(kCamThreads), 0, s, D);
        VS_LAUNCH_CHECK(ctx, "ba_linearize");
      }
    } else if (spec_win && nfp > 0) {  // the points of an accepted state were linearised by ba_point_trial: the camera role is all that is left
      hipLaunchKernelGGL(ba_linearize_cameras, dim3(nfp * cam_split), dim3(kCamThreads), 0, s, D);
      VS_LAUNCH_CHECK(ctx, "ba_linearize_cameras");
    }
    if (first) {
      hipLaunchKernelGGL(ba_lambda_init, dim3(1), dim3(64), 0, s, D);
      VS_LAUNCH_CHECK(ctx, "ba_lambda_init");
    }
    if (ns > 0 && tiled) {
      if (small) {
        hipLaunchKernelGGL(ba_schur_small, dim3(ns + (spec ? nfp * cam_split : 0)), dim3(kSmallThreads), 0, s, D, first ? 0 : 1);
        VS_LAUNCH_CHECK(ctx, "ba_schur_small");
      } else {
        if (ntile > 1) hipLaunchKernelGGL(ba_dinv, dim3((unsigned)((nfl + 255) / 256)), dim3(256), 0, s, D);  // single tile: fused
        if (win) {
          hipLaunchKernelGGL(ba_schur_window, dim3(ns), dim3(256), 0, s, D);
          VS_LAUNCH_CHECK(ctx, "ba_schur_window");
        } else {
          hipLaunchKernelGGL(ba_schur_tile, dim3(ns, ntile * (ntile + 1) / 2), dim3(256), tile_lds, s, D);  // lower triangle
          VS_LAUNCH_CHECK(ctx, "ba_schur_tile");
        }
      }
    } else if (ns > 0) {
      if (lds_slab) hipLaunchKernelGGL(ba_schur<true>, dim3(ns), dim3(kSchurThreads), schur_lds, s, D);
      else hipLaunchKernelGGL(ba_schur<false>, dim3(ns), dim3(kSchurThreads), schur_lds, s, D);
      VS_LAUNCH_CHECK(ctx, "ba_schur");
    }
    if (np > 0 && win) {
      hipLaunchKernelGGL(ba_reduce_window, dim3((unsigned)((slab_elems + kRedWinElems - 1) / kRedWinElems)), dim3(256), sizeof(int) * 2 * (size_t)ns, s, D);
      VS_LAUNCH_CHECK(ctx, "ba_reduce_window");
    } else if (np > 0) {
      hipLaunchKernelGGL(ba_reduce, dim3((unsigned)((slab_elems + 63) / 64)), dim3(64 * kRedSplit), 0, s, D);
      VS_LAUNCH_CHECK(ctx, "ba_reduce");
    }
    VS_TRY(launch_solve(ctx, s, D, splan));
    hipLaunchKernelGGL(ba_point_trial, dim3(nb_pt), dim3(kPtThreads), 0, s, D);  // its last workgroup decides
    VS_LAUNCH_CHECK(ctx, "ba_point_trial");
    return VS_OK;
  };

  if (motion_only && !nothing) {
    // one launch per LM trial (+1: iteration 0 is linearised on its own, lambda_0 needs all cameras); the launches are
    // predicated on the device-resident state, the host polls it once per batch
    ba_dev Dm = D;
    Dm.st = reinterpret_cast<lm_state*>(d_mst);
    mo_state* hms = reinterpret_cast<mo_state*>(ctx->h_pin.p);
    const int max_steps = 1 + q.max_iterations * 10;
    int step = 0;
    int cam_obs_max = 0;
    for (int c = 0; c < nfp; ++c) cam_obs_max = std::max(cam_obs_max, cam_start[c + 1] - cam_start[c]);
    bool persistent = vsba::mo_persistent_ok(ctx, nfp, max_steps);
    for (;;) {
      if (persistent) {  // the whole solve in one launch; the final record lands in both state slots
        if (cam_obs_max <= kMoPersistObs) hipLaunchKernelGGL(ba_motion_persistent<false>, dim3(nfp), dim3(kMoThreads), 0, s, Dm, max_steps);
        else hipLaunchKernelGGL(ba_motion_persistent<true>, dim3(nfp), dim3(kMoThreads), 0, s, Dm, max_steps);
        VS_LAUNCH_CHECK(ctx, "ba_motion_persistent");
        VS_HIP(ctx, hipMemcpyAsync(hms, d_mst, sizeof(mo_state), hipMemcpyDeviceToHost, s));
        VS_HIP(ctx, hipStreamSynchronize(s));
        if (hms->terminated != 3) break;
        // the camera workgroups did not meet (the device was shared with long-running foreign work, or is smaller than it
        // reports): restore the start state from the pinned arena mirror and run the launch-per-step form once instead
        VS_HIP(ctx, hipMemcpyAsync(A.base + upload_begin, A.host + upload_begin, upload_bytes - upload_begin, hipMemcpyHostToDevice, s));
        VS_HIP(ctx, hipMemcpyAsync(D.cam[1], D.cam[0], sizeof(double) * (size_t)F * kCamStride, hipMemcpyDeviceToDevice, s));
        VS_HIP(ctx, hipMemcpyAsync(D.pts[1], D.pts[0], sizeof(double) * 3 * (size_t)P, hipMemcpyDeviceToDevice, s));
        persistent = false;
        ctx->mo_persist_cap = 0;  // and do not try the one-launch form again on this context
      }
      const int batch = std::min(max_steps + 1 - step, q.max_iterations + 2);  // LIN + trials + the deciding launch
      for (int k = 0; k < batch; ++k, ++step) {
        hipLaunchKernelGGL(ba_motion_step, dim3(nfp), dim3(kMoThreads), 0, s, Dm, step);
        VS_LAUNCH_CHECK(ctx, "ba_motion_step");
      }
      VS_HIP(ctx, hipMemcpyAsync(hms, d_mst + ((step - 1) & 1), sizeof(mo_state), hipMemcpyDeviceToHost, s));
      VS_HIP(ctx, hipStreamSynchronize(s));
      if (hms->done || step > max_steps) break;
    }
    // translate to the common read-back record
    mo_state fin = *hms;
    hst->cur = fin.cur;
    hst->it = fin.it;
    hst->trials = fin.trials;
    hst->not_pd = fin.not_pd;
    hst->terminated = fin.terminated;
    hst->chi0 = fin.chi0;
    hst->current_chi = fin.current_chi;
    hst->lambda = fin.lambda;
    hst->done = fin.done;
  } else if (!nothing) {
    // slots are predicated on the device-resident LM state; the host only polls `done` after each batch
    const int max_slots = q.max_iterations * 10;
    int launched = 0;
    bool first = true;
    while (launched < max_slots) {
      const int batch = std::min(max_slots - launched, std::max(1, q.max_iterations));
      if (ctx->tune.ba_graph == 1) {
        // experiment (vs_tune_ba_graph, profiles/tried_and_dropped.md): the batch -- its slots, the export kernel and the
        // read-back -- captured from the stream and replayed as ONE hipGraph launch.  Capture + instantiation are host time
        // outside the measured interval; ba_batch_us = graph launch to completion, to be compared with the same interval of
        // the launch-by-launch form below.
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        VS_HIP(ctx, hipStreamSynchronize(s));
        VS_HIP(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int crc = VS_OK;
        for (int k = 0; k < batch && crc == VS_OK; ++k) {
          crc = launch_slot(first);
          first = false;
        }
        if (crc == VS_OK) crc = enqueue_export();
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        if (crc != VS_OK) return crc;
        if (ce != hipSuccess || !graph) return vs_fail(ctx, VS_EHIP, "%s: stream capture of the trial batch failed: %s", "vs_ba_solve", hipGetErrorString(ce));
        VS_HIP(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        const auto tg0 = now();
        VS_HIP(ctx, hipGraphLaunch(exec, s));
        VS_HIP(ctx, hipStreamSynchronize(s));
        ctx->ba_batch_us = us(tg0, now());
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        memcpy(hst, h_out, sizeof(lm_state));
        exported = true;
        launched += batch;
        if (hst->done) break;
        continue;
      }
      if (ctx->tune.ba_graph == 2) VS_HIP(ctx, hipStreamSynchronize(s));  // measurement only: the same interval as the graph form's
      const auto tb0 = now();  // (the uploads may still be in flight: the product does not wait for them before it enqueues)
      for (int k = 0; k < batch; ++k) {
        VS_TRY(launch_slot(first));
        first = false;
      }
      launched += batch;
      VS_TRY(export_and_fetch());
      ctx->ba_batch_us = us(tb0, now());
      if (hst->done) break;
    }
  } else {
    // chi2 of the start state only
    hipLaunchKernelGGL(ba_linearize, dim3(nb_pt + nfp * cam_split), dim3(kCamThreads), 0, s, D);
    VS_LAUNCH_CHECK(ctx, "ba_linearize");
    hipLaunchKernelGGL(ba_lambda_init, dim3(1), dim3(64), 0, s, D);
    VS_LAUNCH_CHECK(ctx, "ba_lambda_init");
    VS_TRY(export_and_fetch());
  }

  const auto t_solved = now();
  // ---- the accepted state: already in the export block, except on the motion-only path
  const int cur = hst->cur;
  double* h_cam = reinterpret_cast<double*>(A.host);  // motion-only: reuse the pinned mirror
  double* h_pts = h_cam + (size_t)F * kCamStride + 8;
  double* h_tr = h_pts + 3 * (size_t)P + 8;
  if (exported) {
    h_cam = h_out + 32;
    h_pts = h_cam + (size_t)F * kCamStride;
    h_tr = h_pts + 3 * (size_t)P;
    if (trial_cap) memcpy(res->trial_trace, h_tr + 2 * (size_t)q.max_iterations, sizeof(double) * 4 * (size_t)std::min(trial_cap, hst->trials));
  } else {
    VS_HIP(ctx, hipMemcpyAsync(h_cam, D.cam[cur], sizeof(double) * (size_t)F * kCamStride, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipMemcpyAsync(h_pts, D.pts[cur], sizeof(double) * 3 * (size_t)P, hipMemcpyDeviceToHost, s));
    if (q.max_iterations > 0) {
      VS_HIP(ctx, hipMemcpyAsync(h_tr, D.chi_trace, sizeof(double) * q.max_iterations, hipMemcpyDeviceToHost, s));
      VS_HIP(ctx, hipMemcpyAsync(h_tr + q.max_iterations, D.lambda_trace, sizeof(double) * q.max_iterations,
                                 hipMemcpyDeviceToHost, s));
    }
    if (trial_cap)
      VS_HIP(ctx, hipMemcpyAsync(res->trial_trace, D.trial_trace, sizeof(double) * 4 * (size_t)std::min(trial_cap, hst->trials),
                                 hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  if (timing)
    fprintf(stderr, "vs_ba_solve: structure %.1f us, arena fill %.1f us, upload + kernels %.1f us, read-back %.1f us (upload %zu B; window plan %.1f us inside the fill, %d slabs)\n",
            us(t_begin, t_struct), us(t_struct, t_filled), us(t_filled, t_solved), us(t_solved, now()), upload_bytes, win_plan_us, win ? ns : 0);
  if (timing && dev)
    fprintf(stderr, "  structure on the device: reserve + carve %.1f, slot tables %.1f, start state + DMA of the observation arrays + build kernels + read-back %.1f us\n", lap_us[0],
            lap_us[1], lap_us[2]);
  else if (timing)
    fprintf(stderr, "  fill: reserve + carve %.1f, tables + records %.1f, per-point pass %.1f, ranks %.1f, window plan %.1f, states %.1f us (%d threads)\n", lap_us[0],
            lap_us[1], lap_us[2], lap_us[3], lap_us[4], lap_us[5], T);
  res->iterations = hst->it;
  res->trials = hst->trials;
  res->not_pd = hst->not_pd;
  res->terminated = hst->terminated;
  res->chi2_initial = hst->chi0;
  res->chi2_final = hst->current_chi;
  res->lambda_final = hst->lambda;
  for (int i = 0; i < hst->it && i < q.max_iterations; ++i) {
    if (res->chi2_trace) res->chi2_trace[i] = h_tr[i];
    if (res->lambda_trace) res->lambda_trace[i] = h_tr[q.max_iterations + i];
  }
  if (res->poses_out)
    for (int i = 0; i < F; ++i) {
      const double* c = h_cam + (size_t)i * kCamStride;
      double* o = res->poses_out + 16 * (size_t)i;
      // R = (w2n rotation)^T
      for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) o[4 * r + k] = c[7 + 4 * k + r];
        o[4 * r + 3] = c[r];
      }
      o[12] = o[13] = o[14] = 0.0;
      o[15] = 1.0;
    }
  if (res->points_out) memcpy(res->points_out, h_pts, sizeof(double) * 3 * (size_t)P);
  return VS_OK;
}
