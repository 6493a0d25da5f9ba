// vs_pnp.hip -- PnP-RANSAC (gfx950).
#include "vs_ba_internal.h"

#include <type_traits>

using namespace vsba;

// The LM of the PnP kernels is a long dependent FP64 chain executed by one wave: fused multiply-adds halve it.  The
// library is built with -ffp-contract=off for the bundle adjustment (it rounds like the oracle, operation for
// operation); PnP-RANSAC only has to agree with its oracle to 1e-9 on the pose and exactly on the inlier set (tests/
// test_pnp.py), which contraction does not touch.
#pragma clang fp contract(fast)

namespace {

// ------------------------------------------------------------------------------------------------ PnP-RANSAC
// cv2.solvePnPRansac as the reference calls it (src/v2/main.py:196-197; useExtrinsicGuess, ITERATIVE, 100 iterations,
// 8 px, 0.99): hypothesis h refines the extrinsic guess on 5 sampled correspondences, inliers are counted over all
// points, RANSACUpdateNumIters shrinks the budget, the best model is refined on its inliers.
//   pnp_ransac_kernel, hypothesis waves: one wave per hypothesis.  The 5 sampled edges sit on lanes 0..4 of every group of 8
//     lanes; the 28 sums of a linearisation (21 H, 6 b, chi2) are butterfly-reduced inside the group, so every lane holds
//     the same normal equations and runs the same LM step redundantly in registers - no LDS, no broadcast.  Then the 64
//     lanes score the n points and a ballot counts the inliers.
//   pnp_ransac_kernel, finishing workgroup: replays the sequential budget rule over the per-hypothesis counts as they arrive
//     (which yields exactly the sequential algorithm's winner), lists the winner's inliers in order and refines the pose on
//     them with the same cooperative LM (edges strided over 256 threads).  One launch for all of it.


__device__ inline int pnp_count(const pnp_args& P) { return P.n_dev ? *P.n_dev : P.n; }
// Coordinate k of object point i.  obj_f32: rounded to float32 first -- what the solver sees when the caller hands it
// objectPoints.astype(np.float32) as the reference does (src/v2/main.py:196) while the period's resident rows are float64.
__device__ __forceinline__ double pnp_obj(const pnp_args& P, const double* obj, size_t i, int k) {
  const double v = obj[3 * i + k];
  return P.obj_f32 ? (double)(float)v : v;
}
// What a launch works out for itself (the count read on the device, the rows behind the period's offset, the guess in the
// state buffer the previous solve ended on).  Kept apart from pnp_args ON PURPOSE: the argument struct is never written, so
// it stays in the kernel-argument segment -- written to, it was copied to scratch memory at kernel entry (736 bytes per lane),
// every field access became a scratch load, and a kernel with scratch cannot start on a queue before the runtime has
// provided it, which on a fresh process happened to wait for the very kernels this one waits for in chained tracking.
struct pnp_view {
  int n, lm_cur;
  const double *obj, *img;
  const double* guess;  // the record to start from, or nullptr: pnp_args::cam0
};
// bounded, sleeping poll of a tagged word (nullptr: nothing to wait for); acquire: what the publisher wrote is visible after it.
// The words carry sequence numbers that only grow (one word is shared by both buffer sets and every later front half / solve
// overwrites it), so the wait is for "at least this tag", wrap-safe: a later publication can never hide an earlier one.
// Bound: 2^15 polls of ~1.5 us (a sleep of 1024 cycles + the load) = ~50 ms -- what is waited for was enqueued BEFORE this launch
// and takes ~0.1 ms; a workgroup that gives up leaves, the host then redoes the frame host-paced (vs_track.hip, track_recover).
constexpr int kPnpWaitPolls = 1 << 15;
__device__ inline bool pnp_wait_tag(const unsigned* word, unsigned tag) {
  if (!word) return true;
  for (int it = 0; it < kPnpWaitPolls; ++it) {
    if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tag) >= 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one cache invalidation, when the word has arrived -- not one per poll
      return true;
    }
    __builtin_amdgcn_s_sleep(16);
  }
  return false;
}
// chained tracking: the state-buffer index of the previous solve and with it the guess (valid once that solve has ended)
__device__ inline void pnp_resolve_guess(const pnp_args& P, pnp_view& V) {
  const int cur = P.cur_dev->cur & 1;
  V.lm_cur = cur;
  V.guess = cur ? P.guess_dev[1] : P.guess_dev[0];  // (no run-time index into the argument struct: that would put it in scratch)
}
// class-API period: "the outcome is in pinned memory" (every thread's stores first, then the tag with a system-scope release)
__device__ inline void pnp_announce(const pnp_args& P) {
  if (!P.host_tag_word) return;  // uniform
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(P.host_tag_word, P.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (chained tracking: guess_dev[0] and lm_cur have been resolved from the device-side state by then)
__device__ inline double pnp_guess(const pnp_args& P, const pnp_view& V, int k) { return V.guess ? V.guess[k] : P.cam0[k]; }

__device__ inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ inline void pnp_err(const pnp_args& P, const double* cam, const double* X, const double* uv, double& eu,
                               double& ev, double* pc) {
  const double* w = cam + 7;
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = w[4 * i] * X[0] + w[4 * i + 1] * X[1] + w[4 * i + 2] * X[2] + w[4 * i + 3];
  const double iz = vs_fast_rcp(pc[2]);
  eu = (P.fx * pc[0] + P.cx * pc[2]) * iz - uv[0];
  ev = (P.fy * pc[1] + P.cy * pc[2]) * iz - uv[1];
}

// one edge into acc[28] = H upper triangle row-major (21), b (6), chi2 (identity information, no robust kernel)
template <bool JAC>
__device__ inline void pnp_edge(const pnp_args& P, const double* cam, const double* X, const double* uv, double* acc) {
  double pc[3], eu, ev;
  pnp_err(P, cam, X, uv, eu, ev, pc);
  acc[27] += eu * eu + ev * ev;
  if (!JAC) return;
  const double* w = cam + 7;
  const double px = pc[0], py = pc[1], pz = pc[2];
  const double ipz2 = vs_fast_rcp(pz * pz);
  const double ipz2fx = ipz2 * P.fx, ipz2fy = ipz2 * P.fy;
  const double p0 = X[0] - cam[0], p1 = X[1] - cam[1], p2 = X[2] - cam[2];
  double r[3], J[2][6];
#pragma unroll
  for (int k = 0; k < 3; ++k) r[k] = w[4 * k] * p0 + w[4 * k + 1] * p1 + w[4 * k + 2] * p2;
  const double dpx[3] = {0.0, 2 * r[2], -2 * r[1]}, dpy[3] = {-2 * r[2], 0.0, 2 * r[0]}, dpz[3] = {2 * r[1], -2 * r[0], 0.0};
  J[0][3] = (pz * dpx[0] - px * dpx[2]) * ipz2fx;
  J[1][3] = (pz * dpx[1] - py * dpx[2]) * ipz2fy;
  J[0][4] = (pz * dpy[0] - px * dpy[2]) * ipz2fx;
  J[1][4] = (pz * dpy[1] - py * dpy[2]) * ipz2fy;
  J[0][5] = (pz * dpz[0] - px * dpz[2]) * ipz2fx;
  J[1][5] = (pz * dpz[1] - py * dpz[2]) * ipz2fy;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    J[0][a] = -((pz * w[a] - px * w[8 + a]) * ipz2fx);
    J[1][a] = -((pz * w[4 + a] - py * w[8 + a]) * ipz2fy);
  }
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int c = a; c < 6; ++c) acc[k++] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[21 + a] += J[0][a] * (-eu) + J[1][a] * (-ev);
}

__device__ inline void cam_apply(const double* src, const double* x, double* dst) {
  double t[3] = {src[0] + x[0], src[1] + x[1], src[2] + x[2]};
  const double bx = x[3], by = x[4], bz = x[5];
  const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));  // NaN for an oversized step, as in SBACam::update
  const double ax = src[3], ay = src[4], az = src[5], aw = src[6];
  const double w = aw * bw - ax * bx - ay * by - az * bz;
  const double xx = aw * bx + ax * bw + ay * bz - az * by;
  const double yy = aw * by + ay * bw + az * bx - ax * bz;
  const double zz = aw * bz + az * bw + ax * by - ay * bx;
  const double inrm = vs_fast_rsq(xx * xx + yy * yy + zz * zz + w * w);
  double q[4] = {xx * inrm, yy * inrm, zz * inrm, w * inrm};
  for (int k = 0; k < 3; ++k) dst[k] = t[k];
  for (int k = 0; k < 4; ++k) dst[3 + k] = q[k];
  quat_to_w2n(t, q, dst + 7);
}

// sum of v[0..NV) over the lanes of aligned groups of 2^STEPS lanes (butterfly: every lane of the group ends with the
// same bits), then - for workgroups of several waves - over the waves through LDS in wave order
// LDS scratch of the multi-wave reductions: rows of the transposed 28-vector reduction, group sums, totals
constexpr int kPnpRedRows = kPnpFinish * 29, kPnpRedDoubles = kPnpRedRows + 8 * 28 + 32;

template <int NV, int STEPS, int NWAVES>
__device__ inline void pnp_reduce(double* v, double* s_red) {
  if (NWAVES > 1 && NV == 28) {
    // workgroup-wide sum of 28-vectors through a transposed LDS layout: every thread stores its row, (value, group of
    // 32 rows) threads add their rows in row order, 28 threads add the group sums in group order, everybody reads the
    // totals -- fixed order, and far fewer cross-lane operations than 28 butterflies per wave
    constexpr int kGroups = NWAVES * 2;
    double* rows = s_red;
    double* grp = s_red + kPnpRedRows;
    double* tot = grp + 8 * 28;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __syncthreads();  // previous readers of the scratch are done
#pragma unroll
    for (int k = 0; k < NV; ++k) rows[tid * 29 + k] = v[k];
    __syncthreads();
    if (lane < 56) {
      const int k = lane % 28, g = 2 * wv + lane / 28;
      double a = 0.0;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) a += rows[(g * 32 + j) * 29 + k];
      grp[g * 28 + k] = a;
    }
    __syncthreads();
    if (tid < 28) {
      double a = grp[tid];
      for (int g = 1; g < kGroups; ++g) a += grp[g * 28 + tid];
      tot[tid] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = tot[k];
    return;
  }
  // xor-butterfly over aligned groups of 2^STEPS lanes with data-parallel-primitive moves instead of ds_bpermute round trips
  // (__shfl_xor): partner lane^1 and lane^2 by quad permutes; for lane^4 the half-row mirror (lane 7 - i of the group of 8, which
  // holds the same bits as lane i^4 once the quads are uniform), for lane^8 the row mirror likewise; across the rows of 16 the
  // four row values are read out (readlane) and added as the butterfly would -- (r0 + r1) + (r2 + r3), addition commutes --
  // so every lane ends with exactly the bits the __shfl_xor form gave it.
  auto dpp_add = [](double x, auto ctrl) {
    constexpr int kCtrl = decltype(ctrl)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), kCtrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), kCtrl, 0xF, 0xF, false);
    return x + __hiloint2double(hi, lo);
  };
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
    if (STEPS >= 1) x = dpp_add(x, std::integral_constant<int, 0xB1>());   // quad_perm [1,0,3,2]
    if (STEPS >= 2) x = dpp_add(x, std::integral_constant<int, 0x4E>());   // quad_perm [2,3,0,1]
    if (STEPS >= 3) x = dpp_add(x, std::integral_constant<int, 0x141>());  // row_half_mirror
    if (STEPS >= 4) x = dpp_add(x, std::integral_constant<int, 0x140>());  // row_mirror
    if (STEPS == 6) {
      const int lo = __double2loint(x), hi = __double2hiint(x);
      const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
      const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
      const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
      const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
      x = (r0 + r1) + (r2 + r3);
    }
    v[k] = x;
  }
  static_assert(STEPS == 3 || STEPS == 6, "groups of 8 lanes (hypotheses) or whole waves (finishing workgroup)");
  if (NWAVES > 1) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // previous readers of s_red are done
    if (lane == 0)
      for (int k = 0; k < NV; ++k) s_red[wv * NV + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double a = s_red[k];
      for (int w = 1; w < NWAVES; ++w) a += s_red[w * NV + k];
      v[k] = a;
    }
  }
}

// Levenberg-Marquardt as OptimizationAlgorithmLevenberg drives it (one camera, fixed points, no robust kernel) plus a
// stop once a solved step is small (|x|^2 < kPnpStep2 = 1e-14, i.e. |x| < 1e-7: OpenCV's iterative solver stops when the
// parameter change falls below FLT_EPSILON = 1.2e-7 relative), run redundantly by every thread on identical sums; at most
// max_it iterations (the final refinement: the caller's refine_iters; a hypothesis: kPnpHypIters of them -- a hypothesis
// only has to be good enough to count inliers at a threshold of pixels, and the slowest of the first few hypotheses is what
// the finishing workgroup waits for).  Edge e of the thread: e = first, first + stride, ... < m; sel maps to
// the correspondence (nullptr: identity).
// The thread's edges are fetched once and kept in registers when they fit (NREG per thread: always for the 5-point
// hypotheses, up to NREG * stride inliers in the final refinement): every LM iteration evaluates them twice, and from
// memory each evaluation starts with two dependent round trips (index, then point and pixel) on an otherwise idle CU --
// in-kernel stamps put the refinement at 5.7 us per iteration, mostly waiting for those.
template <int STEPS, int NWAVES, int NREG>
__device__ inline void pnp_lm(const pnp_args& P, const pnp_view& V, const int* sel, int m, int first, int stride, double* cam, double* s_red, int max_it) {
  double trial[kCamStride];
  double lambda = 0.0, ni = 2.0;
  const bool in_regs = m <= NREG * stride;  // uniform
  double rX[NREG][3], rUV[NREG][2];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
      const int e = first + j * stride;
      rX[j][0] = rX[j][1] = rX[j][2] = rUV[j][0] = rUV[j][1] = 0.0;
      if (e < m) {
        const int i = sel ? sel[e] : e;
#pragma unroll
        for (int k = 0; k < 3; ++k) rX[j][k] = pnp_obj(P, V.obj, (size_t)i, k);
        rUV[j][0] = V.img[2 * (size_t)i];
        rUV[j][1] = V.img[2 * (size_t)i + 1];
      }
    }
  }
  for (int it = 0; it < max_it; ++it) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < NREG; ++j)
        if (first + j * stride < m) pnp_edge<true>(P, cam, rX[j], rUV[j], acc);
    } else {
      for (int e = first; e < m; e += stride) {
        const int i = sel ? sel[e] : e;
        const double X[3] = {pnp_obj(P, V.obj, (size_t)i, 0), pnp_obj(P, V.obj, (size_t)i, 1), pnp_obj(P, V.obj, (size_t)i, 2)};
        const double uv[2] = {V.img[2 * (size_t)i], V.img[2 * (size_t)i + 1]};
        pnp_edge<true>(P, cam, X, uv, acc);
      }
    }
    pnp_reduce<28, STEPS, NWAVES>(acc, s_red);
    double H[6][6], b[6];
    {
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = a; c < 6; ++c) {
          H[a][c] = acc[k];
          H[c][a] = acc[k];
          ++k;
        }
        b[a] = acc[21 + a];
      }
    }
    double cur = acc[27];
    if (it == 0) {
      double mx = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) mx = fmax(mx, fabs(H[a][a]));
      lambda = 1e-5 * mx;
      ni = 2.0;
    }
    double rho = 0.0;
    int qmax = 0, stop = 0, conv = 0;
    do {
      double A[6][6], x[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; ++c) A[a][c] = H[a][c];
        A[a][a] += lambda;
        x[a] = b[a];
      }
      int ok = 1;
      double rinv[6];  // one division per pivot; every other division by a pivot is a multiplication (oracle alike)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double sdiag = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= A[j][k] * A[j][k];
        if (!(sdiag > 0.0)) ok = 0;
        const double ri = vs_fast_rsq(sdiag);
        A[j][j] = sdiag * ri;
        rinv[j] = ri;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double v = A[i][j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
          A[i][j] = v * rinv[j];
        }
      }
      double temp = 1.7976931348623157e308;
      if (ok) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double v = x[i];
#pragma unroll
          for (int k = 0; k < i; ++k) v -= A[i][k] * x[k];
          x[i] = v * rinv[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
          double v = x[i];
#pragma unroll
          for (int k = i + 1; k < 6; ++k) v -= A[k][i] * x[k];
          x[i] = v * rinv[i];
        }
        cam_apply(cam, x, trial);
        double step2 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) step2 += x[a] * x[a];
        conv = step2 < kPnpStep2;  // a step this small: this trial is the last one
      } else {
#pragma unroll
        for (int a = 0; a < 6; ++a) x[a] = 0.0;
      }
      // the trial chi2 is reduced in any case: the reduction is a workgroup-wide rendezvous
      double tacc[28];
      tacc[27] = 0.0;
      if (ok) {
        if (in_regs) {
#pragma unroll
          for (int j = 0; j < NREG; ++j)
            if (first + j * stride < m) pnp_edge<false>(P, trial, rX[j], rUV[j], tacc);
        } else {
          for (int e = first; e < m; e += stride) {
            const int i = sel ? sel[e] : e;
            const double X[3] = {pnp_obj(P, V.obj, (size_t)i, 0), pnp_obj(P, V.obj, (size_t)i, 1), pnp_obj(P, V.obj, (size_t)i, 2)};
            const double uv[2] = {V.img[2 * (size_t)i], V.img[2 * (size_t)i + 1]};
            pnp_edge<false>(P, trial, X, uv, tacc);
          }
        }
      }
      pnp_reduce<1, STEPS, NWAVES>(tacc + 27, s_red);
      if (ok) temp = tacc[27];
      rho = cur - temp;
      double scale = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) scale += x[a] * (lambda * x[a] + b[a]);
      scale += 1e-3;
      rho = rho * vs_fast_rcp(scale);  // scale > 0: x^T (lambda x + b) >= 0 for a solved step
      if (rho > 0 && isfinite(temp)) {
        const double g = 2 * rho - 1;
        double alpha = 1.0 - g * g * g;
        alpha = fmin(alpha, 2.0 / 3.0);
        lambda *= fmax(1.0 / 3.0, alpha);
        ni = 2.0;
        cur = temp;
#pragma unroll
        for (int k = 0; k < kCamStride; ++k) cam[k] = trial[k];
      } else {
        lambda *= ni;
        ni *= 2;
        if (!isfinite(lambda)) {
          stop = 1;
          break;
        }
      }
      ++qmax;
    } while (rho < 0 && qmax < 10 && !conv);
    if (qmax == 10 || rho == 0 || stop || conv) break;
  }
}

}  // namespace

namespace {
__device__ inline int ransac_update_iters_dev(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.0), 1.0);
  ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1 - p, 2.2250738585072014e-308);
  double denom = 1 - pow(1 - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}


}  // namespace

namespace vsba {
// ONE launch: workgroups 0 .. nhw-1 hold four hypotheses each (one per wave; hypothesis h = wave * nhw + workgroup, so the
// first hypotheses -- the ones the budget replay reads -- sit on different compute units), workgroup nhw finishes.
//   hypothesis wave: sample, refine the guess on the five correspondences (pnp_lm on lanes 0..4 of every group of 8),
//     write the model (write-through), count the inliers over all points, then publish ONE tagged word
//     (call epoch << 32 | count): the word validates itself, stale words of earlier calls never carry this call's epoch.
//   finishing workgroup: wave 0 polls the tagged words 64 at a time and replays the sequential budget rule over the
//     leading run that has arrived -- it needs hypothesis k only when the budget still reaches k, which is typically a
//     handful -- then acquires, lists the winner's inliers in order and refines on them.  It therefore starts as soon as the
//     hypotheses the sequential algorithm would have looked at are in, not when the slowest of all of them is; hypotheses
//     beyond the budget finish on their own compute units meanwhile, nothing waits for them but the end of the launch.
// Two launches and a kernel boundary before (pnp_hypothesis_kernel 35 us + pnp_finish_kernel 30 us per tracked frame).
constexpr int kPnpWaves = kPnpFinish / 64;

__device__ inline void st_wt(double* p, double v) {  // write-through (sc1) store: the reader is on another compute unit
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void pnp_stamp(const pnp_args& P, int row, int col) {  // diagnostic only
  if (P.stamps) P.stamps[(size_t)row * 8 + col] = wall_clock64();
}

__device__ inline void pnp_hypothesis_role(const pnp_args& P, pnp_view& V, int nhw) {
  __shared__ int s_idx[kPnpWaves][8];
  __shared__ int s_cand[kPnpWaves][8];
  __shared__ double s_rec[kPnpWaves][kPnpModel];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = wv * nhw + blockIdx.x;
  if (h >= P.iterations) return;  // whole wave
  if (lane == 0) pnp_stamp(P, h, 0);
  const unsigned long long tag_hi = (unsigned long long)P.epoch << 32;
  if (V.n < 5) {
    if (lane == 0) __hip_atomic_store(&P.tag[h], tag_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  int* idx = s_idx[wv];
  if (V.n == 5) {
    if (lane < 5) idx[lane] = lane;
  } else {
    // the first five distinct values of the stream c_k = splitmix64(base ^ ((h << 20) + k)) % n, k = 0, 1, ...: eight
    // lanes draw eight candidates at a time (a 64-bit modulo is ~200 instructions), lane 0 picks in order
    int* cand = s_cand[wv];
    const unsigned long long base = splitmix64(P.seed);  // unrelated streams for neighbouring seeds
    int got = 0;
    for (unsigned long long k0 = 0; got < 5; k0 += 8) {
      if (lane < 8)
        cand[lane] = (int)(splitmix64(base ^ (((unsigned long long)h << 20) + k0 + (unsigned long long)lane)) % (unsigned long long)V.n);
      wave_lds_sync();
      if (lane == 0) {
        for (int k = 0; k < 8 && got < 5; ++k) {
          const int c = cand[k];
          bool dup = false;
          for (int j = 0; j < got; ++j) dup |= idx[j] == c;
          if (!dup) idx[got++] = c;
        }
        idx[7] = got;
      }
      wave_lds_sync();
      got = idx[7];
    }
  }
  wave_lds_sync();
  if (lane == 0) pnp_stamp(P, h, 1);
  if (P.off_dev) {  // chained tracking: sampled; now the previous frame's solve has to be through
    // one poller per workgroup (its four hypotheses sample in step): a hundred waves polling one word slowed the solve they wait for
    __shared__ int s_back;
    __syncthreads();
    if (threadIdx.x == 0) s_back = pnp_wait_tag(P.back_tag_dev, P.back_tag);
    __syncthreads();
    if (!s_back) return;  // the tags stay unpublished
    pnp_resolve_guess(P, V);
  }
  double cam[kCamStride];
#pragma unroll
  for (int k = 0; k < kCamStride; ++k) cam[k] = pnp_guess(P, V, k);
  pnp_lm<3, 1, 1>(P, V, idx, 5, lane & 7, 8, cam, nullptr, min(P.iters_lm, kPnpHypIters));
  if (lane == 0) pnp_stamp(P, h, 2);
  // the model is handed on as a 4x4 pose (as the sequential algorithm does): re-derive the record from that matrix
  double m[16];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) m[4 * r + k] = cam[7 + 4 * k + r];
    m[4 * r + 3] = cam[r];
  }
  quat_from_pose(m, cam + 3);
  quat_to_w2n(cam, cam + 3, cam + 7);
  if (lane == 0) pnp_stamp(P, h, 4);
  // the model leaves as ONE wave-wide write-through store: [pose rows 12 | record 19 | pad] = 32 doubles, lane l stores
  // word l (through LDS: every lane holds the whole model, a store needs lane-indexed words).  One store instruction = one
  // round trip to wait for before the tag; 31 scalar write-through stores by one lane cost ~12 us here.
  double* rec = s_rec[wv];
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) rec[k] = m[k];
#pragma unroll
    for (int k = 0; k < kCamStride; ++k) rec[12 + k] = cam[k];
    rec[31] = 0.0;
  }
  wave_lds_sync();
  if (lane < kPnpModel) st_wt(P.model_out + (size_t)h * kPnpModel + lane, rec[lane]);
  if (lane == 0) pnp_stamp(P, h, 5);
  int good = 0;
  for (int i0 = 0; i0 < V.n; i0 += 64) {
    const int i = i0 + lane;
    bool in = false;
    if (i < V.n) {
      const double X[3] = {pnp_obj(P, V.obj, (size_t)i, 0), pnp_obj(P, V.obj, (size_t)i, 1), pnp_obj(P, V.obj, (size_t)i, 2)};
      const double uv[2] = {V.img[2 * (size_t)i], V.img[2 * (size_t)i + 1]};
      double eu, ev, pc[3];
      pnp_err(P, cam, X, uv, eu, ev, pc);
      in = eu * eu + ev * ev <= P.thr2;
    }
    good += __popcll(__ballot(in));
  }
  if (lane == 0) pnp_stamp(P, h, 6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the model has left this compute unit before the word that announces it
  if (lane == 0) {
    __hip_atomic_store(&P.tag[h], tag_hi | (unsigned long long)(unsigned)good, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pnp_stamp(P, h, 3);
  }
}

__device__ inline void pnp_finish_role(const pnp_args& P, pnp_view& V) {
  __shared__ double s_red[kPnpRedDoubles];
  __shared__ int s_best[3], s_cnt[4], s_base, s_g[64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int srow = P.iterations;  // diagnostic row of the finishing workgroup
  if (tid == 0) pnp_stamp(P, srow, 0);
  // this workgroup's share of the correspondences (thread t: t, t + 256) is requested now and is in registers long before the
  // hypotheses report: the inlier listing below then starts without a round trip
  constexpr int kListPre = 2;
  double lX[kListPre][3], lU[kListPre][2];
#pragma unroll
  for (int j = 0; j < kListPre; ++j) {
    const int i = kPnpFinish * j + tid;
    lX[j][0] = lX[j][1] = lX[j][2] = lU[j][0] = lU[j][1] = 0.0;
    if (i < V.n) {
      lX[j][0] = pnp_obj(P, V.obj, (size_t)i, 0);
      lX[j][1] = pnp_obj(P, V.obj, (size_t)i, 1);
      lX[j][2] = pnp_obj(P, V.obj, (size_t)i, 2);
      lU[j][0] = V.img[2 * (size_t)i];
      lU[j][1] = V.img[2 * (size_t)i + 1];
    }
  }
  if (P.lm_init && tid == 64) {  // the motion-only solve that follows starts from fresh LM records (no upload in between)
    mo_state z;
    memset(&z, 0, sizeof z);
    z.cur = V.lm_cur;
    P.lm_init[0] = z;
    z.need_lin = 1;  // step 0 reads the record of parity 1
    z.ni = 2.0;
    P.lm_init[1] = z;
  }
  if (wv == 0) {
    // replay of the sequential RANSAC loop (budget update after every improvement) over the per-hypothesis counts, as they
    // arrive: 64 tagged words per poll, the leading run that carries this call's epoch is consumed in order.  Every lane
    // of the wave runs the same replay on the same values.
    int max_good = 0, niters = V.n >= 5 ? P.iterations : 0, best = -1, k = 0, timed_out = 0;
    int rounds = 0;
    while (k < niters && k < P.iterations) {
      const int kk = min(k + lane, P.iterations - 1);
      const unsigned long long w = __hip_atomic_load(&P.tag[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ready = (unsigned)(w >> 32) == P.epoch && k + lane < P.iterations;
      const unsigned long long bal = __ballot(ready);
      const int run = bal == ~0ull ? 64 : __ffsll((long long)~bal) - 1;  // lanes 0 .. run-1 have arrived
      if (run == 0) {
        if (++rounds > (1 << 16)) {  // bounded wait (~50 ms): hypothesis workgroups that never ran, or that gave up their own wait
          timed_out = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      s_g[lane] = (int)(unsigned)(w & 0xFFFFFFFFull);
      wave_lds_sync();
      for (int j = 0; j < run && k < niters; ++j, ++k) {
        const int g = s_g[j];
        if (g > (max_good > 4 ? max_good : 4)) {
          max_good = g;
          best = k;
          niters = ransac_update_iters_dev(P.confidence, (double)(V.n - g) / V.n, 5, niters);
        }
      }
      wave_lds_sync();
    }
    if (lane == 0) {
      pnp_stamp(P, srow, 1);
      s_best[0] = best;
      s_best[1] = k;
      s_best[2] = timed_out;
      s_base = 0;
      // the winner's model was stored write-through by another compute unit: agent-scope acquire before anybody loads it
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const int best = s_best[2] ? -1 : s_best[0];
  if (best < 0) {
    if (tid == 0) {
      P.result[16] = s_best[2] ? -1.0 : 0.0;  // -1: the hypotheses never arrived (reported as an error by the host)
      P.result[17] = 0.0;
      P.result[18] = -1.0;
      P.result[19] = (double)s_best[1];
      if (P.host_result) {
        P.host_result[16] = s_best[2] ? -1.0 : 0.0;
        P.host_result[17] = 0.0;
        P.host_result[18] = -1.0;
        P.host_result[19] = (double)s_best[1];
      }
    }
    if (P.rec_out[0] && tid < kCamStride) {
      double g = 0.0;  // the guess's element `tid`, picked with compile-time indices (see pnp_view)
#pragma unroll
      for (int k = 0; k < kCamStride; ++k) g = tid == k ? pnp_guess(P, V, k) : g;
      P.rec_out[0][tid] = g;
      P.rec_out[1][tid] = g;
    }
    pnp_announce(P);
    return;
  }
  double cam[kCamStride];
#pragma unroll
  for (int k = 0; k < kCamStride; ++k) cam[k] = P.model_out[(size_t)best * kPnpModel + 12 + k];
  // ordered inlier list of the best model
  for (int i0 = 0, j = 0; i0 < V.n; i0 += kPnpFinish, ++j) {
    const int i = i0 + tid;
    bool in = false;
    if (i < V.n) {
      double X[3], uv[2];
      if (j < kListPre) {
        // (static indices only: a register array indexed by the loop counter would go to scratch)
        X[0] = j == 0 ? lX[0][0] : lX[1][0];
        X[1] = j == 0 ? lX[0][1] : lX[1][1];
        X[2] = j == 0 ? lX[0][2] : lX[1][2];
        uv[0] = j == 0 ? lU[0][0] : lU[1][0];
        uv[1] = j == 0 ? lU[0][1] : lU[1][1];
      } else {
        X[0] = pnp_obj(P, V.obj, (size_t)i, 0);
        X[1] = pnp_obj(P, V.obj, (size_t)i, 1);
        X[2] = pnp_obj(P, V.obj, (size_t)i, 2);
        uv[0] = V.img[2 * (size_t)i];
        uv[1] = V.img[2 * (size_t)i + 1];
      }
      double eu, ev, pc[3];
      pnp_err(P, cam, X, uv, eu, ev, pc);
      in = eu * eu + ev * ev <= P.thr2;
    }
    const unsigned long long bal = __ballot(in);
    if (lane == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = s_base + __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wv; ++w) off += s_cnt[w];
    if (in) {
      P.inl_out[off] = i;
      if (P.host_inl) P.host_inl[off] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
  const int m = s_base;
  if (tid == 0) pnp_stamp(P, srow, 2);
  double pose[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) pose[k] = P.model_out[(size_t)best * kPnpModel + k];
  if (m >= 1 && P.iters_lm > 0) {
    // final refinement on the inliers (solvePnP(inliers, useExtrinsicGuess) in OpenCV)
    __threadfence_block();
    pnp_lm<6, kPnpFinish / 64, 4>(P, V, P.inl_out, m, tid, kPnpFinish, cam, s_red, P.iters_lm);
    if (tid == 0) pnp_stamp(P, srow, 4);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) pose[4 * r + k] = cam[7 + 4 * k + r];
      pose[4 * r + 3] = cam[r];
    }
  }
  if (P.rec_out[0] && tid == 0) {
    // the record is re-derived from the 4x4 result, as a caller that passes the pose matrix on would do
    double mat[16], rec[kCamStride];
#pragma unroll
    for (int k = 0; k < 12; ++k) mat[k] = pose[k];
    rec[0] = mat[3];
    rec[1] = mat[7];
    rec[2] = mat[11];
    quat_from_pose(mat, rec + 3);
    quat_to_w2n(rec, rec + 3, rec + 7);
#pragma unroll
    for (int k = 0; k < kCamStride; ++k) {
      P.rec_out[0][k] = rec[k];
      P.rec_out[1][k] = rec[k];
    }
  }
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) P.result[k] = pose[k];
    P.result[12] = P.result[13] = P.result[14] = 0.0;
    P.result[15] = 1.0;
    P.result[16] = 1.0;
    P.result[17] = (double)m;
    P.result[18] = (double)best;
    P.result[19] = (double)s_best[1];
    if (P.host_result) {  // the class-API period reads the PnP outcome from pinned memory, without a copy launch
#pragma unroll
      for (int k = 0; k < 12; ++k) P.host_result[k] = pose[k];
      P.host_result[12] = P.host_result[13] = P.host_result[14] = 0.0;
      P.host_result[15] = 1.0;
      P.host_result[16] = 1.0;
      P.host_result[17] = (double)m;
      P.host_result[18] = (double)best;
      P.host_result[19] = (double)s_best[1];
    }
    pnp_stamp(P, srow, 3);
  }
  pnp_announce(P);
}

__global__ __launch_bounds__(kPnpFinish) void pnp_ransac_kernel(const pnp_args P) {
  // chains of dependent FP64 instructions on single waves: when the next frame's detector shares the CUs (pipelined tracking),
  // the issue arbiter should serve these waves first
  __builtin_amdgcn_s_setprio(3);
  const int nhw = (int)gridDim.x - 1;
  // chained tracking: the correspondences come from the front half on another stream, the guess from the previous frame's
  // motion-only solve on yet another.  Their last workgroups publish tagged words (agent-scope release behind everybody's
  // results); the waits are bounded -- both were enqueued before this launch and neither depends on it -- and whoever gives up
  // leaves the launch (the finishing workgroup then reports the hypotheses as missing, which the host turns into an error).
  // Order: front half -> count and row offset -> [hypotheses: sampling] -> previous solve -> guess.  The launch is resident
  // and has sampled by the time the previous solve ends.
  const bool fin = (int)blockIdx.x == nhw;
  if (P.front_tag_dev || (fin && P.back_tag_dev)) {
    __shared__ int s_front, s_back;
    if (threadIdx.x == 0) {
      s_front = pnp_wait_tag(P.front_tag_dev, P.front_tag);
      s_back = !fin || pnp_wait_tag(P.back_tag_dev, P.back_tag);
    }
    __syncthreads();
    // the previous frame's results to the host, before anything here overwrites them -- they depend on the previous solve only,
    // so they go out even when THIS frame's front half never reported (the host then redoes this frame, not the previous one)
    if (fin && P.pub_src && s_back) {
      for (int i = threadIdx.x; i < P.pub_n16; i += kPnpFinish) P.pub_dst[i] = P.pub_src[i];
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(P.pub_tag_word, P.pub_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!s_front || !s_back) {
      if (fin && threadIdx.x == 0) {
        P.result[16] = -1.0;
        if (P.host_result) P.host_result[16] = -1.0;
      }
      if (fin) pnp_announce(P);
      return;
    }
  } else if (fin && P.pub_src) {
    for (int i = threadIdx.x; i < P.pub_n16; i += kPnpFinish) P.pub_dst[i] = P.pub_src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.pub_tag_word, P.pub_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  pnp_view V;
  V.n = pnp_count(P);
  V.lm_cur = P.lm_cur;
  V.obj = P.obj;
  V.img = P.img;
  V.guess = nullptr;
  if (P.off_dev) {
    const int off = *P.off_dev;
    V.obj += 3 * (size_t)off;
    V.img += 2 * (size_t)off;
    if (fin) pnp_resolve_guess(P, V);  // (the hypothesis waves do it after their sampling, behind their own wait)
  }
  if ((int)blockIdx.x < nhw) pnp_hypothesis_role(P, V, nhw);
  else pnp_finish_role(P, V);
}

// grid of pnp_ransac_kernel for `iterations` hypotheses
int pnp_grid(int iterations) { return (iterations + kPnpWaves - 1) / kPnpWaves + 1; }

// the tagged words of the hypotheses: a buffer of its own, zero when (re)allocated, and a call epoch that only grows --
// no word written by an earlier call, and nothing else that ever lived at these addresses, can carry the current epoch
int pnp_tags(vs_ctx* ctx, int iterations, hipStream_t s, unsigned long long** tag, unsigned* epoch) {
  const size_t bytes = sizeof(unsigned long long) * (size_t)(iterations > 0 ? iterations : 1);
  if (bytes > ctx->d_pnp_tag.cap || !ctx->d_pnp_tag.p) {
    VS_TRY(vs_reserve(ctx, &ctx->d_pnp_tag, bytes));
    VS_HIP(ctx, hipMemsetAsync(ctx->d_pnp_tag.p, 0, ctx->d_pnp_tag.cap, s));
  }
  ctx->pnp_epoch = ctx->pnp_epoch == 0xFFFFFFFFu ? 1u : ctx->pnp_epoch + 1u;
  *tag = (unsigned long long*)ctx->d_pnp_tag.p;
  *epoch = ctx->pnp_epoch;
  return VS_OK;
}

// diagnostic stamps of the next launch (nullptr unless vs_pnp_profile switched them on)
int pnp_stamps(vs_ctx* ctx, int iterations, hipStream_t s, unsigned long long** stamps) {
  *stamps = nullptr;
  if (!ctx->pnp_profile) return VS_OK;
  const size_t bytes = sizeof(unsigned long long) * 8 * (size_t)(iterations + 1);
  VS_TRY(vs_reserve(ctx, &ctx->d_pnp_stamps, bytes));
  VS_HIP(ctx, hipMemsetAsync(ctx->d_pnp_stamps.p, 0, bytes, s));
  ctx->pnp_profile_h = iterations;
  *stamps = (unsigned long long*)ctx->d_pnp_stamps.p;
  return VS_OK;
}
}  // namespace vsba

// diagnostic hooks (not part of the stable ABI): phase stamps of pnp_ransac_kernel.  vs_pnp_profile_read synchronises and
// returns the stamps of the newest profiled launch as microseconds since the launch's first stamp: rows 0 .. H-1 =
// hypotheses (start, sampled, refined, published), row H = the finishing workgroup (start, winner known, inliers listed,
// results written, refinement done); 8 doubles per row, zero = not reached.  Returns the number of rows.
VS_API int vs_pnp_profile(vs_ctx* ctx, int enable) {
  if (!ctx) return VS_EINVAL;
  ctx->pnp_profile = enable != 0;
  return VS_OK;
}
VS_API int vs_pnp_profile_read(vs_ctx* ctx, double* out, int cap_rows) {
  if (!ctx || !out) return VS_EINVAL;
  const int rows = ctx->pnp_profile_h + 1;
  if (!ctx->d_pnp_stamps.p || ctx->pnp_profile_h <= 0 || cap_rows < rows) return 0;
  std::vector<unsigned long long> h((size_t)rows * 8);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(h.data(), ctx->d_pnp_stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return vs_fail(ctx, VS_EHIP, "%s: read-back failed", "vs_pnp_profile_read");
  unsigned long long t0 = ~0ull;
  for (unsigned long long v : h)
    if (v && v < t0) t0 = v;
  for (size_t i = 0; i < h.size(); ++i) out[i] = h[i] ? (double)(h[i] - t0) * 0.01 : 0.0;  // wall_clock64: 100 MHz
  return rows;
}

VS_API int vs_pnp_ransac(vs_ctx* ctx, const double* obj, const double* img, int n, double fx, double fy, double cx,
                         double cy, const double* pose0, int iterations, double reproj_err, double confidence,
                         uint64_t seed, int refine_iters, double* pose_out, int32_t* inliers, int* n_inliers,
                         int* found) {
  if (!ctx) return VS_EINVAL;
  if (!pose0 || !pose_out || !n_inliers || !found || n < 0 || iterations < 0 || refine_iters < 0 ||
      (n > 0 && (!obj || !img || !inliers)))
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_pnp_ransac");
  *found = 0;
  *n_inliers = 0;
  memcpy(pose_out, pose0, 16 * sizeof(double));
  if (n < 5 || iterations == 0) return VS_OK;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int H = iterations;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // [obj | img] uploaded; [result | inliers] read back; [cam | pose | good] stay on the device
  const size_t off_img = up(sizeof(double) * 3 * (size_t)n);
  const size_t off_res = off_img + up(sizeof(double) * 2 * (size_t)n);
  const size_t off_inl = off_res + 256;
  const size_t off_cam = off_inl + up(sizeof(int) * (size_t)n);
  const size_t total = off_cam + up(sizeof(double) * kPnpModel * (size_t)H);
  VS_TRY(vs_reserve(ctx, &ctx->d_xy_in, total));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, off_cam));
  VS_HIP(ctx, hipStreamSynchronize(s));
  uint8_t* h = (uint8_t*)ctx->h_pin_big.p;
  uint8_t* d = (uint8_t*)ctx->d_xy_in.p;
  memcpy(h, obj, sizeof(double) * 3 * (size_t)n);
  memcpy(h + off_img, img, sizeof(double) * 2 * (size_t)n);
  VS_HIP(ctx, hipMemcpyAsync(d, h, off_res, hipMemcpyHostToDevice, s));
  pnp_args P;
  memset(&P, 0, sizeof P);
  P.obj = (const double*)d;
  P.img = (const double*)(d + off_img);
  P.n = n;
  P.iters_lm = refine_iters;
  P.iterations = H;
  P.fx = fx;
  P.fy = fy;
  P.cx = cx;
  P.cy = cy;
  P.thr2 = reproj_err * reproj_err;
  P.confidence = confidence;
  P.seed = seed;
  P.cam0[0] = pose0[3];
  P.cam0[1] = pose0[7];
  P.cam0[2] = pose0[11];
  quat_from_pose(pose0, P.cam0 + 3);
  quat_to_w2n(P.cam0, P.cam0 + 3, P.cam0 + 7);
  P.result = (double*)(d + off_res);
  P.inl_out = (int*)(d + off_inl);
  P.model_out = (double*)(d + off_cam);
  VS_TRY(pnp_tags(ctx, H, s, &P.tag, &P.epoch));
  VS_TRY(pnp_stamps(ctx, H, s, &P.stamps));
  hipLaunchKernelGGL(pnp_ransac_kernel, dim3(pnp_grid(H)), dim3(kPnpFinish), 0, s, P);
  VS_LAUNCH_CHECK(ctx, "pnp_ransac_kernel");
  VS_HIP(ctx, hipMemcpyAsync(h + off_res, d + off_res, off_cam - off_res, hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  const double* res = (const double*)(h + off_res);
  if (res[16] < 0.0) return vs_fail(ctx, VS_EHIP, "%s: the hypothesis workgroups did not report", "vs_pnp_ransac");
  if (res[16] != 0.0) {
    const int m = (int)res[17];
    *found = 1;
    *n_inliers = m;
    memcpy(pose_out, res, 16 * sizeof(double));
    memcpy(inliers, h + off_inl, sizeof(int32_t) * (size_t)m);
  }
  return VS_OK;
}
