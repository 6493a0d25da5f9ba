// vs_twoview.hip -- two-view initialisation (gfx950): essential-matrix RANSAC and pose recovery.
//
// Replaces estimateEssential / estimateRelativePose (reference src/v2/helper_functions.py:47-70,164-195), i.e.
// cv2.findEssentialMat(pts1, pts2, method=RANSAC, prob=0.999, threshold) on K-normalised points and
// cv2.recoverPose(E, pts1, pts2, K, distanceThresh=50).  Runs once per sequence (src/v2/main.py:88-148).
// Kept from OpenCV: Sampson error against threshold^2, RANSACUpdateNumIters after every strictly better model,
// decomposeEssentialMat's (R1, R2, +-t) and recoverPose's four-candidate cheirality vote.
// Own specification (OpenCV cannot be pinned here): 8-point minimal solver (null vector of the 8x9 system by Gaussian
// elimination with full pivoting, then the closest matrix with singular values (1,1,0)), counter-based sampling
// (splitmix64), one-sided Jacobi SVDs, one linear re-fit of the winner to its inliers (adopted if not worse).
//   ess_hypothesis_kernel: one wave per hypothesis -- lane 0 solves the sample, the 64 lanes score all correspondences,
//     a ballot counts inliers; hypotheses are independent because the sampler is counter-based, and the host replays the
//     sequential budget rule over the counts (exactly the sequential algorithm's winner).
//   recover_kernel: one thread per correspondence triangulates it against all four (R, t) candidates (4x4 DLT, Jacobi
//     SVD in registers) and writes the four cheirality verdicts; the host tallies them and picks the candidate.
#include "vs_internal.h"

#include <math.h>

#include <vector>

namespace {

#define HD __host__ __device__ inline

template <int N>
HD void jacobi_svd(double* A /* N x N row-major, becomes U*Sigma */, double* V) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) V[r * N + c] = r == c ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < N; ++r) {
          alpha += A[r * N + p] * A[r * N + p];
          beta += A[r * N + q] * A[r * N + q];
          gamma += A[r * N + p] * A[r * N + q];
        }
        const double lim = sqrt(alpha * beta);
        if (lim > 0.0) off = fmax(off, fabs(gamma) / lim);
        if (fabs(gamma) > 1e-300 && fabs(gamma) > 1e-17 * lim) {
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
          for (int r = 0; r < N; ++r) {
            const double ap = A[r * N + p], aq = A[r * N + q];
            A[r * N + p] = cs * ap - sn * aq;
            A[r * N + q] = sn * ap + cs * aq;
            const double vp = V[r * N + p], vq = V[r * N + q];
            V[r * N + p] = cs * vp - sn * vq;
            V[r * N + q] = sn * vp + cs * vq;
          }
        }
      }
    if (off < 1e-15) break;
  }
}

// E = U diag(s) V^T with s0 >= s1 >= s2, U and V proper rotations (third columns by cross product)
HD void svd3_sorted(const double* E, double* U, double* s, double* V) {
  double A[9], W[9];
  for (int k = 0; k < 9; ++k) A[k] = E[k];
  jacobi_svd<3>(A, W);
  double nn[3];
  int ord[3] = {0, 1, 2};
  for (int c = 0; c < 3; ++c) nn[c] = A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2 - a; ++b)
      if (nn[ord[b]] < nn[ord[b + 1]]) {
        const int t = ord[b];
        ord[b] = ord[b + 1];
        ord[b + 1] = t;
      }
  for (int k = 0; k < 2; ++k) {
    const int c = ord[k];
    s[k] = sqrt(nn[c]);
    for (int r = 0; r < 3; ++r) {
      U[3 * r + k] = s[k] > 0 ? A[3 * r + c] / s[k] : 0.0;
      V[3 * r + k] = W[3 * r + c];
    }
  }
  s[2] = sqrt(nn[ord[2]]);
  U[2] = U[3] * U[7] - U[6] * U[4];
  U[5] = U[6] * U[1] - U[0] * U[7];
  U[8] = U[0] * U[4] - U[3] * U[1];
  V[2] = V[3] * V[7] - V[6] * V[4];
  V[5] = V[6] * V[1] - V[0] * V[7];
  V[8] = V[0] * V[4] - V[3] * V[1];
}

// 8 correspondences (a, b) <-> (c, d), K-normalised -> essential matrix row-major (x2^T E x1 = 0); false: degenerate
HD bool eight_point(const double (*pt)[4], double* E) {
  double A[8][9];
  int perm[9];
  for (int k = 0; k < 8; ++k) {
    const double a = pt[k][0], b = pt[k][1], c = pt[k][2], d = pt[k][3];
    A[k][0] = c * a;
    A[k][1] = c * b;
    A[k][2] = c;
    A[k][3] = d * a;
    A[k][4] = d * b;
    A[k][5] = d;
    A[k][6] = a;
    A[k][7] = b;
    A[k][8] = 1.0;
  }
  for (int j = 0; j < 9; ++j) perm[j] = j;
  for (int k = 0; k < 8; ++k) {
    int pi = k, pj = k;
    double best = -1.0;
    for (int i = k; i < 8; ++i)
      for (int j = k; j < 9; ++j)
        if (fabs(A[i][j]) > best) {
          best = fabs(A[i][j]);
          pi = i;
          pj = j;
        }
    if (!(best > 1e-12)) return false;
    for (int j = 0; j < 9; ++j) {
      const double t = A[k][j];
      A[k][j] = A[pi][j];
      A[pi][j] = t;
    }
    for (int i = 0; i < 8; ++i) {
      const double t = A[i][k];
      A[i][k] = A[i][pj];
      A[i][pj] = t;
    }
    const int tp = perm[k];
    perm[k] = perm[pj];
    perm[pj] = tp;
    for (int i = k + 1; i < 8; ++i) {
      const double f = A[i][k] / A[k][k];
      for (int j = k; j < 9; ++j) A[i][j] -= f * A[k][j];
    }
  }
  double x[9], e[9];
  x[8] = 1.0;
  for (int k = 7; k >= 0; --k) {
    double sacc = 0.0;
    for (int j = k + 1; j < 9; ++j) sacc += A[k][j] * x[j];
    x[k] = -sacc / A[k][k];
  }
  double nrm = 0.0;
  for (int j = 0; j < 9; ++j) nrm += x[j] * x[j];
  nrm = sqrt(nrm);
  for (int j = 0; j < 9; ++j) e[perm[j]] = x[j] / nrm;
  double U[9], s[3], V[9];
  svd3_sorted(e, U, s, V);
  if (!(s[1] > 1e-12)) return false;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1];
  return true;
}

HD double sampson(const double* E, double a, double b, double c, double d) {
  const double l0 = E[0] * a + E[1] * b + E[2], l1 = E[3] * a + E[4] * b + E[5], l2 = E[6] * a + E[7] * b + E[8];
  const double m0 = E[0] * c + E[3] * d + E[6], m1 = E[1] * c + E[4] * d + E[7];
  const double r = c * l0 + d * l1 + l2;
  return r * r / (l0 * l0 + l1 * l1 + m0 * m0 + m1 * m1);
}

// least-squares 8-point over the correspondences with mask != 0: smallest eigenvector of A^T A (9x9, Jacobi), then the
// same projection onto the essential manifold  [host: runs once, after the RANSAC winner is known]
inline bool eight_point_lsq(const double* x1, const double* x2, const uint8_t* mask, int n, double* E) {
  double M[81], V[81];
  memset(M, 0, sizeof M);
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    const double a = x1[2 * (size_t)i], b = x1[2 * (size_t)i + 1], c = x2[2 * (size_t)i], d = x2[2 * (size_t)i + 1];
    const double row[9] = {c * a, c * b, c, d * a, d * b, d, a, b, 1.0};
    for (int r = 0; r < 9; ++r)
      for (int q = 0; q < 9; ++q) M[9 * r + q] += row[r] * row[q];
    ++cnt;
  }
  if (cnt < 8) return false;
  jacobi_svd<9>(M, V);
  double best = 1.7976931348623157e308;
  int bc = 8;
  for (int k = 0; k < 9; ++k) {
    double nn = 0;
    for (int r = 0; r < 9; ++r) nn += M[9 * r + k] * M[9 * r + k];
    if (nn < best) {
      best = nn;
      bc = k;
    }
  }
  double e[9], U[9], sv[3], W[9];
  for (int r = 0; r < 9; ++r) e[r] = V[9 * r + bc];
  svd3_sorted(e, U, sv, W);
  if (!(sv[1] > 1e-12)) return false;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E[3 * r + c] = U[3 * r] * W[3 * c] + U[3 * r + 1] * W[3 * c + 1];
  return true;
}

HD void decompose_essential(const double* E, double* R1, double* R2, double* t) {
  double U[9], s[3], V[9];
  svd3_sorted(E, U, s, V);
  // W = [0 -1 0; 1 0 0; 0 0 1]:  U W = [u1, -u0, u2],  U W^T = [-u1, u0, u2]
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      R1[3 * r + c] = U[3 * r + 1] * V[3 * c] - U[3 * r] * V[3 * c + 1] + U[3 * r + 2] * V[3 * c + 2];
      R2[3 * r + c] = -U[3 * r + 1] * V[3 * c] + U[3 * r] * V[3 * c + 1] + U[3 * r + 2] * V[3 * c + 2];
    }
  for (int r = 0; r < 3; ++r) t[r] = U[3 * r + 2];
}

HD unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct ess_args {
  const double* x1;  // [n][2]
  const double* x2;
  int n, pad;
  double thr2;
  unsigned long long seed;
  double* E_out;  // [H][9]
  int* good_out;  // [H]
};

__global__ __launch_bounds__(64) void ess_hypothesis_kernel(ess_args P) {
  __shared__ double s_E[9];
  __shared__ int s_ok;
  const int h = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) {
    int idx[8];
    if (P.n == 8) {
      for (int k = 0; k < 8; ++k) idx[k] = k;
    } else {
      int got = 0;
      const unsigned long long base = splitmix64(P.seed);
      for (unsigned long long k = 0; got < 8; ++k) {
        const int c = (int)(splitmix64(base ^ (((unsigned long long)h << 20) + k)) % (unsigned long long)P.n);
        bool dup = false;
        for (int j = 0; j < got; ++j) dup |= idx[j] == c;
        if (!dup) idx[got++] = c;
      }
    }
    double pt[8][4], E[9];
    for (int k = 0; k < 8; ++k) {
      pt[k][0] = P.x1[2 * (size_t)idx[k]];
      pt[k][1] = P.x1[2 * (size_t)idx[k] + 1];
      pt[k][2] = P.x2[2 * (size_t)idx[k]];
      pt[k][3] = P.x2[2 * (size_t)idx[k] + 1];
    }
    const bool ok = eight_point(pt, E);
    s_ok = ok;
    for (int k = 0; k < 9; ++k) {
      s_E[k] = ok ? E[k] : 0.0;
      P.E_out[(size_t)h * 9 + k] = s_E[k];
    }
  }
  __syncthreads();
  int good = 0;
  if (s_ok) {
    double E[9];
    for (int k = 0; k < 9; ++k) E[k] = s_E[k];
    for (int i0 = 0; i0 < P.n; i0 += 64) {
      const int i = i0 + lane;
      bool in = false;
      if (i < P.n)
        in = sampson(E, P.x1[2 * (size_t)i], P.x1[2 * (size_t)i + 1], P.x2[2 * (size_t)i], P.x2[2 * (size_t)i + 1]) <= P.thr2;
      good += __popcll(__ballot(in));
    }
  }
  if (lane == 0) P.good_out[h] = good;
}

struct rec_args {
  const double* x1;
  const double* x2;
  int n, pad;
  double dist;
  double R[4][9], t[4][3];
  double* Q;      // [4][n][4]
  uint8_t* mask;  // [4][n]
};

__global__ __launch_bounds__(256) void recover_kernel(rec_args P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.n) return;
  const double a = P.x1[2 * (size_t)i], b = P.x1[2 * (size_t)i + 1], c = P.x2[2 * (size_t)i], d = P.x2[2 * (size_t)i + 1];
  for (int k = 0; k < 4; ++k) {
    const double* R = P.R[k];
    const double* t = P.t[k];
    // DLT against P0 = [I|0], P1 = [R|t]
    double A[16], V[16];
    A[0] = -1.0;
    A[1] = 0.0;
    A[2] = a;
    A[3] = 0.0;
    A[4] = 0.0;
    A[5] = -1.0;
    A[6] = b;
    A[7] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[8 + j] = c * R[6 + j] - R[j];
      A[12 + j] = d * R[6 + j] - R[3 + j];
    }
    A[11] = c * t[2] - t[0];
    A[15] = d * t[2] - t[1];
    jacobi_svd<4>(A, V);
    double best = 1.7976931348623157e308;
    int bc = 3;
    for (int j = 0; j < 4; ++j) {
      double nn = 0;
      for (int r = 0; r < 4; ++r) nn += A[4 * r + j] * A[4 * r + j];
      if (nn < best) {
        best = nn;
        bc = j;
      }
    }
    double Q[4], nrm = 0;
    for (int r = 0; r < 4; ++r) {
      Q[r] = V[4 * r + bc];
      nrm += Q[r] * Q[r];
    }
    nrm = sqrt(nrm);
    if (Q[3] < 0) nrm = -nrm;
    if (nrm != 0.0)
      for (int r = 0; r < 4; ++r) Q[r] /= nrm;
    bool ok = Q[2] * Q[3] > 0;
    const double X0 = Q[0] / Q[3], X1 = Q[1] / Q[3], X2 = Q[2] / Q[3];
    ok = ok && X2 < P.dist;
    const double z2 = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
    ok = ok && z2 > 0 && z2 < P.dist;
    P.mask[(size_t)k * P.n + i] = ok ? 255 : 0;
    for (int r = 0; r < 4; ++r) P.Q[((size_t)k * P.n + i) * 4 + r] = Q[r];
  }
}

int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
  if (p < 0) p = 0;
  if (p > 1) p = 1;
  if (ep < 0) ep = 0;
  if (ep > 1) ep = 1;
  double num = 1 - p > 2.2250738585072014e-308 ? 1 - p : 2.2250738585072014e-308;
  double denom = 1 - pow(1 - ep, model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

VS_API int vs_essential_ransac(vs_ctx* ctx, const double* x1, const double* x2, int n, double threshold, double prob,
                               int max_iters, uint64_t seed, double* E_out, uint8_t* mask, int* n_inliers, int* found) {
  if (!ctx) return VS_EINVAL;
  if (!E_out || !n_inliers || !found || n < 0 || max_iters < 0 || (n > 0 && (!x1 || !x2 || !mask)))
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_essential_ransac");
  *found = 0;
  *n_inliers = 0;
  memset(E_out, 0, 9 * sizeof(double));
  if (n > 0) memset(mask, 0, (size_t)n);
  if (n < 8 || max_iters == 0) return VS_OK;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int H = max_iters;
  const size_t pts = up256(sizeof(double) * 2 * (size_t)n);
  const size_t off_E = 2 * pts, off_good = off_E + up256(sizeof(double) * 9 * (size_t)H);
  const size_t total = off_good + up256(sizeof(int) * (size_t)H);
  VS_TRY(vs_reserve(ctx, &ctx->d_xy_in, total));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, total));
  VS_HIP(ctx, hipStreamSynchronize(s));
  uint8_t* h = (uint8_t*)ctx->h_pin_big.p;
  uint8_t* d = (uint8_t*)ctx->d_xy_in.p;
  memcpy(h, x1, sizeof(double) * 2 * (size_t)n);
  memcpy(h + pts, x2, sizeof(double) * 2 * (size_t)n);
  VS_HIP(ctx, hipMemcpyAsync(d, h, off_E, hipMemcpyHostToDevice, s));
  ess_args P;
  P.x1 = (const double*)d;
  P.x2 = (const double*)(d + pts);
  P.n = n;
  P.pad = 0;
  P.thr2 = threshold * threshold;
  P.seed = seed;
  P.E_out = (double*)(d + off_E);
  P.good_out = (int*)(d + off_good);
  hipLaunchKernelGGL(ess_hypothesis_kernel, dim3(H), dim3(64), 0, s, P);
  VS_LAUNCH_CHECK(ctx, "ess_hypothesis_kernel");
  VS_HIP(ctx, hipMemcpyAsync(h + off_E, d + off_E, total - off_E, hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  const double* Es = (const double*)(h + off_E);
  const int* good = (const int*)(h + off_good);
  // replay of the sequential loop (budget update after every improvement) over the per-hypothesis counts
  int max_good = 0, niters = H, best = -1;
  for (int k = 0; k < niters && k < H; ++k)
    if (good[k] > (max_good > 7 ? max_good : 7)) {
      max_good = good[k];
      best = k;
      niters = ransac_update_iters(prob, (double)(n - good[k]) / n, 8, niters);
    }
  if (best < 0) return VS_OK;
  double E[9], Els[9];
  memcpy(E, Es + (size_t)best * 9, sizeof E);
  auto apply = [&](const double* M) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      mask[i] = sampson(M, x1[2 * (size_t)i], x1[2 * (size_t)i + 1], x2[2 * (size_t)i], x2[2 * (size_t)i + 1]) <= P.thr2;
      m += mask[i];
    }
    return m;
  };
  int m = apply(E);
  // local optimisation: linear 8-point fit to all inliers, adopted if it explains at least as many correspondences
  if (eight_point_lsq(x1, x2, mask, n, Els)) {
    int m2 = 0;
    for (int i = 0; i < n; ++i)
      m2 += sampson(Els, x1[2 * (size_t)i], x1[2 * (size_t)i + 1], x2[2 * (size_t)i], x2[2 * (size_t)i + 1]) <= P.thr2;
    if (m2 >= m) {
      memcpy(E, Els, sizeof E);
      m = apply(E);
    }
  }
  memcpy(E_out, E, 9 * sizeof(double));
  *n_inliers = m;
  *found = 1;
  return VS_OK;
}

VS_API int vs_recover_pose(vs_ctx* ctx, const double* E, const double* x1, const double* x2, int n, double dist_thresh,
                           double* R_out, double* t_out, uint8_t* mask, double* X, int* n_good) {
  if (!ctx) return VS_EINVAL;
  if (!E || !R_out || !t_out || !n_good || n < 0 || (n > 0 && (!x1 || !x2 || !mask || !X)))
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_recover_pose");
  rec_args P;
  double R1[9], R2[9], t[3];
  decompose_essential(E, R1, R2, t);
  for (int k = 0; k < 4; ++k) {
    memcpy(P.R[k], (k & 1) ? R2 : R1, sizeof R1);
    for (int r = 0; r < 3; ++r) P.t[k][r] = k < 2 ? t[r] : -t[r];
  }
  int good[4] = {0, 0, 0, 0};
  std::vector<uint8_t> m4;
  std::vector<double> q4;
  if (n > 0) {
    VS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t pts = up256(sizeof(double) * 2 * (size_t)n);
    const size_t off_q = 2 * pts, off_m = off_q + up256(sizeof(double) * 16 * (size_t)n);
    const size_t total = off_m + up256(4 * (size_t)n);
    VS_TRY(vs_reserve(ctx, &ctx->d_xy_in, total));
    VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, total));
    VS_HIP(ctx, hipStreamSynchronize(s));
    uint8_t* h = (uint8_t*)ctx->h_pin_big.p;
    uint8_t* d = (uint8_t*)ctx->d_xy_in.p;
    memcpy(h, x1, sizeof(double) * 2 * (size_t)n);
    memcpy(h + pts, x2, sizeof(double) * 2 * (size_t)n);
    VS_HIP(ctx, hipMemcpyAsync(d, h, off_q, hipMemcpyHostToDevice, s));
    P.x1 = (const double*)d;
    P.x2 = (const double*)(d + pts);
    P.n = n;
    P.pad = 0;
    P.dist = dist_thresh;
    P.Q = (double*)(d + off_q);
    P.mask = d + off_m;
    hipLaunchKernelGGL(recover_kernel, dim3((n + 255) / 256), dim3(256), 0, s, P);
    VS_LAUNCH_CHECK(ctx, "recover_kernel");
    VS_HIP(ctx, hipMemcpyAsync(h + off_q, d + off_q, total - off_q, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
    const uint8_t* hm = h + off_m;
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < n; ++i) good[k] += hm[(size_t)k * n + i] != 0;
    int pick = 3;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) pick = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) pick = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) pick = 2;
    memcpy(mask, hm + (size_t)pick * n, (size_t)n);
    memcpy(X, (const double*)(h + off_q) + (size_t)pick * n * 4, sizeof(double) * 4 * (size_t)n);
    memcpy(R_out, P.R[pick], sizeof R1);
    memcpy(t_out, P.t[pick], sizeof t);
    *n_good = good[pick];
    return VS_OK;
  }
  memcpy(R_out, P.R[0], sizeof R1);
  memcpy(t_out, P.t[0], sizeof t);
  *n_good = 0;
  return VS_OK;
}
