// vs_triangulate.hip -- two-view DLT triangulation (gfx950).
//
// Replaces helper_functions.triangulate (reference src/v2/helper_functions.py:281-291), which is pure NumPy: per match
//     A = [u1*P1[2]-P1[0]; v1*P1[2]-P1[1]; u2*P2[2]-P2[0]; v2*P2[2]-P2[1]],   _, _, vt = np.linalg.svd(A);  X = vt[3]
// and the cheirality / depth filter of the caller (src/v2/main.py:291-309).
// One thread per match: one-sided (Hestenes) Jacobi SVD of the 4x4 matrix in registers -- it orthogonalises the columns
// of A by plane rotations accumulated in V, which keeps the small singular vector accurate (no A^T A squaring) -- then the
// column of V belonging to the smallest column norm.  The sign is normalised to w >= 0 (LAPACK's sign is arbitrary and
// the caller divides by w immediately, main.py:286).  FP64; embarrassingly parallel, latency-bound at SLAM sizes.
#include "vs_internal.h"

#include <math.h>

namespace {

__global__ __launch_bounds__(256) void triangulate_kernel(const double* __restrict__ Pm /*[2][12]*/,
                                                          const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                          int n, int stride, double* __restrict__ X4,
                                                          const double* __restrict__ Tm /*[2][12] world->camera or null*/,
                                                          double* __restrict__ depth /*[n][2] or null*/) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double u1 = pts1[(size_t)i * stride], v1 = pts1[(size_t)i * stride + 1];
  const double u2 = pts2[(size_t)i * stride], v2 = pts2[(size_t)i * stride + 1];
  double A[4][4], V[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    A[0][c] = u1 * Pm[8 + c] - Pm[c];
    A[1][c] = v1 * Pm[8 + c] - Pm[4 + c];
    A[2][c] = u2 * Pm[20 + c] - Pm[12 + c];
    A[3][c] = v2 * Pm[20 + c] - Pm[16 + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) V[r][c] = r == c ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          alpha += A[r][p] * A[r][p];
          beta += A[r][q] * A[r][q];
          gamma += A[r][p] * A[r][q];
        }
        const double lim = sqrt(alpha * beta);
        if (lim > 0.0) off = fmax(off, fabs(gamma) / lim);
        if (fabs(gamma) > 1e-300 && fabs(gamma) > 1e-17 * lim) {
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double ap = A[r][p], aq = A[r][q];
            A[r][p] = cs * ap - sn * aq;
            A[r][q] = sn * ap + cs * aq;
            const double vp = V[r][p], vq = V[r][q];
            V[r][p] = cs * vp - sn * vq;
            V[r][q] = sn * vp + cs * vq;
          }
        }
      }
    if (off < 1e-15) break;
  }
  // column with the smallest norm = right singular vector of the smallest singular value
  double best = 1.7976931348623157e308;
  double x[4] = {0, 0, 0, 1};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double nn = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) nn += A[r][c] * A[r][c];
    if (nn < best) {
      best = nn;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = V[r][c];
    }
  }
  double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  if (x[3] < 0.0) nrm = -nrm;  // sign convention: w >= 0
  if (nrm != 0.0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] /= nrm;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) X4[4 * (size_t)i + r] = x[r];
  if (depth) {
    // main.py:286-292: X /= X[3]; proj = p @ X; the filter reads proj[2] of both world->camera transforms
    const double w = x[3];
    const double X = x[0] / w, Y = x[1] / w, Z = x[2] / w;
    depth[2 * (size_t)i] = Tm[8] * X + Tm[9] * Y + Tm[10] * Z + Tm[11] * 1.0;
    depth[2 * (size_t)i + 1] = Tm[20] * X + Tm[21] * Y + Tm[22] * Z + Tm[23] * 1.0;
  }
}

}  // namespace

VS_API int vs_triangulate_dlt(vs_ctx* ctx, const double* P1, const double* P2, const double* pts1, const double* pts2,
                              int n, int stride, double* X4, const double* T1, const double* T2, double* depth) {
  if (!ctx) return VS_EINVAL;
  if (!P1 || !P2 || n < 0 || stride < 2 || (n > 0 && (!pts1 || !pts2 || !X4)) || (depth && (!T1 || !T2)))
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_triangulate_dlt");
  if (n == 0) return VS_OK;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pts_bytes = sizeof(double) * (size_t)n * stride;
  const size_t off_p1 = 512, off_p2 = off_p1 + ((pts_bytes + 255) & ~(size_t)255);
  const size_t off_x = off_p2 + ((pts_bytes + 255) & ~(size_t)255);
  const size_t off_d = off_x + sizeof(double) * 4 * (size_t)n;
  const size_t total = off_d + sizeof(double) * 2 * (size_t)n;
  VS_TRY(vs_reserve(ctx, &ctx->d_ba, total));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, total));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint8_t* h = (uint8_t*)ctx->h_pin_big.p;
  uint8_t* d = (uint8_t*)ctx->d_ba.p;
  double* hm = (double*)h;
  memcpy(hm, P1, 12 * sizeof(double));
  memcpy(hm + 12, P2, 12 * sizeof(double));
  if (depth) {
    memcpy(hm + 24, T1, 12 * sizeof(double));
    memcpy(hm + 36, T2, 12 * sizeof(double));
  }
  memcpy(h + off_p1, pts1, pts_bytes);
  memcpy(h + off_p2, pts2, pts_bytes);
  hipStream_t s = ctx->stream;
  VS_HIP(ctx, hipMemcpyAsync(d, h, off_x, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(triangulate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double*)d,
                     (const double*)(d + off_p1), (const double*)(d + off_p2), n, stride, (double*)(d + off_x),
                     depth ? (const double*)d + 24 : (const double*)nullptr, depth ? (double*)(d + off_d) : (double*)nullptr);
  VS_LAUNCH_CHECK(ctx, "triangulate_kernel");
  VS_HIP(ctx, hipMemcpyAsync(h + off_x, d + off_x, total - off_x, hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  memcpy(X4, h + off_x, sizeof(double) * 4 * (size_t)n);
  if (depth) memcpy(depth, h + off_d, sizeof(double) * 2 * (size_t)n);
  return VS_OK;
}
