// dev aid: v_mfma_f64_16x16x4_f64 rate of ONE wave per SIMD when its operands (a) are the same registers every time, (b) rotate over
// registers, (c) are read from LDS one k step ahead (five tiles per step, as ba_schur_window_ws does).
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma64_probe3 tools/mfma64_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NQ = 5, STRIDE = 112;
template <int MODE>
__global__ void probe(double* out, long long* ticks, int iters) {
  __shared__ double sY[24 * STRIDE], sH[24 * STRIDE];
  for (int i = threadIdx.x; i < 24 * STRIDE; i += blockDim.x) {
    sY[i] = 1e-3 * i;
    sH[i] = 1.0 + 1e-4 * i;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, krow = lane >> 4, kcol = lane & 15;
  d4 acc[NQ];
  for (int k = 0; k < NQ; ++k) acc[k] = d4{0, 0, 0, 0};
  const long long t0 = wall_clock64();
  if (MODE == 0) {
    double a = lane * 1e-3, b = 1.0 + lane * 1e-4;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  } else if (MODE == 1) {
    double a[NQ], b[NQ];
    for (int q = 0; q < NQ; ++q) {
      a[q] = lane * 1e-3 * (q + 1);
      b[q] = 1.0 + lane * 1e-4 * (q + 1);
    }
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[(q + ks) % NQ], acc[q], 0, 0, 0);
  } else {
    for (int i = 0; i < iters; ++i) {
      double av[2][NQ], bv[2][NQ];
      const int o0 = krow * STRIDE + kcol;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        av[0][q] = sY[o0 + 16 * (q % 4)];
        bv[0][q] = sH[o0 + 16 * ((q + 1) % 4)];
      }
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        if (ks + 1 < 6) {
          const int o = (4 * (ks + 1) + krow) * STRIDE + kcol;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            av[(ks + 1) & 1][q] = sY[o + 16 * (q % 4)];
            bv[(ks + 1) & 1][q] = sH[o + 16 * ((q + 1) % 4)];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks & 1][q], bv[ks & 1][q], acc[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  const long long t1 = wall_clock64();
  double s = 0;
  for (int k = 0; k < NQ; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int blocks) {
  double* out;
  long long* ticks;
  hipMalloc(&out, sizeof(double) * threads * blocks);
  hipMalloc(&ticks, 8);
  const int iters = 2000;
  probe<MODE><<<blocks, threads>>>(out, ticks, iters);
  hipDeviceSynchronize();
  probe<MODE><<<blocks, threads>>>(out, ticks, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-58s %4d threads x %3d blocks: %6.1f ns per MFMA of one wave\n", name, threads, blocks, c * 10.0 / (iters * 6.0 * NQ));
  hipFree(out);
  hipFree(ticks);
}
int main() {
  run<0>("same operand registers", 256, 1);
  run<1>("operand registers rotate", 256, 1);
  run<2>("operands from LDS, one k step ahead", 256, 1);
  run<2>("operands from LDS, one k step ahead", 128, 1);
  run<3>("... and a workgroup barrier per six k steps", 256, 1);
  run<2>("operands from LDS, one k step ahead", 256, 256);
  run<2>("operands from LDS, one k step ahead", 128, 256);
  return 0;
}
