// dev aid: does a wave's VALU work wait for another wave's v_mfma_f64_16x16x4_f64 on the same SIMD?  One workgroup of 512 threads:
// waves 0-3 (one per SIMD) issue MFMAs back to back, waves 4-7 (their SIMD neighbours) run a chain of FP64 / FP32 / integer VALU
// instructions or LDS stores; nanoseconds per instruction of the second kind with and without the MFMA waves running.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma64_probe2 tools/mfma64_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void probe(double* out, int iters, int with_mfma, int indep) {
  __shared__ double lds[2048];
  const int wv = threadIdx.x >> 6;
  double r = 0;
  if (wv < 4) {
    if (with_mfma) {
      d4 acc[3] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
      double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
      r = acc[0][0] + acc[1][1] + acc[2][2];
    }
  } else {
    const int n = iters / 4;
    const long long w0 = wall_clock64();
    if (KIND == 0) {  // FP64 multiply / add, eight independent chains
      double x[8];
      for (int k = 0; k < 8; ++k) x[k] = 1.0 + threadIdx.x * 1e-9 * (k + 1);
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] * 1.0000001 + 1e-9;
      for (int k = 0; k < 8; ++k) r += x[k];
    } else if (KIND == 1) {  // FP32
      float x[8];
      for (int k = 0; k < 8; ++k) x[k] = 1.0f + threadIdx.x * 1e-6f * (k + 1);
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] * 1.0000001f + 1e-9f;
      for (int k = 0; k < 8; ++k) r += x[k];
    } else if (KIND == 2) {  // integer
      unsigned x[8];
      for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 2654435761u + k;
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (x[k] ^ (x[k] >> 3)) + 0x9e3779b9u;
      for (int k = 0; k < 8; ++k) r += x[k];
    } else {  // LDS stores (16 bytes)
      double2* p = reinterpret_cast<double2*>(lds) + (threadIdx.x - 256);
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          p[256 * k] = double2{(double)i, (double)k};
          asm volatile("" ::: "memory");
        }
      r = lds[threadIdx.x & 255];
    }
    const long long w1 = wall_clock64();
    if (threadIdx.x == 256) out[512 * 255] = (double)(w1 - w0) + (r == 12345.678 ? 1 : 0);  // 100 MHz ticks of wave 4
    return;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + indep;
}
template <int KIND>
void run(const char* name, int per_iter) {
  double* out;
  hipMalloc(&out, sizeof(double) * 512 * 256);
  const int iters = 40000;
  for (int with = 0; with < 2; ++with) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<KIND><<<1, 512>>>(out, iters, with, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<1, 512>>>(out, iters, with, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ticks;
    hipMemcpy(&ticks, out + 512 * 255, 8, hipMemcpyDeviceToHost);
    printf("%-28s %s MFMA waves: kernel %.1f us (%.1f ns per MFMA); second wave alone %.1f us = %.2f ns per instruction\n", name,
           with ? "with   " : "without", ms * 1e3, ms * 1e6 / (3.0 * iters), ticks * 0.01, ticks * 10.0 / ((double)(iters / 4) * per_iter));
  }
  hipFree(out);
}
int main() {
  run<0>("FP64 mul+add (16 per iter)", 16);
  run<1>("FP32 mul+add (16 per iter)", 16);
  run<2>("integer (24 per iter)", 24);
  run<3>("LDS 16-byte stores (4)", 4);
  return 0;
}
