// dev aid: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- cycles per instruction for one wave per SIMD with 1..6 independent
// accumulators, and with two / four waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o build/mfma64_probe tools/mfma64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ void probe(double* out, long long* cyc, int iters) {
  d4 acc[K];
  for (int k = 0; k < K; ++k) acc[k] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < K; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int K>
void run(int threads, int blocks, int iters) {
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * threads * blocks);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<K><<<blocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<K><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * K;
  printf("accumulators %d, %4d threads x %4d blocks: %7.1f counter ticks per MFMA of one wave, %7.1f ns per MFMA per wave; %.2f TFLOP/s\n", K, threads, blocks, c / n,
         ms * 1e6 / n, 2048.0 * n * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(cyc);
}
int main() {
  const int it = 20000;
  run<1>(256, 1, it);
  run<2>(256, 1, it);
  run<3>(256, 1, it);
  run<6>(256, 1, it);
  run<3>(512, 1, it);
  run<3>(1024, 1, it);
  run<3>(256, 256, it);
  run<3>(512, 256, it);
  run<3>(256, 512, it);
  run<6>(256, 1024, it);
  return 0;
}
