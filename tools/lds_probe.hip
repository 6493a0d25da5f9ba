// lds_probe.hip -- latency of the primitives the BA kernels chain together (dev tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k_barrier(int iters, double* out) {
  __shared__ double s[256];
  s[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < iters; ++i) __syncthreads();
  out[threadIdx.x] = s[threadIdx.x];
}
__global__ void k_lds_chain(int iters, double* out) {  // dependent read -> fma -> write -> barrier-free
  __shared__ double s[512];
  s[threadIdx.x] = threadIdx.x + 1.0;
  s[threadIdx.x + 256] = 0.5;
  __syncthreads();
  double acc = 0;
  int idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    double v = s[idx];
    s[idx] = v * 1.0000001 + acc;
    idx = (idx + 1) & 255;  // next iteration reads what a neighbour wrote (same wave for most lanes)
    acc = v * 1e-9;
  }
  out[threadIdx.x] = acc + s[threadIdx.x];
}
__global__ void k_lds_rmw_barrier(int iters, double* out) {  // read, fma, write, __syncthreads per step
  __shared__ double s[512];
  s[threadIdx.x] = threadIdx.x + 1.0;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    double v = s[(threadIdx.x + 1) & 255];
    __syncthreads();
    s[threadIdx.x] = v * 1.0000001;
    __syncthreads();
  }
  out[threadIdx.x] = s[threadIdx.x];
}
__global__ void k_sqrt_div(int iters, double* out) {
  double x = 2.0 + threadIdx.x;
  for (int i = 0; i < iters; ++i) x = 1.0 / sqrt(x) + 3.0;
  out[threadIdx.x] = x;
}
__global__ void k_rsq(int iters, double* out) {
  double x = 2.0 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    x = r + 3.0;
  }
  out[threadIdx.x] = x;
}
__global__ void k_fma_chain(int iters, double* out) {
  double x = 2.0 + threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
  out[threadIdx.x] = x;
}
template <class F>
void timeit(const char* name, F f, int iters) {
  double* d;
  hipMalloc(&d, 4096);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f(10, d);
  hipDeviceSynchronize();
  hipEventRecord(a);
  f(iters, d);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-28s %8.1f ns per iteration (%.0f cycles @2.4GHz)\n", name, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}
int main() {
  const int N = 20000;
  timeit("__syncthreads (256 thr)", [](int n, double* d) { k_barrier<<<1, 256>>>(n, d); }, N);
  timeit("__syncthreads (1024 thr)", [](int n, double* d) { k_barrier<<<1, 1024>>>(n, d); }, N);
  timeit("lds read->fma->write chain", [](int n, double* d) { k_lds_chain<<<1, 256>>>(n, d); }, N);
  timeit("lds rd,bar,wr,bar (256)", [](int n, double* d) { k_lds_rmw_barrier<<<1, 256>>>(n, d); }, N);
  timeit("f64 1/sqrt chain", [](int n, double* d) { k_sqrt_div<<<1, 64>>>(n, d); }, N);
  timeit("f64 rsq+2 newton chain", [](int n, double* d) { k_rsq<<<1, 64>>>(n, d); }, N);
  timeit("f64 fma chain", [](int n, double* d) { k_fma_chain<<<1, 64>>>(n, d); }, N);
  return 0;
}
