// dev probe: what a dependent kernel boundary costs in a stream, launch by launch against a captured hipGraph (gfx950).
// Forty kernels of ~10 us that each read what the previous one wrote (the shape of a BA trial sequence).
// build: hipcc --offload-arch=gfx950 -O3 -o build/graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(const double* in, double* out, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = i < n ? in[i] : 0.0;
  for (int k = 0; k < spin; ++k) v = v * 1.0000001 + 1e-9;
  if (i < n) out[i] = v;
}

int main() {
  const int n = 64 * 256, chain = 40;
  double *a, *b;
  CK(hipMalloc(&a, n * sizeof(double)));
  CK(hipMalloc(&b, n * sizeof(double)));
  CK(hipMemset(a, 0, n * sizeof(double)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int spin : {0, 600, 2400}) {
    auto enqueue = [&]() {
      for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n, spin);
    };
    // one kernel alone
    float one = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, a, b, n, spin);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      one = std::min(one, ms);
    }
    std::vector<float> plain, graph;
    for (int rep = 0; rep < 30; ++rep) {
      CK(hipEventRecord(e0, s));
      enqueue();
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      plain.push_back(ms);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 30; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      graph.push_back(ms);
    }
    std::sort(plain.begin(), plain.end());
    std::sort(graph.begin(), graph.end());
    printf("spin %4d: one kernel between events %.2f us; chain of %d: launches %.2f us per kernel (median; min %.2f), graph %.2f us per kernel (median; min %.2f)\n", spin,
           one * 1e3, chain, plain[15] * 1e3 / chain, plain[0] * 1e3 / chain, graph[15] * 1e3 / chain, graph[0] * 1e3 / chain);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
