// fence_probe.hip -- what an in-launch "everybody is through" costs as a function of the number of workgroups (dev tool).
// Pricing for kernels that hand over to a consumer that is already resident and waits in-kernel (the tracking chain does it
// with 8-20 producer workgroups): every producer workgroup writes a little, fences, takes a ticket; the last one publishes a
// tag.  Variants: no fence (kernel end only), agent-scope fence + ticket, system-scope fence + ticket.
//   hipcc --offload-arch=gfx950 -O3 tools/fence_probe.hip -o /tmp/fence_probe && /tmp/fence_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int MODE>  // 0: none, 1: agent fence + ticket, 2: system fence + ticket, 3: agent fence only, 4: ticket only (no fence)
__global__ void k_produce(double* out, int per_wg, unsigned* ticket, unsigned* tag, unsigned want) {
  double* p = out + (size_t)blockIdx.x * per_wg;
  for (int i = threadIdx.x; i < per_wg; i += blockDim.x) p[i] = (double)(i + want);
  if (MODE == 0) return;
  if (MODE == 1 || MODE == 3) __threadfence();
  else if (MODE == 2) __threadfence_system();
  __syncthreads();
  if (MODE == 3) return;
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tag, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int MODE>
static double run(int wgs, int threads, int per_wg, double* d_out, unsigned* d_sync) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ms;
  for (int rep = 0; rep < 30; ++rep) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(k_produce<MODE>, dim3(wgs), dim3(threads), 0, 0, d_out, per_wg, d_sync, d_sync + 64, (unsigned)(rep * 50 + k + 1));
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    ms.push_back(t / 50);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2] * 1e3;
}

int main() {
  double* d_out;
  unsigned* d_sync;
  hipMalloc(&d_out, sizeof(double) * 2048 * 4096);
  hipMalloc(&d_sync, 1024);
  hipMemset(d_sync, 0, 1024);
  printf("per launch, median of 30 x 50 back-to-back launches (us); every workgroup writes 4 KB\n");
  printf("%8s %8s | %10s %14s %15s %12s %12s\n", "wgs", "threads", "no fence", "agent + ticket", "system + ticket", "fence only", "ticket only");
  for (int wgs : {1, 8, 20, 64, 250, 330, 1024}) {
    const int threads = 256, per = 512;
    const double a = run<0>(wgs, threads, per, d_out, d_sync), b = run<1>(wgs, threads, per, d_out, d_sync), c = run<2>(wgs, threads, per, d_out, d_sync);
    const double d = run<3>(wgs, threads, per, d_out, d_sync), e = run<4>(wgs, threads, per, d_out, d_sync);
    printf("%8d %8d | %10.2f %14.2f %15.2f %12.2f %12.2f\n", wgs, threads, a, b, c, d, e);
  }
  return 0;
}
