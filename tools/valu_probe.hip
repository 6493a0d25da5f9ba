// valu_probe.hip -- measures the issue rate of the integer VALU ops the Hamming kernel uses (dev tool).
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void probe(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t s = seed | 1;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
    if (OP == 1) { REP8(asm volatile("v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %4\n v_bcnt_u32_b32 %4, %4, %5\n v_bcnt_u32_b32 %5, %5, %6\n v_bcnt_u32_b32 %6, %6, %7\n v_bcnt_u32_b32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 2) { REP8(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
    if (OP == 3) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
    if (OP == 4) { REP8(asm volatile("v_med3_u32 %0, %0, %1, %2\n v_med3_u32 %1, %1, %2, %3\n v_med3_u32 %2, %2, %3, %4\n v_med3_u32 %3, %3, %4, %5\n v_med3_u32 %4, %4, %5, %6\n v_med3_u32 %5, %5, %6, %7\n v_med3_u32 %6, %6, %7, %0\n v_med3_u32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 5) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
    if (OP == 6) { REP8(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %4\n v_min_u32 %4, %4, %5\n v_min_u32 %5, %5, %6\n v_min_u32 %6, %6, %7\n v_min_u32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int OP>
void run(const char* name, uint32_t* d) {
  const int iters = 2000;  // 64 instr per iter
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    dim3 grid(256 * wps), block(256);  // 4 waves per block -> one per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<OP><<<grid, block>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<grid, block>>>(d, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 64 * wps;
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD %d : %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)  chip %.1f Tlane-ops/s\n", name, wps, ms,
           ns_per_instr, ns_per_instr * 2.4, 1024.0 * 64 / ns_per_instr / 1e3);
  }
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_xor_b32", d);
  run<1>("v_bcnt_u32_b32", d);
  run<2>("v_add_u32", d);
  run<3>("v_fma_f32", d);
  run<4>("v_med3_u32", d);
  run<5>("v_lshl_or_b32", d);
  run<6>("v_min_u32", d);
  return 0;
}
