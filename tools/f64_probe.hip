// dev probe: issue cost and dependent latency of the FP64 vector instructions the bundle adjustment's factorisations are made of
// (gfx950), at one, two and four waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o build/f64_probe tools/f64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REGS "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s4", "s5"
#define INIT asm volatile("v_mov_b32 v8, 0\nv_mov_b32 v9, 0x3ff00000\nv_mov_b32 v10, 1\nv_mov_b32 v11, 0x3ff00000\nv_mov_b32 v12, 2\nv_mov_b32 v13, 0x3ff00000\nv_mov_b32 v14, 3\nv_mov_b32 v15, 0x3ff00000\n" \
  "v_mov_b32 v16, 4\nv_mov_b32 v17, 0x3ff00000\nv_mov_b32 v18, 5\nv_mov_b32 v19, 0x3ff00000\nv_mov_b32 v20, 6\nv_mov_b32 v21, 0x3ff00000\nv_mov_b32 v22, 7\nv_mov_b32 v23, 0x3ff00000\n" \
  "v_mov_b32 v24, 9\nv_mov_b32 v25, 0x3ff00000\nv_mov_b32 v26, 11\nv_mov_b32 v27, 0x3ff00000" ::: REGS)

#define KERNEL(name, n_, body)                                                              \
  __global__ void name(double* out, int iters, long long* cyc) {                            \
    INIT;                                                                                   \
    const long long t0 = __builtin_readcyclecounter();                                      \
    for (int i = 0; i < iters; ++i) asm volatile(body ::: REGS);                            \
    const long long t1 = __builtin_readcyclecounter();                                      \
    double r;                                                                               \
    unsigned lo_, hi_;\
    asm volatile("v_mov_b32 %0, v8\nv_mov_b32 %1, v9" : "=v"(lo_), "=v"(hi_)::REGS);                \
    r = __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                \
  }                                                                                         \
  static const int name##_n = n_;

KERNEL(mul_indep, 8, "v_mul_f64 v[8:9], v[24:25], v[26:27]\nv_mul_f64 v[10:11], v[24:25], v[26:27]\nv_mul_f64 v[12:13], v[24:25], v[26:27]\nv_mul_f64 v[14:15], v[24:25], v[26:27]\nv_mul_f64 v[16:17], v[24:25], v[26:27]\nv_mul_f64 v[18:19], v[24:25], v[26:27]\nv_mul_f64 v[20:21], v[24:25], v[26:27]\nv_mul_f64 v[22:23], v[24:25], v[26:27]")
KERNEL(add_indep, 8, "v_add_f64 v[8:9], v[24:25], v[26:27]\nv_add_f64 v[10:11], v[24:25], v[26:27]\nv_add_f64 v[12:13], v[24:25], v[26:27]\nv_add_f64 v[14:15], v[24:25], v[26:27]\nv_add_f64 v[16:17], v[24:25], v[26:27]\nv_add_f64 v[18:19], v[24:25], v[26:27]\nv_add_f64 v[20:21], v[24:25], v[26:27]\nv_add_f64 v[22:23], v[24:25], v[26:27]")
KERNEL(fma_indep, 8, "v_fma_f64 v[8:9], v[24:25], v[26:27], v[8:9]\nv_fma_f64 v[10:11], v[24:25], v[26:27], v[10:11]\nv_fma_f64 v[12:13], v[24:25], v[26:27], v[12:13]\nv_fma_f64 v[14:15], v[24:25], v[26:27], v[14:15]\nv_fma_f64 v[16:17], v[24:25], v[26:27], v[16:17]\nv_fma_f64 v[18:19], v[24:25], v[26:27], v[18:19]\nv_fma_f64 v[20:21], v[24:25], v[26:27], v[20:21]\nv_fma_f64 v[22:23], v[24:25], v[26:27], v[22:23]")
KERNEL(mul_chain, 8, "v_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]\nv_mul_f64 v[8:9], v[8:9], v[26:27]")
KERNEL(add_chain, 8, "v_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]\nv_add_f64 v[8:9], v[8:9], v[26:27]")
KERNEL(fma_chain, 8, "v_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]\nv_fma_f64 v[8:9], v[8:9], v[26:27], v[24:25]")
KERNEL(mulsub_4acc, 8, "v_mul_f64 v[16:17], v[24:25], v[26:27]\nv_mul_f64 v[18:19], v[24:25], v[26:27]\nv_mul_f64 v[20:21], v[24:25], v[26:27]\nv_mul_f64 v[22:23], v[24:25], v[26:27]\nv_add_f64 v[8:9], v[8:9], -v[16:17]\nv_add_f64 v[10:11], v[10:11], -v[18:19]\nv_add_f64 v[12:13], v[12:13], -v[20:21]\nv_add_f64 v[14:15], v[14:15], -v[22:23]")
KERNEL(rsq_indep, 4, "v_rsq_f64 v[8:9], v[24:25]\nv_rsq_f64 v[10:11], v[24:25]\nv_rsq_f64 v[12:13], v[24:25]\nv_rsq_f64 v[14:15], v[24:25]")
KERNEL(rsq_chain, 4, "v_rsq_f64 v[24:25], v[24:25]\nv_rsq_f64 v[24:25], v[24:25]\nv_rsq_f64 v[24:25], v[24:25]\nv_rsq_f64 v[24:25], v[24:25]")
KERNEL(add_f32_indep, 8, "v_add_f32 v8, v24, v26\nv_add_f32 v10, v24, v26\nv_add_f32 v12, v24, v26\nv_add_f32 v14, v24, v26\nv_add_f32 v16, v24, v26\nv_add_f32 v18, v24, v26\nv_add_f32 v20, v24, v26\nv_add_f32 v22, v24, v26")
KERNEL(xor_indep, 8, "v_xor_b32 v8, v24, v26\nv_xor_b32 v10, v24, v26\nv_xor_b32 v12, v24, v26\nv_xor_b32 v14, v24, v26\nv_xor_b32 v16, v24, v26\nv_xor_b32 v18, v24, v26\nv_xor_b32 v20, v24, v26\nv_xor_b32 v22, v24, v26")
KERNEL(bcnt_indep, 8, "v_bcnt_u32_b32 v8, v24, v8\nv_bcnt_u32_b32 v10, v24, v10\nv_bcnt_u32_b32 v12, v24, v12\nv_bcnt_u32_b32 v14, v24, v14\nv_bcnt_u32_b32 v16, v24, v16\nv_bcnt_u32_b32 v18, v24, v18\nv_bcnt_u32_b32 v20, v24, v20\nv_bcnt_u32_b32 v22, v24, v22")
KERNEL(xor_bcnt_mix, 8, "v_xor_b32 v9, v24, v26\nv_bcnt_u32_b32 v8, v9, v8\nv_xor_b32 v11, v25, v27\nv_bcnt_u32_b32 v10, v11, v10\nv_xor_b32 v13, v24, v27\nv_bcnt_u32_b32 v12, v13, v12\nv_xor_b32 v15, v25, v26\nv_bcnt_u32_b32 v14, v15, v14")
KERNEL(readlane_mul, 2, "v_readlane_b32 s4, v8, 3\nv_readlane_b32 s5, v9, 3\nv_mul_f64 v[8:9], s[4:5], v[26:27]")

template <class K>
static void run(const char* name, K k, int per_iter, double* d_out, long long* d_cyc) {
  const int iters = 20000;
  for (int threads : {64, 256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
    const double per = (double)cyc / ((double)iters * per_iter);
    const int waves_per_simd = threads <= 256 ? 1 : threads / 256;  // 768 threads: 3
    printf("%-14s %4d threads (%d wave%s/SIMD%s): %6.2f cycles per instruction of one wave, %6.2f per SIMD issue slot\n", name, threads, waves_per_simd, waves_per_simd > 1 ? "s" : "",
           threads == 64 ? ", one SIMD" : "", per, per / waves_per_simd);
  }
}

// the same streams with every CU busy: `blocks` workgroups of 1024 threads (2 per CU = eight waves per SIMD at this register count)
template <class K>
static void run_chip(const char* name, K k, int per_iter, double* d_out, long long* d_cyc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {1, 256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d_out, iters, d_cyc);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
    const double per_wave = (double)cyc / ((double)iters * per_iter);
    // chip-wide: wave-instructions per second and SIMD, against the clock workgroup 0 saw (its cycles over the kernel's time if it ran the whole time)
    const double winstr = (double)blocks * 16 * iters * per_iter, ns_per_simd_instr = ms * 1e6 / (winstr / 1024.0);
    printf("%-14s %4d workgroups of 1024 threads: %6.2f cycles per instruction of one wave (workgroup 0); kernel %.3f ms = %.3f ns per instruction and SIMD\n", name, blocks,
           per_wave, ms, ns_per_simd_instr);
  }
}

int main() {
  double* d_out;
  long long* d_cyc;
  hipMalloc(&d_out, (size_t)1024 * 1024 * sizeof(double));  // the largest launch: 1024 workgroups of 1024 threads, one double each
  hipMalloc(&d_cyc, sizeof(long long));
#define RUN(name) run(#name, name, name##_n, d_out, d_cyc)
  RUN(mul_indep);
  RUN(add_indep);
  RUN(fma_indep);
  RUN(mulsub_4acc);
  RUN(mul_chain);
  RUN(add_chain);
  RUN(fma_chain);
  RUN(rsq_indep);
  RUN(rsq_chain);
  RUN(add_f32_indep);
  RUN(readlane_mul);
  RUN(xor_indep);
  RUN(bcnt_indep);
  RUN(xor_bcnt_mix);
  run_chip("xor_indep", xor_indep, xor_indep_n, d_out, d_cyc);
  run_chip("bcnt_indep", bcnt_indep, bcnt_indep_n, d_out, d_cyc);
  run_chip("xor_bcnt_mix", xor_bcnt_mix, xor_bcnt_mix_n, d_out, d_cyc);
  run_chip("fma_indep", fma_indep, fma_indep_n, d_out, d_cyc);
  return 0;
}
