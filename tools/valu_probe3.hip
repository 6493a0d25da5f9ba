// valu_probe3.hip -- follow-up to valu_probe2: does the 2-cycle rate of v_xor/v_and/v_or survive (a) an SGPR operand,
// (b) mixing with 4-cycle ops (v_bcnt), and how should the mix be ordered?  (dev tool, round 2)
// build: hipcc --offload-arch=gfx950 -O3 -o build/valu_probe3 tools/valu_probe3.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x

#define DEF_PROBE(NAME, NINSTR, BODY)                                                                      \
  __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                                         \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, \
             a6 = a0 * 17, a7 = a0 * 19, b0 = a0 * 23, b1 = a0 * 29, b2 = a0 * 31, b3 = a0 * 37,           \
             b4 = a0 * 41, b5 = a0 * 43, b6 = a0 * 47, b7 = a0 * 53;                                       \
    uint32_t s = seed | 1;                                                                                 \
    for (int i = 0; i < iters; ++i) {                                                                      \
      REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                        "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "s"(s));) \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
  }                                                                                                        \
  static const int NAME##_n = NINSTR;

// %0..%7 = a, %8..%15 = b, %16 = sgpr
DEF_PROBE(xor_vv, 8, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %10\n v_xor_b32 %3, %3, %11\n v_xor_b32 %4, %4, %12\n v_xor_b32 %5, %5, %13\n v_xor_b32 %6, %6, %14\n v_xor_b32 %7, %7, %15")
DEF_PROBE(xor_sv, 8, "v_xor_b32 %0, %16, %0\n v_xor_b32 %1, %16, %1\n v_xor_b32 %2, %16, %2\n v_xor_b32 %3, %16, %3\n v_xor_b32 %4, %16, %4\n v_xor_b32 %5, %16, %5\n v_xor_b32 %6, %16, %6\n v_xor_b32 %7, %16, %7")
DEF_PROBE(xor_const, 8, "v_xor_b32 %0, 5, %0\n v_xor_b32 %1, 5, %1\n v_xor_b32 %2, 5, %2\n v_xor_b32 %3, 5, %3\n v_xor_b32 %4, 5, %4\n v_xor_b32 %5, 5, %5\n v_xor_b32 %6, 5, %6\n v_xor_b32 %7, 5, %7")
DEF_PROBE(and_sv, 8, "v_and_b32 %0, %16, %0\n v_and_b32 %1, %16, %1\n v_and_b32 %2, %16, %2\n v_and_b32 %3, %16, %3\n v_and_b32 %4, %16, %4\n v_and_b32 %5, %16, %5\n v_and_b32 %6, %16, %6\n v_and_b32 %7, %16, %7")
DEF_PROBE(add_u32_vv, 8, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %10\n v_add_u32 %3, %3, %11\n v_add_u32 %4, %4, %12\n v_add_u32 %5, %5, %13\n v_add_u32 %6, %6, %14\n v_add_u32 %7, %7, %15")
DEF_PROBE(sub_u32_vv, 8, "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %9\n v_sub_u32 %2, %2, %10\n v_sub_u32 %3, %3, %11\n v_sub_u32 %4, %4, %12\n v_sub_u32 %5, %5, %13\n v_sub_u32 %6, %6, %14\n v_sub_u32 %7, %7, %15")
DEF_PROBE(lshl_imm, 8, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7")
DEF_PROBE(lshr_imm, 8, "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7")
DEF_PROBE(not_b32, 8, "v_not_b32 %0, %8\n v_not_b32 %1, %9\n v_not_b32 %2, %10\n v_not_b32 %3, %11\n v_not_b32 %4, %12\n v_not_b32 %5, %13\n v_not_b32 %6, %14\n v_not_b32 %7, %15")
DEF_PROBE(min_u32_vv, 8, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %9\n v_min_u32 %2, %2, %10\n v_min_u32 %3, %3, %11\n v_min_u32 %4, %4, %12\n v_min_u32 %5, %5, %13\n v_min_u32 %6, %6, %14\n v_min_u32 %7, %7, %15")
DEF_PROBE(max_f32_vv, 8, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_max_f32 %2, %2, %10\n v_max_f32 %3, %3, %11\n v_max_f32 %4, %4, %12\n v_max_f32 %5, %5, %13\n v_max_f32 %6, %6, %14\n v_max_f32 %7, %7, %15")
DEF_PROBE(sub_f32_vv, 8, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %10\n v_sub_f32 %3, %3, %11\n v_sub_f32 %4, %4, %12\n v_sub_f32 %5, %5, %13\n v_sub_f32 %6, %6, %14\n v_sub_f32 %7, %7, %15")
// bcnt only, 8 independent accumulate chains (b = popc(a) + b)
DEF_PROBE(bcnt8, 8, "v_bcnt_u32_b32 %8, %0, %8\n v_bcnt_u32_b32 %9, %1, %9\n v_bcnt_u32_b32 %10, %2, %10\n v_bcnt_u32_b32 %11, %3, %11\n v_bcnt_u32_b32 %12, %4, %12\n v_bcnt_u32_b32 %13, %5, %13\n v_bcnt_u32_b32 %14, %6, %14\n v_bcnt_u32_b32 %15, %7, %15")
// the matcher's distance chain for ONE (query, row): 8 x (xor -> bcnt accumulate), strictly alternating, xor VGPR x VGPR
DEF_PROBE(chain_alt_vv, 16, "v_xor_b32 %8, %0, %1\n v_bcnt_u32_b32 %15, %8, %15\n v_xor_b32 %9, %1, %2\n v_bcnt_u32_b32 %15, %9, %15\n v_xor_b32 %10, %2, %3\n v_bcnt_u32_b32 %15, %10, %15\n v_xor_b32 %11, %3, %4\n v_bcnt_u32_b32 %15, %11, %15\n v_xor_b32 %12, %4, %5\n v_bcnt_u32_b32 %15, %12, %15\n v_xor_b32 %13, %5, %6\n v_bcnt_u32_b32 %15, %13, %15\n v_xor_b32 %14, %6, %7\n v_bcnt_u32_b32 %15, %14, %15\n v_xor_b32 %8, %7, %0\n v_bcnt_u32_b32 %15, %8, %15")
// same work, grouped: 8 xors first, then the 8-long bcnt chain
DEF_PROBE(chain_grp_vv, 16, "v_xor_b32 %8, %0, %1\n v_xor_b32 %9, %1, %2\n v_xor_b32 %10, %2, %3\n v_xor_b32 %11, %3, %4\n v_xor_b32 %12, %4, %5\n v_xor_b32 %13, %5, %6\n v_xor_b32 %14, %6, %7\n v_xor_b32 %15, %7, %0\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %10, %0\n v_bcnt_u32_b32 %0, %11, %0\n v_bcnt_u32_b32 %0, %12, %0\n v_bcnt_u32_b32 %0, %13, %0\n v_bcnt_u32_b32 %0, %14, %0\n v_bcnt_u32_b32 %0, %15, %0")
// same with the SGPR as one xor operand (what the scalar-load kernel does)
DEF_PROBE(chain_alt_sv, 16, "v_xor_b32 %8, %16, %1\n v_bcnt_u32_b32 %15, %8, %15\n v_xor_b32 %9, %16, %2\n v_bcnt_u32_b32 %15, %9, %15\n v_xor_b32 %10, %16, %3\n v_bcnt_u32_b32 %15, %10, %15\n v_xor_b32 %11, %16, %4\n v_bcnt_u32_b32 %15, %11, %15\n v_xor_b32 %12, %16, %5\n v_bcnt_u32_b32 %15, %12, %15\n v_xor_b32 %13, %16, %6\n v_bcnt_u32_b32 %15, %13, %15\n v_xor_b32 %14, %16, %7\n v_bcnt_u32_b32 %15, %14, %15\n v_xor_b32 %8, %16, %0\n v_bcnt_u32_b32 %15, %8, %15")
// two independent distance chains interleaved (what the compiler's schedule looks like), xor VGPR x VGPR
DEF_PROBE(chain_2way_vv, 16, "v_xor_b32 %8, %0, %1\n v_xor_b32 %9, %2, %3\n v_bcnt_u32_b32 %14, %8, %14\n v_bcnt_u32_b32 %15, %9, %15\n v_xor_b32 %10, %1, %2\n v_xor_b32 %11, %3, %4\n v_bcnt_u32_b32 %14, %10, %14\n v_bcnt_u32_b32 %15, %11, %15\n v_xor_b32 %8, %4, %5\n v_xor_b32 %9, %5, %6\n v_bcnt_u32_b32 %14, %8, %14\n v_bcnt_u32_b32 %15, %9, %15\n v_xor_b32 %10, %6, %7\n v_xor_b32 %11, %7, %0\n v_bcnt_u32_b32 %14, %10, %14\n v_bcnt_u32_b32 %15, %11, %15")

typedef void (*kfn)(uint32_t*, int, uint32_t);

static void run(const char* name, kfn fn, int ninstr, uint32_t* d) {
  const int iters = 1500;
  printf("%-16s", name);
  for (int wps : {1, 2, 4, 5, 8}) {  // waves per SIMD
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    fn<<<grid, block>>>(d, 10, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    fn<<<grid, block>>>(d, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_instr = ms * 1e6 / ((double)iters * 8 * ninstr * wps);
    printf("  w%d %.3f ns", wps, ns_per_instr);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  printf("   (ns per wave64 instruction per SIMD)\n");
}

#define RUN(n) run(#n, n, n##_n, d)
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  RUN(xor_vv); RUN(xor_sv); RUN(xor_const); RUN(and_sv); RUN(add_u32_vv); RUN(sub_u32_vv); RUN(lshl_imm); RUN(lshr_imm);
  RUN(not_b32); RUN(min_u32_vv); RUN(max_f32_vv); RUN(sub_f32_vv); RUN(bcnt8);
  RUN(chain_alt_vv); RUN(chain_grp_vv); RUN(chain_alt_sv); RUN(chain_2way_vv);
  (void)hipFree(d);
  return 0;
}
