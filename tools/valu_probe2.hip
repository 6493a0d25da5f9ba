// valu_probe2.hip -- issue rate of candidate VALU ops for the Hamming kernel's bookkeeping (dev tool, round 2).
// Question: which ops besides v_fma_f32 issue at the 2-cycle wave64 rate on gfx950?  (r01: xor/bcnt/min_u32/med3_u32 = 4)
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe2 tools/valu_probe2.hip && /tmp/valu_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x

// 8 independent chains, register operands only; OPSTR uses %0..%7 as in/out and %8 as a scalar
#define CHAIN2(op)                                                                                                   \
  op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" \
     op " %6, %6, %7\n" op " %7, %7, %0"
#define CHAIN3(op)                                                                                                  \
  op " %0, %0, %1, %2\n" op " %1, %1, %2, %3\n" op " %2, %2, %3, %4\n" op " %3, %3, %4, %5\n" op " %4, %4, %5, %6\n" \
     op " %5, %5, %6, %7\n" op " %6, %6, %7, %0\n" op " %7, %7, %0, %1"

#define DEF_PROBE(NAME, BODY)                                                                              \
  __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                                         \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, \
             a6 = a0 * 17, a7 = a0 * 19;                                                                   \
    for (int i = 0; i < iters; ++i) {                                                                      \
      REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                   \
  }

DEF_PROBE(p_min_f32, CHAIN2("v_min_f32"))
DEF_PROBE(p_max_f32, CHAIN2("v_max_f32"))
DEF_PROBE(p_med3_f32, CHAIN3("v_med3_f32"))
DEF_PROBE(p_min3_f32, CHAIN3("v_min3_f32"))
DEF_PROBE(p_min3_u32, CHAIN3("v_min3_u32"))
DEF_PROBE(p_add_f32, CHAIN2("v_add_f32"))
DEF_PROBE(p_mul_f32, CHAIN2("v_mul_f32"))
DEF_PROBE(p_and_b32, CHAIN2("v_and_b32"))
DEF_PROBE(p_or_b32, CHAIN2("v_or_b32"))
DEF_PROBE(p_xnor_b32, CHAIN2("v_xnor_b32"))
DEF_PROBE(p_bfi_b32, CHAIN3("v_bfi_b32"))
DEF_PROBE(p_perm_b32, CHAIN3("v_perm_b32"))
DEF_PROBE(p_pk_min_u16, CHAIN2("v_pk_min_u16"))
DEF_PROBE(p_pk_max_u16, CHAIN2("v_pk_max_u16"))
DEF_PROBE(p_pk_add_u16, CHAIN2("v_pk_add_u16"))
DEF_PROBE(p_pk_add_f16, CHAIN2("v_pk_add_f16"))
DEF_PROBE(p_pk_min_f16, CHAIN2("v_pk_min_f16"))
DEF_PROBE(p_add3_u32, CHAIN3("v_add3_u32"))
DEF_PROBE(p_and_or_b32, CHAIN3("v_and_or_b32"))
DEF_PROBE(p_or3_b32, CHAIN3("v_or3_b32"))
DEF_PROBE(p_lshl_add_u32, CHAIN3("v_lshl_add_u32"))
DEF_PROBE(p_mad_u32_u24, CHAIN3("v_mad_u32_u24"))
DEF_PROBE(p_sad_u8, CHAIN3("v_sad_u8"))
DEF_PROBE(p_sad_u16, CHAIN3("v_sad_u16"))
DEF_PROBE(p_sad_u32, CHAIN3("v_sad_u32"))
DEF_PROBE(p_msad_u8, CHAIN3("v_msad_u8"))
DEF_PROBE(p_dot4_u32_u8, CHAIN3("v_dot4_u32_u8"))
DEF_PROBE(p_dot8_u32_u4, CHAIN3("v_dot8_u32_u4"))
DEF_PROBE(p_dot2_u32_u16, CHAIN3("v_dot2_u32_u16"))
DEF_PROBE(p_alignbit, CHAIN3("v_alignbit_b32"))
DEF_PROBE(p_bfe_u32, CHAIN3("v_bfe_u32"))
DEF_PROBE(p_cndmask, CHAIN2("v_cndmask_b32"))
DEF_PROBE(p_mov_b32,
          "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n "
          "v_mov_b32 %6, %7\n v_mov_b32 %7, %0")
DEF_PROBE(p_fma_f32, CHAIN3("v_fma_f32"))
DEF_PROBE(p_xor_b32, CHAIN2("v_xor_b32"))
DEF_PROBE(p_bcnt, CHAIN2("v_bcnt_u32_b32"))
// the matcher's inner pattern for one (query, train word): xor then bcnt-accumulate, interleaved over 4 chains
DEF_PROBE(p_xor_bcnt_pair,
          "v_xor_b32 %4, %0, %1\n v_bcnt_u32_b32 %0, %4, %0\n v_xor_b32 %5, %1, %2\n v_bcnt_u32_b32 %1, %5, %1\n "
          "v_xor_b32 %6, %2, %3\n v_bcnt_u32_b32 %2, %6, %2\n v_xor_b32 %7, %3, %0\n v_bcnt_u32_b32 %3, %7, %3")
// mixed: f32 min / med3 between integer ops (does the 2-cycle op pair up with a 4-cycle neighbour?)
DEF_PROBE(p_mix_bcnt_minf,
          "v_bcnt_u32_b32 %0, %4, %0\n v_min_f32 %4, %4, %5\n v_bcnt_u32_b32 %1, %5, %1\n v_min_f32 %5, %5, %6\n "
          "v_bcnt_u32_b32 %2, %6, %2\n v_min_f32 %6, %6, %7\n v_bcnt_u32_b32 %3, %7, %3\n v_min_f32 %7, %7, %4")

typedef void (*kfn)(uint32_t*, int, uint32_t);

static void run(const char* name, kfn fn, uint32_t* d) {
  const int iters = 1500;  // 64 instr per iter
  printf("%-16s", name);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fn<<<grid, block>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fn<<<grid, block>>>(d, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_instr = ms * 1e6 / ((double)iters * 64 * wps);
    printf("  w%d %.2f cyc", wps, ns_per_instr * 2.4);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  printf("   (cycles per wave64 instr per SIMD @2.4GHz)\n");
}

#define RUN(n) run(#n, n, d)
int main() {
  uint32_t* d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  RUN(p_fma_f32); RUN(p_xor_b32); RUN(p_bcnt); RUN(p_xor_bcnt_pair);
  RUN(p_min_f32); RUN(p_max_f32); RUN(p_med3_f32); RUN(p_min3_f32); RUN(p_min3_u32); RUN(p_add_f32); RUN(p_mul_f32);
  RUN(p_mix_bcnt_minf);
  RUN(p_and_b32); RUN(p_or_b32); RUN(p_xnor_b32); RUN(p_bfi_b32); RUN(p_perm_b32);
  RUN(p_pk_min_u16); RUN(p_pk_max_u16); RUN(p_pk_add_u16); RUN(p_pk_add_f16); RUN(p_pk_min_f16);
  RUN(p_add3_u32); RUN(p_and_or_b32); RUN(p_or3_b32); RUN(p_lshl_add_u32); RUN(p_mad_u32_u24);
  RUN(p_sad_u8); RUN(p_sad_u16); RUN(p_sad_u32); RUN(p_msad_u8);
  RUN(p_dot4_u32_u8); RUN(p_dot8_u32_u4); RUN(p_dot2_u32_u16);
  RUN(p_alignbit); RUN(p_bfe_u32); RUN(p_cndmask); RUN(p_mov_b32);
  hipFree(d);
  return 0;
}
