#!/usr/bin/env python3
"""Generates tools/valu_probe4.hip: VALU issue rate with EXPLICIT registers -- do the source operands' VGPR banks (v mod 4)
matter for v_xor_b32 (2-cycle class) and v_bcnt_u32_b32 (4-cycle class), and what does the matcher's main-loop body issue at
with the compiler's register allocation (query word k and train word k in the same bank) against a conflict-free one?"""
import sys

probes = []


def probe(name, lines, regs):
    probes.append((name, lines, sorted(set(regs))))


def xor_case(name, s0, s1, d):
    lines, regs = [], []
    for i in range(8):
        a, b, c = s0(i), s1(i), d(i)
        lines.append("v_xor_b32 v%d, v%d, v%d" % (c, a, b))
        regs += [a, b, c]
    probe(name, lines, regs)


# sources in the same bank / different banks; destinations elsewhere
xor_case("xor_same_bank", lambda i: 8 + 4 * (i % 4), lambda i: 24 + 4 * (i % 4), lambda i: 40 + i)
xor_case("xor_diff_bank", lambda i: 8 + 4 * (i % 4), lambda i: 25 + 4 * (i % 4), lambda i: 40 + i)
xor_case("xor_diff2_bank", lambda i: 8 + 4 * (i % 4), lambda i: 26 + 4 * (i % 4), lambda i: 40 + i)
xor_case("xor_dst_eq_src_bank", lambda i: 8 + 4 * (i % 4), lambda i: 25 + 4 * (i % 4), lambda i: 40 + 4 * (i % 4))


def bcnt_case(name, s0, s1, d):
    lines, regs = [], []
    for i in range(8):
        a, b, c = s0(i), s1(i), d(i)
        lines.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (c, a, b))
        regs += [a, b, c]
    probe(name, lines, regs)


bcnt_case("bcnt_same_bank", lambda i: 8 + 4 * (i % 4), lambda i: 40 + 4 * i, lambda i: 40 + 4 * i)
bcnt_case("bcnt_diff_bank", lambda i: 8 + 4 * (i % 4), lambda i: 41 + 4 * i, lambda i: 41 + 4 * i)


def body(name, qbase, tbase_a, tbase_b, word_of, lds=0):
    """the matcher's word-major step: 2 rows x 4 queries, 8 chains; q[r][k] = v(qbase + 8 r + k), row words at tbase + word_of(k)"""
    lines, regs = [], []
    if lds == 1:    # as the kernel: the two rows are read from LDS (broadcast) at the top of the step, then waited for
        lines += ["ds_read_b128 v[%d:%d], v98" % (tbase_a, tbase_a + 3), "ds_read_b128 v[%d:%d], v98 offset:32" % (tbase_b, tbase_b + 3),
                  "ds_read_b128 v[%d:%d], v98 offset:16" % (tbase_a + 4, tbase_a + 7), "ds_read_b128 v[%d:%d], v98 offset:48" % (tbase_b + 4, tbase_b + 7),
                  "s_waitcnt lgkmcnt(0)"]
        regs += [98]
    if lds == 2:    # software-pipelined: wait for the rows requested during the previous step, compute on them; the next
        lines += ["s_waitcnt lgkmcnt(0)"]    # step's rows are requested at the END of this body (see below) into the other set
        regs += [98]
    acc = [100 + i for i in range(8)]      # accumulators
    tmp = [110 + i for i in range(8)]      # xor results
    zero = 99
    for k in range(8):
        for r in range(4):
            for row, tb in ((0, tbase_a), (1, tbase_b)):
                d = tmp[2 * r + row]
                a, b = qbase + 8 * r + k, tb + word_of(k)
                lines.append("v_xor_b32 v%d, v%d, v%d" % (d, a, b))
                regs += [d, a, b]
        for c in range(8):
            lines.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (acc[c], tmp[c], acc[c] if k else zero))
            regs += [acc[c], tmp[c], zero]
    best = [120 + i for i in range(8)]
    for r in range(4):
        ka, kb = acc[2 * r], acc[2 * r + 1]
        lines.append("v_lshl_or_b32 v%d, v%d, 20, s4" % (ka, ka))
        lines.append("v_lshl_or_b32 v%d, v%d, 20, s5" % (kb, kb))
        lines.append("v_med3_u32 v%d, v%d, v%d, v%d" % (tmp[r], best[2 * r], ka, kb))
        lines.append("v_min_u32 v%d, v%d, v%d" % (best[2 * r + 1], best[2 * r + 1], tmp[r]))
        lines.append("v_min3_u32 v%d, v%d, v%d, v%d" % (best[2 * r], best[2 * r], ka, kb))
        regs += [best[2 * r], best[2 * r + 1]]
    if lds == 2:
        # issue the loads for the "next" step right after the last use of the row registers is not expressible with one register
        # set; emulate the pipelined form: loads into a SECOND set (v80..v95) that nobody reads in this probe -- same LDS traffic,
        # same issue slots, no wait in front of the VALU work
        lines[1:1] = ["ds_read_b128 v[80:83], v98", "ds_read_b128 v[84:87], v98 offset:32", "ds_read_b128 v[88:91], v98 offset:16",
                      "ds_read_b128 v[92:95], v98 offset:48"]
        regs += list(range(80, 96))
    probe(name, lines, regs)


ident = lambda k: k
body("body_conflict", 2, 62, 70, ident)              # q[k] and t[k] in the same bank (what the compiler allocated in round 3/4)
body("body_free2", 2, 64, 72, ident)                 # train tuples two banks away
body("body_free1", 2, 63, 71, ident)                 # one bank away
body("body_rot2", 2, 62, 70, lambda k: (k & 4) | ((k + 2) & 3))  # same tuples, words rotated by two inside each quad
body("body_lds_wait", 2, 62, 70, ident, lds=1)       # + the four broadcast ds_read_b128 of the step, waited for at once
body("body_lds_piped", 2, 62, 70, ident, lds=2)      # + the same reads, issued behind the wait (nothing waits for them in this step)


out = ['// GENERATED by tools/gen_valu_probe4.py -- see there.  build: hipcc --offload-arch=gfx950 -O3 -o build/valu_probe4 tools/valu_probe4.hip',
       '#include <hip/hip_runtime.h>', '#include <stdint.h>', '#include <stdio.h>', '']
for name, lines, regs in probes:
    clob = ", ".join('"v%d"' % r for r in regs)
    init = "\\n".join("v_mov_b32 v%d, %d" % (r, 0 if r == 98 else (r * 2654435761) & 0x7FFFFFFF) for r in regs)
    asm = "\\n".join(lines)
    out.append("__global__ void %s(uint32_t* out, int iters) {" % name)
    out.append("  __shared__ uint32_t lds_buf[4096];")
    out.append("  lds_buf[threadIdx.x] = threadIdx.x; lds_buf[threadIdx.x + 256] = 7u * threadIdx.x; __syncthreads();")
    out.append('  asm volatile("%s" ::: %s);' % (init, clob))
    out.append('  asm volatile("s_mov_b32 s4, 3\\ns_mov_b32 s5, 4" ::: "s4", "s5");')
    out.append("  for (int i = 0; i < iters; ++i) {")
    out.append('    asm volatile("%s" ::: %s, "s4", "s5");' % (asm, clob))
    out.append("  }")
    out.append("  uint32_t r;")
    out.append('  asm volatile("v_mov_b32 %%0, v%d" : "=v"(r) :: %s);' % (regs[-1], clob))
    out.append("  out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds_buf[(threadIdx.x * 3) & 255];")
    out.append("}")
    out.append("static const int %s_n = %d;" % (name, sum(1 for ln in lines if ln.startswith("v_"))))
    out.append("")
out.append('''typedef void (*kfn)(uint32_t*, int);
static void run(const char* name, kfn fn, int ninstr, uint32_t* d) {
  const int iters = ninstr > 64 ? 400 : 6000;
  printf("%-20s", name);
  for (int wps : {1, 2, 4, 5, 6, 8}) {  // waves per SIMD
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    fn<<<grid, block>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    fn<<<grid, block>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  w%d %.3f", wps, ms * 1e6 / ((double)iters * ninstr * wps));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  printf("   ns per wave64 instruction per SIMD (%d instructions per iteration)\\n", ninstr);
}
#define RUN(n) run(#n, n, n##_n, d)
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * 4);''')
for name, _, _ in probes:
    out.append("  RUN(%s);" % name)
out.append("  (void)hipFree(d);\n  return 0;\n}")
open(sys.argv[1] if len(sys.argv) > 1 else "tools/valu_probe4.hip", "w").write("\n".join(out) + "\n")
