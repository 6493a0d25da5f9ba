// vs_match.hip -- brute-force Hamming 2-NN on 256-bit descriptors + Lowe ratio compaction (gfx950).
//
// Replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(k=2) and the ratio loop of FeatureMatcher.match_features
// (reference src/v2/frame.py:18,23,25-47).
//
// Kernel design (VALU-bound integer work, no MFMA):
//   * lane = query.  Every lane keeps kQPL query descriptors (8 dwords each) in VGPRs for the whole kernel.
//   * train descriptors are wave-uniform.  Default (TSTAGE): every wave stages 64 train rows at a time in its own 2 KB LDS
//     slice (one coalesced 32-byte load per lane, prefetched one batch ahead, wave-synchronous: no barrier) and reads each
//     row back as two broadcast ds_read_b128 -- the first half one step ahead of its use -- so both v_xor_b32 operands are
//     VGPRs (on gfx950 a VGPR-only v_xor issues at the 2-cycle wave64 rate, with an SGPR source at the 4-cycle rate:
//     tools/valu_probe2.hip).  The older form -- scalar loads (s_load_dwordx8), the row as the SGPR operand of v_xor_b32 --
//     remains as the !TSTAGE instantiation (sweeps and tests).
//   * distance = 8 x (v_xor_b32 + v_bcnt_u32_b32 with accumulate); a step is 2 rows x kQPL queries = 8 chains advanced word by
//     word (8 xors, then 8 popcount-accumulates, held in that order by scheduling barriers).
//   * top-2 bookkeeping on packed keys  key = dist << 20 | train index, two train rows per update:
//       second = v_min_u32(second, v_med3_u32(best, ka, kb));  best = v_min3_u32(best, ka, kb)
//     (the second smallest of {best, second, ka, kb} with best <= second is min(second, median(best, ka, kb))), so the
//     tie rule "lower train index first" falls out of the packing.  16 + 1 + 1.5 = 18.5 VALU ops per distance.
//   * the grid is (query tiles) x (train chunks), ~5 workgroups per compute unit at 10k x 10k, the train rows cut evenly.  Every
//     (tile, chunk) workgroup publishes one self-validating 8-byte partial per query (launch epoch | second key | best key, one
//     write-through store); the workgroup of a tile's last chunk polls the tile's words and folds them in chunk order and writes
//     the final (idx, dist) rows: ONE launch, no merge kernel, no ticket.
//   * the ratio test + ordered compaction is one workgroup using wave ballots and popcounts for the prefix.
// What the body can and cannot do (round 4, tools/valu_probe4.hip, tools/match_stamps.py): under this load the shader clock is
// 2.1 - 2.4 GHz depending on the box (bench.py measures it per run); on a 2.12 GHz box v_xor (VGPR, VGPR) issues in ~2.8 cycles,
// v_bcnt / v_lshl_or / v_min3 / v_med3 in ~4.1; operand banks do not matter; the 148 instructions of a step issue at 1.72 ns
// apiece with four or five waves per SIMD -- the kernel's 29 440 instructions per SIMD are 50.7 us of issue there, the launch
// takes 57 - 58 us (53 us on a 2.38 GHz box).  Waves of a SIMD are served oldest first: the workgroups of a compute unit finish
// one after the other (14, 25, 36, 47, 57 us), not together.
#include "vs_internal.h"

#include <dlfcn.h>

#include <algorithm>

#include <vector>

namespace {

constexpr int kWaves = 4;                  // waves per workgroup
constexpr int kQPL = 4;                    // queries per lane
constexpr int kTU = 4;                     // train rows per step (two pair-updates)
constexpr int kTileQ = 64 * kQPL;          // queries per workgroup (= threads per workgroup: the fold is thread = query)
constexpr int kIdxBits = 20;
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;
constexpr int kMaxChunk = 1 << kIdxBits;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
static_assert(kTileQ == 64 * kWaves, "the fold maps one thread to one query of the tile");

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// one train descriptor = one 32-byte scalar load (s_load_dwordx8); rows are 32-byte aligned
__device__ __forceinline__ u32x8 load_train(const uint32_t* p) { return *reinterpret_cast<const u32x8*>(p); }

__device__ __forceinline__ uint32_t dist_key(const uint32_t (&q)[8], const u32x8& tw, uint32_t j) {
  uint32_t acc = bcnt_acc(q[0] ^ tw[0], 0u);
#pragma unroll
  for (int k = 1; k < 8; ++k) acc = bcnt_acc(q[k] ^ tw[k], acc);
  return (acc << kIdxBits) | j;
}

// two train rows (indices j, j + 1 within the chunk) against the lane's kQPL queries
__device__ __forceinline__ void accumulate2(const uint32_t (&qv)[kQPL][8], const u32x8& ta, const u32x8& tb, uint32_t j,
                                            uint32_t (&b1)[kQPL], uint32_t (&b2)[kQPL]) {
#pragma unroll
  for (int r = 0; r < kQPL; ++r) {
    const uint32_t ka = dist_key(qv[r], ta, j), kb = dist_key(qv[r], tb, j + 1);
    b2[r] = min(b2[r], umed3(b1[r], ka, kb));
    b1[r] = umin3(b1[r], ka, kb);
  }
}
__device__ __forceinline__ void accumulate1(const uint32_t (&qv)[kQPL][8], const u32x8& tw, uint32_t j,
                                            uint32_t (&b1)[kQPL], uint32_t (&b2)[kQPL]) {
#pragma unroll
  for (int r = 0; r < kQPL; ++r) {
    const uint32_t key = dist_key(qv[r], tw, j);
    b2[r] = umed3(b1[r], b2[r], key);
    b1[r] = min(b1[r], key);
  }
}

// (best, second) of two sets whose keys are all distinct
__device__ __forceinline__ void fold64(unsigned long long& B1, unsigned long long& B2, unsigned long long K) {
  B2 = min(B2, max(B1, K));
  B1 = min(B1, K);
}

// Workgroup (tile, chunk): kWaves waves hold the SAME kTileQ queries; wave w scans the w-th quarter of the chunk and the
// four (best, second) pairs are merged through LDS, so one 8-byte partial per (chunk, query) is published:
//   partial[(chunk * qtiles + tile) * kTileQ + local query]      (2 KB rows: no cache line is shared by two workgroups)
// Chunks and quarters are cut evenly from the ACTUAL number of train rows (chunk c = rows [c nt / C, (c + 1) nt / C)): every
// wave of the launch scans the same number of rows to within one (with chunk lengths rounded up to a multiple of 16 the last
// chunk of a 10 000-row set was 80 rows against 320, and everybody else carried 2.4 % more than their share).
//
// Hand-off of the partials (round 4).  A partial is ONE self-validating 64-bit word
//   [63:58] epoch of the launch (1 .. 63, from the host)   [57:29] second key   [28:0] best key     (key = dist << 20 | index)
// stored write-through by one instruction.  The workgroup of the tile's LAST chunk -- dispatched last, so every other chunk
// of the tile has been dispatched before it: no wait on a workgroup that cannot start -- folds the tile: thread = query polls
// the tile's words until every one carries this launch's epoch.  One store and one (polled) load lie between the end of the
// last body and the fold; the form before (store, drain, barrier, agent-scope ticket, barrier, acquire, loads) had three
// dependent round trips to the coherence point there, and only then did the last arriver start loading.
// A word of an earlier launch with the same geometry carries the previous epoch (every slot is rewritten by every launch);
// when the geometry of a stream's launches changes, or after 63 launches... the host clears the slots (knn2_dev_impl).
// PACKED: one 16-byte row (idx0, idx1, dist0, dist1) per query -- the layout the query-sharded matcher all-gathers.
// TSTAGE (how the wave-uniform train rows reach the VALU):
//   false: scalar loads (s_load_dwordx16), the row is the SGPR operand of v_xor_b32 (sweeps only; no private segment either:
//          tests/test_kernel_resources.py holds both instantiations at scratch 0);
//   true : every wave stages 64 rows at a time in its own LDS slice (one coalesced 32-B load per lane, prefetched one
//          batch ahead) and reads each row back as two broadcast ds_read_b128 -- both v_xor_b32 operands are then VGPRs.
//          tools/valu_probe2.hip: v_xor/v_and/v_or with VGPR sources issue at the 2-cycle wave64 rate on gfx950, with an
//          SGPR source (and every v_bcnt / v_min3 / v_med3) at the 4-cycle rate.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32x8 row_from(const uint4 a, const uint4 b) {
  u32x8 r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
  r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  return r;
}

constexpr uint32_t kKey29 = (1u << 29) - 1u;  // a partial's keys: 9 bits of distance (0 .. 256), 20 bits of index; all ones = none
constexpr int kEpochs = 63;                   // epochs 1 .. 63 (0: a cleared slot)

// distances of 2 rows x kQPL queries as 2 * kQPL chains advanced word by word (all xors of a word, then all popcount-accumulates
// of that word: no instruction depends on the one issued just before it in its wave); keys out
__device__ __forceinline__ void keys2(const uint32_t (&qv)[kQPL][8], const u32x8& ta, const u32x8& tb, uint32_t j,
                                      uint32_t (&ka)[kQPL], uint32_t (&kb)[kQPL]) {
  uint32_t a[kQPL], b[kQPL];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t xa[kQPL], xb[kQPL];
#pragma unroll
    for (int r = 0; r < kQPL; ++r) {
      xa[r] = qv[r][k] ^ ta[k];
      xb[r] = qv[r][k] ^ tb[k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kQPL; ++r) {
      a[r] = bcnt_acc(xa[r], k ? a[r] : 0u);
      b[r] = bcnt_acc(xb[r], k ? b[r] : 0u);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < kQPL; ++r) {
    ka[r] = (a[r] << kIdxBits) | j;
    kb[r] = (b[r] << kIdxBits) | (j + 1);
  }
}
// (best, second) of a query updated with the keys of two rows
__device__ __forceinline__ void update2(const uint32_t (&ka)[kQPL], const uint32_t (&kb)[kQPL], uint32_t (&b1)[kQPL],
                                        uint32_t (&b2)[kQPL]) {
#pragma unroll
  for (int r = 0; r < kQPL; ++r) {
    b2[r] = min(b2[r], umed3(b1[r], ka[r], kb[r]));
    b1[r] = umin3(b1[r], ka[r], kb[r]);
  }
}
__device__ __forceinline__ void step2(const uint32_t (&qv)[kQPL][8], const u32x8& ta, const u32x8& tb, uint32_t j,
                                      uint32_t (&b1)[kQPL], uint32_t (&b2)[kQPL]) {
  uint32_t ka[kQPL], kb[kQPL];
  keys2(qv, ta, tb, j, ka, kb);
  update2(ka, kb, b1, b2);
}

template <bool TSTAGE>
__global__ __launch_bounds__(64 * kWaves, TSTAGE ? 4 : 5) void hamming_knn2_kernel(const uint4* __restrict__ q4, int nq,
                                                                     const uint32_t* __restrict__ t, int nt, int packed,
                                                                     unsigned long long* partial, unsigned epoch,
                                                                     unsigned* host_flag, int2* __restrict__ idx,
                                                                     int2* __restrict__ dist, const int* __restrict__ nt_dev,
                                                                     unsigned long long* __restrict__ stamps) {
  __shared__ uint2 lds[kWaves][kTileQ];
  // diagnostic (vs_match_stamps): wall-clock stamps of this workgroup's phases, thread 0 only
  unsigned long long* st = stamps ? stamps + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
  long long cyc0 = 0;
  if (st && threadIdx.x == 0) {
    cyc0 = (long long)__builtin_readcyclecounter();
    st[0] = wall_clock64();
    st[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |   // HW_ID
            ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);  // XCC_ID
  }
  // keys carry the train index in their low kIdxBits bits: the global row index when all of them fit (the PLANNED nt <= 2^20
  // -- keys of different chunks are then comparable as they are and the fold below is 32-bit min3 / med3), else the index within
  // the workgroup's chunk (the fold rebuilds 64-bit keys from the chunk's first row)
  const bool gkey = nt <= kMaxChunk;
  // nt_dev: the number of train rows lives on the device (a detector's key-point count the host has not read); the launch
  // was planned for nt rows at most, with fewer rows every chunk is shorter (or empty) and still publishes its partial
  if (nt_dev) nt = min(nt, *nt_dev);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> SGPR
  const int tile = blockIdx.x, chunk = blockIdx.y, qtiles = gridDim.x, nchunks = gridDim.y;
  const int qbase = tile * kTileQ;
  const int cbegin = (int)((long long)chunk * nt / nchunks);
  const int cend = (int)((long long)(chunk + 1) * nt / nchunks);
  const int clen = cend - cbegin;
  const int begin = cbegin + (int)((long long)wave * clen / kWaves);
  const int end = cbegin + (int)((long long)(wave + 1) * clen / kWaves);

  uint32_t qv[kQPL][8];
#pragma unroll
  for (int r = 0; r < kQPL; ++r) {
    int qi = min(qbase + r * 64 + lane, nq - 1);  // clamp: the tail computes a duplicate, the store is guarded
    uint4 a = q4[2 * (size_t)qi], b = q4[2 * (size_t)qi + 1];
    qv[r][0] = a.x; qv[r][1] = a.y; qv[r][2] = a.z; qv[r][3] = a.w;
    qv[r][4] = b.x; qv[r][5] = b.y; qv[r][6] = b.z; qv[r][7] = b.w;
  }
  uint32_t b1[kQPL], b2[kQPL];
#pragma unroll
  for (int r = 0; r < kQPL; ++r) b1[r] = b2[r] = kEmpty;

  const uint32_t* tp = t + (size_t)begin * 8;  // wave-uniform -> scalar loads
  const uint32_t j0 = (uint32_t)(gkey ? begin : begin - cbegin);
  const int n = end - begin;
  if constexpr (TSTAGE) {
    __shared__ uint4 stage[kWaves][2][128];  // per wave: two batches of 64 rows x 32 B
    const uint4* t4 = reinterpret_cast<const uint4*>(t);
    const int nb = (n + 63) >> 6;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (nb > 0) {
      const int row = min(begin + lane, nt - 1);
      r0 = t4[2 * (size_t)row];
      r1 = t4[2 * (size_t)row + 1];
    }
    for (int b = 0; b < nb; ++b) {
      uint4* buf = stage[wave][b & 1];
      buf[2 * lane] = r0;
      buf[2 * lane + 1] = r1;
      if (b + 1 < nb) {  // prefetch the next batch while this one is consumed
        const int row = min(begin + 64 * (b + 1) + lane, nt - 1);
        r0 = t4[2 * (size_t)row];
        r1 = t4[2 * (size_t)row + 1];
      }
      wave_lds_order();
      if (st && threadIdx.x == 0 && b == 0) st[1] = wall_clock64();  // queries and the first train batch have arrived
      const int cnt = min(64, n - 64 * b);
      const uint32_t jb = j0 + (uint32_t)(64 * b);
      int j = 0;
      // The FIRST halves (words 0..3) of the rows of step s + 1 are requested from LDS before the arithmetic of step s (two
      // register sets in turn), the second halves at the top of their own step -- they are needed only four word groups later.
      // No instruction of a step waits for a read issued just before it.  (Waves of a SIMD are served oldest first, so a wave
      // that waits for its own reads at the top of every step hands the SIMD over and takes it back 39 times per scan:
      // 57 -> 54.5 us per launch at 10k x 10k, 4.74 -> 4.53 ms at 100k x 100k, although the kernel now needs 108 VGPRs and
      // holds four waves per SIMD instead of five.)
      uint4 a0 = buf[0], c0 = buf[2];
      for (; j + 4 <= cnt; j += 4) {
        const uint4 a0h = buf[2 * j + 1], c0h = buf[2 * j + 3];
        const uint4 a1 = buf[2 * j + 4], c1 = buf[2 * j + 6];
        step2(qv, row_from(a0, a0h), row_from(c0, c0h), jb + (uint32_t)j, b1, b2);
        const uint4 a1h = buf[2 * j + 5], c1h = buf[2 * j + 7];
        const int jn = min(j + 4, 62);  // (the batch holds 64 rows; beyond a full batch's last step nothing uses these)
        a0 = buf[2 * jn];
        c0 = buf[2 * jn + 2];
        step2(qv, row_from(a1, a1h), row_from(c1, c1h), jb + (uint32_t)(j + 2), b1, b2);
      }
      if (j + 2 <= cnt) {
        step2(qv, row_from(a0, buf[2 * j + 1]), row_from(c0, buf[2 * j + 3]), jb + (uint32_t)j, b1, b2);
        j += 2;
        if (j < cnt) accumulate1(qv, row_from(buf[2 * j], buf[2 * j + 1]), jb + (uint32_t)j, b1, b2);
      } else if (j < cnt) {
        accumulate1(qv, row_from(a0, buf[2 * j + 1]), jb + (uint32_t)j, b1, b2);
      }
      wave_lds_order();
    }
  } else {
    int j = 0;
    for (; j + kTU <= n; j += kTU) {
      u32x8 tw[kTU];
#pragma unroll
      for (int u = 0; u < kTU; ++u) tw[u] = load_train(tp + (size_t)(j + u) * 8);
#pragma unroll
      for (int u = 0; u < kTU; u += 2) accumulate2(qv, tw[u], tw[u + 1], j0 + (uint32_t)(j + u), b1, b2);
    }
    for (; j < n; ++j) {
      u32x8 tw = load_train(tp + (size_t)j * 8);
      accumulate1(qv, tw, j0 + (uint32_t)j, b1, b2);
    }
  }
  if (st && threadIdx.x == 0) {  // wave 0's scan is through
    st[2] = wall_clock64();
    st[chunk != nchunks - 1 ? 5 : 4] = (1ull << 63) | (unsigned long long)((long long)__builtin_readcyclecounter() - cyc0);  // (the slot this role leaves free)
  }
#pragma unroll
  for (int r = 0; r < kQPL; ++r) lds[wave][r * 64 + lane] = make_uint2(b1[r], b2[r]);
  __syncthreads();
  if (st && threadIdx.x == 0) st[3] = wall_clock64();  // all four waves' scans are through
  // keys of different waves are distinct (disjoint index ranges), so min / max merge them exactly; thread = query
  const int lq = threadIdx.x, qi = qbase + lq;
  uint2 m = lds[0][lq];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) {
    const uint2 o = lds[w][lq];
    const uint32_t s2 = min(max(m.x, o.x), min(m.y, o.y));
    m.x = min(m.x, o.x);
    m.y = s2;
  }
  unsigned long long B1 = ~0ull, B2 = ~0ull;
  if (nchunks > 1) {
    const unsigned long long tag = (unsigned long long)epoch << 58;
    if (chunk != nchunks - 1) {
      // ---- publish and leave: one write-through store (agent-scope relaxed atomic store = global_store_dwordx2 ... sc1)
      __hip_atomic_store(partial + ((size_t)chunk * qtiles + tile) * kTileQ + lq,
                         tag | ((unsigned long long)(m.y & kKey29) << 29) | (unsigned long long)(m.x & kKey29), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (st && threadIdx.x == 0) st[4] = wall_clock64();
      return;
    }
    // ---- the tile's last chunk folds: its own pair first, then the other chunks' words in chunk order, as they arrive
    constexpr int kFoldBatch = 16;
    uint32_t g1 = kKey29, g2 = kKey29;
    bool lost = false;
    for (int c0 = 0; c0 < nchunks - 1; c0 += kFoldBatch) {
      unsigned long long w[kFoldBatch];
      for (int spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) {
          const int c = min(c0 + u, nchunks - 2);
          w[u] = __hip_atomic_load(partial + ((size_t)c * qtiles + tile) * kTileQ + lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) ok &= (w[u] >> 58) == (unsigned long long)epoch;
        if (ok) break;
        if (spin > (1 << 18)) {  // bounded (~a second): a chunk's workgroup never ran.  Reported through the pinned flag.
          lost = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (lost) break;
      if (gkey) {  // uniform
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) {
          const bool in = c0 + u < nchunks - 1;  // beyond the last chunk the (clamped) load repeated a word: neutralise it
          const uint32_t kx = in ? (uint32_t)(w[u] & kKey29) : kKey29, ky = in ? (uint32_t)((w[u] >> 29) & kKey29) : kKey29;
          g2 = min(g2, umed3(g1, kx, ky));
          g1 = umin3(g1, kx, ky);
        }
      } else {
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) {
          if (c0 + u >= nchunks - 1) break;
          const unsigned long long base = (unsigned long long)((long long)(c0 + u) * nt / nchunks);
          const uint32_t kx = (uint32_t)(w[u] & kKey29), ky = (uint32_t)((w[u] >> 29) & kKey29);
          if (kx != kKey29) fold64(B1, B2, ((unsigned long long)(kx >> kIdxBits) << 32) | (base + (kx & kIdxMask)));
          if (ky != kKey29) fold64(B1, B2, ((unsigned long long)(ky >> kIdxBits) << 32) | (base + (ky & kIdxMask)));
        }
      }
    }
    if (st && threadIdx.x == 0) st[5] = wall_clock64();  // every chunk's word of thread 0's query has arrived
    if (lost) {
      if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      g1 = g2 = kKey29;
      B1 = B2 = ~0ull;
      m.x = m.y = kEmpty;
    }
    if (gkey) {
      const uint32_t mx = m.x & kKey29, my = m.y & kKey29;
      g2 = min(g2, umed3(g1, mx, my));
      g1 = umin3(g1, mx, my);
      if (g1 != kKey29) B1 = ((unsigned long long)(g1 >> kIdxBits) << 32) | (unsigned long long)(g1 & kIdxMask);
      if (g2 != kKey29) B2 = ((unsigned long long)(g2 >> kIdxBits) << 32) | (unsigned long long)(g2 & kIdxMask);
    } else {
      const unsigned long long base = (unsigned long long)cbegin;
      if (m.x != kEmpty) fold64(B1, B2, ((unsigned long long)(m.x >> kIdxBits) << 32) | (base + (m.x & kIdxMask)));
      if (m.y != kEmpty) fold64(B1, B2, ((unsigned long long)(m.y >> kIdxBits) << 32) | (base + (m.y & kIdxMask)));
    }
  } else {
    if (m.x != kEmpty) fold64(B1, B2, ((unsigned long long)(m.x >> kIdxBits) << 32) | (unsigned long long)(m.x & kIdxMask));
    if (m.y != kEmpty) fold64(B1, B2, ((unsigned long long)(m.y >> kIdxBits) << 32) | (unsigned long long)(m.y & kIdxMask));
  }
  if (qi < nq) {
    if (packed) {
      reinterpret_cast<int4*>(idx)[qi] =
          make_int4((int)(B1 & 0xFFFFFFFFull), (int)(B2 & 0xFFFFFFFFull), (int)(B1 >> 32), (int)(B2 >> 32));
    } else {
      idx[qi] = make_int2((int)(B1 & 0xFFFFFFFFull), (int)(B2 & 0xFFFFFFFFull));
      dist[qi] = make_int2((int)(B1 >> 32), (int)(B2 >> 32));
    }
  }
  if (st && threadIdx.x == 0) st[6] = wall_clock64();
}

// Lowe ratio test + ordered compaction: one workgroup, wave ballots + popcounts give the ranks.
__global__ __launch_bounds__(1024) void ratio_compact_kernel(const int2* __restrict__ idx, const int2* __restrict__ dist,
                                                              int nq, double ratio, int32_t* __restrict__ mq,
                                                              int32_t* __restrict__ mt, int32_t* __restrict__ md,
                                                              int32_t* __restrict__ n_out, const int* __restrict__ nt_dev,
                                                              int32_t* __restrict__ h_mirror, int h_stride) {
  // h_mirror != nullptr: pinned host block [count (4 ints) | mq (h_stride) | mt (h_stride) | md (h_stride)] that receives
  // every result word as well (the tracking period fed by the class API reads matches without a copy launch)
  __shared__ int wave_cnt[16];
  if (nt_dev && *nt_dev < 2) {  // fewer than two train rows: no second neighbour, no match (frame.py:30 could not unpack (m, n))
    if (threadIdx.x == 0) {
      *n_out = 0;
      if (h_mirror) h_mirror[0] = 0;
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int running = 0;
  for (int start = 0; start < nq; start += 1024) {
    const int q = start + threadIdx.x;
    int2 i2 = make_int2(0, 0), d2 = make_int2(0, 0);
    bool pass = false;
    if (q < nq) {
      i2 = idx[q];
      d2 = dist[q];
      pass = (double)d2.x < ratio * (double)d2.y;  // m.distance < ratio * n.distance (frame.py:33)
    }
    const unsigned long long bal = __ballot(pass);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      int c = wave_cnt[w];
      before += (w < wave) ? c : 0;
      total += c;
    }
    if (pass) {
      const int pos = running + before + rank;
      mq[pos] = q;
      mt[pos] = i2.x;
      md[pos] = d2.x;
      if (h_mirror) {
        h_mirror[4 + pos] = q;
        h_mirror[4 + h_stride + pos] = i2.x;
        h_mirror[4 + 2 * h_stride + pos] = d2.x;
      }
    }
    running += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *n_out = running;
    if (h_mirror) h_mirror[0] = running;
  }
}

// nchunks workgroups along y share the train rows evenly (the kernel cuts chunk c = rows [c nt / C, (c + 1) nt / C) itself).
// Automatic plan (target_blocks == 0; > 0 is the sweep hook vs_tune_match): pick the number of train chunks that minimises
//     ceil(workgroups / 256 CUs) * chunk length  +  fold cost per chunk
// over plans with 1000..4600 workgroups -- the first term is the busiest CU's share of train rows (workgroups are
// spread round-robin, 4..16 resident per CU), the second the extra partial words the folding workgroup of a tile reads.  At
// 10k x 10k this gives 32 chunks x 40 query tiles = 1280 workgroups = 5 per CU, at 100k x 100k 11 x 391 = 4301 (98.8 %
// balanced; a fixed 1024-workgroup plan loses 25 % there to 4-vs-3 workgroups per CU).
void plan_chunks(int target_blocks, int nq, int nt, int* nchunks) {
  const int qtiles = (nq + kTileQ - 1) / kTileQ;
  const int q1 = qtiles > 0 ? qtiles : 1;
  long nch = 1;
  if (target_blocks > 0) {
    nch = target_blocks / q1;
  } else {
    const long lo = (1000 + q1 - 1) / q1, hi = 4600 / q1 > lo ? 4600 / q1 : lo;
    double best = 1e300;
    for (long c = lo; c <= hi; ++c) {
      const long len = (nt + c - 1) / c;
      if (len < 32 && c > lo) break;
      const double cost = (double)((q1 * c + 255) / 256) * (double)len + (double)c * 4.6 * (double)q1 / 40.0;
      if (cost < best) {
        best = cost;
        nch = c;
      }
    }
  }
  // at least 32 rows per chunk (8 per wave): below that a workgroup is all prologue and hand-off -- the 600 x 3000 match of a
  // tracked frame takes 94 chunks (10 us); 334 chunks of 9 rows, which the cost model alone would pick, took 27 us
  const long most = std::max(1L, (long)nt / 32);
  if (nch > most) nch = most;
  const long need = ((long)nt + kMaxChunk - 1) / kMaxChunk;  // a chunk's rows are indexed with kIdxBits bits
  if (nch < need) nch = need;
  if (nch > 65535) nch = 65535;  // gridDim.y
  if (nch < 1) nch = 1;
  *nchunks = (int)nch;
}

int check_args(vs_ctx* ctx, const void* q, int nq, const void* t, int nt, const char* fn) {
  if (!ctx) return VS_EINVAL;
  if (nq < 0 || nt < 2) return vs_fail(ctx, VS_EINVAL, "%s: need nq >= 0 and nt >= 2 (k = 2 neighbours)", fn);
  if ((nq > 0 && !q) || !t) return vs_fail(ctx, VS_EINVAL, "%s: null descriptor pointer", fn);
  return VS_OK;
}

}  // namespace

// tuning hooks for bench sweeps and tests (not part of the stable ABI; per context): a negative value leaves a knob as it is
VS_API int vs_tune_match(vs_ctx* ctx, int target_blocks, int tstage) {
  if (!ctx) return VS_EINVAL;
  if (target_blocks >= 0) ctx->tune.match_target_blocks = target_blocks;
  if (tstage == 0 || tstage == 1) ctx->tune.match_tstage = tstage, ctx->chain_scratch = -1;  // (another kernel: ask again)
  return VS_OK;
}

// scratch of the stream `s` (launches on different streams may run concurrently and must not share it)
static int match_scratch_for(vs_ctx* ctx, hipStream_t s, vs_match_scratch** out) {
  vs_match_scratch* ms = nullptr;
  for (vs_match_scratch& m : ctx->match_scratch)
    if (m.used && m.stream == s) ms = &m;
  if (!ms) {
    // a stream not seen before takes a free set, else the least recently used one -- once the device has drained: the set's
    // old stream may have been destroyed by its owner meanwhile, so its handle is never touched again
    vs_match_scratch* lru = &ctx->match_scratch[0];
    for (vs_match_scratch& m : ctx->match_scratch) {
      if (!m.used) {
        lru = &m;
        break;
      }
      if (m.stamp < lru->stamp) lru = &m;
    }
    if (lru->used) VS_HIP(ctx, hipDeviceSynchronize());
    lru->used = true;
    lru->stream = s;
    ms = lru;
  }
  ms->stamp = ++ctx->desc_stamp;
  *out = ms;
  return VS_OK;
}

// after a synchronisation of stream `s`: did a folding workgroup of a launch on it give up its bounded wait?  (the device entry
// points cannot know -- they only enqueue -- and find out on the stream's next launch; the host entry points ask right away)
static int match_check_flag(vs_ctx* ctx, hipStream_t s, const char* who) {
  for (vs_match_scratch& m : ctx->match_scratch)
    if (m.used && m.stream == s && m.flag.p && *(volatile unsigned*)m.flag.p != 0u) {
      *(volatile unsigned*)m.flag.p = 0u;
      m.qtiles = m.nchunks = 0;  // the slots are cleared before the next launch
      return vs_fail(ctx, VS_EHIP, "%s: a train chunk of the match launch did not report within the bounded wait", who);
    }
  return VS_OK;
}

// the same question over EVERY scratch set (any stream): asked where results of device-side launches are handed out -- the
// tracking period's back half, vs_track_end, vs_match_status (the sharded matcher's collect), vs_destroy -- so that a launch
// that lost a chunk is reported against itself, not against the next, innocent launch on its stream (or never)
int vs_match_lost_check(vs_ctx* ctx, const char* who) {
  bool lost = false;
  for (vs_match_scratch& m : ctx->match_scratch)
    if (m.used && m.flag.p && *(volatile unsigned*)m.flag.p != 0u) {
      *(volatile unsigned*)m.flag.p = 0u;
      m.qtiles = m.nchunks = 0;
      lost = true;
    }
  if (lost)
    return vs_fail(ctx, VS_EHIP, "%s: a train chunk of a match launch did not report within the bounded wait (its rows are -1)", who);
  return VS_OK;
}

VS_API int vs_match_status(vs_ctx* ctx) { return ctx ? vs_match_lost_check(ctx, "vs_match_status") : VS_EINVAL; }

// developer hook (include/vslam_hip_dev.h): raise the pinned "lost a chunk" word of every scratch set in use, exactly as a
// folding workgroup that gave up would -- tests of where the report surfaces; returns the number of words raised
VS_API int vs_match_debug_raise(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  int n = 0;
  for (vs_match_scratch& m : ctx->match_scratch)
    if (m.used && m.flag.p) {
      *(volatile unsigned*)m.flag.p = 1u;
      ++n;
    }
  return n;
}

static int knn2_dev_impl(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_idx, void* d_dist,
                         bool packed, void* stream, const int* nt_dev = nullptr) {
  VS_TRY(check_args(ctx, d_q, nq, d_t, nt, "vs_hamming_knn2_dev"));
  if (nq == 0) return VS_OK;
  if (!d_idx || (!packed && !d_dist)) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_hamming_knn2_dev");
  hipStream_t s = vs_pick_stream(ctx, stream);
  int nchunks;
  plan_chunks(ctx->tune.match_target_blocks, nq, nt, &nchunks);
  if (((uintptr_t)d_q | (uintptr_t)d_t) & 31)
    return vs_fail(ctx, VS_EINVAL, "%s: descriptor arrays must be 32-byte aligned", "vs_hamming_knn2_dev");
  const int qtiles = (nq + kTileQ - 1) / kTileQ;
  vs_match_scratch* ms = nullptr;
  VS_TRY(match_scratch_for(ctx, s, &ms));
  if (!ms->flag.p) {  // one pinned word per scratch set: a folding workgroup that gave up waiting says so here
    VS_TRY(vs_reserve_pinned(ctx, &ms->flag, 64));
    *(volatile unsigned*)ms->flag.p = 0u;
  }
  if (*(volatile unsigned*)ms->flag.p != 0u) {
    *(volatile unsigned*)ms->flag.p = 0u;
    ms->qtiles = ms->nchunks = 0;
    return vs_fail(ctx, VS_EHIP, "%s: an earlier match launch on this stream did not see all its train chunks report", "vs_hamming_knn2_dev");
  }
  // the partial words validate themselves by the launch's epoch (1 .. 63, see the kernel).  Every launch of one geometry
  // rewrites every slot, so a slot never carries the current epoch before its workgroup stores it -- unless the slots were
  // laid out for another geometry, or the buffer is new: then they are cleared first (stream-ordered, in front of the launch)
  const size_t pbytes = sizeof(unsigned long long) * (size_t)nchunks * qtiles * kTileQ;
  const bool fresh = pbytes > ms->partial.cap || !ms->partial.p;
  VS_TRY(vs_reserve(ctx, &ms->partial, pbytes));
  if (fresh || ms->qtiles != qtiles || ms->nchunks != nchunks) {
    if (nchunks > 1) VS_HIP(ctx, hipMemsetAsync(ms->partial.p, 0, pbytes, s));
    ms->qtiles = qtiles;
    ms->nchunks = nchunks;
    ms->epoch = 0;
  }
  ms->epoch = ms->epoch % kEpochs + 1;
  dim3 grid(qtiles, nchunks);
  vs_prof_rec pr{};
  if (ctx->tune.match_profile) {
    VS_HIP(ctx, hipEventCreate(&pr.e0));
    VS_HIP(ctx, hipEventCreate(&pr.e1));
    VS_HIP(ctx, hipEventRecord(pr.e0, s));
  }
  typedef void (*knn2_fn)(const uint4*, int, const uint32_t*, int, int, unsigned long long*, unsigned, unsigned*, int2*, int2*, const int*,
                          unsigned long long*);
  static const knn2_fn kFn[2] = {hamming_knn2_kernel<false>, hamming_knn2_kernel<true>};
  const int which = ctx->tune.match_tstage ? 1 : 0;
  unsigned long long* d_stamps = nullptr;
  if (ctx->match_stamps_on) {  // diagnostic: per-workgroup phase stamps of THIS launch (vs_match_stamps)
    const size_t sb = sizeof(unsigned long long) * 8 * (size_t)qtiles * nchunks;
    VS_TRY(vs_reserve(ctx, &ctx->d_match_stamps, sb));
    VS_HIP(ctx, hipMemsetAsync(ctx->d_match_stamps.p, 0, sb, s));
    ctx->match_stamps_rows = qtiles * nchunks;
    d_stamps = (unsigned long long*)ctx->d_match_stamps.p;
  }
  hipLaunchKernelGGL(kFn[which], grid, dim3(64 * kWaves), 0, s, (const uint4*)d_q, nq, (const uint32_t*)d_t, nt, packed ? 1 : 0,
                     (unsigned long long*)ms->partial.p, (unsigned)ms->epoch, (unsigned*)ms->flag.p, (int2*)d_idx,
                     packed ? (int2*)nullptr : (int2*)d_dist, nt_dev, d_stamps);
  VS_LAUNCH_CHECK(ctx, "hamming_knn2_kernel");
  if (ctx->tune.match_profile) {
    VS_HIP(ctx, hipEventRecord(pr.e1, s));
    ctx->match_prof.push_back(pr);
  }
  return VS_OK;
}

VS_API int vs_hamming_knn2_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_idx,
                               void* d_dist, void* stream) {
  return knn2_dev_impl(ctx, d_q, nq, d_t, nt, d_idx, d_dist, false, stream);
}

VS_API int vs_hamming_knn2_packed_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_out,
                                      void* stream) {
  if (ctx && ((uintptr_t)d_out & 15)) return vs_fail(ctx, VS_EINVAL, "%s: output must be 16-byte aligned", "vs_hamming_knn2_packed_dev");
  return knn2_dev_impl(ctx, d_q, nq, d_t, nt, d_out, nullptr, true, stream);
}

// ---- query-sharded step: local match into this rank's slot of the gather buffer + one in-place ncclAllGather
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
nccl_allgather_fn g_nccl_allgather = nullptr;
bool g_nccl_tried = false;

nccl_allgather_fn resolve_nccl() {
  if (g_nccl_tried) return g_nccl_allgather;
  g_nccl_tried = true;
  // the RCCL that the process already uses (torch.distributed loads one): never a second copy
  for (const char* name : {"librccl.so", "librccl.so.1"}) {
    void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) continue;
    if (void* f = dlsym(h, "ncclAllGather")) {
      g_nccl_allgather = (nccl_allgather_fn)f;
      return g_nccl_allgather;
    }
  }
  if (void* f = dlsym(RTLD_DEFAULT, "ncclAllGather")) {
    g_nccl_allgather = (nccl_allgather_fn)f;
    return g_nccl_allgather;
  }
  if (void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL))
    g_nccl_allgather = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
  return g_nccl_allgather;
}
}  // namespace

VS_API int vs_hamming_knn2_sharded_dev(vs_ctx* ctx, const void* d_q_shard, int nq_shard, const void* d_t, int nt,
                                       void* d_gathered, int per, int rank, int world, void* nccl_comm,
                                       void* compute_stream, void* comm_stream, void* done_event, void* after_stream) {
  if (!ctx) return VS_EINVAL;
  if (world < 1 || rank < 0 || rank >= world || per < 0 || nq_shard < 0 || nq_shard > per || !d_gathered ||
      ((uintptr_t)d_gathered & 15))
    return vs_fail(ctx, VS_EINVAL, "%s: bad shard geometry / gather buffer", "vs_hamming_knn2_sharded_dev");
  if (world > 1 && (!nccl_comm || !comm_stream))
    return vs_fail(ctx, VS_EINVAL, "%s: world > 1 needs a communicator and a stream for it", "vs_hamming_knn2_sharded_dev");
  hipStream_t cs = vs_pick_stream(ctx, compute_stream);
  // ---- everything that may still touch d_gathered, or that produces this step's inputs, comes first on the compute stream:
  //  (1) done_event as recorded by the PREVIOUS step on this gather buffer -- its all-gather sends from, and receives into,
  //      the rows the kernel is about to overwrite (an event that was never recorded counts as complete);
  //  (2) the caller's stream up to now: the consumers of the previous results of d_gathered and the producers of q / t.
  if (done_event) {
    // (a wait for an event that has already completed is still a barrier packet in front of the kernel -- with the record behind
    // it, 17 us per step on a single compute stream; round 5, tools/shard_modes.py -- so it is only enqueued when it can matter)
    const hipError_t qe = hipEventQuery((hipEvent_t)done_event);
    if (qe == hipErrorNotReady) {
      (void)hipGetLastError();
      VS_HIP(ctx, hipStreamWaitEvent(cs, (hipEvent_t)done_event, 0));
    } else if (qe != hipSuccess) {
      return vs_fail(ctx, VS_EHIP, "%s: hipEventQuery(done_event) failed: %s", "vs_hamming_knn2_sharded_dev", hipGetErrorString(qe));
    }
  }
  if (after_stream && (hipStream_t)after_stream != cs) {
    if (!ctx->ev_after) VS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_after, hipEventDisableTiming));
    VS_HIP(ctx, hipEventRecord(ctx->ev_after, (hipStream_t)after_stream));
    VS_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_after, 0));
  }
  uint8_t* slot = (uint8_t*)d_gathered + (size_t)rank * per * 16;
  VS_TRY(knn2_dev_impl(ctx, d_q_shard, nq_shard, d_t, nt, slot, nullptr, true, cs));
  if (!nccl_comm) {  // no collective: the step is complete when the kernel is
    if (done_event) VS_HIP(ctx, hipEventRecord((hipEvent_t)done_event, cs));
    return VS_OK;
  }
  nccl_allgather_fn ag = resolve_nccl();
  if (!ag) return vs_fail(ctx, VS_ENCCL, "%s: ncclAllGather not found (is RCCL loaded in this process?)", "vs_hamming_knn2_sharded_dev");
  hipStream_t ms = comm_stream ? (hipStream_t)comm_stream : cs;
  if (ms != cs) {
    if (!ctx->ev_shard) VS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_shard, hipEventDisableTiming));
    VS_HIP(ctx, hipEventRecord(ctx->ev_shard, cs));
    VS_HIP(ctx, hipStreamWaitEvent(ms, ctx->ev_shard, 0));
  }
  // in place: every rank's send buffer is its own slot of the receive buffer (ncclInt32 = 2)
  const int rc = ag(slot, d_gathered, (size_t)per * 4, 2, nccl_comm, ms);
  if (rc != 0) return vs_fail(ctx, VS_ENCCL, "%s: ncclAllGather failed", "vs_hamming_knn2_sharded_dev");
  if (done_event) VS_HIP(ctx, hipEventRecord((hipEvent_t)done_event, ms));
  return VS_OK;
}

// bench hooks (not part of the stable ABI): HIP-event timing of the match kernel on its launch stream
VS_API int vs_match_profile(vs_ctx* ctx, int enable) {
  if (!ctx) return VS_EINVAL;
  ctx->tune.match_profile = enable != 0;
  return VS_OK;
}
// synchronises, returns the number of profiled calls and their mean kernel duration in milliseconds, then clears
VS_API int vs_match_profile_read(vs_ctx* ctx, float* kernel_ms) {
  if (!ctx) return VS_EINVAL;
  double a = 0;
  int n = 0;
  for (vs_prof_rec& r : ctx->match_prof) {
    float x = 0;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&x, r.e0, r.e1) == hipSuccess) {
      a += x;
      ++n;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  ctx->match_prof.clear();
  if (kernel_ms) *kernel_ms = n ? (float)(a / n) : 0.f;
  return n;
}

static size_t func_scratch(const void* fn) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, fn) != hipSuccess) {
    (void)hipGetLastError();
    return (size_t)-1;
  }
  return a.localSizeBytes;
}
size_t vs_match_chain_scratch_bytes(vs_ctx* ctx) {
  const size_t a = ctx->tune.match_tstage ? func_scratch((const void*)hamming_knn2_kernel<true>) : func_scratch((const void*)hamming_knn2_kernel<false>);
  const size_t b = func_scratch((const void*)ratio_compact_kernel);
  return a > b ? a : b;
}

// diagnostic hooks (include/vslam_hip_dev.h): per-workgroup phase stamps of the match kernel.  vs_match_stamps_read synchronises and
// returns the newest stamped launch as rows of 8 doubles, one per workgroup in (chunk, tile) order: microseconds since the
// launch's first stamp of [0] start, [1] operands arrived, [2] wave 0 through its scan, [3] all waves through, [4] partial
// stored (publishing workgroups), [5] all partials arrived (folding workgroups), [6] results written; [7] = XCC_ID * 2^16 +
// HW_ID bits (where the workgroup ran).  0 = phase not reached.  Returns the number of rows.
VS_API int vs_match_stamps(vs_ctx* ctx, int enable) {
  if (!ctx) return VS_EINVAL;
  ctx->match_stamps_on = enable != 0;
  return VS_OK;
}
VS_API int vs_match_stamps_read(vs_ctx* ctx, double* out, int cap_rows) {
  if (!ctx || !out) return VS_EINVAL;
  const int rows = ctx->match_stamps_rows;
  if (!ctx->d_match_stamps.p || rows <= 0 || cap_rows < rows) return 0;
  std::vector<unsigned long long> h((size_t)rows * 8);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(h.data(), ctx->d_match_stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return vs_fail(ctx, VS_EHIP, "%s: read-back failed", "vs_match_stamps_read");
  unsigned long long t0 = ~0ull;
  for (size_t i = 0; i < h.size(); ++i)
    if ((i & 7) != 7 && h[i] && !(h[i] >> 63) && h[i] < t0) t0 = h[i];
  for (size_t i = 0; i < h.size(); ++i) {
    if ((i & 7) == 7) out[i] = (double)(((h[i] >> 32) & 0xFFFF) * 65536ull + (h[i] & 0xFFFF));
    else if (h[i] >> 63) out[i] = -(double)(h[i] & ~(1ull << 63));  // a raw cycle count (negative: not a time)
    else out[i] = h[i] ? (double)(h[i] - t0) * 0.01 : 0.0;  // wall_clock64: 100 MHz
  }
  return rows;
}

// nt_dev != nullptr: at most nt train rows, the actual count is read on the device (internal: vs_track.hip matches a frame
// whose key-point count the host has not seen)
int vs_match_ratio_dev_n(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, const int* nt_dev, double ratio,
                         void* d_match_q, void* d_match_t, void* d_match_d, void* d_n_out, void* stream, int32_t* h_mirror,
                         int h_stride) {
  VS_TRY(check_args(ctx, d_q, nq, d_t, nt, "vs_match_ratio_dev"));
  if (!d_n_out || (nq > 0 && (!d_match_q || !d_match_t || !d_match_d)))
    return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_match_ratio_dev");
  hipStream_t s = vs_pick_stream(ctx, stream);
  // the 2-NN rows between the two kernels belong to the launch stream's scratch set: calls on different streams (a host
  // match while a pipelined front half is in flight, two streams of a caller) never share them
  vs_match_scratch* ms = nullptr;
  VS_TRY(match_scratch_for(ctx, s, &ms));
  VS_TRY(vs_reserve(ctx, &ms->idx, sizeof(int2) * (size_t)(nq > 0 ? nq : 1)));
  VS_TRY(vs_reserve(ctx, &ms->dist, sizeof(int2) * (size_t)(nq > 0 ? nq : 1)));
  VS_HIP(ctx, hipSetDevice(ctx->device));
  VS_TRY(knn2_dev_impl(ctx, d_q, nq, d_t, nt, ms->idx.p, ms->dist.p, false, s, nt_dev));
  hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, s, (const int2*)ms->idx.p,
                     (const int2*)ms->dist.p, nq, ratio, (int32_t*)d_match_q, (int32_t*)d_match_t,
                     (int32_t*)d_match_d, (int32_t*)d_n_out, nt_dev, h_mirror, h_stride);
  VS_LAUNCH_CHECK(ctx, "ratio_compact_kernel");
  return VS_OK;
}

VS_API int vs_match_ratio_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, double ratio,
                              void* d_match_q, void* d_match_t, void* d_match_d, void* d_n_out, void* stream) {
  return vs_match_ratio_dev_n(ctx, d_q, nq, d_t, nt, nullptr, ratio, d_match_q, d_match_t, d_match_d, d_n_out, stream, nullptr, 0);
}

VS_API int vs_hamming_knn2(vs_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                           int32_t* dist) {
  VS_TRY(check_args(ctx, q, nq, t, nt, "vs_hamming_knn2"));
  if (nq == 0) return VS_OK;
  if (!idx || !dist) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_hamming_knn2");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const void *dq, *dt;  // device-resident copies (uploaded unless these very bytes are already resident: exact compare)
  VS_TRY(vs_desc_resident(ctx, q, nq, 0, &dq));
  VS_TRY(vs_desc_resident(ctx, t, nt, 1, &dt));
  // results go to a private pair of buffers (d_mq/d_mt) so they never alias the ratio path's d_idx/d_dist
  VS_TRY(vs_reserve(ctx, &ctx->d_mq, sizeof(int2) * (size_t)nq));
  VS_TRY(vs_reserve(ctx, &ctx->d_mt, sizeof(int2) * (size_t)nq));
  VS_TRY(vs_hamming_knn2_dev(ctx, dq, nq, dt, nt, ctx->d_mq.p, ctx->d_mt.p, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(idx, ctx->d_mq.p, sizeof(int2) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(dist, ctx->d_mt.p, sizeof(int2) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return match_check_flag(ctx, ctx->stream, "vs_hamming_knn2");
}

VS_API int vs_match_ratio(vs_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                          int32_t* match_q, int32_t* match_t, int32_t* match_d, int* n_out) {
  VS_TRY(check_args(ctx, q, nq, t, nt, "vs_match_ratio"));
  if (!n_out) return vs_fail(ctx, VS_EINVAL, "%s: n_out is NULL", "vs_match_ratio");
  *n_out = 0;
  if (nq == 0) return VS_OK;
  if (!match_q || !match_t || !match_d) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_match_ratio");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const void *dq, *dt;
  VS_TRY(vs_desc_resident(ctx, q, nq, 0, &dq));
  VS_TRY(vs_desc_resident(ctx, t, nt, 1, &dt));
  // one device block [count (16 B) | match_q | match_t | match_d], copied back in one piece: one synchronisation
  const size_t row = (sizeof(int32_t) * (size_t)nq + 15) & ~(size_t)15;
  const size_t bytes = 16 + 3 * row;
  VS_TRY(vs_reserve(ctx, &ctx->d_mq, bytes));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, bytes));
  uint8_t* blk = (uint8_t*)ctx->d_mq.p;
  VS_TRY(vs_match_ratio_dev(ctx, dq, nq, dt, nt, ratio, blk + 16, blk + 16 + row, blk + 16 + 2 * row, blk, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(ctx->h_pin.p, blk, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  VS_TRY(match_check_flag(ctx, ctx->stream, "vs_match_ratio"));
  const uint8_t* hp = (const uint8_t*)ctx->h_pin.p;
  const int n = *(const int32_t*)hp;
  if (n < 0 || n > nq) return vs_fail(ctx, VS_EHIP, "%s: device returned an impossible match count", "vs_match_ratio");
  memcpy(match_q, hp + 16, sizeof(int32_t) * (size_t)n);
  memcpy(match_t, hp + 16 + row, sizeof(int32_t) * (size_t)n);
  memcpy(match_d, hp + 16 + 2 * row, sizeof(int32_t) * (size_t)n);
  *n_out = n;
  return VS_OK;
}
