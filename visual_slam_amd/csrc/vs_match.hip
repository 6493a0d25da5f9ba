// vs_match.hip -- brute-force Hamming 2-NN on 256-bit descriptors + Lowe ratio compaction (gfx950).
//
// Replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(k=2) and the ratio loop of FeatureMatcher.match_features
// (reference src/v2/frame.py:18,23,25-47).
//
// Kernel design (VALU-bound integer work, no MFMA):
//   * lane = query.  Every lane keeps QPL query descriptors (8 dwords each) in VGPRs for the whole kernel.
//   * train descriptors are wave-uniform: they are read with scalar loads (s_load_dwordx8) into SGPRs and fed to
//     v_xor_b32 as the scalar operand -- no LDS traffic, no VGPRs, one 32-byte scalar load per 64*QPL distances.
//   * distance = 8 x (v_xor_b32 + v_bcnt_u32_b32 with accumulate), a dependent chain per query, QPL chains in flight.
//   * top-2 bookkeeping on packed keys  key = dist << 20 | (train index within the chunk):
//       second = v_med3_u32(best, second, key);  best = v_min_u32(best, key)
//     so the tie rule "lower train index first" falls out of the packing.  19 VALU ops per distance.
//   * the grid is (query tiles) x (train chunks) so that ~4 waves sit on every SIMD even at 10k x 10k; each
//     (chunk, query) pair writes one 8-byte partial; a second tiny kernel merges the chunks per query in chunk order.
//   * the ratio test + ordered compaction is one workgroup using wave ballots and popcounts for the prefix.
#include "vs_internal.h"

#include <vector>

namespace {

constexpr int kWaves = 4;                  // waves per workgroup
constexpr int kIdxBits = 20;
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;
constexpr int kMaxChunk = 1 << kIdxBits;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// one train descriptor = one 32-byte scalar load (s_load_dwordx8); rows are 32-byte aligned
__device__ __forceinline__ u32x8 load_train(const uint32_t* p) { return *reinterpret_cast<const u32x8*>(p); }

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  return min(max(a, b), max(min(a, b), c));  // hipcc folds this to v_med3_u32
}

// asm-issued scalar load: invisible to the compiler's waitcnt insertion, so the wait is placed by hand (swait)
__device__ __forceinline__ u32x8 sload8_async(const uint32_t* p) {
  u32x8 r;
  asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(p));
  return r;
}
template <int N>
__device__ __forceinline__ void swait(u32x8 (&v)[N]) {
  static_assert(N == 2 || N == 4, "");
  if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v[0]), "+s"(v[1]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]));
}

template <int QPL>
__device__ __forceinline__ void accumulate(const uint32_t (&qv)[QPL][8], const u32x8& tw, uint32_t j,
                                           uint32_t (&b1)[QPL], uint32_t (&b2)[QPL]) {
#pragma unroll
  for (int r = 0; r < QPL; ++r) {
    uint32_t acc = bcnt_acc(qv[r][0] ^ tw[0], 0u);
#pragma unroll
    for (int k = 1; k < 8; ++k) acc = bcnt_acc(qv[r][k] ^ tw[k], acc);
    uint32_t key = (acc << kIdxBits) | j;
    b2[r] = umed3(b1[r], b2[r], key);
    b1[r] = min(b1[r], key);
  }
}

// partial[chunk][q] = (best key, second key) of query q over the trains of `chunk`.
// A workgroup = kWaves waves that hold the SAME 64*QPL queries; wave w scans the w-th quarter of the chunk and the
// four (best, second) pairs are merged through LDS, so only one 8-byte partial per (chunk, query) reaches HBM.
// QPL: queries per lane; TU: train descriptors per step; PIPE: software-pipelined scalar loads (asm + manual wait)
template <int QPL, int TU, bool PIPE>
__global__ __launch_bounds__(64 * kWaves) void hamming_partial_kernel(const uint4* __restrict__ q4, int nq,
                                                                        const uint32_t* __restrict__ t, int nt,
                                                                        int chunk_len, int sub_len,
                                                                        uint2* __restrict__ partial) {
  __shared__ uint2 lds[kWaves][64 * QPL];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> SGPR
  const int qbase = blockIdx.x * (64 * QPL);
  const int cbegin = blockIdx.y * chunk_len;
  const int cend = min(cbegin + chunk_len, nt);
  const int begin = min(cbegin + wave * sub_len, cend);
  const int end = min(begin + sub_len, cend);

  uint32_t qv[QPL][8];
#pragma unroll
  for (int r = 0; r < QPL; ++r) {
    int qi = min(qbase + r * 64 + lane, nq - 1);  // clamp: the tail computes a duplicate, the store is guarded
    uint4 a = q4[2 * (size_t)qi], b = q4[2 * (size_t)qi + 1];
    qv[r][0] = a.x; qv[r][1] = a.y; qv[r][2] = a.z; qv[r][3] = a.w;
    qv[r][4] = b.x; qv[r][5] = b.y; qv[r][6] = b.z; qv[r][7] = b.w;
  }
  uint32_t b1[QPL], b2[QPL];
#pragma unroll
  for (int r = 0; r < QPL; ++r) b1[r] = b2[r] = kEmpty;

  const uint32_t* tp = t + (size_t)begin * 8;  // wave-uniform -> scalar loads
  const uint32_t j0 = (uint32_t)(begin - cbegin); // keys carry the index within the workgroup's chunk
  const int n = end - begin;
  int j = 0;
  if constexpr (PIPE) {
    if (n >= TU) {
      u32x8 cur[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) cur[u] = sload8_async(tp + (size_t)u * 8);
      swait(cur);
      for (; j + TU <= n; j += TU) {
        const int jn = min(j + TU, n - TU);  // the last step re-reads itself: always in bounds
        u32x8 nxt[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) nxt[u] = sload8_async(tp + (size_t)(jn + u) * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TU; ++u) accumulate<QPL>(qv, cur[u], j0 + (uint32_t)(j + u), b1, b2);
        __builtin_amdgcn_sched_barrier(0);
        swait(nxt);
#pragma unroll
        for (int u = 0; u < TU; ++u) cur[u] = nxt[u];
      }
    }
  } else {
    for (; j + TU <= n; j += TU) {
      u32x8 tw[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) tw[u] = load_train(tp + (size_t)(j + u) * 8);
#pragma unroll
      for (int u = 0; u < TU; ++u) accumulate<QPL>(qv, tw[u], j0 + (uint32_t)(j + u), b1, b2);
    }
  }
  for (; j < n; ++j) {
    u32x8 tw = load_train(tp + (size_t)j * 8);
    accumulate<QPL>(qv, tw, j0 + (uint32_t)j, b1, b2);
  }
#pragma unroll
  for (int r = 0; r < QPL; ++r) lds[wave][r * 64 + lane] = make_uint2(b1[r], b2[r]);
  __syncthreads();
  // keys of different waves are distinct (disjoint index ranges), so min / med3 merge them exactly
  for (int i = threadIdx.x; i < 64 * QPL; i += 64 * kWaves) {
    uint2 m = lds[0][i];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      uint2 o = lds[w][i];
      uint32_t s2 = min(max(m.x, o.x), min(m.y, o.y));
      m.x = min(m.x, o.x);
      m.y = s2;
    }
    const int qi = qbase + i;
    if (qi < nq) partial[(size_t)blockIdx.y * nq + qi] = m;
  }
}

// kMergeLanes lanes per query: lane g folds chunks g, g + L, ... (all its loads are issued before the first use),
// then log2(L) xor-shuffles combine the lanes.  Ties cannot occur between chunks (distinct global indices).
// PACKED: one 16-byte row (idx0, idx1, dist0, dist1) per query -- the layout the query-sharded matcher all-gathers.
constexpr int kMergeLanes = 8;
template <bool PACKED>
__global__ __launch_bounds__(256) void hamming_merge_kernel(const uint2* __restrict__ partial, int nq, int nchunks,
                                                             int chunk_len, int2* __restrict__ idx,
                                                             int2* __restrict__ dist) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid / kMergeLanes, g = tid % kMergeLanes;
  const int qc = min(q, nq - 1);
  unsigned long long B1 = ~0ull, B2 = ~0ull;
  for (int c0 = g; c0 < nchunks; c0 += 4 * kMergeLanes) {
    uint2 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * kMergeLanes;
      p[u] = c < nchunks ? partial[(size_t)c * nq + qc] : make_uint2(kEmpty, kEmpty);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long base = (unsigned long long)(c0 + u * kMergeLanes) * (unsigned long long)chunk_len;
      if (p[u].x != kEmpty) {
        unsigned long long K = ((unsigned long long)(p[u].x >> kIdxBits) << 32) | (base + (p[u].x & kIdxMask));
        B2 = min(B2, max(B1, K));
        B1 = min(B1, K);
      }
      if (p[u].y != kEmpty) {
        unsigned long long K = ((unsigned long long)(p[u].y >> kIdxBits) << 32) | (base + (p[u].y & kIdxMask));
        B2 = min(B2, max(B1, K));
        B1 = min(B1, K);
      }
    }
  }
#pragma unroll
  for (int off = 1; off < kMergeLanes; off <<= 1) {
    unsigned long long O1 = __shfl_xor(B1, off), O2 = __shfl_xor(B2, off);
    unsigned long long S2 = min(max(B1, O1), min(B2, O2));
    B1 = min(B1, O1);
    B2 = S2;
  }
  if (g == 0 && q < nq) {
    if (PACKED) {
      reinterpret_cast<int4*>(idx)[q] =
          make_int4((int)(B1 & 0xFFFFFFFFull), (int)(B2 & 0xFFFFFFFFull), (int)(B1 >> 32), (int)(B2 >> 32));
    } else {
      idx[q] = make_int2((int)(B1 & 0xFFFFFFFFull), (int)(B2 & 0xFFFFFFFFull));
      dist[q] = make_int2((int)(B1 >> 32), (int)(B2 >> 32));
    }
  }
}

// Lowe ratio test + ordered compaction: one workgroup, wave ballots + popcounts give the ranks.
__global__ __launch_bounds__(1024) void ratio_compact_kernel(const int2* __restrict__ idx, const int2* __restrict__ dist,
                                                              int nq, double ratio, int32_t* __restrict__ mq,
                                                              int32_t* __restrict__ mt, int32_t* __restrict__ md,
                                                              int32_t* __restrict__ n_out) {
  __shared__ int wave_cnt[16];
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
    }
    running += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = running;
}

int g_target_blocks = 0;     // 0: automatic plan (plan_chunks); > 0: fixed number of workgroups (tuning hook)

// optional per-kernel timing (bench.py): hipEvents on the launch stream around the two kernels of each call
bool g_profile = false;
struct prof_rec {
  hipEvent_t e0, e1, e2;
};
std::vector<prof_rec> g_prof;
int g_variant = 0;           // see launch_partial

typedef void (*partial_fn)(const uint4*, int, const uint32_t*, int, int, int, uint2*);
struct variant_t {
  partial_fn fn;
  int qpl, tu;
};
const variant_t kVariants[] = {
    {hamming_partial_kernel<4, 4, false>, 4, 4},
    {hamming_partial_kernel<4, 4, true>, 4, 4},
    {hamming_partial_kernel<2, 4, true>, 2, 4},
    {hamming_partial_kernel<4, 2, true>, 4, 2},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// chunk_len trains per workgroup (split into kWaves sub-ranges of sub_len), nchunks workgroups along y.
// Automatic plan (g_target_blocks == 0): pick the number of train chunks that minimises
//     ceil(workgroups / 256 CUs) * chunk_len  +  merge cost per chunk
// over plans with 1000..4600 workgroups -- the first term is the busiest CU's share of train rows (workgroups are
// spread round-robin, 4..16 resident per CU), the second the extra partial rows the merge kernel folds.  At 10k x 10k
// this gives 32 chunks x 40 query tiles = 1280 workgroups (25 chunks cost 0.2 % more), at 100k x 100k 11 x 391 = 4301 (98.8 % balanced; a fixed
// 1024-workgroup plan loses 25 % there to 4-vs-3 workgroups per CU).
void plan_chunks(int nq, int nt, int* chunk_len, int* sub_len, int* nchunks) {
  const variant_t& v = kVariants[g_variant];
  const int qtiles = (nq + 64 * v.qpl - 1) / (64 * v.qpl);
  const int q1 = qtiles > 0 ? qtiles : 1;
  long nch = 1;
  if (g_target_blocks > 0) {
    nch = g_target_blocks / q1;
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
  if (nch < 1) nch = 1;
  long len = (nt + nch - 1) / nch;
  long sub = (len + kWaves - 1) / kWaves;
  sub = (sub + v.tu - 1) / v.tu * v.tu;
  if (sub < 2 * v.tu) sub = 2 * v.tu;
  len = sub * kWaves;
  if (len > kMaxChunk) {
    len = kMaxChunk;
    sub = len / kWaves;
  }
  *chunk_len = (int)len;
  *sub_len = (int)sub;
  *nchunks = (int)((nt + len - 1) / len);
}

int check_args(vs_ctx* ctx, const void* q, int nq, const void* t, int nt, const char* fn) {
  if (!ctx) return VS_EINVAL;
  if (nq < 0 || nt < 2) return vs_fail(ctx, VS_EINVAL, "%s: need nq >= 0 and nt >= 2 (k = 2 neighbours)", fn);
  if ((nq > 0 && !q) || !t) return vs_fail(ctx, VS_EINVAL, "%s: null descriptor pointer", fn);
  return VS_OK;
}

}  // namespace

// tuning hook for bench sweeps (not part of the stable ABI)
VS_API int vs_match_set_target_blocks(int blocks) {
  if (blocks >= 0) g_target_blocks = blocks;
  return g_target_blocks;
}
VS_API int vs_match_set_variant(int v) {
  if (v >= 0 && v < kNumVariants) g_variant = v;
  return g_variant;
}

static int knn2_dev_impl(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_idx, void* d_dist,
                         bool packed, void* stream) {
  VS_TRY(check_args(ctx, d_q, nq, d_t, nt, "vs_hamming_knn2_dev"));
  if (nq == 0) return VS_OK;
  if (!d_idx || (!packed && !d_dist)) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_hamming_knn2_dev");
  hipStream_t s = vs_pick_stream(ctx, stream);
  int chunk_len, sub_len, nchunks;
  plan_chunks(nq, nt, &chunk_len, &sub_len, &nchunks);
  if (((uintptr_t)d_q | (uintptr_t)d_t) & 31)
    return vs_fail(ctx, VS_EINVAL, "%s: descriptor arrays must be 32-byte aligned", "vs_hamming_knn2_dev");
  VS_TRY(vs_reserve(ctx, &ctx->d_partial, sizeof(uint2) * (size_t)nchunks * nq));
  const int tile_q = 64 * kVariants[g_variant].qpl;
  dim3 grid((nq + tile_q - 1) / tile_q, nchunks);
  prof_rec pr{};
  if (g_profile) {
    VS_HIP(ctx, hipEventCreate(&pr.e0));
    VS_HIP(ctx, hipEventCreate(&pr.e1));
    VS_HIP(ctx, hipEventCreate(&pr.e2));
    VS_HIP(ctx, hipEventRecord(pr.e0, s));
  }
  hipLaunchKernelGGL(kVariants[g_variant].fn, grid, dim3(64 * kWaves), 0, s, (const uint4*)d_q, nq,
                     (const uint32_t*)d_t, nt, chunk_len, sub_len, (uint2*)ctx->d_partial.p);
  VS_LAUNCH_CHECK(ctx, "hamming_partial_kernel");
  if (g_profile) VS_HIP(ctx, hipEventRecord(pr.e1, s));
  const long merge_threads = (long)nq * kMergeLanes;
  if (packed)
    hipLaunchKernelGGL(hamming_merge_kernel<true>, dim3((unsigned)((merge_threads + 255) / 256)), dim3(256), 0, s,
                       (const uint2*)ctx->d_partial.p, nq, nchunks, chunk_len, (int2*)d_idx, (int2*)nullptr);
  else
    hipLaunchKernelGGL(hamming_merge_kernel<false>, dim3((unsigned)((merge_threads + 255) / 256)), dim3(256), 0, s,
                       (const uint2*)ctx->d_partial.p, nq, nchunks, chunk_len, (int2*)d_idx, (int2*)d_dist);
  VS_LAUNCH_CHECK(ctx, "hamming_merge_kernel");
  if (g_profile) {
    VS_HIP(ctx, hipEventRecord(pr.e2, s));
    g_prof.push_back(pr);
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

// bench hooks (not part of the stable ABI): HIP-event timing of the two match kernels on their launch stream
VS_API int vs_match_profile(int enable) {
  g_profile = enable != 0;
  return 0;
}
// synchronises, returns the number of profiled calls and their mean kernel durations in milliseconds, then clears
VS_API int vs_match_profile_read(float* partial_ms, float* merge_ms) {
  double a = 0, b = 0;
  int n = 0;
  for (prof_rec& r : g_prof) {
    float x = 0, y = 0;
    if (hipEventSynchronize(r.e2) == hipSuccess && hipEventElapsedTime(&x, r.e0, r.e1) == hipSuccess &&
        hipEventElapsedTime(&y, r.e1, r.e2) == hipSuccess) {
      a += x;
      b += y;
      ++n;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    (void)hipEventDestroy(r.e2);
  }
  g_prof.clear();
  if (partial_ms) *partial_ms = n ? (float)(a / n) : 0.f;
  if (merge_ms) *merge_ms = n ? (float)(b / n) : 0.f;
  return n;
}

VS_API int vs_match_ratio_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, double ratio,
                              void* d_match_q, void* d_match_t, void* d_match_d, void* d_n_out, void* stream) {
  VS_TRY(check_args(ctx, d_q, nq, d_t, nt, "vs_match_ratio_dev"));
  if (!d_n_out || (nq > 0 && (!d_match_q || !d_match_t || !d_match_d)))
    return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_match_ratio_dev");
  hipStream_t s = vs_pick_stream(ctx, stream);
  VS_TRY(vs_reserve(ctx, &ctx->d_idx, sizeof(int2) * (size_t)(nq > 0 ? nq : 1)));
  VS_TRY(vs_reserve(ctx, &ctx->d_dist, sizeof(int2) * (size_t)(nq > 0 ? nq : 1)));
  VS_TRY(vs_hamming_knn2_dev(ctx, d_q, nq, d_t, nt, ctx->d_idx.p, ctx->d_dist.p, s));
  hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, s, (const int2*)ctx->d_idx.p,
                     (const int2*)ctx->d_dist.p, nq, ratio, (int32_t*)d_match_q, (int32_t*)d_match_t,
                     (int32_t*)d_match_d, (int32_t*)d_n_out);
  VS_LAUNCH_CHECK(ctx, "ratio_compact_kernel");
  return VS_OK;
}

VS_API int vs_hamming_knn2(vs_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                           int32_t* dist) {
  VS_TRY(check_args(ctx, q, nq, t, nt, "vs_hamming_knn2"));
  if (nq == 0) return VS_OK;
  if (!idx || !dist) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", "vs_hamming_knn2");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const void *dq, *dt;  // device-resident copies (uploaded only when this content has not been seen at this address)
  VS_TRY(vs_desc_resident(ctx, q, nq, &dq));
  VS_TRY(vs_desc_resident(ctx, t, nt, &dt));
  // results go to a private pair of buffers (d_mq/d_mt) so they never alias the ratio path's d_idx/d_dist
  VS_TRY(vs_reserve(ctx, &ctx->d_mq, sizeof(int2) * (size_t)nq));
  VS_TRY(vs_reserve(ctx, &ctx->d_mt, sizeof(int2) * (size_t)nq));
  VS_TRY(vs_hamming_knn2_dev(ctx, dq, nq, dt, nt, ctx->d_mq.p, ctx->d_mt.p, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(idx, ctx->d_mq.p, sizeof(int2) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(dist, ctx->d_mt.p, sizeof(int2) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
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
  VS_TRY(vs_desc_resident(ctx, q, nq, &dq));
  VS_TRY(vs_desc_resident(ctx, t, nt, &dt));
  // one device block [count (16 B) | match_q | match_t | match_d], copied back in one piece: one synchronisation
  const size_t row = (sizeof(int32_t) * (size_t)nq + 15) & ~(size_t)15;
  const size_t bytes = 16 + 3 * row;
  VS_TRY(vs_reserve(ctx, &ctx->d_mq, bytes));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, bytes));
  uint8_t* blk = (uint8_t*)ctx->d_mq.p;
  VS_TRY(vs_match_ratio_dev(ctx, dq, nq, dt, nt, ratio, blk + 16, blk + 16 + row, blk + 16 + 2 * row, blk, ctx->stream));
  VS_HIP(ctx, hipMemcpyAsync(ctx->h_pin.p, blk, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint8_t* hp = (const uint8_t*)ctx->h_pin.p;
  const int n = *(const int32_t*)hp;
  if (n < 0 || n > nq) return vs_fail(ctx, VS_EHIP, "%s: device returned an impossible match count", "vs_match_ratio");
  memcpy(match_q, hp + 16, sizeof(int32_t) * (size_t)n);
  memcpy(match_t, hp + 16 + row, sizeof(int32_t) * (size_t)n);
  memcpy(match_d, hp + 16 + 2 * row, sizeof(int32_t) * (size_t)n);
  *n_out = n;
  return VS_OK;
}
