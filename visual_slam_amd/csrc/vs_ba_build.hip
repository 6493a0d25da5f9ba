// vs_ba_build.hip -- the sparsity structure of a LARGE bundle-adjustment problem, built on the device (gfx950).
//
// vs_ba_solve's host passes (vs_ba.hip) turn the caller's observation list -- what the reference hands to g2o edge by edge,
// src/v2/LocalBA.py:115-131 -- into the index arrays the kernels work from: observation ranges per point, the Hpl block of
// every observation, the observation list of every free camera, camera-tile masks, and the banded-window plan.  At 2 000 000
// observations those passes cost 4.6 ms on twelve host threads, as much as all the kernels of the solve.  The observation
// arrays are DMA-ed to the device anyway, so for large problems whose list arrives grouped by point (as the reference adds
// its edges) the same arrays are produced here, from the device copy, in ten short launches:
//
//   build_obs_keys      thread = observation: index checks, "grouped by point" check, camera key, Hpl flag, and -- where the
//                       point changes -- the start of every point's observation range
//   split_count / split_scan / split_scatter
//                       a stable counting sort by a small key (camera slot: <= 1 000 keys): per-block histograms in LDS, a
//                       column-wise exclusive scan over the blocks, then the scatter with wave-ordered ranks -- every
//                       observation list comes out in ascending observation order, exactly as the host pass fills it.  The
//                       camera pass carries the exclusive scan of the Hpl flags along (block index of every Hpl block).
//   build_points        thread = point: tile mask, lowest / highest camera slot, duplicate cameras, most blocks per point
//   split_* again       the free points ordered by their lowest camera slot (the banded-window order)
//   build_window_plan   workgroup = slab of that order: first camera and camera span
//   build_window_first  first slab at or behind every camera slot
//
// Everything is integer work on index arrays: no floating point, integer atomics only (histograms, flags, maxima), so the
// arrays are bit-identical to the host passes' -- and so is the solve (tests/test_gpu_ba.py compares both).  Anything the
// fast path does not cover (an index out of range, a list not grouped by point, inactive observations or points, a camera
// twice in one point) raises a flag in `info`; the host then runs its own passes, which also own the error reporting.
#include "vs_ba_internal.h"

#include <limits.h>

using namespace vsba;

namespace {

constexpr int kSplitThreads = 256, kSplitRounds = 4, kSplitItems = kSplitThreads * kSplitRounds;
constexpr int kScanSegs = 16;

__global__ __launch_bounds__(256) void build_obs_keys(ba_build B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B.n_obs) return;
  const int ci = B.o_cam[i], pj = B.o_pt[i];
  int prev = i > 0 ? B.o_pt[i - 1] : -1;
  const bool bad = ci < 0 || ci >= B.F || pj < 0 || pj >= B.P;
  if (bad) {
    atomicOr(B.info + kBuildBad, 1);
    B.ckey[i] = B.nfp;
    B.o_hpl[i] = 0;
    return;
  }
  prev = max(-1, min(prev, B.P - 1));  // a bad neighbour is reported by its own thread; keep the loops below in range
  if (pj < prev) atomicOr(B.info + kBuildUngrouped, 1);
  const int cs = B.pose_slot[ci], ls = B.pt_slot[pj];
  if (cs < 0 && ls < 0) atomicOr(B.info + kBuildInactive, 1);
  B.ckey[i] = cs >= 0 ? cs : B.nfp;
  B.o_hpl[i] = cs >= 0 && ls >= 0;  // the flag; split_scatter turns it into the block index
  for (int j = prev + 1; j <= pj; ++j) B.pt_start[j] = i;  // points without observations in between start here, too
  if (i == B.n_obs - 1)
    for (int j = pj + 1; j <= B.P; ++j) B.pt_start[j] = B.n_obs;
}

// histogram of one block of kSplitItems keys (+ the number of set flags in column nkeys + 1)
template <bool FLAG>
__global__ __launch_bounds__(kSplitThreads) void split_count(const int* keys, const int* flags, int n, int nkeys, int* hist) {
  __shared__ int s_hist[kBuildMaxKeys + 2];
  const int tid = threadIdx.x, ncols = nkeys + 2;
  for (int k = tid; k < ncols; k += kSplitThreads) s_hist[k] = 0;
  __syncthreads();
  const int base = blockIdx.x * kSplitItems;
#pragma unroll
  for (int r = 0; r < kSplitRounds; ++r) {
    const int i = base + r * kSplitThreads + tid;
    if (i < n) {
      atomicAdd(&s_hist[keys[i]], 1);
      if (FLAG && flags[i]) atomicAdd(&s_hist[nkeys + 1], 1);
    }
  }
  __syncthreads();
  for (int k = tid; k < ncols; k += kSplitThreads) hist[(size_t)blockIdx.x * ncols + k] = s_hist[k];
}

// per column: exclusive prefix over the blocks, in place; the column's total to tot[c].  Workgroup = 64 columns x 16 row segments.
__global__ __launch_bounds__(64 * kScanSegs) void split_scan(int* hist, int nblk, int ncols, int* tot) {
  __shared__ int s_sum[kScanSegs][64];
  const int cl = threadIdx.x & 63, seg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const int per = (nblk + kScanSegs - 1) / kScanSegs, r0 = seg * per, r1 = min(nblk, r0 + per);
  int sum = 0;
  if (c < ncols)
    for (int r = r0; r < r1; ++r) sum += hist[(size_t)r * ncols + c];
  s_sum[seg][cl] = sum;
  __syncthreads();
  int run = 0;
  for (int s = 0; s < seg; ++s) run += s_sum[s][cl];
  if (c < ncols) {
    for (int r = r0; r < r1; ++r) {
      const int v = hist[(size_t)r * ncols + c];
      hist[(size_t)r * ncols + c] = run;
      run += v;
    }
    if (seg == kScanSegs - 1) tot[c] = run;
  }
}

// MODE 0: items = free points keyed by their lowest camera slot -> out_items (the window order)
// MODE 1: items = observations keyed by camera slot -> the camera lists (cam_obs, cam_pt), and the Hpl blocks: o_hpl, fp_slot, fp_start
template <int MODE>
__global__ __launch_bounds__(kSplitThreads) void split_scatter(ba_build B, const int* keys, int n, int nkeys, const int* hist, const int* tot,
                                                               int* out_items, int* out_start) {
  __shared__ int s_base[kBuildMaxKeys + 2];
  __shared__ int s_scan[kSplitThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ncols = nkeys + 2;
  // exclusive scan of the key totals (nkeys + 1 key columns: the last one holds the items without a key)
  {
    // thread t owns keys [t * per, (t + 1) * per)
    const int per = (nkeys + 1 + kSplitThreads - 1) / kSplitThreads;
    int sum = 0;
    for (int k = tid * per; k < min(nkeys + 1, (tid + 1) * per); ++k) sum += tot[k];
    // wave scan, then across the four waves
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    if (lane == 63) s_scan[wv] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wv; ++w) run += s_scan[w];
    for (int k = tid * per; k < min(nkeys + 1, (tid + 1) * per); ++k) {
      s_base[k] = run;
      run += tot[k];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int k = tid; k <= nkeys; k += kSplitThreads) out_start[k] = s_base[k];
    if (MODE == 1) {
      int mx = 0;
      for (int k = tid; k < nkeys; k += kSplitThreads) mx = max(mx, tot[k]);
      atomicMax(B.info + kBuildCamMax, mx);
      if (tid == 0) {
        B.info[kBuildHpl] = tot[nkeys + 1];
        B.fp_start[B.nfl] = tot[nkeys + 1];
      }
    } else if (tid == 0) {
      B.info[kBuildWinN] = s_base[nkeys];
    }
  }
  __syncthreads();
  for (int k = tid; k <= nkeys; k += kSplitThreads) s_base[k] += hist[(size_t)blockIdx.x * ncols + k];
  if (tid == 0) s_base[nkeys + 1] = MODE == 1 ? hist[(size_t)blockIdx.x * ncols + nkeys + 1] : 0;
  __syncthreads();
  const int base = blockIdx.x * kSplitItems;
  const unsigned long long lt = lane ? ~0ull >> (64 - lane) : 0ull;
  for (int r = 0; r < kSplitRounds; ++r) {
    const int i = base + r * kSplitThreads + tid;
    const bool valid = i < n;
    const int key = valid ? keys[i] : -1;
    const int flag = MODE == 1 && valid ? B.o_hpl[i] : 0;
    // rank among the lanes of this wave with the same key (lanes in ascending item order), and how many there are
    int rank = 0, cnt = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const int k = __shfl(key, lead);
      const unsigned long long same = __ballot(valid && key == k);
      if (key == k) {
        rank = __popcll(same & lt);
        cnt = __popcll(same);
      }
      todo &= ~same;
    }
    const unsigned long long fb = __ballot(flag != 0);
    const int frank = __popcll(fb & lt), fcnt = __popcll(fb);
    // the waves take their turns in item order
    int pos = 0, fpos = 0;
    for (int w = 0; w < kSplitThreads / 64; ++w) {
      if (wv == w) {
        if (valid) pos = s_base[key] + rank;
        fpos = s_base[nkeys + 1] + frank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == cnt - 1) s_base[key] = pos + 1;
        if (lane == 0) s_base[nkeys + 1] += fcnt;
      }
      __syncthreads();
    }
    if (!valid) continue;
    out_items[pos] = i;
    if (MODE == 1) {
      const int pj = B.o_pt[i];
      B.cam_pt[pos] = pj;
      B.o_hpl[i] = flag ? fpos : -1;
      if (flag) B.fp_slot[fpos] = key;
      // the first Hpl block of every free point whose range starts here (see build_obs_keys), and of those behind the last
      if (pj >= 0 && pj < B.P) {
        const int prev = i > 0 ? max(-1, min(B.o_pt[i - 1], B.P - 1)) : -1;
        for (int j = prev + 1; j <= pj; ++j) {
          const int ls = B.pt_slot[j];
          if (ls >= 0) B.fp_start[ls] = fpos;
        }
        if (i == n - 1)
          for (int j = pj + 1; j < B.P; ++j) {
            const int ls = B.pt_slot[j];
            if (ls >= 0) B.fp_start[ls] = fpos + flag;
          }
      }
    }
  }
}

__global__ __launch_bounds__(256) void build_points(ba_build B) {
  __shared__ int s_mm[4], s_dd[4], s_in[4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  int mf = 0, dup = 0, inact = 0;
  if (j < B.P) {
    // (a list with an index out of range or not grouped by point leaves entries of pt_start unwritten or out of order: the host
    // passes will take the problem over, but this kernel still runs -- keep every range inside the observation list)
    const int i0 = min(max(B.pt_start[j], 0), B.n_obs), i1 = min(max(B.pt_start[j + 1], i0), B.n_obs), ls = B.pt_slot[j];
    B.act_pt[j] = j;
    inact = i0 >= i1 && ls < 0;
    if (ls >= 0) {
      int lo = INT_MAX, hi = -1;
      unsigned long long mask = 0ull;
      if (i1 - i0 > kBuildMaxPerPoint) {
        dup = 1;  // not worth a quadratic scan here: the host pass takes it
      } else {
        for (int i = i0; i < i1; ++i) {
          const int cs = B.ckey[i];
          if (cs >= B.nfp) continue;
          ++mf;
          mask |= 1ull << ((cs / B.tile_cams) & 63);
          lo = min(lo, cs);
          hi = max(hi, cs);
          for (int k = i0; k < i; ++k) dup |= B.ckey[k] == cs;
        }
      }
      if (B.fp_mask) B.fp_mask[ls] = mask;
      B.wlo[ls] = lo;
      B.whi[ls] = hi;
      B.wkey[ls] = hi >= 0 ? lo : B.nfp;
    }
  }
  // block maxima (integer: any order)
  for (int d = 32; d; d >>= 1) {
    mf = max(mf, __shfl_xor(mf, d));
    dup |= __shfl_xor(dup, d);
    inact |= __shfl_xor(inact, d);
  }
  if ((threadIdx.x & 63) == 0) {
    s_mm[threadIdx.x >> 6] = mf;
    s_dd[threadIdx.x >> 6] = dup;
    s_in[threadIdx.x >> 6] = inact;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int m = max(max(s_mm[0], s_mm[1]), max(s_mm[2], s_mm[3]));
    if (m > 1) atomicMax(B.info + kBuildMmax, m);
    if (s_dd[0] | s_dd[1] | s_dd[2] | s_dd[3]) atomicOr(B.info + kBuildDups, 1);
    if (s_in[0] | s_in[1] | s_in[2] | s_in[3]) atomicOr(B.info + kBuildInactive, 1);
  }
}

// slabs of the window order: the same cut as the host plan in vs_ba_solve (points per slab from the number of contributing points)
__global__ __launch_bounds__(256) void build_window_plan(ba_build B) {
  __shared__ int s_hi[4];
  const int win_n = B.win_start[B.nfp];
  int per = min(B.win_per_max, max(32, (win_n + B.win_target - 1) / B.win_target));
  if (B.win_per_tune > 0) per = min(B.win_per_max, B.win_per_tune);
  const int ns = (win_n + per - 1) / per;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    B.info[kBuildWinPer] = per;
    B.info[kBuildWinSlabs] = ns;
  }
  const int sl = blockIdx.x;
  if (sl >= ns || sl >= B.ns_cap) return;
  const int a = sl * per, b = min(a + per, win_n);
  const int lo = B.wlo[B.win_order[a]];  // sorted by it
  int hi = lo;
  for (int i = a + threadIdx.x; i < b; i += 256) hi = max(hi, B.whi[B.win_order[i]]);
  for (int d = 32; d; d >>= 1) hi = max(hi, __shfl_xor(hi, d));
  if ((threadIdx.x & 63) == 0) s_hi[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    B.win_w0[sl] = lo;
    B.win_len[sl] = hi - lo + 1;
    atomicMax(B.info + kBuildWinCams, hi - lo + 1);
  }
}

// first slab that starts at camera c or later (win_w0 ascends: the slabs cut an order sorted by it)
__global__ __launch_bounds__(256) void build_window_first(ba_build B) {
  const int ns = min(B.info[kBuildWinSlabs], B.ns_cap);
  for (int c = threadIdx.x; c <= B.nfp; c += 256) {
    int lo = 0, hi = ns;  // first sl in [0, ns] with w0[sl] >= c
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (B.win_w0[mid] < c) lo = mid + 1;
      else hi = mid;
    }
    B.win_first[c] = lo;
  }
}

}  // namespace

namespace vsba {

size_t ba_build_temp_ints(int n_obs, int nfp, int nfl) {
  const size_t nblk_o = ((size_t)n_obs + kSplitItems - 1) / kSplitItems, nblk_p = ((size_t)nfl + kSplitItems - 1) / kSplitItems;
  return (size_t)n_obs /*ckey*/ + 3 * (size_t)nfl /*wkey, wlo, whi*/ + (nblk_o + nblk_p + 2) * ((size_t)nfp + 2) /*hist, tot*/ + (nfp + 2) /*win_start*/ +
         kBuildInfoInts + 64 * 8 /*alignment of the eight pieces*/;
}

int ba_build_enqueue(vs_ctx* ctx, hipStream_t s, const ba_build& B) {
  if (B.nfp + 2 > kBuildMaxKeys + 2 || B.n_obs <= 0 || B.nfl <= 0) return vs_fail(ctx, VS_EINVAL, "%s: not a problem for the device-side structure", "ba_build_enqueue");
  const int nblk_o = (B.n_obs + kSplitItems - 1) / kSplitItems, nblk_p = (B.nfl + kSplitItems - 1) / kSplitItems;
  const int ncols = B.nfp + 2;
  VS_HIP(ctx, hipMemsetAsync(B.info, 0, sizeof(int) * kBuildInfoInts, s));
  hipLaunchKernelGGL(build_obs_keys, dim3((B.n_obs + 255) / 256), dim3(256), 0, s, B);
  VS_LAUNCH_CHECK(ctx, "build_obs_keys");
  hipLaunchKernelGGL(split_count<true>, dim3(nblk_o), dim3(kSplitThreads), 0, s, (const int*)B.ckey, (const int*)B.o_hpl, B.n_obs, B.nfp, B.hist_o);
  VS_LAUNCH_CHECK(ctx, "split_count");
  hipLaunchKernelGGL(split_scan, dim3((ncols + 63) / 64), dim3(64 * kScanSegs), 0, s, B.hist_o, nblk_o, ncols, B.tot_o);
  VS_LAUNCH_CHECK(ctx, "split_scan");
  hipLaunchKernelGGL(split_scatter<1>, dim3(nblk_o), dim3(kSplitThreads), 0, s, B, (const int*)B.ckey, B.n_obs, B.nfp, (const int*)B.hist_o, (const int*)B.tot_o,
                     B.cam_obs, B.cam_start);
  VS_LAUNCH_CHECK(ctx, "split_scatter");
  hipLaunchKernelGGL(build_points, dim3((B.P + 255) / 256), dim3(256), 0, s, B);
  VS_LAUNCH_CHECK(ctx, "build_points");
  if (B.win_order) {
    hipLaunchKernelGGL(split_count<false>, dim3(nblk_p), dim3(kSplitThreads), 0, s, (const int*)B.wkey, (const int*)nullptr, B.nfl, B.nfp, B.hist_p);
    VS_LAUNCH_CHECK(ctx, "split_count");
    hipLaunchKernelGGL(split_scan, dim3((ncols + 63) / 64), dim3(64 * kScanSegs), 0, s, B.hist_p, nblk_p, ncols, B.tot_p);
    VS_LAUNCH_CHECK(ctx, "split_scan");
    hipLaunchKernelGGL(split_scatter<0>, dim3(nblk_p), dim3(kSplitThreads), 0, s, B, (const int*)B.wkey, B.nfl, B.nfp, (const int*)B.hist_p, (const int*)B.tot_p,
                       B.win_order, B.win_start);
    VS_LAUNCH_CHECK(ctx, "split_scatter");
    if (B.ns_cap > 0) {
      hipLaunchKernelGGL(build_window_plan, dim3(B.ns_cap), dim3(256), 0, s, B);
      VS_LAUNCH_CHECK(ctx, "build_window_plan");
      hipLaunchKernelGGL(build_window_first, dim3(1), dim3(256), 0, s, B);
      VS_LAUNCH_CHECK(ctx, "build_window_first");
    }
  }
  return VS_OK;
}

}  // namespace vsba
