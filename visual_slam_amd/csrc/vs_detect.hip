// vs_detect.hip -- gray conversion, FAST-9/16 + 3x3 NMS + cap selection, BRIEF-256 (gfx950).
//
// Replaces FeatureExtractor.compute_features (reference src/v2/frame.py:10-14): np.mean gray, the keypoint detector
// (goodFeaturesToTrack there; FAST as in cv2.ORB per BASELINE.json) and extractor.compute.
//
// Two launches per frame:
//   detect_band_kernel   one workgroup per band of kBand image rows.  The band (+4 halo rows) is converted BGR->gray
//                        straight into LDS, every pixel gets one thread-iteration: 16-pixel circle read from LDS,
//                        brighter/darker bit masks, "9 contiguous" bit test; a wave only enters the exact-score
//                        path when a ballot says one of its lanes is a corner.  Scores for the band (+1 halo row)
//                        stay in LDS for the 3x3 non-maximum suppression; survivors are compacted IN ROW-MAJOR ORDER
//                        with wave ballots + popcounts into the band's slot of `raw`.  The same workgroup also
//                        writes the 5x5 box-sum image (u16) that BRIEF samples.
//   select_describe_kernel  one workgroup per band: prefix over the band counts (and, only when more than max_kp
//                        survive, an LDS histogram of all scores to find the cut) gives every survivor its final
//                        row-major output slot; one wave per keypoint evaluates the 256 BRIEF tests, 64 per ballot.
// Everything is integer work and bit-exact against oracle/vs_oracle.c.
#include "vs_internal.h"

#include <chrono>
#include "../../include/vs_brief_pattern.h"

namespace {

#ifndef VS_DET_BAND
#define VS_DET_BAND 2
#endif
constexpr int kBand = VS_DET_BAND;  // image rows per workgroup (2: 240 workgroups at 640x480; 3 and 4 measured slower, see profiles/tried_and_dropped.md)
constexpr int kHalo = 4;         // 3 (circle radius) + 1 (NMS neighbour)
#ifndef VS_DET_THREADS
#define VS_DET_THREADS 1024
#endif
constexpr int kDetThreads = VS_DET_THREADS; // 16 waves: 4 per SIMD hide the LDS latency of the box / score phases (512 and 256 measured in round 4, see profiles/tried_and_dropped.md)
constexpr int kStageUnroll = 2; // staging groups per thread with their loads in flight together
constexpr int kSelThreads = 256;
constexpr int kMaxDim = 4096;    // x, y packed in 12 bits each

struct int8x4 {
  int8_t v[4];
};
__constant__ int8x4 c_brief[VS_BRIEF_NTESTS] = VS_BRIEF_PATTERN_INIT;

// Bresenham circle, radius 3, clockwise from 12 o'clock (same table as the oracle); compile-time so that the
// unrolled circle reads use immediate offsets
struct circle_t {
  int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
};
constexpr circle_t kCircle{};

__device__ __forceinline__ bool has_run9(uint32_t m16) {
  uint32_t m = m16 | (m16 << 16);
  uint32_t r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m >> 8;
  return (r & 0xFFFFu) != 0;
}

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

// m = max over the 16 arcs of 9 contiguous circle pixels of min(p - c) and of min(c - p); corner at t iff m > t
__device__ __forceinline__ int fast9_maxmin(const int (&d)[16]) {
  int a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = min3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    b[i] = max3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
  }
  int mb = -255, md = 255;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    mb = max(mb, min3i(a[i], a[(i + 3) & 15], a[(i + 6) & 15]));  // min over d[i..i+8]
    md = min(md, max3i(b[i], b[(i + 3) & 15], b[(i + 6) & 15]));  // max over d[i..i+8]
  }
  return max(mb, -md);
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  return v;
}

// ------------------------------------------------------------------------------------------------- detect
// LDS carve (dynamic): gray[(kBand+2*kHalo)][P] u8 | score[(kBand+2)][w4] u8 | hsum[(kBand+4)][w4] u16 | ballots u64[nseg]
// | segoff int[nseg]
// i / d for 0 <= i < 2^20 with magic = ceil(2^32 / d): one v_mul_hi_u32 instead of the ~40-instruction runtime division
__device__ __forceinline__ int fast_div(int i, uint32_t magic) { return (int)__umulhi((uint32_t)i, magic); }
__host__ inline uint32_t div_magic(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

template <bool FROM_BGR, bool DO_BOX, bool WRITE_GRAY, bool HOSTSRC = false>
__global__ __launch_bounds__(kDetThreads) void detect_band_kernel(const uint8_t* __restrict__ img, int pitch, int w,
                                                                   int h, int thr, int border,
                                                                   uint8_t* __restrict__ gray_out,
                                                                   uint16_t* __restrict__ box_out,
                                                                   uint32_t* __restrict__ raw, int band_cap,
                                                                   int* __restrict__ bandcnt, int* __restrict__ bandhist,
                                                                   uint32_t magic_w, uint32_t magic_w4, uint32_t magic_gpr,
                                                                   int* __restrict__ framehist, unsigned* band_flag, unsigned seq,
                                                                   int* host_err) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_bhist[256];  // scores of this band's survivors (the cap of select_describe_kernel needs their histogram)
  if (threadIdx.x < 256) s_bhist[threadIdx.x] = 0;
  const int w4 = (w + 3) & ~3;
  const int P = w4;
  const int nseg = (kBand * w + 63) >> 6;
  uint8_t* s_gray = smem;                                                    // (kBand + 2*kHalo) * P
  uint8_t* s_score = s_gray + (kBand + 2 * kHalo) * P;                       // (kBand + 2) * w4
  uint16_t* s_hsum = reinterpret_cast<uint16_t*>(s_score + (kBand + 2) * w4); // (kBand + 4) * w4
  unsigned long long* s_ballot =
      reinterpret_cast<unsigned long long*>(s_hsum + (DO_BOX ? (kBand + 4) * w4 : 0));  // nseg (8-byte aligned: w4 % 4 == 0)
  int* s_segoff = reinterpret_cast<int*>(s_ballot + nseg);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = blockIdx.x * kBand;

  // ---- 1. stage gray rows y0-kHalo .. y0+kBand+kHalo-1 in LDS, 4 pixels per group.  The global loads of up to
  // kStageUnroll groups per thread are all issued before the first use (the frame is read once, at HBM latency).
  const int groups_per_row = w4 >> 2;
  const int ngroups = (kBand + 2 * kHalo) * groups_per_row;
  if (HOSTSRC) {
    // The frame is read WHERE IT LIES, in the caller's pinned host memory, and every workgroup fetches only its own kBand rows
    // over the bus (with the halo each row would cross it five times): it converts them, keeps them in LDS and publishes them as
    // gray words in HBM -- write-through stores, drained, then the band's flag = this frame's sequence number.  The halo rows
    // are its neighbours' bands: it waits for their flags (all bands are resident at once -- the host checks that there are
    // at least as many compute units -- so everybody's rows are on their way) and reads them from HBM past the L1.  The
    // transfer and the detection overlap: a band starts as soon as its ten rows have arrived, not when the whole frame has.
    uint32_t* gray_words = reinterpret_cast<uint32_t*>(gray_out);  // pitch w4
    const int band = blockIdx.x, nbands_ = gridDim.x, own = kBand * groups_per_row;
    // (16-byte pieces parked in LDS and converted from there were tried: 36 us per launch against 34 with the three dword loads
    // per pixel group below -- a kernel reads the uncached host memory at ~27 GB/s either way, the copy engine at 51)
    for (int g = tid; g < own; g += kDetThreads) {
      const int r = fast_div(g, magic_gpr), x = (g - r * groups_per_row) << 2, y = y0 + r;
      uint32_t packed = 0;
      if (y < h) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(img + (size_t)y * pitch + 3 * x);
        const uint32_t d0 = __builtin_nontemporal_load(p), d1 = __builtin_nontemporal_load(p + 1), d2 = __builtin_nontemporal_load(p + 2);
        const uint32_t g0_ = ((d0 & 255) + ((d0 >> 8) & 255) + ((d0 >> 16) & 255)) / 3u;
        const uint32_t g1_ = ((d0 >> 24) + (d1 & 255) + ((d1 >> 8) & 255)) / 3u;
        const uint32_t g2_ = (((d1 >> 16) & 255) + (d1 >> 24) + (d2 & 255)) / 3u;
        const uint32_t g3_ = (((d2 >> 8) & 255) + ((d2 >> 16) & 255) + (d2 >> 24)) / 3u;
        packed = g0_ | (g1_ << 8) | (g2_ << 16) | (g3_ << 24);
        __hip_atomic_store(gray_words + (size_t)y * groups_per_row + (x >> 2), packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *reinterpret_cast<uint32_t*>(s_gray + (kHalo + r) * P + x) = packed;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(band_flag + band, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 4) {  // the two bands above and the two below (kHalo = 2 kBand)
      const int nb = band + (tid < 2 ? tid - 2 : tid - 1);
      if (nb >= 0 && nb < nbands_) {
        int it = 0;
        while ((int)(__hip_atomic_load(band_flag + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
          if (++it > (1 << 20)) {  // ~0.1 s: cannot happen with every band resident; reported, never silent
            *host_err = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
    for (int g = tid; g < 2 * kHalo * groups_per_row; g += kDetThreads) {
      const int rr = fast_div(g, magic_gpr), x = (g - rr * groups_per_row) << 2;
      const int r = rr < kHalo ? rr : rr + kBand, y = y0 - kHalo + r;
      uint32_t packed = 0;
      if (y >= 0 && y < h) packed = __hip_atomic_load(gray_words + (size_t)y * groups_per_row + (x >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *reinterpret_cast<uint32_t*>(s_gray + r * P + x) = packed;
    }
  }
  for (int g0 = 0; !HOSTSRC && g0 < ngroups; g0 += kDetThreads * kStageUnroll) {
    uint32_t d[kStageUnroll][3];
    int lds_off[kStageUnroll];
#pragma unroll
    for (int u = 0; u < kStageUnroll; ++u) {
      const int g = g0 + u * kDetThreads + tid;
      lds_off[u] = -1;
      d[u][0] = d[u][1] = d[u][2] = 0;
      if (g < ngroups) {
        const int r = fast_div(g, magic_gpr), x = (g - r * groups_per_row) << 2;
        const int y = y0 - kHalo + r;
        lds_off[u] = r * P + x;
        if (y >= 0 && y < h) {
          const uint8_t* row = img + (size_t)y * pitch;
          if (FROM_BGR) {
            // 4 pixels = 12 bytes = 3 aligned dwords (pitch % 4 == 0, x % 4 == 0); rows are padded to a dword
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row + 3 * x);
            d[u][0] = p[0];
            d[u][1] = p[1];
            d[u][2] = p[2];
          } else {
            d[u][0] = *reinterpret_cast<const uint32_t*>(row + x);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kStageUnroll; ++u) {
      if (lds_off[u] < 0) continue;
      uint32_t packed;
      if (FROM_BGR) {
        const uint32_t d0 = d[u][0], d1 = d[u][1], d2 = d[u][2];
        const uint32_t g0_ = ((d0 & 255) + ((d0 >> 8) & 255) + ((d0 >> 16) & 255)) / 3u;
        const uint32_t g1_ = ((d0 >> 24) + (d1 & 255) + ((d1 >> 8) & 255)) / 3u;
        const uint32_t g2_ = (((d1 >> 16) & 255) + (d1 >> 24) + (d2 & 255)) / 3u;
        const uint32_t g3_ = (((d2 >> 8) & 255) + ((d2 >> 16) & 255) + (d2 >> 24)) / 3u;
        packed = g0_ | (g1_ << 8) | (g2_ << 16) | (g3_ << 24);
      } else {
        packed = d[u][0];
      }
      if (WRITE_GRAY) {
        const int r = fast_div(lds_off[u], magic_w4), x = lds_off[u] - r * P;
        const int y = y0 - kHalo + r;
        if (r >= kHalo && r < kHalo + kBand && y < h) {  // interior rows are written exactly once
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (x + k < w) gray_out[(size_t)y * w + x + k] = (uint8_t)(packed >> (8 * k));
        }
      }
      *reinterpret_cast<uint32_t*>(s_gray + lds_off[u]) = packed;
    }
  }
  __syncthreads();

  // ---- 2. 5x5 box sums of the band rows (separable: horizontal 5-sums of rows y0-2..y0+kBand+1, then vertical)
  if (DO_BOX) {
    for (int i = tid; i < (kBand + 4) * w; i += kDetThreads) {
      const int r = fast_div(i, magic_w), x = i - r * w;  // r = 0 is image row y0 - 2 = tile row kHalo - 2
      int s = 0;
      if (x >= 2 && x < w - 2) {
        const uint8_t* g = s_gray + (r + kHalo - 2) * P + x;
        s = g[-2] + g[-1] + g[0] + g[1] + g[2];
      }
      s_hsum[r * w4 + x] = (uint16_t)s;
    }
    __syncthreads();
    for (int i = tid; i < kBand * w; i += kDetThreads) {
      const int r = fast_div(i, magic_w), x = i - r * w;
      const int y = y0 + r;
      if (y < h) {
        int s = 0;
        if (y >= 2 && y < h - 2 && x >= 2 && x < w - 2) {
          const uint16_t* c = s_hsum + r * w4 + x;  // rows r .. r+4 of hsum are image rows y-2 .. y+2
          s = c[0] + c[w4] + c[2 * w4] + c[3 * w4] + c[4 * w4];
        }
        box_out[(size_t)y * w + x] = (uint16_t)s;
      }
    }
  }

  // ---- 3. FAST-9 scores for rows y0-1 .. y0+kBand into LDS
  for (int i = tid; i < (kBand + 2) * w4; i += kDetThreads) {  // w4 % 64 may be != 0: lanes past w idle below
    const int r = fast_div(i, magic_w4), x = i - r * w4;
    const int y = y0 - 1 + r;
    int score = 0;
    bool cand = false;
    int d[16];
    if (x < w && y >= border && y < h - border && x >= border && x < w - border) {
      const uint8_t* cp = s_gray + (r + kHalo - 1) * P + x;
      const int c = cp[0];
      uint32_t mb = 0, md = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int p = cp[kCircle.dy[k] * P + kCircle.dx[k]];
        d[k] = p - c;
        mb |= (uint32_t)(d[k] > thr) << k;
        md |= (uint32_t)(d[k] < -thr) << k;
      }
      cand = has_run9(mb) || has_run9(md);
    }
    if (__ballot(cand)) {  // wave-uniform: most waves of a natural image skip the exact score
      if (cand) score = fast9_maxmin(d) - 1;
    }
    s_score[r * w4 + x] = (uint8_t)score;
  }
  __syncthreads();

  // ---- 4. 3x3 non-maximum suppression + ordered compaction of the band
  const int npx = kBand * w;
  for (int seg = wave; seg < nseg; seg += kDetThreads / 64) {
    const int p = seg * 64 + lane;
    bool keep = false;
    if (p < npx) {
      const int r = fast_div(p, magic_w), x = p - r * w;
      if (y0 + r < h) {
        const uint8_t* sp = s_score + (r + 1) * w4 + x;
        const int s = sp[0];
        if (s > 0) {  // s > 0 implies border <= x < w - border, so x-1 and x+1 are inside the row
          const int m = max(max(max((int)sp[-1], (int)sp[1]), max((int)sp[-w4 - 1], (int)sp[-w4])),
                            max(max((int)sp[-w4 + 1], (int)sp[w4 - 1]), max((int)sp[w4], (int)sp[w4 + 1])));
          keep = s > m;
        }
      }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_ballot[seg] = bal;
  }
  __syncthreads();
  if (wave == 0) {
    int running = 0;
    for (int base = 0; base < nseg; base += 64) {
      const int c = (base + lane < nseg) ? __popcll(s_ballot[base + lane]) : 0;
      const int incl = wave_incl_scan(c, lane);
      if (base + lane < nseg) s_segoff[base + lane] = running + incl - c;
      running += __shfl(incl, 63);
    }
    if (lane == 0) bandcnt[blockIdx.x] = running;
  }
  __syncthreads();
  for (int seg = wave; seg < nseg; seg += kDetThreads / 64) {
    const unsigned long long bal = s_ballot[seg];
    if ((bal >> lane) & 1ull) {
      const int p = seg * 64 + lane;
      const int r = fast_div(p, magic_w), x = p - r * w;
      const int pos = s_segoff[seg] + __popcll(bal & ((1ull << lane) - 1ull));
      const uint32_t s = s_score[(r + 1) * w4 + x];
      if (pos < band_cap) {
        raw[(size_t)blockIdx.x * band_cap + pos] = ((uint32_t)(y0 + r) << 20) | ((uint32_t)x << 8) | s;
        atomicAdd(&s_bhist[s], 1);  // LDS integer atomic: order-independent
      }
    }
  }
  __syncthreads();
  // What the cap of select_describe_kernel needs, in the form that makes it cheap there: the band's histogram as SUFFIX
  // sums, c[s] = survivors of this band with score >= s -- "how many score above the cut / equal to it" is then two words
  // per band instead of a 256-bin row (240 dependent row reads per select workgroup made a frame over the cap cost twice a
  // frame under it).  One wave, four bins per lane, no further barrier.  The band's plain histogram goes into the frame's
  // with integer atomics (order-independent, so still bit-exact): only the bins that occur, a handful per band.
  if (tid >= 64 && tid < 320) {
    const int c = s_bhist[tid - 64];
    if (c) atomicAdd(&framehist[tid - 64], c);
  }
  if (tid < 64) {
    int v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // lane l holds bins 255 - 4l - k, k = 0..3: descending score order
      v[k] = s_bhist[255 - 4 * lane - k];
      sum += v[k];
    }
    int run = wave_incl_scan(sum, lane) - sum;  // survivors with a score above this lane's bins
    int4 out;
    run += v[0];
    out.w = run;  // bin 255 - 4l
    run += v[1];
    out.z = run;
    run += v[2];
    out.y = run;
    run += v[3];
    out.x = run;  // bin 252 - 4l
    *reinterpret_cast<int4*>(bandhist + (size_t)blockIdx.x * 256 + 252 - 4 * lane) = out;
  }
}

// ------------------------------------------------------------------------------------------------- select + describe
__device__ __forceinline__ int block_sum(int v, int* s_tmp /* >= 4 ints */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_tmp[wave] = v;
  __syncthreads();
  int t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_tmp[i];
  return t;
}

// one wave evaluates the 256 tests of one keypoint: test k = 64*round + lane; a ballot packs 64 bits = 8 bytes
__device__ __forceinline__ void brief_wave(const uint16_t* __restrict__ box, int w, int x, int y,
                                           uint8_t* __restrict__ out32, int lane, uint8_t* __restrict__ mirror32 = nullptr) {
  unsigned long long* out = reinterpret_cast<unsigned long long*>(out32);
  int a[4], b[4];
#pragma unroll
  for (int rnd = 0; rnd < 4; ++rnd) {  // 8 independent gathers in flight
    const int8x4 t = c_brief[rnd * 64 + lane];
    a[rnd] = box[(size_t)(y + t.v[1]) * w + (x + t.v[0])];
    b[rnd] = box[(size_t)(y + t.v[3]) * w + (x + t.v[2])];
  }
  unsigned long long m[4];
#pragma unroll
  for (int rnd = 0; rnd < 4; ++rnd) m[rnd] = __ballot(a[rnd] < b[rnd]);
  if (lane < 4) {
    const unsigned long long v = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
    out[lane] = v;
    if (mirror32) reinterpret_cast<unsigned long long*>(mirror32)[lane] = v;
  }
}

template <bool DO_BRIEF>
__global__ __launch_bounds__(kSelThreads) void select_describe_kernel(const uint32_t* __restrict__ raw, int band_cap,
                                                                       const int* __restrict__ bandcnt,
                                                                       const int* __restrict__ bandhist,
                                                                       const int* __restrict__ framehist,
                                                                       int* __restrict__ framehist_next, int nbands,
                                                                       int max_kp, const uint16_t* __restrict__ box,
                                                                       int w, float* __restrict__ xy,
                                                                       uint8_t* __restrict__ score,
                                                                       uint8_t* __restrict__ desc,
                                                                       int* __restrict__ n_out, uint8_t* __restrict__ host_block,
                                                                       unsigned h_off_score, unsigned h_off_xy,
                                                                       unsigned h_off_desc) {
  // host_block != nullptr: a pinned host block laid out like the device result block; every result word is written there as
  // well (a few tens of KB over PCIe straight from the kernel), so the host entry point needs no copy launch behind this one
  float* const h_xy = host_block ? reinterpret_cast<float*>(host_block + h_off_xy) : nullptr;
  __shared__ int s_tmp[8];
  __shared__ int s_hist[256];
  __shared__ int s_wcnt[kSelThreads / 64];
  __shared__ uint32_t s_sel[kSelThreads];  // (pos << 0) stored separately below
  __shared__ int s_pos[kSelThreads];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int band = blockIdx.x;

  int part_total = 0, part_before = 0;
  for (int b = tid; b < nbands; b += kSelThreads) {
    const int c = min(bandcnt[b], band_cap);
    part_total += c;
    if (b < band) part_before += c;
  }
  const int total = block_sum(part_total, s_tmp);
  const int before = block_sum(part_before, s_tmp);
  const int nb = min(bandcnt[band], band_cap);

  int cut = 0, quota = 0, base = before, eq_before = 0;
  const bool over = total > max_kp;
  if (band == 0) framehist_next[tid] = 0;  // the table the NEXT frame's bands add into (this frame's is framehist)
  if (over) {
    // framehist[s] = survivors of the whole frame with score s; its suffix sums ge[s] = survivors with score >= s are
    // non-increasing in s.  cut = the score at which the cap is reached = the largest s with ge[s] > max_kp (0 if none);
    // quota = how many of score == cut still fit.  Thread t holds bin 255 - t.
    static_assert(kSelThreads == 256, "one thread per score bin");
    const int hv = framehist[255 - tid];
    const int incl = wave_incl_scan(hv, lane);
    if (lane == 63) s_wcnt[wave] = incl;
    __syncthreads();
    int ge = incl;
    for (int k = 0; k < wave; ++k) ge += s_wcnt[k];
    s_hist[255 - tid] = ge;
    __syncthreads();
    const unsigned long long bal = __ballot(ge > max_kp);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int nabove = 0;
    for (int k = 0; k < kSelThreads / 64; ++k) nabove += s_wcnt[k];
    cut = max(nabove - 1, 0);
    const int above = cut < 255 ? s_hist[cut + 1] : 0;   // survivors with score > cut
    quota = max_kp - above;
    __syncthreads();  // s_wcnt is reused by the selection loop below
    // the same two numbers for the bands before this one, from their suffix-sum rows: two words per band
    int gt_part = 0, ge_part = 0;
    for (int b = tid; b < band; b += kSelThreads) {
      ge_part += bandhist[(size_t)b * 256 + cut];
      gt_part += cut < 255 ? bandhist[(size_t)b * 256 + cut + 1] : 0;
    }
    const int gt_before = block_sum(gt_part, s_tmp);
    const int ge_before = block_sum(ge_part, s_tmp);
    eq_before = ge_before - gt_before;
    base = gt_before + min(eq_before, quota);
  }
  if (band == 0 && tid == 0) {
    *n_out = over ? max_kp : total;
    if (host_block) *reinterpret_cast<int*>(host_block) = over ? max_kp : total;
  }

  int running = 0, eq_running = eq_before;
  for (int start = 0; start < nb; start += kSelThreads) {
    const int i = start + tid;
    uint32_t e = 0;
    bool sel = false, is_eq = false;
    if (i < nb) {
      e = raw[(size_t)band * band_cap + i];
      const int s = e & 255u;
      if (!over) sel = true;
      else if (s > cut) sel = true;
      else if (s == cut) is_eq = true;
    }
    // rank among the score == cut survivors (row-major): first `quota` of them are kept
    const unsigned long long beq = __ballot(is_eq);
    if (lane == 0) s_wcnt[wave] = __popcll(beq);
    __syncthreads();
    int eq_pre = 0, eq_tot = 0;
    for (int k = 0; k < kSelThreads / 64; ++k) {
      const int c = s_wcnt[k];
      eq_pre += (k < wave) ? c : 0;
      eq_tot += c;
    }
    if (is_eq) sel = (int)(eq_running + eq_pre + __popcll(beq & ((1ull << lane) - 1ull))) < quota;
    eq_running += eq_tot;
    __syncthreads();
    const unsigned long long bsel = __ballot(sel);
    if (lane == 0) s_wcnt[wave] = __popcll(bsel);
    __syncthreads();
    int pre = 0, tot = 0;
    for (int k = 0; k < kSelThreads / 64; ++k) {
      const int c = s_wcnt[k];
      pre += (k < wave) ? c : 0;
      tot += c;
    }
    const int local = pre + __popcll(bsel & ((1ull << lane) - 1ull));
    if (sel) {
      const int pos = base + running + local;
      const int x = (e >> 8) & 4095u, y = e >> 20;
      xy[2 * (size_t)pos] = (float)x;
      xy[2 * (size_t)pos + 1] = (float)y;
      if (score) score[pos] = (uint8_t)(e & 255u);
      if (host_block) {
        h_xy[2 * (size_t)pos] = (float)x;
        h_xy[2 * (size_t)pos + 1] = (float)y;
        host_block[h_off_score + pos] = (uint8_t)(e & 255u);
      }
      s_sel[local] = e;
      s_pos[local] = pos;
    }
    __syncthreads();
    if (DO_BRIEF) {
      for (int k = wave; k < tot; k += kSelThreads / 64) {
        const uint32_t ee = s_sel[k];
        brief_wave(box, w, (ee >> 8) & 4095u, ee >> 20, desc + (size_t)s_pos[k] * VS_DESC_BYTES, lane,
                   host_block ? host_block + h_off_desc + (size_t)s_pos[k] * VS_DESC_BYTES : nullptr);
      }
    }
    running += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------- stand-alone pieces
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, int pitch, int w, int h,
                                                   uint8_t* __restrict__ gray) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int y = i / w, x = i - y * w;
  const uint8_t* p = bgr + (size_t)y * pitch + 3 * x;
  gray[i] = (uint8_t)(((uint32_t)p[0] + p[1] + p[2]) / 3u);
}

__global__ __launch_bounds__(256) void boxsum_kernel(const uint8_t* __restrict__ gray, int pitch, int w, int h,
                                                     uint16_t* __restrict__ box) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int y = i / w, x = i - y * w;
  int s = 0;
  if (x >= 2 && x < w - 2 && y >= 2 && y < h - 2) {
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) s += gray[(size_t)(y + dy) * pitch + x + dx];
  }
  box[i] = (uint16_t)s;
}

// BRIEF for caller-supplied keypoints: drop the ones whose patch leaves the image, keep input order
__global__ __launch_bounds__(1024) void brief_points_kernel(const float* __restrict__ xy, int n,
                                                             const uint16_t* __restrict__ box, int w, int h,
                                                             uint8_t* __restrict__ desc, int32_t* __restrict__ keep_idx,
                                                             int* __restrict__ n_out) {
  __shared__ int s_wcnt[16];
  __shared__ int s_x[1024], s_y[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int running = 0;
  for (int start = 0; start < n; start += 1024) {
    const int i = start + tid;
    bool ok = false;
    int x = 0, y = 0;
    if (i < n) {
      x = __float2int_rn(xy[2 * (size_t)i]);  // round half to even, as lrintf in the oracle
      y = __float2int_rn(xy[2 * (size_t)i + 1]);
      ok = x >= VS_BRIEF_BORDER && y >= VS_BRIEF_BORDER && x < w - VS_BRIEF_BORDER && y < h - VS_BRIEF_BORDER;
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      const int c = s_wcnt[k];
      pre += (k < wave) ? c : 0;
      tot += c;
    }
    const int local = pre + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok) {
      keep_idx[running + local] = i;
      s_x[local] = x;
      s_y[local] = y;
    }
    __syncthreads();
    for (int k = wave; k < tot; k += 16)
      brief_wave(box, w, s_x[k], s_y[k], desc + (size_t)(running + k) * VS_DESC_BYTES, lane);
    running += tot;
    __syncthreads();
  }
  if (tid == 0) *n_out = running;
}

size_t detect_lds_bytes(int w, bool do_box) {
  const int w4 = (w + 3) & ~3;
  const int nseg = (kBand * w + 63) >> 6;
  size_t b = (size_t)(kBand + 2 * kHalo) * w4 + (size_t)(kBand + 2) * w4;
  if (do_box) b += (size_t)(kBand + 4) * w4 * 2;
  b = (b + 7) & ~(size_t)7;
  return b + (size_t)nseg * 8 + (size_t)nseg * 4 + 16;
}

int band_capacity(int w) { return (kBand / 2 + 1) * (w / 2 + 1); }

int check_image(vs_ctx* ctx, const void* img, int w, int h, int stride, int bpp, const char* fn) {
  if (!ctx) return VS_EINVAL;
  if (!img) return vs_fail(ctx, VS_EINVAL, "%s: image pointer is NULL", fn);
  if (w <= 0 || h <= 0 || w > kMaxDim || h > kMaxDim || stride < bpp * w)
    return vs_fail(ctx, VS_EINVAL, "%s: bad image geometry (1..4096 per side, stride >= row bytes)", fn);
  return VS_OK;
}

// copy a host image into a device buffer whose pitch is a multiple of 4 (+4 bytes of slack so dword reads of the
// last pixels stay inside the allocation)
int upload_image(vs_ctx* ctx, vs_buf* dst, const uint8_t* src, int row_bytes, int h, int stride, int* pitch_out) {
  const int pitch = (row_bytes + 3) & ~3;
  VS_TRY(vs_reserve(ctx, dst, (size_t)pitch * h + 16));
  if (stride == pitch && vs_is_pinned(src)) {  // caller's frame is pinned (vs_host_alloc): DMA it directly
    VS_HIP(ctx, hipMemcpyAsync(dst->p, src, (size_t)pitch * (h - 1) + row_bytes, hipMemcpyHostToDevice, ctx->stream));
    *pitch_out = pitch;
    return VS_OK;
  }
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin_big, (size_t)pitch * h + 16));
  uint8_t* stage = (uint8_t*)ctx->h_pin_big.p;
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer may still feed an earlier copy
  if (stride == pitch) {
    memcpy(stage, src, (size_t)pitch * (h - 1) + row_bytes);
  } else {
    for (int y = 0; y < h; ++y) memcpy(stage + (size_t)y * pitch, src + (size_t)y * stride, row_bytes);
  }
  VS_HIP(ctx, hipMemcpyAsync(dst->p, stage, (size_t)pitch * h, hipMemcpyHostToDevice, ctx->stream));
  *pitch_out = pitch;
  return VS_OK;
}

// the frame's suffix-sum score table: two of them, used alternately -- the bands of frame f add into table f & 1, which the
// select kernel of frame f - 1 zeroed (zero when allocated)
int frame_tables(vs_ctx* ctx, hipStream_t s, int** cur, int** next) {
  if (!ctx->d_framehist.p) {
    VS_TRY(vs_reserve(ctx, &ctx->d_framehist, 2 * 256 * sizeof(int)));
    VS_HIP(ctx, hipMemsetAsync(ctx->d_framehist.p, 0, 2 * 256 * sizeof(int), s));
    VS_HIP(ctx, hipStreamSynchronize(s));  // launches on the context's other streams must see the zeros, too
  }
  ctx->det_parity ^= 1;
  *cur = (int*)ctx->d_framehist.p + 256 * ctx->det_parity;
  *next = (int*)ctx->d_framehist.p + 256 * (ctx->det_parity ^ 1);
  return VS_OK;
}

template <bool FROM_BGR, bool DO_BOX, bool WRITE_GRAY, bool HOSTSRC = false>
int launch_detect(vs_ctx* ctx, hipStream_t s, const uint8_t* d_img, int pitch, int w, int h, int thr, int border,
                  uint8_t* d_gray, uint16_t* d_box, uint32_t* d_raw, int band_cap, int* d_bandcnt, int* d_bandhist,
                  int* d_framehist, unsigned* d_bandflag = nullptr, unsigned seq = 0, int* host_err = nullptr) {
  const int nbands = (h + kBand - 1) / kBand;
  const size_t lds = detect_lds_bytes(w, DO_BOX);
  auto fn = detect_band_kernel<FROM_BGR, DO_BOX, WRITE_GRAY, HOSTSRC>;
  if (lds > 64 * 1024)
    VS_HIP(ctx, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int w4 = (w + 3) & ~3;
  hipLaunchKernelGGL(fn, dim3(nbands), dim3(kDetThreads), lds, s, d_img, pitch, w, h, thr, border, d_gray, d_box, d_raw,
                     band_cap, d_bandcnt, d_bandhist, div_magic(w), div_magic(w4), div_magic(w4 >> 2), d_framehist, d_bandflag, seq, host_err);
  VS_LAUNCH_CHECK(ctx, "detect_band_kernel");
  return VS_OK;
}

// device layout of the result block: [n int32 | pad to 16][score max_kp, padded to 16][xy max_kp*8][desc max_kp*32]
struct result_layout {
  size_t off_score, off_xy, off_desc, total;
};
result_layout make_layout(int max_kp) {
  result_layout L;
  L.off_score = 16;
  L.off_xy = L.off_score + (((size_t)max_kp + 15) & ~(size_t)15);
  L.off_desc = L.off_xy + (size_t)max_kp * 8;
  L.total = L.off_desc + (size_t)max_kp * VS_DESC_BYTES;
  return L;
}

int detect_common(vs_ctx* ctx, bool from_bgr, const uint8_t* host_img, int w, int h, int stride, int thr, int border,
                  int max_kp, bool describe, float* xy, uint8_t* score, uint8_t* desc, int* n_out, const char* fn) {
  VS_TRY(check_image(ctx, host_img, w, h, stride, from_bgr ? 3 : 1, fn));
  if (!xy || !n_out || (describe && !desc)) return vs_fail(ctx, VS_EINVAL, "%s: null output pointer", fn);
  if (thr < 1 || thr > 254 || border < 3 || max_kp < 0)
    return vs_fail(ctx, VS_EINVAL, "%s: need 1 <= thr <= 254, border >= 3, max_kp >= 0", fn);
  VS_HIP(ctx, hipSetDevice(ctx->device));
  *n_out = 0;
  const int nbands = (h + kBand - 1) / kBand;
  const int band_cap = band_capacity(w);
  const int cap = max_kp > 0 ? max_kp : 1;
  const result_layout L = make_layout(cap);
  int pitch = 0;
  static const bool kTiming = getenv("VS_DET_TIMING") != nullptr;  // developer aid: host phases on stderr
  const auto t_begin = std::chrono::steady_clock::now();
  // A pinned BGR frame (vs_host_alloc) is read where it lies by the detection kernel itself (detect_band_kernel<.., HOSTSRC>):
  // no copy in front of the kernels, and the transfer overlaps the detection.  Needs every band resident at the same time.
  static const bool kNoZeroCopy = getenv("VS_DET_NO_ZEROCOPY") != nullptr;  // developer aid (A/B)
  bool zero_copy = false;
  void* dev_view = nullptr;
  if (from_bgr && !kNoZeroCopy && !ctx->det_copy_only && describe && (w & 3) == 0 && stride == 3 * w && ((uintptr_t)host_img & 3) == 0 && nbands <= ctx->prop.multiProcessorCount &&
      kHalo == 2 * kBand && vs_is_pinned(host_img)) {
    if (hipHostGetDevicePointer(&dev_view, const_cast<uint8_t*>(host_img), 0) == hipSuccess && dev_view) {
      VS_TRY(vs_reserve(ctx, &ctx->d_gray, (size_t)w * h + 16));
      if (ctx->d_bandflag.cap < sizeof(unsigned) * (size_t)nbands) {
        VS_TRY(vs_reserve(ctx, &ctx->d_bandflag, sizeof(unsigned) * (size_t)nbands));
        // the whole capacity, not this frame's bands: vs_reserve over-allocates, and a later frame with more bands that still
        // fits must not meet whatever hipMalloc returned (a word that compares >= seq lets a neighbour skip its wait)
        VS_HIP(ctx, hipMemsetAsync(ctx->d_bandflag.p, 0, ctx->d_bandflag.cap, ctx->stream));
        ctx->det_seq = 0;
      }
      zero_copy = true;
      pitch = 3 * w;
    } else {
      (void)hipGetLastError();
    }
  }
  if (!zero_copy) VS_TRY(upload_image(ctx, from_bgr ? &ctx->d_bgr : &ctx->d_gray, host_img, (from_bgr ? 3 : 1) * w, h, stride, &pitch));
  VS_TRY(vs_reserve(ctx, &ctx->d_box, sizeof(uint16_t) * (size_t)w * h));
  VS_TRY(vs_reserve(ctx, &ctx->d_raw, sizeof(uint32_t) * (size_t)nbands * band_cap));
  VS_TRY(vs_reserve(ctx, &ctx->d_bandcnt, sizeof(int) * (size_t)nbands));
  VS_TRY(vs_reserve(ctx, &ctx->d_hist, sizeof(int) * 256 * (size_t)nbands));
  VS_TRY(vs_reserve(ctx, &ctx->d_xy, L.total));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, L.total));
  uint8_t* res = (uint8_t*)ctx->d_xy.p;
  const uint8_t* d_img = (const uint8_t*)(from_bgr ? ctx->d_bgr.p : ctx->d_gray.p);
  hipStream_t s = ctx->stream;
  int *fh = nullptr, *fh_next = nullptr;
  VS_TRY(frame_tables(ctx, s, &fh, &fh_next));
  // descriptors are written straight into the least recently used slot of the descriptor cache (when they fit one)
  vs_desc_entry* slot = nullptr;
  uint8_t* d_desc_out = res + L.off_desc;
  if (describe) {
    VS_TRY(vs_desc_slot_for_output(ctx, (size_t)cap * VS_DESC_BYTES, &slot));
    if (slot) d_desc_out = (uint8_t*)slot->dev.p;
  }
  if (zero_copy) {
    if (++ctx->det_seq == 0x7FFFFFFFu) {  // the flags only ever grow: start over long before the comparison could wrap
      VS_HIP(ctx, hipMemsetAsync(ctx->d_bandflag.p, 0, ctx->d_bandflag.cap, s));  // every flag the buffer holds (see above)
      ctx->det_seq = 1;
    }
    int* host_err = reinterpret_cast<int*>(ctx->h_pin.p) + 1;  // second word of the pinned result block's header
    *host_err = 0;
    VS_TRY((launch_detect<true, true, false, true>(ctx, s, (const uint8_t*)dev_view, pitch, w, h, thr, border, (uint8_t*)ctx->d_gray.p, (uint16_t*)ctx->d_box.p,
                                                   (uint32_t*)ctx->d_raw.p, band_cap, (int*)ctx->d_bandcnt.p, (int*)ctx->d_hist.p, fh,
                                                   (unsigned*)ctx->d_bandflag.p, ctx->det_seq, host_err)));
  } else if (from_bgr) {
    VS_TRY((launch_detect<true, true, false>(ctx, s, d_img, pitch, w, h, thr, border, nullptr, (uint16_t*)ctx->d_box.p,
                                             (uint32_t*)ctx->d_raw.p, band_cap, (int*)ctx->d_bandcnt.p, (int*)ctx->d_hist.p, fh)));
  } else if (describe) {
    VS_TRY((launch_detect<false, true, false>(ctx, s, d_img, pitch, w, h, thr, border, nullptr, (uint16_t*)ctx->d_box.p,
                                              (uint32_t*)ctx->d_raw.p, band_cap, (int*)ctx->d_bandcnt.p, (int*)ctx->d_hist.p, fh)));
  } else {
    VS_TRY((launch_detect<false, false, false>(ctx, s, d_img, pitch, w, h, thr, border, nullptr, nullptr,
                                               (uint32_t*)ctx->d_raw.p, band_cap, (int*)ctx->d_bandcnt.p, (int*)ctx->d_hist.p, fh)));
  }
  if (describe) {
    hipLaunchKernelGGL(select_describe_kernel<true>, dim3(nbands), dim3(kSelThreads), 0, s, (const uint32_t*)ctx->d_raw.p,
                       band_cap, (const int*)ctx->d_bandcnt.p, (const int*)ctx->d_hist.p, (const int*)fh, fh_next, nbands, max_kp,
                       (const uint16_t*)ctx->d_box.p, w, (float*)(res + L.off_xy), res + L.off_score, d_desc_out, (int*)res,
                       (uint8_t*)ctx->h_pin.p, (unsigned)L.off_score, (unsigned)L.off_xy, (unsigned)L.off_desc);
  } else {
    hipLaunchKernelGGL(select_describe_kernel<false>, dim3(nbands), dim3(kSelThreads), 0, s,
                       (const uint32_t*)ctx->d_raw.p, band_cap, (const int*)ctx->d_bandcnt.p, (const int*)ctx->d_hist.p, (const int*)fh,
                       fh_next, nbands, max_kp, (const uint16_t*)nullptr, w, (float*)(res + L.off_xy), res + L.off_score,
                       (uint8_t*)nullptr, (int*)res, (uint8_t*)ctx->h_pin.p, (unsigned)L.off_score, (unsigned)L.off_xy,
                       (unsigned)L.off_desc);
  }
  VS_LAUNCH_CHECK(ctx, "select_describe_kernel");
  const auto t_enq = std::chrono::steady_clock::now();
  VS_HIP(ctx, hipStreamSynchronize(s));  // the results are in the pinned block already (written by the kernel itself)
  const auto t_sync = std::chrono::steady_clock::now();
  const uint8_t* hp = (const uint8_t*)ctx->h_pin.p;
  const int n = *(const int*)hp;
  if (zero_copy && reinterpret_cast<const int*>(hp)[1] != 0) {
    // a band gave up waiting for its neighbours' rows (bounded wait; the device must have been shared with long-running foreign
    // work): this frame is done again through the copy, and so is every later one of this context
    ctx->det_copy_only = true;
    return detect_common(ctx, true, host_img, w, h, stride, thr, border, max_kp, describe, xy, score, desc, n_out, fn);
  }
  if (n < 0 || n > cap) return vs_fail(ctx, VS_EHIP, "%s: device returned an impossible keypoint count", fn);
  memcpy(xy, hp + L.off_xy, (size_t)n * 8);
  if (score) memcpy(score, hp + L.off_score, (size_t)n);
  if (describe) {
    memcpy(desc, hp + L.off_desc, (size_t)n * VS_DESC_BYTES);
    // the matcher will be handed `desc` next: the kernel wrote the descriptors into a descriptor-cache slot, which now
    // becomes the resident copy of `desc` -- no upload later, no device copy now
    if (slot) vs_desc_adopt_slot(ctx, slot, desc, n, hp + L.off_desc);
  }
  *n_out = n;
  if (kTiming) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "%s: enqueue %.1f us (%s), wait %.1f us, copy out %.1f us\n", fn, us(t_begin, t_enq), zero_copy ? "frame read where it lies" : "frame copied first",
            us(t_enq, t_sync), us(t_sync, std::chrono::steady_clock::now()));
  }
  return VS_OK;
}

}  // namespace

size_t vs_detect_chain_scratch_bytes() {
  size_t worst = 0;
  for (const void* fn : {(const void*)detect_band_kernel<true, true, false>, (const void*)select_describe_kernel<true>}) {
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, fn) != hipSuccess) {
      (void)hipGetLastError();
      return (size_t)-1;
    }
    if (a.localSizeBytes > worst) worst = a.localSizeBytes;
  }
  return worst;
}

VS_API int vs_gray_mean3_u8(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray) {
  VS_TRY(check_image(ctx, bgr, w, h, stride, 3, "vs_gray_mean3_u8"));
  if (!gray) return vs_fail(ctx, VS_EINVAL, "%s: gray is NULL", "vs_gray_mean3_u8");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  int pitch = 0;
  VS_TRY(upload_image(ctx, &ctx->d_bgr, bgr, 3 * w, h, stride, &pitch));
  VS_TRY(vs_reserve(ctx, &ctx->d_gray, (size_t)w * h + 16));
  hipLaunchKernelGGL(gray_kernel, dim3((w * h + 255) / 256), dim3(256), 0, ctx->stream, (const uint8_t*)ctx->d_bgr.p,
                     pitch, w, h, (uint8_t*)ctx->d_gray.p);
  VS_LAUNCH_CHECK(ctx, "gray_kernel");
  VS_HIP(ctx, hipMemcpyAsync(gray, ctx->d_gray.p, (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream));
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}

VS_API int vs_fast9_detect(vs_ctx* ctx, const uint8_t* gray, int w, int h, int stride, int thr, int border, int max_kp,
                           float* xy, uint8_t* score, int* n_out) {
  return detect_common(ctx, false, gray, w, h, stride, thr, border, max_kp, false, xy, score, nullptr, n_out,
                       "vs_fast9_detect");
}

VS_API int vs_detect_describe_bgr(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int thr, int max_kp,
                                  float* xy, uint8_t* score, uint8_t* desc, int* n_out) {
  return detect_common(ctx, true, bgr, w, h, stride, thr, VS_BRIEF_BORDER, max_kp, true, xy, score, desc, n_out,
                       "vs_detect_describe_bgr");
}

int vs_detect_describe_dev_mirror(vs_ctx* ctx, const void* d_bgr, int w, int h, int pitch, int thr, int max_kp, void* d_xy,
                                  void* d_score, void* d_desc, void* d_n_out, void* stream, uint8_t* h_block,
                                  unsigned h_off_score, unsigned h_off_xy, unsigned h_off_desc) {
  VS_TRY(check_image(ctx, d_bgr, w, h, pitch, 3, "vs_detect_describe_bgr_dev"));
  if ((pitch & 3) || ((uintptr_t)d_bgr & 3) || ((uintptr_t)d_desc & 7))
    return vs_fail(ctx, VS_EINVAL, "%s: pitch and base must be multiples of 4 bytes, desc of 8", "vs_detect_describe_bgr_dev");
  if (!d_xy || !d_desc || !d_n_out || thr < 1 || thr > 254 || max_kp < 0)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_detect_describe_bgr_dev");
  hipStream_t s = vs_pick_stream(ctx, stream);
  const int nbands = (h + kBand - 1) / kBand;
  const int band_cap = band_capacity(w);
  VS_TRY(vs_reserve(ctx, &ctx->d_box, sizeof(uint16_t) * (size_t)w * h));
  VS_TRY(vs_reserve(ctx, &ctx->d_raw, sizeof(uint32_t) * (size_t)nbands * band_cap));
  VS_TRY(vs_reserve(ctx, &ctx->d_bandcnt, sizeof(int) * (size_t)nbands));
  VS_TRY(vs_reserve(ctx, &ctx->d_hist, sizeof(int) * 256 * (size_t)nbands));
  int *fh = nullptr, *fh_next = nullptr;
  VS_TRY(frame_tables(ctx, s, &fh, &fh_next));
  VS_TRY((launch_detect<true, true, false>(ctx, s, (const uint8_t*)d_bgr, pitch, w, h, thr, VS_BRIEF_BORDER, nullptr,
                                           (uint16_t*)ctx->d_box.p, (uint32_t*)ctx->d_raw.p, band_cap,
                                           (int*)ctx->d_bandcnt.p, (int*)ctx->d_hist.p, fh)));
  hipLaunchKernelGGL(select_describe_kernel<true>, dim3(nbands), dim3(kSelThreads), 0, s, (const uint32_t*)ctx->d_raw.p,
                     band_cap, (const int*)ctx->d_bandcnt.p, (const int*)ctx->d_hist.p, (const int*)fh, fh_next, nbands, max_kp,
                     (const uint16_t*)ctx->d_box.p, w, (float*)d_xy, (uint8_t*)d_score, (uint8_t*)d_desc, (int*)d_n_out,
                     h_block, h_off_score, h_off_xy, h_off_desc);
  VS_LAUNCH_CHECK(ctx, "select_describe_kernel");
  return VS_OK;
}

VS_API int vs_detect_describe_bgr_dev(vs_ctx* ctx, const void* d_bgr, int w, int h, int pitch, int thr, int max_kp,
                                      void* d_xy, void* d_score, void* d_desc, void* d_n_out, void* stream) {
  return vs_detect_describe_dev_mirror(ctx, d_bgr, w, h, pitch, thr, max_kp, d_xy, d_score, d_desc, d_n_out, stream, nullptr, 0u,
                                       0u, 0u);
}

VS_API int vs_brief256(vs_ctx* ctx, const uint8_t* gray, int w, int h, int stride, const float* xy, int n,
                       uint8_t* desc, int32_t* keep_idx, int* n_out) {
  VS_TRY(check_image(ctx, gray, w, h, stride, 1, "vs_brief256"));
  if (!n_out || n < 0 || (n > 0 && (!xy || !desc)))
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_brief256");
  *n_out = 0;
  if (n == 0) return VS_OK;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  int pitch = 0;
  VS_TRY(upload_image(ctx, &ctx->d_gray, gray, w, h, stride, &pitch));
  VS_TRY(vs_reserve(ctx, &ctx->d_box, sizeof(uint16_t) * (size_t)w * h));
  VS_TRY(vs_reserve(ctx, &ctx->d_xy_in, sizeof(float) * 2 * (size_t)n));
  VS_TRY(vs_reserve(ctx, &ctx->d_desc, (size_t)VS_DESC_BYTES * n));
  VS_TRY(vs_reserve(ctx, &ctx->d_keep, sizeof(int32_t) * (size_t)n));
  VS_TRY(vs_reserve(ctx, &ctx->d_n, 16));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_pin, 64));
  hipStream_t s = ctx->stream;
  VS_HIP(ctx, hipMemcpyAsync(ctx->d_xy_in.p, xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(boxsum_kernel, dim3((w * h + 255) / 256), dim3(256), 0, s, (const uint8_t*)ctx->d_gray.p, pitch, w,
                     h, (uint16_t*)ctx->d_box.p);
  VS_LAUNCH_CHECK(ctx, "boxsum_kernel");
  hipLaunchKernelGGL(brief_points_kernel, dim3(1), dim3(1024), 0, s, (const float*)ctx->d_xy_in.p, n,
                     (const uint16_t*)ctx->d_box.p, w, h, (uint8_t*)ctx->d_desc.p, (int32_t*)ctx->d_keep.p,
                     (int*)ctx->d_n.p);
  VS_LAUNCH_CHECK(ctx, "brief_points_kernel");
  VS_HIP(ctx, hipMemcpyAsync(ctx->h_pin.p, ctx->d_n.p, sizeof(int), hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  const int m = *(const int*)ctx->h_pin.p;
  if (m < 0 || m > n) return vs_fail(ctx, VS_EHIP, "%s: device returned an impossible keypoint count", "vs_brief256");
  if (m > 0) {
    VS_HIP(ctx, hipMemcpyAsync(desc, ctx->d_desc.p, (size_t)VS_DESC_BYTES * m, hipMemcpyDeviceToHost, s));
    if (keep_idx) VS_HIP(ctx, hipMemcpyAsync(keep_idx, ctx->d_keep.p, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  *n_out = m;
  return VS_OK;
}
