// vs_internal.h -- private helpers shared by the translation units of libvslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/vslam_hip_dev.h"  // includes vslam_hip.h; the compiler checks both sets of prototypes

#define VS_API extern "C" __attribute__((visibility("default")))

// growable device / pinned-host scratch buffer owned by the context
struct vs_buf {
  void* p = nullptr;
  size_t cap = 0;
};

// Device-resident copies of descriptor sets the host API has seen: keyed by host pointer + row count and verified
// byte-for-byte against a pinned host shadow of what was uploaded (memcmp -- no hash, so a re-used or modified host
// buffer can never alias a stale copy).  A frame's descriptors are uploaded once although they are matched many times
// (as train this frame, as the key frame's query set for every following frame), and the descriptors
// vs_detect_describe_bgr just produced are never uploaded at all.
struct vs_desc_entry {
  const void* host = nullptr;
  int n = 0;
  vs_buf dev;
  vs_buf shadow;  // pinned host copy of the uploaded bytes (also the staging buffer of the upload)
  uint64_t stamp = 0;
};
constexpr int VS_DESC_CACHE = 6;
constexpr size_t VS_DESC_CACHE_MAX_BYTES = 8u << 20;  // larger sets are uploaded on every call

// scratch of the match kernel (per-chunk partial rows + per-tile arrival tickets), one set per stream it is launched on:
// launches on different streams may overlap on the GPU, launches on one stream never do
struct vs_match_scratch {
  hipStream_t stream = nullptr;
  bool used = false;
  uint64_t stamp = 0;  // last use (least recently used set is recycled for a new stream)
  vs_buf partial;      // [nchunks][qtiles][256] self-validating partial words (epoch | second key | best key)
  vs_buf flag;         // pinned: set by a folding workgroup whose wait ran out (checked by the next launch on the stream)
  int qtiles = 0, nchunks = 0, epoch = 0;  // geometry the slots were last used with; epoch of the newest launch (1 .. 63)
  vs_buf idx, dist;    // 2-NN rows between the match kernel and the ratio kernel (vs_match_ratio_dev)
};
constexpr int VS_MATCH_STREAMS = 4;

// Test / sweep hooks.  They live in the context (one per process and GPU), not in process-wide globals: two contexts, or a
// test that flips a knob, never change what another context computes.  Set through vs_tune_* (not part of the stable ABI).
struct vs_tuning {
  int match_target_blocks = 0;  // 0: automatic plan (plan_chunks); > 0: fixed number of workgroups
  int match_tstage = 1;         // train rows staged through LDS into VGPRs (1) or fed from SGPRs (0)
  bool match_profile = false;   // HIP events around every match launch (bench.py)
  int ba_host_structure = 0;    // 1: large problems build their structure with the host passes, never on the device (vs_tune_ba_structure)
  int schur_variant = 0;        // single-tile windows: 0 ba_schur_small + speculative linearisation, 1 tile kernel, 2 ba_schur_small + linearise launch; 3: as 0, and banded windows of several tiles on ba_schur_tile instead of ba_schur_window
  int small_per = 0;            // points per ba_schur_small workgroup (0: kSmallPts)
  int small_ns_cap = 512;       // cap on its slab count
  int win_per = 0;              // banded large windows: points per ba_schur_window workgroup (0: automatic)
  int motion_variant = 0;       // 0: one-launch motion-only solve where it applies, 1: one launch per LM step
  int ba_graph = 0;             // experiment: the LM slot batches of vs_ba_solve replayed as captured hipGraphs (vs_tune_ba_graph)
  int ba_solve_packed = 0;      // dense solve in LDS (vs_tune_ba_solve): 0 automatic (packed triangle from 127 to 198 unknowns), 1 never packed, 2 packed at every size it fits
  int poison_alloc = -1;        // >= 0: every NEW device buffer of vs_reserve is filled with this byte (vs_debug_poison_alloc; tests: nothing may rely on what hipMalloc returns)
};
struct vs_prof_rec {
  hipEvent_t e0, e1;
};

constexpr int VS_AUX_STREAMS = 2;

// Host worker threads of a context (the structure passes of large bundle adjustments): created on first use, parked on a
// condition variable between uses -- starting a dozen std::threads per pass cost 0.3-0.4 ms each time.  run(nt, fn, arg) calls
// fn(arg, t, nt) for t = 0 .. nt-1, t = 0 on the caller, and returns when all have finished.  One run at a time.
struct vs_pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  void (*fn)(void*, int, int) = nullptr;
  void* arg = nullptr;
  int nt = 0, pending = 0;
  unsigned gen = 0;
  bool stop = false;
  void run(int n, void (*f)(void*, int, int), void* a);
  void shutdown();
  ~vs_pool() { shutdown(); }
};

struct vs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // Further streams of the context, created TOGETHER with `stream` in vs_create: the runtime spreads streams over a few
  // hardware queues as they are created, and two streams that end up on one queue never overlap.  Measured (round 3,
  // tools/stream_pressure.py): with the front-half stream created lazily, after the process had made other streams (one
  // torch.cuda.Stream() creates torch's whole pool), the pipelined tracking period ran at 210 us per frame instead of 139.
  hipStream_t aux_stream[VS_AUX_STREAMS] = {nullptr, nullptr};  // compute streams for callers that keep steps in flight (vs_aux_stream)
  hipDeviceProp_t prop;
  char err[512];
  // matcher
  vs_buf d_q, d_t, d_mq, d_mt, d_md, d_cnt;
  vs_match_scratch match_scratch[VS_MATCH_STREAMS];
  // detector
  vs_buf d_bgr, d_gray, d_box, d_raw, d_bandcnt, d_hist, d_xy, d_score, d_desc, d_n, d_xy_in, d_keep;
  vs_buf d_framehist;  // two 256-entry suffix-sum score tables of whole frames, used alternately (vs_detect.hip)
  int det_parity = 0;
  vs_buf d_bandflag;       // per band: sequence number of the newest frame whose gray rows of that band are in d_gray (frames read where they lie)
  unsigned det_seq = 0;
  bool det_copy_only = false;  // a band once gave up waiting for its neighbours' rows: this context copies its frames from then on
  // bundle adjustment
  vs_buf d_ba;      // one arena, carved per solve
  vs_buf h_pin;     // pinned staging (small read-backs)
  vs_buf h_pin_big; // pinned staging (frames, descriptors)
  vs_desc_entry desc_cache[VS_DESC_CACHE];
  uint64_t desc_stamp = 0;
  // tracking session (vs_track_*): one key-frame period resident on the device
  vs_buf d_track, h_track, h_api;  // h_api: pinned block the class-API entry points' kernels mirror their results into
  struct {
    int active = 0, n_points = 0, cap_frames = 0, max_kp = 0, pnp_iters = 0;
    int n_frames = 0;   // frames tracked so far in this period (pose index of the newest one)
    int obs_used = 0;   // observations appended so far
    int cur = 0;        // state buffer holding the accepted estimate
    double K[4] = {0, 0, 0, 0};
    double last_rec[19];  // camera record of the newest pose (the PnP guess of the next frame)
    int lm_steps_hint = 0;  // motion-only LM launches the previous solve of this period needed (0: unknown)
    int solve_epoch = 0;    // tag of the newest one-launch motion-only solve (ba_motion_persistent's mailboxes)
    // pipelined use (vs_track_frame_pipelined): the front half (upload, detect, match) of frame k+1 runs on its own
    // stream while the back half (PnP, BA) of frame k runs on the context's stream; two sets of per-frame buffers
    hipStream_t front_stream = nullptr;
    hipEvent_t ev_front[2] = {nullptr, nullptr};
    int pending_published = 0;  // ... and its read-back block is on its way to the host (a kernel that publishes it has been enqueued)
    int pending_chained = 0;  // the pending back half was enqueued with its inputs read on the device (see vs_track_frame_pipelined)
    unsigned front_seq = 0, front_tag[2] = {0, 0};  // tags of the front halves (track_append_kernel publishes them on the device)
    unsigned api_seq = 0;        // tags of the class-API entry points' pinned results (front half / PnP outcome)
    unsigned back_tag_sync = 0;  // ... and of the host-paced back half's block (pinned block 0)
    unsigned back_seq = 0, back_tag[2] = {0, 0};    // tags of the chained back halves' read-back blocks (track_publish_kernel)
    unsigned ba_seq = 0, ba_tag[2] = {0, 0};        // tags the chained back halves' motion-only solves publish on the device when they are through
    int quiet = 1;     // no tracking work can be outstanding on any of the context's streams (set where results were handed out last)
    int mst_both = 0;  // both LM records on the device name the state buffer of the newest estimate (fresh period, or the last solve ran in one launch)
    int pending = -1;   // buffer set of the frame whose front half is done and whose back half is not, or -1
    int pending_step = -1;  // >= 0: that frame's back half is enqueued already, this many LM launches so far
    int next_set = 0;
    // class-API feeding (vs_track_front / vs_track_back_begin / vs_track_back_end): 0 idle, 1 front half done, 2 back half running
    int api_stage = 0, api_step = 0, api_matches = 0;
    int api_back = 0;  // the back half being enqueued (or redone) is vs_track_back_begin's
    int api_obj_f32 = 0, api_guess_set = 0;  // vs_track_back_begin: object points as float32 / an explicit PnP guess (else the previous pose)
    double api_guess_rec[19];
    hipEvent_t ev_api = nullptr;  // recorded behind the PnP kernel of vs_track_back_begin
    struct {
      double reproj_err, confidence, huber;
      unsigned long long seed;
      int lm_iterations;
    } params[2];
    // recovery (vs_track.hip, track_redo): a back half whose in-kernel hand-offs did not complete is redone host-paced once
    std::vector<uint8_t> good;  // read-back block [LM records | flags | PnP result | both camera buffers] of the newest frame handed out
    double key_rec[19];         // the key frame's camera record (what both camera buffers start from)
    double redo_rec[19];        // start record of a host-fed frame (vs_track_push_frame), kept for a redo
    int recoveries = 0;         // back halves redone so far (vs_track_debug)
    int recoverable = 0;        // set by track_back_finish next to an error a redo can cure
    int in_redo = 0;            // a redo is running: the motion-only solve takes the launch-per-step form, nothing is chained
    int last_set = -1, last_nkp = 0, last_M = 0;  // buffer set / counts of the newest frame handed out (vs_track_last_frame)
    int inject = 0;             // developer aid (vs_track_debug): the next chained PnP launch waits for a tag nobody publishes
    int dirty = 0;              // an entry point returned an error after enqueueing: vs_track_end / vs_track_begin synchronise fully
  } track;
  vs_buf d_bgr2;  // image buffer of the second set
  hipEvent_t ev_shard = nullptr;  // orders the all-gather stream behind the match kernel (vs_hamming_knn2_sharded_dev)
  hipEvent_t ev_after = nullptr;  // orders a compute stream behind the caller's stream (same entry point, after_stream)
  vs_tuning tune;
  std::vector<vs_prof_rec> match_prof;
  vs_buf d_mo_stamps;       // diagnostic step stamps of the newest one-launch motion-only solve of a tracking period (vs_mo_profile)
  bool mo_profile = false;
  bool ba_aux_copy_pending = false;  // a copy into the BA arena is in flight on aux_stream[0] and the main stream has not been ordered behind it yet
  double ba_batch_us = 0;    // wall time of the newest LM slot batch, first enqueue (or graph launch) to results on the host
  int poison_count = 0;      // device allocations made while vs_debug_poison_alloc is on
  hipStream_t poison_stream = nullptr;  // non-blocking stream of the poison fills (created on first use)
  int ba_path[6] = {-1, -1, 0, 0, 0, 0};  // kernels of the newest vs_ba_solve (vs_ba_last_path)
  int ba_structure_dev = 0;  // 1: the newest vs_ba_solve built its structure on the device (vs_ba_structure_on_device)
  vs_buf d_match_stamps;    // diagnostic phase stamps of the newest stamped match launch (vs_match_stamps)
  bool match_stamps_on = false;
  int match_stamps_rows = 0;
  vs_buf d_pnp_tag;         // tagged per-hypothesis words of pnp_ransac_kernel (zero when allocated)
  unsigned pnp_epoch = 0;   // epoch of the newest PnP call
  vs_buf d_pnp_stamps;      // diagnostic phase stamps of the newest PnP launch (vs_pnp_profile)
  bool pnp_profile = false;
  int pnp_profile_h = 0;
  vs_pool pool;
  int mo_persist_cap = -1;  // camera workgroups of ba_motion_persistent the device keeps resident together (-1: not asked yet)
  long long chain_scratch = -1;  // largest private segment among the kernels of a chained tracking period (-1: not asked yet; vs_track.hip)
};

// device copy of the n x 32-byte descriptor set at host pointer `h` (uploads unless the very same bytes are already
// resident); role 0 = query, 1 = train (buffers of the uncached path); the returned pointer stays valid until
// VS_DESC_CACHE other sets have been used
int vs_desc_resident(vs_ctx* ctx, const uint8_t* h, int n, int role, const void** dev_out);
// registers a device-side descriptor set that equals the host array `h`; host_src = the bytes just copied into `h`
int vs_desc_adopt(vs_ctx* ctx, const uint8_t* h, int n, const void* dev_src, const uint8_t* host_src);
// a cache slot (the least recently used one) whose device buffer holds `bytes`, for a producer kernel to write a descriptor
// set into; *slot = nullptr if the set is too large to be cached.  vs_desc_adopt_slot then binds it to the host array.
int vs_desc_slot_for_output(vs_ctx* ctx, size_t bytes, vs_desc_entry** slot);
void vs_desc_adopt_slot(vs_ctx* ctx, vs_desc_entry* slot, const uint8_t* h, int n, const uint8_t* host_src);
// true if `p` is pinned (hipHostMalloc / hipHostRegister) host memory
bool vs_is_pinned(const void* p);

extern char g_vs_create_error[512];

static inline int vs_fail(vs_ctx* ctx, int code, const char* fmt, const char* a = "", const char* b = "") {
  char* dst = ctx ? ctx->err : g_vs_create_error;
  snprintf(dst, 512, fmt, a, b);
  return code;
}

#define VS_HIP(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return vs_fail(ctx, VS_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define VS_LAUNCH_CHECK(ctx, name)                                                                   \
  do {                                                                                               \
    hipError_t e_ = hipGetLastError();                                                               \
    if (e_ != hipSuccess) return vs_fail(ctx, VS_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

// grow-only device buffer
static inline int vs_reserve(vs_ctx* ctx, vs_buf* b, size_t bytes) {
  if (bytes <= b->cap && b->p) return VS_OK;
  if (bytes < 256) bytes = 256;
  size_t want = bytes + bytes / 4;  // headroom so a growing map does not reallocate every frame
  if (b->p) {
    VS_HIP(ctx, hipDeviceSynchronize());  // the buffer may be in use on any stream (front half, match streams of a plan)
    VS_HIP(ctx, hipFree(b->p));
    b->p = nullptr;
    b->cap = 0;
  }
  hipError_t e = hipMalloc(&b->p, want);
  if (e != hipSuccess) return vs_fail(ctx, VS_ENOMEM, "hipMalloc(%s) failed: %s", "scratch", hipGetErrorString(e));
  b->cap = want;
  if (ctx->tune.poison_alloc >= 0) {  // developer aid (vs_debug_poison_alloc; VS_POISON_SKIP=n leaves the n-th allocation alone, VS_POISON_LOG=1 lists them)
    static const int kSkip = getenv("VS_POISON_SKIP") ? atoi(getenv("VS_POISON_SKIP")) : -1;
    static const int kUpto = getenv("VS_POISON_UPTO") ? atoi(getenv("VS_POISON_UPTO")) : 1 << 30;  // only the first n allocations
    static const bool kLog = getenv("VS_POISON_LOG") != nullptr;
    const int ordinal = ctx->poison_count++;
    const bool skip = ordinal == kSkip || ordinal >= kUpto;
    if (kLog) fprintf(stderr, "[poison] allocation %d: %zu bytes at %p (vs_buf at +%td)%s\n", ordinal, want, b->p, (char*)b - (char*)ctx, skip ? " SKIPPED" : "");
    if (!skip) {
      // on a non-blocking stream of its own: hipMemset runs on the legacy default stream, which waits for -- and holds back --
      // every blocking stream of the process; with a chained tracking period in flight (kernels that wait in-kernel for work
      // enqueued behind them) that is a deadlock until the bounded waits give up (seen in round 5 as "the device did not
      // publish its results" under VS_TEST_POISON)
      if (!ctx->poison_stream) VS_HIP(ctx, hipStreamCreateWithFlags(&ctx->poison_stream, hipStreamNonBlocking));
      VS_HIP(ctx, hipMemsetAsync(b->p, ctx->tune.poison_alloc & 255, want, ctx->poison_stream));
      VS_HIP(ctx, hipStreamSynchronize(ctx->poison_stream));
    }
  }
  return VS_OK;
}

static inline int vs_reserve_pinned(vs_ctx* ctx, vs_buf* b, size_t bytes) {
  if (bytes <= b->cap && b->p) return VS_OK;
  if (bytes < 4096) bytes = 4096;
  if (b->p) {
    VS_HIP(ctx, hipDeviceSynchronize());
    VS_HIP(ctx, hipHostFree(b->p));
    b->p = nullptr;
    b->cap = 0;
  }
  hipError_t e = hipHostMalloc(&b->p, bytes + bytes / 4, hipHostMallocDefault);
  if (e != hipSuccess) return vs_fail(ctx, VS_ENOMEM, "hipHostMalloc(%s) failed: %s", "staging", hipGetErrorString(e));
  b->cap = bytes + bytes / 4;
  return VS_OK;
}

#define VS_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != VS_OK) return rc_; \
  } while (0)

static inline hipStream_t vs_pick_stream(vs_ctx* ctx, void* s) { return s ? (hipStream_t)s : ctx->stream; }
// vs_match_ratio_dev with the number of train rows read on the device (at most nt of them); vs_match.hip
// h_mirror: optional pinned host block [count (4 ints) | mq | mt | md], h_stride ints apart, written by the kernel as well
int vs_match_ratio_dev_n(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, const int* nt_dev, double ratio,
                         void* d_match_q, void* d_match_t, void* d_match_d, void* d_n_out, void* stream, int32_t* h_mirror,
                         int h_stride);
// vs_detect_describe_bgr_dev with an optional pinned host block [n | pad 16][score][xy][desc] (offsets as given) that the
// select kernel writes as well; vs_detect.hip
int vs_detect_describe_dev_mirror(vs_ctx* ctx, const void* d_bgr, int w, int h, int pitch, int thr, int max_kp, void* d_xy,
                                  void* d_score, void* d_desc, void* d_n_out, void* stream, uint8_t* h_block,
                                  unsigned h_off_score, unsigned h_off_xy, unsigned h_off_desc);

// largest private segment (scratch bytes per lane) among the kernels this unit contributes to a tracking period's front half,
// as the runtime reports it for the loaded code objects ((size_t)-1: could not be asked)
size_t vs_match_chain_scratch_bytes(vs_ctx* ctx);
size_t vs_detect_chain_scratch_bytes();

// VS_EHIP (and the flags cleared) when a match launch on ANY stream of the context gave up its bounded wait since the last check
int vs_match_lost_check(vs_ctx* ctx, const char* who);
// implemented in vs_match.hip / vs_detect.hip / vs_ba.hip
void vs_ctx_free_buffers(vs_ctx* ctx);
