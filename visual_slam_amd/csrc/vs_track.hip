// vs_track.hip -- one tracking period resident on the device (gfx950).
#include "vs_ba_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>

using namespace vsba;

namespace {
// developer aid (VS_TRACK_TIMING=1): host-side phase times of the pipelined entry point, printed by vs_track_end
struct track_timing {
  bool on = getenv("VS_TRACK_TIMING") != nullptr;
  double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long n = 0;
  std::chrono::steady_clock::time_point t;
  void start() {
    if (on) t = std::chrono::steady_clock::now();
  }
  void lap(int i) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    sum[i] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
  }
} g_tt;

// ------------------------------------------------------------------------------------------------ tracking session
// appends the new frame's observations to the period's camera-major arrays: obs i = (map point mq[i], keypoint mt[i])
__global__ __launch_bounds__(256) void track_append_kernel(const double* xyz, const float* fxy, const int* mq, const int* mt,
                                                           const int* d_M, double* mo_X, double* mo_uv, int* cam_start,
                                                           int slot, int cap_obs, int* flags, const int* d_nkp,
                                                           unsigned* front_sync, unsigned front_tag, unsigned* host_tag_word,
                                                           unsigned host_tag) {
  const int base = cam_start[slot];
  int M = *d_M;
  if (base + M > cap_obs) {
    M = max(0, cap_obs - base);
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 1;  // capacity exceeded: the host reports it
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
    const int q = mq[i], t = mt[i];
    mo_X[3 * (size_t)(base + i)] = xyz[3 * (size_t)q];
    mo_X[3 * (size_t)(base + i) + 1] = xyz[3 * (size_t)q + 1];
    mo_X[3 * (size_t)(base + i) + 2] = xyz[3 * (size_t)q + 2];
    mo_uv[2 * (size_t)(base + i)] = (double)fxy[2 * (size_t)t];
    mo_uv[2 * (size_t)(base + i) + 1] = (double)fxy[2 * (size_t)t + 1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cam_start[slot + 1] = base + M;
    flags[1] = M;  // the count PnP and the host see (clamped)
    flags[2] = *d_nkp;  // the frame's key-point count: the host reads it with the results, not in the middle of the frame
  }
  // "this frame's rows are complete": the workgroup that arrives last publishes the frame's tag.  A chained back half
  // (pnp_ransac_kernel on the other stream) waits for that word instead of for an event in front of its launch.
  __threadfence();  // this thread's rows are visible device-wide before its workgroup is counted
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(front_sync, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(front_sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next frame's append counts from zero
      __hip_atomic_store(front_sync + 64, front_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      // class-API period: the front half's kernels mirrored their results into pinned memory (complete: they ended before
      // this launch began); the host polls this word instead of synchronising with the stream
      if (host_tag_word) __hip_atomic_store(host_tag_word, host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// chained back halves: the read-back block goes to pinned host memory from a kernel of ours, followed by the frame's tag
// (system-scope release) -- the host polls the tag.  One short launch instead of a copy command + an event record, both of
// which are barrier packets on the chain of back halves.
__global__ __launch_bounds__(256) void track_publish_kernel(const uint4* src, uint4* dst, int n16, unsigned* tag_word, unsigned tag) {
  for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(tag_word, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// host-fed variant (vs_track_push_frame): obs i = (map point idx[i], image point uv[i]) as the caller matched them
__global__ __launch_bounds__(256) void track_push_kernel(const double* xyz, const int* idx, const double* uv, int m,
                                                         int n_points, double* mo_X, double* mo_uv, int* cam_start, int slot,
                                                         int cap_obs, int* flags) {
  const int base = cam_start[slot];
  int M = m;
  if (base + M > cap_obs) {
    M = max(0, cap_obs - base);
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 1;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
    const int q = min(max(idx[i], 0), n_points - 1);  // validated on the host; clamped here so no access can leave the map
    mo_X[3 * (size_t)(base + i)] = xyz[3 * (size_t)q];
    mo_X[3 * (size_t)(base + i) + 1] = xyz[3 * (size_t)q + 1];
    mo_X[3 * (size_t)(base + i) + 2] = xyz[3 * (size_t)q + 2];
    mo_uv[2 * (size_t)(base + i)] = uv[2 * (size_t)i];
    mo_uv[2 * (size_t)(base + i) + 1] = uv[2 * (size_t)i + 1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cam_start[slot + 1] = base + M;
    flags[1] = M;
    flags[2] = 0;  // no key points on this path
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tracking session
// One key-frame period of the reference's tracking loop (src/v2/main.py:173-214) kept resident on the device: the key
// frame's map points and descriptors are uploaded once (vs_track_begin); every vs_track_frame uploads only the image
// and runs detect+describe -> match against the map -> PnP-RANSAC from the previous pose -> append the observations
// -> motion-only BA over all poses of the period, with ONE host wait, at the end (the matcher is launched for max_kp train
// rows and reads the key-point count on the device; the wait is a poll of a tagged word in pinned memory that the last kernel
// writes behind the results).  Same kernels, same arithmetic as the separate entry points.
namespace {
struct track_front {
  size_t fxy, fscore, fdesc, fn, mq, mt, md, M;
};
struct track_layout {
  size_t xyz, mapdesc;
  track_front f[2];  // two sets of per-frame buffers (the synchronous entry point uses set 0 only)
  size_t flags, cam0, cam1, moX, moUV, cam_start, slot_pose, part, H, box, mst, pnp_cam, pnp_res, pnp_inl,
      push_idx, push_uv, rb_end, front_sync, back_sync, init_end, total;
  int cap_obs;
};

track_layout track_layout_of(int P, int F, int max_kp, int H) {
  track_layout L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  // matches per frame: one per surviving QUERY (= map point), so up to P of them whatever max_kp is
  const int per = P;
  L.cap_obs = F * (per > 0 ? per : 1);
  L.xyz = take(sizeof(double) * 3 * (size_t)P);
  L.mapdesc = take(32 * (size_t)P + 32);
  for (track_front& f : L.f) {
    f.fxy = take(sizeof(float) * 2 * (size_t)max_kp);
    f.fscore = take((size_t)max_kp);
    f.fdesc = take(32 * (size_t)max_kp + 32);
    f.fn = take(sizeof(int));
    f.mq = take(sizeof(int) * (size_t)P);
    f.mt = take(sizeof(int) * (size_t)P);
    f.md = take(sizeof(int) * (size_t)P);
    f.M = take(sizeof(int));
  }
  // read-back block: [LM state x2 | flags | PnP result | both camera buffers] is fetched with one copy per frame
  L.mst = take(2 * sizeof(mo_state));
  L.flags = take(2 * 4 * sizeof(int));  // two sets of {capacity exceeded, matches, key points, -}: one per per-frame buffer set
  L.pnp_res = take(sizeof(double) * 20);
  L.cam0 = take(sizeof(double) * kCamStride * (size_t)(F + 1));
  L.cam1 = take(sizeof(double) * kCamStride * (size_t)(F + 1));
  L.rb_end = off;
  // initialised by vs_track_begin together with the block above: ONE upload of a host-built image [mst .. init_end)
  L.cam_start = take(sizeof(int) * (size_t)(F + 2));
  L.slot_pose = take(sizeof(int) * (size_t)(F + 1));
  L.box = take(sizeof(unsigned long long) * 2 * kMoPersistCameras * 8);  // mailboxes of ba_motion_persistent
  L.front_sync = take(512);  // [0] arrival counter of track_append_kernel, [64] tag of the newest complete front half
  L.back_sync = take(512);   // [0] arrival counter of ba_motion_persistent's workgroups, [64] tag of the newest finished solve
  L.init_end = off;
  L.moX = take(sizeof(double) * 3 * (size_t)L.cap_obs);
  L.moUV = take(sizeof(double) * 2 * (size_t)L.cap_obs);
  L.part = take(sizeof(double) * 8 * (size_t)F);
  L.H = take(sizeof(double) * 42 * (size_t)F);
  L.pnp_cam = take(sizeof(double) * kPnpModel * (size_t)H);
  L.pnp_inl = take(sizeof(int) * (size_t)(per > 0 ? per : 1));
  L.push_idx = take(sizeof(int) * (size_t)(per > 0 ? per : 1));
  L.push_uv = take(sizeof(double) * 2 * (size_t)(per > 0 ? per : 1));
  L.total = off;
  return L;
}

void rec_from_pose(const double* pose16, double* rec) {
  rec[0] = pose16[3];
  rec[1] = pose16[7];
  rec[2] = pose16[11];
  quat_from_pose(pose16, rec + 3);
  quat_to_w2n(rec, rec + 3, rec + 7);
}

void pose_from_rec(const double* c, double* o) {
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) o[4 * r + k] = c[7 + 4 * k + r];
    o[4 * r + 3] = c[r];
  }
  o[12] = o[13] = o[14] = 0.0;
  o[15] = 1.0;
}

track_layout layout_of(const vs_ctx* ctx) {
  const auto& T = ctx->track;
  return track_layout_of(T.n_points, T.cap_frames, T.max_kp, T.pnp_iters > 0 ? T.pnp_iters : 1);
}

constexpr size_t kPinRb = 4096;  // pinned staging: 1024 LM start state, 2048 record, 4096 read-back

// Front half of a frame on stream `s`: image upload, detect+describe, match against the map, append of the matches as the
// observations of free-camera slot `slot` -- enqueue only, no host synchronisation.  Ends with ev_front[set] recorded on `s`.
// pinned block of the class-API entry points (vs_track_front / vs_track_back_begin): what the kernels mirror to the host
struct api_layout {
  size_t det, det_score, det_xy, det_desc, match, match_stride, pnp_res, pnp_inl, tags, total;
};
api_layout api_layout_of(int P, int max_kp) {
  api_layout A;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  A.det = 0;
  A.det_score = 16;
  A.det_xy = A.det_score + (((size_t)max_kp + 15) & ~(size_t)15);
  A.det_desc = A.det_xy + (size_t)max_kp * 8;
  A.match = up(A.det_desc + (size_t)max_kp * VS_DESC_BYTES);
  A.match_stride = ((size_t)P + 3) & ~(size_t)3;  // ints
  A.pnp_res = up(A.match + 16 + 3 * 4 * A.match_stride);
  A.pnp_inl = A.pnp_res + 256;
  A.tags = up(A.pnp_inl + 4 * (size_t)P);  // [0] front half complete, [16] PnP outcome complete (words the kernels write last)
  A.total = A.tags + 256;
  return A;
}

// Waits until a pinned word carries `want` (written last, with a system-scope release, by the kernel whose results precede
// it): a poll of host memory instead of a stream / event synchronisation (~10 us less per wait).  Bounded: after 1.5 s (every
// in-kernel wait has given up long before: ~50 - 100 ms each) the streams are synchronised and the word is looked at once more.
int track_poll(vs_ctx* ctx, const volatile unsigned* word, unsigned want, const char* who) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; __atomic_load_n(word, __ATOMIC_ACQUIRE) != want; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1500)) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipStreamSynchronize(ctx->aux_stream[1]);
      if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) break;
      return vs_fail(ctx, VS_EHIP, "%s: the device did not publish its results", who);
    }
  }
  return VS_OK;
}

int track_front_half(vs_ctx* ctx, int set, int slot, const uint8_t* bgr, int w, int h_img, int stride, int thr, double ratio,
                     hipStream_t s, bool mirror = false) {
  auto& T = ctx->track;
  const track_layout L = layout_of(ctx);
  const track_front& F = L.f[set];
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  vs_buf* img = set ? &ctx->d_bgr2 : &ctx->d_bgr;
  const int pitch = (3 * w + 3) & ~3;  // 3w rounded up to 4 bytes, 16 bytes of slack after the last row
  VS_TRY(vs_reserve(ctx, img, (size_t)pitch * h_img + 16));
  if (stride == pitch) VS_HIP(ctx, hipMemcpyAsync(img->p, bgr, (size_t)pitch * h_img, hipMemcpyHostToDevice, s));
  else VS_HIP(ctx, hipMemcpy2DAsync(img->p, pitch, bgr, stride, 3 * (size_t)w, h_img, hipMemcpyHostToDevice, s));
  const api_layout AL = api_layout_of(T.n_points, T.max_kp);
  uint8_t* hb = mirror ? (uint8_t*)ctx->h_api.p : nullptr;
  VS_TRY(vs_detect_describe_dev_mirror(ctx, img->p, w, h_img, pitch, thr, T.max_kp, d + F.fxy, d + F.fscore, d + F.fdesc, d + F.fn, s,
                                       hb, (unsigned)AL.det_score, (unsigned)AL.det_xy, (unsigned)AL.det_desc));
  // the matcher is launched for max_kp train rows at most and reads the actual key-point count on the device (fewer than
  // two: no matches): the front half needs no host synchronisation, the count reaches the host with the frame's results
  VS_TRY(vs_match_ratio_dev_n(ctx, d + L.mapdesc, T.n_points, d + F.fdesc, T.max_kp, (const int*)(d + F.fn), ratio, d + F.mq,
                              d + F.mt, d + F.md, d + F.M, s, hb ? (int32_t*)(hb + AL.match) : nullptr, (int)AL.match_stride));
  // The new frame's observations are appended here, in the FRONT half: the rows lie behind everything the motion-only
  // solve of the previous frame reads (cameras up to its own), cam_start[slot] was written by the previous frame's append on
  // this same stream, and the counts go to this buffer set's own flag words -- so in pipelined use the append (and the
  // dispatch gap behind it) is off the critical path, the chain of back halves.
  int* flags = (int*)(d + L.flags) + 4 * set;
  if (++T.front_seq == 0) ++T.front_seq;  // never the zero the word starts with
  T.front_tag[set] = T.front_seq;
  hipLaunchKernelGGL(track_append_kernel, dim3(8), dim3(256), 0, s, (const double*)(d + L.xyz), (const float*)(d + F.fxy),
                     (const int*)(d + F.mq), (const int*)(d + F.mt), (const int*)(d + F.M), (double*)(d + L.moX),
                     (double*)(d + L.moUV), (int*)(d + L.cam_start), slot, L.cap_obs, flags, (const int*)(d + F.fn),
                     (unsigned*)(d + L.front_sync), T.front_tag[set], hb ? (unsigned*)(hb + AL.tags) : nullptr, T.api_seq);
  VS_LAUNCH_CHECK(ctx, "track_append_kernel");
  VS_HIP(ctx, hipEventRecord(T.ev_front[set], s));
  return VS_OK;
}

// Chained back halves alternate between two streams (buffer set 0: the context's, set 1: an auxiliary one): the PnP launch of
// frame k+1 does not queue behind frame k's solve -- it is resident, has waited for its front half and sampled when that
// solve publishes its tag.
hipStream_t track_back_stream(vs_ctx* ctx, int set, bool chained) { return chained && set ? ctx->aux_stream[1] : ctx->stream; }

size_t rb_len(const track_layout& L) { return (L.rb_end - L.mst + 255) & ~(size_t)255; }
size_t rb_stride(const track_layout& L) { return rb_len(L) + 256; }  // the block, then the word track_publish_kernel tags it with

// 1 when none of the kernels a chained period puts on its streams has a private segment (asked once per context from the
// runtime: hipFuncGetAttributes on the loaded code objects, i.e. on what actually runs, whatever the tuning knobs select)
bool track_chain_scratch_free(vs_ctx* ctx) {
  if (ctx->chain_scratch < 0) {
    size_t worst = 0;
    auto add = [&](const void* fn) {
      hipFuncAttributes a;
      if (hipFuncGetAttributes(&a, fn) != hipSuccess) {
        (void)hipGetLastError();
        worst = (size_t)-1;
      } else if (a.localSizeBytes > worst) {
        worst = a.localSizeBytes;
      }
    };
    add((const void*)track_append_kernel);
    add((const void*)track_publish_kernel);
    add((const void*)pnp_ransac_kernel);
    add((const void*)ba_motion_persistent<false>);
    size_t front = vs_match_chain_scratch_bytes(ctx), det = vs_detect_chain_scratch_bytes();
    worst = std::max(worst, std::max(front, det));
    ctx->chain_scratch = (long long)worst;
  }
  return ctx->chain_scratch == 0;
}

// May the back half of pose k be enqueued before the previous one's results are known?  Only the form that needs no host
// decision in between: PnP on, LM on, the motion-only solve in one launch.
bool track_can_chain(vs_ctx* ctx, int set, int k) {
  const auto& T = ctx->track;
  const int lm = T.params[set].lm_iterations;
  static const bool off = getenv("VS_TRACK_NOCHAIN") != nullptr;  // developer aid: A/B against the host-paced form
  if (off || T.in_redo) return false;
  // build property the chain relies on, checked on the loaded code objects: no kernel a chained period launches may use
  // scratch memory (a kernel that needs scratch may not be able to start while the kernel that waits for it in-kernel is
  // resident -- the first chained period of a fresh process timed out on exactly that in round 3)
  if (!track_chain_scratch_free(ctx)) return false;
  // (n_points: only the register-resident instantiation of the one-launch solve -- the other one uses scratch memory, and a
  // kernel that needs scratch may not be able to start while the kernel that waits for it in-kernel is running)
  return T.mst_both && T.pnp_iters > 0 && lm > 0 && T.n_points <= kMoPersistObs && mo_persistent_ok(ctx, k, 1 + lm * 10);
}

// Back half, enqueue only (context stream): wait for the front half, PnP-RANSAC from the
// previous pose, motion-only BA over the k free poses, one read-back copy.  *steps_out = LM launches enqueued.
// `k` is the pose index of the new frame.  chained: the previous frame's back half may still be running -- the row offset of
// the correspondences, the guess (the previous pose, in the state buffer its solve ended on) and that buffer's index are read
// by the PnP kernel on the device instead of being passed from the host.
int track_back_enqueue(vs_ctx* ctx, int set, int* steps_out, int k, bool chained = false, int publish_set = -1) {
  auto& T = ctx->track;
  const auto& Q = T.params[set];
  const track_layout L = layout_of(ctx);
  hipStream_t s = track_back_stream(ctx, set, chained);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  uint8_t* hp = (uint8_t*)ctx->h_track.p;
  const int H = T.pnp_iters > 0 ? T.pnp_iters : 1;
  if (!chained) VS_HIP(ctx, hipStreamWaitEvent(s, T.ev_front[set], 0));  // chained: the PnP kernel waits for the front half's tag itself
  double* cam0 = (double*)(d + L.cam0);
  double* cam1 = (double*)(d + L.cam1);
  // ---- PnP-RANSAC from the previous pose; its result becomes the new pose's record in both state buffers
  pnp_args A;
  memset(&A, 0, sizeof A);
  A.obj = (const double*)(d + L.moX) + (chained ? 0 : 3 * (size_t)T.obs_used);
  A.img = (const double*)(d + L.moUV) + (chained ? 0 : 2 * (size_t)T.obs_used);
  A.n = 0;
  A.n_dev = (const int*)(d + L.flags) + 4 * set + 1;
  A.iters_lm = Q.lm_iterations;
  A.iterations = T.pnp_iters;
  A.fx = T.K[0];
  A.fy = T.K[1];
  A.cx = T.K[2];
  A.cy = T.K[3];
  A.thr2 = Q.reproj_err * Q.reproj_err;
  A.confidence = Q.confidence;
  A.seed = Q.seed;
  memcpy(A.cam0, T.last_rec, sizeof A.cam0);
  if (T.api_back) {  // class-API back half (vs_track_back_begin / its redo): the caller's own guess and object-point precision
    if (T.api_guess_set) memcpy(A.cam0, T.api_guess_rec, sizeof A.cam0);
    A.obj_f32 = T.api_obj_f32;
  }
  if (chained) {
    A.off_dev = (const int*)(d + L.cam_start) + (k - 1);  // written by the previous frames' appends
    A.guess_dev[0] = cam0 + (size_t)(k - 1) * kCamStride;
    A.guess_dev[1] = cam1 + (size_t)(k - 1) * kCamStride;
    A.cur_dev = reinterpret_cast<const mo_state*>(d + L.mst);  // a finished solve leaves its record in both slots
    A.front_tag_dev = (const unsigned*)(d + L.front_sync) + 64;
    A.front_tag = T.front_tag[set];
    if (T.inject == 1) {  // developer aid (vs_track_debug): a tag nobody will publish -- every workgroup's wait runs out
      A.front_tag += 0x40000000u;
      T.inject = 0;
    }
    if (publish_set >= 0) {  // a chained frame is in flight on the other stream: its solve announces its end on the device
      A.back_tag_dev = (const unsigned*)(d + L.back_sync) + 64;
      A.back_tag = T.ba_tag[publish_set];
    }
    if (publish_set >= 0) {  // the previous frame's read-back rides on this launch's finishing workgroup
      uint8_t* rbp = (uint8_t*)ctx->h_track.p + kPinRb + (size_t)(1 + publish_set) * rb_stride(L);
      A.pub_src = (const uint4*)(d + L.mst);
      A.pub_dst = (uint4*)rbp;
      A.pub_n16 = (int)(rb_len(L) / 16);
      A.pub_tag_word = (unsigned*)(rbp + rb_len(L));
      A.pub_tag = T.back_tag[publish_set];
    }
  }
  A.model_out = (double*)(d + L.pnp_cam);
  A.result = (double*)(d + L.pnp_res);
  A.inl_out = (int*)(d + L.pnp_inl);
  A.rec_out[0] = cam0 + (size_t)k * kCamStride;
  A.rec_out[1] = cam1 + (size_t)k * kCamStride;
  const bool lm_on_device = T.pnp_iters > 0 && Q.lm_iterations > 0;  // pnp_ransac_kernel's finishing workgroup resets the LM records itself
  A.lm_init = lm_on_device ? reinterpret_cast<mo_state*>(d + L.mst) : nullptr;
  A.lm_cur = T.cur;
  if (T.api_stage == 1) {  // class-API period: the PnP outcome is read from pinned memory (vs_track_back_begin)
    const api_layout AL = api_layout_of(T.n_points, T.max_kp);
    A.host_result = (double*)((uint8_t*)ctx->h_api.p + AL.pnp_res);
    A.host_inl = (int*)((uint8_t*)ctx->h_api.p + AL.pnp_inl);
    A.host_tag_word = (unsigned*)((uint8_t*)ctx->h_api.p + AL.tags + 64);
    A.host_tag = T.api_seq;
  }
  if (T.pnp_iters > 0) {
    VS_TRY(pnp_tags(ctx, H, s, &A.tag, &A.epoch));
    VS_TRY(pnp_stamps(ctx, H, s, &A.stamps));
    hipLaunchKernelGGL(pnp_ransac_kernel, dim3(pnp_grid(H)), dim3(kPnpFinish), 0, s, A);
    VS_LAUNCH_CHECK(ctx, "pnp_ransac_kernel");
  } else {  // no PnP: the previous pose is the start
    memcpy(hp + 2048, T.last_rec, sizeof T.last_rec);
    VS_HIP(ctx, hipMemcpyAsync(A.rec_out[0], hp + 2048, sizeof T.last_rec, hipMemcpyHostToDevice, s));
    VS_HIP(ctx, hipMemcpyAsync(A.rec_out[1], hp + 2048, sizeof T.last_rec, hipMemcpyHostToDevice, s));
  }
  *steps_out = 0;
  if (Q.lm_iterations > 0 && !lm_on_device) {
    mo_state* h_st = (mo_state*)(hp + 1024);
    memset(h_st, 0, 256);
    h_st[1].need_lin = 1;
    h_st[1].ni = 2.0;
    h_st[1].cur = T.cur;
    h_st[0].cur = T.cur;
    VS_HIP(ctx, hipMemcpyAsync(d + L.mst, h_st, 256, hipMemcpyHostToDevice, s));
  }
  return VS_OK;
}

ba_dev track_ba_dev(vs_ctx* ctx, int set, int k, bool chained = false) {
  auto& T = ctx->track;
  const track_layout L = layout_of(ctx);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  ba_dev D;
  memset(&D, 0, sizeof D);
  D.n_poses = k + 1;
  D.nfp = k;
  D.np = 6 * k;
  D.max_it = T.params[set].lm_iterations;
  D.fx = T.K[0];
  D.fy = T.K[1];
  D.cx = T.K[2];
  D.cy = T.K[3];
  D.huber = T.params[set].huber;
  D.dcs = 1.0;
  D.slot_pose = (const int*)(d + L.slot_pose);
  D.cam_start = (const int*)(d + L.cam_start);
  D.cam[0] = (double*)(d + L.cam0);
  D.cam[1] = (double*)(d + L.cam1);
  D.mo_X = (const double*)(d + L.moX);
  D.mo_uv = (const double*)(d + L.moUV);
  D.mo_part = (double*)(d + L.part);
  D.mo_H = (double*)(d + L.H);
  D.st = reinterpret_cast<lm_state*>(d + L.mst);
  D.mo_box = reinterpret_cast<unsigned long long*>(d + L.box);
  D.mo_epoch = (unsigned)T.solve_epoch & 0xFFFFFu;
  if (chained) {  // the solve announces its end on the device (the next frame's PnP launch, on the other stream, waits for it)
    if (++T.ba_seq == 0) ++T.ba_seq;
    T.ba_tag[set] = T.ba_seq;
    D.mo_done = (unsigned*)(d + L.back_sync);
    D.mo_done_tag = T.ba_tag[set];
  }
  return D;
}

// enqueues up to one batch of LM launches followed by the read-back copy; returns the number of launches so far
// chained: the read-back goes to the buffer set's own pinned block through track_publish_kernel, which tags it (the host polls
// that tag; it cannot wait for the stream, which holds the next back half already)
int track_ba_batch(vs_ctx* ctx, int set, int* step, int k, bool chained = false) {
  auto& T = ctx->track;
  const track_layout L = layout_of(ctx);
  hipStream_t s = track_back_stream(ctx, set, chained);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  uint8_t* rb = (uint8_t*)ctx->h_track.p + kPinRb + (chained ? (size_t)(1 + set) * rb_stride(L) : 0);
  const int lm = T.params[set].lm_iterations;
  if (lm > 0) {
    if (*step == 0) T.solve_epoch = T.solve_epoch % 0xFFFFF + 1;  // 1 .. 2^20 - 1: never the zero the mailboxes start with
    const ba_dev D = track_ba_dev(ctx, set, k, chained);
    const int max_steps = 1 + lm * 10;
    // Launches after the one that finds the solve finished are predicated no-ops of ~5 us each on the critical path of the
    // frame, and consecutive frames of a stream need about the same number of LM steps: the first batch is as long as the
    // previous solve was (+1); a solve that needs more gets further batches (the results do not depend on the split).
    if (*step == 0 && !T.in_redo && mo_persistent_ok(ctx, k, max_steps)) {
      ba_dev Dp = D;
      if (ctx->mo_profile) {  // diagnostic: step stamps of camera 0's workgroup (vs_mo_profile)
        VS_TRY(vs_reserve(ctx, &ctx->d_mo_stamps, sizeof(unsigned long long) * 64 * 8));
        VS_HIP(ctx, hipMemsetAsync(ctx->d_mo_stamps.p, 0, sizeof(unsigned long long) * 64 * 8, s));
        Dp.mo_stamps = (unsigned long long*)ctx->d_mo_stamps.p;
      }
      // the whole solve in one launch (a frame has at most n_points matches); the final record lands in both state slots
      if (T.n_points <= kMoPersistObs) hipLaunchKernelGGL(ba_motion_persistent<false>, dim3(k), dim3(kMoThreads), 0, s, Dp, max_steps);
      else hipLaunchKernelGGL(ba_motion_persistent<true>, dim3(k), dim3(kMoThreads), 0, s, Dp, max_steps);
      VS_LAUNCH_CHECK(ctx, "ba_motion_persistent");
      *step = max_steps + 2;  // nothing left to enqueue
      T.mst_both = 1;
    } else {
      T.mst_both = 0;
      int batch = std::min(max_steps + 1 - *step, lm + 2);
      if (*step == 0 && T.lm_steps_hint > 0) batch = std::min(batch, std::max(3, T.lm_steps_hint + 1));
      for (int b = 0; b < batch; ++b, ++*step) {
        hipLaunchKernelGGL(ba_motion_step, dim3(k), dim3(kMoThreads), 0, s, D, *step);
        VS_LAUNCH_CHECK(ctx, "ba_motion_step");
      }
    }
  }
  // one copy brings back everything the host wants; if the solve needs another batch it is simply repeated
  if (chained) {
    // the block is published later: by the next frame's PnP launch (its finishing workgroup idles while the hypotheses
    // run), or by track_publish below when no chained frame follows
    if (++T.back_seq == 0) ++T.back_seq;
    T.back_tag[set] = T.back_seq;
  } else {
    // host-paced: the block goes out right behind the solve, by the same kernel (one launch; a copy command + a stream
    // synchronisation cost the host ~10 us more than polling the tag that kernel writes last)
    if (++T.back_seq == 0) ++T.back_seq;
    T.back_tag_sync = T.back_seq;
    hipLaunchKernelGGL(track_publish_kernel, dim3(1), dim3(256), 0, s, (const uint4*)(d + L.mst), (uint4*)rb, (int)(rb_len(L) / 16),
                       (unsigned*)(rb + rb_len(L)), T.back_tag_sync);
    VS_LAUNCH_CHECK(ctx, "track_publish_kernel");
  }
  return VS_OK;
}

// the read-back block of a chained back half to its pinned block, tagged (when no later PnP launch carries it)
int track_publish(vs_ctx* ctx, int set) {
  auto& T = ctx->track;
  const track_layout L = layout_of(ctx);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  uint8_t* rb = (uint8_t*)ctx->h_track.p + kPinRb + (size_t)(1 + set) * rb_stride(L);
  hipLaunchKernelGGL(track_publish_kernel, dim3(1), dim3(256), 0, track_back_stream(ctx, set, true), (const uint4*)(d + L.mst), (uint4*)rb,
                     (int)(rb_len(L) / 16), (unsigned*)(rb + rb_len(L)), T.back_tag[set]);
  VS_LAUNCH_CHECK(ctx, "track_publish_kernel");
  return VS_OK;
}

// Back half, completion: synchronise, run further LM batches if the solve is not finished, hand out the results.
int track_back_finish(vs_ctx* ctx, int set, int* step, double* poses_out, int* n_poses_out, int* n_matches, int* pnp_found,
                      float* xy_out, uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t, bool pnp_ran = true,
                      bool chained = false) {
  auto& T = ctx->track;
  const track_layout L = layout_of(ctx);
  const track_front& F = L.f[set];
  // chained: the streams hold the next back half already -- wait for this one's tag (below), and fetch the optional
  // per-frame arrays on another stream (this buffer set is not written again before the next call)
  hipStream_t s = chained ? ctx->aux_stream[0] : ctx->stream;
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  const uint8_t* rb = (const uint8_t*)ctx->h_track.p + kPinRb + (chained ? (size_t)(1 + set) * rb_stride(L) : 0);
  const mo_state* rb_st = (const mo_state*)rb;
  const int lm = T.params[set].lm_iterations, k = T.n_frames + 1;
  mo_state fin;
  memset(&fin, 0, sizeof fin);
  fin.cur = T.cur;
  // The four ways a back half's in-kernel hand-offs can fail (a tag that never came: the read-back, the PnP hypotheses, the
  // solve's rendezvous, an unfinished one-launch solve) are marked recoverable: the caller redoes the frame host-paced
  // (track_redo).  Nothing of the period's host-side state is touched before all of them have been ruled out.
  T.recoverable = 1;
  for (;;) {
    // the tag behind the block, written last (system-scope release) by the kernel that copied the block to pinned memory
    VS_TRY(track_poll(ctx, (const volatile unsigned*)(rb + rb_len(L)), chained ? T.back_tag[set] : T.back_tag_sync, "vs_track_frame"));
    if (lm == 0) break;
    fin = rb_st[(*step - 1) & 1];
    if (fin.terminated == 3) return vs_fail(ctx, VS_EHIP, "%s: the camera workgroups of the motion-only solve did not rendezvous", "vs_track_frame");
    if (fin.done || *step > 1 + lm * 10) break;
    if (chained) return vs_fail(ctx, VS_EHIP, "%s: the one-launch motion-only solve returned unfinished", "vs_track_frame_pipelined");
    VS_TRY(track_ba_batch(ctx, set, step, k));
  }
  const int* rb_flags = (const int*)(rb + (L.flags - L.mst)) + 4 * set;
  const double* rb_res = (const double*)(rb + (L.pnp_res - L.mst));
  const int M = rb_flags[1], n_kp = rb_flags[2];
  if (pnp_ran && T.pnp_iters > 0 && rb_res[16] < 0.0)
    return vs_fail(ctx, VS_EHIP, "%s: the PnP hypothesis workgroups did not report", "vs_track_frame");
  T.recoverable = 0;
  // the frame's front half has completed (its counts are in the block just read): did its match launch lose a train chunk?
  // Its rows were appended as "no match" then -- not redone (the front half is not repeatable once appended), but reported
  // HERE, against this frame, instead of on the stream's next launch or never.
  VS_TRY(vs_match_lost_check(ctx, "vs_track_frame"));
  if (rb_flags[0]) return vs_fail(ctx, VS_ENOMEM, "%s: observation capacity of the period exceeded", "vs_track_frame");
  bool copies = false;
  if (xy_out && n_kp > 0) {
    VS_HIP(ctx, hipMemcpyAsync(xy_out, d + F.fxy, sizeof(float) * 2 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (desc_out && n_kp > 0) {
    VS_HIP(ctx, hipMemcpyAsync(desc_out, d + F.fdesc, 32 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (match_q && match_t && M > 0) {
    VS_HIP(ctx, hipMemcpyAsync(match_q, d + F.mq, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipMemcpyAsync(match_t, d + F.mt, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (copies) VS_HIP(ctx, hipStreamSynchronize(s));
  T.cur = fin.cur;
  if (lm > 0 && fin.done) T.lm_steps_hint = fin.seq + 1;  // launches this solve needed (fin.seq: the launch that found it done)
  const double* h_cam = (const double*)(rb + ((T.cur ? L.cam1 : L.cam0) - L.mst));
  for (int i = 0; i <= k; ++i) pose_from_rec(h_cam + (size_t)i * kCamStride, poses_out + 16 * (size_t)i);
  memcpy(T.last_rec, h_cam + (size_t)k * kCamStride, sizeof T.last_rec);
  T.n_frames = k;
  T.obs_used += M;
  T.last_set = set;  // where vs_track_last_frame finds this frame's key points and matches (until the set's next front half)
  T.last_nkp = n_kp;
  T.last_M = M;
  *n_poses_out = k + 1;
  *n_matches = M;
  if (n_kp_out) *n_kp_out = n_kp;
  T.good.assign(rb, rb + (L.rb_end - L.mst));  // what a redo of the next frame would start from
  if (pnp_found) *pnp_found = pnp_ran && T.pnp_iters > 0 && rb_res[16] != 0.0 ? (int)rb_res[17] : 0;  // inliers of the PnP model
  return VS_OK;
}

// A back half whose in-kernel hand-offs did not complete (track_back_finish set T.recoverable) is redone ONCE, host-paced:
// drain every stream of the period, put the state the failed kernels may have written -- the LM records and both camera buffers
// -- back to what the newest frame handed out left (or, for the period's first frame, to the key frame's record), and run the
// frame's back half again in the form that needs no in-kernel wait: event-ordered behind the front half, the motion-only solve
// one launch per LM step.  Same arithmetic: the poses equal the undisturbed run's bit for bit (tests).  The frame's front half
// is not repeated -- its rows were appended by kernels that depend on no back half.
// ba_motion_step (what runs here, and in every host-paced solve) HAS a private segment: a kernel that needs scratch may be unable
// to start while another kernel waits for it in-kernel, so it may only be launched when no kernel of the period is waiting
// on a device-side tag -- which holds here (every stream is drained first) and in track_ba_batch's non-chained use (the host
// paces the launches; chained periods use ba_motion_persistent<false>, the register-resident instantiation, only).
int track_redo(vs_ctx* ctx, int set, double* poses_out, int* n_poses_out, int* n_matches, int* pnp_found, float* xy_out,
               uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t, bool pnp_ran) {
  auto& T = ctx->track;
  if (T.in_redo) return VS_EHIP;  // (the message of the failed redo stands)
  T.recoverable = 0;
  T.in_redo = 1;
  ++T.recoveries;
  struct leave {
    int* p;
    ~leave() { *p = 0; }
  } guard{&T.in_redo};
  for (hipStream_t st : {T.front_stream, ctx->stream, ctx->aux_stream[0], ctx->aux_stream[1]})
    if (st) VS_HIP(ctx, hipStreamSynchronize(st));
  const track_layout L = layout_of(ctx);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  const size_t blk = L.rb_end - L.mst;
  if (T.good.size() == blk) {
    // everything but the flag words (the front half of the frame being redone has written its counts there since)
    VS_HIP(ctx, hipMemcpy(d + L.mst, T.good.data(), L.flags - L.mst, hipMemcpyHostToDevice));
    VS_HIP(ctx, hipMemcpy(d + L.pnp_res, T.good.data() + (L.pnp_res - L.mst), L.rb_end - L.pnp_res, hipMemcpyHostToDevice));
  } else {  // no frame of this period has been handed out yet: the control image of vs_track_begin
    std::vector<uint8_t> img(blk, 0);
    memcpy(img.data() + (L.cam0 - L.mst), T.key_rec, sizeof T.key_rec);
    memcpy(img.data() + (L.cam1 - L.mst), T.key_rec, sizeof T.key_rec);
    VS_HIP(ctx, hipMemcpy(d + L.mst, img.data(), L.flags - L.mst, hipMemcpyHostToDevice));
    VS_HIP(ctx, hipMemcpy(d + L.pnp_res, img.data() + (L.pnp_res - L.mst), L.rb_end - L.pnp_res, hipMemcpyHostToDevice));
  }
  int step = 0;
  const int k = T.n_frames + 1;
  if (pnp_ran) {
    VS_TRY(track_back_enqueue(ctx, set, &step, k));
  } else {  // host-fed frame (vs_track_push_frame): its start record went up with the call and was overwritten by the restore
    uint8_t* hp = (uint8_t*)ctx->h_track.p;
    memcpy(hp + 2048, T.redo_rec, sizeof T.redo_rec);
    VS_HIP(ctx, hipMemcpyAsync((double*)(d + L.cam0) + (size_t)k * kCamStride, hp + 2048, sizeof T.redo_rec, hipMemcpyHostToDevice, ctx->stream));
    VS_HIP(ctx, hipMemcpyAsync((double*)(d + L.cam1) + (size_t)k * kCamStride, hp + 2048, sizeof T.redo_rec, hipMemcpyHostToDevice, ctx->stream));
    if (T.params[set].lm_iterations > 0) {
      mo_state* h_st = (mo_state*)(hp + 1024);
      memset(h_st, 0, 256);
      h_st[1].need_lin = 1;
      h_st[1].ni = 2.0;
      h_st[1].cur = T.cur;
      h_st[0].cur = T.cur;
      VS_HIP(ctx, hipMemcpyAsync(d + L.mst, h_st, 256, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  VS_TRY(track_ba_batch(ctx, set, &step, k));
  return track_back_finish(ctx, set, &step, poses_out, n_poses_out, n_matches, pnp_found, xy_out, desc_out, n_kp_out, match_q, match_t,
                           pnp_ran, false);
}

int track_check_frame(vs_ctx* ctx, const uint8_t* bgr, int w, int h_img, int stride, int lm_iterations, const char* who) {
  if (!ctx->track.active) return vs_fail(ctx, VS_EINVAL, "%s: no tracking period (call vs_track_begin)", who);
  if (!bgr || w < 31 || h_img < 31 || stride < 3 * w || lm_iterations < 0) return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", who);
  if (ctx->track.n_frames + (ctx->track.pending >= 0 ? 1 : 0) >= ctx->track.cap_frames)
    return vs_fail(ctx, VS_ENOMEM, "%s: the period holds max_frames frames already", who);
  return VS_OK;
}
}  // namespace

VS_API int vs_track_begin(vs_ctx* ctx, const double* xyz, const uint8_t* desc, int n_points, const double* key_pose,
                          double fx, double fy, double cx, double cy, int max_frames, int max_kp, int pnp_iterations) {
  if (!ctx) return VS_EINVAL;
  if (!xyz || !desc || !key_pose || n_points < 1 || max_frames < 1 || max_kp < 2 || pnp_iterations < 0 || pnp_iterations > 4096)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_begin");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto& T = ctx->track;
  T.last_set = -1;
  T.inject = 0;  // vs_track_debug arms a fault for the period it is called in
  // the one synchronisation of this call: the staging is free to be rewritten (or reallocated), the buffers are idle
  // (skipped when the previous period ended with all its results handed out: only tracking kernels touch these buffers, and
  // a synchronisation costs ~14 us per stream even when the stream is idle)
  if (!T.quiet || T.dirty) {
    VS_HIP(ctx, hipStreamSynchronize(T.front_stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->aux_stream[1]));
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  T.quiet = 0;
  T.dirty = 0;
  T.in_redo = 0;
  T.recoverable = 0;
  T.good.clear();
  const track_layout L = track_layout_of(n_points, max_frames, max_kp, pnp_iterations > 0 ? pnp_iterations : 1);
  VS_TRY(vs_reserve(ctx, &ctx->d_track, L.total));
  const size_t up = L.f[0].fxy;  // [xyz | mapdesc] are uploaded
  const size_t init = L.init_end - L.mst;  // the control image: LM records, flags, both camera buffers, counters, mailboxes
  // pinned staging: [control block kPinRb | three read-back blocks | push staging: idx + uv of one frame], and at least the
  // map + the control image of this call
  const size_t map_at = (kPinRb + 3 * rb_stride(L) + 256 + 20 * (size_t)n_points + 255) & ~(size_t)255;
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_track, std::max(map_at + up + init + 512, (size_t)1 << 16)));
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_api, api_layout_of(n_points, max_kp).total));
  if (!T.ev_api) VS_HIP(ctx, hipEventCreateWithFlags(&T.ev_api, hipEventDisableTiming));
  uint8_t* h = (uint8_t*)ctx->h_track.p + map_at;
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  memcpy(h + L.xyz, xyz, sizeof(double) * 3 * (size_t)n_points);
  memcpy(h + L.mapdesc, desc, 32 * (size_t)n_points);
  // the control image, built here and uploaded with one copy: everything zero (LM records: state buffer 0 holds the
  // estimate; flags; counters; mailboxes: no word carries a tag) except the key frame's record in both camera buffers and
  // the slot table (free-camera slot c = pose c + 1, pose 0 is the fixed key frame)
  uint8_t* hi = h + up;
  memset(hi, 0, init);
  double rec[kCamStride];
  rec_from_pose(key_pose, rec);
  memcpy(hi + (L.cam0 - L.mst), rec, sizeof rec);
  memcpy(hi + (L.cam1 - L.mst), rec, sizeof rec);
  int* sp = (int*)(hi + (L.slot_pose - L.mst));
  for (int c = 0; c < max_frames; ++c) sp[c] = c + 1;
  VS_HIP(ctx, hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(d + L.mst, hi, init, hipMemcpyHostToDevice, s));
  // no wait here: the frames' back halves run on this stream, and the front halves' stream is ordered behind the uploads
  VS_HIP(ctx, hipEventRecord(T.ev_api, s));
  VS_HIP(ctx, hipStreamWaitEvent(T.front_stream, T.ev_api, 0));
  VS_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[1], T.ev_api, 0));  // chained back halves of buffer set 1 run there
  T.active = 1;
  T.n_points = n_points;
  T.cap_frames = max_frames;
  T.max_kp = max_kp;
  T.pnp_iters = pnp_iterations;
  T.n_frames = 0;
  T.obs_used = 0;
  T.cur = 0;
  T.lm_steps_hint = 0;
  T.solve_epoch = 0;
  T.pending = -1;
  T.pending_step = -1;
  T.pending_chained = 0;
  T.pending_published = 0;
  T.mst_both = 1;
  T.next_set = 0;
  T.api_stage = 0;
  T.api_back = 0;
  T.K[0] = fx;
  T.K[1] = fy;
  T.K[2] = cx;
  T.K[3] = cy;
  memcpy(T.last_rec, rec, sizeof rec);
  memcpy(T.key_rec, rec, sizeof rec);
  return VS_OK;
}

VS_API int vs_track_end(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  // Nothing is outstanding when the last frame's results have been handed out (every entry point that returns results has
  // waited for them, and the front half of a frame precedes its back half): then the three synchronisations (~14 us apiece
  // even on idle streams) are skipped.  vs_track_begin and vs_destroy synchronise before they touch the period's buffers.
  if (ctx->track.pending >= 0 || ctx->track.api_stage != 0 || !ctx->track.active || ctx->track.dirty) {
    if (ctx->track.front_stream) (void)hipStreamSynchronize(ctx->track.front_stream);
    (void)hipStreamSynchronize(ctx->aux_stream[1]);
    (void)hipStreamSynchronize(ctx->stream);
  }
  ctx->track.quiet = 1;
  ctx->track.dirty = 0;
  ctx->track.active = 0;
  ctx->track.inject = 0;  // a fault armed for a period that never chained must not fire in a later one
  ctx->track.pending = -1;
  ctx->track.pending_step = -1;
  ctx->track.api_stage = 0;
  if (g_tt.on && g_tt.n) {
    fprintf(stderr, "[vs_track timing] per pipelined call over %ld calls: front enqueue %.1f us, wait+finish %.1f, PnP enqueue %.1f, BA+copy enqueue %.1f\n",
            g_tt.n, g_tt.sum[0] / g_tt.n, g_tt.sum[1] / g_tt.n, g_tt.sum[2] / g_tt.n, g_tt.sum[3] / g_tt.n);
    g_tt = track_timing();
  }
  return vs_match_lost_check(ctx, "vs_track_end");  // everything of the period has been waited for: nothing may go unreported
}

VS_API int vs_track_frame(vs_ctx* ctx, const uint8_t* bgr, int w, int h_img, int stride, int thr, double ratio,
                          double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                          double huber_delta, double* poses_out, int* n_poses_out, int* n_matches, int* pnp_found,
                          float* xy_out, uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t) {
  if (!ctx) return VS_EINVAL;
  VS_TRY(track_check_frame(ctx, bgr, w, h_img, stride, lm_iterations, "vs_track_frame"));
  if (!poses_out || !n_poses_out || !n_matches) return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_frame");
  if (ctx->track.pending >= 0) return vs_fail(ctx, VS_EINVAL, "%s: a pipelined frame is pending (flush it first)", "vs_track_frame");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  auto& T = ctx->track;
  T.params[0] = {pnp_reproj_err, pnp_confidence, huber_delta, seed, lm_iterations};
  T.api_stage = 0;  // a front half of the class-API entry points nobody followed up on: its rows are overwritten below
  T.api_back = 0;
  T.dirty = 1;      // (until this call has handed out its results: an error return leaves work in flight)
  VS_TRY(track_front_half(ctx, 0, T.n_frames, bgr, w, h_img, stride, thr, ratio, ctx->stream));
  int step = 0;
  VS_TRY(track_back_enqueue(ctx, 0, &step, T.n_frames + 1));
  VS_TRY(track_ba_batch(ctx, 0, &step, T.n_frames + 1));
  int rc = track_back_finish(ctx, 0, &step, poses_out, n_poses_out, n_matches, pnp_found, xy_out, desc_out, n_kp_out, match_q, match_t);
  if (rc != VS_OK && T.recoverable)
    rc = track_redo(ctx, 0, poses_out, n_poses_out, n_matches, pnp_found, xy_out, desc_out, n_kp_out, match_q, match_t, true);
  if (rc == VS_OK) T.dirty = 0;
  return rc;
}

// The per-frame arrays of the newest frame handed out, fetched AFTERWARDS: a caller that needs a frame's key points, descriptors
// and match lists only when the frame turns out to be a key frame (main.py:221-236 -- one frame in twenty) calls vs_track_frame
// with those outputs NULL and comes here for the one frame that needs them.  The frame's buffer set is not written again before
// the front half of the next frame but one (pipelined) / the next frame (frame by frame), so the rows are still there; the
// copies run on an auxiliary stream that holds nothing of the period.  Destinations in pinned memory are DMA-ed directly.
VS_API int vs_track_last_frame(vs_ctx* ctx, float* xy_out, uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t,
                               int* n_matches_out) {
  if (!ctx) return VS_EINVAL;
  auto& T = ctx->track;
  if (!T.active || T.last_set < 0)
    return vs_fail(ctx, VS_EINVAL, "%s: no frame of an open tracking period has been handed out yet", "vs_track_last_frame");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const track_layout L = layout_of(ctx);
  const track_front& F = L.f[T.last_set];
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  hipStream_t s = ctx->aux_stream[0];
  const int n_kp = T.last_nkp, M = T.last_M;
  bool copies = false;
  if (xy_out && n_kp > 0) {
    VS_HIP(ctx, hipMemcpyAsync(xy_out, d + F.fxy, sizeof(float) * 2 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (desc_out && n_kp > 0) {
    VS_HIP(ctx, hipMemcpyAsync(desc_out, d + F.fdesc, 32 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (match_q && match_t && M > 0) {
    VS_HIP(ctx, hipMemcpyAsync(match_q, d + F.mq, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipMemcpyAsync(match_t, d + F.mt, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    copies = true;
  }
  if (copies) VS_HIP(ctx, hipStreamSynchronize(s));
  if (n_kp_out) *n_kp_out = n_kp;
  if (n_matches_out) *n_matches_out = M;
  return VS_OK;
}

// Host-fed back half: the caller (the class API: FeatureMatcher.match_features + solvePnPRansac, src/v2/main.py:185-204)
// matched and estimated the start pose itself.  Appends the frame's observations and start pose to the resident period and
// runs the motion-only BA over all its poses -- what BundleAdjustment.motionOnlyBundleAdjustement (LocalBA.py:195-229)
// does on an otherwise unchanged local map, without rebuilding or re-uploading the period.
VS_API int vs_track_push_frame(vs_ctx* ctx, const int32_t* point_idx, const double* uv, int m, const double* pose16,
                               int lm_iterations, double huber_delta, double* poses_out, int* n_poses_out) {
  if (!ctx) return VS_EINVAL;
  auto& T = ctx->track;
  if (!T.active) return vs_fail(ctx, VS_EINVAL, "%s: no tracking period (call vs_track_begin)", "vs_track_push_frame");
  if (m < 0 || (m > 0 && (!point_idx || !uv)) || !pose16 || lm_iterations < 0 || !poses_out || !n_poses_out)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_push_frame");
  if (T.pending >= 0) return vs_fail(ctx, VS_EINVAL, "%s: a pipelined frame is pending (flush it first)", "vs_track_push_frame");
  if (T.api_stage == 2) return vs_fail(ctx, VS_EINVAL, "%s: a back half is running (vs_track_back_end first)", "vs_track_push_frame");
  T.api_stage = 0;  // a front half nobody followed up on: its rows are overwritten below
  T.api_back = 0;
  if (T.n_frames >= T.cap_frames) return vs_fail(ctx, VS_ENOMEM, "%s: the period holds max_frames frames already", "vs_track_push_frame");
  if (m > T.n_points) return vs_fail(ctx, VS_EINVAL, "%s: more observations than map points", "vs_track_push_frame");
  for (int i = 0; i < m; ++i)
    if (point_idx[i] < 0 || point_idx[i] >= T.n_points)
      return vs_fail(ctx, VS_EINVAL, "%s: map point index out of range", "vs_track_push_frame");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  const track_layout L = layout_of(ctx);
  hipStream_t s = ctx->stream;
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  uint8_t* hp = (uint8_t*)ctx->h_track.p;
  uint8_t* stage = hp + kPinRb + 3 * rb_stride(L);
  T.params[0] = {0.0, 0.0, huber_delta, 0ull, lm_iterations};
  const int slot = T.n_frames, k = T.n_frames + 1;
  if (m > 0) {
    memcpy(stage, point_idx, sizeof(int) * (size_t)m);
    double* suv = (double*)(stage + ((sizeof(int) * (size_t)m + 15) & ~(size_t)15));
    memcpy(suv, uv, sizeof(double) * 2 * (size_t)m);
    VS_HIP(ctx, hipMemcpyAsync(d + L.push_idx, stage, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, s));
    VS_HIP(ctx, hipMemcpyAsync(d + L.push_uv, suv, sizeof(double) * 2 * (size_t)m, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(track_push_kernel, dim3(8), dim3(256), 0, s, (const double*)(d + L.xyz), (const int*)(d + L.push_idx),
                     (const double*)(d + L.push_uv), m, T.n_points, (double*)(d + L.moX), (double*)(d + L.moUV),
                     (int*)(d + L.cam_start), slot, L.cap_obs, (int*)(d + L.flags));
  VS_LAUNCH_CHECK(ctx, "track_push_kernel");
  double rec[kCamStride];
  rec_from_pose(pose16, rec);
  memcpy(hp + 2048, rec, sizeof rec);
  memcpy(T.redo_rec, rec, sizeof rec);
  T.dirty = 1;
  double* cam0 = (double*)(d + L.cam0);
  double* cam1 = (double*)(d + L.cam1);
  VS_HIP(ctx, hipMemcpyAsync(cam0 + (size_t)k * kCamStride, hp + 2048, sizeof rec, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(cam1 + (size_t)k * kCamStride, hp + 2048, sizeof rec, hipMemcpyHostToDevice, s));
  if (lm_iterations > 0) {
    mo_state* h_st = (mo_state*)(hp + 1024);
    memset(h_st, 0, 256);
    h_st[1].need_lin = 1;
    h_st[1].ni = 2.0;
    h_st[1].cur = T.cur;
    h_st[0].cur = T.cur;
    VS_HIP(ctx, hipMemcpyAsync(d + L.mst, h_st, 256, hipMemcpyHostToDevice, s));
  }
  int step = 0, n_matches = 0;
  VS_TRY(track_ba_batch(ctx, 0, &step, k));
  int rc = track_back_finish(ctx, 0, &step, poses_out, n_poses_out, &n_matches, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, false);
  if (rc != VS_OK && T.recoverable)
    rc = track_redo(ctx, 0, poses_out, n_poses_out, &n_matches, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false);
  if (rc == VS_OK) T.dirty = 0;
  return rc;
}

VS_API int vs_track_frame_pipelined(vs_ctx* ctx, const uint8_t* bgr, int w, int h_img, int stride, int thr, double ratio,
                                    double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                                    double huber_delta, int* has_result, double* poses_out, int* n_poses_out,
                                    int* n_matches, int* pnp_found, float* xy_out, uint8_t* desc_out, int* n_kp_out,
                                    int32_t* match_q, int32_t* match_t) {
  if (!ctx) return VS_EINVAL;
  if (!has_result || !poses_out || !n_poses_out || !n_matches)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_frame_pipelined");
  *has_result = 0;
  auto& T = ctx->track;
  if (!T.active) return vs_fail(ctx, VS_EINVAL, "%s: no tracking period (call vs_track_begin)", "vs_track_frame_pipelined");
  if (bgr) VS_TRY(track_check_frame(ctx, bgr, w, h_img, stride, lm_iterations, "vs_track_frame_pipelined"));
  VS_HIP(ctx, hipSetDevice(ctx->device));
  // The back half of the previous frame is on the GPU already.  Chained form (PnP on, LM on, the motion-only solve in one
  // launch): the back half of THIS frame is enqueued too (on the other of two streams), before the previous one's results
  // are known -- the PnP kernel reads the row offset of its correspondences, its guess and the state-buffer index on the
  // device -- so the GPU passes from one back half to the next without the host in between (the host round trip and the
  // dispatch of the next kernel onto an idle queue were ~20 us per frame on the chain of back halves, which is the critical
  // path).  The previous frame's results are then awaited through the tag behind ITS read-back block.  Otherwise (a solve
  // that may need further LM batches decided on the host): the back half is enqueued once the previous results are in.
  T.api_stage = 0;
  T.api_back = 0;
  T.dirty = 1;  // (cleared where this call returns with nothing but the pending frame's own work outstanding)
  const int solve = T.pending;
  const bool solve_chained = solve >= 0 && T.pending_chained;
  int step = T.pending_step >= 0 ? T.pending_step : 0;
  if (solve >= 0 && T.pending_step < 0) {
    VS_TRY(track_back_enqueue(ctx, solve, &step, T.n_frames + 1));
    VS_TRY(track_ba_batch(ctx, solve, &step, T.n_frames + 1));
  }
  g_tt.start();
  int submitted = -1, next_step = 0;
  bool chain = false;
  if (bgr) {
    submitted = T.next_set;
    T.next_set ^= 1;
    T.params[submitted] = {pnp_reproj_err, pnp_confidence, huber_delta, seed, lm_iterations};
    // its pose index: behind the frames tracked so far and the one whose back half is still running
    const int k_new = T.n_frames + (solve >= 0 ? 2 : 1);
    VS_TRY(track_front_half(ctx, submitted, k_new - 1, bgr, w, h_img, stride, thr, ratio, T.front_stream));
    g_tt.lap(0);  // front half enqueued
    chain = (solve < 0 || solve_chained) && track_can_chain(ctx, submitted, k_new);
    if (chain) {
      const bool carry = solve_chained && !T.pending_published;  // frame k's read-back rides on frame k+1's PnP launch
      VS_TRY(track_back_enqueue(ctx, submitted, &next_step, k_new, true, carry ? solve : -1));
      if (carry) T.pending_published = 1;
      VS_TRY(track_ba_batch(ctx, submitted, &next_step, k_new, true));
      g_tt.lap(2);
    }
  }
  if (solve_chained && !T.pending_published) {  // nobody carries it (flush, or a frame that cannot be chained)
    VS_TRY(track_publish(ctx, solve));
    T.pending_published = 1;
  }
  T.pending = -1;
  T.pending_step = -1;
  T.pending_chained = 0;
  if (solve >= 0) {
    int rc = track_back_finish(ctx, solve, &step, poses_out, n_poses_out, n_matches, pnp_found, xy_out, desc_out, n_kp_out,
                               match_q, match_t, true, solve_chained);
    if (rc != VS_OK && T.recoverable) {
      // a hand-off inside the chain did not complete: redo this frame host-paced (everything is drained first).  The back
      // half of the frame submitted by this call -- enqueued behind the failed one, on its results -- is void with it: that
      // frame stays pending with only its front half done, and its back half is enqueued below, host-paced.
      rc = track_redo(ctx, solve, poses_out, n_poses_out, n_matches, pnp_found, xy_out, desc_out, n_kp_out, match_q, match_t, true);
      chain = false;
    }
    if (rc != VS_OK) {
      for (hipStream_t st : {T.front_stream, ctx->stream, ctx->aux_stream[1]})  // whatever this call enqueued: let it drain
        if (st) (void)hipStreamSynchronize(st);
      return rc;
    }
    *has_result = 1;
  }
  g_tt.lap(1);  // previous back half waited for and handed out
  if (submitted >= 0) {
    T.pending = submitted;  // from here on the frame counts as pending, whatever happens below
    T.pending_chained = chain;
    T.pending_published = 0;
    if (!chain) {
      VS_TRY(track_back_enqueue(ctx, submitted, &next_step, T.n_frames + 1));
      g_tt.lap(2);  // PnP enqueued
      VS_TRY(track_ba_batch(ctx, submitted, &next_step, T.n_frames + 1));
      g_tt.lap(3);  // BA + read-back enqueued
    }
    T.pending_step = next_step;
  }
  ++g_tt.n;
  T.dirty = 0;
  return VS_OK;
}

// ---- The same period driven call by call through the reference's class API (src/v2/main.py:181-214), without giving up the
// residency: Frame.process_frame -> vs_track_front (upload, detect + describe, match against the key frame's map points,
// append; ONE synchronisation, results mirrored into pinned memory by the kernels themselves); cv2.solvePnPRansac ->
// vs_track_back_begin (PnP-RANSAC and, right behind it on the stream, the motion-only BA; returns as soon as the PnP outcome is
// in); BundleAdjustment.motionOnlyBundleAdjustement -> vs_track_back_end (waits for the BA, which ran while the caller did
// its bookkeeping).  The Python side (map.py, _PeriodMirror) checks that what the caller passes to each of these calls is
// what the device already worked on, and otherwise starts the period afresh from the map.
VS_API int vs_track_front(vs_ctx* ctx, const uint8_t* bgr, int w, int h_img, int stride, int thr, double ratio, float* xy_out,
                          uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t, int32_t* match_d,
                          int* n_matches) {
  if (!ctx) return VS_EINVAL;
  VS_TRY(track_check_frame(ctx, bgr, w, h_img, stride, 0, "vs_track_front"));
  if (!xy_out || !desc_out || !n_kp_out || !match_q || !match_t || !match_d || !n_matches)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_front");
  auto& T = ctx->track;
  if (T.pending >= 0 || T.api_stage == 2) return vs_fail(ctx, VS_EINVAL, "%s: a back half is still running", "vs_track_front");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  if (++T.api_seq == 0) ++T.api_seq;
  T.dirty = 1;
  VS_TRY(track_front_half(ctx, 0, T.n_frames, bgr, w, h_img, stride, thr, ratio, ctx->stream, true));
  const api_layout AL = api_layout_of(T.n_points, T.max_kp);
  const uint8_t* hb = (const uint8_t*)ctx->h_api.p;
  VS_TRY(track_poll(ctx, (const volatile unsigned*)(hb + AL.tags), T.api_seq, "vs_track_front"));  // written by track_append_kernel
  const int n_kp = *(const int*)(hb + AL.det);
  const int32_t* hm = (const int32_t*)(hb + AL.match);
  const int M = hm[0];
  if (n_kp < 0 || n_kp > T.max_kp || M < 0 || M > T.n_points)
    return vs_fail(ctx, VS_EHIP, "%s: device returned impossible counts", "vs_track_front");
  memcpy(xy_out, hb + AL.det_xy, sizeof(float) * 2 * (size_t)n_kp);
  memcpy(desc_out, hb + AL.det_desc, (size_t)VS_DESC_BYTES * n_kp);
  memcpy(match_q, hm + 4, sizeof(int32_t) * (size_t)M);
  memcpy(match_t, hm + 4 + AL.match_stride, sizeof(int32_t) * (size_t)M);
  memcpy(match_d, hm + 4 + 2 * AL.match_stride, sizeof(int32_t) * (size_t)M);
  *n_kp_out = n_kp;
  *n_matches = M;
  T.api_stage = 1;
  T.api_matches = M;
  return VS_OK;
}

VS_API int vs_track_back_begin(vs_ctx* ctx, double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                               double huber_delta, const double* guess_pose16, int obj_as_f32, int* found, double* pose16,
                               int32_t* inliers, int* n_inliers) {
  if (!ctx) return VS_EINVAL;
  auto& T = ctx->track;
  if (!T.active || T.api_stage != 1) return vs_fail(ctx, VS_EINVAL, "%s: no front half to continue (vs_track_front)", "vs_track_back_begin");
  if (!found || !pose16 || !inliers || !n_inliers || lm_iterations < 0 || T.pnp_iters <= 0)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_back_begin");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  T.params[0] = {pnp_reproj_err, pnp_confidence, huber_delta, seed, lm_iterations};
  T.api_obj_f32 = obj_as_f32 != 0;
  T.api_guess_set = guess_pose16 != nullptr;
  if (guess_pose16) rec_from_pose(guess_pose16, T.api_guess_rec);
  T.api_back = 1;
  int step = 0;
  if (++T.api_seq == 0) ++T.api_seq;
  VS_TRY(track_back_enqueue(ctx, 0, &step, T.n_frames + 1));  // PnP-RANSAC; its outcome also goes to pinned memory, tagged
  VS_TRY(track_ba_batch(ctx, 0, &step, T.n_frames + 1));      // the motion-only BA + read-back, right behind it (no event between)
  T.api_step = step;
  T.api_stage = 2;
  const api_layout AL = api_layout_of(T.n_points, T.max_kp);
  VS_TRY(track_poll(ctx, (const volatile unsigned*)((const uint8_t*)ctx->h_api.p + AL.tags + 64), T.api_seq, "vs_track_back_begin"));
  const double* res = (const double*)((const uint8_t*)ctx->h_api.p + AL.pnp_res);
  if (res[16] < 0.0) return vs_fail(ctx, VS_EHIP, "%s: the PnP hypothesis workgroups did not report", "vs_track_back_begin");
  *found = res[16] != 0.0;
  *n_inliers = 0;
  if (*found) {
    const int m = (int)res[17];
    if (m < 0 || m > T.api_matches) return vs_fail(ctx, VS_EHIP, "%s: device returned an impossible inlier count", "vs_track_back_begin");
    memcpy(pose16, res, 16 * sizeof(double));
    memcpy(inliers, (const uint8_t*)ctx->h_api.p + AL.pnp_inl, sizeof(int32_t) * (size_t)m);
    *n_inliers = m;
  } else {
    pose_from_rec(T.api_guess_set ? T.api_guess_rec : T.last_rec, pose16);  // nothing found: the guess, as cv2 leaves rvec / tvec untouched
  }
  return VS_OK;
}

VS_API int vs_track_back_end(vs_ctx* ctx, double* poses_out, int* n_poses_out) {
  if (!ctx) return VS_EINVAL;
  auto& T = ctx->track;
  if (!T.active || T.api_stage != 2) return vs_fail(ctx, VS_EINVAL, "%s: no back half is running (vs_track_back_begin)", "vs_track_back_end");
  if (!poses_out || !n_poses_out) return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_back_end");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  int step = T.api_step, n_matches = 0, pnp_found = 0;
  T.api_stage = 0;
  int rc = track_back_finish(ctx, 0, &step, poses_out, n_poses_out, &n_matches, &pnp_found, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc != VS_OK && T.recoverable)  // (the PnP outcome the caller already holds is reproduced: same inputs, same seed, same guess)
    rc = track_redo(ctx, 0, poses_out, n_poses_out, &n_matches, &pnp_found, nullptr, nullptr, nullptr, nullptr, nullptr, true);
  T.api_back = 0;
  if (rc == VS_OK) T.dirty = 0;
  return rc;
}

// diagnostic hooks (include/vslam_hip_dev.h): per-step phase stamps of ba_motion_persistent inside a tracking period (camera 0's
// workgroup, thread 0).  vs_mo_profile_read synchronises and returns the newest stamped solve as rows of 8 doubles, one per LM
// step: columns 0..6 in shader-clock cycles since the solve's first stamp ([0] step entered, [1] partials of all cameras arrived,
// [2] decision taken, [3] linearised + reduced, [4] 6x6 solved + trial record, [5] trial chi2 summed, [6] partials posted; 0 = phase
// skipped), column 7 the wall clock in microseconds since the first step.  Returns the number of steps stamped.
VS_API int vs_mo_profile(vs_ctx* ctx, int enable) {
  if (!ctx) return VS_EINVAL;
  ctx->mo_profile = enable != 0;
  return VS_OK;
}
VS_API int vs_mo_profile_read(vs_ctx* ctx, double* out, int cap_rows) {
  if (!ctx || !out) return VS_EINVAL;
  if (!ctx->d_mo_stamps.p || cap_rows < 64) return 0;
  unsigned long long h[64 * 8];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, ctx->d_mo_stamps.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
    return vs_fail(ctx, VS_EHIP, "%s: read-back failed", "vs_mo_profile_read");
  int rows = 0;
  while (rows < 64 && h[rows * 8]) ++rows;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < 8; ++c) {
      const unsigned long long v = h[r * 8 + c], base = c == 7 ? h[7] : h[0];
      out[r * 8 + c] = !v ? 0.0 : c == 7 ? (double)(v - base) * 0.01 : (double)(v - base);
    }
  return rows;
}

// developer entry point (include/vslam_hip_dev.h): fault injection and the redo counter of the tracking period
VS_API int vs_track_debug(vs_ctx* ctx, int inject_fault, int* recoveries_out) {
  if (!ctx) return VS_EINVAL;
  if (inject_fault == 1) ctx->track.inject = 1;  // armed for this period only: vs_track_begin / vs_track_end disarm it
  if (recoveries_out) *recoveries_out = ctx->track.recoveries;
  return VS_OK;
}
