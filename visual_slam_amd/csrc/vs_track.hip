// vs_track.hip -- one tracking period resident on the device (gfx950).
#include "vs_ba_internal.h"

#include <algorithm>

using namespace vsba;

namespace {

// ------------------------------------------------------------------------------------------------ tracking session
// appends the new frame's observations to the period's camera-major arrays: obs i = (map point mq[i], keypoint mt[i])
__global__ __launch_bounds__(256) void track_append_kernel(const double* xyz, const float* fxy, const int* mq, const int* mt,
                                                           const int* d_M, double* mo_X, double* mo_uv, int* cam_start,
                                                           int slot, int cap_obs, int* flags) {
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
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tracking session
// One key-frame period of the reference's tracking loop (src/v2/main.py:173-214) kept resident on the device: the key
// frame's map points and descriptors are uploaded once (vs_track_begin); every vs_track_frame uploads only the image
// and runs detect+describe -> match against the map -> PnP-RANSAC from the previous pose -> append the observations
// -> motion-only BA over all poses of the period, with one host synchronisation for the keypoint count (the matcher's
// launch geometry needs it) and one at the end.  Same kernels, same arithmetic as the separate entry points.
namespace {
struct track_layout {
  size_t xyz, mapdesc, fxy, fscore, fdesc, fn, mq, mt, md, M, flags, cam0, cam1, moX, moUV, cam_start, slot_pose, part,
      H, mst, pnp_cam, pnp_pose, pnp_good, pnp_res, pnp_inl, rb_end, total;
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
  const int per = P < max_kp ? P : max_kp;
  L.cap_obs = F * (per > 0 ? per : 1);
  L.xyz = take(sizeof(double) * 3 * (size_t)P);
  L.mapdesc = take(32 * (size_t)P + 32);
  L.fxy = take(sizeof(float) * 2 * (size_t)max_kp);
  L.fscore = take((size_t)max_kp);
  L.fdesc = take(32 * (size_t)max_kp + 32);
  L.fn = take(sizeof(int));
  L.mq = take(sizeof(int) * (size_t)P);
  L.mt = take(sizeof(int) * (size_t)P);
  L.md = take(sizeof(int) * (size_t)P);
  L.M = take(sizeof(int));
  // read-back block: [LM state x2 | flags | PnP result | both camera buffers] is fetched with one copy per frame
  L.mst = take(2 * sizeof(mo_state));
  L.flags = take(4 * sizeof(int));
  L.pnp_res = take(sizeof(double) * 20);
  L.cam0 = take(sizeof(double) * kCamStride * (size_t)(F + 1));
  L.cam1 = take(sizeof(double) * kCamStride * (size_t)(F + 1));
  L.rb_end = off;
  L.moX = take(sizeof(double) * 3 * (size_t)L.cap_obs);
  L.moUV = take(sizeof(double) * 2 * (size_t)L.cap_obs);
  L.cam_start = take(sizeof(int) * (size_t)(F + 2));
  L.slot_pose = take(sizeof(int) * (size_t)(F + 1));
  L.part = take(sizeof(double) * 8 * (size_t)F);
  L.H = take(sizeof(double) * 42 * (size_t)F);
  L.pnp_cam = take(sizeof(double) * kCamStride * (size_t)H);
  L.pnp_pose = take(sizeof(double) * 12 * (size_t)H);
  L.pnp_good = take(sizeof(int) * (size_t)H);
  L.pnp_inl = take(sizeof(int) * (size_t)(per > 0 ? per : 1));
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
}  // namespace

VS_API int vs_track_begin(vs_ctx* ctx, const double* xyz, const uint8_t* desc, int n_points, const double* key_pose,
                          double fx, double fy, double cx, double cy, int max_frames, int max_kp, int pnp_iterations) {
  if (!ctx) return VS_EINVAL;
  if (!xyz || !desc || !key_pose || n_points < 1 || max_frames < 1 || max_kp < 2 || pnp_iterations < 0 || pnp_iterations > 4096)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_begin");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const track_layout L = track_layout_of(n_points, max_frames, max_kp, pnp_iterations > 0 ? pnp_iterations : 1);
  VS_TRY(vs_reserve(ctx, &ctx->d_track, L.total));
  const size_t up = L.fxy;  // [xyz | mapdesc] are uploaded
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_track, std::max(up, (size_t)1 << 16)));
  VS_HIP(ctx, hipStreamSynchronize(s));
  uint8_t* h = (uint8_t*)ctx->h_track.p;
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  memcpy(h + L.xyz, xyz, sizeof(double) * 3 * (size_t)n_points);
  memcpy(h + L.mapdesc, desc, 32 * (size_t)n_points);
  VS_HIP(ctx, hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemsetAsync(d + L.cam_start, 0, sizeof(int) * (size_t)(max_frames + 2), s));
  VS_HIP(ctx, hipMemsetAsync(d + L.flags, 0, 4 * sizeof(int), s));
  double rec[kCamStride];
  rec_from_pose(key_pose, rec);
  VS_HIP(ctx, hipStreamSynchronize(s));  // the pinned mirror is reused below
  memcpy(h, rec, sizeof rec);
  int* sp = (int*)(h + 1024);
  for (int c = 0; c < max_frames; ++c) sp[c] = c + 1;  // free-camera slot c = pose c + 1 (pose 0 is the fixed key frame)
  VS_HIP(ctx, hipMemcpyAsync(d + L.cam0, h, sizeof rec, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipMemcpyAsync(d + L.cam1, h, sizeof rec, hipMemcpyHostToDevice, s));
  if (sizeof(int) * (size_t)max_frames + 1024 > ctx->h_track.cap) return vs_fail(ctx, VS_ENOMEM, "%s: staging too small", "vs_track_begin");
  VS_HIP(ctx, hipMemcpyAsync(d + L.slot_pose, sp, sizeof(int) * (size_t)max_frames, hipMemcpyHostToDevice, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  ctx->track.active = 1;
  ctx->track.n_points = n_points;
  ctx->track.cap_frames = max_frames;
  ctx->track.max_kp = max_kp;
  ctx->track.pnp_iters = pnp_iterations;
  ctx->track.n_frames = 0;
  ctx->track.obs_used = 0;
  ctx->track.cur = 0;
  ctx->track.K[0] = fx;
  ctx->track.K[1] = fy;
  ctx->track.K[2] = cx;
  ctx->track.K[3] = cy;
  memcpy(ctx->track.last_rec, rec, sizeof rec);
  return VS_OK;
}

VS_API int vs_track_end(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  ctx->track.active = 0;
  return VS_OK;
}

VS_API int vs_track_frame(vs_ctx* ctx, const uint8_t* bgr, int w, int h_img, int stride, int thr, double ratio,
                          double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                          double huber_delta, double* poses_out, int* n_poses_out, int* n_matches, int* pnp_found,
                          float* xy_out, uint8_t* desc_out, int* n_kp_out, int32_t* match_q, int32_t* match_t) {
  if (!ctx) return VS_EINVAL;
  if (!ctx->track.active) return vs_fail(ctx, VS_EINVAL, "%s: no tracking period (call vs_track_begin)", "vs_track_frame");
  if (!bgr || w < 31 || h_img < 31 || stride < 3 * w || !poses_out || !n_poses_out || !n_matches || lm_iterations < 0)
    return vs_fail(ctx, VS_EINVAL, "%s: bad arguments", "vs_track_frame");
  auto& T = ctx->track;
  if (T.n_frames >= T.cap_frames) return vs_fail(ctx, VS_ENOMEM, "%s: the period holds max_frames frames already", "vs_track_frame");
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int P = T.n_points, H = T.pnp_iters > 0 ? T.pnp_iters : 1;
  const track_layout L = track_layout_of(P, T.cap_frames, T.max_kp, H);
  uint8_t* d = (uint8_t*)ctx->d_track.p;
  // ---- image upload (pitch = 3w rounded up to 4 bytes, 16 bytes of slack after the last row)
  const int pitch = (3 * w + 3) & ~3;
  VS_TRY(vs_reserve(ctx, &ctx->d_bgr, (size_t)pitch * h_img + 16));
  if (stride == pitch) VS_HIP(ctx, hipMemcpyAsync(ctx->d_bgr.p, bgr, (size_t)pitch * h_img, hipMemcpyHostToDevice, s));
  else VS_HIP(ctx, hipMemcpy2DAsync(ctx->d_bgr.p, pitch, bgr, stride, 3 * (size_t)w, h_img, hipMemcpyHostToDevice, s));
  VS_TRY(vs_detect_describe_bgr_dev(ctx, ctx->d_bgr.p, w, h_img, pitch, thr, T.max_kp, d + L.fxy, d + L.fscore, d + L.fdesc,
                                    d + L.fn, s));
  const size_t rb_bytes = L.rb_end - L.mst;  // the read-back block is mirrored at hp + 4096
  VS_TRY(vs_reserve_pinned(ctx, &ctx->h_track, 4096 + rb_bytes));
  uint8_t* hp = (uint8_t*)ctx->h_track.p;
  uint8_t* rb = hp + 4096;
  int* h_n = (int*)hp;  // [0]: keypoints
  VS_HIP(ctx, hipMemcpyAsync(h_n, d + L.fn, sizeof(int), hipMemcpyDeviceToHost, s));
  VS_HIP(ctx, hipStreamSynchronize(s));
  const int n_kp = h_n[0];
  if (n_kp_out) *n_kp_out = n_kp;
  if (xy_out && n_kp > 0) VS_HIP(ctx, hipMemcpyAsync(xy_out, d + L.fxy, sizeof(float) * 2 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
  if (desc_out && n_kp > 0) VS_HIP(ctx, hipMemcpyAsync(desc_out, d + L.fdesc, 32 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
  // ---- match the map's descriptors (query) against the frame's (train), Lowe ratio, ordered compaction
  if (n_kp >= 2) {
    VS_TRY(vs_match_ratio_dev(ctx, d + L.mapdesc, P, d + L.fdesc, n_kp, ratio, d + L.mq, d + L.mt, d + L.md, d + L.M, s));
  } else {
    VS_HIP(ctx, hipMemsetAsync(d + L.M, 0, sizeof(int), s));
  }
  const int slot = T.n_frames, k = T.n_frames + 1;  // new free-camera slot / pose index
  hipLaunchKernelGGL(track_append_kernel, dim3(8), dim3(256), 0, s, (const double*)(d + L.xyz), (const float*)(d + L.fxy),
                     (const int*)(d + L.mq), (const int*)(d + L.mt), (const int*)(d + L.M), (double*)(d + L.moX),
                     (double*)(d + L.moUV), (int*)(d + L.cam_start), slot, L.cap_obs, (int*)(d + L.flags));
  VS_LAUNCH_CHECK(ctx, "track_append_kernel");
  double* cam0 = (double*)(d + L.cam0);
  double* cam1 = (double*)(d + L.cam1);
  // ---- PnP-RANSAC from the previous pose; its result becomes the new pose's record in both state buffers
  pnp_args A;
  memset(&A, 0, sizeof A);
  A.obj = (const double*)(d + L.moX) + 3 * (size_t)T.obs_used;
  A.img = (const double*)(d + L.moUV) + 2 * (size_t)T.obs_used;
  A.n = 0;
  A.n_dev = (const int*)(d + L.flags) + 1;
  A.iters_lm = lm_iterations;
  A.iterations = T.pnp_iters;
  A.fx = T.K[0];
  A.fy = T.K[1];
  A.cx = T.K[2];
  A.cy = T.K[3];
  A.thr2 = pnp_reproj_err * pnp_reproj_err;
  A.confidence = pnp_confidence;
  A.seed = seed;
  memcpy(A.cam0, T.last_rec, sizeof A.cam0);
  A.cam_out = (double*)(d + L.pnp_cam);
  A.pose_out = (double*)(d + L.pnp_pose);
  A.good_out = (int*)(d + L.pnp_good);
  A.result = (double*)(d + L.pnp_res);
  A.inl_out = (int*)(d + L.pnp_inl);
  A.rec_out[0] = cam0 + (size_t)k * kCamStride;
  A.rec_out[1] = cam1 + (size_t)k * kCamStride;
  if (T.pnp_iters > 0) {
    hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3(H), dim3(64), 0, s, A);
    VS_LAUNCH_CHECK(ctx, "pnp_hypothesis_kernel");
    hipLaunchKernelGGL(pnp_finish_kernel, dim3(1), dim3(kPnpFinish), 0, s, A);
    VS_LAUNCH_CHECK(ctx, "pnp_finish_kernel");
  } else {  // no PnP: the previous pose is the start (pinned staging: h_track + 2048)
    memcpy(hp + 2048, T.last_rec, sizeof T.last_rec);
    VS_HIP(ctx, hipMemcpyAsync(A.rec_out[0], hp + 2048, sizeof T.last_rec, hipMemcpyHostToDevice, s));
    VS_HIP(ctx, hipMemcpyAsync(A.rec_out[1], hp + 2048, sizeof T.last_rec, hipMemcpyHostToDevice, s));
  }
  // ---- motion-only BA over the k free poses of the period
  ba_dev D;
  memset(&D, 0, sizeof D);
  D.n_poses = k + 1;
  D.nfp = k;
  D.np = 6 * k;
  D.max_it = lm_iterations;
  D.fx = T.K[0];
  D.fy = T.K[1];
  D.cx = T.K[2];
  D.cy = T.K[3];
  D.huber = huber_delta;
  D.dcs = 1.0;
  D.slot_pose = (const int*)(d + L.slot_pose);
  D.cam_start = (const int*)(d + L.cam_start);
  D.cam[0] = cam0;
  D.cam[1] = cam1;
  D.mo_X = (const double*)(d + L.moX);
  D.mo_uv = (const double*)(d + L.moUV);
  D.mo_part = (double*)(d + L.part);
  D.mo_H = (double*)(d + L.H);
  mo_state* d_mst = (mo_state*)(d + L.mst);
  D.st = reinterpret_cast<lm_state*>(d_mst);
  mo_state* h_st = (mo_state*)(hp + 1024);  // initial state (uploaded)
  memset(h_st, 0, 2 * sizeof(mo_state));
  h_st[1].need_lin = 1;
  h_st[1].ni = 2.0;
  h_st[1].cur = T.cur;
  h_st[0].cur = T.cur;
  mo_state fin;
  memset(&fin, 0, sizeof fin);
  fin.cur = T.cur;
  const mo_state* rb_st = (const mo_state*)(rb + (L.mst - L.mst));
  if (lm_iterations > 0) {
    VS_HIP(ctx, hipMemcpyAsync(d_mst, h_st, 2 * sizeof(mo_state), hipMemcpyHostToDevice, s));
    const int max_steps = 1 + lm_iterations * 10;
    int step = 0;
    for (;;) {
      const int batch = std::min(max_steps + 1 - step, lm_iterations + 2);
      for (int b = 0; b < batch; ++b, ++step) {
        hipLaunchKernelGGL(ba_motion_step, dim3(k), dim3(kMoThreads), 0, s, D, step);
        VS_LAUNCH_CHECK(ctx, "ba_motion_step");
      }
      // one copy brings back everything the host wants; if the solve needs another batch it is simply repeated
      VS_HIP(ctx, hipMemcpyAsync(rb, d + L.mst, rb_bytes, hipMemcpyDeviceToHost, s));
      VS_HIP(ctx, hipStreamSynchronize(s));
      if (rb_st[(step - 1) & 1].done || step > max_steps) break;
    }
    fin = rb_st[(step - 1) & 1];
  } else {
    VS_HIP(ctx, hipMemcpyAsync(rb, d + L.mst, rb_bytes, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  const int* rb_flags = (const int*)(rb + (L.flags - L.mst));
  const double* rb_res = (const double*)(rb + (L.pnp_res - L.mst));
  const int M = rb_flags[1];
  if (rb_flags[0]) return vs_fail(ctx, VS_ENOMEM, "%s: observation capacity of the period exceeded", "vs_track_frame");
  if (match_q && match_t && M > 0) {
    VS_HIP(ctx, hipMemcpyAsync(match_q, d + L.mq, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipMemcpyAsync(match_t, d + L.mt, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, s));
    VS_HIP(ctx, hipStreamSynchronize(s));
  } else if ((xy_out || desc_out) && lm_iterations == 0) {
    VS_HIP(ctx, hipStreamSynchronize(s));
  }
  T.cur = fin.cur;
  const double* h_cam = (const double*)(rb + ((T.cur ? L.cam1 : L.cam0) - L.mst));
  for (int i = 0; i <= k; ++i) pose_from_rec(h_cam + (size_t)i * kCamStride, poses_out + 16 * (size_t)i);
  memcpy(T.last_rec, h_cam + (size_t)k * kCamStride, sizeof T.last_rec);
  T.n_frames = k;
  T.obs_used += M;
  *n_poses_out = k + 1;
  *n_matches = M;
  if (pnp_found) *pnp_found = T.pnp_iters > 0 && rb_res[16] != 0.0;
  return VS_OK;
}
