// vs_ctx.hip -- context lifetime and error reporting of libvslam_hip.so
#include "vs_internal.h"

char g_vs_create_error[512] = "";

VS_API int vs_abi_version(void) { return VS_ABI_VERSION; }

VS_API int vs_create(vs_ctx** out, int device) {
  if (!out) return vs_fail(nullptr, VS_EINVAL, "vs_create: %s", "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return vs_fail(nullptr, VS_EHIP, "vs_create: no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "count 0");
  if (device < 0 || device >= n) return vs_fail(nullptr, VS_EINVAL, "vs_create: %s", "device index out of range");
  vs_ctx* ctx = new (std::nothrow) vs_ctx();
  if (!ctx) return vs_fail(nullptr, VS_ENOMEM, "vs_create: %s", "out of host memory");
  ctx->err[0] = 0;
  ctx->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
    vs_fail(nullptr, VS_EHIP, "vs_create: %s", hipGetErrorString(e));
    delete ctx;
    return VS_EHIP;
  }
  if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    vs_fail(nullptr, VS_EHIP, "vs_create: this library is built for gfx950 only, device is %s", ctx->prop.gcnArchName);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VS_EHIP;
  }
  *out = ctx;
  return VS_OK;
}

static void free_dev(vs_buf* b) {
  if (b->p) (void)hipFree(b->p);  // teardown: nothing useful can be done with an error here
  b->p = nullptr;
  b->cap = 0;
}

VS_API int vs_destroy(vs_ctx* ctx) {
  if (!ctx) return VS_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  vs_buf* dev[] = {&ctx->d_q,   &ctx->d_t,    &ctx->d_idx,     &ctx->d_dist, &ctx->d_partial, &ctx->d_mq,
                   &ctx->d_mt,  &ctx->d_md,   &ctx->d_cnt,     &ctx->d_bgr,  &ctx->d_gray,    &ctx->d_box,
                   &ctx->d_raw, &ctx->d_bandcnt, &ctx->d_hist, &ctx->d_xy,   &ctx->d_score,   &ctx->d_desc,
                   &ctx->d_n,   &ctx->d_xy_in, &ctx->d_keep,   &ctx->d_ba,  &ctx->d_track, &ctx->d_bgr2};
  for (vs_buf* b : dev) free_dev(b);
  for (vs_desc_entry& e : ctx->desc_cache) free_dev(&e.dev);
  if (ctx->h_pin.p) (void)hipHostFree(ctx->h_pin.p);
  if (ctx->h_pin_big.p) (void)hipHostFree(ctx->h_pin_big.p);
  if (ctx->h_track.p) (void)hipHostFree(ctx->h_track.p);
  for (hipEvent_t e : ctx->track.ev_front)
    if (e) (void)hipEventDestroy(e);
  if (ctx->track.front_stream) (void)hipStreamDestroy(ctx->track.front_stream);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return VS_OK;
}

uint64_t vs_fingerprint(const void* p, size_t bytes) {
  // four independent multiply-xor lanes over 64-bit words (vectorises; ~10+ GB/s), tail bytes folded at the end
  const uint64_t* w = (const uint64_t*)p;
  const size_t nw = bytes / 8;
  uint64_t h0 = 0x9E3779B97F4A7C15ull, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull, h3 = 0x27D4EB2F165667C5ull;
  size_t i = 0;
  for (; i + 4 <= nw; i += 4) {
    h0 = (h0 ^ w[i]) * 0x100000001B3ull;
    h1 = (h1 ^ w[i + 1]) * 0x100000001B3ull;
    h2 = (h2 ^ w[i + 2]) * 0x100000001B3ull;
    h3 = (h3 ^ w[i + 3]) * 0x100000001B3ull;
  }
  for (; i < nw; ++i) h0 = (h0 ^ w[i]) * 0x100000001B3ull;
  const uint8_t* b = (const uint8_t*)p + nw * 8;
  for (size_t k = 0; k < (bytes & 7); ++k) h1 = (h1 ^ b[k]) * 0x100000001B3ull;
  uint64_t h = h0 ^ (h1 << 1 | h1 >> 63) ^ (h2 << 2 | h2 >> 62) ^ (h3 << 3 | h3 >> 61);
  return h ^ (uint64_t)bytes;
}

bool vs_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory reports an error: clear it
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

static vs_desc_entry* desc_slot(vs_ctx* ctx, const uint8_t* h, int n, uint64_t fp, bool* hit) {
  vs_desc_entry* lru = &ctx->desc_cache[0];
  for (vs_desc_entry& e : ctx->desc_cache) {
    if (e.host == h && e.n == n && e.fp == fp && e.dev.p) {
      e.stamp = ++ctx->desc_stamp;
      *hit = true;
      return &e;
    }
    if (e.stamp < lru->stamp) lru = &e;
  }
  *hit = false;
  lru->host = h;
  lru->n = n;
  lru->fp = fp;
  lru->stamp = ++ctx->desc_stamp;
  return lru;
}

int vs_desc_resident(vs_ctx* ctx, const uint8_t* h, int n, const void** dev_out) {
  const size_t bytes = (size_t)VS_DESC_BYTES * n;
  bool hit;
  vs_desc_entry* e = desc_slot(ctx, h, n, vs_fingerprint(h, bytes), &hit);
  if (!hit) {
    e->host = nullptr;  // invalid while the upload is prepared
    VS_TRY(vs_reserve(ctx, &e->dev, bytes));
    VS_HIP(ctx, hipMemcpyAsync(e->dev.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    e->host = h;
  }
  *dev_out = e->dev.p;
  return VS_OK;
}

int vs_desc_adopt(vs_ctx* ctx, const uint8_t* h, int n, const void* dev_src) {
  if (n <= 0) return VS_OK;
  const size_t bytes = (size_t)VS_DESC_BYTES * n;
  bool hit;
  vs_desc_entry* e = desc_slot(ctx, h, n, vs_fingerprint(h, bytes), &hit);
  if (!hit) {
    e->host = nullptr;
    VS_TRY(vs_reserve(ctx, &e->dev, bytes));
    VS_HIP(ctx, hipMemcpyAsync(e->dev.p, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    e->host = h;
  }
  return VS_OK;
}

// pinned host memory for callers that want DMA without a staging copy (frames decoded straight into it)
VS_API int vs_host_alloc(vs_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return VS_EINVAL;
  *out = nullptr;
  VS_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return vs_fail(ctx, VS_ENOMEM, "hipHostMalloc(%s) failed: %s", "vs_host_alloc", hipGetErrorString(e));
  return VS_OK;
}
VS_API int vs_host_free(vs_ctx* ctx, void* p) {
  if (!ctx) return VS_EINVAL;
  if (p) VS_HIP(ctx, hipHostFree(p));
  return VS_OK;
}

VS_API const char* vs_last_error(const vs_ctx* ctx) { return ctx ? ctx->err : g_vs_create_error; }

VS_API void* vs_stream(vs_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

VS_API int vs_synchronize(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}
