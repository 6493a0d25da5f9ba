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
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return VS_EHIP;
  }
  *out = ctx;
  return VS_OK;
}

static void free_dev(vs_buf* b) {
  if (b->p) hipFree(b->p);
  b->p = nullptr;
  b->cap = 0;
}

VS_API int vs_destroy(vs_ctx* ctx) {
  if (!ctx) return VS_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  vs_buf* dev[] = {&ctx->d_q,   &ctx->d_t,    &ctx->d_idx,     &ctx->d_dist, &ctx->d_partial, &ctx->d_mq,
                   &ctx->d_mt,  &ctx->d_md,   &ctx->d_cnt,     &ctx->d_bgr,  &ctx->d_gray,    &ctx->d_box,
                   &ctx->d_raw, &ctx->d_bandcnt, &ctx->d_hist, &ctx->d_xy,   &ctx->d_score,   &ctx->d_desc,
                   &ctx->d_n,   &ctx->d_xy_in, &ctx->d_keep,   &ctx->d_ba};
  for (vs_buf* b : dev) free_dev(b);
  if (ctx->h_pin.p) hipHostFree(ctx->h_pin.p);
  if (ctx->h_pin_big.p) hipHostFree(ctx->h_pin_big.p);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return VS_OK;
}

VS_API const char* vs_last_error(const vs_ctx* ctx) { return ctx ? ctx->err : g_vs_create_error; }

VS_API void* vs_stream(vs_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

VS_API int vs_synchronize(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}
