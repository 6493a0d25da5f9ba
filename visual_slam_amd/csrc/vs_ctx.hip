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
  // all streams of the context at once, so that they land on hardware queues of their own (see vs_internal.h).  The
  // front-half stream has the lowest priority: in pipelined tracking the front half (detect, match) of frame k+1 shares
  // the GPU with the back half of frame k, which is the critical path -- a chain of short launches that should not queue
  // behind the detector's 240 workgroups (ba_motion_step: 7 us alone, 10-13 us behind them at equal priority)
  int prio_lo = 0, prio_hi = 0;
  e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->track.front_stream, hipStreamNonBlocking, prio_lo);
  for (int i = 0; i < VS_AUX_STREAMS && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking);
  for (hipEvent_t& ev : ctx->track.ev_front)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    vs_fail(nullptr, VS_EHIP, "vs_create: %s", hipGetErrorString(e));
    vs_destroy(ctx);
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

void vs_pool::run(int n, void (*f)(void*, int, int), void* a) {
  if (n <= 1) {
    f(a, 0, 1);
    return;
  }
  {
    std::lock_guard<std::mutex> lk(m);
    while ((int)th.size() < n - 1) {
      const int id = (int)th.size() + 1;
      const unsigned seen0 = gen;  // a worker born now waits for the NEXT generation
      th.emplace_back([this, id, seen0] {
        unsigned seen = seen0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv_work.wait(lk, [&] { return stop || gen != seen; });
          if (stop) return;
          seen = gen;
          if (id >= nt) continue;  // not part of this run
          void (*const f_)(void*, int, int) = fn;
          void* const a_ = arg;
          const int nt_ = nt;
          lk.unlock();
          f_(a_, id, nt_);
          lk.lock();
          if (--pending == 0) cv_done.notify_one();
        }
      });
    }
    fn = f;
    arg = a;
    nt = n;
    pending = n - 1;
    ++gen;
  }
  cv_work.notify_all();
  f(a, 0, n);
  std::unique_lock<std::mutex> lk(m);
  cv_done.wait(lk, [&] { return pending == 0; });
}

void vs_pool::shutdown() {
  {
    std::lock_guard<std::mutex> lk(m);
    stop = true;
  }
  cv_work.notify_all();
  for (std::thread& t : th)
    if (t.joinable()) t.join();
  th.clear();
}

static void free_dev(vs_buf* b) {
  if (b->p) (void)hipFree(b->p);  // teardown: nothing useful can be done with an error here
  b->p = nullptr;
  b->cap = 0;
}

VS_API int vs_destroy(vs_ctx* ctx) {
  if (!ctx) return VS_OK;
  // A context destroyed after the HIP runtime has begun to shut down (a static destructor, a late finaliser) must not
  // call into it: the first call tells, and then only the host object is released.
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamQuery(ctx->stream) == hipErrorContextIsDestroyed) {
    (void)hipGetLastError();
    delete ctx;
    return VS_OK;
  }
  (void)hipDeviceSynchronize();  // every stream the context has launched on (front half, a plan's match streams)
  const int lost_rc = vs_match_lost_check(ctx, "vs_destroy");  // a last launch nobody asked about: at least the return value says so
  if (lost_rc != VS_OK) fprintf(stderr, "libvslam_hip: %s\n", ctx->err);
  vs_buf* dev[] = {&ctx->d_q,   &ctx->d_t,    &ctx->d_mq,
                   &ctx->d_mt,  &ctx->d_md,   &ctx->d_cnt,     &ctx->d_bgr,  &ctx->d_gray,    &ctx->d_box,
                   &ctx->d_raw, &ctx->d_bandcnt, &ctx->d_hist, &ctx->d_xy,   &ctx->d_score,   &ctx->d_desc,
                   &ctx->d_n,   &ctx->d_xy_in, &ctx->d_keep,   &ctx->d_ba,  &ctx->d_track, &ctx->d_bgr2, &ctx->d_pnp_tag, &ctx->d_pnp_stamps, &ctx->d_framehist, &ctx->d_match_stamps, &ctx->d_mo_stamps, &ctx->d_bandflag};
  for (vs_buf* b : dev) free_dev(b);
  for (vs_match_scratch& m : ctx->match_scratch) {
    free_dev(&m.partial);
    if (m.flag.p) (void)hipHostFree(m.flag.p);
    m.flag = vs_buf();
    m.qtiles = m.nchunks = m.epoch = 0;
    free_dev(&m.idx);
    free_dev(&m.dist);
  }
  for (vs_prof_rec& r : ctx->match_prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (vs_desc_entry& e : ctx->desc_cache) {
    free_dev(&e.dev);
    if (e.shadow.p) (void)hipHostFree(e.shadow.p);
  }
  if (ctx->h_pin.p) (void)hipHostFree(ctx->h_pin.p);
  if (ctx->h_pin_big.p) (void)hipHostFree(ctx->h_pin_big.p);
  if (ctx->h_track.p) (void)hipHostFree(ctx->h_track.p);
  if (ctx->h_api.p) (void)hipHostFree(ctx->h_api.p);
  if (ctx->track.ev_api) (void)hipEventDestroy(ctx->track.ev_api);
  for (hipEvent_t e : ctx->track.ev_front)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_shard) (void)hipEventDestroy(ctx->ev_shard);
  if (ctx->ev_after) (void)hipEventDestroy(ctx->ev_after);
  if (ctx->poison_stream) (void)hipStreamDestroy(ctx->poison_stream);
  if (ctx->track.front_stream) (void)hipStreamDestroy(ctx->track.front_stream);
  for (hipStream_t a : ctx->aux_stream)
    if (a) (void)hipStreamDestroy(a);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return lost_rc;
}

bool vs_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory reports an error: clear it
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// ---- device-resident descriptor sets of the host matcher -----------------------------------------------------------
// An entry is (host address, row count) -> device copy + a pinned host SHADOW of the bytes that were uploaded.  A later
// call with the same address and count is a hit only if the caller's bytes are still byte-for-byte equal to the shadow
// (memcmp); anything else re-uploads.  No hash is involved, so a stale device copy can never be served.  The shadow is
// also the DMA staging buffer of the upload, i.e. it costs nothing extra on a miss.
static vs_desc_entry* desc_slot(vs_ctx* ctx, const uint8_t* h, int n, size_t bytes, bool* hit) {
  vs_desc_entry* lru = &ctx->desc_cache[0];
  for (vs_desc_entry& e : ctx->desc_cache) {
    if (e.host == h && e.n == n && e.dev.p && e.shadow.p) {
      e.stamp = ++ctx->desc_stamp;
      *hit = memcmp(h, e.shadow.p, bytes) == 0;
      return &e;  // same buffer with new contents re-uses its own entry
    }
    if (e.stamp < lru->stamp) lru = &e;
  }
  *hit = false;
  lru->stamp = ++ctx->desc_stamp;
  return lru;
}

int vs_desc_resident(vs_ctx* ctx, const uint8_t* h, int n, int role, const void** dev_out) {
  const size_t bytes = (size_t)VS_DESC_BYTES * n;
  if (bytes > VS_DESC_CACHE_MAX_BYTES) {
    // large sets (beyond anything a frame produces) are uploaded on every call into the role's own buffer
    vs_buf* b = role == 0 ? &ctx->d_q : &ctx->d_t;
    VS_TRY(vs_reserve(ctx, b, bytes));
    VS_HIP(ctx, hipMemcpyAsync(b->p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev_out = b->p;
    return VS_OK;
  }
  bool hit;
  vs_desc_entry* e = desc_slot(ctx, h, n, bytes, &hit);
  if (!hit) {
    e->host = nullptr;  // invalid while the upload is prepared
    VS_TRY(vs_reserve(ctx, &e->dev, bytes));
    VS_TRY(vs_reserve_pinned(ctx, &e->shadow, bytes));
    // every host entry point synchronises before it returns, so no earlier copy still reads this shadow
    memcpy(e->shadow.p, h, bytes);
    VS_HIP(ctx, hipMemcpyAsync(e->dev.p, e->shadow.p, bytes, hipMemcpyHostToDevice, ctx->stream));
    e->host = h;
    e->n = n;
  }
  *dev_out = e->dev.p;
  return VS_OK;
}

int vs_desc_adopt(vs_ctx* ctx, const uint8_t* h, int n, const void* dev_src, const uint8_t* host_src) {
  if (n <= 0) return VS_OK;
  const size_t bytes = (size_t)VS_DESC_BYTES * n;
  if (bytes > VS_DESC_CACHE_MAX_BYTES) return VS_OK;
  bool hit;
  vs_desc_entry* e = desc_slot(ctx, h, n, bytes, &hit);
  if (!hit) {
    e->host = nullptr;
    VS_TRY(vs_reserve(ctx, &e->dev, bytes));
    VS_TRY(vs_reserve_pinned(ctx, &e->shadow, bytes));
    memcpy(e->shadow.p, host_src, bytes);
    VS_HIP(ctx, hipMemcpyAsync(e->dev.p, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    e->host = h;
    e->n = n;
  }
  return VS_OK;
}

int vs_desc_slot_for_output(vs_ctx* ctx, size_t bytes, vs_desc_entry** slot) {
  *slot = nullptr;
  if (bytes > VS_DESC_CACHE_MAX_BYTES) return VS_OK;
  vs_desc_entry* lru = &ctx->desc_cache[0];
  for (vs_desc_entry& e : ctx->desc_cache)
    if (e.stamp < lru->stamp) lru = &e;
  lru->host = nullptr;  // invalid until adopted
  lru->n = 0;
  lru->stamp = ++ctx->desc_stamp;
  VS_TRY(vs_reserve(ctx, &lru->dev, bytes));
  VS_TRY(vs_reserve_pinned(ctx, &lru->shadow, bytes));
  *slot = lru;
  return VS_OK;
}

void vs_desc_adopt_slot(vs_ctx* ctx, vs_desc_entry* slot, const uint8_t* h, int n, const uint8_t* host_src) {
  if (n <= 0) return;
  for (vs_desc_entry& e : ctx->desc_cache)  // the host array may have been bound to another slot before: one binding only
    if (&e != slot && e.host == h) e.host = nullptr;
  memcpy(slot->shadow.p, host_src, (size_t)VS_DESC_BYTES * n);
  slot->host = h;
  slot->n = n;
  slot->stamp = ++ctx->desc_stamp;
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

VS_API void* vs_aux_stream(vs_ctx* ctx, int index) {
  return ctx && index >= 0 && index < VS_AUX_STREAMS ? (void*)ctx->aux_stream[index] : nullptr;
}

VS_API int vs_synchronize(vs_ctx* ctx) {
  if (!ctx) return VS_EINVAL;
  VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VS_OK;
}
