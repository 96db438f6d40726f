// extern "C" surface of libmodsgpu.so (include/mods_hip.h): context management, timing,
// detector entry points and introspection for the parity tests.
#include "common.hpp"
#include "detmath.hpp"
#include "ransac_host.hpp"
#include <cstdarg>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <mutex>
#include <vector>
#include <time.h>
#include <sys/prctl.h>

namespace mods {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// How a pipeline thread waits for a stream.  hipStreamQuery on a stream that is not done makes the runtime put a marker behind the
// stream's work and has the runtime's OWN thread spin until that marker completes: tools/ubench/rt_thread_probe.hip - a 5 ms kernel
// polled with hipStreamQuery every 50 us keeps that thread at 98.5 % of a core for the whole 5 ms, the same kernel behind a recorded
// event polled with hipEventQuery at 0.0 % (a word in pinned memory written by a kernel: 0.0 % as well).  That spinning was the "one
// core per process" of rounds 4 and 5 (1.1 ms per pair at 900 pairs/s).  So: record an event of the calling thread behind the work
// and poll THAT, sleeping in between.
// (A first attempt this round - a one-thread kernel that stores a sequence number to pinned memory - showed no gain only because
// other waits of the process still used hipStreamQuery; one pending query-marker is enough to keep the runtime's thread spinning.)
// Events / streams a thread borrows for as long as it lives: handed back to a process-wide list when the thread ends (no runtime call
// in a thread's exit path - the runtime may be shutting down by then -, and a pipeline that is created and destroyed again and again
// reuses the same few objects)
template <class T> struct HandlePool {
  std::mutex mu;
  std::vector<T> free_[16];
  T take(int dev, T (*make)()) {
    { std::lock_guard<std::mutex> lk(mu); if (!free_[dev].empty()) { T h = free_[dev].back(); free_[dev].pop_back(); return h; } }
    return make();
  }
  void give(int dev, T h) { std::lock_guard<std::mutex> lk(mu); free_[dev].push_back(h); }
};
template <class T> struct Borrowed {
  HandlePool<T> *pool; T h[16] = {};
  explicit Borrowed(HandlePool<T> *p) : pool(p) {}
  ~Borrowed() { for (int d = 0; d < 16; d++) if (h[d]) pool->give(d, h[d]); }
};
static HandlePool<hipEvent_t> *event_pool() { static HandlePool<hipEvent_t> *p = new HandlePool<hipEvent_t>(); return p; }     // (never destroyed: threads may outlive statics)
static HandlePool<hipStream_t> *stream_pool() { static HandlePool<hipStream_t> *p = new HandlePool<hipStream_t>(); return p; }
// the thread's event on the device that OWNS stream s (an event can only be recorded on a stream of its own device, and a polling
// thread may wait for a stream of another device than its current one: the plane getters, the multi-device helpers); the event is
// created with that device current, and the caller's device is put back
static hipEvent_t thread_event(hipStream_t s) {
  static thread_local Borrowed<hipEvent_t> t(event_pool());
  int cur = 0, dev = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  dev = cur;
  if (s && hipStreamGetDevice(s, &dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (dev < 0 || dev >= 16) return nullptr;
  if (!t.h[dev]) {
    if (dev != cur && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    t.h[dev] = t.pool->take(dev, +[]() -> hipEvent_t { hipEvent_t e = nullptr; return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? e : nullptr; });
    if (dev != cur) (void)hipSetDevice(cur);
  }
  return t.h[dev];
}
hipError_t stream_wait(hipStream_t s) {
  const long ns = tl_wait_sleep_ns;
  if (ns <= 0) return hipStreamSynchronize(s);
  hipEvent_t ev = thread_event(s);
  if (!ev) return hipStreamSynchronize(s);
  hipError_t e = hipEventRecord(ev, s);
  if (e != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(s); }   // (the runtime's own wait works from any current device)
  e = hipEventQuery(ev);
  for (int i = 0; i < tl_wait_spin_polls && e == hipErrorNotReady; i++) e = hipEventQuery(ev);
  // every poll is a runtime call and a wake-up of this thread (~4 us of CPU): the sleep grows by half per poll up to its bound, so a
  // batch's 10 ms cost a GPU worker ~45 polls instead of 200 and a scoring round a verify thread ~6 instead of 25 (process CPU per
  // pair 2.35 -> ~2.0 ms at the same rate, profiles/r05_wait_interval_sweep.log); the wait overshoots by at most a third of itself
  long cur = ns;
  const long cap = std::max(ns, tl_wait_sleep_max_ns);
  while (e == hipErrorNotReady) {
    const timespec ts = {0, cur};
    nanosleep(&ts, nullptr);
    e = hipEventQuery(ev);
    cur = std::min(cap, cur + cur / 2);
  }
  // a "not ready" answer may have been left as the thread's last error: it is not one
  const hipError_t last = hipGetLastError();
  if (e == hipSuccess && last != hipSuccess && last != hipErrorNotReady) return last;
  return e;
}

// (a blocking copy / fill inside a stream capture would be recorded as a node and then waited for - which a capture forbids: refused
// here, so that no upload can end up inside a recorded detect + describe graph)
static bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
hipError_t copy_wait(hipStream_t s, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return hipSuccess;
  if (capturing(s)) return hipErrorStreamCaptureUnsupported;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  return e != hipSuccess ? e : stream_wait(s);
}
hipError_t fill_wait(hipStream_t s, void *dst, int value, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (capturing(s)) return hipErrorStreamCaptureUnsupported;
  const hipError_t e = hipMemsetAsync(dst, value, bytes, s);
  return e != hipSuccess ? e : stream_wait(s);
}
hipStream_t thread_stream(int dev) {
  static thread_local Borrowed<hipStream_t> t(stream_pool());
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;                             // (nullptr = the legacy stream: what the call used before)
  if (dev < 0) dev = cur;
  if (dev >= 16) return nullptr;
  if (!t.h[dev]) {
    if (dev != cur && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    t.h[dev] = t.pool->take(dev, +[]() -> hipStream_t { hipStream_t q = nullptr; return hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess ? q : nullptr; });
    if (dev != cur) (void)hipSetDevice(cur);
  }
  return t.h[dev];
}
// the device a device pointer lives on (-1: not a device pointer the runtime knows - the calling thread's current device then)
int device_of_pointer(const void *p) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.type == hipMemoryTypeDevice ? a.device : -1;
}

void wait_mode_for_worker(long default_sleep_ns) {
  long ns = default_sleep_ns, cap = 250000;
  if (const char *m = getenv("MODS_SYNC")) {
    if (!strncmp(m, "spin", 4)) ns = 0;
    else if (!strncmp(m, "sleep", 5)) { if (m[5] == ':') { ns = std::max(1l, atol(m + 6)) * 1000; cap = ns; } }   // a given interval is kept as it is
    else fprintf(stderr, "mods: MODS_SYNC=%s not understood (spin | sleep[:microseconds]); keeping the default\n", m);
  }
  tl_wait_sleep_ns = ns;
  tl_wait_sleep_max_ns = cap;
  if (ns > 0) prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0);   // default slack 50 us: a 30 us sleep would take 80
}

StageScope::StageScope(mods_ctx *c, int s, double bytes) : ctx(c), stage(s) {
  on = (c->timing_mask >> s) & 1;
  if (!on) return;
  StageTimer &t = c->timers[s];
  auto get = [&]() {
    hipEvent_t e;
    if (!t.pool.empty()) { e = t.pool.back(); t.pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  };
  e0 = get(); e1 = get();
  t.bytes += bytes;
  (void)hipEventRecord(e0, c->stream);
}
StageScope::~StageScope() {
  if (!on) return;
  (void)hipEventRecord(e1, ctx->stream);
  ctx->timers[stage].pending.emplace_back(e0, e1);
}

}  // namespace mods

using namespace mods;

namespace mods { int ransac_failed(); }   // ransac.hip: the calling thread's last exp_ransac*custom hit a device failure

extern "C" {

const char *mods_last_error(void) { return g_err; }

int mods_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ctx_create_impl(int device, int max_w, int max_h, int batch, unsigned stream_flags, mods_ctx **out);

int mods_ctx_create(int device, int max_w, int max_h, int batch, mods_ctx **out) {
  // default: a stream that orders itself after the legacy default stream, so that images produced by
  // another library on the default stream (e.g. a torch .cuda() copy) are complete when kernels read them
  return ctx_create_impl(device, max_w, max_h, batch, hipStreamDefault, out);
}

// flags bit 0: non-blocking stream (no implicit ordering with the default stream; the caller guarantees
// that inputs are complete).  Needed for several contexts to overlap on one GPU.
int mods_ctx_create_ex(int device, int max_w, int max_h, int batch, int flags, mods_ctx **out) {
  return ctx_create_impl(device, max_w, max_h, batch, (flags & 1) ? hipStreamNonBlocking : hipStreamDefault, out);
}

static int ctx_create_impl(int device, int max_w, int max_h, int batch, unsigned stream_flags, mods_ctx **out) {
  if (!out || max_w <= 0 || max_h <= 0 || batch <= 0) { set_error("mods_ctx_create: bad arguments"); return MODS_E_ARG; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) {
    set_error("no HIP device available (count=%d, requested %d): libmodsgpu has no CPU path", n, device);
    return MODS_E_NODEVICE;
  }
  MODS_HIP_CHECK(hipSetDevice(device));
  mods_ctx *c = new mods_ctx();
  c->device = device; c->max_w = max_w; c->max_h = max_h; c->batch = batch;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cu = cus; }
  MODS_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, stream_flags));
  const size_t px = (size_t)max_w * max_h;
  size_t mc = px / 8;
  mc = std::max<size_t>(mc, 1u << 16);
  mc = std::min<size_t>(mc, 1u << 22);
  c->max_cand = (int)mc;
  MODS_HIP_CHECK(hipMalloc(&c->pyr_dev, sizeof(PyramidDev)));
  MODS_HIP_CHECK(hipMalloc(&c->input_dev, px * batch * sizeof(float)));
  MODS_HIP_CHECK(hipMalloc(&c->tmp_dev, px * batch * sizeof(float)));
  MODS_HIP_CHECK(hipMalloc(&c->gauss_taps_dev, 16 * 64 * sizeof(float)));
  MODS_HIP_CHECK(hipMalloc(&c->smm_mask_dev, 32 * 32 * sizeof(float)));
  MODS_HIP_CHECK(hipMalloc(&c->cand, sizeof(CandDev) * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->cand_count, sizeof(int) * 3 * batch));
  MODS_HIP_CHECK(hipMalloc(&c->keys_dev, sizeof(mods_affkey) * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->sort_keys, sizeof(unsigned long long) * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->sort_idx, sizeof(int) * 2 * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->rank_dev, sizeof(int) * mc * batch));
  c->nms_mask_words = (px / 64 + (size_t)max_h + 64) * kMaxLevels * batch;
  MODS_HIP_CHECK(hipMalloc(&c->nms_mask, sizeof(unsigned long long) * c->nms_mask_words));
  MODS_HIP_CHECK(hipHostMalloc(&c->host_counts, sizeof(int) * (5 * batch + 4)));   // cand / acc / key / region / inside counts, error flag
  MODS_HIP_CHECK(hipMalloc(&c->ori_dev, 48 * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->regions_dev, sizeof(mods_region) * mc * batch));
  MODS_HIP_CHECK(hipMalloc(&c->region_count, sizeof(int) * 3 * batch));   // regions, then 2 tier counts per image
  MODS_HIP_CHECK(hipMalloc(&c->inside_count, sizeof(int) * batch));
  MODS_HIP_CHECK(mods::fill_wait(c->stream, c->inside_count, 0, sizeof(int) * batch));
  *out = c;
  return MODS_OK;
}

static void dd_graph_drop(mods_ctx *c) {
  for (auto &e : c->dd_cache) (void)hipGraphExecDestroy(e.second);
  c->dd_cache.clear();
}

void mods_ctx_destroy(mods_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)mods::stream_wait(c->stream);
  for (auto &t : c->timers) {
    for (auto &p : t.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &e : t.pool) (void)hipEventDestroy(e);
  }
  (void)hipFree(c->pyr_dev); (void)hipFree(c->plane_pool); (void)hipFree(c->omap_pool); (void)hipFree(c->input_dev); (void)hipFree(c->u8_stage_dev);
  (void)hipFree(c->tmp_dev); (void)hipFree(c->alt_taps_dev); (void)hipFree(c->alt_planes); (void)hipFree(c->view_dev); (void)hipFree(c->gauss_taps_dev); (void)hipFree(c->smm_mask_dev); if (c->baum_stats_dev) (void)hipFree(c->baum_stats_dev); (void)hipFree(c->cand);
  (void)hipFree(c->cand_count); (void)hipFree(c->keys_dev); (void)hipFree(c->sort_keys); (void)hipFree(c->sort_idx); (void)hipFree(c->rank_dev); (void)hipFree(c->nms_mask);
  (void)hipHostFree(c->host_counts); (void)hipHostFree(c->pin_arena);
  (void)hipFree(c->ori_dev); (void)hipFree(c->ori_multi_dev); (void)hipFree(c->regions_dev); (void)hipFree(c->regions_half_dev); (void)hipFree(c->region_count); (void)hipFree(c->inside_count); (void)hipFree(c->desc_tables_dev); (void)hipFree(c->blur_table_dev);
  (void)hipFree(c->desc_err_dev); (void)hipFree(c->desc_scratch);
  (void)hipFree(c->m_desc); (void)hipFree(c->m_c); (void)hipFree(c->m_xy); (void)hipFree(c->m_u64); (void)hipFree(c->m_int); (void)hipFree(c->m_mid);
  (void)hipFree(c->m_p2); (void)hipFree(c->dd_buf); (void)hipFree(c->m_tent2); (void)hipFree(c->m_tent); (void)hipFree(c->m_tent_batch); (void)hipHostFree(c->m_count); (void)hipFree(c->m_regs);
  mser_release(c);
  for (mods_ctx *h : c->helpers) if (h) mods_ctx_destroy(h);
  for (auto &a : c->helper_stage) (void)hipFree(a.buf);
  (void)hipSetDevice(c->device);
  dd_graph_drop(c);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

int mods_ctx_sync(mods_ctx *c) { MODS_HIP_CHECK(mods::stream_wait(c->stream)); return MODS_OK; }
void *mods_ctx_stream(mods_ctx *c) { return (void *)c->stream; }

int mods_ctx_timing_enable(mods_ctx *c, int stage_mask) { c->timing_mask = stage_mask; return MODS_OK; }
int mods_ctx_pyramid_streams(mods_ctx *c, int n) {
  if (!c || n < 1 || n > 2) { set_error("pyramid streams: 1 or 2"); return MODS_E_ARG; }
  c->pyr_streams = n;
  return MODS_OK;
}

static int resolve_timers(mods_ctx *c) {
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  for (auto &t : c->timers) {
    for (auto &p : t.pending) {
      float ms = 0;
      MODS_HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second));
      t.total_ms += ms;
      t.launches++;
      t.pool.push_back(p.first); t.pool.push_back(p.second);
    }
    t.pending.clear();
  }
  return MODS_OK;
}

int mods_ctx_timing_read(mods_ctx *c, int stage, double *total_ms, int *launches, double *bytes) {
  if (stage < 0 || stage >= MODS_STAGE_COUNT) return MODS_E_ARG;
  int rc = resolve_timers(c);
  if (rc) return rc;
  if (total_ms) *total_ms = c->timers[stage].total_ms;
  if (launches) *launches = c->timers[stage].launches;
  if (bytes) *bytes = c->timers[stage].bytes;
  return MODS_OK;
}

int mods_ctx_timing_reset(mods_ctx *c) {
  int rc = resolve_timers(c);
  if (rc) return rc;
  for (auto &t : c->timers) { t.total_ms = 0; t.launches = 0; t.bytes = 0; }
  return MODS_OK;
}

// ---- detector -----------------------------------------------------------------------------
static int detect_common(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride,
                         const mods_hessaff_params *par, mods_affkey *out_host, int max_out, int *n_out_host) {
  if (!c || !img_dev || !par || !n_out_host) { set_error("detect: null argument"); return MODS_E_ARG; }
  if (w <= 0 || h <= 0 || (size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  if (stride < w) { set_error("detect: stride %d < width %d", stride, w); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = detect_any(c, img_dev, n_img, w, h, stride, par, 1.0, 1.0))) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(c->host_counts, c->cand_count, sizeof(int) * 3 * c->batch, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  for (int b = 0; b < n_img; b++) {
    if (c->host_counts[b] > c->max_cand) { set_error("NMS hit list overflow: %d > %d", c->host_counts[b], c->max_cand); return MODS_E_CAPACITY; }
    const int n = c->host_counts[2 * c->batch + b];
    n_out_host[b] = n;
    if (out_host) {
      if (n > max_out) { set_error("keypoint output overflow: %d > %d", n, max_out); return MODS_E_CAPACITY; }
      MODS_HIP_CHECK(mods::copy_wait(c->stream, out_host + (size_t)b * max_out, c->keys_dev + (size_t)b * c->max_cand, sizeof(mods_affkey) * n, hipMemcpyDeviceToHost));
    }
  }
  return MODS_OK;
}

int mods_detect_hessian_affine_dev(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride,
                                   const mods_hessaff_params *par, mods_affkey *out_host, int max_out, int *n_out_host) {
  return detect_common(c, img_dev, n_img, w, h, stride, par, out_host, max_out, n_out_host);
}

int mods_detect_hessian_affine(mods_ctx *c, const float *img, int w, int h, int stride, const mods_hessaff_params *par,
                               mods_affkey *out, int max_out, int *n_out) {
  if (!c || !img) { set_error("detect: null argument"); return MODS_E_ARG; }
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpy2DAsync(c->input_dev, sizeof(float) * w, img, sizeof(float) * stride, sizeof(float) * w, h,
                                  hipMemcpyHostToDevice, c->stream));
  return detect_common(c, c->input_dev, 1, w, h, w, par, out, max_out, n_out);
}

int mods_pyramid_octaves(mods_ctx *c) { return c->pyr.n_oct; }
int mods_pyramid_dims(mods_ctx *c, int o, int *w, int *h) {
  if (o < 0 || o >= c->pyr.n_oct) return MODS_E_ARG;
  *w = c->pyr.oct[o].w; *h = c->pyr.oct[o].h;
  return MODS_OK;
}
int mods_pyramid_plane(mods_ctx *c, int img, int o, int level, int kind, float *dst) {
  if (o < 0 || o >= c->pyr.n_oct || level < 0 || level >= c->pyr.n_levels || img < 0 || img >= c->last_n_img) return MODS_E_ARG;
  const OctaveDev &oc = c->pyr.oct[o];
  const float *p = (kind ? oc.resp[level] : oc.blur[level]) + (size_t)oc.w * oc.h * img;
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  MODS_HIP_CHECK(mods::copy_wait(c->stream, dst, p, sizeof(float) * (size_t)oc.w * oc.h, hipMemcpyDeviceToHost));
  return MODS_OK;
}
// accepted (post-dedup) localisation records of image `img`, in list (arbitrary) order
int mods_pyramid_candidates(mods_ctx *c, int img, mods_candidate *out, int max_out, int *n_out) {
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  int count = 0;
  MODS_HIP_CHECK(mods::copy_wait(c->stream, &count, c->cand_count + img, sizeof(int), hipMemcpyDeviceToHost));
  int n = std::min(count, c->max_cand);
  std::vector<CandDev> v(n);
  MODS_HIP_CHECK(mods::copy_wait(c->stream, v.data(), c->cand + (size_t)img * c->max_cand, sizeof(CandDev) * n, hipMemcpyDeviceToHost));
  // state: 2 accepted (claimed its octaveMap cell), 3 Baumberg converged, 4 Baumberg rejected
  int m = 0;
  for (auto &cd : v) {
    if (!(cd.state == 2 || cd.state == 3 || cd.state == 4)) continue;
    if (m < max_out) {
      mods_candidate &o = out[m];
      o.octave = cd.octave; o.level = cd.level; o.r0 = cd.r0; o.c0 = cd.c0; o.r = cd.r; o.c = cd.c;
      o.x = cd.x; o.y = cd.y; o.s = cd.s; o.pixelDistance = cd.pixelDistance; o.response = cd.response; o.type = cd.type;
    }
    m++;
  }
  *n_out = m;
  return m > max_out ? MODS_E_CAPACITY : MODS_OK;
}

// ---- orientation + description ----------------------------------------------------------------
static int check_desc_err(mods_ctx *c) {
  int e = 0;
  MODS_HIP_CHECK(hipMemcpyAsync(&e, c->desc_err_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  if (e) {
    MODS_HIP_CHECK(hipMemsetAsync(c->desc_err_dev, 0, sizeof(int), c->stream));
    set_error("measurement region larger than the descriptor scratch (P2 > 3*max(w,h) or > 4096 blur taps)");
    return MODS_E_CAPACITY;
  }
  return MODS_OK;
}

// The launches of one detect + describe call: scale space, NMS, localisation, Baumberg, order, orientation, extraction, SIFT and the
// read-back of the batch's counters (pinned host words) - ~70 dispatches for a 1080p batch, no host round trip between them.
static int dd_enqueue(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride, const mods_hessaff_params *det,
                      const mods_describe_params *desc) {
  int rc;
  if ((rc = detect_any(c, img_dev, n_img, w, h, stride, det, 1.0, 1.0))) return rc;
  const float *planes = img_dev;
  if (stride != w) planes = c->tmp_dev;   // pyramid_build repacked the batch there
  if ((rc = describe_run(c, planes, n_img, w, h, desc))) return rc;
  // every count and the error flag of the batch in one round trip (pinned host words)
  int *hc = c->host_counts;
  MODS_HIP_CHECK(hipMemcpyAsync(hc, c->cand_count, sizeof(int) * 3 * c->batch, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 3 * c->batch, c->region_count, sizeof(int) * c->batch, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 4 * c->batch, c->inside_count, sizeof(int) * n_img, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 5 * c->batch, c->desc_err_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  return MODS_OK;
}

static unsigned long long fnv1a(const void *p, size_t n, unsigned long long h) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// Replay instead of re-issue (mods_ctx_graphs).  A pipeline worker calls this with the same arguments batch after batch: every kernel
// argument is either a pointer into the context's pools or a parameter, every data-dependent quantity lives on the device, so the
// sequence of launches is the same each time.  The first call with a set of arguments runs eagerly (it may allocate pools, upload tap
// tables, set kernel attributes); the second one is recorded with hipStreamBeginCapture (both streams of the forked pyramid end up in
// the graph through their fork / join events) and launched; later ones are one hipGraphLaunch.  A recording that fails - a call in
// the chain that cannot be captured - switches the context back to eager launches for good.  Off while stage timers are on (their
// events belong to single launches), with external hooks (host round trips) and for the MSER detector (host threads).
// Only recordings with TWO branches (the forked scale space of a large batch) are replayed.  A recording that is one linear chain
// of launches faults on replay with this runtime (ROCm 7.0.2, HIP 7.0.51831: "illegal memory access" in the first replay, 12 of 12
// linear configurations of tools/exp_graph.py, while all 16 - linear and forked - replay bit-identically with
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, the switch for the runtime's replay of pre-built AQL packets, which linear kernel chains
// take; profiles/r05_graph_replay_matrix*.log).  That switch is read when the runtime starts, out of a library's reach, so calls
// whose scale space does not fork (small images, small batches, mods_ctx_pyramid_streams(1)) stay eager.
static int dd_run(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride, const mods_hessaff_params *det,
                  const mods_describe_params *desc) {
  const bool can = c->dd_graphs && c->timing_mask == 0 && !c->ext_fn && !c->shape_fn && !c->ori_fn && det->detectorType != MODS_DET_MSER;
  if (!can) { mods::dev_state_changed(c); return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc); }
  mods_ctx::DdKey key;
  key.img = img_dev; key.n_img = n_img; key.w = w; key.h = h; key.stride = stride;
  key.par_hash = fnv1a(desc, sizeof(*desc), fnv1a(det, sizeof(*det), 1469598103934665603ull)) ^ (unsigned long long)c->pyr_streams;
  key.epoch = c->dev_state_epoch;
  // A recording is made, and replayed, only right behind a call with the SAME arguments: tables that live on the device and are
  // refreshed from the host when the arguments change (the octave table, tap slots, masks) are then exactly what the recorded
  // launches expect, and no such refresh can end up inside a recording.  A worker's steady state - full batches of one geometry -
  // is a run of identical calls; a change of batch size or geometry costs one eager call.
  if (c->dd_stale) { dd_graph_drop(c); c->dd_linear.clear(); c->dd_stale = false; }
  const bool repeat = key == c->dd_prev;
  c->dd_prev = key;
  if (!repeat) return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
  for (auto &e : c->dd_cache)
    if (e.first == key) {
      MODS_HIP_CHECK(hipGraphLaunch(e.second, c->stream));
      c->dd_replays++;
      return MODS_OK;
    }
  for (const auto &k2 : c->dd_linear)
    if (k2 == key) return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
  hipStream_t s = c->stream;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    c->dd_graphs = false;
    return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
  }
  const int rc = dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
  hipGraph_t g = nullptr;
  const hipError_t e_end = hipStreamEndCapture(s, &g);
  hipGraphExec_t ex = nullptr;
  if (rc == MODS_OK && e_end == hipSuccess && g && !c->pyr_forked) {      // a linear recording: not replayed (above); nothing has run yet
    (void)hipGraphDestroy(g);
    if (c->dd_linear.size() >= 16) c->dd_linear.erase(c->dd_linear.begin());
    c->dd_linear.push_back(key);
    return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
  }
  if (rc == MODS_OK && e_end == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(g);
    if (c->dd_cache.size() >= 8) { (void)hipGraphExecDestroy(c->dd_cache.front().second); c->dd_cache.erase(c->dd_cache.begin()); }
    c->dd_cache.emplace_back(key, ex);
    MODS_HIP_CHECK(hipGraphLaunch(ex, c->stream));
    c->dd_replays++;
    return MODS_OK;
  }
  // not capturable here (or the chain itself failed): nothing has run - clean up and issue the launches directly
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  c->pyr_side = false;                    // (a fork recorded into the dead capture)
  c->dd_graphs = false;
  c->omap_dirty = true;
  return dd_enqueue(c, img_dev, n_img, w, h, stride, det, desc);
}

int mods_ctx_graphs(mods_ctx *c, int on) {
  if (!c) return MODS_E_ARG;
  c->dd_graphs = on != 0;
  if (!on) { dd_graph_drop(c); c->dd_linear.clear(); }
  mods::dev_state_changed(c);
  return MODS_OK;
}
long mods_ctx_graph_replays(const mods_ctx *c) { return c ? c->dd_replays : 0; }

int mods_detect_describe_dev(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride,
                             const mods_hessaff_params *det, const mods_describe_params *desc, int *n_detected_host,
                             int *n_regions_host) {
  if (!c || !img_dev || !det || !desc) { set_error("detect_describe: null argument"); return MODS_E_ARG; }
  if (w <= 0 || h <= 0 || (size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("detect_describe: image larger than the context"); return MODS_E_ARG; }
  if (stride < w) { set_error("detect_describe: stride %d < width %d", stride, w); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = dd_run(c, img_dev, n_img, w, h, stride, det, desc))) return rc;
  int *hc = c->host_counts;
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  for (int b = 0; b < n_img; b++) {
    if (hc[b] > c->max_cand) { set_error("NMS hit list overflow: %d > %d", hc[b], c->max_cand); return MODS_E_CAPACITY; }
    if (n_detected_host) n_detected_host[b] = hc[2 * c->batch + b];
    if (hc[3 * c->batch + b] > std::min(c->max_cand, 1 << 17)) { set_error("region list overflow: %d", hc[3 * c->batch + b]); return MODS_E_CAPACITY; }
    if (n_regions_host) n_regions_host[b] = hc[3 * c->batch + b];
  }
  c->last_region_counts.assign(hc + 3 * c->batch, hc + 3 * c->batch + n_img);
  c->last_inside_counts.assign(hc + 4 * c->batch, hc + 4 * c->batch + n_img);
  if (hc[5 * c->batch]) {
    MODS_HIP_CHECK(hipMemsetAsync(c->desc_err_dev, 0, sizeof(int), c->stream));
    set_error("measurement region larger than the descriptor scratch (P2 > 3*max(w,h) or > 4096 blur taps)");
    return MODS_E_CAPACITY;
  }
  return MODS_OK;
}

// Route the description of every following call through `fn` (NULL: back to the built-in RootSIFT).  The patches are
// ExtractPatchesColumn's (synth-detection.cpp:38-132; fp32, no photometric normalisation): patchSize x patchSize at mrSize.
int mods_ctx_set_external_descriptor(mods_ctx *c, mods_descriptor_fn fn, void *user, double mrSize, int patchSize) {
  if (!c) return MODS_E_ARG;
  c->ext_fn = fn; c->ext_user = user; c->ext_mr = mrSize; c->ext_ps = patchSize;
  return MODS_OK;
}

int mods_ctx_set_external_shape(mods_ctx *c, mods_descriptor_fn fn, void *user, double mrSize, int patchSize) {
  if (!c) return MODS_E_ARG;
  if (fn && (patchSize < 8 || patchSize > 63)) { set_error("external shape: patch size %d unsupported", patchSize); return MODS_E_ARG; }
  c->shape_fn = fn; c->shape_user = user; c->shape_mr = mrSize; c->shape_ps = patchSize;
  return MODS_OK;
}

int mods_ctx_set_external_orientation(mods_ctx *c, mods_descriptor_fn fn, void *user, double mrSize, int patchSize) {
  if (!c) return MODS_E_ARG;
  if (fn && (patchSize < 8 || patchSize > 63)) { set_error("external orientation: patch size %d unsupported", patchSize); return MODS_E_ARG; }
  c->ori_fn = fn; c->ori_user = user; c->ori_mr = mrSize; c->ori_ps = patchSize;
  return MODS_OK;
}
// host copy of the patches the describe stage extracted for image slot img in its last call ([n][ps][ps] fp32)
int mods_patches_fetch(mods_ctx *c, int img, int ps, float *out, int max_regions, int *n_out) {
  if (!c || !out || !n_out || img < 0 || img >= c->batch || !c->desc_scratch) { set_error("patches_fetch: nothing described"); return MODS_E_ARG; }
  const int reg_cap = std::min(c->max_cand, 1 << 17);
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  int n = 0;
  MODS_HIP_CHECK(mods::copy_wait(c->stream, &n, c->region_count + img, sizeof(int), hipMemcpyDeviceToHost));
  if (n > reg_cap) n = reg_cap;
  if (n > max_regions) { set_error("patch output overflow: %d > %d", n, max_regions); return MODS_E_CAPACITY; }
  MODS_HIP_CHECK(mods::copy_wait(c->stream, out, c->desc_scratch + (size_t)img * reg_cap * ps * ps, sizeof(float) * (size_t)n * ps * ps, hipMemcpyDeviceToHost));
  *n_out = n;
  return MODS_OK;
}

// size of the reference's unoriented ("None") region list of image slot img after the last describe call:
// detections whose centre, in the original frame, lies inside the image (imagerepresentation.cpp:867, 939)
// Baumberg work of image slot img, accumulated since mods_baumberg_stats_enable(ctx, 1): keypoints that entered the iteration and
// iterations run (each = one smmWindowSize^2 window of bilinear taps, affine.cpp:26-158).  Off by default: two atomics per keypoint.
int mods_baumberg_stats_enable(mods_ctx *c, int on) {
  if (!c) return MODS_E_ARG;
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  if (c->baum_stats_dev) { (void)hipFree(c->baum_stats_dev); c->baum_stats_dev = nullptr; }
  if (on) {
    MODS_HIP_CHECK(hipMalloc(&c->baum_stats_dev, sizeof(unsigned long long) * 2 * 64 * (size_t)c->batch));
    MODS_HIP_CHECK(hipMemset(c->baum_stats_dev, 0, sizeof(unsigned long long) * 2 * 64 * (size_t)c->batch));
  }
  mods::dev_pool_reallocated(c);         // (a recorded detect + describe graph holds the old pointer)
  return MODS_OK;
}
int mods_baumberg_stats(mods_ctx *c, int img, unsigned long long *keypoints, unsigned long long *iterations) {
  if (!c || img < 0 || img >= c->batch || !c->baum_stats_dev) return MODS_E_ARG;
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  unsigned long long v[2 * 64], kp = 0, it = 0;   // 64 slots per image (the kernel spreads its adds)
  MODS_HIP_CHECK(hipMemcpy(v, c->baum_stats_dev + 2 * 64 * (size_t)img, sizeof(v), hipMemcpyDeviceToHost));
  for (int q = 0; q < 64; q++) { kp += v[2 * q]; it += v[2 * q + 1]; }
  if (keypoints) *keypoints = kp;
  if (iterations) *iterations = it;
  return MODS_OK;
}

int mods_unoriented_count(mods_ctx *c, int img) {
  if (!c || img < 0 || img >= (int)c->last_inside_counts.size()) return 0;
  return c->last_inside_counts[img];
}

int mods_regions_fetch(mods_ctx *c, int img, mods_region *out, int max_out, int *n_out) {
  if (!c || img < 0 || img >= c->batch || !n_out) return MODS_E_ARG;
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  int n = 0;
  MODS_HIP_CHECK(mods::copy_wait(c->stream, &n, c->region_count + img, sizeof(int), hipMemcpyDeviceToHost));
  *n_out = n;
  if (out) {
    if (n > max_out) { set_error("region output overflow: %d > %d", n, max_out); return MODS_E_CAPACITY; }
    MODS_HIP_CHECK(mods::copy_wait(c->stream, out, c->regions_dev + (size_t)img * c->max_cand, sizeof(mods_region) * n, hipMemcpyDeviceToHost));
  }
  return MODS_OK;
}

int mods_regions_fetch_half(mods_ctx *c, int img, mods_region *out, int max_out, int *n_out) {
  if (!c || img < 0 || img >= c->batch || !n_out) return MODS_E_ARG;
  if (!c->have_half || !c->regions_half_dev) { set_error("no HalfRootSIFT descriptors: the last describe call did not ask for them"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  int n = 0;
  MODS_HIP_CHECK(mods::copy_wait(c->stream, &n, c->region_count + img, sizeof(int), hipMemcpyDeviceToHost));
  *n_out = n;
  if (out) {
    if (n > max_out) { set_error("region output overflow: %d > %d", n, max_out); return MODS_E_CAPACITY; }
    MODS_HIP_CHECK(mods::copy_wait(c->stream, out, c->regions_half_dev + (size_t)img * c->max_cand, sizeof(mods_region) * n, hipMemcpyDeviceToHost));
  }
  return MODS_OK;
}
const mods_region *mods_regions_half_dev(mods_ctx *c, int img) {
  return (c && c->have_half && c->regions_half_dev) ? c->regions_half_dev + (size_t)img * c->max_cand : nullptr;
}

int mods_orient_describe(mods_ctx *c, const float *img, int w, int h, int stride, const mods_affkey *keys, int n_keys,
                         const mods_describe_params *par, mods_region *out, int max_out, int *n_out) {
  if (!c || !img || !par || !n_out || (n_keys > 0 && !keys)) { set_error("orient_describe: null argument"); return MODS_E_ARG; }
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  if (n_keys > c->max_cand) { set_error("too many keypoints for the context: %d > %d", n_keys, c->max_cand); return MODS_E_CAPACITY; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpy2DAsync(c->input_dev, sizeof(float) * w, img, sizeof(float) * stride, sizeof(float) * w, h,
                                  hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(c->keys_dev, keys, sizeof(mods_affkey) * n_keys, hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(c->cand_count + 2 * c->batch, &n_keys, sizeof(int), hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));   // n_keys is a stack variable
  int rc = describe_run(c, c->input_dev, 1, w, h, par);
  if (rc) return rc;
  if ((rc = mods_regions_fetch(c, 0, out, max_out, n_out))) return rc;
  return check_desc_err(c);
}

int mods_dominant_angle(mods_ctx *c, const float *patch, int ps, double th, float *angle, int *found) {
  mods_describe_params dp = {5.1962, ps, 1, th, 5.1962, c->desc_ps ? c->desc_ps : 41, 1, 1, 0.2};
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = describe_configure(c, &dp);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, patch, sizeof(float) * ps * ps, hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_dominant_angle_test(c, c->input_dev, ps, th, c->tmp_dev))) return rc;
  float res[2];
  MODS_HIP_CHECK(hipMemcpyAsync(res, c->tmp_dev, sizeof(res), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  *found = res[0] != 0.f;
  *angle = res[1];
  return MODS_OK;
}

int mods_selftest_fast_sqrt(mods_ctx *c, unsigned long long *out5) {
  if (!c || !out5) { set_error("selftest: null argument"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  return launch_fast_sqrt_selftest(c, out5);
}

int mods_sift_patch(mods_ctx *c, const float *patch, int ps, int rootsift, double maxBinValue, uint8_t *out128) {
  mods_describe_params dp = {5.1962, c->desc_ori_ps ? c->desc_ori_ps : 32, 1, 0.8, 5.1962, ps, 1, rootsift, maxBinValue};
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = describe_configure(c, &dp);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, patch, sizeof(float) * ps * ps, hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_sift_patch_test(c, c->input_dev, ps, rootsift, maxBinValue, (uint8_t *)c->tmp_dev))) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(out128, c->tmp_dev, 128, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

// ---- matching ----------------------------------------------------------------------------------
// The packed output of the last search (n tentatives | correspondences | frames, common.hpp) in ONE device-to-host copy,
// split into the caller's arrays (any of them may be null).  Synchronises the context's stream.
int mods_match_copy_out(mods_ctx *c, int n, mods_tentative *tent, double *u6, double *laf) {
  if (n <= 0) return MODS_OK;
  static thread_local std::vector<char> stage;
  stage.resize(tent_bytes((size_t)n));
  MODS_HIP_CHECK(hipMemcpyAsync(stage.data(), c->m_tent, stage.size(), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  if (tent) memcpy(tent, stage.data(), sizeof(mods_tentative) * n);
  if (u6) memcpy(u6, stage.data() + tent_u6_off((size_t)n), sizeof(double) * 6 * n);
  if (laf) memcpy(laf, stage.data() + tent_laf_off((size_t)n), sizeof(double) * 14 * n);
  return MODS_OK;
}

int mods_match_fetch_internal(mods_ctx *c, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out) {
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  const int n = *(volatile int *)c->m_count;
  *n_out = n;
  if (n > max_out) { set_error("tentative output overflow: %d > %d", n, max_out); return MODS_E_CAPACITY; }
  return mods_match_copy_out(c, n, out, u6_out, laf_out);
}

int mods_match_fginn(mods_ctx *c, const mods_region *q, int n_q, const mods_region *t, int n_t, double ratio,
                     double contradDist, int nn, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out) {
  if (!c || !n_out || (n_q > 0 && !q) || (n_t > 0 && !t)) { set_error("match: null argument"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = match_ensure_buffers(c);
  if (rc) return rc;
  if (n_q > c->max_cand || n_t > c->max_cand) { set_error("match: list larger than the context capacity"); return MODS_E_CAPACITY; }
  MODS_HIP_CHECK(hipMemcpyAsync(c->m_regs, q, sizeof(mods_region) * n_q, hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(c->m_regs + c->max_cand, t, sizeof(mods_region) * n_t, hipMemcpyHostToDevice, c->stream));
  if ((rc = match_run(c, c->m_regs, n_q, c->m_regs + c->max_cand, n_t, ratio, contradDist, nn))) return rc;
  return mods_match_fetch_internal(c, out, u6_out, laf_out, max_out, n_out);
}

// MatchFLANNDistance (matching.cpp:572-633): nearest neighbour by Hamming distance over the descriptor bytes, kept when the
// distance is at most (int)(float)threshold; exact search
int mods_match_distance(mods_ctx *c, const mods_region *q, int n_q, const mods_region *t, int n_t, double threshold,
                        mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out) {
  if (!c || !n_out || (n_q > 0 && !q) || (n_t > 0 && !t)) { set_error("match: null argument"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = match_ensure_buffers(c);
  if (rc) return rc;
  if (n_q > c->max_cand || n_t > c->max_cand) { set_error("match: list larger than the context capacity"); return MODS_E_CAPACITY; }
  MODS_HIP_CHECK(hipMemcpyAsync(c->m_regs, q, sizeof(mods_region) * n_q, hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(c->m_regs + c->max_cand, t, sizeof(mods_region) * n_t, hipMemcpyHostToDevice, c->stream));
  if ((rc = match_run_distance(c, c->m_regs, n_q, c->m_regs + c->max_cand, n_t, threshold))) return rc;
  return mods_match_fetch_internal(c, out, u6_out, laf_out, max_out, n_out);
}

int mods_match_dev(mods_ctx *c, int img_q, int img_t, double ratio, double contradDist, int nn, mods_tentative *out,
                   double *u6_out, double *laf_out, int max_out, int *n_out) {
  if (!c || !n_out) { set_error("match: null argument"); return MODS_E_ARG; }
  const int nb = (int)c->last_region_counts.size();
  if (img_q < 0 || img_q >= nb || img_t < 0 || img_t >= nb) { set_error("match: image index outside the last described batch"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = match_run(c, c->regions_dev + (size_t)img_q * c->max_cand, c->last_region_counts[img_q],
                     c->regions_dev + (size_t)img_t * c->max_cand, c->last_region_counts[img_t], ratio, contradDist, nn);
  if (rc) return rc;
  return mods_match_fetch_internal(c, out, u6_out, laf_out, max_out, n_out);
}

// DuplicateFiltering, matching.cpp:2615-2679: optional stable sort, then the first correspondence
// (in list order) of every group whose two endpoints lie within r of each other survives.  A uniform
// grid over the first-image points (cell = r) bounds the candidates to the 3x3 neighbourhood; the
// accept/reject sequence is the reference's.
int mods_duplicate_filter(mods_tentative *tent, double *u6, double *laf, int n, double r, int mode, int *n_out) {
  if (!n_out || (n > 0 && (!tent || !u6))) { set_error("duplicate_filter: null argument"); return MODS_E_ARG; }
  *n_out = n;
  if (r <= 0 || n <= 0) return MODS_OK;
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  if (mode >= 1 && mode <= 3) {
    // (key, index) pairs sorted by value: the order of a stable sort by key, without a random access per comparison
    if (mode == 3 && !laf) { set_error("duplicate_filter: biggerRegion needs the local affine frames"); return MODS_E_ARG; }   // CompareCorrespondenceByScale, matching.cpp:74
    std::vector<std::pair<double, int>> keyed(n);
    for (int i = 0; i < n; i++)
      keyed[i] = {mode == 1 ? std::fabs(tent[i].ratio) : mode == 2 ? std::fabs((double)tent[i].d1) : std::fabs(laf[(size_t)i * 14 + 6]), i};
    std::sort(keyed.begin(), keyed.end());
    for (int i = 0; i < n; i++) order[i] = keyed[i].second;
  }
  std::vector<mods_tentative> ts(n);
  std::vector<double> us((size_t)n * 6), ls(laf ? (size_t)n * 14 : 0);
  for (int i = 0; i < n; i++) {
    ts[i] = tent[order[i]];
    memcpy(&us[(size_t)i * 6], &u6[(size_t)order[i] * 6], 6 * sizeof(double));
    if (laf) memcpy(&ls[(size_t)i * 14], &laf[(size_t)order[i] * 14], 14 * sizeof(double));
  }
  const double r_sq = r * r;
  // hash grid of the kept correspondences, keyed by the first-image cell: chained buckets in two flat arrays (which
  // kept correspondence is met first inside a cell does not matter, only whether one is met)
  // ~8 buckets per correspondence: a probe of an empty neighbour cell then rarely walks a chain of unrelated entries (with a fixed
  // 16 k buckets the 24 k correspondences of a 4096^2 pair cost ~13 useless distance tests each: 8 of the pair's 43 ms)
  size_t kBuckets = 1 << 14;
  while (kBuckets < (size_t)n * 8) kBuckets <<= 1;
  std::vector<int> head(kBuckets, -1), next(n);
  auto cell_of = [&](double v) { return (long long)std::floor(v / r); };
  auto hash = [&](long long cx, long long cy) { return (size_t)(((unsigned long long)cx * 73856093ull) ^ ((unsigned long long)cy * 19349663ull)) & (kBuckets - 1); };
  std::vector<char> keep(n, 0);
  int m = 0;
  for (int j = 0; j < n; j++) {
    const double x1 = us[(size_t)j * 6], y1 = us[(size_t)j * 6 + 1], x2 = us[(size_t)j * 6 + 3], y2 = us[(size_t)j * 6 + 4];
    const long long cx = cell_of(x1), cy = cell_of(y1);
    bool dup = false;
    for (long long dy = -1; dy <= 1 && !dup; dy++)
      for (long long dx = -1; dx <= 1 && !dup; dx++)
        for (int i = head[hash(cx + dx, cy + dy)]; i >= 0; i = next[i]) {
          double ex = us[(size_t)i * 6] - x1, ey = us[(size_t)i * 6 + 1] - y1;
          if (ex * ex + ey * ey > r_sq) continue;
          ex = us[(size_t)i * 6 + 3] - x2; ey = us[(size_t)i * 6 + 4] - y2;
          if (ex * ex + ey * ey <= r_sq) { dup = true; break; }
        }
    if (!dup) { keep[j] = 1; const size_t b = hash(cx, cy); next[j] = head[b]; head[b] = j; }
  }
  for (int j = 0; j < n; j++)
    if (keep[j]) {
      tent[m] = ts[j];
      memcpy(&u6[(size_t)m * 6], &us[(size_t)j * 6], 6 * sizeof(double));
      if (laf) memcpy(&laf[(size_t)m * 14], &ls[(size_t)j * 14], 14 * sizeof(double));
      m++;
    }
  *n_out = m;
  return MODS_OK;
}

// ---- verification ----------------------------------------------------------------------------------
extern "C" {
typedef struct { unsigned I; double J; } mods_score;
mods_score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int iter_type,
                             int *data_out, int oriented_constraint, unsigned inlLimit, double **resids,
                             void (*HDS1)(const double *, const double *, const double *, double *, int),
                             void (*HDSi1)(const double *, const double *, const double *, double *, int, int *, int),
                             void (*HDSidx1)(const double *, const double *, const double *, double *, int, int *, int), int doSymCheck);
int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out, int do_lo,
                      unsigned inlLimit, double **resids, double *H_best, int *Ih,
                      void (*EXFDS1)(const double *, const double *, double *, double *, int),
                      void (*FDS1)(const double *, const double *, double *, int), int doSymCheck);
void FDs(const double *, const double *, double *, int);
void FDsSym(const double *, const double *, double *, int);
void exFDs(const double *, const double *, double *, double *, int);
void exFDsSym(const double *, const double *, double *, double *, int);
void HDs(const double *, const double *, const double *, double *, int);
void HDsSym(const double *, const double *, const double *, double *, int);
void HDsSymMax(const double *, const double *, const double *, double *, int);
void HDsi(const double *, const double *, const double *, double *, int, int *, int);
void HDsiSym(const double *, const double *, const double *, double *, int, int *, int);
void HDsiSymMax(const double *, const double *, const double *, double *, int, int *, int);
void HDsidx(const double *, const double *, const double *, double *, int, int *, int);
void HDsSymidx(const double *, const double *, const double *, double *, int, int *, int);
void HDsSymidxMax(const double *, const double *, const double *, double *, int, int *, int);
}

// cv::invert(3x3, DECOMP_LU) as OpenCV evaluates small matrices: determinant and cofactors in
// double (OpenCV is not in the image: closed form assumed; result 0 when the determinant is 0).
static bool invert3(const double *S, double *t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0) { for (int i = 0; i < 9; i++) t[i] = 0; return false; }
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return true;
}

// What LORANSACFiltering does with the model of the homography branch (matching.cpp:745-805): H -> row-major img1->img2 by inv(H^T),
// NaiveHCheck (10 px, :1014-1043) over the RANSAC inliers, H_LAF_check (:250-308: HDsSymMax on the three frame points).  mask[i] = 1
// for the correspondences that survive; returns their number.  ops = the host SIMD table (ransac_simd.hpp): both checks are the
// symmetric transfer error of point pairs, evaluated lanes-wide over structure-of-arrays copies of the inliers (every lane does the
// scalar code's operations in its order: the same bits); ops = nullptr runs the scalar statement (kept for the self-test).
static int h_post_checks(const double *u6, const double *laf, int n, const unsigned char *inl2, const double *Hloran,
                         const mods_ransac_params *par, const mods::rs::SimdOps *ops, unsigned char *mask, double *H_out) {
  const int MIN_POINTS = 8;
  // H: inv(Hloran^T); reading the column-major h as a row-major matrix is H^T, its transpose is h read column-wise
  const double Ht[9] = {Hloran[0], Hloran[3], Hloran[6], Hloran[1], Hloran[4], Hloran[7], Hloran[2], Hloran[5], Hloran[8]};
  double Hinv[9];
  invert3(Ht, Hinv);
  bool nonzero = false;
  for (int i = 0; i < 9; i++) nonzero = nonzero || (Hinv[i] != 0.0);
  if (!nonzero) return 0;
  for (int i = 0; i < 9; i++) H_out[i] = Hinv[i];
  std::vector<int> cur;
  for (int i = 0; i < n; i++) if (inl2[i]) cur.push_back(i);
  const double affineFerror = 3.0 * par->HLAFCoef * par->err_threshold;
  const double err_sq = 10.0 * 10.0;
  const double ks = 3.0;   // matching.cpp:171
  double Hi[9];
  invert3(Hinv, Hi);
  if (ops) {
    const int m = (int)cur.size();
    if (m == 0) return 0;
    const int m_pad = (m + mods::rs::SIMD_PAD - 1) / mods::rs::SIMD_PAD * mods::rs::SIMD_PAD;
    std::vector<double> buf((size_t)7 * m_pad);
    double *c0 = buf.data(), *c1 = c0 + m_pad, *c2 = c1 + m_pad, *c3 = c2 + m_pad, *e0 = c3 + m_pad, *e1 = e0 + m_pad, *e2 = e1 + m_pad;
    const double *soa[5] = {c0, c1, c2, c3, c3};
    for (int k = 0; k < m_pad; k++) {
      const double *p = u6 + (size_t)cur[k < m ? k : m - 1] * 6;
      c0[k] = p[0]; c1[k] = p[1]; c2[k] = p[3]; c3[k] = p[4];
    }
    ops->hsym_both_all(soa, m_pad, Hinv, Hi, e0, e1);
    int ok = 0;
    for (int k = 0; k < m; k++) if ((e0[k] <= err_sq) && (e1[k] <= err_sq)) ok++;
    if (ok < MIN_POINTS) return 0;
    if (affineFerror > 0 && laf) {
      mods::rs::SymH sh;
      mods::rs::sym_prepare(Hloran, &sh);
      // the centre pair is the correspondence itself (u[0..1], u[3..4] = f[0..1], f[7..8]); then the two frame points
      for (int k = 0; k < m_pad; k++) { const double *f = laf + (size_t)cur[k < m ? k : m - 1] * 14; c0[k] = f[0]; c1[k] = f[1]; c2[k] = f[7]; c3[k] = f[8]; }
      ops->hsym_all(soa, m_pad, sh.H1, sh.Hinv, 1, e0);
      for (int k = 0; k < m_pad; k++) {
        const double *f = laf + (size_t)cur[k < m ? k : m - 1] * 14;
        c0[k] = f[0] + ks * f[3] * f[6]; c1[k] = f[1] + ks * f[5] * f[6]; c2[k] = f[7] + ks * f[10] * f[13]; c3[k] = f[8] + ks * f[12] * f[13];
      }
      ops->hsym_all(soa, m_pad, sh.H1, sh.Hinv, 1, e1);
      for (int k = 0; k < m_pad; k++) {
        const double *f = laf + (size_t)cur[k < m ? k : m - 1] * 14;
        c0[k] = f[0] + ks * f[2] * f[6]; c1[k] = f[1] + ks * f[4] * f[6]; c2[k] = f[7] + ks * f[9] * f[13]; c3[k] = f[8] + ks * f[11] * f[13];
      }
      ops->hsym_all(soa, m_pad, sh.H1, sh.Hinv, 1, e2);
      int kept = 0;
      for (int k = 0; k < m; k++) {
        const double sumErr = std::sqrt(e0[k] + e1[k] + e2[k]);
        if (!(sumErr > affineFerror)) cur[kept++] = cur[k];
      }
      cur.resize(kept);
    }
  } else {
    // NaiveHCheck over the RANSAC inliers
    {
      int ok = 0;
      for (int i : cur) {
        const double *p = u6 + (size_t)i * 6;
        const double *Hm = Hinv;
        double xa = (Hm[0] * p[0] + Hm[1] * p[1] + Hm[2]) / (Hm[6] * p[0] + Hm[7] * p[1] + Hm[8]);
        double ya = (Hm[3] * p[0] + Hm[4] * p[1] + Hm[5]) / (Hm[6] * p[0] + Hm[7] * p[1] + Hm[8]);
        const double d1 = (p[3] - xa) * (p[3] - xa) + (p[4] - ya) * (p[4] - ya);
        xa = (Hi[0] * p[3] + Hi[1] * p[4] + Hi[2]) / (Hi[6] * p[3] + Hi[7] * p[4] + Hi[8]);
        ya = (Hi[3] * p[3] + Hi[4] * p[4] + Hi[5]) / (Hi[6] * p[3] + Hi[7] * p[4] + Hi[8]);
        const double d2 = (p[0] - xa) * (p[0] - xa) + (p[1] - ya) * (p[1] - ya);
        if ((d1 <= err_sq) && (d2 <= err_sq)) ok++;
      }
      if (ok < MIN_POINTS) cur.clear();
    }
    // H_LAF_check with HDsSymMax on the three frame points
    if (affineFerror > 0 && laf) {
      std::vector<int> good;
      for (int i : cur) {
        const double *f = laf + (size_t)i * 14;
        double u[18], err[3];
        u[0] = f[0]; u[1] = f[1]; u[2] = 1.0;
        u[3] = f[7]; u[4] = f[8]; u[5] = 1.0;
        u[6] = u[0] + ks * f[3] * f[6]; u[7] = u[1] + ks * f[5] * f[6]; u[8] = 1.0;
        u[9] = u[3] + ks * f[10] * f[13]; u[10] = u[4] + ks * f[12] * f[13]; u[11] = 1.0;
        u[12] = u[0] + ks * f[2] * f[6]; u[13] = u[1] + ks * f[4] * f[6]; u[14] = 1.0;
        u[15] = u[3] + ks * f[9] * f[13]; u[16] = u[4] + ks * f[11] * f[13]; u[17] = 1.0;
        HDsSymMax(nullptr, u, Hloran, err, 3);
        const double sumErr = std::sqrt(err[0] + err[1] + err[2]);
        if (!(sumErr > affineFerror)) good.push_back(i);
      }
      cur.swap(good);
    }
  }
  if ((int)cur.size() < MIN_POINTS) cur.clear();
  for (int i : cur) mask[i] = 1;
  return (int)cur.size();
}

// LORANSACFiltering for the homography branch (useF = 0), matching.cpp:637-805: degensac LO-RANSAC,
// H -> row-major img1->img2 by inv(H^T), NaiveHCheck (10 px, :1014-1043), H_LAF_check (:250-308).
// mask[i] = 1 for the correspondences that survive every check.
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// clock of the MODS_RANSAC_PROF lines: wall time, or the calling thread's CPU time with MODS_RANSAC_PROF=cpu (ransac.hip: rs_now_us)
static double prof_ms() {
  if (ransac_profile_mode() == 2) { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
  return now_ms();
}

int mods_loransac_h(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask, double *H_out,
                    int *n_inliers, int *stats3) {
  const double t_enter = prof_ms();
  if (!par || !mask || !H_out || !n_inliers || (n > 0 && !u6)) { set_error("loransac_h: null argument"); return MODS_E_ARG; }
  *n_inliers = 0;
  for (int i = 0; i < 9; i++) H_out[i] = -1;   // TentativeCorrespListExt(): H[i] = -1 (matching.hpp:92-99)
  for (int i = 0; i < n; i++) mask[i] = 0;
  if (stats3) { stats3[0] = stats3[1] = stats3[2] = 0; }
  const int MIN_POINTS = 8;
  if (n < MIN_POINTS) return MODS_OK;
  int max_samples = par->max_samples;
  if (n <= 20) max_samples = 1000;
  std::vector<double> u2(u6, u6 + (size_t)n * 6);
  std::vector<unsigned char> inl2(n);
  std::vector<int> data_out(18);   // (the reference sizes it 18 n, matching.cpp:700; three counters are written)
  double Hloran[9];
  void (*f0)(const double *, const double *, const double *, double *, int);
  void (*f1)(const double *, const double *, const double *, double *, int, int *, int);
  void (*f2)(const double *, const double *, const double *, double *, int, int *, int);
  if (par->errorType == 0) { f0 = &HDs; f1 = &HDsi; f2 = &HDsidx; }
  else if (par->errorType == 1) { f0 = &HDsSymMax; f1 = &HDsiSymMax; f2 = &HDsSymidxMax; }
  else { f0 = &HDsSym; f1 = &HDsiSym; f2 = &HDsSymidx; }
  exp_ransacHcustom(u2.data(), n, par->err_threshold * par->err_threshold, par->confidence, max_samples, Hloran, inl2.data(), 4,
                    data_out.data(), 1, 0, nullptr, f0, f1, f2, par->doSymmCheck);   // (no residual rows: nobody reads them here)
  if (mods::ransac_failed()) return MODS_E_HIP;   // device failure inside the control loop; mods_last_error() says which
  const double t_post0 = prof_ms();
  if (stats3) { stats3[0] = data_out[0]; stats3[1] = data_out[1]; stats3[2] = data_out[2]; }
  *n_inliers = h_post_checks(u6, laf, n, inl2.data(), Hloran, par, mods::rs::simd_ops(), mask, H_out);
  if (ransac_profile_on()) fprintf(stderr, "loransac_h prof: whole %.0f us, checks after RANSAC %.0f us\n", 1e3 * (prof_ms() - t_enter), 1e3 * (prof_ms() - t_post0));
  return MODS_OK;
}

int mods_test_host_hchecks(const double *u6, const double *laf14, int n, const unsigned char *inl, const double *Hloran,
                           const mods_ransac_params *par, int lanes, unsigned char *mask, double *H_out) {
  if (!u6 || !inl || !Hloran || !par || !mask || !H_out || n < 0) return MODS_E_ARG;
  const mods::rs::SimdOps *ops = lanes ? mods::rs::simd_ops_lanes(lanes) : nullptr;
  if (lanes && !ops) return MODS_E_ARG;
  for (int i = 0; i < 9; i++) H_out[i] = -1;
  for (int i = 0; i < n; i++) mask[i] = 0;
  return h_post_checks(u6, laf14, n, inl, Hloran, par, ops, mask, H_out);
}

// useF branch of LORANSACFiltering (matching.cpp:711-726, 804-816): DEGENSAC + F_LAF_check (:192-249)
int mods_loransac_f(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask, double *F_out,
                    int *n_inliers, int *stats3) {
  if (!par || !mask || !F_out || !n_inliers || (n > 0 && !u6)) { set_error("loransac_f: null argument"); return MODS_E_ARG; }
  *n_inliers = 0;
  for (int i = 0; i < 9; i++) F_out[i] = -1;
  for (int i = 0; i < n; i++) mask[i] = 0;
  if (stats3) { stats3[0] = stats3[1] = stats3[2] = 0; }
  const int MIN_POINTS = 8;
  if (n < MIN_POINTS) return MODS_OK;
  std::vector<double> u2(u6, u6 + (size_t)n * 6);
  std::vector<unsigned char> inl2(n);
  std::vector<int> data_out((size_t)n * 18, 0);
  double Floran[9], HinF[9];
  double *resids = nullptr;
  int I_H = 0;
  void (*fds)(const double *, const double *, double *, int) = par->errorType == 0 ? &FDs : &FDsSym;
  void (*exfds)(const double *, const double *, double *, double *, int) = par->errorType == 0 ? &exFDs : &exFDsSym;
  exp_ransacFcustom(u2.data(), n, par->err_threshold * par->err_threshold, par->confidence, par->max_samples, Floran, inl2.data(),
                    data_out.data(), par->localOptimization, 0, &resids, HinF, &I_H, exfds, fds, par->doSymmCheck);
  free(resids);
  if (mods::ransac_failed()) return MODS_E_HIP;
  if (stats3) { stats3[0] = data_out[0]; stats3[1] = data_out[1]; stats3[2] = I_H; }
  std::vector<int> cur;
  for (int i = 0; i < n; i++) if (inl2[i]) cur.push_back(i);
  const double affineFerror = par->LAFCoef * par->err_threshold;
  if (affineFerror > 0 && laf) {
    std::vector<int> good;
    const double ks = 3.0;   // k_sigma, matching.cpp:171
    for (int i : cur) {
      const double *f = laf + (size_t)i * 14;
      double u[18], err[3];
      u[0] = f[0]; u[1] = f[1]; u[2] = 1.0;
      u[3] = f[7]; u[4] = f[8]; u[5] = 1.0;
      u[6] = u[0] + ks * f[3] * f[6]; u[7] = u[1] + ks * f[5] * f[6]; u[8] = 1.0;
      u[9] = u[3] + ks * f[10] * f[13]; u[10] = u[4] + ks * f[12] * f[13]; u[11] = 1.0;
      u[12] = u[0] + ks * f[2] * f[6]; u[13] = u[1] + ks * f[4] * f[6]; u[14] = 1.0;
      u[15] = u[3] + ks * f[9] * f[13]; u[16] = u[4] + ks * f[11] * f[13]; u[17] = 1.0;
      fds(u, Floran, err, 3);
      const double sumErr = std::sqrt(err[0]) + std::sqrt(err[1]) + std::sqrt(err[2]);
      if (!(sumErr > affineFerror)) good.push_back(i);
    }
    cur.swap(good);
  }
  if ((int)cur.size() < MIN_POINTS) cur.clear();
  for (int i : cur) mask[i] = 1;
  *n_inliers = (int)cur.size();
  for (int i = 0; i < 9; i++) F_out[i] = Floran[i];     // ransac_corresp.H[i] = Hloran[i], matching.cpp:814-815
  return MODS_OK;
}

// ---- one pair end to end -------------------------------------------------------------------------------

int mods_duplicate_filter_gpu(mods_ctx *c, mods_tentative *tent, double *u6, double *laf, int n, double r, int mode, int *n_out, int *on_device) {
  if (!c || !n_out || (n > 0 && (!tent || !u6 || !laf))) { set_error("duplicate_filter_gpu: null argument"); return MODS_E_ARG; }
  *n_out = n;
  if (on_device) *on_device = 0;
  if (r <= 0 || n <= 0) return MODS_OK;
  if (n > c->max_cand || mode < 0 || mode > 3) return mods_duplicate_filter(tent, u6, laf, n, r, mode, n_out);
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = match_ensure_buffers(c))) return rc;
  if (!c->m_tent2) MODS_HIP_CHECK(hipMalloc(&c->m_tent2, tent_bytes(((size_t)c->max_cand + 127) & ~(size_t)63) + 64));
  std::vector<char> stage(tent_bytes((size_t)n));
  memcpy(stage.data(), tent, sizeof(mods_tentative) * n);
  memcpy(stage.data() + tent_u6_off(n), u6, sizeof(double) * 6 * n);
  memcpy(stage.data() + tent_laf_off(n), laf, sizeof(double) * 14 * n);
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  MODS_HIP_CHECK(mods::copy_wait(c->stream, c->m_tent, stage.data(), stage.size(), hipMemcpyHostToDevice));
  c->m_count[0] = n;
  const DupJob job = {(const char *)c->m_tent, (char *)c->m_tent2, c->m_count, c->m_count + 64, c->m_count + 128};
  if ((rc = dup_filter_dev(c, &job, 1, n, r, mode))) return rc;
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  if (((volatile int *)c->m_count)[128] != 0) return mods_duplicate_filter(tent, u6, laf, n, r, mode, n_out);
  const int m = ((volatile int *)c->m_count)[64];
  if (m > 0) {
    MODS_HIP_CHECK(mods::copy_wait(c->stream, stage.data(), c->m_tent2, tent_bytes((size_t)m), hipMemcpyDeviceToHost));
    memcpy(tent, stage.data(), sizeof(mods_tentative) * m);
    memcpy(u6, stage.data() + tent_u6_off(m), sizeof(double) * 6 * m);
    memcpy(laf, stage.data() + tent_laf_off(m), sizeof(double) * 14 * m);
  }
  *n_out = m;
  if (on_device) *on_device = 1;
  return MODS_OK;
}

// DuplicateFiltering ahead of RANSAC runs on the device, behind the search (dedup.hip)
static bool dedup_on_device(const mods_pair_params *par) {
  return par->dup_before_ransac && par->dup_dist > 0 && par->dup_mode >= 0 && par->dup_mode <= 3;
}

// GPU half of a pair: detect + describe both images, match, bring the tentatives to the host.
int mods_pair_gpu_stage(mods_ctx *c, const float *img_dev, int w, int h, int stride, const mods_pair_params *par,
                        mods_pair_result *res, std::vector<mods_tentative> *tent, std::vector<double> *u6, std::vector<double> *laf) {
  if (!c || !img_dev || !par || !res) { set_error("match_pair: null argument"); return MODS_E_ARG; }
  if (c->batch < 2) { set_error("match_pair needs a context created with batch >= 2"); return MODS_E_ARG; }
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  int rc;
  const double t0 = now_ms();
  if ((rc = mods_detect_describe_dev(c, img_dev, 2, w, h, stride, &par->det, &par->desc, res->n_detected, res->n_described))) return rc;
  const double t1 = now_ms();
  res->ms_detect_describe = t1 - t0;
  if ((rc = match_run(c, c->regions_dev, res->n_described[0], c->regions_dev + c->max_cand, res->n_described[1],
                      par->fginn_ratio, par->contradDist, par->nn))) return rc;
  const bool dedup = dedup_on_device(par);
  if (dedup) {
    if (!c->m_tent2) MODS_HIP_CHECK(hipMalloc(&c->m_tent2, tent_bytes(((size_t)c->max_cand + 127) & ~(size_t)63) + 64));
    const DupJob job = {(const char *)c->m_tent, (char *)c->m_tent2, c->m_count, c->m_count + 64, c->m_count + 128};
    if ((rc = dup_filter_dev(c, &job, 1, res->n_described[0], par->dup_dist, par->dup_mode))) return rc;
  }
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  int n = *(volatile int *)c->m_count;
  res->n_tentatives = n;
  if (n > c->max_cand) { set_error("tentative list overflow"); return MODS_E_CAPACITY; }
  const bool filtered = dedup && ((volatile int *)c->m_count)[128] == 0;
  if (filtered) n = ((volatile int *)c->m_count)[64];
  tent->resize(n); u6->resize((size_t)n * 6); laf->resize((size_t)n * 14);
  if (filtered) {          // the kept correspondences in their sorted order; the verify stage sees n_unique == list length and does not filter again
    res->n_unique = n;
    if (n > 0) {
      std::vector<char> stage(tent_bytes((size_t)n));
      MODS_HIP_CHECK(hipMemcpyAsync(stage.data(), c->m_tent2, stage.size(), hipMemcpyDeviceToHost, c->stream));
      MODS_HIP_CHECK(mods::stream_wait(c->stream));
      memcpy(tent->data(), stage.data(), sizeof(mods_tentative) * n);
      memcpy(u6->data(), stage.data() + tent_u6_off(n), sizeof(double) * 6 * n);
      memcpy(laf->data(), stage.data() + tent_laf_off(n), sizeof(double) * 14 * n);
    }
  } else if ((rc = mods_match_copy_out(c, n, tent->data(), u6->data(), laf->data()))) return rc;
  res->ms_match = now_ms() - t1;
  return MODS_OK;
}

constexpr size_t kPinArena = (size_t)24 << 20;   // pinned staging of a batch's tentative lists

// 8-bit grey -> float (the ImageRepresentation constructor's convertTo(CV_32F), imagerepresentation.cpp:293-302): exact
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char *__restrict__ src, float *__restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const uchar4 v = ((const uchar4 *)src)[i];
    ((float4 *)dst)[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  }
}

// GPU half of up to batch/2 pairs in one pass: the images of all pairs go through the pyramid / detector /
// describe kernels as ONE batch (launches n times larger, the many tiny launches of the small octaves amortised
// over n pairs), then every pair is matched on its own.  img[i]: [2][h][w] of pair i; kinds[i] (NULL = all 0):
// 0 fp32 in HBM, 1 fp32 in (pinned) host memory, 2 8-bit grey in (pinned) host memory - host images are uploaded on
// the context's stream, so the transfer of one worker overlaps the kernels of the others.
// One batch of a synthetic blob lattice through detect / describe and one match: allocates every pool a batch of n_img
// images of w x h needs (they are sized by the geometry and the context's capacities) and loads every kernel of the path.
int mods_ctx_warmup(mods_ctx *c, int n_img, int w, int h, const mods_pair_params *par) {
  if (!c || !par || n_img < 1 || n_img > c->batch) { set_error("warmup: bad argument"); return MODS_E_ARG; }
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("warmup: image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  if (!c->u8_stage_dev) MODS_HIP_CHECK(hipMalloc(&c->u8_stage_dev, (size_t)c->max_w * c->max_h * c->batch + 16));
  if (!c->pin_arena) { MODS_HIP_CHECK(hipHostMalloc(&c->pin_arena, kPinArena)); c->pin_arena_cap = kPinArena; }
  std::vector<float> img((size_t)w * h);
  // blobs every 14 px on top of blobs every 90 px: some ten thousand regions of both patch tiers on a 2-megapixel image; on larger
  // images the lattice is stretched so that the count stays there (a 4096 x 4096 image at the 14-px period overflows the lists)
  const float f = sqrtf(std::min(1.0f, 1920.f * 1080.f / ((float)w * (float)h)));
  const float f1 = 0.22f * f, f2 = 0.035f * f;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      img[(size_t)y * w + x] = 128.f + 70.f * sinf(f1 * x) * sinf(f1 * y) + 50.f * sinf(f2 * x + 1.f) * sinf(f2 * y);
  const size_t plane = (size_t)w * h;
  MODS_HIP_CHECK(mods::copy_wait(c->stream, c->input_dev, img.data(), sizeof(float) * plane, hipMemcpyHostToDevice));
  for (int i = 1; i < n_img; i++)
    MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev + plane * i, c->input_dev, sizeof(float) * plane, hipMemcpyDeviceToDevice, c->stream));
  std::vector<int> nd(n_img), nr(n_img);
  int rc = mods_detect_describe_dev(c, c->input_dev, n_img, w, h, w, &par->det, &par->desc, nd.data(), nr.data());
  if (rc) return rc;
  const int last = n_img - 1;
  if ((rc = match_ensure_buffers(c, std::min(16, std::max(1, n_img / 2))))) return rc;     // the searches of a batch's pairs run as one group
  if ((rc = match_run(c, c->regions_dev, nr[0], c->regions_dev + (size_t)last * c->max_cand, nr[last], par->fginn_ratio, par->contradDist, par->nn))) return rc;
  if (dedup_on_device(par)) {
    // the duplicate filter's scratch for the lists of a whole batch and its kernels (a first hipMalloc inside the running pipeline
    // would synchronise the device): the filter runs once over the warm-up search's list, repeated as every job of a batch
    if (!c->m_tent2) MODS_HIP_CHECK(hipMalloc(&c->m_tent2, tent_bytes(((size_t)c->max_cand + 127) & ~(size_t)63) + 64));
    const int n_jobs = std::min(DUP_MAX_JOBS, std::max(1, n_img / 2));
    std::vector<DupJob> jobs(n_jobs, DupJob{(const char *)c->m_tent, (char *)c->m_tent2, c->m_count, c->m_count + 64, c->m_count + 128});
    if ((rc = dup_filter_dev(c, jobs.data(), 1, nr[0], par->dup_dist, par->dup_mode))) return rc;     // kernels + the single-pair path
    if (n_jobs > 1) {   // the allocation for a batch's jobs (every job of this call would write the same output: run none of them twice)
      MODS_HIP_CHECK(mods::stream_wait(c->stream));
      if ((rc = dup_filter_reserve(c, n_jobs))) return rc;
    }
  }
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

int mods_pairs_gpu_stage(mods_ctx *c, const void *const *img, const int *kinds, int n_pairs, int w, int h, const mods_pair_params *par,
                         mods_pair_result **res, std::vector<mods_tentative> **tent, std::vector<double> **u6, std::vector<double> **laf) {
  if (!c || !img || !par || !res || n_pairs < 1) { set_error("match_pairs: null argument"); return MODS_E_ARG; }
  if (n_pairs == 1 && (!kinds || kinds[0] == 0)) return mods_pair_gpu_stage(c, (const float *)img[0], w, h, w, par, res[0], tent[0], u6[0], laf[0]);
  if (c->batch < 2 * n_pairs) { set_error("match_pairs: context batch %d < %d images", c->batch, 2 * n_pairs); return MODS_E_ARG; }
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("match_pairs: image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  const size_t plane2 = (size_t)2 * w * h;
  const int n_img = 2 * n_pairs;
  for (int i = 0; i < n_pairs; i++) {
    memset(res[i], 0, sizeof(*res[i]));
    for (int q = 0; q < 9; q++) res[i]->H[q] = -1;
    const int kind = kinds ? kinds[i] : 0;
    if (kind == 2) {
      if (!c->u8_stage_dev) MODS_HIP_CHECK(hipMalloc(&c->u8_stage_dev, (size_t)c->max_w * c->max_h * c->batch + 16));
      unsigned char *st = c->u8_stage_dev + plane2 * i;
      if ((plane2 & 3) || ((uintptr_t)st & 3)) { set_error("match_pairs: 8-bit input needs w*h*2 divisible by 4"); return MODS_E_ARG; }
      // (reading page-locked host images from the conversion kernel itself - no staging copy - was measured: 610 against 636
      // pairs/s, the kernel's waves sit on PCIe reads; the copy engine path stays)
      MODS_HIP_CHECK(hipMemcpyAsync(st, img[i], plane2, hipMemcpyHostToDevice, c->stream));
      const unsigned char *src = st;
      hipLaunchKernelGGL(u8_to_f32_kernel, dim3(1024), dim3(256), 0, c->stream, src, c->input_dev + plane2 * i, plane2 / 4);
    } else {
      MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev + plane2 * i, img[i], sizeof(float) * plane2,
                                    kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    }
  }
  MODS_HIP_CHECK(hipGetLastError());
  std::vector<int> nd(n_img), nr(n_img);
  int rc;
  const double t0 = now_ms();
  if ((rc = mods_detect_describe_dev(c, c->input_dev, n_img, w, h, w, &par->det, &par->desc, nd.data(), nr.data()))) return rc;
  const double t1 = now_ms();
  // The tentative lists of the batch go to the host through a pinned arena: the packed list of a pair (tentatives |
  // correspondences | frames) is ONE transfer queued behind its match kernels, the stream is synchronised once per pair for the COUNT only (4 bytes) and once
  // per batch for the lists; a pair that does not fit the arena takes the direct (pageable, synchronous) path.
  if (!c->pin_arena) { MODS_HIP_CHECK(hipHostMalloc(&c->pin_arena, kPinArena)); c->pin_arena_cap = kPinArena; }
  // every pair's search is queued without waiting: the packed list of pair i goes to its own segment of a device arena (a
  // list is at most as long as the query list), its length to slot i of the pinned counter array.  Then ONE synchronisation
  // for the lengths, the transfers of exactly those bytes, and one more for the lists (before: a synchronisation per pair).
  if (n_pairs > 63) { set_error("match_pairs: at most 63 pairs per batch"); return MODS_E_ARG; }
  std::vector<size_t> seg(n_pairs + 1, 0);
  for (int i = 0; i < n_pairs; i++) seg[i + 1] = seg[i] + ((tent_bytes((size_t)std::max(nr[2 * i], 1)) + 255) & ~(size_t)255);
  if ((rc = match_ensure_buffers(c))) return rc;
  const bool dedup = dedup_on_device(par);
  const bool direct = dedup && seg[n_pairs] <= c->pin_arena_cap;
  // (the second half of the arena takes the filtered lists of the device duplicate filter)
  if (2 * seg[n_pairs] > c->m_tent_batch_cap) {
    if (c->m_tent_batch) MODS_HIP_CHECK(hipFree(c->m_tent_batch));
    c->m_tent_batch = nullptr; c->m_tent_batch_cap = 0;
    MODS_HIP_CHECK(hipMalloc(&c->m_tent_batch, 2 * seg[n_pairs] + seg[n_pairs] / 2));
    c->m_tent_batch_cap = 2 * seg[n_pairs] + seg[n_pairs] / 2;
  }
  const double tm0 = now_ms();
  // the searches of the batch's pairs in grouped launches (csrc/match.hip: match_run_group), up to 16 pairs per set of launches
  for (int i0 = 0; i0 < n_pairs; i0 += 16) {
    const int g = std::min(16, n_pairs - i0);
    const mods_region *qv[16], *tv[16];
    int nq[16], nt[16];
    mods_tentative *to[16];
    int *co[16];
    for (int e = 0; e < g; e++) {
      const int i = i0 + e;
      mods_pair_result *r = res[i];
      r->n_detected[0] = nd[2 * i]; r->n_detected[1] = nd[2 * i + 1];
      r->n_described[0] = nr[2 * i]; r->n_described[1] = nr[2 * i + 1];
      r->ms_detect_describe = (t1 - t0) / n_pairs;
      qv[e] = c->regions_dev + (size_t)(2 * i) * c->max_cand; nq[e] = nr[2 * i];
      tv[e] = c->regions_dev + (size_t)(2 * i + 1) * c->max_cand; nt[e] = nr[2 * i + 1];
      to[e] = (mods_tentative *)(c->m_tent_batch + seg[i]);
      co[e] = c->m_count + 1 + i;
    }
    if ((rc = match_run_group(c, g, qv, nq, tv, nt, to, co, par->fginn_ratio, par->contradDist, par->nn))) return rc;
  }
  if (dedup) {     // the lists of the whole batch through the device duplicate filter in one set of launches
    // when the batch's lists fit the pinned arena (they do unless the images are very large) the filter's compaction writes the
    // kept lists straight into it - host memory the device can address - at the segments' offsets: no copy launch per pair and one
    // synchronisation per batch
    std::vector<DupJob> jobs(n_pairs);
    int grid_n = 1;
    for (int i = 0; i < n_pairs; i++) {
      char *dst = direct ? c->pin_arena + seg[i] : c->m_tent_batch + seg[n_pairs] + seg[i];
      jobs[i] = {c->m_tent_batch + seg[i], dst, c->m_count + 1 + i, c->m_count + 64 + 1 + i, c->m_count + 128 + 1 + i};
      grid_n = std::max(grid_n, nr[2 * i]);
    }
    if ((rc = dup_filter_dev(c, jobs.data(), n_pairs, grid_n, par->dup_dist, par->dup_mode))) return rc;
  }
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  std::vector<size_t> off(n_pairs, (size_t)-1);
  size_t used = direct ? seg[n_pairs] : 0;       // (lists that were not filtered on the device go behind the segments)
  bool copies = false;
  for (int i = 0; i < n_pairs; i++) {
    int n = ((volatile int *)c->m_count)[1 + i];
    res[i]->n_tentatives = n;
    if (n > c->max_cand || n > nr[2 * i]) { set_error("tentative list overflow"); return MODS_E_CAPACITY; }
    // filtered on the device: the kept correspondences come over, the verify stage sees n_unique == list length and does not filter again
    const bool filtered = dedup && ((volatile int *)c->m_count)[128 + 1 + i] == 0;
    const char *list = c->m_tent_batch + seg[i];
    if (filtered) { n = ((volatile int *)c->m_count)[64 + 1 + i]; res[i]->n_unique = n; list = c->m_tent_batch + seg[n_pairs] + seg[i]; }
    tent[i]->resize(n); u6[i]->resize((size_t)n * 6); laf[i]->resize((size_t)n * 14);
    if (n > 0) {
      const size_t bytes = tent_bytes((size_t)n);
      if (filtered && direct) off[i] = seg[i];          // already in the arena
      else if (used + bytes <= c->pin_arena_cap) {
        MODS_HIP_CHECK(hipMemcpyAsync(c->pin_arena + used, list, bytes, hipMemcpyDeviceToHost, c->stream));
        off[i] = used; used += (bytes + 15) & ~(size_t)15;
        copies = true;
      } else {      // a list that does not fit the arena: the direct (pageable, synchronous) path
        std::vector<char> stage(bytes);
        MODS_HIP_CHECK(mods::copy_wait(c->stream, stage.data(), list, bytes, hipMemcpyDeviceToHost));
        memcpy(tent[i]->data(), stage.data(), sizeof(mods_tentative) * n);
        memcpy(u6[i]->data(), stage.data() + tent_u6_off(n), sizeof(double) * 6 * n);
        memcpy(laf[i]->data(), stage.data() + tent_laf_off(n), sizeof(double) * 14 * n);
      }
    }
  }
  if (copies) MODS_HIP_CHECK(mods::stream_wait(c->stream));
  const double tm1 = now_ms();
  for (int i = 0; i < n_pairs; i++) {
    res[i]->ms_match = (tm1 - tm0) / n_pairs;
    if (off[i] == (size_t)-1) continue;
    const size_t n = tent[i]->size();
    const char *a = c->pin_arena + off[i];
    memcpy(tent[i]->data(), a, sizeof(mods_tentative) * n);
    memcpy(u6[i]->data(), a + tent_u6_off(n), sizeof(double) * 6 * n);
    memcpy(laf[i]->data(), a + tent_laf_off(n), sizeof(double) * 14 * n);
  }
  return MODS_OK;
}

// Host-driven half: duplicate filtering + LO-RANSAC (hypotheses scored on `device`) + checks.
// The verification half of one step of the reference's loop (mods.cpp:278-368), in place on (tent, u6, laf):
//   [DuplicateFiltering] doBeforeRANSAC = 1: DuplicateFiltering on the tentatives, then LORANSACFiltering;
//   doBeforeRANSAC = 0: LORANSACFiltering on every tentative, then DuplicateFiltering on the VERIFIED list (mods.cpp:357-368;
//   TrueMatch1st, which also drives the minMatches stop, is the size of the de-duplicated list).
// On return the first *n_verified entries of the three arrays are the verified correspondences in output order;
// *n_unique = the size of the list RANSAC ran on.
// HMatrixFiltering, matching.cpp:917-1012: the reference stacks (second image point, first image point) and hands the
// column-major H to the error function; th = (float)(err_threshold^2) compared in double
int mods_hmatrix_filter(const double *u6, int n, const double *H_rowmajor, const mods_ransac_params *par, unsigned char *mask, int *n_true) {
  if (!par || !H_rowmajor || !n_true || (n > 0 && (!u6 || !mask))) { set_error("hmatrix_filter: null argument"); return MODS_E_ARG; }
  *n_true = 0;
  if (n <= 0) return MODS_OK;
  std::vector<double> u2((size_t)n * 6), d(n);
  for (int i = 0; i < n; i++) {
    const double *s = u6 + (size_t)i * 6;
    double *q = &u2[(size_t)i * 6];
    q[0] = s[3]; q[1] = s[4]; q[2] = 1.; q[3] = s[0]; q[4] = s[1]; q[5] = 1.;
  }
  const double *M = H_rowmajor;
  const double Hc[9] = {M[0], M[3], M[6], M[1], M[4], M[7], M[2], M[5], M[8]};   // Hready[0], [3], [6] <- first row of the file
  if (par->errorType == 0) HDs(nullptr, u2.data(), Hc, d.data(), n);
  else if (par->errorType == 1) HDsSymMax(nullptr, u2.data(), Hc, d.data(), n);
  else HDsSym(nullptr, u2.data(), Hc, d.data(), n);
  const float th = (float)(par->err_threshold * par->err_threshold);
  int c = 0;
  for (int i = 0; i < n; i++) { mask[i] = d[i] <= th ? 1 : 0; c += mask[i]; }
  *n_true = c;
  return MODS_OK;
}

int mods_verify_tentatives(int device, const mods_pair_params *par, mods_tentative *tent, double *u6, double *laf, int n,
                           int *n_unique, int *n_verified, double *H_out, int *stats3, double *ms_dup, double *ms_ransac) {
  return mods_verify_tentatives_ex(device, par, tent, u6, laf, n, n_unique, n_verified, H_out, stats3, nullptr, ms_dup, ms_ransac);
}

int mods_verify_tentatives_ex(int device, const mods_pair_params *par, mods_tentative *tent, double *u6, double *laf, int n,
                              int *n_unique, int *n_verified, double *H_out, int *stats3, int *gt3, double *ms_dup, double *ms_ransac) {
  if (gt3) gt3[0] = gt3[1] = gt3[2] = 0;
  if (!par || !n_unique || !n_verified || (n > 0 && (!tent || !u6 || !laf))) { set_error("verify_tentatives: null argument"); return MODS_E_ARG; }
  int rc;
  const double t0 = now_ms();
  int nu = n;
  if (par->dup_before_ransac && n > 0)
    if ((rc = mods_duplicate_filter(tent, u6, laf, n, par->dup_dist, par->dup_mode, &nu))) return rc;
  *n_unique = nu;
  const double t1 = now_ms();
  int stats[3] = {0, 0, 0}, ninl = 0;
  mods_ransac_set_device(device);
  std::vector<unsigned char> mask(nu > 0 ? nu : 1);
  double H[9];
  if (par->ransac.groundTruth) {
    // GR_TRUTH, mods.cpp:292-320: HMatrixFiltering of all unique tentatives (TrueMatch1st); with doBothRANSACgroundTruth the
    // verified list is instead the LORANSAC inliers that the ground truth confirms
    int n_true = 0;
    if ((rc = mods_hmatrix_filter(u6, nu, par->ransac.gtH, &par->ransac, mask.data(), &n_true))) return rc;
    if (gt3) gt3[0] = n_true;
    ninl = n_true;
    if (par->ransac.groundTruth >= 2) {
      std::vector<unsigned char> mr(nu > 0 ? nu : 1);
      double Hr[9];
      if ((rc = mods_loransac_h(u6, laf, nu, &par->ransac, mr.data(), Hr, &ninl, stats))) return rc;
      std::vector<double> ur((size_t)(ninl > 0 ? ninl : 1) * 6);
      std::vector<unsigned char> mt(ninl > 0 ? ninl : 1);
      int q = 0;
      for (int i = 0; i < nu; i++)
        if (mr[i]) { memcpy(&ur[(size_t)q * 6], &u6[(size_t)i * 6], 6 * sizeof(double)); q++; }
      int n_tr = 0;
      if ((rc = mods_hmatrix_filter(ur.data(), ninl, par->ransac.gtH, &par->ransac, mt.data(), &n_tr))) return rc;
      if (gt3) { gt3[1] = ninl; gt3[2] = n_tr; }
      q = 0;
      for (int i = 0; i < nu; i++) { mask[i] = mr[i] ? mt[q++] : 0; }
      ninl = n_tr;
    }
    memcpy(H, par->ransac.gtH, sizeof(H));     // true_corresp.H = the ground truth, row-major again (matching.cpp:1002-1010)
  } else {
    if (par->ransac.useF) rc = mods_loransac_f(u6, laf, nu, &par->ransac, mask.data(), H, &ninl, stats);
    else rc = mods_loransac_h(u6, laf, nu, &par->ransac, mask.data(), H, &ninl, stats);
    if (rc) return rc;
  }
  const double t2 = now_ms();
  int m = 0;
  for (int i = 0; i < nu; i++)
    if (mask[i]) {
      if (m != i) {
        tent[m] = tent[i];
        memcpy(&u6[(size_t)m * 6], &u6[(size_t)i * 6], 6 * sizeof(double));
        memcpy(&laf[(size_t)m * 14], &laf[(size_t)i * 14], 14 * sizeof(double));
      }
      m++;
    }
  if (!par->dup_before_ransac && m > 0)
    if ((rc = mods_duplicate_filter(tent, u6, laf, m, par->dup_dist, par->dup_mode, &m))) return rc;
  *n_verified = m;
  if (H_out) memcpy(H_out, H, sizeof(H));
  if (stats3) memcpy(stats3, stats, sizeof(stats));
  const double t3 = now_ms();
  if (ms_dup) *ms_dup = (t1 - t0) + (t3 - t2);
  if (ms_ransac) *ms_ransac = t2 - t1;
  return MODS_OK;
}

int mods_pair_verify_stage(int device, const mods_pair_params *par, mods_pair_result *res, std::vector<mods_tentative> *tent,
                           std::vector<double> *u6, std::vector<double> *laf, double *matches_out, int max_matches) {
  int stats[3] = {0, 0, 0};
  // a list the GPU stage has already filtered (DuplicateFiltering on the device, dedup.hip) arrives with n_unique = its length
  mods_pair_params p2;
  if (!tent->empty() && res->n_unique == (int)tent->size() && par->dup_before_ransac) { p2 = *par; p2.dup_dist = 0; par = &p2; }
  const int rc = mods_verify_tentatives(device, par, tent->data(), u6->data(), laf->data(), (int)tent->size(), &res->n_unique,
                                        &res->n_inliers, res->H, stats, &res->ms_duplicates, &res->ms_ransac);
  if (rc) return rc;
  res->ransac_samples = stats[0]; res->ransac_lo = stats[1]; res->ransac_rejects = stats[2];
  if (matches_out)
    for (int m = 0; m < res->n_inliers && m < max_matches; m++) {
      const double *p = &(*u6)[(size_t)m * 6];
      matches_out[4 * m] = p[0]; matches_out[4 * m + 1] = p[1]; matches_out[4 * m + 2] = p[3]; matches_out[4 * m + 3] = p[4];
    }
  return MODS_OK;
}

int mods_match_pair_dev(mods_ctx *c, const float *img_dev, int w, int h, int stride, const mods_pair_params *par,
                        mods_pair_result *res, double *matches_out, int max_matches) {
  if (!c) { set_error("match_pair: null context"); return MODS_E_ARG; }
  int rc = mods_pair_gpu_stage(c, img_dev, w, h, stride, par, res, &c->h_tent, &c->h_u6, &c->h_laf);
  if (rc) return rc;
  return mods_pair_verify_stage(c->device, par, res, &c->h_tent, &c->h_u6, &c->h_laf, matches_out, max_matches);
}

// ---- single primitives -------------------------------------------------------------------
int mods_gauss_blur(mods_ctx *c, const float *src, int w, int h, float sigma, float *dst) {
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, src, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice, c->stream));
  int rc = launch_gauss_blur(c, c->input_dev, c->tmp_dev, w, h, 1, sigma);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(dst, c->tmp_dev, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

int mods_hessian_response(mods_ctx *c, const float *src, int w, int h, float norm, float *dst) {
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, src, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice, c->stream));
  int rc = launch_hessian_response(c, c->input_dev, c->tmp_dev, w, h, 1, norm);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(dst, c->tmp_dev, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

int mods_resize_half(mods_ctx *c, const float *src, int w, int h, float *dst, int *dw, int *dh) {
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  resize_half_dims(w, h, dw, dh);
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, src, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice, c->stream));
  int rc = launch_resize_half(c, c->input_dev, c->tmp_dev, w, h, *dw, *dh, 1);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(dst, c->tmp_dev, sizeof(float) * (size_t)(*dw) * (*dh), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

}  // extern "C"
