// Pair pipeline: overlaps the GPU stages of pair i+1 (detect, describe, match) with the host-driven
// verification of pair i (duplicate filtering, LO-RANSAC control loop), the way mods.cpp overlaps its two
// images with OpenMP tasks (mods.cpp:234-251) - here across pairs, with worker threads that each own a
// context.  Results come back in submission order.
#include "common.hpp"
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <algorithm>
#include <chrono>
#include <time.h>
#include <pthread.h>
#include <sched.h>
#include <dirent.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <map>
#include <set>
#include <tuple>

extern "C" int mods_ctx_create_ex(int device, int max_w, int max_h, int batch, int flags, mods_ctx **out);
extern "C" int mods_pair_gpu_stage(mods_ctx *c, const float *img_dev, int w, int h, int stride, const mods_pair_params *par,
                                   mods_pair_result *res, std::vector<mods_tentative> *tent, std::vector<double> *u6,
                                   std::vector<double> *laf);
extern "C" int mods_pairs_gpu_stage(mods_ctx *c, const void *const *img, const int *kinds, int n_pairs, int w, int h, const mods_pair_params *par,
                                    mods_pair_result **res, std::vector<mods_tentative> **tent, std::vector<double> **u6,
                                    std::vector<double> **laf);
extern "C" int mods_pair_verify_stage(int device, const mods_pair_params *par, mods_pair_result *res, std::vector<mods_tentative> *tent,
                                      std::vector<double> *u6, std::vector<double> *laf, double *matches_out, int max_matches);

namespace mods {

struct Job {
  long tag = 0;
  const void *img = nullptr;
  int kind = 0;                   // 0 fp32 in HBM, 1 fp32 on the host, 2 8-bit grey on the host
  mods_pair_result res;
  std::vector<mods_tentative> tent;
  std::vector<double> u6, laf;
  int rc = MODS_OK;
  std::string err;
  bool done = false;
};

}  // namespace mods

struct mods_pipeline {
  int device = 0, w = 0, h = 0;
  mods_pair_params par;
  std::vector<mods_ctx *> ctxs;
  std::vector<std::thread> gpu_threads, verify_threads;
  std::mutex mu;
  std::condition_variable cv_gpu, cv_verify, cv_done, cv_space;
  std::deque<std::shared_ptr<mods::Job>> q_gpu, q_verify, q_order;
  int max_in_flight = 8;
  int pairs_per_batch = 1;        // pairs a GPU worker pushes through detect/describe in one batch of launches
  bool stop = false;
  int ready = 0, warm_rc = MODS_OK;   // workers that finished their warm-up (mods_pipeline_create_ex waits for all of them)
  std::string warm_err;
  std::condition_variable cv_ready;
  // CPU time the workers' own threads spent inside their stages (CLOCK_THREAD_CPUTIME_ID; the RANSAC task pool's helper threads,
  // which the verify stage of a hard pair spreads its model fits over, are not in it: the process clock is)
  std::atomic<long long> cpu_gpu_ns{0}, cpu_verify_ns{0};
};
static long long thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; }

using namespace mods;

extern "C" int mods_ransac_warmup(int device, int len);
extern "C" int mods_ctx_warmup(mods_ctx *c, int n_img, int w, int h, const mods_pair_params *par);

static void worker_ready(mods_pipeline *p, int rc) {
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (rc && p->warm_rc == MODS_OK) { p->warm_rc = rc; p->warm_err = mods_last_error(); }
    p->ready++;
  }
  p->cv_ready.notify_all();
}

static void gpu_worker(mods_pipeline *p, mods_ctx *ctx) {
  (void)hipSetDevice(p->device);
  pthread_setname_np(pthread_self(), "mods-gpu");
  wait_mode_for_worker(50000);    // a batch takes tens of milliseconds: 50 us between looks
  // every pool a full batch needs (pyramid planes, candidate / region / matcher buffers, code objects) is allocated now: a
  // hipMalloc inside the running pipeline synchronises the whole device
  worker_ready(p, mods_ctx_warmup(ctx, 2 * p->pairs_per_batch, p->w, p->h, &p->par));
  for (;;) {
    std::vector<std::shared_ptr<Job>> js;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_gpu.wait(lk, [&] { return p->stop || !p->q_gpu.empty(); });
      if (p->stop && p->q_gpu.empty()) return;
      // whatever is queued, up to the batch size: no waiting for a full batch
      while (!p->q_gpu.empty() && (int)js.size() < p->pairs_per_batch) { js.push_back(p->q_gpu.front()); p->q_gpu.pop_front(); }
    }
    const int n = (int)js.size();
    std::vector<const void *> imgs(n);
    std::vector<int> kinds(n);
    std::vector<mods_pair_result *> res(n);
    std::vector<std::vector<mods_tentative> *> tent(n);
    std::vector<std::vector<double> *> u6(n), laf(n);
    for (int i = 0; i < n; i++) { imgs[i] = js[i]->img; kinds[i] = js[i]->kind; res[i] = &js[i]->res; tent[i] = &js[i]->tent; u6[i] = &js[i]->u6; laf[i] = &js[i]->laf; }
    const long long c0 = thread_cpu_ns();
    const int rc = mods_pairs_gpu_stage(ctx, imgs.data(), kinds.data(), n, p->w, p->h, &p->par, res.data(), tent.data(), u6.data(), laf.data());
    p->cpu_gpu_ns.fetch_add(thread_cpu_ns() - c0, std::memory_order_relaxed);
    const std::string err = rc ? mods_last_error() : "";
    {
      std::lock_guard<std::mutex> lk(p->mu);
      for (auto &j : js) { j->rc = rc; j->err = err; p->q_verify.push_back(j); }
    }
    p->cv_verify.notify_all();
  }
}

static void verify_worker(mods_pipeline *p) {
  (void)hipSetDevice(p->device);
  pthread_setname_np(pthread_self(), "mods-verify");
  wait_mode_for_worker(20000);    // scoring launches are short but queue behind the other workers' kernels
  worker_ready(p, mods_ransac_warmup(p->device, 16384));   // this thread's scoring stream and workspace
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_verify.wait(lk, [&] { return p->stop || !p->q_verify.empty(); });
      if (p->stop && p->q_verify.empty()) return;
      j = p->q_verify.front(); p->q_verify.pop_front();
    }
    if (j->rc == MODS_OK) {
      const long long c0 = thread_cpu_ns();
      j->rc = mods_pair_verify_stage(p->device, &p->par, &j->res, &j->tent, &j->u6, &j->laf, nullptr, 0);
      p->cpu_verify_ns.fetch_add(thread_cpu_ns() - c0, std::memory_order_relaxed);
      if (j->rc) j->err = mods_last_error();
    }
    {
      std::lock_guard<std::mutex> lk(p->mu);
      j->done = true;
    }
    p->cv_done.notify_all();
  }
}


// (Rounds 4 and 5 carried a watch here that moved the HIP runtime's busy thread to SCHED_IDLE on request, MODS_RUNTIME_THREAD=idle.
// The thread was busy because it SPINS while a hipStreamQuery marker, a dependency between two streams or a graph launch is pending
// (tools/ubench/rt_thread_probe.hip); the pipeline no longer hands it any of the three - see stream_wait in capi.hip and the
// workers' contexts below - so the watch is gone: docs/history/r05_removed_paths.patch.)

extern "C" {

int mods_pipeline_create_ex(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                            int pairs_per_batch, mods_pipeline **out);

int mods_pipeline_create(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                         mods_pipeline **out) {
  return mods_pipeline_create_ex(device, w, h, par, gpu_workers, verify_workers, 1, out);
}

int mods_pipeline_capacity(const mods_pipeline *p) { return p ? p->max_in_flight : 0; }

// per-kernel HIP-event timing of the GPU workers' contexts (see mods_ctx_timing_*); call while nothing is in flight
int mods_pipeline_timing_enable(mods_pipeline *p, int stage_mask) {
  if (!p) return MODS_E_ARG;
  for (auto *c : p->ctxs) { int rc = mods_ctx_timing_enable(c, stage_mask); if (rc) return rc; rc = mods_ctx_timing_reset(c); if (rc) return rc; }
  return MODS_OK;
}
int mods_pipeline_timing_read(mods_pipeline *p, int stage, double *total_ms, int *launches, double *bytes) {
  if (!p) return MODS_E_ARG;
  double ms = 0, by = 0; int n = 0;
  for (auto *c : p->ctxs) {
    double m1 = 0, b1 = 0; int n1 = 0;
    const int rc = mods_ctx_timing_read(c, stage, &m1, &n1, &b1);
    if (rc) return rc;
    ms += m1; by += b1; n += n1;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (bytes) *bytes = by;
  return MODS_OK;
}

// detect + describe calls of the workers that were served by a graph replay so far (mods_ctx_graphs)
long mods_pipeline_graph_replays(mods_pipeline *p) {
  long n = 0;
  if (p) for (auto *c : p->ctxs) n += mods_ctx_graph_replays(c);
  return n;
}

// CPU seconds the GPU workers / the verify workers have spent inside their stages since the last reset (reset != 0 clears them)
int mods_pipeline_cpu_seconds(mods_pipeline *p, double *gpu_workers_s, double *verify_workers_s, int reset) {
  if (!p) return MODS_E_ARG;
  if (gpu_workers_s) *gpu_workers_s = p->cpu_gpu_ns.load() * 1e-9;
  if (verify_workers_s) *verify_workers_s = p->cpu_verify_ns.load() * 1e-9;
  if (reset) { p->cpu_gpu_ns.store(0); p->cpu_verify_ns.store(0); }
  return MODS_OK;
}

int mods_pipeline_create_ex(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                            int pairs_per_batch, mods_pipeline **out) {
  if (!par || !out || gpu_workers < 1 || verify_workers < 1 || gpu_workers > 8 || verify_workers > 32 || pairs_per_batch < 1 ||
      pairs_per_batch > 16) { set_error("pipeline: bad arguments"); return MODS_E_ARG; }
  std::unique_ptr<mods_pipeline> p(new mods_pipeline());
  p->device = device; p->w = w; p->h = h; p->par = *par;
  p->pairs_per_batch = pairs_per_batch;
  p->max_in_flight = 2 * gpu_workers * pairs_per_batch + 2 * verify_workers;
  for (int i = 0; i < gpu_workers; i++) {
    mods_ctx *c = nullptr;
    int rc = mods_ctx_create_ex(device, w, h, 2 * pairs_per_batch, 1, &c);   // non-blocking streams: the workers overlap on the GPU
    if (rc) { for (auto *q : p->ctxs) mods_ctx_destroy(q); return rc; }
    // One stream per worker, eager launches.  A worker's own side stream (the small octaves of the scale space, pyramid.hip) and the
    // replay of its launch chain as a hipGraph (capi.hip: dd_run) both put a dependency between two streams in front of the runtime,
    // and this runtime resolves such a dependency on its own thread, SPINNING until the awaited work is done (tools/ubench/
    // rt_thread_probe.hip: a fork / join over two streams keeps that thread at 66 % of a core for its duration, the same chain as a
    // graph launch at 98 %, one stream at 0.3 %): with six workers that thread never slept - the "one core per process" of rounds 4
    // and 5.  Six workers overlap on the GPU anyway, so the side stream buys the pipeline nothing: 895-906 pairs/s at 0.95 ms of
    // process CPU per pair on one stream against 841-862 at 2.0 ms with side stream + replay on the same box
    // (profiles/r05_pipeline_streams_ab.log).  MODS_PIPELINE_STREAMS=2 restores side stream + replay.
    static const int streams_env = getenv("MODS_PIPELINE_STREAMS") ? atoi(getenv("MODS_PIPELINE_STREAMS")) : 1;
    if (streams_env >= 2) (void)mods_ctx_graphs(c, 1);
    else (void)mods_ctx_pyramid_streams(c, 1);
    p->ctxs.push_back(c);
  }
  for (int i = 0; i < gpu_workers; i++) p->gpu_threads.emplace_back(gpu_worker, p.get(), p->ctxs[i]);
  for (int i = 0; i < verify_workers; i++) p->verify_threads.emplace_back(verify_worker, p.get());
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_ready.wait(lk, [&] { return p->ready == gpu_workers + verify_workers; });
  }
  if (p->warm_rc) {
    const int rc = p->warm_rc;
    const std::string err = p->warm_err;
    mods_pipeline_destroy(p.release());
    set_error("pipeline warm-up: %s", err.c_str());
    return rc;
  }
  *out = p.release();
  return MODS_OK;
}

// Queues one pair ([2][h][w] fp32 in HBM; must stay valid until its result has been fetched).  Blocks
// only while too many pairs are in flight.
static int submit_any(mods_pipeline *p, const void *img, int kind, long tag) {
  if (!p || !img) return MODS_E_ARG;
  auto j = std::make_shared<Job>();
  j->tag = tag; j->img = img; j->kind = kind;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_space.wait(lk, [&] { return (int)p->q_order.size() < p->max_in_flight; });
    p->q_gpu.push_back(j);
    p->q_order.push_back(j);
  }
  p->cv_gpu.notify_one();
  return MODS_OK;
}

int mods_pipeline_submit(mods_pipeline *p, const float *img_dev, long tag) { return submit_any(p, img_dev, 0, tag); }
// The same with the pair in host memory ([2][h][w], fp32 or 8-bit grey; pinned memory makes the upload asynchronous): the
// boundary of the reference's step loop, mods.cpp:184-383 (decoded images in host memory in, verified matches + H / F out).
int mods_pipeline_submit_host(mods_pipeline *p, const float *img_host, long tag) { return submit_any(p, img_host, 1, tag); }
int mods_pipeline_submit_host_u8(mods_pipeline *p, const unsigned char *img_host, long tag) { return submit_any(p, img_host, 2, tag); }

// Result of the oldest submitted pair (blocks until it is verified).  Returns MODS_E_ARG when nothing
// is in flight.
int mods_pipeline_next_matches(mods_pipeline *p, mods_pair_result *res, long *tag, double *matches_out, int max_matches) {
  if (!p || !res) return MODS_E_ARG;
  std::shared_ptr<Job> j;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->q_order.empty()) { set_error("pipeline: nothing in flight"); return MODS_E_ARG; }
    j = p->q_order.front();
    p->cv_done.wait(lk, [&] { return j->done; });
    p->q_order.pop_front();
  }
  p->cv_space.notify_one();
  *res = j->res;
  if (tag) *tag = j->tag;
  if (j->rc) set_error("%s", j->err.c_str());
  // the verify stage leaves the verified correspondences in the first n_inliers rows of the job's list
  if (!j->rc && matches_out)
    for (int m = 0; m < j->res.n_inliers && m < max_matches && (size_t)m * 6 + 5 < j->u6.size(); m++) {
      const double *q = &j->u6[(size_t)m * 6];
      matches_out[4 * m] = q[0]; matches_out[4 * m + 1] = q[1]; matches_out[4 * m + 2] = q[3]; matches_out[4 * m + 3] = q[4];
    }
  return j->rc;
}

int mods_pipeline_next(mods_pipeline *p, mods_pair_result *res, long *tag) { return mods_pipeline_next_matches(p, res, tag, nullptr, 0); }

void mods_pipeline_destroy(mods_pipeline *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv_gpu.notify_all(); p->cv_verify.notify_all();
  for (auto &t : p->gpu_threads) t.join();
  for (auto &t : p->verify_threads) t.join();
  for (auto *c : p->ctxs) mods_ctx_destroy(c);
  delete p;
}

}  // extern "C"
