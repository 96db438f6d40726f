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

extern "C" int mods_ctx_create_ex(int device, int max_w, int max_h, int batch, int flags, mods_ctx **out);
extern "C" int mods_pair_gpu_stage(mods_ctx *c, const float *img_dev, int w, int h, int stride, const mods_pair_params *par,
                                   mods_pair_result *res, std::vector<mods_tentative> *tent, std::vector<double> *u6,
                                   std::vector<double> *laf);
extern "C" int mods_pair_verify_stage(int device, const mods_pair_params *par, mods_pair_result *res, std::vector<mods_tentative> *tent,
                                      std::vector<double> *u6, std::vector<double> *laf, double *matches_out, int max_matches);

namespace mods {

struct Job {
  long tag = 0;
  const float *img = nullptr;
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
  bool stop = false;
};

using namespace mods;

static void gpu_worker(mods_pipeline *p, mods_ctx *ctx) {
  (void)hipSetDevice(p->device);
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_gpu.wait(lk, [&] { return p->stop || !p->q_gpu.empty(); });
      if (p->stop && p->q_gpu.empty()) return;
      j = p->q_gpu.front(); p->q_gpu.pop_front();
    }
    j->rc = mods_pair_gpu_stage(ctx, j->img, p->w, p->h, p->w, &p->par, &j->res, &j->tent, &j->u6, &j->laf);
    if (j->rc) j->err = mods_last_error();
    {
      std::lock_guard<std::mutex> lk(p->mu);
      p->q_verify.push_back(j);
    }
    p->cv_verify.notify_one();
  }
}

static void verify_worker(mods_pipeline *p) {
  (void)hipSetDevice(p->device);
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_verify.wait(lk, [&] { return p->stop || !p->q_verify.empty(); });
      if (p->stop && p->q_verify.empty()) return;
      j = p->q_verify.front(); p->q_verify.pop_front();
    }
    if (j->rc == MODS_OK) {
      j->rc = mods_pair_verify_stage(p->device, &p->par, &j->res, &j->tent, &j->u6, &j->laf, nullptr, 0);
      if (j->rc) j->err = mods_last_error();
    }
    {
      std::lock_guard<std::mutex> lk(p->mu);
      j->done = true;
    }
    p->cv_done.notify_all();
  }
}

extern "C" {

int mods_pipeline_create(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                         mods_pipeline **out) {
  if (!par || !out || gpu_workers < 1 || verify_workers < 1 || gpu_workers > 8 || verify_workers > 32) { set_error("pipeline: bad arguments"); return MODS_E_ARG; }
  std::unique_ptr<mods_pipeline> p(new mods_pipeline());
  p->device = device; p->w = w; p->h = h; p->par = *par;
  p->max_in_flight = 2 * (gpu_workers + verify_workers);
  for (int i = 0; i < gpu_workers; i++) {
    mods_ctx *c = nullptr;
    int rc = mods_ctx_create_ex(device, w, h, 2, 1, &c);   // non-blocking streams: the workers overlap on the GPU
    if (rc) { for (auto *q : p->ctxs) mods_ctx_destroy(q); return rc; }
    p->ctxs.push_back(c);
  }
  for (int i = 0; i < gpu_workers; i++) p->gpu_threads.emplace_back(gpu_worker, p.get(), p->ctxs[i]);
  for (int i = 0; i < verify_workers; i++) p->verify_threads.emplace_back(verify_worker, p.get());
  *out = p.release();
  return MODS_OK;
}

// Queues one pair ([2][h][w] fp32 in HBM; must stay valid until its result has been fetched).  Blocks
// only while too many pairs are in flight.
int mods_pipeline_submit(mods_pipeline *p, const float *img_dev, long tag) {
  if (!p || !img_dev) return MODS_E_ARG;
  auto j = std::make_shared<Job>();
  j->tag = tag; j->img = img_dev;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_space.wait(lk, [&] { return (int)p->q_order.size() < p->max_in_flight; });
    p->q_gpu.push_back(j);
    p->q_order.push_back(j);
  }
  p->cv_gpu.notify_one();
  return MODS_OK;
}

// Result of the oldest submitted pair (blocks until it is verified).  Returns MODS_E_ARG when nothing
// is in flight.
int mods_pipeline_next(mods_pipeline *p, mods_pair_result *res, long *tag) {
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
  return j->rc;
}

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
