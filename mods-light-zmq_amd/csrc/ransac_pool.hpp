// Host threads for the INDEPENDENT pieces of the DEGENSAC degenerate branch (innerFH: the evaluations of its 15 samples
// and the u2Fit refinements they trigger; rFtH: the innerFH calls of successive triggers).  Every task is the serial code
// on its own inputs, and the results are folded in the serial order afterwards, so what comes out is what one thread
// would have produced - only sooner.  The reference runs all of it on one core (DegUtils.c:233-584).
//
// One pool per process, started on first use: MODS_RANSAC_THREADS threads in total with the caller (default: the cores this
// process may use - affinity mask and cgroup quota - capped at 16; 1 = everything inline).  Round 6, 4096 x 4096 planar pair, 23 545
// correspondences (profiles/r06_degensac_pool.log): the degenerate branch's 105 sample evaluations and ~20 refinements per round of
// triggers are throughput work - 8 / 12 / 16 threads: innerFH 13.0 / 8.9 / 6.8 ms, the whole call 25.0 / 19.8 / 17.3 ms; with the samples' counts made on the device the ~20 refinements of a
// round are what is left: 16 / 24 threads: innerFH 5.2 / 4.1-5.5 ms, the whole call 12.0-12.7 / 11.2-13.8 ms over four leases - no
// gain that repeats, so the cap stays at 16.  A caller that finds the pool
// taken (several verification threads in a pipeline) runs its tasks inline: the pool never queues and never oversubscribes.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mods {
namespace rs {

inline int usable_cores() {
  int n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  long quota = 0, period = 100000;
  char a[64] = {0};
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                        // cgroup v2: "<quota|max> <period>"
    if (fscanf(f, "%63s %ld", a, &period) >= 1 && a[0] != 'm') quota = atol(a);
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // v1: quota (-1 = none) and period in two files
    if (fscanf(g, "%ld", &quota) != 1) quota = 0;
    fclose(g);
    if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 100000; fclose(h); }
  }
  if (quota > 0 && period > 0) { const int q = (int)((quota + period - 1) / period); if (q < n) n = q; }
  return n < 1 ? 1 : n;
}

class TaskPool {
 public:
  static TaskPool &get() { static TaskPool *p = new TaskPool(); return *p; }   // never destroyed: its threads outlive main()
  int threads() const { return n_threads_; }

  // fn(i) for every i in [0, n), on the pool's threads and the caller; returns when all are done.  Tasks must not call run().
  void run(int n, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    if (n == 1 || n_threads_ <= 1 || !busy_.try_lock()) { for (int i = 0; i < n; i++) fn(i); return; }
    start_workers();
    {
      std::lock_guard<std::mutex> lk(m_);   // no worker is inside work() here: the previous run waited for them
      fn_ = &fn; n_ = n;
      done_.store(0, std::memory_order_relaxed);
      next_.store(0, std::memory_order_relaxed);
      open_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    work();
    while (done_.load(std::memory_order_acquire) < n) cpu_relax();
    {
      std::lock_guard<std::mutex> lk(m_);   // a worker decides under this lock whether it joins: none joins from here on
      open_ = false;
    }
    while (in_work_.load(std::memory_order_acquire) > 0) cpu_relax();
    busy_.unlock();
    // a task that threw (bad_alloc in a fit, say) was caught where it ran - on a detached worker an escaping exception would end
    // the process, on the caller it would leave busy_ locked - and is rethrown here, after the pool is back in its idle state
    if (failed_.exchange(false, std::memory_order_acq_rel)) { std::exception_ptr e = error_; error_ = nullptr; std::rethrow_exception(e); }
  }

 private:
  TaskPool() {
    const char *e = getenv("MODS_RANSAC_THREADS");
    int n = e ? atoi(e) : usable_cores();
    if (!e && n > 16) n = 16;
    n_threads_ = n < 1 ? 1 : n;
  }
  static void cpu_relax() { __builtin_ia32_pause(); }
  void start_workers() {
    if (started_) return;
    started_ = true;
    for (int i = 1; i < n_threads_; i++) std::thread([this] { loop(); }).detach();
  }
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_acq_rel);
      if (i >= n_) break;
      try { (*fn_)(i); }
      catch (...) {
        std::lock_guard<std::mutex> lk(m_);
        if (!failed_.load(std::memory_order_relaxed)) { error_ = std::current_exception(); failed_.store(true, std::memory_order_release); }
      }
      done_.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void loop() {
    pthread_setname_np(pthread_self(), "mods-pool");
    unsigned seen = 0;
    for (;;) {
      // spin for a while (the tasks of one run are tens of microseconds to a millisecond, and runs follow each other closely),
      // then sleep
      bool got = false;
      for (int spin = 0; spin < 20000; spin++) {
        if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
        cpu_relax();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        seen = gen_.load(std::memory_order_acquire);
        if (!open_) continue;                 // that run is over already
        in_work_.fetch_add(1, std::memory_order_acq_rel);
      }
      work();
      in_work_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }

  int n_threads_ = 1;
  bool started_ = false, open_ = false;
  std::mutex busy_, m_;
  std::condition_variable cv_;
  std::atomic<unsigned> gen_{0};
  std::atomic<int> next_{0}, done_{0}, in_work_{0};
  std::atomic<bool> failed_{false};
  std::exception_ptr error_;
  const std::function<void(int)> *fn_ = nullptr;
  int n_ = 0;
};

}  // namespace rs
}  // namespace mods
