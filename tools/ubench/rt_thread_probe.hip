// Which runtime calls wake the HIP runtime's own thread?  One pattern at a time for ~2 s at a fixed rate of operations from the main
// thread (sleeping polls in between, as the pipeline's workers wait), then the CPU time every OTHER thread of the process used.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/rt_thread_probe.hip -o /tmp/rt_probe && /tmp/rt_probe
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

__global__ void tiny(int *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
__global__ void spin_us(int *p, long long us) {   // ~us microseconds (s_memrealtime counts at 100 MHz)
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < us * 100) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1;
}
__global__ void to_host(volatile int *h, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) h[0] = v; }

static std::map<int, double> task_cpu() {
  std::map<int, double> out;
  if (DIR *d = opendir("/proc/self/task")) {
    while (dirent *e = readdir(d)) {
      const int tid = atoi(e->d_name);
      if (tid <= 0) continue;
      char path[64]; snprintf(path, sizeof(path), "/proc/self/task/%d/schedstat", tid);
      if (FILE *f = fopen(path, "r")) { long long ns = 0; if (fscanf(f, "%lld", &ns) == 1) out[tid] = ns * 1e-9; fclose(f); }
    }
    closedir(d);
  }
  return out;
}
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static void nap(long us) { timespec ts = {0, us * 1000}; nanosleep(&ts, nullptr); }
static hipEvent_t g_ev;
static void wait_event(hipStream_t s) { hipEventRecord(g_ev, s); while (hipEventQuery(g_ev) == hipErrorNotReady) nap(50); (void)hipGetLastError(); }
static void wait_poll(hipStream_t s) { while (hipStreamQuery(s) == hipErrorNotReady) nap(50); (void)hipGetLastError(); }

int main(int argc, char **argv) {
  const bool only_fork = argc > 1 && !strcmp(argv[1], "fork");   // only the two-stream patterns (to try environment settings on them)
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int *d; hipMalloc(&d, 1 << 20);
  char *pin; hipHostMalloc(&pin, 1 << 20);
  volatile int *flag; hipHostMalloc((void **)&flag, 64); *flag = 0;
  char *page = (char *)malloc(1 << 20);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventCreateWithFlags(&g_ev, hipEventDisableTiming);
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t fork, join; hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
  char *dbig; hipMalloc(&dbig, 4 << 20); char *pinbig; hipHostMalloc(&pinbig, 4 << 20);
  hipGraph_t graph; hipGraphExec_t gexec = nullptr;
  { hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll);
    hipEventRecord(fork, s); hipStreamWaitEvent(s2, fork, 0);
    hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s2, d + 64, 1000ll);
    hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll);
    hipEventRecord(join, s2); hipStreamWaitEvent(s, join, 0);
    hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll);
    hipStreamEndCapture(s, &graph); hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0); }
  const int self = (int)syscall(SYS_gettid);
  struct Pat { const char *name; int kind; };
  const Pat pats[] = {{"nothing (naps only)", 0}, {"kernel launches, one poll-wait per 20", 1}, {"kernel launch + poll-wait each", 2},
                      {"kernel + flag in pinned memory, no runtime call to wait", 3}, {"H2D 64 KB pinned async + poll-wait", 4},
                      {"D2H 64 KB pinned async + poll-wait", 5}, {"H2D 64 KB pageable async + poll-wait", 6}, {"D2H 4 B pinned async + poll-wait", 7},
                      {"kernel + hipEventRecord + hipEventQuery polls", 8}, {"kernel + hipStreamSynchronize", 9},
                      {"20 kernels then ONE poll-wait (rate x20)", 10}, {"5 ms kernel, hipStreamQuery every 50 us", 11},
                      {"5 ms kernel, hipEventQuery every 50 us", 12}, {"5 ms kernel + 4 B D2H behind it, hipStreamQuery every 50 us", 13},
                      {"5 ms kernel, flag in pinned memory read every 50 us", 14},
                      {"E: 5 ms kernel + 4 B D2H pinned, event wait", 15}, {"E: 5 ms kernel + 64 KB D2H pinned, event wait", 16},
                      {"E: 2 MB H2D pinned + 5 ms kernel, event wait", 17}, {"E: 5 ms kernel + hipMemsetAsync 4 KB, event wait", 18},
                      {"E: fork / join over two streams (4 x 1 ms kernels), event wait", 19}, {"E: the same as a hipGraphLaunch, event wait", 20},
                      {"E: 100 x 50 us kernels, event wait", 21}, {"E: 5 ms kernel + 4 B D2H to PAGEABLE memory, event wait", 22},
                      {"E: 5 ms kernel + 64 KB H2D pageable, event wait", 23}, {"E: 5 ms kernel + kernel storing to pinned memory, event wait", 24}};
  for (const Pat &p : pats) {
    if (only_fork && p.kind != 19 && p.kind != 20 && p.kind != 11) continue;
    const auto c0 = task_cpu();
    const double t0 = now();
    long ops = 0; int seq = 0;
    while (now() - t0 < 2.0) {
      switch (p.kind) {
        case 0: nap(100); break;
        case 1: for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, i); nap(100); } wait_poll(s); ops += 20; break;
        case 2: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, 1); wait_poll(s); nap(100); ops++; break;
        case 3: seq++; hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, flag, seq); while (*flag != seq) nap(50); nap(100); ops++; break;
        case 4: hipMemcpyAsync(d, pin, 65536, hipMemcpyHostToDevice, s); wait_poll(s); nap(100); ops++; break;
        case 5: hipMemcpyAsync(pin, d, 65536, hipMemcpyDeviceToHost, s); wait_poll(s); nap(100); ops++; break;
        case 6: hipMemcpyAsync(d, page, 65536, hipMemcpyHostToDevice, s); wait_poll(s); nap(100); ops++; break;
        case 7: hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s); wait_poll(s); nap(100); ops++; break;
        case 8: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, 1); hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) nap(50); (void)hipGetLastError(); nap(100); ops++; break;
        case 9: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, 1); hipStreamSynchronize(s); nap(100); ops++; break;
        case 11: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); wait_poll(s); ops++; break;
        case 12: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) nap(50); (void)hipGetLastError(); ops++; break;
        case 13: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s); wait_poll(s); ops++; break;
        case 14: seq++; hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, flag, seq); while (*flag != seq) nap(50); ops++; break;
        case 15: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s); wait_event(s); ops++; break;
        case 16: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemcpyAsync(pin, d, 65536, hipMemcpyDeviceToHost, s); wait_event(s); ops++; break;
        case 17: hipMemcpyAsync(dbig, pinbig, 2 << 20, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); wait_event(s); ops++; break;
        case 18: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemsetAsync(d + 1024, 0, 4096, s); wait_event(s); ops++; break;
        case 19: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll);
                 hipEventRecord(fork, s); hipStreamWaitEvent(s2, fork, 0);
                 hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s2, d + 64, 1000ll);
                 hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll);
                 hipEventRecord(join, s2); hipStreamWaitEvent(s, join, 0);
                 hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 1000ll); wait_event(s); ops++; break;
        case 20: hipGraphLaunch(gexec, s); wait_event(s); ops++; break;
        case 21: for (int i = 0; i < 100; i++) hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 50ll); wait_event(s); ops++; break;
        case 22: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemcpyAsync(page, d, 4, hipMemcpyDeviceToHost, s); wait_event(s); ops++; break;
        case 23: hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipMemcpyAsync(d + 2048, page, 65536, hipMemcpyHostToDevice, s); wait_event(s); ops++; break;
        case 24: seq++; hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, s, d, 5000ll); hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, flag, seq); wait_event(s); ops++; break;
        case 10: for (int i = 0; i < 20; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, i); wait_poll(s); nap(100); ops += 20; break;
      }
    }
    const double dt = now() - t0;
    const auto c1 = task_cpu();
    double others = 0, busiest = 0, mine = 0;
    for (auto &kv : c1) {
      const double used = kv.second - (c0.count(kv.first) ? c0.at(kv.first) : 0.0);
      if (kv.first == self) mine = used; else { others += used; if (used > busiest) busiest = used; }
    }
    printf("%-58s %7.0f ops/s | main thread %5.1f %% | other threads %5.1f %% of a core (busiest %5.1f %%) = %6.1f us per op\n", p.name, ops / dt, 100 * mine / dt,
           100 * others / dt, 100 * busiest / dt, ops ? 1e6 * others / ops : 0.0);
    fflush(stdout);
  }
  return 0;
}
