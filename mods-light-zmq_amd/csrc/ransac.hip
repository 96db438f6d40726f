// LO-RANSAC homography verification: GPU hypothesis scoring + host control loop.
//
// Reference behaviour: exp_ransacHcustom, degensac/exp_ranH.c:796-1236 (LO: exp_inHranicustom
// :741-793, exp_iterHcustom :617-737), called from LORANSACFiltering, matching/matching.cpp:637-823.
//
// Structure.  The reference draws one 4-point sample per iteration from a libc generator that
// is re-seeded every iteration (srand(seed); 4 x random(); seed = rand(), exp_ranH.c:861-863), so
// the whole sample sequence is a function of the first seed only and never depends on scores.
// It is therefore generated ahead of time on the host (bit-exact glibc generator), minimal
// solutions are solved on the host (9x9 Gauss-Jordan, ~1 us each) and a batch of hypotheses is
// scored over all correspondences on the GPU in one launch pair:
//   score kernel : thread (correspondence i, hypothesis k): error d, symmetric-check error,
//                  truncated-quadratic gain; inlier counts by wave ballots + one atomic per wave
//   gain kernel  : lane k adds the gains of hypothesis k in correspondence order (the MSAC
//                  score J is a sequential fp64 sum in the reference; same order here)
// The host then replays the reference's decision sequence over the batch in order (best-so-far,
// symmetric check, LO trigger, adaptive stopping); hypotheses scored beyond the stopping point
// are discarded.  Local optimisation (least squares on inliers, iterated re-estimation) stays on
// the host: it is a chain of data-dependent small solves.
#include "common.hpp"
#include "ransac_host.hpp"
#include "ransac_gpu.hpp"
#include "ransac_soa.hpp"
#include "ransac_dev.hpp"
#include <ctime>
#include <cstdlib>
#include <mutex>
#include <unistd.h>
#include <sys/syscall.h>

namespace mods {

using rs::Score;

enum { ERR_SAMPSON = 0, ERR_SYMSUM = 1, ERR_SYMMAX = 2 };

struct HypDev {            // one scored hypothesis
  double h[9];
  double Hinv[9], H1[9];   // symmetric-error operands (Hinv = h^T as stored, H1 = minv(Hinv))
};
static_assert(sizeof(HypDev) <= HYP_SLOT_BYTES, "hypothesis slot too small");

// grid = (ceil(len/256), n_hyp), block 256.  d[k][i], gain[i][kstride], counts[k] = {I, Isym}.
__global__ __launch_bounds__(256) void ransac_score_kernel(const double *__restrict__ u, int len, const HypDev *__restrict__ hyp,
                                                           int err_type, int do_sym, double th, double th_check,
                                                           double *__restrict__ d_out, double *__restrict__ gain, int kstride,
                                                           int *__restrict__ counts) {
  __shared__ HypDev sh;
  const int k = blockIdx.y;
  for (int q = threadIdx.x; q < (int)(sizeof(HypDev) / sizeof(double)); q += 256) ((double *)&sh)[q] = ((const double *)&hyp[k])[q];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool inl = false, inls = false;
  if (i < len) {
    double uu[6];
#pragma unroll
    for (int q = 0; q < 6; q++) uu[q] = u[(size_t)i * 6 + q];
    double d, d1 = 0, d2 = 0;
    if (err_type != ERR_SAMPSON || do_sym) hsym_dev(uu, sh.Hinv, sh.H1, &d1, &d2);
    if (err_type == ERR_SAMPSON) d = hds_dev(uu, sh.h);
    else if (err_type == ERR_SYMSUM) d = d1 + d2;
    else d = d1 > d2 ? d1 : d2;     // MAX(d1,d2), Htools.c:12,281
    d_out[(size_t)k * len + i] = d;
    gain[(size_t)i * kstride + k] = trunc_quad_dev(d, th);
    inl = d <= th;
    inls = do_sym && (d1 + d2) <= th_check;
  }
  const unsigned long long m1 = __ballot(inl), m2 = __ballot(inls);
  if ((threadIdx.x & 63) == 0) {
    if (m1) atomicAdd(&counts[2 * k], __popcll(m1));
    if (m2) atomicAdd(&counts[2 * k + 1], __popcll(m2));
  }
}

// grid = ceil(n_hyp / COLS), block 256: lane k of wave 0 sums gain[i][k] for i = 0..len-1 in order.  The sum is a serial
// chain, so one wave adds; what it would wait for is the memory latency of 8 bytes per row, so all four waves fetch the
// next ROWS rows into registers while wave 0 adds the current ones out of LDS.  The gain matrix has hyp_cap (a multiple of 64)
// columns, so every lane's column exists.  COLS = 64, ROWS = 128 for a full batch; the first batch of a call has 8 hypotheses:
// 16 columns wide the same LDS holds 384 rows, a third of the chunks and of the exposed fetch latencies (114 -> ~50 us for 6 000
// correspondences, on the path of a single pair's latency).
template <int COLS, int ROWS>
__device__ __forceinline__ void gain_sums(const double *__restrict__ gain, int len, int n_hyp, int kstride, int *__restrict__ counts,
                                          double *__restrict__ J_out, int *__restrict__ counts_out, double *s_rows) {
  constexpr int RG = 256 / COLS;            // row groups of a fetch step
  constexpr int PER = ROWS / RG;
  const int col = threadIdx.x % COLS, rsub = threadIdx.x / COLS;
  const int kf = blockIdx.x * COLS + col;   // column this thread fetches
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * COLS + lane;   // column lane `lane` of wave 0 sums (lane < COLS)
  double v[PER];
  double sum = 0;
#define GAIN_FETCH(c0)                                                                  \
  _Pragma("unroll") for (int j = 0; j < PER; j++) {                                     \
    const int row = (c0) + rsub + RG * j;                                               \
    v[j] = row < len ? gain[(size_t)row * kstride + kf] : 0.0;                          \
  }
  GAIN_FETCH(0)
  for (int c0 = 0; c0 < len; c0 += ROWS) {
#pragma unroll
    for (int j = 0; j < PER; j++) s_rows[(rsub + RG * j) * COLS + col] = v[j];
    __syncthreads();
    if (c0 + ROWS < len) { GAIN_FETCH(c0 + ROWS) }
    if (wv == 0 && lane < COLS) {
      const int rows = min(ROWS, len - c0);
      int r = 0;
      for (; r + 15 < rows; r += 16) {
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = s_rows[(r + q) * COLS + lane];
#pragma unroll
        for (int q = 0; q < 16; q++) sum += t[q];
      }
      for (; r < rows; r++) sum += s_rows[r * COLS + lane];
    }
    __syncthreads();
  }
#undef GAIN_FETCH
  // J and the inlier counts of hypothesis k go straight to the host's (pinned, device-visible) result block, and the device counters
  // are left at zero for the next batch: no copy and no fill launch per scoring round
  if (wv == 0 && lane < COLS && k < n_hyp) {
    J_out[k] = sum;
    counts_out[2 * k] = counts[2 * k]; counts_out[2 * k + 1] = counts[2 * k + 1];
    counts[2 * k] = 0; counts[2 * k + 1] = 0;
  }
}
constexpr int GAIN_ROWS = 128;
__global__ void __launch_bounds__(256) ransac_gain_kernel(const double *__restrict__ gain, int len, int n_hyp, int kstride,
                                                          int *__restrict__ counts, double *__restrict__ J_out, int *__restrict__ counts_out) {
  __shared__ double s_rows[GAIN_ROWS * 64];
  gain_sums<64, GAIN_ROWS>(gain, len, n_hyp, kstride, counts, J_out, counts_out, s_rows);
}
constexpr int GAIN_FEW = 16, GAIN_FEW_ROWS = 384;      // up to 16 hypotheses: 16 columns x 384 rows in the same 48 KB
__global__ void __launch_bounds__(256) ransac_gain_few_kernel(const double *__restrict__ gain, int len, int n_hyp, int kstride,
                                                              int *__restrict__ counts, double *__restrict__ J_out, int *__restrict__ counts_out) {
  __shared__ double s_rows[GAIN_FEW_ROWS * GAIN_FEW];
  gain_sums<GAIN_FEW, GAIN_FEW_ROWS>(gain, len, n_hyp, kstride, counts, J_out, counts_out, s_rows);
}

RansacGpu::~RansacGpu() {
  if (device < 0) return;
  // The workspace is thread-local.  A worker thread returns its HBM when it ends; the main thread's copy is
  // destroyed during process exit, when the HIP runtime (or a profiler layered on it) may already be shutting
  // down and a hipFree can block forever - the process is going away, so nothing is released there.
  if ((long)getpid() == (long)syscall(SYS_gettid)) return;
  (void)hipSetDevice(device);
  (void)hipFree(u_dev); (void)hipFree(hyp_dev); (void)hipHostFree(hyp_host); (void)hipFree(d_dev); (void)hipFree(gain_dev);
  (void)hipFree(J_dev); (void)hipHostFree(J_host);   // (the counts live behind the J values in the same allocations)
  (void)hipHostFree(row_host); (void)hipFree(aux_dev);
  (void)hipHostFree(cand_host); (void)hipHostFree(candc_host); (void)hipHostFree(cntf_host); (void)hipHostFree(cntc_host);   // (cand_dev / candc_dev are their device addresses)
  for (int q = 0; q < 2; q++) if (cand_ev[q]) (void)hipEventDestroy(cand_ev[q]);
  if (stream) (void)hipStreamDestroy(stream);
}

static int g_ransac_device = 0;
static long g_pinned_seed = -1;
static std::mutex g_cfg_mutex;

static thread_local int g_ransac_failed = 0;
void ransac_set_failed(int failed) { g_ransac_failed = failed; }
int ransac_failed() { return g_ransac_failed; }

long ransac_pinned_seed() {
  long pinned;
  { std::lock_guard<std::mutex> lk(g_cfg_mutex); pinned = g_pinned_seed; }
  if (pinned < 0) { const char *e = getenv("MODS_RANSAC_SEED"); if (e) pinned = atol(e); }
  return pinned;
}

RansacGpu *ransac_gpu() {
  static thread_local RansacGpu ws;
  if (ws.device < 0) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device: RANSAC scoring has no CPU path"); return nullptr; }
    int dev;
    { std::lock_guard<std::mutex> lk(g_cfg_mutex); dev = g_ransac_device; }
    if (hipSetDevice(dev) != hipSuccess) { set_error("hipSetDevice(%d) failed", dev); return nullptr; }
    // the scoring launches are tiny and a host thread waits for each of them: highest priority, so that they do not
    // queue behind a describe batch of the pipeline's GPU workers (16-25 ms per call when they did)
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (hipStreamCreateWithPriority(&ws.stream, hipStreamNonBlocking, prio_high) != hipSuccess) { set_error("stream creation failed"); return nullptr; }
    ws.device = dev;
  }
  (void)hipSetDevice(ws.device);
  return &ws;
}

bool ransac_ws_reserve(RansacGpu *ws, int len, int n_hyp) {
  if ((size_t)len * 6 > ws->u_cap) {
    if (ws->u_dev) RS_CHECK(hipFree(ws->u_dev));
    ws->u_cap = (size_t)len * 6 * 2;
    if (ws->u_cap < 6 * 16384) ws->u_cap = 6 * 16384;
    RS_CHECK(hipMalloc(&ws->u_dev, ws->u_cap * sizeof(double)));
  }
  if (n_hyp > ws->hyp_cap) {
    if (ws->hyp_dev) { RS_CHECK(hipFree(ws->hyp_dev)); RS_CHECK(hipHostFree(ws->hyp_host));
                       RS_CHECK(hipFree(ws->J_dev)); RS_CHECK(hipHostFree(ws->J_host)); }
    ws->hyp_cap = 64;
    while (ws->hyp_cap < n_hyp) ws->hyp_cap *= 2;
    n_hyp = ws->hyp_cap;
    RS_CHECK(hipMalloc(&ws->hyp_dev, (size_t)HYP_SLOT_BYTES * n_hyp));
    RS_CHECK(hipHostMalloc(&ws->hyp_host, (size_t)HYP_SLOT_BYTES * n_hyp));
    // J[hyp_cap] | counts[2 * hyp_cap] in ONE allocation on either side: the scores of a batch come back in one copy
    RS_CHECK(hipMalloc(&ws->J_dev, 2 * sizeof(double) * n_hyp));
    RS_CHECK(hipHostMalloc(&ws->J_host, 2 * sizeof(double) * n_hyp));
    ws->counts_dev = (int *)(ws->J_dev + n_hyp);
    ws->counts_host = (int *)(ws->J_host + n_hyp);
    // on the workspace's own stream: a memset on the legacy stream is refused while ANY blocking stream of the process is being
    // captured (another thread recording a context's launch chain), and it would wait for every blocking stream besides
    RS_CHECK(hipMemsetAsync(ws->J_dev, 0, 2 * sizeof(double) * n_hyp, ws->stream));
    RS_CHECK(mods::stream_wait(ws->stream));
    ws->counts_dirty = false;
    ws->dg_cap = 0;
  }
  // hipFree / hipMalloc synchronise the whole device (20+ ms under a running pipeline): grow in powers of two from a
  // floor that covers an ordinary pair, so that a worker stops reallocating after its first call
  size_t len_cap = 16384;
  while (len_cap < (size_t)len) len_cap *= 2;
  const size_t need = len_cap * ws->hyp_cap;
  if (need > ws->dg_cap) {
    if (ws->d_dev) { RS_CHECK(hipFree(ws->d_dev)); RS_CHECK(hipFree(ws->gain_dev)); }
    ws->dg_cap = need;
    RS_CHECK(hipMalloc(&ws->d_dev, need * sizeof(double)));
    RS_CHECK(hipMalloc(&ws->gain_dev, need * sizeof(double)));
  }
  if ((size_t)len > ws->row_cap) {
    if (ws->row_host) RS_CHECK(hipHostFree(ws->row_host));
    ws->row_cap = (size_t)len * 2;
    if (ws->row_cap < 16384) ws->row_cap = 16384;
    RS_CHECK(hipHostMalloc(&ws->row_host, ws->row_cap * sizeof(double)));
  }
  return true;
}

// The device counters are zero between scoring rounds: ransac_ws_reserve clears them when it allocates, ransac_gain_kernel
// after it has read them.  A round that fails between its score and gain launches (or is abandoned) leaves counts behind; the
// flag makes the next round of this workspace clear them first.
bool ransac_counts_begin(RansacGpu *ws) {
  if (ws->counts_dirty) RS_CHECK(hipMemsetAsync(ws->counts_dev, 0, 2 * sizeof(int) * (size_t)ws->hyp_cap, ws->stream));
  ws->counts_dirty = true;      // until the gain kernel of this round has run and been waited for
  return true;
}

// scores hyp_host[0..n) over all correspondences; fills counts_host / J_host
static bool gpu_score(RansacGpu *ws, int len, int n, int err_type, int do_sym, double th, double th_check) {
  if (!ransac_counts_begin(ws)) return false;
  RS_CHECK(hipMemcpyAsync(ws->hyp_dev, ws->hyp_host, sizeof(HypDev) * n, hipMemcpyHostToDevice, ws->stream));
  hipLaunchKernelGGL(ransac_score_kernel, dim3((len + 255) / 256, n), dim3(256), 0, ws->stream, ws->u_dev, len, (const HypDev *)ws->hyp_dev, err_type,
                     do_sym, th, th_check, ws->d_dev, ws->gain_dev, ws->hyp_cap, ws->counts_dev);
  if (n <= GAIN_FEW) hipLaunchKernelGGL(ransac_gain_few_kernel, dim3(1), dim3(256), 0, ws->stream, ws->gain_dev, len, n, ws->hyp_cap, ws->counts_dev, ws->J_host,
                     ws->counts_host);
  else hipLaunchKernelGGL(ransac_gain_kernel, dim3((n + 63) / 64), dim3(256), 0, ws->stream, ws->gain_dev, len, n, ws->hyp_cap, ws->counts_dev, ws->J_host,
                     ws->counts_host);
  RS_CHECK(hipGetLastError());
  RS_CHECK(mods::stream_wait(ws->stream));
  ws->counts_dirty = false;
  ws->launches += 2;
  return true;
}

bool ransac_fetch_row(RansacGpu *ws, int len, int k, double *dst) {
  RS_CHECK(hipMemcpyAsync(ws->row_host, ws->d_dev + (size_t)k * len, sizeof(double) * len, hipMemcpyDeviceToHost, ws->stream));
  RS_CHECK(mods::stream_wait(ws->stream));
  memcpy(dst, ws->row_host, sizeof(double) * len);
  return true;
}

}  // namespace mods

using namespace mods;

extern "C" {

typedef void (*HDsPtr)(const double *, const double *, const double *, double *, int);
typedef void (*HDsiPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*HDsidxPtr)(const double *, const double *, const double *, double *, int, int *, int);

// Error functions with the reference's signatures (Htools.h).  `lin` is ignored: the design-matrix
// rows are rebuilt from u (same values).  Host-side; used by LO and by the LAF checks.
void HDs(const double *lin, const double *u, const double *H, double *p, int len) {
  (void)lin;
  for (int i = 0; i < len; i++) p[i] = rs::hds_point(u + 6 * i, H);
}
void HDsSym(const double *lin, const double *u, const double *H, double *p, int len) {
  (void)lin;
  rs::SymH s; rs::sym_prepare(H, &s);
  for (int i = 0; i < len; i++) { double d1, d2; rs::hsym_point(u + 6 * i, &s, &d1, &d2); p[i] = d1 + d2; }
}
void HDsSymMax(const double *lin, const double *u, const double *H, double *p, int len) {
  (void)lin;
  rs::SymH s; rs::sym_prepare(H, &s);
  for (int i = 0; i < len; i++) { double d1, d2; rs::hsym_point(u + 6 * i, &s, &d1, &d2); p[i] = d1 > d2 ? d1 : d2; }
}
void HDsi(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) {
  (void)lin; (void)len;
  for (int i = 0; i < ni; i++) p[i] = rs::hds_point(u6 + 6 * pts[i], H);
}
void HDsiSym(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) {
  (void)lin; (void)len;
  rs::SymH s; rs::sym_prepare(H, &s);
  for (int i = 0; i < ni; i++) { double d1, d2; rs::hsym_point(u6 + 6 * pts[i], &s, &d1, &d2); p[i] = d1 + d2; }
}
void HDsiSymMax(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni) {
  (void)lin; (void)len;
  rs::SymH s; rs::sym_prepare(H, &s);
  for (int i = 0; i < ni; i++) { double d1, d2; rs::hsym_point(u6 + 6 * pts[i], &s, &d1, &d2); p[i] = d1 > d2 ? d1 : d2; }
}
// HDsidx / HDsSymidx / HDsSymidxMax are passed through LORANSACFiltering but never invoked on the
// live path (exp_ranH.c uses them only under __LSBL_MCE__); exported for link compatibility.
void HDsidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz) { HDsi(lin, mu, H, p, len, idx, siz); }
void HDsSymidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz) { HDsiSym(lin, mu, H, p, len, idx, siz); }
void HDsSymidxMax(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz) { HDsiSymMax(lin, mu, H, p, len, idx, siz); }

// first n outputs of srand(seed); rand() as the sampler sees them (host-only self-test hook: lets the
// CPU test-suite compare the restated generator with the libc of the machine it runs on)
void mods_test_glibc_rand(unsigned seed, int n, int *out) {
  rs::GlibcRand g;
  g.seed(seed);
  for (int i = 0; i < n; i++) out[i] = g.next();
}

// creates the calling thread's scoring stream and workspace for correspondence lists up to `len` (what the first
// exp_ransac*custom call of a thread would otherwise do, with its hipMallocs, in the middle of a running pipeline)
int mods_ransac_warmup(int device, int len) {
  mods_ransac_set_device(device);
  RansacGpu *ws = ransac_gpu();
  if (!ws) return MODS_E_NODEVICE;
  if (!ransac_ws_reserve(ws, len > 0 ? len : 1, 64)) return MODS_E_HIP;
  hipLaunchKernelGGL(ransac_gain_kernel, dim3(1), dim3(256), 0, ws->stream, ws->gain_dev, 0, 0, ws->hyp_cap, ws->counts_dev, ws->J_host, ws->counts_host);   // loads the code object
  if (mods::stream_wait(ws->stream) != hipSuccess) { set_error("ransac warm-up failed"); return MODS_E_HIP; }
  return MODS_OK;
}

int mods_ransac_set_device(int device) { std::lock_guard<std::mutex> lk(g_cfg_mutex); g_ransac_device = device; return MODS_OK; }
// seed >= 0: every call behaves as if time(NULL) returned `seed`; < 0: back to the wall clock
void mods_ransac_pin_seed(long seed) { std::lock_guard<std::mutex> lk(g_cfg_mutex); g_pinned_seed = seed; }

}  // extern "C"

namespace mods {

// MODS_RANSAC_PROF=1: per-call breakdown of the verification on stderr (development aid)
struct RsProf { double us[8]; };   // 0 hypotheses, 1 gpu_score, 2 fetch_row, 3 LO u2h, 4 LO error function, 5 LO, 6 upload, 7 whole call
static thread_local RsProf g_rsprof;
// MODS_RANSAC_PROF: 0 = off, 1 = phase times on stderr in wall time, 2 ("cpu") = in the calling thread's CPU time (a sleeping wait costs none)
int ransac_profile_mode() { static const int mode = [] { const char *e = getenv("MODS_RANSAC_PROF"); return !e ? 0 : (!strcmp(e, "cpu") ? 2 : 1); }(); return mode; }
bool ransac_profile_on() { return ransac_profile_mode() != 0; }
static int rsprof_on() { return ransac_profile_on() ? 1 : 0; }
static inline double rs_now_us() {
  timespec ts; clock_gettime(ransac_profile_mode() == 2 ? CLOCK_THREAD_CPUTIME_ID : CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
struct RsTimer {
  int slot; double t0;
  explicit RsTimer(int s) : slot(s), t0(rsprof_on() ? rs_now_us() : 0) {}
  ~RsTimer() { if (rsprof_on()) g_rsprof.us[slot] += rs_now_us() - t0; }
};

// The LO step of one call fits and scores the same inlier lists over and over: its ten inner samples (exp_ranH.c:741-793) mostly lead
// to the same wide-threshold sets, and the reference's own hash table only ends an iteration once the NARROW set repeats - 9 to 12 of
// the 14 full least-squares fits of a 1080p pair (and the error passes behind them) repeat an earlier one of the same call.  A fit is
// a function of (u, list), an error vector of (u, H): both are kept per call and looked up by exact comparison (list / the nine
// doubles bit by bit), so a hit returns the bits the computation would return.  Fits with a random subset (inlLimit) draw from the
// generator and are never looked up; a caller's own error function is never cached.
struct LoMemo {
  struct Fit { int n; std::vector<int> list; double h[9]; };
  struct Err { double h[9]; std::vector<double> d; };
  static constexpr size_t kFits = 12, kErrs = 8;
  std::vector<Fit> fits;
  std::vector<Err> errs;
  size_t fit_next = 0, err_next = 0;
  const double *find_fit(const int *list, int n) const {
    for (const Fit &f : fits)
      if (f.n == n && !memcmp(f.list.data(), list, sizeof(int) * (size_t)n)) return f.h;
    return nullptr;
  }
  void add_fit(const int *list, int n, const double *h) {
    if (fits.size() < kFits) fits.emplace_back();
    Fit &f = fits[fit_next % fits.size()];
    fit_next = (fit_next + 1) % kFits;
    f.n = n; f.list.assign(list, list + n); memcpy(f.h, h, sizeof(f.h));
  }
  const double *find_err(const double *h) const {
    for (const Err &e : errs)
      if (!memcmp(e.h, h, sizeof(e.h))) return e.d.data();
    return nullptr;
  }
  void add_err(const double *h, const double *d, int n) {
    if (errs.size() < kErrs) errs.emplace_back();
    Err &e = errs[err_next % errs.size()];
    err_next = (err_next + 1) % kErrs;
    memcpy(e.h, h, sizeof(e.h)); e.d.assign(d, d + n);
  }
};

struct ErrFn {   // host-side error function of the run (LO path); d has room for n_pad values
  int type; HDsPtr custom; const PointsSoA *pts; LoMemo *memo;
  void operator()(const double *u, const double *H, double *d, int len) const {
    RsTimer t_(4);
    if (custom) { custom(nullptr, u, H, d, len); return; }
    if (memo)
      if (const double *known = memo->find_err(H)) { memcpy(d, known, sizeof(double) * (size_t)pts->n_pad); return; }
    if (type == ERR_SAMPSON) pts->ops->hds_all(pts->col, pts->n_pad, H, d);
    else {
      rs::SymH s; rs::sym_prepare(H, &s);
      pts->ops->hsym_all(pts->col, pts->n_pad, s.H1, s.Hinv, type == ERR_SYMSUM ? 0 : 1, d);
    }
    if (memo) memo->add_err(H, d, pts->n_pad);
  }
};
// u2h of a whole list (no random subset) through the call's memo
static void u2h_memo(LoMemo *memo, const double *u, const int *list, int n, double *h) {
  if (memo && n >= 32)
    if (const double *known = memo->find_fit(list, n)) { memcpy(h, known, 9 * sizeof(double)); return; }
  rs::u2h(u, list, n, h, nullptr);
  if (memo && n >= 32) memo->add_fit(list, n, h);
}

struct LoState {
  const double *u; int len; double th;
  double *errs[5];
  double *buffer;
  rs::GlibcRand *rng;
  rs::HashTable *ht;
  unsigned inlLimit;
  ErrFn errfn;
  PointsSoA *pts;
  LoMemo *memo;
};

// inlidxs, rtools.c:155-166: the gains are computed lanes-wide, the MSAC sum and the index list stay sequential
static rs::Score inlidxs_v(const LoState &L, const double *err, double th, int *inl) {
  if (th == 0) return rs::inlidxs(err, L.len, th, inl);
  double *g = L.pts->gains.data();
  L.pts->ops->gains_all(err, L.pts->n_pad, th * 9 / 4, g);
  rs::Score s = {0, 0};
  const int len = L.len;
  unsigned n = 0;
  for (int i = 0; i < len; ++i) {
    s.J += g[i];
    inl[n] = i;                       // (written unconditionally, kept when the test holds: no branch on the data)
    n += err[i] <= th ? 1u : 0u;
  }
  s.I = n;
  return s;
}
// the same where the caller reads the count and the list only (the wide-threshold sets the least-squares steps are fed with,
// exp_ranH.c:650, 673, 1046, 1050): the sum - a serial chain of len additions behind len divisions - is not formed, J stays 0
static rs::Score inlidxs_list(const LoState &L, const double *err, double th, int *inl) {
  rs::Score s = {0, 0};
  const int len = L.len;
  unsigned n = 0;
  for (int i = 0; i < len; ++i) {
    inl[n] = i;
    n += err[i] <= th ? 1u : 0u;
  }
  s.I = n;
  return s;
}

// exp_iterHcustom, exp_ranH.c:617-737 (__D3__ with inlLimit = 1e6 => least squares on all
// inliers; __HASHING__ on)
static Score lo_iter(LoState &L, int *inliers, double th, double ths, int steps, double *H, int iterID, double *resids) {
  const int len = L.len;
  double *d = L.errs[1];
  double h[9];
  Score maxS = {0, 0}, S = {0, 0}, Ss;
  const double dth = (ths - th) / (steps);
  auto lsq = [&](const Score &Sc) {
    unsigned detached = (unsigned)(int)(Sc.I * 1);
    if (detached > L.inlLimit) detached = L.inlLimit;
    if (detached < 4) detached = 4;
    RsTimer t_(3);
    if (detached >= Sc.I) u2h_memo(L.memo, L.u, inliers, (int)Sc.I, h);
    else {
      int *sub = rs::randsubset(*L.rng, inliers, (int)Sc.I, (int)detached);
      rs::u2h(L.u, sub, (int)detached, h, L.buffer);
    }
  };
  maxS = inlidxs_v(L, L.errs[4], th, inliers);
  if (maxS.I < 4) return S;
  S = inlidxs_list(L, L.errs[4], th * 2, inliers);   // th*MWM, MWM = (9/4) = 2 (rtools.h:33); only S.I and the list are read
  lsq(S);
  for (int it = 0; it < steps; it++) {
    L.errfn(L.u, h, d, len);
    if (resids) memcpy(resids + (size_t)it * len, d, len * sizeof(double));
    Ss = inlidxs_v(L, d, th, inliers);
    const uint32_t hash = rs::super_fast_hash((const char *)inliers, (int)(Ss.I * sizeof(*inliers)));
    const int ret = L.ht->contains(hash, (int)Ss.I, iterID);
    if (ret != -1 && ret != iterID) { S.I = 0; S.J = 0; return S; }
    if (ret == -1) L.ht->insert(hash, (int)Ss.I, iterID);
    S = inlidxs_list(L, d, ths * 2, inliers);
    if (rs::score_less(maxS, Ss)) {
      maxS = Ss;
      L.errs[1] = L.errs[0];
      L.errs[0] = d;
      d = L.errs[1];
      memcpy(H, h, 9 * sizeof(double));
    }
    if (S.I < 4) return maxS;
    lsq(S);
    ths -= dth;
  }
  L.errfn(L.u, h, d, len);
  if (resids) memcpy(resids + (size_t)4 * len, d, len * sizeof(double));
  S = inlidxs_v(L, d, th, inliers);
  if (rs::score_less(maxS, S)) {
    maxS = S;
    L.errs[1] = L.errs[0];
    L.errs[0] = d;
    memcpy(H, h, 9 * sizeof(double));
  }
  return maxS;
}

// exp_inHranicustom, exp_ranH.c:741-793 (RAN_REP = 10, ILSQ_ITERS = 4, TC = 4)
static Score lo_inner(LoState &L, int *inliers, int ninl, double th, double *H, int rep, int *iterID, double *resids) {
  const int len = L.len;
  Score S, maxS = {0, 0};
  double *d, h[9];
  std::vector<int> intbuff(len);
  if (ninl < 8) {
    if (resids) memset(resids, 0xFF, (size_t)(62 - 2) * len * sizeof(double));   // RESIDS_M - 2
    return maxS;
  }
  int ssiz = ninl / 2;
  if (ssiz > 12) ssiz = 12;
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  for (int i = 0; i < rep; i++) {
    int *sample = rs::randsubset(*L.rng, inliers, ninl, ssiz);
    rs::u2h(L.u, sample, ssiz, h, L.buffer);
    L.errfn(L.u, h, L.errs[0], len);
    if (resids) memcpy(resids + (size_t)i * 6 * len, L.errs[0], len * sizeof(double));
    L.errs[4] = L.errs[0];
    S = lo_iter(L, intbuff.data(), th, 4 * th, 4, h, ++*iterID, resids ? resids + (size_t)i * 6 * len + len : nullptr);
    if (rs::score_less(maxS, S)) {
      maxS = S;
      d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
      memcpy(H, h, 9 * sizeof(double));
    }
  }
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  return maxS;
}

static double h_tol3(const double *h) {   // exp_ranH.c:877-885
  double tol = h[8];
  if (tol == 0) {
    for (int i = 0; i < 9; ++i) tol += h[i] * h[i];
    tol = std::sqrt(tol);
    tol *= 0.001;
  }
  return tol * tol * tol;
}

}  // namespace mods

// ---- host-only self-test hooks of the LO step's fast forms (no device needed) ----------------------------------------
extern "C" {
// type 0 Sampson, 1 symmetric sum, 2 symmetric max; lanes 0 = the scalar functions above, 1 / 4 / 8 = ransac_simd at that
// width (MODS_E_ARG when this CPU lacks it)
int mods_test_host_errfn(int type, const double *u6, int len, const double *H, int lanes, double *out) {
  if (!u6 || !H || !out || len < 1 || type < 0 || type > 2) return MODS_E_ARG;
  if (lanes == 0) {
    if (type == 0) HDs(nullptr, u6, H, out, len); else if (type == 1) HDsSym(nullptr, u6, H, out, len); else HDsSymMax(nullptr, u6, H, out, len);
    return MODS_OK;
  }
  PointsSoA pts;
  pts.ops = rs::simd_ops_lanes(lanes);
  if (!pts.ops) return MODS_E_ARG;
  pts.build(u6, len);
  std::vector<double> d(pts.n_pad);
  ErrFn f = {type == 0 ? ERR_SAMPSON : type == 1 ? ERR_SYMSUM : ERR_SYMMAX, nullptr, &pts};
  f(u6, H, d.data(), len);
  memcpy(out, d.data(), sizeof(double) * len);
  return MODS_OK;
}
// u2h (Htools.c:100-132) of the correspondences inl[0..n): reference_form 1 = lin_hgN + cov_mat as written there, 0 = cov_hgN
int mods_test_host_u2h(const double *u6, const int *inl, int n, int reference_form, double *H) {
  if (!u6 || !inl || !H || n < 4) return MODS_E_ARG;
  std::vector<double> buffer((size_t)18 * n + 81);
  rs::u2h(u6, inl, n, H, buffer.data(), reference_form != 0);
  return MODS_OK;
}
// the 9 x 9 moment matrix alone: reference_form 1 = normu + lin_hgN + cov_mat as written in the reference, 0 = the product's path
// (the 30 folded sums through the host SIMD table), 2 = the folded sums in scalar code, 100 + lanes = the table of that
// lane count (MODS_E_ARG when this CPU lacks it)
int mods_test_host_cov(const double *u6, const int *inl, int n, int reference_form, double *Cv) {
  if (!u6 || !inl || !Cv || n < 1) return MODS_E_ARG;
  double A1[3], A2[3];
  if (reference_form >= 100) {
    const rs::SimdOps *ops = rs::simd_ops_lanes(reference_form - 100);
    if (!ops) return MODS_E_ARG;
    double sums[30];
    rs::normu(u6, inl, n, A1, A2);
    ops->cov_hg_all(u6, inl, n, A1, A2, sums);
    rs::cov_hgN_unfold(sums, Cv);
    return MODS_OK;
  }
  rs::normu(u6, inl, n, A1, A2);
  if (reference_form == 0) { rs::cov_hgN(u6, inl, n, A1, A2, Cv); return MODS_OK; }
  if (reference_form == 1) {
    std::vector<double> Z((size_t)18 * n);
    rs::lin_hgN(u6, Z.data(), inl, n, A1, A2);
    rs::cov_mat(Cv, Z.data(), 2 * n, 9);
  } else rs::cov_hgN_scalar(u6, inl, n, A1, A2, Cv);
  return MODS_OK;
}
// inlidxs (rtools.c:155-166): lanes 0 = scalar
int mods_test_host_inlidxs(const double *err, int len, double th, int lanes, int *inl, unsigned *I, double *J) {
  if (!err || !inl || !I || !J || len < 1) return MODS_E_ARG;
  rs::Score s;
  if (lanes == 0) s = rs::inlidxs(err, len, th, inl);
  else if (lanes < 0) {     // the list-only form (count and list; J stays 0)
    LoState L = {};
    L.len = len;
    s = inlidxs_list(L, err, th, inl);
  } else {
    PointsSoA pts;
    pts.ops = rs::simd_ops_lanes(lanes);
    if (!pts.ops) return MODS_E_ARG;
    pts.len = len; pts.n_pad = (len + rs::SIMD_PAD - 1) / rs::SIMD_PAD * rs::SIMD_PAD;
    pts.gains.assign(pts.n_pad, 0.0);
    std::vector<double> e(pts.n_pad, 0.0);
    memcpy(e.data(), err, sizeof(double) * len);
    LoState L = {};
    L.len = len; L.pts = &pts;
    s = inlidxs_v(L, e.data(), th, inl);
  }
  *I = s.I; *J = s.J;
  return MODS_OK;
}
}  // extern "C"

static Score ransac_h_run(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                                    int iter_type, int *data_out, int oriented_constraint, unsigned inlLimit, double **resids,
                                    HDsPtr HDS1, HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck) {
  (void)HDSi1; (void)HDSidx1;
  Score maxS = {0, 0}, maxSs = {0, 0}, S = {0, 0};
  const int RESIDS_M = 2 + 10 * (1 + 4 + 1);
  if (resids) *resids = (double *)malloc(0 * sizeof(double) + 8);
  if (len < 4 || !u || !H || !inl || !data_out) { if (data_out) { data_out[0] = 0; data_out[1] = 0; data_out[2] = 0; } return maxS; }
  const double t_call0 = rsprof_on() ? rs_now_us() : 0;
  RansacGpu *ws = ransac_gpu();
  if (!ws) ransac_fail();   // no CPU fallback
  int err_type; HDsPtr custom = nullptr;
  if (HDS1 == &HDs || HDS1 == nullptr) err_type = ERR_SAMPSON;
  else if (HDS1 == &HDsSym) err_type = ERR_SYMSUM;
  else if (HDS1 == &HDsSymMax) err_type = ERR_SYMMAX;
  else { err_type = -1; custom = HDS1; }   // foreign error function: evaluated where it lives, on the host
  if (inlLimit == 0) inlLimit = 1000000;

  const long pinned = ransac_pinned_seed();
  rs::GlibcRand rng;
  rng.seed((unsigned)(pinned >= 0 ? (time_t)pinned : time(NULL)));   // srand(time(NULL)), exp_ranH.c:823
  rs::HashTable ht;

  std::vector<int> pool(len), inliers(len);
  for (int i = 0; i < len; i++) pool[i] = i;
  PointsSoA pts;
  pts.build(u, len);
  std::vector<double> err((size_t)pts.n_pad * 4), d_check(len);   // (u2h's scratch matrix is for its reference form only: not used here)
  double *errs[5];
  for (int i = 0; i < 4; i++) errs[i] = err.data() + (size_t)i * pts.n_pad;
  errs[4] = errs[3];
  int no_sam = 0, iter_cnt = 0, no_rej = 0, iterID = 0;
  unsigned seed = (unsigned)rng.next();   // seed = rand()
  double h[9];
  const double CHECK_COEF = 9.0;
  const unsigned MIN_GOOD_SYM_PTS = 5;
  const double th_check = CHECK_COEF * th;
  LoMemo memo;
  ErrFn errfn = {err_type, custom, &pts, &memo};
  LoState L = {u, len, th, {errs[0], errs[1], errs[2], errs[3], errs[4]}, nullptr, &rng, &ht, inlLimit, errfn, &pts, &memo};

  const double t_up0 = rsprof_on() ? rs_now_us() : 0;
  if (!ransac_ws_reserve(ws, len, 64)) ransac_fail();
  if (hipMemcpyAsync(ws->u_dev, u, sizeof(double) * 6 * len, hipMemcpyHostToDevice, ws->stream) != hipSuccess ||
      mods::stream_wait(ws->stream) != hipSuccess) { set_error("upload of the correspondences failed"); ransac_fail(); }
  if (rsprof_on()) g_rsprof.us[6] += rs_now_us() - t_up0;

  // sym check of a host-side model (LO results); the per-sample check comes from the GPU counts
  auto sym_bad = [&](const double *hh) -> bool {
    HDsSym(nullptr, u, hh, d_check.data(), len);
    unsigned c = 0;
    for (int j = 0; j < len; j++) if (d_check[j] <= th_check) c++;
    return c <= MIN_GOOD_SYM_PTS;
  };
  // LO block of the main loop, exp_ranH.c:961-1074 (iter_type 4 = inner RANSAC + iterated LSQ; other
  // types follow the same switch)
  auto run_lo = [&](bool *new_max) {
    RsTimer t_lo(5);
    iter_cnt++;
    // the residual rows of every LO (exp_ranH.c:961-968: RESIDS_M rows of len doubles each) are kept for a caller that asks for them
    double *rbase = nullptr;
    if (resids) {
      *resids = (double *)realloc(*resids, (size_t)iter_cnt * RESIDS_M * len * sizeof(double));
      rbase = *resids + (size_t)RESIDS_M * (iter_cnt - 1) * len;
    }
    double *d;
    switch (iter_type) {
      case 0: break;
      case 1:
        S = inlidxs_v(L, L.errs[4], 4 * th, inliers.data());
        u2h_memo(&memo, u, inliers.data(), (int)S.I, h);
        d = L.errs[0];
        errfn(u, h, d, len);
        S.I = 0; S.J = 0;
        for (int j = 0; j < len; j++) { if (d[j] <= th) S.I++; S.J += rs::trunc_quad(d[j], th); }
        break;
      case 2:
        S = lo_iter(L, inliers.data(), th, 4 * th, 4, h, ++iterID, rbase ? rbase + 2 * len : nullptr);
        break;
      case 3:
        d = L.errs[0];
        S = inlidxs_v(L, L.errs[4], 4 * th, inliers.data());
        u2h_memo(&memo, u, inliers.data(), (int)S.I, h);
        errfn(u, h, d, len);
        S = inlidxs_v(L, d, th, inliers.data());
        break;
      default: {
        if (rbase) memcpy(rbase, L.errs[4], len * sizeof(double));
        d = L.errs[0];
        S = inlidxs_list(L, L.errs[4], 4 * th * 2, inliers.data());   // TC*th*MWM
        u2h_memo(&memo, u, inliers.data(), (int)S.I, h);
        errfn(u, h, d, len);
        S = inlidxs_list(L, d, th, inliers.data());   // (lo_inner takes the count; what it returns replaces S)
        if (rbase) memcpy(rbase + len, d, len * sizeof(double));
        S = lo_inner(L, inliers.data(), (int)S.I, th, h, 10, &iterID, rbase ? rbase + 2 * len : nullptr);
        break;
      }
    }
    const double tol = h_tol3(h);
    if (rs::score_less(maxS, S) && (std::fabs(rs::det3(h) / tol) > 10e-2)) {
      bool bad_model = false;
      if (doSymCheck) bad_model = sym_bad(h);
      if (!bad_model) {
        double *dd = L.errs[0]; L.errs[0] = L.errs[3]; L.errs[3] = dd;
        maxS = S;
        *new_max = true;
        memcpy(H, h, 9 * sizeof(double));
      }
    }
  };

  // ---- main loop, batched ---------------------------------------------------------------------------
  struct Sample { unsigned seed_before; int idx[4]; int valid; double h[9]; };
  std::vector<Sample> batch;
  // (the reference samples rows of the 2len x 9 design matrix lin_hg made of all correspondences, exp_ranH.c:849, rtools.c:127-151;
  // the two rows of a sampled correspondence are written from its coordinates instead - the same products, Htools.c:19-58 - so the
  // 18 len doubles are neither filled nor read)
  // batches: 8 samples first - a pair with many inliers stops after three to five -, then 64, 128, ... as before; the samples and
  // their order do not depend on the batching
  int batch_size = 8;
  bool bad_model = false;
  while (no_sam < max_sam) {
    int want = batch_size;
    if (want > max_sam - no_sam) want = max_sam - no_sam;
    batch.resize(want);
    if (!ransac_ws_reserve(ws, len, want)) ransac_fail();
    int n_valid = 0;
    const double t_hyp0 = rsprof_on() ? rs_now_us() : 0;
    for (int b = 0; b < want; b++) {
      Sample &sm = batch[b];
      sm.seed_before = seed;
      rng.seed(seed);                               // srand(seed)
      double M[9 * 9];
      for (int i = 0; i < 4; i++) {                 // multirsampleT(Z, 9, 2, pool, 4, len, M), rtools.c:127-151
        const int s = rng.next() % (len - i);
        const int j = len - i - 1;
        const int q = pool[s];
        pool[s] = pool[j];
        pool[j] = q;
        const double *sq = u + 6 * (size_t)q;
        double *r0 = M + (2 * i) * 9, *r1 = r0 + 9;
        r0[0] = sq[3]; r0[3] = sq[4]; r0[6] = sq[5];
        r0[1] = 0; r0[4] = 0; r0[7] = 0;
        r0[2] = -sq[0] * sq[3]; r0[5] = -sq[0] * sq[4]; r0[8] = -sq[0] * sq[5];
        r1[0] = 0; r1[3] = 0; r1[6] = 0;
        r1[1] = sq[3]; r1[4] = sq[4]; r1[7] = sq[5];
        r1[2] = -sq[1] * sq[3]; r1[5] = -sq[1] * sq[4]; r1[8] = -sq[1] * sq[5];
      }
      seed = (unsigned)rng.next();                  // seed = rand()
      for (int i = 0; i < 4; i++) sm.idx[i] = pool[len - 4 + i];
      sm.valid = 0;
      if (oriented_constraint && !rs::all_Hori_valid(u, sm.idx)) { sm.valid = -1; continue; }   // counts as no_rej
      for (int i = 9 * 8; i < 9 * 9; ++i) M[i] = 0.0;
      double sol[9 * 9];
      int nb[18];
      memset(sol, 0, sizeof(sol));
      if (rs::nullspace(M, sol, 9, nb) != 1) continue;
      const double v = rs::det3(sol);
      const double tol = h_tol3(sol);
      if (std::fabs(v / tol) < 10e-2) continue;     // close to singular
      memcpy(sm.h, sol, sizeof(sm.h));
      sm.valid = 1;
      HypDev *hyp_host = (HypDev *)ws->hyp_host;
      HypDev &hd = hyp_host[n_valid];
      memcpy(hd.h, sol, sizeof(hd.h));
      rs::SymH sh; rs::sym_prepare(sol, &sh);
      memcpy(hd.Hinv, sh.Hinv, sizeof(hd.Hinv)); memcpy(hd.H1, sh.H1, sizeof(hd.H1));
      sm.valid = 1 + n_valid;                       // 1-based slot in the scored batch
      n_valid++;
    }
    if (rsprof_on()) g_rsprof.us[0] += rs_now_us() - t_hyp0;
    std::vector<double> custom_d;
    if (n_valid > 0) {
      RsTimer t_(1);
      if (custom) {
        // foreign error function: scores come from the caller's code, on the host
        custom_d.resize((size_t)n_valid * len);
        for (int kq = 0; kq < n_valid; kq++) {
          double *dd = custom_d.data() + (size_t)kq * len;
          custom(nullptr, u, ((HypDev *)ws->hyp_host)[kq].h, dd, len);
          unsigned I = 0, Is = 0; double J = 0;
          for (int j = 0; j < len; j++) { if (dd[j] <= th) I++; J += rs::trunc_quad(dd[j], th); }
          if (doSymCheck) { HDsSym(nullptr, u, ((HypDev *)ws->hyp_host)[kq].h, d_check.data(), len); for (int j = 0; j < len; j++) if (d_check[j] <= th_check) Is++; }
          ws->counts_host[2 * kq] = (int)I; ws->counts_host[2 * kq + 1] = (int)Is; ws->J_host[kq] = J;
        }
      } else if (!gpu_score(ws, len, n_valid, err_type, doSymCheck, th, th_check)) ransac_fail();
    }
    auto fetch_row = [&](int slot, double *dst) {
      RsTimer t_(2);
      if (custom) memcpy(dst, custom_d.data() + (size_t)slot * len, sizeof(double) * len);
      else if (!ransac_fetch_row(ws, len, slot, dst)) ransac_fail();
    };
    // replay the reference's per-iteration decisions in order (exp_ranH.c:858-1083)
    int b = 0;
    for (; b < want && no_sam < max_sam; b++) {
      const Sample &sm = batch[b];
      no_sam++;
      if (sm.valid == -1) { no_rej++; continue; }
      if (sm.valid == 0) continue;
      const int slot = sm.valid - 1;
      bool new_max = false;
      S.I = (unsigned)ws->counts_host[2 * slot];
      S.J = ws->J_host[slot];
      double *d = L.errs[0];
      bool have_d = false;
      if (rs::score_less(maxS, S)) {
        if (doSymCheck) bad_model = (unsigned)ws->counts_host[2 * slot + 1] <= MIN_GOOD_SYM_PTS;
        if (bad_model) continue;
        fetch_row(slot, d); have_d = true;
        L.errs[0] = L.errs[3];
        L.errs[3] = d;
        maxS = S;
        new_max = true;
        memcpy(H, sm.h, 9 * sizeof(double));
      }
      bool do_iterate;
      if (rs::score_less(maxSs, S)) {
        do_iterate = no_sam > 50;                   // ITER_SAM
        maxSs = S;
        if (!have_d) { fetch_row(slot, d); have_d = true; }
        L.errs[4] = d;
      } else do_iterate = false;
      if ((no_sam >= 50) && (iter_cnt == 0) && (maxSs.I > 4)) do_iterate = true;
      if (do_iterate) {
        // generator state as the reference has it here: srand(seed_k); 4 x random(); rand()
        rng.seed(sm.seed_before);
        for (int i = 0; i < 5; i++) (void)rng.next();
        memcpy(h, sm.h, sizeof(h));
        run_lo(&new_max);
      }
      if (new_max) {
        const int new_sam = rs::nsamples((int)maxS.I + 1, len, 4, conf);
        if (new_sam < max_sam) max_sam = new_sam;
      }
    }
    if (b < want) {
      // stopped inside the batch: the generator continues from the last executed iteration
      const Sample &last = batch[b - 1];
      rng.seed(last.seed_before);
      for (int i = 0; i < 5; i++) (void)rng.next();
      break;
    }
    batch_size = batch_size < 64 ? 64 : (batch_size < 1024 ? batch_size * 2 : batch_size);
  }
  // "If there were no LOs, do at least one NOW!", exp_ranH.c:1085-1197
  if (iter_cnt == 0 && iter_type != 0) {
    bool nm = false;
    memset(h, 0, sizeof(h));
    run_lo(&nm);
  }
  {
    const double *d = L.errs[3];
    for (int j = 0; j < len; j++) inl[j] = d[j] <= th ? 1 : 0;
  }
  if (rsprof_on()) {
    RsProf &p = g_rsprof;
    fprintf(stderr, "ransac H prof: len %d samples %d LO %d | call %.0f us: upload %.0f, hypotheses %.0f, gpu_score %.0f, fetch_row %.0f, LO %.0f (u2h %.0f, errfn %.0f)\n", len,
            no_sam, iter_cnt, rs_now_us() - t_call0, p.us[6], p.us[0], p.us[1], p.us[2], p.us[5], p.us[3], p.us[4]);
    p = RsProf();
  }
  data_out[0] = no_sam;
  data_out[1] = iter_type == 0 ? 0 : iter_cnt;
  data_out[2] = no_rej;
  return maxS;
}

extern "C" Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                                    int iter_type, int *data_out, int oriented_constraint, unsigned inlLimit, double **resids,
                                    HDsPtr HDS1, HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck) {
  mods::ransac_set_failed(0);
  try {
    return ransac_h_run(u, len, th, conf, max_sam, H, inl, iter_type, data_out, oriented_constraint, inlLimit, resids, HDS1, HDSi1, HDSidx1,
                        doSymCheck);
  } catch (const mods::RansacDeviceError &) {
    mods::ransac_set_failed(1);
    if (data_out) { data_out[0] = 0; data_out[1] = 0; data_out[2] = 0; }
    if (inl) memset(inl, 0, (size_t)(len > 0 ? len : 0));
    const Score none = {0, 0};
    return none;
  }
}
