// DEGENSAC fundamental-matrix verification: GPU hypothesis scoring + host control loop.
//
// Reference behaviour: exp_ransacFcustom, degensac/exp_ranF.c:805-1202 (LO: exp_inFranicustom
// :748-800, exp_iterFcustom :622-745; degeneracy handling: DegUtils.c), called from the useF branch
// of LORANSACFiltering, matching/matching.cpp:714-726.
//
// Structure (same idea as ransac.hip).  One 7-point sample per iteration comes from a libc
// generator that is re-seeded every iteration (srand(seed); 7 x random(); seed = rand(),
// exp_ranF.c:884-891), so the sample sequence is a function of the first seed only.  It is generated
// ahead of time on the host together with the 1..3 real solutions of each sample (9x9 null space +
// cubic), and a batch of candidate matrices is scored over all correspondences on the GPU:
//   score kernel : thread (correspondence i, candidate k): error d (Sampson or symmetric epipolar),
//                  truncated-quadratic gain, symmetric-check vote; counts by wave ballots
//   gain kernel  : MSAC score J_k = sum_i gain[i][k] in correspondence order (ransac.hip)
// The host replays the reference's decisions per sample and per root in order.  The rare branches
// stay on the host because they are chains of small data-dependent solves: the plane-degeneracy
// test of a new best sample, the homography LO, the local optimisation of F.  The one O(N * 10^4)
// piece of the degenerate branch, the plane-and-parallax search (rFtH), counts its two-point
// candidates on the GPU in blocks.
#include "common.hpp"
#include "ransac_f_host.hpp"
#include "ransac_gpu.hpp"
#include "ransac_soa.hpp"
#include "ransac_dev.hpp"
#include <ctime>
#include <cstdlib>
#include <memory>
#include <chrono>

namespace mods {

using rs::Score;

enum { FERR_SAMPSON = 0, FERR_SYM = 1 };

struct HypF { double f[9]; };
static_assert(sizeof(HypF) <= HYP_SLOT_BYTES, "hypothesis slot too small");

__device__ __forceinline__ double trunc_quad_f(double epsilon, double thr) {
  if (thr == 0) return 0;
  if (epsilon >= thr * 9 / 4) return 0;
  return 1 - (epsilon / (thr * 9 / 4));
}

// grid = (ceil(len/256), n_hyp), block 256.  d[k][i], gain[i][kstride], counts[k] = {I, Isym}.
__global__ void __launch_bounds__(256) ransacf_score_kernel(const double *__restrict__ u, int len, const HypF *__restrict__ hyp, int err_type,
                                                            int do_sym, double th, double th_check, double *__restrict__ d_out,
                                                            double *__restrict__ gain, int kstride, int *__restrict__ counts) {
  __shared__ double F[9];
  const int k = blockIdx.y;
  if (threadIdx.x < 9) F[threadIdx.x] = hyp[k].f[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool inl = false, inls = false;
  if (i < len) {
    double uu[6];
#pragma unroll
    for (int q = 0; q < 6; q++) uu[q] = u[(size_t)i * 6 + q];
    double d, ds = 0;
    if (err_type == FERR_SYM || do_sym) {
      const FTerms t = f_terms(uu, F);
      ds = t.r * t.r * (t.a + t.b) / (t.a * t.b);
    }
    d = err_type == FERR_SYM ? ds : fds_from(uu, F);
    d_out[(size_t)k * len + i] = d;
    gain[(size_t)i * kstride + k] = trunc_quad_f(d, th);
    inl = d <= th;
    inls = do_sym && ds <= th_check;
  }
  const unsigned long long m1 = __ballot(inl), m2 = __ballot(inls);
  if ((threadIdx.x & 63) == 0) {
    if (m1) atomicAdd(&counts[2 * k], __popcll(m1));
    if (m2) atomicAdd(&counts[2 * k + 1], __popcll(m2));
  }
}

// plane-and-parallax candidates: counts[k] = #{i < n : FDs(uN_i, F_k) < limit}, one workgroup per candidate (grid = k), with the candidates made on
// the device (round 6; until then the host made k x 9 matrices and uploaded them): a candidate is F = ([e]x H^T)^T with e through the two parallax
// lines of off-plane correspondences p0, p1 (rFtH, DegUtils.c:254-444) - ~80 fp64 operations that the host spent 30 us per block of
// 2048 on, and nine doubles per candidate to upload.  Here a block gets the two indices and thread 0 makes F with the host code's
// operations in the host code's order (rs::rFtH's `candidate`: crossprod x 3, the norm's sqrt and three divisions, skew_sym, the 3 x 3
// product with its zero terms, the transpose; no contraction on either side), so the counts are the host candidates' counts.
// us: per off-plane correspondence (x, y, 1 of image 1 | H^T-side point of image 2), as rFtH builds it; Ht: H transposed.
struct RfthHt { double v[9]; };
__global__ void __launch_bounds__(256) ransacf_count_pairs_kernel(const double *__restrict__ uN, const double *__restrict__ us, int n,
                                                                  const unsigned int *__restrict__ pairs, RfthHt Ht, double limit,
                                                                  int *__restrict__ counts) {
  __shared__ double F[9];
  __shared__ int s_c[4];
  const int k = blockIdx.x;
  if (threadIdx.x == 0) {
    const double *a0 = us + 6 * (size_t)pairs[2 * k], *a1 = us + 6 * (size_t)pairs[2 * k + 1];
    double c1[3], c2[3], ec[3], sk[9], prod[9];
    c1[0] = a0[1] * a0[5] - a0[2] * a0[4]; c1[1] = a0[2] * a0[3] - a0[0] * a0[5]; c1[2] = a0[0] * a0[4] - a0[1] * a0[3];
    c2[0] = a1[1] * a1[5] - a1[2] * a1[4]; c2[1] = a1[2] * a1[3] - a1[0] * a1[5]; c2[2] = a1[0] * a1[4] - a1[1] * a1[3];
    ec[0] = c1[1] * c2[2] - c1[2] * c2[1]; ec[1] = c1[2] * c2[0] - c1[0] * c2[2]; ec[2] = c1[0] * c2[1] - c1[1] * c2[0];
    const double ecNorm = sqrt(ec[0] * ec[0] + ec[1] * ec[1] + ec[2] * ec[2]);
    ec[0] = ec[0] / ecNorm; ec[1] = ec[1] / ecNorm; ec[2] = ec[2] / ecNorm;
    sk[0] = 0; sk[1] = -ec[2]; sk[2] = ec[1];
    sk[3] = ec[2]; sk[4] = 0; sk[5] = -ec[0];
    sk[6] = -ec[1]; sk[7] = ec[0]; sk[8] = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double sacc = 0.;
        for (int q = 0; q < 3; q++) sacc += sk[3 * i + q] * Ht.v[3 * q + j];
        prod[3 * i + j] = sacc;
      }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) F[3 * i + j] = prod[3 * j + i];
  }
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    double uu[6];
#pragma unroll
    for (int q = 0; q < 6; q++) uu[q] = uN[(size_t)i * 6 + q];
    c += fds_from(uu, F) < limit ? 1 : 0;
  }
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[k] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// counts[k * COUNT_PARTS + part] = #{i in the part : FDs(u_i, F_k) < limit} over ALL correspondences of the run, for the k models of
// innerFH's samples at once (DegUtils.c:476-584 evaluates them one after the other): grid = (k, COUNT_PARTS); Fs and counts are
// mapped host memory (a few KB each way), the parts are added on the host.
constexpr int COUNT_PARTS = 4;
__global__ void __launch_bounds__(256) ransacf_count_models_kernel(const double *__restrict__ u, int len, const double *__restrict__ Fs, double limit,
                                                                   int *__restrict__ counts) {
  __shared__ double F[9];
  __shared__ int s_c[4];
  const int k = blockIdx.x;
  if (threadIdx.x < 9) F[threadIdx.x] = Fs[(size_t)k * 9 + threadIdx.x];
  __syncthreads();
  int c = 0;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < len; i += 256 * COUNT_PARTS) {
    double uu[6];
#pragma unroll
    for (int q = 0; q < 6; q++) uu[q] = u[(size_t)i * 6 + q];
    c += fds_from(uu, F) < limit ? 1 : 0;
  }
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[k * COUNT_PARTS + blockIdx.y] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

static bool gpu_score_f(RansacGpu *ws, int len, int n, int err_type, int do_sym, double th, double th_check) {
  // (device counters: zero on entry, read out to the host's pinned result block and cleared again by ransac_gain_kernel; ransac.hip)
  if (!ransac_counts_begin(ws)) return false;
  RS_CHECK(hipMemcpyAsync(ws->hyp_dev, ws->hyp_host, sizeof(HypF) * n, hipMemcpyHostToDevice, ws->stream));
  hipLaunchKernelGGL(ransacf_score_kernel, dim3((len + 255) / 256, n), dim3(256), 0, ws->stream, ws->u_dev, len, (const HypF *)ws->hyp_dev,
                     err_type, do_sym, th, th_check, ws->d_dev, ws->gain_dev, ws->hyp_cap, ws->counts_dev);
  hipLaunchKernelGGL(ransac_gain_kernel, dim3((n + 63) / 64), dim3(256), 0, ws->stream, ws->gain_dev, len, n, ws->hyp_cap, ws->counts_dev, ws->J_host,
                     ws->counts_host);
  RS_CHECK(hipGetLastError());
  RS_CHECK(mods::stream_wait(ws->stream));
  ws->counts_dirty = false;
  ws->launches += 2;
  return true;
}

// PointEval as host SIMD across correspondences (ransac_simd.hpp: the scalar code's operations in the scalar code's order, 4
// or 8 correspondences per instruction).  One evaluation of 24 k correspondences takes ~25 us here against ~100 us for a
// launch + copy + synchronise round trip to the device (round 3's form, docs/history/r05_removed_paths.patch), and the
// degenerate branch of a large planar pair makes over a thousand of them (BASELINE configs[4]).
struct SimdEval : rs::PointEval {
  PointsSoA pts;
  SimdEval(const double *u_, int len_) : rs::PointEval(u_, len_) { pts.build(u_, len_); }
  // padded outputs of one evaluation; per calling thread, so that the tasks of ransac_pool.hpp can share one SimdEval
  static std::vector<double> &scratch(int which, size_t n) {
    static thread_local std::vector<double> v[2];
    if (v[which].size() < n) v[which].resize(n);
    return v[which];
  }
  void f(const double *F, int mode, double *p, double *w) {
    std::vector<double> &tp = scratch(0, pts.n_pad), &tw = scratch(1, pts.n_pad);
    pts.ops->fds_all(pts.col, pts.n_pad, F, mode, tp.data(), tw.data());
    memcpy(p, tp.data(), sizeof(double) * len);
    if (w) memcpy(w, tw.data(), sizeof(double) * len);
  }
  void hds(const double *H, double *out) override {
    std::vector<double> &tp = scratch(0, pts.n_pad);
    pts.ops->hds_all(pts.col, pts.n_pad, H, tp.data());
    memcpy(out, tp.data(), sizeof(double) * len);
  }
  void fds(const double *F, double *out) override { f(F, 0, out, nullptr); }
  void fds_sym(const double *F, double *out) override { f(F, 1, out, nullptr); }
  void exfds(const double *F, double *p, double *w) override { f(F, 2, p, w); }
  void exfds_sym(const double *F, double *p, double *w) override { f(F, 3, p, w); }
  bool concurrent() const override { return true; }
};

// off-plane set of rFtH (uN) and its plane-transferred form (us, may be null) -> aux_dev: uN in the first half of the buffer, us behind it
static bool gpu_upload_aux(RansacGpu *ws, const double *uN, const double *us, unsigned n) {
  if ((size_t)n * 12 > ws->aux_cap) {
    if (ws->aux_dev) RS_CHECK(hipFree(ws->aux_dev));
    ws->aux_cap = (size_t)n * 12 * 2;
    RS_CHECK(hipMalloc(&ws->aux_dev, ws->aux_cap * sizeof(double)));
  }
  RS_CHECK(hipMemcpyAsync(ws->aux_dev, uN, sizeof(double) * 6 * n, hipMemcpyHostToDevice, ws->stream));
  if (us) RS_CHECK(hipMemcpyAsync(ws->aux_dev + (size_t)6 * n, us, sizeof(double) * 6 * n, hipMemcpyHostToDevice, ws->stream));
  RS_CHECK(mods::stream_wait(ws->stream));
  return true;
}
// Count of k candidates given as index pairs into the off-plane set (pairs: 2 k unsigned; us sits behind uN in aux_dev), in two
// calls with a buffer slot (0 / 1) between them: begin queues upload, count and copy back on the workspace's stream and records the
// slot's event; end waits for that event only - the other slot's block may be in flight behind it.
static bool gpu_count_pairs_begin(RansacGpu *ws, int slot, unsigned n, const unsigned *pairs, int k, const double *Ht, double limit) {
  if (k > ws->cand_cap) {
    RS_CHECK(hipStreamSynchronize(ws->stream));                            // (no slot is in flight when the first block of a call is the largest)
    if (ws->cand_host) { RS_CHECK(hipHostFree(ws->cand_host)); RS_CHECK(hipHostFree(ws->candc_host)); }
    const int cap = k > 2048 ? k : 2048;                                   // rFtH's block size: never grown with a slot in flight
    ws->cand_cap = cap;
    // both in pinned host memory the kernel addresses itself (16 KB of index pairs in, 8 KB of counts out per block): a block of
    // candidates is ONE launch and one event, no copies to queue - the calls, not the work, were the cost of a block here
    // (coherent: what the kernel writes is in host memory when its event has completed, whatever HIP_HOST_COHERENT says)
    RS_CHECK(hipHostMalloc(&ws->cand_host, sizeof(unsigned) * 2 * 2 * cap, hipHostMallocMapped | hipHostMallocCoherent));
    RS_CHECK(hipHostMalloc(&ws->candc_host, sizeof(int) * 2 * cap, hipHostMallocMapped | hipHostMallocCoherent));
    RS_CHECK(hipHostGetDevicePointer((void **)&ws->cand_dev, ws->cand_host, 0));
    RS_CHECK(hipHostGetDevicePointer((void **)&ws->candc_dev, ws->candc_host, 0));
  }
  if (!ws->cand_ev[slot]) RS_CHECK(hipEventCreateWithFlags(&ws->cand_ev[slot], hipEventDisableTiming));
  memcpy(ws->cand_host + (size_t)slot * 2 * ws->cand_cap, pairs, sizeof(unsigned) * 2 * k);
  RfthHt ht;
  for (int i = 0; i < 9; i++) ht.v[i] = Ht[i];
  hipLaunchKernelGGL(ransacf_count_pairs_kernel, dim3(k), dim3(256), 0, ws->stream, ws->aux_dev, ws->aux_dev + (size_t)6 * n, (int)n,
                     (const unsigned int *)(ws->cand_dev + (size_t)slot * 2 * ws->cand_cap), ht, limit, ws->candc_dev + (size_t)slot * ws->cand_cap);
  RS_CHECK(hipGetLastError());
  RS_CHECK(hipEventRecord(ws->cand_ev[slot], ws->stream));
  ws->launches += 1;
  return true;
}
// the models of one round of innerFH samples counted over the run's correspondences (ws->u_dev), see ransacf_count_models_kernel
static bool gpu_count_models(RansacGpu *ws, int len, const double *Fs, int k, double limit, unsigned *counts) {
  if (k > ws->cntf_cap) {
    if (ws->cntf_host) { RS_CHECK(hipHostFree(ws->cntf_host)); RS_CHECK(hipHostFree(ws->cntc_host)); }
    const int cap = k > 256 ? k : 256;
    ws->cntf_cap = cap;
    RS_CHECK(hipHostMalloc(&ws->cntf_host, sizeof(double) * 9 * cap, hipHostMallocMapped | hipHostMallocCoherent));
    RS_CHECK(hipHostMalloc(&ws->cntc_host, sizeof(int) * COUNT_PARTS * cap, hipHostMallocMapped | hipHostMallocCoherent));
    RS_CHECK(hipHostGetDevicePointer((void **)&ws->cntf_dev, ws->cntf_host, 0));
    RS_CHECK(hipHostGetDevicePointer((void **)&ws->cntc_dev, ws->cntc_host, 0));
  }
  memcpy(ws->cntf_host, Fs, sizeof(double) * 9 * k);
  hipLaunchKernelGGL(ransacf_count_models_kernel, dim3(k, COUNT_PARTS), dim3(256), 0, ws->stream, (const double *)ws->u_dev, len, (const double *)ws->cntf_dev,
                     limit, ws->cntc_dev);
  RS_CHECK(hipGetLastError());
  RS_CHECK(mods::stream_wait(ws->stream));
  for (int j = 0; j < k; j++) {
    int c = 0;
    for (int q = 0; q < COUNT_PARTS; q++) c += ws->cntc_host[j * COUNT_PARTS + q];
    counts[j] = (unsigned)c;
  }
  ws->launches += 1;
  return true;
}
static bool gpu_count_pairs_end(RansacGpu *ws, int slot, int k, unsigned *counts) {
  hipError_t e = hipEventQuery(ws->cand_ev[slot]);                         // tens of microseconds: polled, no sleep
  while (e == hipErrorNotReady) e = hipEventQuery(ws->cand_ev[slot]);
  (void)hipGetLastError();                                                 // ("not ready" is not an error to leave behind)
  RS_CHECK(e);
  const int *ch = ws->candc_host + (size_t)slot * ws->cand_cap;
  for (int i = 0; i < k; i++) counts[i] = (unsigned)ch[i];
  return true;
}

}  // namespace mods

using namespace mods;

extern "C" {

typedef void (*FDsPtr)(const double *, const double *, double *, int);
typedef void (*exFDsPtr)(const double *, const double *, double *, double *, int);

// Error functions with the reference's signatures (Ftools.h, matching.cpp:44-70).  Host-side; used by
// the local optimisation and by the LAF check.
void FDs(const double *u, const double *F, double *p, int len) { rs::FDs_all(u, F, p, len); }
void FDsSym(const double *u, const double *F, double *p, int len) { rs::FDsSym_all(u, F, p, len); }
void FDsfull(const double *u, const double *F, double *p, int len) { rs::FDsSym_all(u, F, p, len); }   // SYMMETRIC_ERROR_CHECK is defined, Ftools.c:11
void exFDs(const double *u, const double *F, double *p, double *w, int len) { rs::exFDs_all(u, F, p, w, len); }
void exFDsSym(const double *u, const double *F, double *p, double *w, int len) { rs::exFDsSym_all(u, F, p, w, len); }

// ---- host-only self-test hooks (include/mods_hip.h) ----------------------------------------------------
int mods_test_seven_point(const double *u7, double *F27) {
  int id[7] = {0, 1, 2, 3, 4, 5, 6};
  double Z[9 * 7], A[9 * 9], sol[9 * 9], poly[4], roots[3];
  int nb[18];
  rs::lin_fm(u7, Z, id, 7);
  for (int i = 0; i < 7; i++)
    for (int c = 0; c < 9; c++) A[i * 9 + c] = Z[c * 7 + i];
  for (int i = 7 * 9; i < 9 * 9; ++i) A[i] = 0.0;
  memset(sol, 0, sizeof(sol));
  if (rs::nullspace(A, sol, 9, nb) != 2) return -1;
  rs::slcm(sol, sol + 9, poly);
  const int nsol = rs::rroots3(poly, roots);
  for (int i = 0; i < nsol; i++)
    for (int j = 0; j < 9; j++) F27[9 * i + j] = sol[j] * roots[i] + sol[9 + j] * (1 - roots[i]);
  return nsol;
}
void mods_test_u2f(const double *u, const int *idx, int n, const double *w, double *F) {
  std::vector<double> buffer((size_t)9 * n + 96);
  rs::u2fw(u, idx, w, n, F, buffer.data());
}
// reference_form 1: lin_fmN + row weights + cov_mat as written in Ftools.c:302-405; 0: cov_fmN (what u2fw runs)
void mods_test_u2f_form(const double *u, const int *idx, int n, const double *w, int reference_form, double *F) {
  std::vector<double> buffer((size_t)9 * n + 96);
  rs::u2fw(u, idx, w, n, F, buffer.data(), reference_form != 0);
}
// the 9 x 9 moment matrix of u2f / u2fw alone; lanes 0: scalar loop, 1 / 4 / 8: host SIMD across the sums
int mods_test_cov_fm(const double *u, const int *idx, int n, const double *w, int lanes, double *Cv81) {
  if (lanes != 0 && !rs::simd_ops_lanes(lanes)) return MODS_E_ARG;
  double A1[3], A2[3];
  rs::normu(u, idx, n, A1, A2);
  rs::cov_fmN(u, idx, w, n, A1, A2, Cv81, lanes);
  return MODS_OK;
}
int mods_test_checksample(const double *F, const double *u7, double th, double *H) { return rs::checksample(F, u7, th, H); }
unsigned mods_test_inner_h(unsigned seed, double *H, const double *u, unsigned len, double th, unsigned iters, unsigned char *inl) {
  rs::GlibcRand g;
  g.seed(seed);
  std::vector<double> buffer((size_t)18 * len + 96);
  return rs::innerH(H, u, len, th, iters, inl, g, buffer.data());
}
int mods_ransac_host_threads(void) { return rs::TaskPool::get().threads(); }
// the same through the host SIMD evaluation of the production path (simd = 0: scalar); *next_rand = the generator's next value
// after the call; *path = 0 one-thread loop, 1 repetitions side by side, 2 side by side and redone by the one-thread loop
unsigned mods_test_inner_h2(unsigned seed, double *H, const double *u, unsigned len, double th, unsigned iters, unsigned char *inl, int simd,
                            int *next_rand, int *path) {
  rs::GlibcRand g;
  g.seed(seed);
  std::vector<double> buffer((size_t)18 * len + 96);
  SimdEval sev(u, (int)len);
  const unsigned I = rs::innerH(H, u, len, th, iters, inl, g, buffer.data(), simd ? &sev : nullptr);
  if (next_rand) *next_rand = g.next();
  if (path) *path = rs::g_innerh_path;
  return I;
}
unsigned mods_test_rfth(unsigned seed, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F) {
  rs::GlibcRand g;
  g.seed(seed);
  return rs::rFtH(g, u, hinl, th, H, len, F, nullptr, rs::PairCounter());
}
// the same with the host SIMD evaluation of the production path (0: the scalar PointEval) and the generator's next value
// after the call (what the rest of exp_ransacFcustom would draw next); prof10 (optional) receives and clears g_rfth_prof
unsigned mods_test_rfth2(unsigned seed, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F,
                         int simd, int *next_rand, double *prof10) {
  rs::GlibcRand g;
  g.seed(seed);
  SimdEval sev(u, (int)len);
  const unsigned I = rs::rFtH(g, u, hinl, th, H, len, F, nullptr, rs::PairCounter(), simd ? &sev : nullptr);
  if (next_rand) *next_rand = g.next();
  if (prof10) for (int i = 0; i < 10; i++) { prof10[i] = rs::g_rfth_prof[i]; rs::g_rfth_prof[i] = 0; }
  return I;
}

}  // extern "C"

namespace mods {

struct FLo {
  const double *u; int len;
  double *errs[5];
  double *buffer;
  rs::GlibcRand *rng;
  rs::HashTable *ht;
  unsigned inlLimit;
  std::function<void(const double *F, double *d)> fds;
  std::function<void(const double *F, double *d, double *w)> exfds;
};

// least squares on (a subset of) the inliers, exp_ranF.c:647-668 / 706-727 (__D3__: D3_F_RATIO 1, D3_F_MIN 0)
static void lo_lsq(FLo &L, int *inliers, unsigned n, const double *w, double *f) {
  unsigned detached = (unsigned)(int)(n * 1);
  if (detached > L.inlLimit) detached = L.inlLimit;
  if (detached < 8) detached = 8;
  if (detached >= n) rs::u2fw(L.u, inliers, w, (int)n, f, L.buffer);
  else {
    int *sub = rs::randsubset(*L.rng, inliers, (int)n, (int)detached);
    rs::u2fw(L.u, sub, w, (int)detached, f, L.buffer);
  }
}

// exp_iterFcustom, exp_ranF.c:622-745
static Score lo_iter_f(FLo &L, int *inliers, double th, double ths, int iters, double *F, int iterID, double *resids) {
  const int len = L.len;
  double *d = L.errs[1];
  double f[9];
  Score S = {0, 0}, Ss, maxS;
  std::vector<double> w(len);
  const double dth = (ths - th) / 4;   // ILSQ_ITERS
  maxS = rs::inlidxs(L.errs[4], len, th, inliers);
  if (maxS.I < 8) return S;
  S = rs::inlidxs(L.errs[4], len, th * 2, inliers);   // th*MWM
  lo_lsq(L, inliers, S.I, nullptr, f);
  for (int it = 0; it < iters; it++) {
    L.exfds(f, d, w.data());
    memcpy(resids + (size_t)it * len, d, len * sizeof(double));
    S = rs::inlidxs(d, len, th, inliers);
    const uint32_t hash = rs::super_fast_hash((const char *)inliers, (int)(S.I * sizeof(*inliers)));
    const int ret = L.ht->contains(hash, (int)S.I, iterID);
    if (ret != -1 && ret != iterID) { S.I = 0; S.J = 0; return S; }
    if (ret == -1) L.ht->insert(hash, (int)S.I, iterID);
    if (rs::score_less(maxS, S)) {
      maxS = S;
      L.errs[1] = L.errs[0];
      L.errs[0] = d;
      d = L.errs[1];
      memcpy(F, f, 9 * sizeof(double));
    }
    // the reference takes the next LSQ support from `d` AFTER the exchange above, i.e. from the buffer
    // that has just become the scratch one (exp_ranF.c:700) - kept as is
    Ss = rs::inlidxs(d, len, ths * 2, inliers);
    if (Ss.I < 8) return maxS;
    lo_lsq(L, inliers, Ss.I, w.data(), f);
    ths -= dth;
  }
  L.fds(f, d);
  memcpy(resids + (size_t)4 * len, d, len * sizeof(double));
  S = rs::inlidxs(d, len, th, inliers);
  if (rs::score_less(maxS, S)) {
    maxS = S;
    L.errs[1] = L.errs[0];
    L.errs[0] = d;
    memcpy(F, f, 9 * sizeof(double));
  }
  return maxS;
}

// exp_inFranicustom, exp_ranF.c:748-800 (RAN_REP 10, ILSQ_ITERS 4, TC 4)
static Score lo_inner_f(FLo &L, int *inliers, int ninl, double th, double *F, int *iterID, double *resids) {
  const int len = L.len;
  Score S = {0, 0}, maxS = {0, 0};
  double *d, f[9];
  std::vector<int> intbuff(len);
  if (ninl < 16) {
    memset(resids, 0, (size_t)(62 - 2) * len * sizeof(double));   // RESIDS_M - 2
    return maxS;
  }
  int ssiz = ninl / 2;
  if (ssiz > 14) ssiz = 14;
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  for (int i = 0; i < 10; i++) {
    int *sample = rs::randsubset(*L.rng, inliers, ninl, ssiz);
    rs::u2f(L.u, sample, ssiz, f, L.buffer);
    L.fds(f, L.errs[0]);
    memcpy(resids + (size_t)i * 6 * len, L.errs[0], len * sizeof(double));
    L.errs[4] = L.errs[0];
    S = lo_iter_f(L, intbuff.data(), th, 4 * th, 4, f, ++*iterID, resids + (size_t)i * 6 * len + len);
    if (rs::score_less(maxS, S)) {
      maxS = S;
      d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
      memcpy(F, f, 9 * sizeof(double));
    }
  }
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  return maxS;
}

}  // namespace mods

static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define F_FATAL() mods::ransac_fail()

// host-only self-test hook: mode 0 FDs, 1 FDsSym, 2 exFDs, 3 exFDsSym over all correspondences; lanes 0 = the scalar loops,
// 1 / 4 / 8 = ransac_simd at that width (MODS_E_ARG when this CPU lacks it)
extern "C" int mods_test_host_fds(int mode, const double *u6, int len, const double *F, int lanes, double *p, double *w) {
  if (!u6 || !F || !p || len < 1 || mode < 0 || mode > 3 || (mode >= 2 && !w)) return MODS_E_ARG;
  if (lanes == 0) {
    if (mode == 0) rs::FDs_all(u6, F, p, len); else if (mode == 1) rs::FDsSym_all(u6, F, p, len);
    else if (mode == 2) rs::exFDs_all(u6, F, p, w, len); else rs::exFDsSym_all(u6, F, p, w, len);
    return MODS_OK;
  }
  const rs::SimdOps *ops = rs::simd_ops_lanes(lanes);
  if (!ops) return MODS_E_ARG;
  SimdEval ev(u6, len);
  ev.pts.ops = ops;
  ev.f(F, mode, p, mode >= 2 ? w : nullptr);
  return MODS_OK;
}

static int ransac_f_run(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
                                  int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih, exFDsPtr EXFDS1, FDsPtr FDS1,
                                  int doSymCheck) {
  (void)H_best;   // the reference copies zero elements into it (exp_ranF.c:1199)
  const int RESIDS_M = 2 + 10 * (1 + 4 + 1);
  const int ITER_SAM = 50;
  const double CHECK_COEF = 16.0, SYMM_COEF = 0.6;
  if (resids) *resids = (double *)malloc(8);
  if (Ih) *Ih = 0;
  if (len < 7 || !u || !F || !inl || !data_out || !resids) { if (data_out) { data_out[0] = 0; data_out[1] = 0; } return 0; }
  RansacGpu *ws = ransac_gpu();
  if (!ws) F_FATAL();   // no CPU fallback
  if (!FDS1) FDS1 = &FDs;
  if (!EXFDS1) EXFDS1 = FDS1 == &FDsSym ? &exFDsSym : &exFDs;
  int err_type = FDS1 == &FDs ? FERR_SAMPSON : FDS1 == &FDsSym ? FERR_SYM : -1;   // -1: foreign error function, evaluated on the host

  const bool prof = ransac_profile_on();   // MODS_RANSAC_PROF
  const double t_begin = prof ? wall_ms() : 0;
  double t_innerh = 0, t_rfth = 0, t_lo = 0, t_gen = 0, t_score = 0, t_setup = 0;
  const long pinned = ransac_pinned_seed();
  rs::GlibcRand rng, gen;
  rng.seed((unsigned)(pinned >= 0 ? (time_t)pinned : time(NULL)));   // srand(time(NULL)), exp_ranF.c:832
  rs::HashTable ht;

  std::vector<int> pool(len), inliers(len), bufferP(len);
  for (int i = 0; i < len; i++) pool[i] = i;
  std::vector<double> Z((size_t)len * 9), buffer((size_t)len * 18 + 96), err((size_t)len * 4), errorsBest(len), HDsv(len), d_check(len);
  rs::lin_fm(u, Z.data(), pool.data(), len);
  FLo L;
  L.u = u; L.len = len; L.buffer = buffer.data(); L.rng = &rng; L.ht = &ht; L.inlLimit = inlLimit;
  for (int i = 0; i < 4; i++) L.errs[i] = err.data() + (size_t)i * len;
  L.errs[4] = L.errs[3];
  double **errs = L.errs;

  Score maxS = {8, 0}, maxSs = {8, 0}, S = {0, 0};
  int no_sam = 0, iter_cnt = 0, degen_cnt = 0, iterID = 0, Ihmax = 0;
  unsigned non_degen_samples_count = 0;
  int samidxBest[7] = {0, 0, 0, 0, 0, 0, 0};
  double FBest[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, H[9], f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool bad_model = false;
  const double th_check = CHECK_COEF * th;
  unsigned seed = (unsigned)rng.next();   // seed = rand()

  if (!ransac_ws_reserve(ws, len, 96)) F_FATAL();
  if (hipMemcpyAsync(ws->u_dev, u, sizeof(double) * 6 * len, hipMemcpyHostToDevice, ws->stream) != hipSuccess ||
      mods::stream_wait(ws->stream) != hipSuccess) { set_error("upload of the correspondences failed"); F_FATAL(); }
  // O(len) evaluations of one model (LO steps, degenerate branch): host SIMD across the correspondences
  SimdEval simd_eval(u, len);
  rs::PointEval *ev = &simd_eval;
  simd_eval.count_fds = [&](const double *Fs, int k, double limit, unsigned *counts) {
    if (!gpu_count_models(ws, len, Fs, k, limit, counts)) F_FATAL();
  };
  auto eval_fds = [&](const double *Fm, double *dd) {
    if (FDS1 == &FDs) ev->fds(Fm, dd);
    else if (FDS1 == &FDsSym) ev->fds_sym(Fm, dd);
    else FDS1(u, Fm, dd, len);
  };
  L.fds = eval_fds;
  L.exfds = [&](const double *Fm, double *dd, double *ww) {
    if (EXFDS1 == &exFDs) ev->exfds(Fm, dd, ww);
    else if (EXFDS1 == &exFDsSym) ev->exfds_sym(Fm, dd, ww);
    else EXFDS1(u, Fm, dd, ww, len);
  };

  // plane-and-parallax search with its two-point candidates counted on the GPU
  unsigned aux_n = 0;
  auto upload_offplane = [&](const double *uN, const double *us, unsigned n) { aux_n = n; if (!gpu_upload_aux(ws, uN, us, n)) F_FATAL(); };
  rs::PairCounter count_pairs;
  count_pairs.begin = [&](int slot, const unsigned *pairs, int k, const double *Ht) {
    if (!gpu_count_pairs_begin(ws, slot, aux_n, pairs, k, Ht, th * 2)) F_FATAL();
  };
  count_pairs.end = [&](int slot, int k, unsigned *counts) {
    if (!gpu_count_pairs_end(ws, slot, k, counts)) F_FATAL();
  };

  // LO block of the main loop, exp_ranF.c:1032-1070 (__LSQ_BEFORE_LO__); `source` = errs[4] in the loop,
  // errorsBest in the closing "ALO" run
  auto run_lo = [&](const double *source, bool *new_max) {
    iter_cnt++;
    *resids = (double *)realloc(*resids, (size_t)iter_cnt * RESIDS_M * len * sizeof(double));
    double *rbase = *resids + (size_t)RESIDS_M * (iter_cnt - 1) * len;
    memcpy(rbase, errs[4], len * sizeof(double));
    double *d = errs[0];
    S = rs::inlidxs(source, len, 4 * th * 2, inliers.data());   // TC*th*MWM
    rs::u2f(u, inliers.data(), (int)S.I, f, buffer.data());
    eval_fds(f, d);
    S = rs::inlidxs(d, len, th, inliers.data());
    memcpy(rbase + len, d, len * sizeof(double));
    S = lo_inner_f(L, inliers.data(), (int)S.I, th, f, &iterID, rbase + 2 * len);
    if (rs::score_less(maxS, S)) {
      d = errs[0]; errs[0] = errs[3]; errs[3] = d;
      maxS = S;
      memcpy(F, f, 9 * sizeof(double));
      *new_max = true;
    }
  };

  struct Sample {
    unsigned seed_before;
    int idx[7];        // samidx[0..6] = pool[len-7 .. len-1]
    int nsol;          // -1: null space not two-dimensional (iteration skipped)
    double f[3][9];
    int slot[3];       // scored slot, -1 = rejected by the orientation constraint
  };
  std::vector<Sample> batch;
  int batch_size = 8;            // (the first round trip is the latency of an easy pair: 24 candidates instead of 96; doubles per batch up to 512)
  int tag[4] = {-1, -1, -1, -1};   // per error buffer: batch slot of the candidate last evaluated into it (-1: content is on the host)
  bool rng_live = true;          // does `rng` hold the generator state the reference has at this point?
  unsigned live_seed = 0;
  auto ensure_rng = [&]() {
    if (rng_live) return;
    rng.seed(live_seed);                       // srand(seed); 7 x random(); seed = rand()
    for (int i = 0; i < 8; i++) (void)rng.next();
    rng_live = true;
  };

  if (prof) t_setup = wall_ms() - t_begin;
  while (no_sam < max_sam) {
    int want = batch_size;
    if (want > max_sam - no_sam) want = max_sam - no_sam;
    batch.resize(want);
    if (!ransac_ws_reserve(ws, len, 3 * want)) F_FATAL();
    HypF *hyp_host = (HypF *)ws->hyp_host;
    int n_hyp = 0;
    const double tg0 = prof ? wall_ms() : 0;
    for (int b = 0; b < want; b++) {
      Sample &sm = batch[b];
      sm.seed_before = seed;
      gen.seed(seed);
      double A[9 * 9], sol[9 * 9];
      int nb[18];
      for (int i = 0; i < 7; i++) {                 // rsampleT(Z, 9, pool, 7, len, A), rtools.c:12-23, 83-101
        const int s = gen.next() % (len - i);
        const int j = len - i - 1;
        const int q = pool[s];
        pool[s] = pool[j];
        pool[j] = q;
        for (int c = 0; c < 9; c++) A[i * 9 + c] = Z[(size_t)c * len + q];
      }
      seed = (unsigned)gen.next();
      for (int i = 0; i < 7; i++) sm.idx[i] = pool[len - 7 + i];
      for (int i = 7 * 9; i < 9 * 9; ++i) A[i] = 0.0;
      memset(sol, 0, sizeof(sol));
      sm.nsol = -1;
      if (rs::nullspace(A, sol, 9, nb) != 2) continue;
      double poly[4], roots[3];
      double *f1 = sol, *f2 = sol + 9;
      rs::slcm(f1, f2, poly);
      sm.nsol = rs::rroots3(poly, roots);
      for (int i = 0; i < sm.nsol; i++) {
        for (int j = 0; j < 9; j++) sm.f[i][j] = f1[j] * roots[i] + f2[j] * (1 - roots[i]);
        sm.slot[i] = -1;
        if (!rs::all_ori_valid(sm.f[i], u, sm.idx, 7)) continue;
        memcpy(hyp_host[n_hyp].f, sm.f[i], sizeof(sm.f[i]));
        sm.slot[i] = n_hyp++;
      }
    }
    const double tg1 = prof ? wall_ms() : 0;
    t_gen += tg1 - tg0;
    std::vector<double> host_d;
    if (n_hyp > 0) {
      if (err_type < 0) {
        host_d.resize((size_t)n_hyp * len);
        for (int kq = 0; kq < n_hyp; kq++) {
          double *dd = host_d.data() + (size_t)kq * len;
          FDS1(u, hyp_host[kq].f, dd, len);
          unsigned I = 0, Is = 0; double J = 0;
          for (int j = 0; j < len; j++) { if (dd[j] <= th) I++; J += rs::trunc_quad(dd[j], th); }
          if (doSymCheck) { FDsSym(u, hyp_host[kq].f, d_check.data(), len); for (int j = 0; j < len; j++) if (d_check[j] <= th_check) Is++; }
          ws->counts_host[2 * kq] = (int)I; ws->counts_host[2 * kq + 1] = (int)Is; ws->J_host[kq] = J;
        }
      } else if (!gpu_score_f(ws, len, n_hyp, err_type, doSymCheck, th, th_check)) F_FATAL();
    }
    if (prof) t_score += wall_ms() - tg1;
    std::vector<int> cnt(ws->counts_host, ws->counts_host + 2 * n_hyp);
    std::vector<double> Jv(ws->J_host, ws->J_host + n_hyp);
    auto fetch_row = [&](int slot, double *dst) {
      if (err_type < 0) memcpy(dst, host_d.data() + (size_t)slot * len, sizeof(double) * len);
      else if (!ransac_fetch_row(ws, len, slot, dst)) F_FATAL();
    };
    // The reference evaluates every candidate straight into errs[i] (exp_ranF.c:920-921), so a buffer
    // that an older pointer still refers to (errs[4] after its buffer has rotated out of errs[3]) holds
    // the errors of whichever candidate was written there LAST.  The rows stay on the GPU; each of the
    // four buffers carries the slot of its last writer and is filled in when somebody reads it.
    auto buf_of = [&](const double *p) { return (int)((p - err.data()) / len); };
    auto materialise = [&](double *p) {
      const int bi = buf_of(p);
      if (tag[bi] >= 0) { fetch_row(tag[bi], p); tag[bi] = -1; }
    };
    auto materialise_all = [&]() { for (int q = 0; q < 4; q++) materialise(err.data() + (size_t)q * len); };

    // replay of the reference's per-iteration decisions (exp_ranF.c:880-1079)
    int b = 0;
    for (; b < want && no_sam < max_sam; b++) {
      const Sample &sm = batch[b];
      no_sam++;
      rng_live = false; live_seed = sm.seed_before;
      if (sm.nsol < 0) continue;
      double u7[6 * 7];
      for (int i = 0; i < 7; i++) memcpy(u7 + 6 * i, u + 6 * sm.idx[i], 6 * sizeof(double));
      bool new_max = false, do_iterate = false;
      int LmaxI = 0;
      for (int i = 0; i < sm.nsol; i++) {
        memcpy(f, sm.f[i], sizeof(f));
        if (sm.slot[i] < 0) continue;
        const int slot = sm.slot[i];
        double *d = errs[i];
        tag[buf_of(d)] = slot;
        S.I = (unsigned)cnt[2 * slot];
        S.J = Jv[slot];
        if ((int)S.I > LmaxI) LmaxI = (int)S.I;
        if (rs::score_less(maxS, S)) {
          if (doSymCheck) {
            const int SI_min = (int)std::floor(SYMM_COEF * S.I);
            bad_model = cnt[2 * slot + 1] <= SI_min;
          }
          if (bad_model) continue;
          materialise(d);
          errs[i] = errs[3];
          errs[3] = d;
          maxS = S;
          memcpy(F, f, 9 * sizeof(double));
          new_max = true;
        }
        if (rs::score_less(maxSs, S)) {
          maxSs = S;
          if (rs::checksample(f, u7, 3 * th, H)) {
            ev->hds(H, HDsv.data());
            unsigned I = 0;
            for (int j = 0; j < len; ++j) if (HDsv[j] < th * 3) ++I;
            if (I < 8) break;
            ensure_rng();
            const double tp0 = prof ? wall_ms() : 0;
            I = rs::innerH(H, u, (unsigned)len, 16 * th, 10, inl, rng, buffer.data(), ev);
            if (prof) t_innerh += wall_ms() - tp0;
            if ((int)I > Ihmax) Ihmax = (int)I;      // the reference also keeps H here but never hands it out (exp_ranF.c:1199)
            if (I > 6) {
              materialise_all();
              const double tp1 = prof ? wall_ms() : 0;
              I = rs::rFtH(rng, u, inl, th, H, (unsigned)len, f, upload_offplane, count_pairs, ev);
              if (prof) t_rfth += wall_ms() - tp1;
              if (I > maxS.I) {
                eval_fds(f, errs[3]);
                maxS.I = I;                       // maxS.J follows below
                memcpy(F, f, 9 * sizeof(double));
                new_max = true;
                d = errs[3];
              } else {
                eval_fds(f, errs[i]);
                d = errs[i];
              }
              double jj = 0;
              for (int j = 0; j < len; j++) jj += rs::trunc_quad(d[j], th);
              if (new_max) maxS.J = jj;
              ++degen_cnt;
            }
          } else {
            do_iterate = (do_lo > 0 && (no_sam > ITER_SAM));
            materialise(d);
            errs[4] = d;
            non_degen_samples_count++;
            memcpy(samidxBest, sm.idx, sizeof(samidxBest));
            memcpy(errorsBest.data(), d, len * sizeof(double));
            memcpy(FBest, f, sizeof(FBest));
          }
        }
      }
      data_out[LmaxI + 2]++;
      if (do_lo > 0 && (no_sam == ITER_SAM) && non_degen_samples_count) do_iterate = true;
      if (do_iterate) {
        ensure_rng();
        materialise_all();
        const double tp2 = prof ? wall_ms() : 0;
        run_lo(errs[4], &new_max);
        if (prof) t_lo += wall_ms() - tp2;
      }
      if (new_max) {
        const int new_sam = rs::nsamples((int)maxS.I + 1, len, 7, conf);
        if (new_sam < max_sam) max_sam = new_sam;
      }
    }
    materialise_all();          // the rows of this batch are about to be overwritten
    if (b < want) break;        // stopped inside the batch
    if (batch_size < 512) batch_size *= 2;
  }

  // "If there were no LOs, do at least one NOW!", exp_ranF.c:1082-1163.  The sample kept in
  // samidxBest/FBest is one that checksample() has already classified as non-degenerate (it is
  // recorded only in that branch, :1023-1029) and the test is deterministic, so the degenerate arm of
  // the reference's closing block cannot be taken; only its LO arm exists here.
  if (do_lo && (!iter_cnt && !degen_cnt) && non_degen_samples_count) {
    ensure_rng();
    bool nm = false;
    run_lo(errorsBest.data(), &nm);
  }

  {
    const double *d = errs[3];
    for (int j = 0; j < len; j++) inl[j] = d[j] <= th ? 1 : 0;
  }
  data_out[0] = no_sam;
  data_out[1] = iter_cnt;
  if (prof) { fprintf(stderr, "[mods ransacF] rFtH: candidates %.2f ms, counting %.2f ms, %.0f blocks, %.0f off-plane; innerFH %.0f calls (+%.0f run ahead and dropped) in %.0f rounds %.2f ms on %d threads, %.0f fits in u2Fit %.2f thread-ms; set-up %.2f ms, triggers %.2f ms, folding %.2f ms, count launches %.2f ms, innerFH stages %.2f + %.2f ms\n", rs::g_rfth_prof[0], rs::g_rfth_prof[1], rs::g_rfth_prof[2], rs::g_rfth_prof[3], rs::g_rfth_prof[6], rs::g_rfth_prof[8], rs::g_rfth_prof[9], rs::g_rfth_prof[4], rs::TaskPool::get().threads(), rs::g_rfth_prof[7], rs::g_rfth_prof[5], rs::g_rfth_prof[10], rs::g_rfth_prof[11], rs::g_rfth_prof[12], rs::g_rfth_prof[13], rs::g_rfth_prof[14], rs::g_rfth_prof[15]); for (double &x : rs::g_rfth_prof) x = 0; }
  if (prof) fprintf(stderr, "[mods ransacF] len %d samples %d lo %d degen %d | total %.2f ms: set-up %.2f samples %.2f scoring %.2f innerH %.2f rFtH %.2f LO %.2f\n", len, no_sam, iter_cnt,
                    degen_cnt, wall_ms() - t_begin, t_setup, t_gen, t_score, t_innerh, t_rfth, t_lo);
  if (Ih) *Ih = Ihmax;
  return (int)maxS.I;
}

extern "C" int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
                                  int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih, exFDsPtr EXFDS1, FDsPtr FDS1,
                                  int doSymCheck) {
  mods::ransac_set_failed(0);
  try {
    return ransac_f_run(u, len, th, conf, max_sam, F, inl, data_out, do_lo, inlLimit, resids, H_best, Ih, EXFDS1, FDS1, doSymCheck);
  } catch (const mods::RansacDeviceError &) {
    mods::ransac_set_failed(1);
    if (data_out) { data_out[0] = 0; data_out[1] = 0; }
    if (inl) memset(inl, 0, (size_t)(len > 0 ? len : 0));
    if (Ih) *Ih = 0;
    return 0;
  }
}
