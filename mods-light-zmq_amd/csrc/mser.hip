// MSER detector (reference: DetectMSERs, detectors/mser/extrema/extrema.cpp:196-295, behind
// DetectAffineRegions<>, imagerepresentation.cpp:780-783).
//
// Division of labour (see mser_host.hpp for why the growth itself is host code):
//   device  mser_u8_kernel        the float view -> 8-bit image ("(unsigned char)*in_ptr", extrema.cpp:228-230)
//   host    mser::Grower          grey-level growth of every (image, polarity), one per thread: stable (slot, threshold, margin)
//                                 list + the merge tree (pix_slot, tpar, tlev); keypoint order / selection mode (a sort of a
//                                 few thousand integers with std::sort's own tie order, extrema.cpp:31-90)
//   device  mser_inner_kernel     per pixel: the smallest stable region that contains it (walk of the merge tree)
//           mser_nest_kernel      per stable region: the next larger stable region
//           mser_runs_kernel      per pixel: start / end of a row run for every stable region whose left / right neighbour
//                                 lies outside it - what RegionBoundaries + ReduceBoundary + ReducedBoundary2RLE produce by
//                                 one flood fill per region (boundary.cpp:99-208, libExtrema.cpp:162-252)
//           rocprim radix sort    runs of every region in (line, column) order
//           mser_ellipse_kernel   per region: RLE2Ellipse (libExtrema.cpp:117-160) over its runs IN THAT ORDER (double sums),
//                                 Matrix2::schur_sym / sqrt -> A (utls/matrix.cpp:185-216), one lane per region
//           mser_export_kernel    the DetectAffineRegions loop (synth-detection.hpp:96-110) into the context's key list
// Everything downstream (orientation, description, banks, matching) is the common path.
#include "common.hpp"
#include "mser_host.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>

namespace mods {

struct MserState {
  unsigned char *img8 = nullptr; size_t img8_cap = 0;           // [n_img][h][w]
  int32_t *pix_slot = nullptr; uint32_t *tpar = nullptr; uint8_t *tlev = nullptr; int32_t *inner = nullptr; size_t px_cap = 0;   // [jobs][P]
  int *tab = nullptr; size_t tab_cap = 0;
  unsigned long long *keys = nullptr; size_t keys_cap = 0;      // 4 arrays of keys_cap: starts in / out, ends in / out
  void *sort_tmp = nullptr; size_t sort_tmp_cap = 0;
  double *ell = nullptr; size_t ell_cap = 0;                    // [n_u][6]
  unsigned long long *counters = nullptr;                       // device: starts, ends, errors
  unsigned char *h_img8 = nullptr; size_t h_img8_cap = 0;       // pinned
  int32_t *h_pix_slot = nullptr; uint32_t *h_tpar = nullptr; uint8_t *h_tlev = nullptr; size_t h_px_cap = 0;
  int *h_tab = nullptr; size_t h_tab_cap = 0;
  unsigned long long *h_counters = nullptr;
};

void mser_release(mods_ctx *c) {
  MserState *s = (MserState *)c->mser;
  if (!s) return;
  (void)hipFree(s->img8); (void)hipFree(s->pix_slot); (void)hipFree(s->tpar); (void)hipFree(s->tlev); (void)hipFree(s->inner);
  (void)hipFree(s->tab); (void)hipFree(s->keys); (void)hipFree(s->sort_tmp); (void)hipFree(s->ell); (void)hipFree(s->counters);
  (void)hipHostFree(s->h_img8); (void)hipHostFree(s->h_pix_slot); (void)hipHostFree(s->h_tpar); (void)hipHostFree(s->h_tlev);
  (void)hipHostFree(s->h_tab); (void)hipHostFree(s->h_counters);
  delete s;
  c->mser = nullptr;
}

template <typename T>
static int grow_dev(T *&p, size_t &cap, size_t need) {
  if (need <= cap) return MODS_OK;
  if (p) MODS_HIP_CHECK(hipFree(p));
  p = nullptr; cap = 0;
  MODS_HIP_CHECK(hipMalloc(&p, need * sizeof(T)));
  cap = need;
  return MODS_OK;
}
template <typename T>
static int grow_host(T *&p, size_t need) {
  if (p) MODS_HIP_CHECK(hipHostFree(p));
  p = nullptr;
  MODS_HIP_CHECK(hipHostMalloc(&p, need * sizeof(T)));
  return MODS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
struct MserTabs {
  const int *ss_slot, *ss_begin;      // distinct slots with stable thresholds (sorted per job) -> first entry of the u table
  const int *u_slot, *u_thresh;       // unique (slot, threshold) pairs, sorted per job by (slot, threshold)
  int *u_parent;                      // next larger stable region (global u index) or -1
  const int *u_area;                  // area the growth counted for it
  const int *job_ss_off, *job_u_off;  // [jobs + 1]
};

__global__ __launch_bounds__(256) void mser_u8_kernel(const float *__restrict__ img, int w, int h, int stride, unsigned char *__restrict__ out) {
  const size_t n = (size_t)w * h;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
  const float v = img[(size_t)blockIdx.y * stride * h + (size_t)y * stride + x];
  out[(size_t)blockIdx.y * n + i] = (unsigned char)(unsigned)(int)v;   // cvttss2si, low byte
}

// first stable region met on the way up from slot s for a pixel that is part of s's component from level lv on
__device__ __forceinline__ int mser_first_stable(const MserTabs &T, const uint32_t *__restrict__ tpar, const uint8_t *__restrict__ tlev,
                                                 int job, int s, int lv) {
  for (;;) {
    const uint32_t tp = tpar[s];
    const uint32_t par = tp & mser::kNoParent;
    const int top = par == mser::kNoParent ? 256 : (int)tlev[s];       // s is a root for levels < top
    if (tp & mser::kHasStable) {
      int lo = T.job_ss_off[job], hi = T.job_ss_off[job + 1];
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (T.ss_slot[mid] < s) lo = mid + 1; else hi = mid; }
      for (int k = T.ss_begin[lo], e = T.ss_begin[lo + 1]; k < e; k++) {
        const int t = T.u_thresh[k];
        if (t >= lv && t < top) return k;
      }
    }
    if (par == mser::kNoParent) return -1;
    lv = top;
    s = (int)par;
  }
}

__global__ __launch_bounds__(256) void mser_inner_kernel(MserTabs T, const unsigned char *__restrict__ img8, int w, int h, size_t P,
                                                         const int32_t *__restrict__ pix_slot, const uint32_t *__restrict__ tpar,
                                                         const uint8_t *__restrict__ tlev, int32_t *__restrict__ inner) {
  const int job = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)w * h) return;
  const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
  const size_t ofs = (size_t)(y + 1) * (w + 2) + x + 1;
  int lv = img8[(size_t)(job >> 1) * w * h + i];
  if (job & 1) lv = 255 - lv;
  inner[job * P + ofs] = mser_first_stable(T, tpar + job * P, tlev + job * P, job, pix_slot[job * P + ofs], lv);
}

__global__ __launch_bounds__(256) void mser_nest_kernel(MserTabs T, int n_jobs, size_t P, const uint32_t *__restrict__ tpar,
                                                        const uint8_t *__restrict__ tlev) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= T.job_u_off[n_jobs]) return;
  int job = 0;
  while (u >= T.job_u_off[job + 1]) job++;
  const int s = T.u_slot[u];
  int parent;
  if (u + 1 < T.job_u_off[job + 1] && T.u_slot[u + 1] == s) parent = u + 1;      // the same region at its next threshold
  else {
    const uint32_t par = tpar[job * P + s] & mser::kNoParent;
    parent = par == mser::kNoParent ? -1 : mser_first_stable(T, tpar + job * P, tlev + job * P, job, (int)par, (int)tlev[job * P + s]);
  }
  T.u_parent[u] = parent;
}

__device__ __forceinline__ bool mser_in_chain(const int *__restrict__ u_parent, int u, int b) {
  for (; b >= 0; b = u_parent[b]) if (b == u) return true;
  return false;
}

// write = 0: count the run starts / ends (counters[0], [1]); write = 1: emit keys (region << 32 | padded offset)
__global__ __launch_bounds__(256) void mser_runs_kernel(MserTabs T, int w, int h, size_t P, const int32_t *__restrict__ inner, int write,
                                                        unsigned long long *__restrict__ counters, unsigned long long *__restrict__ starts,
                                                        unsigned long long *__restrict__ ends, size_t cap) {
  const int job = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int a = -1, bl = -1, br = -1;
  size_t ofs = 0;
  if (i < (size_t)w * h) {
    const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
    ofs = (size_t)(y + 1) * (w + 2) + x + 1;
    const int32_t *in = inner + job * P;
    a = in[ofs];
    if (a >= 0) {
      bl = x > 0 ? in[ofs - 1] : -1;
      br = x + 1 < w ? in[ofs + 1] : -1;
    }
  }
  // the regions of a's chain that do not contain the neighbour: a prefix of the chain (an ancestor of a common region is common)
  int ns = 0, ne = 0;
  if (a >= 0) {
    if (bl != a) for (int u = a; u >= 0 && !mser_in_chain(T.u_parent, u, bl); u = T.u_parent[u]) ns++;
    if (br != a) for (int u = a; u >= 0 && !mser_in_chain(T.u_parent, u, br); u = T.u_parent[u]) ne++;
  }
  int is = ns, ie = ne;
  for (int d = 1; d < 64; d <<= 1) {
    const int vs = __shfl_up(is, d), ve = __shfl_up(ie, d);
    if (lane >= d) { is += vs; ie += ve; }
  }
  const int ts = __shfl(is, 63), te = __shfl(ie, 63);
  if (ts == 0 && te == 0) return;
  unsigned long long bs = 0, be = 0;
  if (lane == 63) {
    if (ts) bs = atomicAdd(&counters[0], (unsigned long long)ts);
    if (te) be = atomicAdd(&counters[1], (unsigned long long)te);
  }
  if (!write) return;
  bs = __shfl(bs, 63); be = __shfl(be, 63);
  size_t ps = bs + (is - ns), pe = be + (ie - ne);
  if (ns) for (int u = a, k = 0; k < ns; k++, u = T.u_parent[u]) { if (ps < cap) starts[ps] = ((unsigned long long)u << 32) | ofs; ps++; }
  if (ne) for (int u = a, k = 0; k < ne; k++, u = T.u_parent[u]) { if (pe < cap) ends[pe] = ((unsigned long long)u << 32) | ofs; pe++; }
}

// One lane per stable region: RLE2Ellipse over its runs in (line, column) order, then C -> A = U sqrt(T) U^T.
__global__ __launch_bounds__(64) void mser_ellipse_kernel(MserTabs T, int n_u, int w, const unsigned long long *__restrict__ starts,
                                                          const unsigned long long *__restrict__ ends, size_t n_runs, double *__restrict__ ell,
                                                          unsigned long long *__restrict__ counters) {
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= n_u) return;
  size_t lo = 0, hi = n_runs;
  const unsigned long long k0 = (unsigned long long)u << 32, k1 = (unsigned long long)(u + 1) << 32;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (starts[mid] < k0) lo = mid + 1; else hi = mid; }
  const size_t i0 = lo;
  hi = n_runs;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (starts[mid] < k1) lo = mid + 1; else hi = mid; }
  const size_t i1 = lo;
  const int cols = w + 2;
  double area = 0, sumX = 0, sumY = 0;
  bool bad = false;
  for (size_t i = i0; i < i1; i++) {
    const unsigned so = (unsigned)starts[i], eo = (unsigned)ends[i];
    bad |= (ends[i] >> 32) != (unsigned long long)u || so / cols != eo / cols || eo < so;
    const double line = (double)((int)(so / cols) - 1), m = (double)((int)(so % cols) - 1), n = (double)(1 + ((int)(eo % cols) - 1));
    sumX += (n * n - m * m) / 2;
    sumY += (n - m) * (2 * line + 1) / 2;
    area += n - m;
  }
  if (bad || area != (double)T.u_area[u]) atomicAdd(&counters[2], 1ull);   // the tree and the growth's counters disagree: a bug
  const double barX = sumX / area, barY = sumY / area;
  double sumX2 = 0, sumY2 = 0, sumXY = 0;
  for (size_t i = i0; i < i1; i++) {
    const unsigned so = (unsigned)starts[i], eo = (unsigned)ends[i];
    const double line = (double)((int)(so / cols) - 1) - barY;
    const double m = (double)((int)(so % cols) - 1) - barX, n = (double)(1 + ((int)(eo % cols) - 1)) - barX;
    const double l2 = line * line, m2 = m * m, n2 = n * n;
    sumX2 += (n2 * n - m2 * m) / 3;
    sumY2 += (n - m) * (3 * l2 + 3 * line + 1) / 3;
    sumXY += -.25 * (m2 - n2) * (2 * line + 1);
  }
  sumX2 /= area; sumY2 /= area; sumXY /= area;
  // utls::Matrix2::schur_sym (utls/matrix.cpp:185-216)
  double t, r;
  if (sumXY != 0) {
    r = (sumY2 - sumX2) / (2 * sumXY);
    if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
    else t = -1.0 / (-r + sqrt(1 + r * r));
    r = 1.0 / sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  const double q00 = r, q01 = t, q10 = -t, q11 = r;                    // Q; Q^T = [q00 q10; q01 q11]
  const double m00 = q00 * sumX2 + q10 * sumXY, m01 = q00 * sumXY + q10 * sumY2;   // Q^T * C
  const double m10 = q01 * sumX2 + q11 * sumXY, m11 = q01 * sumXY + q11 * sumY2;
  const double t00 = m00 * q00 + m01 * q10, t11 = m10 * q01 + m11 * q11;           // (Q^T C) Q, off-diagonal set to 0
  const double s00 = sqrt(t00), s11 = sqrt(t11), s01 = sqrt(0.0), s10 = sqrt(0.0);
  const double us00 = q00 * s00 + q01 * s10, us01 = q00 * s01 + q01 * s11;         // U * sqrt(T)
  const double us10 = q10 * s00 + q11 * s10, us11 = q10 * s01 + q11 * s11;
  double *o = ell + (size_t)u * 6;
  o[0] = barX; o[1] = barY;
  o[2] = us00 * q00 + us01 * q01;                                                  // ... * U^T
  o[3] = us00 * q10 + us01 * q11;
  o[4] = us10 * q00 + us11 * q01;
  o[5] = us10 * q10 + us11 * q11;
}

// out_* : the keypoints in output order; DetectAffineRegions (synth-detection.hpp:96-110): s = 1 * sqrt|det A|, rectifyTransformation
__global__ __launch_bounds__(256) void mser_export_kernel(int n_out, const int *__restrict__ out_u, const int *__restrict__ out_margin,
                                                          const int *__restrict__ out_sub, const int *__restrict__ out_img,
                                                          const int *__restrict__ out_pos, const int *__restrict__ u_slot,
                                                          const int *__restrict__ u_thresh, int w, const double *__restrict__ ell, int max_cand,
                                                          mods_affkey *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int u = out_u[i];
  const double *e = ell + (size_t)u * 6;
  const double a = e[2], bb = e[3], c = e[4], d = e[5];
  mods_affkey o;
  o.x = e[0]; o.y = e[1];
  o.s = 1.0 * sqrt(fabs(a * d - bb * c));
  const double det = sqrt(fabs(a * d - bb * c));
  const double b2a2 = sqrt(bb * bb + a * a);
  o.a11 = b2a2 / det;
  o.a12 = 0;
  o.a21 = (d * bb + c * a) / (b2a2 * det);
  o.a22 = det / b2a2;
  o.response = (double)out_margin[i];
  o.sub_type = out_sub[i];
  const int slot = u_slot[u];
  o.octave = u_thresh[u]; o.level = out_sub[i] == 20 ? 1 : 0; o.r0 = slot / (w + 2) - 1; o.c0 = slot % (w + 2) - 1; o.pad = 0;
  keys[(size_t)out_img[i] * max_cand + out_pos[i]] = o;
}

// ---------------------------------------------------------------------------------------------------------------------
struct MserOut { double response; int u, sub; };

int mser_detect(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride, const mods_hessaff_params *par, double tilt,
                double zoom) {
  if (w <= 0 || h <= 0 || n_img <= 0 || n_img > c->batch) { set_error("bad image batch %dx%d x%d (ctx batch %d)", w, h, n_img, c->batch); return MODS_E_ARG; }
  // the context's per-image scratch planes (tmp_dev takes the packed copy of a strided input below, the describe stage reads a plane
  // of this size) hold max_w * max_h pixels: the same check as pyramid_configure, which this detector does not go through
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image %dx%d larger than the context (%dx%d)", w, h, c->max_w, c->max_h); return MODS_E_ARG; }
  if (par->mserMinSize < 1 || !(par->mserMaxArea > 0) || !(par->mserMinMargin >= 1)) {
    set_error("MSER: min_size >= 1, max_area > 0 and min_margin >= 1 are required (got %d, %g, %g)", par->mserMinSize, par->mserMaxArea, par->mserMinMargin);
    return MODS_E_ARG;
  }
  if (par->mode < 0 || par->mode > MODS_DET_NOT_LESS_THAN_REGIONS) { set_error("unknown detector mode %d", par->mode); return MODS_E_ARG; }
  if (par->mode == MODS_DET_RELATIVE_REG_NUMBER && !(par->relativeRegionsNumber >= 0.f && par->relativeRegionsNumber <= 1.f)) { set_error("relativeRegionsNumber must be in [0, 1]"); return MODS_E_ARG; }
  if (par->mode == MODS_DET_NOT_LESS_THAN_REGIONS && par->regionsNumber < 0) { set_error("regionsNumber must be >= 0 in this mode"); return MODS_E_ARG; }
  if ((size_t)(w + 2) * (h + 2) >= (1ull << 31)) { set_error("MSER: image too large"); return MODS_E_ARG; }
  if (!c->mser) c->mser = new MserState();
  MserState &S = *(MserState *)c->mser;
  const int n_jobs = 2 * n_img, cols = w + 2, rows = h + 2;
  const size_t npx = (size_t)w * h, P = (size_t)rows * cols;
  int rc;
  if ((rc = grow_dev(S.img8, S.img8_cap, npx * n_img))) return rc;
  if (P * n_jobs > S.px_cap) {
    size_t cap = 0;
    if ((rc = grow_dev(S.pix_slot, cap, P * n_jobs))) return rc;
    cap = 0; if ((rc = grow_dev(S.tpar, cap, P * n_jobs))) return rc;
    cap = 0; if ((rc = grow_dev(S.tlev, cap, P * n_jobs))) return rc;
    cap = 0; if ((rc = grow_dev(S.inner, cap, P * n_jobs))) return rc;
    S.px_cap = P * n_jobs;
  }
  if (npx * n_img > S.h_img8_cap) { if ((rc = grow_host(S.h_img8, npx * n_img))) return rc; S.h_img8_cap = npx * n_img; }
  if (P * n_jobs > S.h_px_cap) {
    if ((rc = grow_host(S.h_pix_slot, P * n_jobs)) || (rc = grow_host(S.h_tpar, P * n_jobs)) || (rc = grow_host(S.h_tlev, P * n_jobs))) return rc;
    S.h_px_cap = P * n_jobs;
  }
  if (!S.counters) MODS_HIP_CHECK(hipMalloc(&S.counters, 4 * sizeof(unsigned long long)));
  if (!S.h_counters) MODS_HIP_CHECK(hipHostMalloc(&S.h_counters, 4 * sizeof(unsigned long long)));
  c->par = *par;
  c->last_w = w; c->last_h = h; c->last_n_img = n_img;
  c->last_img_dev = img_dev; c->last_stride = stride;
  c->pyr.n_oct = 0;                                  // no scale space behind these keys
  int *key_count = c->cand_count + 2 * c->batch;
  MODS_HIP_CHECK(hipMemsetAsync(c->cand_count, 0, sizeof(int) * 3 * c->batch, c->stream));
  if (stride != w)   // the describe stage reads dense planes from tmp_dev in this case (as pyramid_build leaves them)
    MODS_HIP_CHECK(hipMemcpy2DAsync(c->tmp_dev, sizeof(float) * w, img_dev, sizeof(float) * stride, sizeof(float) * w, (size_t)h * n_img,
                                    hipMemcpyDeviceToDevice, c->stream));

  // 1. 8-bit image, to the host
  hipLaunchKernelGGL(mser_u8_kernel, dim3((unsigned)((npx + 255) / 256), n_img), dim3(256), 0, c->stream, img_dev, w, h, stride, S.img8);
  MODS_HIP_CHECK(hipGetLastError());
  MODS_HIP_CHECK(hipMemcpyAsync(S.h_img8, S.img8, npx * n_img, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));

  // 2. growth of every (image, polarity) on host threads
  mser::GrowParams gp;
  gp.min_size = par->mserMinSize; gp.max_area = par->mserMaxArea; gp.relative = false;
  gp.min_margin = par->mode != MODS_DET_FIXED_TH ? 1.0 : par->mserMinMargin;          // extrema.cpp:206-211
  std::vector<std::vector<mser::Stable>> stable(n_jobs);
  {
    std::atomic<int> next(0);
    std::atomic<bool> oom(false);
    auto work = [&]() {
      try {
        mser::Grower g;
        std::vector<uint8_t> padded(P, 0);
        for (int j; (j = next.fetch_add(1)) < n_jobs;) {
          const unsigned char *src = S.h_img8 + (size_t)(j >> 1) * npx;
          const bool inv = (j & 1) != 0;
          for (int y = 0; y < h; y++) {
            uint8_t *d = padded.data() + (size_t)(y + 1) * cols + 1;
            const unsigned char *s = src + (size_t)y * w;
            if (inv) for (int x = 0; x < w; x++) d[x] = (uint8_t)(255 - s[x]);          // InvertImageAndHistogram, sortPixels.cpp:134-153
            else std::memcpy(d, s, w);
          }
          mser::GrowParams q = gp;
          q.invert = inv;
          g.run(padded.data(), w, h, q, S.h_pix_slot + j * P, S.h_tpar + j * P, S.h_tlev + j * P, stable[j]);
        }
      } catch (...) { oom = true; }      // std::bad_alloc: nothing else throws in there
    };
    const int n_thr = std::max(1, std::min<int>(n_jobs, std::min(8u, std::max(1u, std::thread::hardware_concurrency()))));
    std::vector<std::thread> pool;
    try {
      for (int t = 1; t < n_thr; t++) pool.emplace_back(work);
    } catch (...) {}                     // fewer threads than asked for: the others (at least the caller) take the jobs
    work();
    for (auto &t : pool) t.join();
    if (oom) { set_error("MSER: the grey-level growth ran out of host memory"); return MODS_E_HIP; }
  }

  // 3. tables: unique (slot, threshold) pairs per job; the keypoints in the reference's order
  std::vector<int> ss_slot, ss_begin, u_slot, u_thresh, u_area, job_ss_off(n_jobs + 1, 0), job_u_off(n_jobs + 1, 0);
  std::vector<std::vector<int>> st_u(n_jobs);
  for (int j = 0; j < n_jobs; j++) {
    std::vector<std::pair<std::pair<int, int>, int>> v;
    for (const mser::Stable &s : stable[j]) v.push_back({{s.slot, s.thresh}, s.area});
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    const int base = (int)u_slot.size();
    for (size_t k = 0; k < v.size(); k++) {
      if (k == 0 || v[k].first.first != v[k - 1].first.first) { ss_slot.push_back(v[k].first.first); ss_begin.push_back(base + (int)k); }
      u_slot.push_back(v[k].first.first); u_thresh.push_back(v[k].first.second); u_area.push_back(v[k].second);
    }
    st_u[j].reserve(stable[j].size());
    for (const mser::Stable &s : stable[j]) {
      const auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(std::make_pair(s.slot, s.thresh), s.area));
      st_u[j].push_back(base + (int)(it - v.begin()));
    }
    job_ss_off[j + 1] = (int)ss_slot.size();
    job_u_off[j + 1] = (int)u_slot.size();
  }
  ss_begin.push_back((int)u_slot.size());
  const int n_u = (int)u_slot.size(), n_ss = (int)ss_slot.size();
  int reg_number = par->regionsNumber;
  if ((tilt > 2.0) || (zoom < 0.5)) reg_number = (int)floor(zoom * 2.0 * reg_number / tilt);   // extrema.cpp:201-202
  std::vector<int> out_u, out_margin, out_sub, out_img, out_pos, n_out(n_img, 0);
  for (int b = 0; b < n_img; b++) {
    std::vector<MserOut> keys;
    for (int pol = 0; pol < 2; pol++) {
      const int j = 2 * b + pol;
      for (size_t k = 0; k < stable[j].size(); k++) keys.push_back({(double)stable[j][k].margin, st_u[j][k], pol ? 20 : 21});
    }
    if (par->mode != MODS_DET_FIXED_TH && !keys.empty()) {              // prepareKeysForExport, extrema.cpp:31-90
      // std::sort as the reference calls it: the margins are small integers and the order inside a group of equal margins is
      // what libstdc++'s introsort leaves for this sequence (a function of positions and comparisons only)
      std::sort(keys.begin(), keys.end(), [](const MserOut &k1, const MserOut &k2) { return fabs(k1.response) > fabs(k2.response); });
      const int regNumber = (int)keys.size();
      auto above = [&](double thr) { int m = 0; while (m < regNumber && fabs(keys[m].response) > fabs(thr)) m++; return m; };
      int keep = regNumber;
      switch (par->mode) {
        case MODS_DET_RELATIVE_TH: keep = above(fabs(keys[0].response) * par->relativeThreshold); break;
        case MODS_DET_FIXED_REG_NUMBER: if (reg_number < regNumber && reg_number >= 0) keep = reg_number; break;
        case MODS_DET_RELATIVE_REG_NUMBER: keep = (int)floor(par->relativeRegionsNumber * (double)keys.size()); break;
        case MODS_DET_NOT_LESS_THAN_REGIONS: {
          const int fixTh = above(1.0);
          keep = fixTh < reg_number ? std::min(reg_number, regNumber) : std::min(fixTh, regNumber);
          break;
        }
      }
      keys.resize((size_t)std::max(0, std::min(keep, regNumber)));
    }
    if ((int)keys.size() > c->max_cand) { set_error("MSER: %zu regions > key capacity %d", keys.size(), c->max_cand); return MODS_E_CAPACITY; }
    n_out[b] = (int)keys.size();
    for (size_t k = 0; k < keys.size(); k++) {
      out_u.push_back(keys[k].u); out_margin.push_back((int)keys[k].response); out_sub.push_back(keys[k].sub);
      out_img.push_back(b); out_pos.push_back((int)k);
    }
  }
  const int n_keys = (int)out_u.size();
  MODS_HIP_CHECK(hipMemcpyAsync(key_count, n_out.data(), sizeof(int) * n_img, hipMemcpyHostToDevice, c->stream));
  if (n_keys == 0 || n_u == 0) { MODS_HIP_CHECK(mods::stream_wait(c->stream)); return MODS_OK; }

  // packed int table: ss_slot | ss_begin | u_slot | u_thresh | u_area | u_parent | job_ss_off | job_u_off | out_*[5]
  const size_t o_ss_slot = 0, o_ss_begin = o_ss_slot + n_ss, o_u_slot = o_ss_begin + n_ss + 1, o_u_thresh = o_u_slot + n_u,
               o_u_area = o_u_thresh + n_u, o_u_parent = o_u_area + n_u, o_job_ss = o_u_parent + n_u, o_job_u = o_job_ss + n_jobs + 1,
               o_out = o_job_u + n_jobs + 1, n_tab = o_out + 5 * (size_t)n_keys;
  if ((rc = grow_dev(S.tab, S.tab_cap, n_tab))) return rc;
  if (n_tab > S.h_tab_cap) { if ((rc = grow_host(S.h_tab, n_tab))) return rc; S.h_tab_cap = n_tab; }
  std::copy(ss_slot.begin(), ss_slot.end(), S.h_tab + o_ss_slot);
  std::copy(ss_begin.begin(), ss_begin.end(), S.h_tab + o_ss_begin);
  std::copy(u_slot.begin(), u_slot.end(), S.h_tab + o_u_slot);
  std::copy(u_thresh.begin(), u_thresh.end(), S.h_tab + o_u_thresh);
  std::copy(u_area.begin(), u_area.end(), S.h_tab + o_u_area);
  std::fill(S.h_tab + o_u_parent, S.h_tab + o_u_parent + n_u, -1);
  std::copy(job_ss_off.begin(), job_ss_off.end(), S.h_tab + o_job_ss);
  std::copy(job_u_off.begin(), job_u_off.end(), S.h_tab + o_job_u);
  std::copy(out_u.begin(), out_u.end(), S.h_tab + o_out);
  std::copy(out_margin.begin(), out_margin.end(), S.h_tab + o_out + n_keys);
  std::copy(out_sub.begin(), out_sub.end(), S.h_tab + o_out + 2 * (size_t)n_keys);
  std::copy(out_img.begin(), out_img.end(), S.h_tab + o_out + 3 * (size_t)n_keys);
  std::copy(out_pos.begin(), out_pos.end(), S.h_tab + o_out + 4 * (size_t)n_keys);
  MODS_HIP_CHECK(hipMemcpyAsync(S.tab, S.h_tab, n_tab * sizeof(int), hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(S.pix_slot, S.h_pix_slot, P * n_jobs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(S.tpar, S.h_tpar, P * n_jobs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(S.tlev, S.h_tlev, P * n_jobs, hipMemcpyHostToDevice, c->stream));
  MODS_HIP_CHECK(hipMemsetAsync(S.counters, 0, 4 * sizeof(unsigned long long), c->stream));
  MserTabs T;
  T.ss_slot = S.tab + o_ss_slot; T.ss_begin = S.tab + o_ss_begin; T.u_slot = S.tab + o_u_slot; T.u_thresh = S.tab + o_u_thresh;
  T.u_area = S.tab + o_u_area; T.u_parent = S.tab + o_u_parent; T.job_ss_off = S.tab + o_job_ss; T.job_u_off = S.tab + o_job_u;

  // 4. membership, nesting, runs
  const dim3 pgrid((unsigned)((npx + 255) / 256), n_jobs);
  hipLaunchKernelGGL(mser_inner_kernel, pgrid, dim3(256), 0, c->stream, T, S.img8, w, h, P, S.pix_slot, S.tpar, S.tlev, S.inner);
  hipLaunchKernelGGL(mser_nest_kernel, dim3((n_u + 255) / 256), dim3(256), 0, c->stream, T, n_jobs, P, S.tpar, S.tlev);
  hipLaunchKernelGGL(mser_runs_kernel, pgrid, dim3(256), 0, c->stream, T, w, h, P, S.inner, 0, S.counters, (unsigned long long *)nullptr,
                     (unsigned long long *)nullptr, (size_t)0);
  MODS_HIP_CHECK(hipGetLastError());
  MODS_HIP_CHECK(hipMemcpyAsync(S.h_counters, S.counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  const size_t n_runs = (size_t)S.h_counters[0];
  if (S.h_counters[1] != S.h_counters[0] || n_runs == 0 || n_runs >= (1ull << 32)) {
    set_error("MSER: %llu run starts, %llu run ends", S.h_counters[0], S.h_counters[1]);
    return MODS_E_HIP;
  }
  if (n_runs > S.keys_cap) {
    if (S.keys) MODS_HIP_CHECK(hipFree(S.keys));
    S.keys = nullptr; S.keys_cap = 0;
    const size_t cap = n_runs + n_runs / 4 + 1024;
    MODS_HIP_CHECK(hipMalloc(&S.keys, 4 * cap * sizeof(unsigned long long)));
    S.keys_cap = cap;
  }
  unsigned long long *s_in = S.keys, *s_out = S.keys + S.keys_cap, *e_in = S.keys + 2 * S.keys_cap, *e_out = S.keys + 3 * S.keys_cap;
  MODS_HIP_CHECK(hipMemsetAsync(S.counters, 0, 2 * sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(mser_runs_kernel, pgrid, dim3(256), 0, c->stream, T, w, h, P, S.inner, 1, S.counters, s_in, e_in, S.keys_cap);
  MODS_HIP_CHECK(hipGetLastError());
  int ubits = 1;
  while ((1ll << ubits) < n_u) ubits++;
  size_t tmp_bytes = 0;
  MODS_HIP_CHECK(rocprim::radix_sort_keys(nullptr, tmp_bytes, s_in, s_out, n_runs, 0, 32 + ubits, c->stream));
  if (tmp_bytes > S.sort_tmp_cap) {
    if (S.sort_tmp) MODS_HIP_CHECK(hipFree(S.sort_tmp));
    S.sort_tmp = nullptr; S.sort_tmp_cap = 0;
    MODS_HIP_CHECK(hipMalloc(&S.sort_tmp, tmp_bytes));
    S.sort_tmp_cap = tmp_bytes;
  }
  MODS_HIP_CHECK(rocprim::radix_sort_keys(S.sort_tmp, tmp_bytes, s_in, s_out, n_runs, 0, 32 + ubits, c->stream));
  MODS_HIP_CHECK(rocprim::radix_sort_keys(S.sort_tmp, tmp_bytes, e_in, e_out, n_runs, 0, 32 + ubits, c->stream));

  // 5. moments, frames, keypoints
  if ((rc = grow_dev(S.ell, S.ell_cap, (size_t)n_u * 6))) return rc;
  hipLaunchKernelGGL(mser_ellipse_kernel, dim3((n_u + 63) / 64), dim3(64), 0, c->stream, T, n_u, w, s_out, e_out, n_runs, S.ell, S.counters);
  const int *ot = S.tab + o_out;
  hipLaunchKernelGGL(mser_export_kernel, dim3((n_keys + 255) / 256), dim3(256), 0, c->stream, n_keys, ot, ot + n_keys, ot + 2 * (size_t)n_keys,
                     ot + 3 * (size_t)n_keys, ot + 4 * (size_t)n_keys, T.u_slot, T.u_thresh, w, S.ell, c->max_cand, c->keys_dev);
  MODS_HIP_CHECK(hipGetLastError());
  MODS_HIP_CHECK(hipMemcpyAsync(S.h_counters, S.counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  if (S.h_counters[2]) { set_error("MSER: %llu regions whose runs disagree with the growth's area", S.h_counters[2]); return MODS_E_HIP; }
  return MODS_OK;
}

// DetectAffineRegions for whichever detector the parameter set names (imagerepresentation.cpp:733-783); tilt, zoom: the view's
// SynthImage fields (1, 1 for the image itself), which scale regionsNumber (scale-space-detector.cpp:20-21, extrema.cpp:201-202)
int detect_any(mods_ctx *c, const float *img_dev, int n_img, int w, int h, int stride, const mods_hessaff_params *par, double tilt, double zoom) {
  if (par->detectorType == MODS_DET_MSER) return mser_detect(c, img_dev, n_img, w, h, stride, par, tilt, zoom);
  int rc;
  if ((rc = pyramid_configure(c, w, h, n_img, par))) return rc;
  if (tilt > 2.0 || zoom < 0.5) c->reg_number_eff = (int)floor(zoom * (double)par->regionsNumber / tilt);
  if ((rc = pyramid_build(c, img_dev, stride))) return rc;
  return detect_run(c);
}

}  // namespace mods

// self-test hook of the host half (no device needed): the stable thresholds of one polarity of an 8-bit image, rows of
// (seed_x, seed_y, thresh, margin, area) in output order; tree_out (optional, 3 ints per pixel of the padded (w + 2) x (h + 2)
// frame): pix_slot, tpar & 0x7fffffff (0x7fffffff = none), tlev
extern "C" int mods_test_mser_grow(const unsigned char *img8, int w, int h, int min_size, double max_area, double min_margin, int invert,
                                   int *out5, int max_out, int *tree_out) {
  if (!img8 || w <= 0 || h <= 0 || !out5) return MODS_E_ARG;
  const int cols = w + 2;
  const size_t P = (size_t)cols * (h + 2);
  std::vector<uint8_t> padded(P, 0);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) padded[(size_t)(y + 1) * cols + x + 1] = invert ? (uint8_t)(255 - img8[(size_t)y * w + x]) : img8[(size_t)y * w + x];
  std::vector<int32_t> ps(P, -1);
  std::vector<uint32_t> tp(P, mods::mser::kNoParent);
  std::vector<uint8_t> tl(P, 0);
  std::vector<mods::mser::Stable> st;
  mods::mser::Grower g;
  mods::mser::GrowParams gp;
  gp.min_size = min_size; gp.max_area = max_area; gp.min_margin = min_margin; gp.relative = false; gp.invert = invert != 0;
  g.run(padded.data(), w, h, gp, ps.data(), tp.data(), tl.data(), st);
  for (size_t i = 0; i < st.size() && (int)i < max_out; i++) {
    int *o = out5 + 5 * i;
    o[0] = st[i].slot % cols - 1; o[1] = st[i].slot / cols - 1; o[2] = st[i].thresh; o[3] = st[i].margin; o[4] = st[i].area;
  }
  if (tree_out)
    for (size_t i = 0; i < P; i++) { tree_out[3 * i] = ps[i]; tree_out[3 * i + 1] = (int)(tp[i] & mods::mser::kNoParent); tree_out[3 * i + 2] = tl[i]; }
  return (int)st.size();
}

