// Multi-view image representation and the MODS step loop for one image pair.
//
// Reference behaviour:
//   SetVSPars                        synth-detection.cpp:191-322   view list of one step, minus the views of earlier steps
//   ImageRepresentation::AddRegions / AddRegionsToList   imagerepresentation.cpp:637-684   regions of all views, appended
//                                    in view order, ids shifted by the list size
//   ImageRepresentation::SynthDetectDescribeKeypoints    imagerepresentation.cpp:686-1104  per-view chain (synth_view.hip)
//   CorrespondenceBank::MatchImgReps (separate detector / descriptor branch)   correspondencebank.cpp:288-340
//                                    all accumulated queries against all accumulated trains, every step
//   main step loop                   mods.cpp:202-383   stop when the verified matches reach minMatches
// One detector (HessianAffine) and one descriptor (RootSIFT): the part of iters_MODS.ini inside the hot
// path (steps [HessianAffine2], [HessianAffine3]).
//
// The accumulated regions live in HBM (208 B each; 31 views of a 1080p image are ~3*10^5 regions = 60 MB),
// so the matcher reads them in place and, for the multi-GPU path, the all-gather moves one dense buffer.
#include "common.hpp"
#include <atomic>
#include <string>
#include <thread>
#include <algorithm>
#include <chrono>
#include <cmath>

struct mods_imgrep {
  int device = 0;
  hipStream_t stream = nullptr;     // the owning context's stream
  mods_region *reg = nullptr;
  int cap = 0, n = 0;
};

namespace mods {

// AddRegionsToList: id and parent_id are shifted by the size of the list they are appended to
__global__ void __launch_bounds__(256) shift_ids_kernel(mods_region *reg, int n, int shift) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { reg[i].id += shift; reg[i].parent += shift; }
}

static double now_ms2() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace mods

using namespace mods;

extern "C" {

int mods_loransac_h(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask, double *H_out,
                    int *n_inliers, int *stats3);
int mods_loransac_f(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask, double *F_out,
                    int *n_inliers, int *stats3);
int mods_duplicate_filter(mods_tentative *tent, double *u6, double *laf, int n, double r, int mode, int *n_out);
int mods_ransac_set_device(int device);
int mods_unoriented_count(mods_ctx *c, int img);
int mods_match_fetch_internal(mods_ctx *c, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);   // capi.hip

int mods_host_alloc(size_t bytes, void **out) { if (!out) return MODS_E_ARG; MODS_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 4, hipHostMallocDefault)); return MODS_OK; }
int mods_host_free(void *p) { MODS_HIP_CHECK(hipHostFree(p)); return MODS_OK; }
int mods_dev_alloc(size_t bytes, void **out) { if (!out) return MODS_E_ARG; MODS_HIP_CHECK(hipMalloc(out, bytes ? bytes : 4)); return MODS_OK; }
int mods_dev_free(void *p) { MODS_HIP_CHECK(hipFree(p)); return MODS_OK; }
// (the copy runs on a stream of the device the buffer lives on; ordering contract: include/mods_hip.h)
int mods_dev_upload(void *dst_dev, const void *src_host, size_t bytes) { MODS_HIP_CHECK(mods::copy_wait(mods::thread_stream(mods::device_of_pointer(dst_dev)), dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return MODS_OK; }
int mods_dev_download(void *dst_host, const void *src_dev, size_t bytes) { MODS_HIP_CHECK(mods::copy_wait(mods::thread_stream(mods::device_of_pointer(src_dev)), dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return MODS_OK; }

// ---- view schedule -------------------------------------------------------------------------------------
// SetVSPars for one detector: the (zoom, tilt, phi) triples of a step that no earlier step has produced.
// `prev` (capacity prev_cap, *n_prev entries) is the history; the new views are appended to it.
// Returns the number of views written to `out`, or a negative error.
int mods_view_schedule(const double *scale_set, int n_scales, const double *tilt_set, int n_tilts, double phi_base,
                       mods_view_par *prev, int *n_prev, int prev_cap, mods_view_par *out, int max_out) {
  if (!n_prev || (*n_prev > 0 && !prev) || !out) { set_error("view_schedule: null argument"); return MODS_E_ARG; }
  const double eps1 = 0.01;   // synth-detection.cpp:22
  std::vector<mods_view_par> tmp;
  if (n_scales == 0 || n_tilts == 0) { mods_view_par v = {0, 0, 0}; tmp.push_back(v); }
  for (int sc = 0; sc < n_scales; sc++)
    for (int t = 0; t < n_tilts; t++) {
      if (std::fabs(tilt_set[t] - 1) > eps1) {
        int n_rot1 = (int)std::floor(180.0 * tilt_set[t] / phi_base);
        double delta_phi = M_PI / n_rot1;
        if (n_rot1 < 0) {            // negative density: no rotations, one vertically tilted view as well
          n_rot1 = 1;
          delta_phi = 0;
          mods_view_par v = {scale_set[sc], -tilt_set[t], 0.0};
          tmp.push_back(v);
        }
        for (int r = 0; r < n_rot1; r++) {
          mods_view_par v = {scale_set[sc], tilt_set[t], delta_phi * r};
          tmp.push_back(v);
        }
      } else {
        mods_view_par v = {scale_set[sc], tilt_set[t], 0.0};
        tmp.push_back(v);
      }
    }
  int n_out = 0;
  const int n_hist = *n_prev;
  for (size_t i = 0; i < tmp.size(); i++) {
    bool unique = true;
    for (int j = 0; j < n_hist; j++)
      if ((std::fabs(tmp[i].zoom - prev[j].zoom) <= eps1) && (std::fabs(tmp[i].tilt - prev[j].tilt) <= eps1) &&
          (std::fabs(tmp[i].phi - prev[j].phi) <= eps1)) { unique = false; break; }
    if (!unique) continue;
    if (n_out >= max_out) { set_error("view_schedule: more than %d views", max_out); return MODS_E_CAPACITY; }
    out[n_out++] = tmp[i];
  }
  if (prev) {
    if (*n_prev + n_out > prev_cap) { set_error("view_schedule: history overflow"); return MODS_E_CAPACITY; }
    for (int i = 0; i < n_out; i++) prev[(*n_prev)++] = out[i];
  }
  return n_out;
}

// ---- accumulated regions of one image -----------------------------------------------------------------------
int mods_imgrep_create(mods_ctx *c, int capacity, mods_imgrep **out) {
  if (!c || !out || capacity <= 0) { set_error("imgrep_create: bad arguments"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  mods_imgrep *r = new mods_imgrep();
  r->device = c->device; r->stream = c->stream; r->cap = capacity;
  MODS_HIP_CHECK(hipMalloc(&r->reg, sizeof(mods_region) * (size_t)capacity));
  *out = r;
  return MODS_OK;
}
void mods_imgrep_destroy(mods_imgrep *r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  (void)mods::stream_wait(r->stream);
  (void)hipFree(r->reg);
  delete r;
}
int mods_imgrep_clear(mods_imgrep *r) { if (!r) return MODS_E_ARG; r->n = 0; return MODS_OK; }
int mods_imgrep_count(const mods_imgrep *r) { return r ? r->n : 0; }
const mods_region *mods_imgrep_regions_dev(const mods_imgrep *r) { return r ? r->reg : nullptr; }

// AddRegions: the regions the context holds for image slot `img` (after mods_detect_describe[_view]_dev)
static int imgrep_append_from(mods_imgrep *r, mods_ctx *c, int img, const mods_region *base) {
  if (!r || !c || img < 0 || img >= (int)c->last_region_counts.size()) { set_error("imgrep_append: nothing described in that slot"); return MODS_E_ARG; }
  const int n = c->last_region_counts[img];
  if (r->n + n > r->cap) { set_error("imgrep: capacity %d exceeded (%d + %d)", r->cap, r->n, n); return MODS_E_CAPACITY; }
  if (n == 0) return MODS_OK;
  MODS_HIP_CHECK(hipSetDevice(r->device));
  MODS_HIP_CHECK(hipMemcpyAsync(r->reg + r->n, base + (size_t)img * c->max_cand, sizeof(mods_region) * (size_t)n,
                                hipMemcpyDeviceToDevice, c->stream));
  if (r->n > 0) hipLaunchKernelGGL(shift_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, r->reg + r->n, n, r->n);
  MODS_HIP_CHECK(hipGetLastError());
  r->n += n;
  return MODS_OK;
}
int mods_imgrep_append_ctx(mods_imgrep *r, mods_ctx *c, int img) { return imgrep_append_from(r, c, img, c ? c->regions_dev : nullptr); }
int mods_imgrep_append_ctx_half(mods_imgrep *r, mods_ctx *c, int img) {
  if (!c || !c->have_half || !c->regions_half_dev) { set_error("imgrep_append: no HalfRootSIFT descriptors in the context"); return MODS_E_ARG; }
  return imgrep_append_from(r, c, img, c->regions_half_dev);
}
int mods_imgrep_append_dev(mods_imgrep *r, const mods_region *src_dev, int n) {
  if (!r || (n > 0 && !src_dev)) { set_error("imgrep_append_dev: null argument"); return MODS_E_ARG; }
  if (r->n + n > r->cap) { set_error("imgrep: capacity %d exceeded (%d + %d)", r->cap, r->n, n); return MODS_E_CAPACITY; }
  if (n == 0) return MODS_OK;
  MODS_HIP_CHECK(hipSetDevice(r->device));
  MODS_HIP_CHECK(hipMemcpyAsync(r->reg + r->n, src_dev, sizeof(mods_region) * (size_t)n, hipMemcpyDeviceToDevice, r->stream));
  if (r->n > 0) hipLaunchKernelGGL(shift_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, r->stream, r->reg + r->n, n, r->n);
  MODS_HIP_CHECK(hipGetLastError());
  r->n += n;
  return MODS_OK;
}
int mods_imgrep_append_host(mods_imgrep *r, const mods_region *src, int n) {
  if (!r || (n > 0 && !src)) { set_error("imgrep_append_host: null argument"); return MODS_E_ARG; }
  if (r->n + n > r->cap) { set_error("imgrep: capacity %d exceeded (%d + %d)", r->cap, r->n, n); return MODS_E_CAPACITY; }
  if (n == 0) return MODS_OK;
  MODS_HIP_CHECK(hipSetDevice(r->device));
  MODS_HIP_CHECK(hipMemcpyAsync(r->reg + r->n, src, sizeof(mods_region) * (size_t)n, hipMemcpyHostToDevice, r->stream));
  MODS_HIP_CHECK(mods::stream_wait(r->stream));
  if (r->n > 0) hipLaunchKernelGGL(shift_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, r->stream, r->reg + r->n, n, r->n);
  MODS_HIP_CHECK(hipGetLastError());
  r->n += n;
  return MODS_OK;
}
int mods_imgrep_fetch(mods_imgrep *r, int begin, int count, mods_region *out) {
  if (!r || begin < 0 || count < 0 || begin + count > r->n || (count > 0 && !out)) { set_error("imgrep_fetch: bad range"); return MODS_E_ARG; }
  if (count == 0) return MODS_OK;
  MODS_HIP_CHECK(hipSetDevice(r->device));
  MODS_HIP_CHECK(hipMemcpyAsync(out, r->reg + begin, sizeof(mods_region) * (size_t)count, hipMemcpyDeviceToHost, r->stream));
  MODS_HIP_CHECK(mods::stream_wait(r->stream));
  return MODS_OK;
}

// MatchImgReps for (HessianAffine, RootSIFT): queries [q_begin, q_end) of rep `q` against every region of
// rep `t` (MatchFlannFGINN, exact search).  Tentative indices refer to the full lists.
int mods_match_reps(mods_ctx *c, const mods_imgrep *q, int q_begin, int q_end, const mods_imgrep *t, double ratio, double contradDist,
                    int nn, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out) {
  if (!c || !q || !t || !n_out) { set_error("match_reps: null argument"); return MODS_E_ARG; }
  if (q_begin < 0 || q_end > q->n || q_begin > q_end) { set_error("match_reps: bad query range"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = match_run(c, q->reg + q_begin, q_end - q_begin, t->reg, t->n, ratio, contradDist, nn);
  if (rc) return rc;
  rc = mods_match_fetch_internal(c, out, u6_out, laf_out, max_out, n_out);
  if (rc) return rc;
  if (out && q_begin)
    for (int i = 0; i < *n_out; i++) out[i].q += q_begin;
  return MODS_OK;
}

// The same for either matcher: distance > 0 runs MatchFLANNDistance (Hamming, threshold `distance`) on the query slice instead
int mods_match_reps_any(mods_ctx *c, const mods_imgrep *q, int q_begin, int q_end, const mods_imgrep *t, double ratio, double contradDist,
                        int nn, double distance, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out) {
  if (!(distance > 0)) return mods_match_reps(c, q, q_begin, q_end, t, ratio, contradDist, nn, out, u6_out, laf_out, max_out, n_out);
  if (!c || !q || !t || !n_out) { set_error("match_reps: null argument"); return MODS_E_ARG; }
  if (q_begin < 0 || q_end > q->n || q_begin > q_end) { set_error("match_reps: bad query range"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  int rc = match_run_distance(c, q->reg + q_begin, q_end - q_begin, t->reg, t->n, distance);
  if (rc) return rc;
  rc = mods_match_fetch_internal(c, out, u6_out, laf_out, max_out, n_out);
  if (rc) return rc;
  if (out && q_begin)
    for (int i = 0; i < *n_out; i++) out[i].q += q_begin;
  return MODS_OK;
}

// ---- the step loop of mods.cpp:202-383 on one GPU ---------------------------------------------------------------
// img1_dev / img2_dev: dense fp32 images in HBM (the two images may differ in size).  Every step adds the step's new views of both images to the two region
// banks, matches bank 1 against bank 2, filters duplicates, verifies, and stops once the verified
// matches reach min_matches.
static void copy_verified(mods_ctx *c, const mods_ladder_result *res, double *matches_out, int max_matches) {
  if (!matches_out) return;
  for (int m = 0; m < res->n_inliers && m < max_matches; m++) {   // mods_verify_tentatives left the verified rows first
    const double *p = &c->h_u6[(size_t)m * 6];
    matches_out[4 * m] = p[0]; matches_out[4 * m + 1] = p[1]; matches_out[4 * m + 2] = p[3]; matches_out[4 * m + 3] = p[4];
  }
}

// One FGINN search of bank q against bank t into a host list (MatchFlannFGINN of one (detector, descriptor) pair,
// correspondencebank.cpp:288-340)
struct TentList { std::vector<mods_tentative> t; std::vector<double> u6, laf; void clear() { t.clear(); u6.clear(); laf.clear(); } };
// distance > 0: MatchFLANNDistance (Hamming, threshold `distance`) instead of the FGINN search
static int match_into(mods_ctx *c, mods_imgrep *q, mods_imgrep *t, double ratio, const mods_pair_params *par, TentList *out, double distance = 0) {
  int rc, m = 0;
  out->clear();
  if (!q || !t) return MODS_OK;
  if (distance > 0) rc = match_run_distance(c, q->reg, q->n, t->reg, t->n, distance);
  else rc = match_run(c, q->reg, q->n, t->reg, t->n, ratio, par->contradDist, par->nn);
  if (rc) return rc;
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  m = *(volatile int *)c->m_count;
  if (m > c->max_cand) { set_error("tentative list overflow"); return MODS_E_CAPACITY; }
  out->t.resize(m); out->u6.resize((size_t)m * 6); out->laf.resize((size_t)m * 14);
  return mods_match_copy_out(c, m, out->t.data(), out->u6.data(), out->laf.data());
}

// CorrespondenceBank::GetCorresponcesVector("All", "All") (correspondencebank.cpp:114-148): the bank is a std::map keyed by
// descriptor name, then by detector name, and the joint list walks it in key order - "HalfRootSIFT" before "RootSIFT",
// detectors in the (name-sorted) order the caller listed them.  lists[desc][det], desc 0 = RootSIFT, 1 = HalfRootSIFT.
static void gather_tentatives(mods_ctx *c, const std::vector<TentList> lists[2]) {
  c->h_tent.clear(); c->h_u6.clear(); c->h_laf.clear();
  for (int desc = 1; desc >= 0; desc--)
    for (const TentList &l : lists[desc]) {
      c->h_tent.insert(c->h_tent.end(), l.t.begin(), l.t.end());
      c->h_u6.insert(c->h_u6.end(), l.u6.begin(), l.u6.end());
      c->h_laf.insert(c->h_laf.end(), l.laf.begin(), l.laf.end());
    }
}

// DuplicateFiltering + LORANSACFiltering of the gathered list (mods.cpp:278-383)
static int verify_gathered(mods_ctx *c, const mods_pair_params *par, mods_ladder_result *res) {
  const int n = (int)c->h_tent.size();
  res->n_tentatives = n;
  int stats[3] = {0, 0, 0};
  double ms_dup = 0, ms_ran = 0;
  int gt3[3] = {0, 0, 0};
  const int rc = mods_verify_tentatives_ex(c->device, par, c->h_tent.data(), c->h_u6.data(), c->h_laf.data(), n, &res->n_unique, &res->n_inliers,
                                           res->H, stats, gt3, &ms_dup, &ms_ran);
  if (rc) return rc;
  res->ms_duplicates += ms_dup; res->ms_ransac += ms_ran;
  res->ransac_samples = stats[0]; res->ransac_lo = stats[1]; res->ransac_rejects = stats[2];
  res->gt_true = gt3[0]; res->gt_ransac_inliers = gt3[1]; res->gt_true_of_ransac = gt3[2];
  return MODS_OK;
}
// curr_matches of the step loop (mods.cpp:286, 357-383): TrueMatch1st - the verified list, or in ground-truth mode
// HMatrixFiltering's count over all unique tentatives (the de-duplicated verified list when duplicates are filtered after the
// verification) - and Tentatives1stRANSAC when [Matching] RANSACforStopping is set in ground-truth mode
static int stop_count(const mods_pair_params *par, const mods_ladder_result *res) {
  if (!par->ransac.groundTruth) return res->n_inliers;
  if (par->ransac.ransacForStopping) return res->gt_ransac_inliers;
  return par->dup_before_ransac ? res->gt_true : res->n_inliers;
}

// match bank 1 against bank 2, drop duplicates, verify (MatchImgReps + DuplicateFiltering + LORANSACFiltering,
// mods.cpp:288-383) for one detector: fills the match / verification fields of res
static int match_verify_banks(mods_ctx *c, mods_imgrep *rep1, mods_imgrep *rep2, double fginn_ratio, const mods_pair_params *par,
                              mods_ladder_result *res, mods_imgrep *rep1h = nullptr, mods_imgrep *rep2h = nullptr, double fginn_ratio_half = 0) {
  int rc;
  const double t1 = now_ms2();
  std::vector<TentList> lists[2];
  lists[0].resize(1); lists[1].resize(1);
  if (fginn_ratio > 0 && (rc = match_into(c, rep1, rep2, fginn_ratio, par, &lists[0][0]))) return rc;
  if (fginn_ratio_half > 0 && (rc = match_into(c, rep1h, rep2h, fginn_ratio_half, par, &lists[1][0]))) return rc;
  gather_tentatives(c, lists);
  res->ms_match += now_ms2() - t1;
  return verify_gathered(c, par, res);
}

// One synthesised view of one image for one detector of a step, and where its regions were left
struct ViewJob {
  int d = 0, im = 0;
  mods_view_par vp;
  double initSigma = 0;
  int doBlur = 1;
  mods_describe_params desc;
  bool want_half = false;
  int nd = 0, nr = 0, unoriented = 0, rc = 0;
  const mods_region *src = nullptr, *src_half = nullptr;
  std::string err;
};

// MODS_LADDER_WORKERS contexts (default 4, 1 = the serial loop) on the GPU of `c`; worker 0 is `c` itself on the calling thread.
// Every worker copies the regions of a finished view into its own staging arena, so the banks can be filled in job order.
// Helper contexts cost device memory (each is a full max_w x max_h context with two image slots): one is made only while the
// device has room for it several times over, a failed attempt is reported once on stderr and not repeated, and what the helpers'
// stage timers measured is added to the caller's.
static int run_view_jobs(mods_ctx *c, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                         const mods_hessaff_params *dets, std::vector<ViewJob> &jobs) {
  static const int env_workers = getenv("MODS_LADDER_WORKERS") ? atoi(getenv("MODS_LADDER_WORKERS")) : 4;
  int n_workers = std::max(1, std::min(env_workers, 8));
  if (c->ext_fn || c->shape_fn || c->ori_fn) n_workers = 1;      // the daemons' hooks belong to one context
  if (jobs.empty()) return MODS_OK;
  // Work units: a view of image 1 and the same view of image 2 go through ONE chain of launches when the images have one size
  // (both images of a pair share their view schedule) and the contexts hold two images; otherwise one job per unit.
  struct Unit { int a, b; };                                     // job indices; b < 0: a single view
  std::vector<Unit> units;
  const bool pair_ok = w1 == w2 && h1 == h2 && !(c->ext_fn || c->shape_fn || c->ori_fn);
  {
    std::vector<char> used(jobs.size(), 0);
    for (size_t i = 0; i < jobs.size(); i++) {
      if (used[i]) continue;
      int mate = -1;
      if (pair_ok && jobs[i].im == 0)
        for (size_t k = i + 1; k < jobs.size() && mate < 0; k++)
          if (!used[k] && jobs[k].im == 1 && jobs[k].d == jobs[i].d && jobs[k].vp.zoom == jobs[i].vp.zoom && jobs[k].vp.tilt == jobs[i].vp.tilt &&
              jobs[k].vp.phi == jobs[i].vp.phi && jobs[k].initSigma == jobs[i].initSigma && jobs[k].doBlur == jobs[i].doBlur)
            mate = (int)k;
      used[i] = 1;
      if (mate >= 0) used[mate] = 1;
      units.push_back({(int)i, mate});
    }
  }
  n_workers = std::min<int>(n_workers, (int)units.size());
  // helper contexts are made by their own threads, side by side and only while enough units are left to be worth the ~35 ms a
  // context takes to set up (a one-shot run with a handful of views is faster on the caller's context alone)
  if ((int)c->helpers.size() < n_workers - 1) c->helpers.resize(n_workers - 1, nullptr);
  if ((int)c->helper_stage.size() < n_workers) c->helper_stage.resize(n_workers);
  // the banks were filled from the staging arenas by copies on c->stream (mods_imgrep_append_dev, asynchronous); the workers are
  // about to overwrite the arenas on their own streams
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  struct Placed { size_t off, off_half; };
  std::vector<Placed> placed(jobs.size());
  std::vector<int> owner(jobs.size(), 0);
  std::atomic<int> next(0);
  auto work = [&](int k) {
    (void)hipSetDevice(c->device);
    if (k > 0 && !c->helpers[k - 1]) {
      if (c->helpers_failed || (int)units.size() - next.load() < 3 * (k + 1)) return;
      // a context of this size takes roughly 100 bytes per pixel and image slot (pyramid pools, patch store, region lists):
      // leave the device at least four such contexts of head room, other users of the GPU included
      size_t free_b = 0, total_b = 0;
      const size_t ctx_bytes = (size_t)c->max_w * c->max_h * 2 * 100;
      mods_ctx *h = nullptr;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 4 * ctx_bytes || mods_ctx_create_ex(c->device, c->max_w, c->max_h, 2, 1, &h)) {
        if (!c->helpers_failed.exchange(true))
          fprintf(stderr, "mods: view worker %d of the step loop not created (%.1f GB free on device %d): the remaining contexts take its views\n", k, free_b / 1e9, c->device);
        return;                                                                        // the other workers take the units
      }
      c->helpers[k - 1] = h;
    }
    mods_ctx *wk = k == 0 ? c : c->helpers[k - 1];
    if (k > 0) wk->timing_mask = c->timing_mask;
    mods_ctx::StageArena &A = c->helper_stage[k];
    size_t used = 0;
    auto park = [&](int ji, int slot) {                         // regions of context slot `slot` -> this worker's arena
      ViewJob &j = jobs[ji];
      const size_t need = used + (size_t)j.nr * (j.want_half ? 2 : 1);
      if (need > A.cap) {                                       // grow, keeping what earlier jobs of this step left
        const size_t cap = std::max<size_t>(need + need / 2, 1 << 14);
        mods_region *nb = nullptr;
        if (hipMalloc(&nb, cap * sizeof(mods_region)) != hipSuccess) { j.rc = MODS_E_HIP; set_error("view staging: out of device memory"); return; }
        (void)mods::stream_wait(wk->stream);                 // copies into the old arena may still be in flight
        if (used) (void)mods::copy_wait(wk->stream, nb, A.buf, used * sizeof(mods_region), hipMemcpyDeviceToDevice);
        (void)hipFree(A.buf);
        A.buf = nb; A.cap = cap;
      }
      // the copies are ordered on the worker's stream before the next view overwrites the context's region lists: no
      // synchronisation per view, one per worker at the end
      placed[ji].off = used;
      if (j.nr > 0 && hipMemcpyAsync(A.buf + used, wk->regions_dev + (size_t)slot * wk->max_cand, sizeof(mods_region) * (size_t)j.nr, hipMemcpyDeviceToDevice,
                                     wk->stream) != hipSuccess) j.rc = MODS_E_HIP;
      used += j.nr;
      if (!j.rc && j.want_half && j.nr > 0) {
        placed[ji].off_half = used;
        if (!wk->have_half || !wk->regions_half_dev) { j.rc = MODS_E_ARG; set_error("view job: no HalfRootSIFT descriptors in the context"); }
        else if (hipMemcpyAsync(A.buf + used, wk->regions_half_dev + (size_t)slot * wk->max_cand, sizeof(mods_region) * (size_t)j.nr, hipMemcpyDeviceToDevice,
                                wk->stream) != hipSuccess) j.rc = MODS_E_HIP;
        used += j.nr;
      }
      if (j.rc == MODS_E_HIP) set_error("view job: device copy failed");
    };
    for (int u; (u = next.fetch_add(1)) < (int)units.size();) {
      const Unit un = units[u];
      ViewJob &j = jobs[un.a];
      if (un.b >= 0 && wk->batch >= 2) {                        // both images of the pair through one chain of launches
        ViewJob &j2 = jobs[un.b];
        int nd[2] = {0, 0}, nr[2] = {0, 0};
        j.rc = mods_detect_describe_view2_dev(wk, img1_dev, img2_dev, w1, h1, w1, j.vp.tilt, j.vp.phi, j.vp.zoom, j.initSigma, j.doBlur, &dets[j.d],
                                              &j.desc, nullptr, nd, nr);
        j2.rc = j.rc;
        if (!j.rc) {
          j.nd = nd[0]; j.nr = nr[0]; j.unoriented = mods_unoriented_count(wk, 0);
          j2.nd = nd[1]; j2.nr = nr[1]; j2.unoriented = mods_unoriented_count(wk, 1);
          park(un.a, 0);
          if (!j.rc) park(un.b, 1);
        }
        if (j.rc) j.err = mods_last_error();
        if (j2.rc) j2.err = mods_last_error();
        owner[un.a] = owner[un.b] = k;
        continue;
      }
      for (int ji : {un.a, un.b}) {
        if (ji < 0) continue;
        ViewJob &q = jobs[ji];
        const float *img = q.im ? img2_dev : img1_dev;
        const int w = q.im ? w2 : w1, h = q.im ? h2 : h1;
        q.rc = mods_detect_describe_view_dev(wk, img, w, h, w, q.vp.tilt, q.vp.phi, q.vp.zoom, q.initSigma, q.doBlur, &dets[q.d], &q.desc, nullptr,
                                             &q.nd, &q.nr);
        if (!q.rc) { q.unoriented = mods_unoriented_count(wk, 0); park(ji, 0); }
        if (q.rc) q.err = mods_last_error();
        owner[ji] = k;
      }
    }
    (void)mods::stream_wait(wk->stream);
  };
  std::vector<std::thread> pool;
  try {
    for (int k = 1; k < n_workers; k++) pool.emplace_back(work, k);
  } catch (...) {}                       // fewer threads than asked for: the others (at least the caller) take the jobs
  work(0);
  for (auto &t : pool) t.join();
  if (c->timing_mask)                                            // what the helpers' stage timers saw belongs to this call
    for (mods_ctx *h : c->helpers) {
      if (!h) continue;
      for (int st = 0; st < MODS_STAGE_COUNT; st++) {
        double ms = 0, by = 0; int n = 0;
        if (mods_ctx_timing_read(h, st, &ms, &n, &by) == MODS_OK) { c->timers[st].total_ms += ms; c->timers[st].launches += n; c->timers[st].bytes += by; }
      }
      (void)mods_ctx_timing_reset(h);
    }
  for (size_t i = 0; i < jobs.size(); i++) {
    ViewJob &j = jobs[i];
    if (j.rc) { set_error("%s", j.err.c_str()); return j.rc; }
    j.src = c->helper_stage[owner[i]].buf + placed[i].off;           // arenas may have moved while growing: resolved here
    j.src_half = j.want_half ? c->helper_stage[owner[i]].buf + placed[i].off_half : nullptr;
  }
  return MODS_OK;
}

int mods_match_ladder_groups_dev(mods_ctx *c, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                                 const mods_ladder_step *steps, const mods_hessaff_params *dets, const mods_ladder_group *groups, int group_pos,
                                 int n_steps, int n_det, int min_matches, const mods_pair_params *par, mods_imgrep **reps1, mods_imgrep **reps2,
                                 mods_ladder_result *res, double *matches_out, int max_matches) {
  if (groups && (group_pos < 0 || group_pos > n_det)) { set_error("match_ladder: bad group position"); return MODS_E_ARG; }
  if (!c || !img1_dev || !img2_dev || !steps || !dets || !par || !reps1 || !reps2 || !res || n_det < 1 || n_det > 8) { set_error("match_ladder: bad argument"); return MODS_E_ARG; }
  for (int d = 0; d < n_det; d++) if (!reps1[d] || !reps2[d]) { set_error("match_ladder: null region bank"); return MODS_E_ARG; }
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  int rc, curr_matches = 0;
  struct PerDet {
    std::vector<mods_view_par> hist = std::vector<mods_view_par>(1024);
    int n_hist = 0;
    mods_imgrep *h1 = nullptr, *h2 = nullptr;    // HalfRootSIFT banks (the reference keeps one region list per descriptor name)
    ~PerDet() { mods_imgrep_destroy(h1); mods_imgrep_destroy(h2); }
  };
  std::vector<PerDet> pd(n_det);
  std::vector<mods_view_par> views(256);
  // per (descriptor, detector): kept from step to step until re-matched.  With grouped matching the bank has one more
  // "detector", named Group, at its place in the name order: slot of detector d = d + (d >= group_pos)
  std::vector<TentList> lists[2];
  const int n_slots = n_det + (groups ? 1 : 0);
  lists[0].resize(n_slots); lists[1].resize(n_slots);
  auto slot_of = [&](int d) { return groups && d >= group_pos ? d + 1 : d; };
  struct GroupBanks { mods_imgrep *q = nullptr, *t = nullptr; ~GroupBanks() { mods_imgrep_destroy(q); mods_imgrep_destroy(t); } } gb;
  for (int d = 0; d < n_det; d++) { mods_imgrep_clear(reps1[d]); mods_imgrep_clear(reps2[d]); }
  for (int step = 0; step < n_steps && curr_matches < min_matches; step++) {
    std::vector<int> new_views(n_det, 0);
    std::vector<ViewJob> jobs;
    const double t0 = now_ms2();
    for (int d = 0; d < n_det; d++) {
      const mods_ladder_step &st = steps[(size_t)step * n_det + d];
      if (st.n_tilts < 0 || st.n_scales < 0) continue;          // the detector has no section in this step
      const int nv = mods_view_schedule(st.scale_set, st.n_scales, st.tilt_set, st.n_tilts, st.phi, pd[d].hist.data(), &pd[d].n_hist,
                                        (int)pd[d].hist.size(), views.data(), (int)views.size());
      if (nv < 0) return nv;
      new_views[d] = nv;
      const bool want_half = st.fginn_ratio_half > 0 || st.dist_threshold_half > 0;
      if (want_half && !pd[d].h1) {
        if ((rc = mods_imgrep_create(c, reps1[d]->cap, &pd[d].h1))) return rc;
        if ((rc = mods_imgrep_create(c, reps2[d]->cap, &pd[d].h2))) return rc;
      }
      mods_describe_params desc = par->desc;
      desc.ori_halfMode = (st.half_orientation || want_half) ? 1 : 0;
      desc.halfDesc = want_half ? 1 : 0;
      for (int im = 0; im < 2; im++)
        for (int v = 0; v < nv; v++) {
          ViewJob j;
          j.d = d; j.im = im; j.vp = views[v]; j.initSigma = st.initSigma; j.doBlur = st.doBlur; j.desc = desc; j.want_half = want_half;
          jobs.push_back(j);
        }
    }
    // the views of a step are independent (the reference runs them as an OpenMP loop, imagerepresentation.cpp:703-704): they are
    // spread over a few contexts of this GPU - the launch chains of small tilted views and the host growth of MSER views
    // overlap - and appended to the banks in the serial order afterwards
    if ((rc = run_view_jobs(c, img1_dev, w1, h1, img2_dev, w2, h2, dets, jobs))) return rc;
    for (const ViewJob &j : jobs) {
      mods_imgrep *rep = j.im ? reps2[j.d] : reps1[j.d];
      if (j.nr > 0 && (rc = mods_imgrep_append_dev(rep, j.src, j.nr))) return rc;
      if (j.want_half && j.nr > 0 && (rc = mods_imgrep_append_dev(j.im ? pd[j.d].h2 : pd[j.d].h1, j.src_half, j.nr))) return rc;
      res->n_views++;
      res->n_detected[j.im] += j.nd;
      res->n_unoriented[j.im] += j.unoriented;
    }
    res->n_described[0] = res->n_described[1] = 0;
    for (int d = 0; d < n_det; d++) { res->n_described[0] += reps1[d]->n; res->n_described[1] += reps2[d]->n; }
    const double t1 = now_ms2();
    res->ms_detect_describe += t1 - t0;
    // MatchImgReps, correspondencebank.cpp:286-340: a detector is matched in a step that brought new views of it; each of its
    // descriptor lists is cleared and searched again over everything accumulated.  Lists that are not touched keep their
    // tentatives (ratio < 0: descriptor / detector not named in [Matching<i>]; 0: named, not searched, so it ends up empty)
    // Grouped matching first, correspondencebank.cpp:245-285: the regions of the GroupDetectors, joined in the order they
    // are named, searched as one list per GroupDescriptor with the [Matching]-wide thresholds; done in every step that names
    // a group, new views or not
    if (groups && groups[step].n_dets > 0) {
      const mods_ladder_group &g = groups[step];
      for (int desc = 0; desc < 2; desc++) {
        const double ratio = desc ? g.fginn_ratio_half : g.fginn_ratio, dist = desc ? g.dist_threshold_half : g.dist_threshold;
        if (ratio < 0) continue;
        TentList &out = lists[desc][group_pos];
        out.clear();
        if (!(ratio > 0) && !(dist > 0)) continue;
        int nq = 0, nt = 0;
        for (int i = 0; i < g.n_dets; i++) {
          const int d = g.dets[i];
          if (d < 0 || d >= n_det) { set_error("match_ladder: group names detector %d of %d", d, n_det); return MODS_E_ARG; }
          const mods_imgrep *a = desc ? pd[d].h1 : reps1[d], *b = desc ? pd[d].h2 : reps2[d];
          if (a) nq += a->n;
          if (b) nt += b->n;
        }
        if (nq == 0 || nt == 0) continue;
        if (!gb.q || gb.q->cap < nq) { mods_imgrep_destroy(gb.q); gb.q = nullptr; if ((rc = mods_imgrep_create(c, nq, &gb.q))) return rc; }
        if (!gb.t || gb.t->cap < nt) { mods_imgrep_destroy(gb.t); gb.t = nullptr; if ((rc = mods_imgrep_create(c, nt, &gb.t))) return rc; }
        mods_imgrep_clear(gb.q); mods_imgrep_clear(gb.t);
        for (int i = 0; i < g.n_dets; i++) {
          const int d = g.dets[i];
          const mods_imgrep *a = desc ? pd[d].h1 : reps1[d], *b = desc ? pd[d].h2 : reps2[d];
          if (a && a->n && (rc = mods_imgrep_append_dev(gb.q, a->reg, a->n))) return rc;
          if (b && b->n && (rc = mods_imgrep_append_dev(gb.t, b->reg, b->n))) return rc;
        }
        if (ratio > 0 && (rc = match_into(c, gb.q, gb.t, ratio, par, &out))) return rc;
        if (dist > 0 && (rc = match_into(c, gb.q, gb.t, 0, par, &out, dist))) return rc;
      }
    }
    for (int d = 0; d < n_det; d++) {
      const mods_ladder_step &st = steps[(size_t)step * n_det + d];
      if (st.n_tilts < 0 || st.n_scales < 0 || new_views[d] == 0) continue;
      // (a descriptor with a DistanceThreshold: MatchFLANNDistance runs after MatchFlannFGINN and clears the list it is
      // given, so its tentatives replace the FGINN ones, correspondencebank.cpp:328-334, matching.cpp:585)
      if (st.fginn_ratio >= 0) {
        lists[0][slot_of(d)].clear();
        if (st.dist_threshold > 0) { if ((rc = match_into(c, reps1[d], reps2[d], 0, par, &lists[0][slot_of(d)], st.dist_threshold))) return rc; }
        else if (st.fginn_ratio > 0 && (rc = match_into(c, reps1[d], reps2[d], st.fginn_ratio, par, &lists[0][slot_of(d)]))) return rc;
      }
      if (st.fginn_ratio_half >= 0) {
        lists[1][slot_of(d)].clear();
        if (st.dist_threshold_half > 0 && pd[d].h1) { if ((rc = match_into(c, pd[d].h1, pd[d].h2, 0, par, &lists[1][slot_of(d)], st.dist_threshold_half))) return rc; }
        else if (st.fginn_ratio_half > 0 && (rc = match_into(c, pd[d].h1, pd[d].h2, st.fginn_ratio_half, par, &lists[1][slot_of(d)]))) return rc;
      }
    }
    gather_tentatives(c, lists);
    res->ms_match += now_ms2() - t1;
    if ((rc = verify_gathered(c, par, res))) return rc;
    curr_matches = stop_count(par, res);
    res->steps_done = step + 1;
  }
  copy_verified(c, res, matches_out, max_matches);
  return MODS_OK;
}

int mods_match_ladder_dets_dev(mods_ctx *c, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                               const mods_ladder_step *steps, const mods_hessaff_params *dets, int n_steps, int n_det, int min_matches,
                               const mods_pair_params *par, mods_imgrep **reps1, mods_imgrep **reps2, mods_ladder_result *res,
                               double *matches_out, int max_matches) {
  return mods_match_ladder_groups_dev(c, img1_dev, w1, h1, img2_dev, w2, h2, steps, dets, nullptr, 0, n_steps, n_det, min_matches, par, reps1, reps2,
                                      res, matches_out, max_matches);
}

int mods_match_ladder_dev(mods_ctx *c, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                          const mods_ladder_step *steps, int n_steps, int min_matches,
                          const mods_pair_params *par, mods_imgrep *rep1, mods_imgrep *rep2, mods_ladder_result *res, double *matches_out,
                          int max_matches) {
  if (!par) { set_error("match_ladder: null argument"); return MODS_E_ARG; }
  return mods_match_ladder_dets_dev(c, img1_dev, w1, h1, img2_dev, w2, h2, steps, &par->det, n_steps, 1, min_matches, par, &rep1, &rep2, res,
                                    matches_out, max_matches);
}

int mods_match_verify_reps(mods_ctx *c, mods_imgrep *rep1, mods_imgrep *rep2, double fginn_ratio, const mods_pair_params *par,
                           mods_ladder_result *res, double *matches_out, int max_matches) {
  if (!c || !rep1 || !rep2 || !par || !res) { set_error("match_verify_reps: null argument"); return MODS_E_ARG; }
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  res->n_described[0] = rep1->n; res->n_described[1] = rep2->n;
  const int rc = match_verify_banks(c, rep1, rep2, fginn_ratio, par, res);
  if (rc) return rc;
  res->steps_done = 1;
  copy_verified(c, res, matches_out, max_matches);
  return MODS_OK;
}

}  // extern "C"
