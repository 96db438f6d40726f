// One hard pair on several GPUs of a node: the synthesised views of a step are sharded over the devices, the described
// regions are exchanged in ONE all-gather per step (RCCL over xGMI, HBM to HBM), every device then searches its slice of the
// query rows against all trains, and device 0's host verifies.
//
// Reference seams: the views loop of ImageRepresentation::SynthDetectDescribeKeypoints (imagerepresentation.cpp:704-1099: views
// are independent until AddRegions), CorrespondenceBank::MatchImgReps (correspondencebank.cpp:234-343: queries are searched one by
// one), the step loop of mods.cpp:202-383.  The result is identical to mods_match_ladder_dev on one GPU: the banks are rebuilt
// in canonical (image, view) order on every device and the tentatives are joined in query order.
//
// One host process drives all devices (one context and one host thread per device, one RCCL communicator per device created
// with ncclCommInitAll): that is what the `mods` command line needs (MODS_DEVICES=0,1,..).  bench.py's throughput scaling uses
// one process per GPU instead (pairs are independent there: no collective at all).
//
// Exchange record = mods_region (208 B): the matcher needs the 128 descriptor bytes and the centre, the emit kernel builds the
// RANSAC correspondences and the local affine frames of the LAF checks (x, y, a11..a22, s = 56 B) from the same lists on the
// device, so the frame travels too; only response / ids (24 B) are not strictly needed.  3*10^5 regions = 62 MB per step.
//
// Round 3: the whole step loop of mods_match_ladder_groups_dev - several detectors per step, each with its own view history
// and banks, HalfRootSIFT lists next to the RootSIFT ones (a job's HalfRootSIFT twins travel right behind its RootSIFT
// regions in the same all-gather), the distance matcher, grouped matching - so that iters_MODS.ini gives the same files on
// N GPUs as on one.
#include "common.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <string>
#include <thread>

extern "C" int mods_ctx_create_ex(int device, int max_w, int max_h, int batch, int flags, mods_ctx **out);
using mods::set_error;

struct mods_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<mods_ctx *> ctx;
  // [dev][det]: every device holds a full copy of every bank (RootSIFT r1 / r2, HalfRootSIFT h1 / h2), created on first use
  std::vector<std::vector<mods_imgrep *>> r1, r2, h1, h2;
  std::vector<mods_imgrep *> gq, gt;  // [dev] joint banks of grouped matching
  std::vector<ncclComm_t> comm;
  bool use_rccl = false;              // distinct devices: RCCL all-gather; a device listed twice (development on one GPU): copies
  std::vector<mods_region *> send;    // [dev] regions of the views a device described in this step, packed in job order
  std::vector<mods_region *> recv;    // [dev] n x cap: what every device contributed
  std::vector<float *> img;           // [dev] both images
  size_t cap = 0;                     // regions per device and step
  int rep_cap = 0;
  size_t img_px = 0;                  // pixels the image buffer of a device holds (both images together)
  int side = 0;                       // side of the contexts' square canvas (>= the diagonal of the images given at creation)
};

extern "C" int mods_match_reps_any(mods_ctx *c, const mods_imgrep *q, int q_begin, int q_end, const mods_imgrep *t, double ratio, double contradDist,
                                   int nn, double distance, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);
extern "C" int mods_regions_half_copy_dev(mods_ctx *c, int img, mods_region *dst_dev, int n);

namespace {

// Greedy longest-processing-time assignment of the view jobs to the devices (tilted views are much smaller than the frontal
// one); deterministic.  The same rule as shard.largest_first_views.
void assign_views(const std::vector<double> &areas, int n_dev, std::vector<std::vector<int>> *out) {
  std::vector<int> order(areas.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return areas[a] > areas[b]; });
  std::vector<double> load(n_dev, 0.0);
  out->assign(n_dev, std::vector<int>());
  for (int i : order) {
    int r = 0;
    for (int q = 1; q < n_dev; q++)
      if (load[q] < load[r]) r = q;
    (*out)[r].push_back(i);
    load[r] += areas[i];
  }
  for (auto &v : *out) std::sort(v.begin(), v.end());   // a device walks its jobs in job order
}

struct Job { int det, im; mods_view_par v; double init_sigma; int do_blur; int half_ori, want_half; };
struct TentList { std::vector<mods_tentative> t; std::vector<double> u6, laf; void clear() { t.clear(); u6.clear(); laf.clear(); } };

int bank(mods_multi *m, std::vector<std::vector<mods_imgrep *>> &v, int d, int det) {
  if ((int)v[d].size() <= det) v[d].resize(det + 1, nullptr);
  if (!v[d][det]) return mods_imgrep_create(m->ctx[d], m->rep_cap, &v[d][det]);
  return MODS_OK;
}

// One list of a (descriptor, detector) pair over all devices: device d searches its slice of the query rows of ITS copy of the
// banks against all trains; joined in query order (MatchFlannFGINN / MatchFLANNDistance walk the queries in order)
int match_sharded(mods_multi *m, const std::vector<mods_imgrep *> &q, const std::vector<mods_imgrep *> &t, double ratio, double distance,
                  const mods_pair_params *par, TentList *out) {
  out->clear();
  const int D = m->n;
  if (!q[0] || !t[0]) return MODS_OK;
  const int nq = mods_imgrep_count(q[0]), nt = mods_imgrep_count(t[0]);
  if (nq == 0 || nt == 0) return MODS_OK;
  std::vector<TentList> part(D);
  std::vector<int> rcs(D, MODS_OK);
  std::vector<std::string> errs(D);
  std::vector<std::thread> th;
  for (int d = 0; d < D; d++)
    th.emplace_back([&, d] {
      if (hipSetDevice(m->dev[d]) != hipSuccess) { rcs[d] = MODS_E_HIP; errs[d] = "hipSetDevice"; return; }
      const int q0 = (int)((long long)nq * d / D), q1 = (int)((long long)nq * (d + 1) / D);
      if (q1 <= q0) return;
      const int cap = q1 - q0;
      part[d].t.resize(cap); part[d].u6.resize((size_t)cap * 6); part[d].laf.resize((size_t)cap * 14);
      int n = 0;
      const int rc = mods_match_reps_any(m->ctx[d], q[d], q0, q1, t[d], ratio, par->contradDist, par->nn, distance, part[d].t.data(),
                                         part[d].u6.data(), part[d].laf.data(), cap, &n);
      if (rc) { rcs[d] = rc; errs[d] = mods_last_error(); return; }
      part[d].t.resize(n); part[d].u6.resize((size_t)n * 6); part[d].laf.resize((size_t)n * 14);
    });
  for (auto &x : th) x.join();
  for (int d = 0; d < D; d++)
    if (rcs[d]) { set_error("device %d: %s", m->dev[d], errs[d].c_str()); return rcs[d]; }
  for (int d = 0; d < D; d++) {
    out->t.insert(out->t.end(), part[d].t.begin(), part[d].t.end());
    out->u6.insert(out->u6.end(), part[d].u6.begin(), part[d].u6.end());
    out->laf.insert(out->laf.end(), part[d].laf.begin(), part[d].laf.end());
  }
  return MODS_OK;
}

}  // namespace

extern "C" {

// host-side piece for the CPU tests: the assignment of `n` view jobs with the given areas to n_dev devices; owner[i] = device
int mods_multi_assign(const double *areas, int n, int n_dev, int *owner) {
  if (!areas || !owner || n < 0 || n_dev < 1) return MODS_E_ARG;
  std::vector<std::vector<int>> a;
  assign_views(std::vector<double>(areas, areas + n), n_dev, &a);
  for (int d = 0; d < n_dev; d++)
    for (int i : a[d]) owner[i] = d;
  return MODS_OK;
}

void mods_multi_destroy(mods_multi *m) {
  if (!m) return;
  for (int d = 0; d < (int)m->ctx.size(); d++) {
    (void)hipSetDevice(m->dev[d]);
    if (d < (int)m->comm.size() && m->comm[d]) ncclCommDestroy(m->comm[d]);
    for (auto *v : {&m->r1, &m->r2, &m->h1, &m->h2})
      if (d < (int)v->size())
        for (mods_imgrep *r : (*v)[d]) mods_imgrep_destroy(r);
    if (d < (int)m->gq.size()) mods_imgrep_destroy(m->gq[d]);
    if (d < (int)m->gt.size()) mods_imgrep_destroy(m->gt[d]);
    if (d < (int)m->send.size()) (void)hipFree(m->send[d]);
    if (d < (int)m->recv.size()) (void)hipFree(m->recv[d]);
    if (d < (int)m->img.size()) (void)hipFree(m->img[d]);
    mods_ctx_destroy(m->ctx[d]);
  }
  delete m;
}

// devices[n]: HIP device ids (a device may be listed more than once: the exchange then uses device copies instead of RCCL -
// a development aid for one-GPU boxes).  w, h: the larger image; rep_capacity: regions per image bank.
int mods_multi_create(const int *devices, int n, int w, int h, int rep_capacity, mods_multi **out) {
  if (!devices || n < 1 || n > 16 || !out || w <= 0 || h <= 0) { set_error("multi_create: bad arguments"); return MODS_E_ARG; }
  std::unique_ptr<mods_multi, void (*)(mods_multi *)> m(new mods_multi(), mods_multi_destroy);
  m->n = n;
  m->dev.assign(devices, devices + n);
  bool distinct = true;
  for (int a = 0; a < n; a++)
    for (int b = a + 1; b < n; b++) distinct = distinct && devices[a] != devices[b];
  const int side = (int)std::ceil(std::hypot((double)w, (double)h));     // a rotated view fits a side x side canvas
  m->img_px = (size_t)w * h * 2;
  m->side = side;
  m->rep_cap = rep_capacity > 0 ? rep_capacity : (1 << 20);
  m->cap = (size_t)m->rep_cap * 2 / n + (1 << 16);
  m->r1.resize(n); m->r2.resize(n); m->h1.resize(n); m->h2.resize(n);
  m->gq.assign(n, nullptr); m->gt.assign(n, nullptr);
  for (int d = 0; d < n; d++) {
    mods_ctx *c = nullptr;
    int rc = mods_ctx_create_ex(devices[d], side, side, 1, 1, &c);
    if (rc) return rc;
    m->ctx.push_back(c);
    if ((rc = bank(m.get(), m->r1, d, 0)) || (rc = bank(m.get(), m->r2, d, 0))) return rc;
    MODS_HIP_CHECK(hipSetDevice(devices[d]));
    mods_region *s = nullptr, *r = nullptr;
    float *im = nullptr;
    MODS_HIP_CHECK(hipMalloc(&s, sizeof(mods_region) * m->cap));
    m->send.push_back(s);
    MODS_HIP_CHECK(hipMalloc(&r, sizeof(mods_region) * m->cap * n));
    m->recv.push_back(r);
    MODS_HIP_CHECK(hipMalloc(&im, sizeof(float) * (size_t)w * h * 2));
    m->img.push_back(im);
  }
  if (distinct) {
    m->comm.assign(n, nullptr);
    const ncclResult_t nr = ncclCommInitAll(m->comm.data(), n, devices);
    if (nr != ncclSuccess) { set_error("ncclCommInitAll: %s", ncclGetErrorString(nr)); return MODS_E_HIP; }
    m->use_rccl = true;
  }
  *out = m.release();
  return MODS_OK;
}

int mods_multi_uses_rccl(const mods_multi *m) { return m && m->use_rccl ? 1 : 0; }
// the accumulated regions of image 1 / 2 after a run (device 0's copy of the banks; every device holds the same lists)
mods_imgrep *mods_multi_bank_det(mods_multi *m, int image, int det) {
  if (!m || det < 0) return nullptr;
  auto &v = image ? m->r2[0] : m->r1[0];
  return det < (int)v.size() ? v[det] : nullptr;
}
mods_imgrep *mods_multi_bank(mods_multi *m, int image) { return mods_multi_bank_det(m, image, 0); }

// The step loop of mods.cpp:202-383 with the views of every step sharded over the devices: mods_match_ladder_groups_dev's
// arguments (steps[step * n_det + det], dets[n_det], groups[n_steps] or NULL, group_pos) with the images in host memory.  Same
// results, field by field.
int mods_match_ladder_groups_multi(mods_multi *m, const float *img1_host, int w1, int h1, const float *img2_host, int w2, int h2,
                                   const mods_ladder_step *steps, const mods_hessaff_params *dets, const mods_ladder_group *groups, int group_pos,
                                   int n_steps, int n_det, int min_matches, const mods_pair_params *par, mods_ladder_result *res,
                                   double *matches_out, int max_matches) {
  if (!m || !img1_host || !img2_host || (!steps && n_steps > 0) || !dets || !par || !res || n_det < 1 || n_det > 8) {
    set_error("match_ladder_multi: bad argument"); return MODS_E_ARG;
  }
  if (groups && (group_pos < 0 || group_pos > n_det)) { set_error("match_ladder_multi: bad group position"); return MODS_E_ARG; }
  if (w1 <= 0 || h1 <= 0 || w2 <= 0 || h2 <= 0 || n_steps < 0) { set_error("match_ladder_multi: bad image size or step count"); return MODS_E_ARG; }
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 9; i++) res->H[i] = -1;
  const int D = m->n;
  const size_t px1 = (size_t)w1 * h1, px2 = (size_t)w2 * h2;
  // the device buffers were sized by mods_multi_create: both images must fit, and a rotated view of either must fit the canvas
  if (px1 + px2 > m->img_px || std::ceil(std::hypot((double)w1, (double)h1)) > m->side || std::ceil(std::hypot((double)w2, (double)h2)) > m->side) {
    set_error("match_ladder_multi: images %dx%d + %dx%d exceed what mods_multi_create was sized for (%zu pixels, canvas side %d)",
              w1, h1, w2, h2, m->img_px, m->side);
    return MODS_E_ARG;
  }
  int rc;
  for (int d = 0; d < D; d++) {
    MODS_HIP_CHECK(hipSetDevice(m->dev[d]));
    hipStream_t st = (hipStream_t)mods_ctx_stream(m->ctx[d]);
    MODS_HIP_CHECK(hipMemcpyAsync(m->img[d], img1_host, sizeof(float) * px1, hipMemcpyHostToDevice, st));
    MODS_HIP_CHECK(hipMemcpyAsync(m->img[d] + px1, img2_host, sizeof(float) * px2, hipMemcpyHostToDevice, st));
    for (int det = 0; det < n_det; det++) {
      if ((rc = bank(m, m->r1, d, det)) || (rc = bank(m, m->r2, d, det))) return rc;
      mods_imgrep_clear(m->r1[d][det]); mods_imgrep_clear(m->r2[d][det]);
    }
    for (auto *v : {&m->h1, &m->h2})
      for (mods_imgrep *r : (*v)[d]) if (r) mods_imgrep_clear(r);
  }
  struct PerDet { std::vector<mods_view_par> hist = std::vector<mods_view_par>(1024); int n_hist = 0; bool half = false; };
  std::vector<PerDet> pd(n_det);
  std::vector<mods_view_par> views(256);
  // per (descriptor, detector): kept from step to step until re-matched; with grouped matching one more slot ("Group")
  std::vector<TentList> lists[2];
  const int n_slots = n_det + (groups ? 1 : 0);
  lists[0].resize(n_slots); lists[1].resize(n_slots);
  auto slot_of = [&](int d) { return groups && d >= group_pos ? d + 1 : d; };
  int curr_matches = 0;
  std::vector<mods_tentative> tent;
  std::vector<double> u6, laf;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int step = 0; step < n_steps && curr_matches < min_matches; step++) {
    // ---- the step's view jobs in canonical order: detector, image, view (the order mods_match_ladder_groups_dev appends in)
    std::vector<Job> jobs;
    std::vector<double> areas;
    std::vector<int> new_views(n_det, 0);
    for (int det = 0; det < n_det; det++) {
      const mods_ladder_step &st = steps[(size_t)step * n_det + det];
      if (st.n_tilts < 0 || st.n_scales < 0) continue;          // the detector has no section in this step
      const int nv = mods_view_schedule(st.scale_set, st.n_scales, st.tilt_set, st.n_tilts, st.phi, pd[det].hist.data(), &pd[det].n_hist,
                                        (int)pd[det].hist.size(), views.data(), (int)views.size());
      if (nv < 0) return nv;
      new_views[det] = nv;
      const bool want_half = st.fginn_ratio_half > 0 || st.dist_threshold_half > 0;
      if (want_half) pd[det].half = true;
      for (int im = 0; im < 2; im++)
        for (int v = 0; v < nv; v++) {
          jobs.push_back({det, im, views[v], st.initSigma, st.doBlur, (st.half_orientation || want_half) ? 1 : 0, want_half ? 1 : 0});
          mods_view_geom g;
          if ((rc = mods_view_geometry(im ? w2 : w1, im ? h2 : h1, views[v].tilt, views[v].phi, views[v].zoom, st.initSigma, &g))) return rc;
          areas.push_back((double)g.w_new * g.h_new);
        }
    }
    for (int d = 0; d < D; d++)
      for (int det = 0; det < n_det; det++)
        if (pd[det].half && ((rc = bank(m, m->h1, d, det)) || (rc = bank(m, m->h2, d, det)))) return rc;
    std::vector<std::vector<int>> mine;
    assign_views(areas, D, &mine);
    const double t0 = now();
    // ---- every device: synthesise / detect / describe its views; a job's regions are packed RootSIFT first, then (when the
    // step asks for HalfRootSIFT lists) its HalfRootSIFT twins
    std::vector<int> job_count(jobs.size(), 0), job_det(jobs.size(), 0), job_unor(jobs.size(), 0);
    std::vector<int> rcs(D, MODS_OK);
    std::vector<std::string> errs(D);
    std::vector<int> sent(D, 0);
    {
      std::vector<std::thread> th;
      for (int d = 0; d < D; d++)
        th.emplace_back([&, d] {
          mods_ctx *c = m->ctx[d];
          if (hipSetDevice(m->dev[d]) != hipSuccess) { rcs[d] = MODS_E_HIP; errs[d] = "hipSetDevice"; return; }
          int off = 0;
          for (int j : mine[d]) {
            const Job &jb = jobs[j];
            mods_describe_params desc = par->desc;
            desc.ori_halfMode = jb.half_ori;
            desc.halfDesc = jb.want_half;
            int nd = 0, nr = 0;
            const float *img = m->img[d] + (jb.im ? px1 : 0);
            int rc2 = mods_detect_describe_view_dev(c, img, jb.im ? w2 : w1, jb.im ? h2 : h1, jb.im ? w2 : w1, jb.v.tilt, jb.v.phi, jb.v.zoom,
                                                    jb.init_sigma, jb.do_blur, &dets[jb.det], &desc, nullptr, &nd, &nr);
            const int slots = jb.want_half ? 2 * nr : nr;
            if (!rc2 && (size_t)off + slots > m->cap) { rc2 = MODS_E_CAPACITY; set_error("multi: exchange buffer too small"); }
            if (!rc2 && nr > 0) rc2 = mods_regions_copy_dev(c, 0, m->send[d] + off, nr);
            if (!rc2 && nr > 0 && jb.want_half) rc2 = mods_regions_half_copy_dev(c, 0, m->send[d] + off + nr, nr);
            if (rc2) { rcs[d] = rc2; errs[d] = mods_last_error(); return; }
            job_count[j] = nr; job_det[j] = nd; job_unor[j] = mods_unoriented_count(c, 0);
            off += slots;
          }
          sent[d] = off;
          mods_ctx_sync(c);
        });
      for (auto &t : th) t.join();
    }
    for (int d = 0; d < D; d++)
      if (rcs[d]) { set_error("device %d: %s", m->dev[d], errs[d].c_str()); return rcs[d]; }
    // ---- one all-gather of the padded blocks (the counts are known to the one host process: nothing to exchange first)
    size_t pad = 1;
    for (int d = 0; d < D; d++) pad = std::max(pad, (size_t)sent[d]);
    if (m->use_rccl) {
      ncclGroupStart();
      for (int d = 0; d < D; d++) {
        (void)hipSetDevice(m->dev[d]);
        const ncclResult_t nr = ncclAllGather(m->send[d], m->recv[d], pad * sizeof(mods_region), ncclChar, m->comm[d],
                                              (hipStream_t)mods_ctx_stream(m->ctx[d]));
        if (nr != ncclSuccess) { ncclGroupEnd(); set_error("ncclAllGather: %s", ncclGetErrorString(nr)); return MODS_E_HIP; }
      }
      const ncclResult_t ge = ncclGroupEnd();
      if (ge != ncclSuccess) { set_error("ncclGroupEnd: %s", ncclGetErrorString(ge)); return MODS_E_HIP; }
    } else {
      for (int dst = 0; dst < D; dst++) {
        (void)hipSetDevice(m->dev[dst]);
        for (int src = 0; src < D; src++)
          if (sent[src] > 0)
            MODS_HIP_CHECK(hipMemcpyAsync(m->recv[dst] + (size_t)src * pad, m->send[src], sizeof(mods_region) * sent[src],
                                          hipMemcpyDeviceToDevice, (hipStream_t)mods_ctx_stream(m->ctx[dst])));
      }
    }
    // ---- every device rebuilds the banks in canonical job order from what it received
    std::vector<int> owner(jobs.size(), 0), off_in_owner(jobs.size(), 0);
    for (int d = 0; d < D; d++) {
      int off = 0;
      for (int j : mine[d]) { owner[j] = d; off_in_owner[j] = off; off += jobs[j].want_half ? 2 * job_count[j] : job_count[j]; }
    }
    for (int d = 0; d < D; d++) {
      MODS_HIP_CHECK(hipSetDevice(m->dev[d]));
      for (size_t j = 0; j < jobs.size(); j++) {
        if (!job_count[j]) continue;
        const Job &jb = jobs[j];
        const mods_region *src = m->recv[d] + (size_t)owner[j] * pad + off_in_owner[j];
        if ((rc = mods_imgrep_append_dev(jb.im ? m->r2[d][jb.det] : m->r1[d][jb.det], src, job_count[j]))) return rc;
        if (jb.want_half && (rc = mods_imgrep_append_dev(jb.im ? m->h2[d][jb.det] : m->h1[d][jb.det], src + job_count[j], job_count[j]))) return rc;
      }
    }
    for (size_t j = 0; j < jobs.size(); j++) {
      res->n_views++;
      res->n_detected[jobs[j].im] += job_det[j];
      res->n_unoriented[jobs[j].im] += job_unor[j];
    }
    res->n_described[0] = res->n_described[1] = 0;
    for (int det = 0; det < n_det; det++) { res->n_described[0] += mods_imgrep_count(m->r1[0][det]); res->n_described[1] += mods_imgrep_count(m->r2[0][det]); }
    const double t1 = now();
    res->ms_detect_describe += t1 - t0;
    // ---- MatchImgReps (correspondencebank.cpp:245-340), the rules of mods_match_ladder_groups_dev, every search sharded by
    // query rows
    auto col = [&](std::vector<std::vector<mods_imgrep *>> &v, int det) {
      std::vector<mods_imgrep *> o(D, nullptr);
      for (int d = 0; d < D; d++) o[d] = det < (int)v[d].size() ? v[d][det] : nullptr;
      return o;
    };
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
          const int det = g.dets[i];
          if (det < 0 || det >= n_det) { set_error("match_ladder: group names detector %d of %d", det, n_det); return MODS_E_ARG; }
          const mods_imgrep *a = col(desc ? m->h1 : m->r1, det)[0], *b = col(desc ? m->h2 : m->r2, det)[0];
          nq += mods_imgrep_count(a); nt += mods_imgrep_count(b);
        }
        if (nq == 0 || nt == 0) continue;
        for (int d = 0; d < D; d++) {
          MODS_HIP_CHECK(hipSetDevice(m->dev[d]));
          if (!m->gq[d] && (rc = mods_imgrep_create(m->ctx[d], 2 * m->rep_cap, &m->gq[d]))) return rc;
          if (!m->gt[d] && (rc = mods_imgrep_create(m->ctx[d], 2 * m->rep_cap, &m->gt[d]))) return rc;
          mods_imgrep_clear(m->gq[d]); mods_imgrep_clear(m->gt[d]);
          for (int i = 0; i < g.n_dets; i++) {
            const int det = g.dets[i];
            const mods_imgrep *a = col(desc ? m->h1 : m->r1, det)[d], *b = col(desc ? m->h2 : m->r2, det)[d];
            if (a && mods_imgrep_count(a) && (rc = mods_imgrep_append_dev(m->gq[d], mods_imgrep_regions_dev(a), mods_imgrep_count(a)))) return rc;
            if (b && mods_imgrep_count(b) && (rc = mods_imgrep_append_dev(m->gt[d], mods_imgrep_regions_dev(b), mods_imgrep_count(b)))) return rc;
          }
        }
        if (ratio > 0 && (rc = match_sharded(m, m->gq, m->gt, ratio, 0, par, &out))) return rc;
        if (dist > 0 && (rc = match_sharded(m, m->gq, m->gt, 0, dist, par, &out))) return rc;
      }
    }
    for (int det = 0; det < n_det; det++) {
      const mods_ladder_step &st = steps[(size_t)step * n_det + det];
      if (st.n_tilts < 0 || st.n_scales < 0 || new_views[det] == 0) continue;
      if (st.fginn_ratio >= 0) {
        TentList &out = lists[0][slot_of(det)];
        out.clear();
        if (st.dist_threshold > 0) { if ((rc = match_sharded(m, col(m->r1, det), col(m->r2, det), 0, st.dist_threshold, par, &out))) return rc; }
        else if (st.fginn_ratio > 0 && (rc = match_sharded(m, col(m->r1, det), col(m->r2, det), st.fginn_ratio, 0, par, &out))) return rc;
      }
      if (st.fginn_ratio_half >= 0) {
        TentList &out = lists[1][slot_of(det)];
        out.clear();
        if (st.dist_threshold_half > 0 && pd[det].half) { if ((rc = match_sharded(m, col(m->h1, det), col(m->h2, det), 0, st.dist_threshold_half, par, &out))) return rc; }
        else if (st.fginn_ratio_half > 0 && (rc = match_sharded(m, col(m->h1, det), col(m->h2, det), st.fginn_ratio_half, 0, par, &out))) return rc;
      }
    }
    // the joint list in the bank's key order: HalfRootSIFT lists before RootSIFT lists, detectors in slot order
    tent.clear(); u6.clear(); laf.clear();
    for (int desc = 1; desc >= 0; desc--)
      for (const TentList &l : lists[desc]) {
        tent.insert(tent.end(), l.t.begin(), l.t.end());
        u6.insert(u6.end(), l.u6.begin(), l.u6.end());
        laf.insert(laf.end(), l.laf.begin(), l.laf.end());
      }
    const double t2 = now();
    res->ms_match += t2 - t1;
    res->n_tentatives = (int)tent.size();
    // ---- device 0's host: duplicate filter + verification
    int stats[3] = {0, 0, 0};
    double ms_dup = 0, ms_ran = 0;
    int gt3[3] = {0, 0, 0};
    rc = mods_verify_tentatives_ex(m->dev[0], par, tent.data(), u6.data(), laf.data(), (int)tent.size(), &res->n_unique, &res->n_inliers,
                                   res->H, stats, gt3, &ms_dup, &ms_ran);
    if (rc) return rc;
    res->ms_duplicates += ms_dup; res->ms_ransac += ms_ran;
    res->ransac_samples = stats[0]; res->ransac_lo = stats[1]; res->ransac_rejects = stats[2];
    res->gt_true = gt3[0]; res->gt_ransac_inliers = gt3[1]; res->gt_true_of_ransac = gt3[2];
    // the stop criterion of the step loop, as in mods_match_ladder_groups_dev (ground-truth mode: mods.cpp:381-383)
    curr_matches = !par->ransac.groundTruth ? res->n_inliers
                   : (par->ransac.ransacForStopping ? res->gt_ransac_inliers : (par->dup_before_ransac ? res->gt_true : res->n_inliers));
    res->steps_done = step + 1;
  }
  if (matches_out)
    for (int i = 0; i < res->n_inliers && i < max_matches; i++) {
      const double *p = &u6[(size_t)i * 6];
      matches_out[4 * i] = p[0]; matches_out[4 * i + 1] = p[1]; matches_out[4 * i + 2] = p[3]; matches_out[4 * i + 3] = p[4];
    }
  return MODS_OK;
}

// one HessianAffine detector (par->det): the form of mods_match_ladder_dev
int mods_match_ladder_multi(mods_multi *m, const float *img1_host, int w1, int h1, const float *img2_host, int w2, int h2,
                            const mods_ladder_step *steps, int n_steps, int min_matches, const mods_pair_params *par,
                            mods_ladder_result *res, double *matches_out, int max_matches) {
  if (!par) { set_error("match_ladder_multi: null argument"); return MODS_E_ARG; }
  return mods_match_ladder_groups_multi(m, img1_host, w1, h1, img2_host, w2, h2, steps, &par->det, nullptr, 0, n_steps, 1, min_matches, par, res,
                                        matches_out, max_matches);
}

}  // extern "C"
