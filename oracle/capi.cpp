// ORACLE — TEST INFRASTRUCTURE ONLY.  Plain-C entry points of the CPU restatement for the
// python tests (ctypes), smoke() and bench.py's cpu_baseline leg.  Never linked into the
// product library.
#include "orc.h"
#include "detmath.h"
#include <cmath>
#include <cstring>

using namespace orc;

extern "C" {

struct orc_hessaff_params {     // mirrors include/mods_hip.h: mods_hessaff_params
  int numberOfScales;
  float initialSigma;
  float threshold;
  float edgeEigenValueRatio;
  int border;
  int maxIterations;
  float convergenceThreshold;
  int smmWindowSize;
  int doBaumberg;
  int mode;
  float relativeThreshold;
  int regionsNumber;
  float relativeRegionsNumber;
  int detectorType;
  int iiDoGMode;
  int sampleFromImage;
  double mserMaxArea, mserMinMargin;
  int mserMinSize, affBmbrgMethod;
};

struct orc_candidate { int octave, level, r0, c0, r, c; float x, y, s, pixelDistance, response; int type; };
struct orc_affkey { double x, y, s, a11, a12, a21, a22, response; int sub_type, octave, level, r0, c0, pad; };

static HessAffParams cvt(const orc_hessaff_params *p) {
  HessAffParams q;
  if (p) {
    q.numberOfScales = p->numberOfScales; q.initialSigma = p->initialSigma; q.threshold = p->threshold;
    q.edgeEigenValueRatio = p->edgeEigenValueRatio; q.border = p->border; q.maxIterations = p->maxIterations;
    q.convergenceThreshold = p->convergenceThreshold; q.smmWindowSize = p->smmWindowSize; q.doBaumberg = p->doBaumberg;
    q.mode = p->mode; q.rel_threshold = p->relativeThreshold; q.reg_number = p->regionsNumber; q.rel_reg_number = p->relativeRegionsNumber;
    q.detector_type = p->detectorType; q.ii_dog = p->iiDoGMode; q.sample_from_image = p->sampleFromImage;
    q.mser_max_area = p->mserMaxArea; q.mser_min_margin = p->mserMinMargin; q.mser_min_size = p->mserMinSize;
    q.aff_bmbrg_method = p->affBmbrgMethod;
  }
  return q;
}
static Img wrap(const float *src, int w, int h) {
  Img im(w, h);
  std::memcpy(im.d.data(), src, sizeof(float) * (size_t)w * h);
  return im;
}

// experiment switches (tools/readme_count_hunt.py only); returns 0 for an unknown name
// threads used inside one oracle call (bench.py's row-parallel CPU baseline); results do not depend on it
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_set_variant(const char *name, int v) {
  if (!std::strcmp(name, "kernel")) g_variant.kernel = v;
  else if (!std::strcmp(name, "row_fma")) g_variant.row_fma = v;
  else if (!std::strcmp(name, "col_fma")) g_variant.col_fma = v;
  else if (!std::strcmp(name, "small_row")) g_variant.small_row = v;
  else if (!std::strcmp(name, "resize_tail")) g_variant.resize_tail = v;
  else if (!std::strcmp(name, "libm")) { g_variant.libm = v; g_libm_variant = v; }
  else return 0;
  return 1;
}
// (B + G + R) / 3.0 of GenerateSynthImageCorr (synth-detection.cpp:343-354) on interleaved 8-bit RGB: the MatExpr becomes
// addWeighted(B + G, 1/3, R, 1/3, 0) with float weights; an FMA build of OpenCV evaluates fma(B + G, a, R * a), a = (float)(1/3.).
void orc_grey_of_rgb(const unsigned char *rgb, long n, float *out) {
  const float a = (float)(1.0 / 3.0);
  for (long i = 0; i < n; i++) {
    const float bg = (float)rgb[3 * i + 2] + (float)rgb[3 * i + 1], r = (float)rgb[3 * i];
    out[i] = std::fmaf(bg, a, r * a);
  }
}
int orc_gauss_ksize(float sigma) { return gauss_ksize(sigma); }
void orc_gauss_kernel(int n, double sigma, float *out) {
  std::vector<float> k = gauss_kernel(n, sigma);
  std::memcpy(out, k.data(), sizeof(float) * n);
}
void orc_gauss_blur(const float *src, int w, int h, float sigma, float *dst) {
  Img a = wrap(src, w, h), b;
  gauss_blur(a, b, sigma);
  std::memcpy(dst, b.d.data(), sizeof(float) * (size_t)w * h);
}
void orc_hessian_response(const float *src, int w, int h, float norm, float *dst) {
  Img a = wrap(src, w, h), b;
  hessian_response(a, b, norm);
  std::memcpy(dst, b.d.data(), sizeof(float) * (size_t)w * h);
}
// kind 1 DoG, 2 Harris, 3 iiDoG (anything else: Hessian)
void orc_response(const float *src, int w, int h, float norm, int kind, float *dst) {
  Img a = wrap(src, w, h), b;
  if (kind == 1 || kind == 3) dog_response(a, b, norm, kind == 3);
  else if (kind == 2) harris_response(a, b, norm);
  else hessian_response(a, b, norm);
  std::memcpy(dst, b.d.data(), sizeof(float) * (size_t)w * h);
}
void orc_resize_half_dims(int w, int h, int *dw, int *dh) {
  Img a(w, h), b;
  resize_half(a, b);
  *dw = b.w; *dh = b.h;
}
void orc_resize_half(const float *src, int w, int h, float *dst) {
  Img a = wrap(src, w, h), b;
  resize_half(a, b);
  std::memcpy(dst, b.d.data(), sizeof(float) * (size_t)b.w * b.h);
}
// out = d0 d1 | U (4, row-major) | Vt (4); returns 1 when a singular value is <= FLT_MIN
int orc_svd2x2(const float *A, float *out) {
  bool deg = false;
  svd2x2_f32(A, out, out + 2, out + 6, &deg);
  return deg ? 1 : 0;
}
int orc_interpolate(const float *src, int w, int h, float ofsx, float ofsy, float a11, float a12, float a21,
                    float a22, int rw, int rh, float *dst) {
  Img a = wrap(src, w, h), r(rw, rh);
  bool t = interpolate(a, ofsx, ofsy, a11, a12, a21, a22, r);
  std::memcpy(dst, r.d.data(), sizeof(float) * (size_t)rw * rh);
  return t ? 1 : 0;
}
void orc_gauss_mask(int size, float *dst) {
  Img m(size, size); compute_gauss_mask(m);
  std::memcpy(dst, m.d.data(), sizeof(float) * (size_t)size * size);
}
void orc_circular_gauss_mask(int size, float sigma, float *dst) {
  Img m(size, size); compute_circular_gauss_mask(m, sigma);
  std::memcpy(dst, m.d.data(), sizeof(float) * (size_t)size * size);
}
float orc_det_pow2f(float x) { return det_pow2f(x); }
float orc_det_expf(float x) { return det_expf(x); }
void orc_det_sincos(double a, double *s, double *c) { det_sincos(a, s, c); }
float orc_atan2_lut(float y, float x) { return atan2_lut_ff(y, x); }

// ---- pyramid handle ----------------------------------------------------------------------
void *orc_pyramid_build(const float *img, int w, int h, const orc_hessaff_params *p) {
  Pyramid *pyr = new Pyramid();
  build_pyramid(wrap(img, w, h), cvt(p), *pyr);
  return pyr;
}
void orc_pyramid_free(void *h) { delete (Pyramid *)h; }
int orc_pyramid_octaves(void *h) { return (int)((Pyramid *)h)->oct.size(); }
void orc_pyramid_dims(void *h, int o, int *w, int *hh) { *w = ((Pyramid *)h)->oct[o].w; *hh = ((Pyramid *)h)->oct[o].h; }
// kind 0 = blur, 1 = response
void orc_pyramid_plane(void *h, int o, int level, int kind, float *dst) {
  const Pyramid::Oct &oc = ((Pyramid *)h)->oct[o];
  const Img &im = kind ? oc.resp[level] : oc.blur[level];
  std::memcpy(dst, im.d.data(), sizeof(float) * (size_t)im.w * im.h);
}
int orc_pyramid_candidates(void *h, const orc_hessaff_params *p, orc_candidate *out, int max_out,
                           int *nms_raw, int max_raw, int *n_raw) {
  std::vector<Candidate> c;
  std::vector<int> raw;
  find_candidates(*(Pyramid *)h, cvt(p), c, &raw);
  int n = (int)c.size();
  for (int i = 0; i < n && i < max_out; i++) {
    orc_candidate &o = out[i];
    o.octave = c[i].octave; o.level = c[i].level; o.r0 = c[i].r0; o.c0 = c[i].c0; o.r = c[i].r; o.c = c[i].c;
    o.x = c[i].x; o.y = c[i].y; o.s = c[i].s; o.pixelDistance = c[i].pixelDistance; o.response = c[i].response;
    o.type = c[i].type;
  }
  int nr = (int)raw.size() / 4;
  if (n_raw) *n_raw = nr;
  if (nms_raw) for (int i = 0; i < nr * 4 && i < max_raw * 4; i++) nms_raw[i] = raw[i];
  return n;
}
// Baumberg for one keypoint on plane (octave, level) of the pyramid.
int orc_affine_shape(void *h, int o, int level, float x, float y, float s, float pixelDistance,
                     const orc_hessaff_params *p, float *a4, int *iters) {
  HessAffParams q = cvt(p);
  Img mask(q.smmWindowSize, q.smmWindowSize);
  compute_gauss_mask(mask);
  return find_affine_shape(((Pyramid *)h)->oct[o].blur[level], x, y, s, pixelDistance, q, mask, a4, iters) ? 1 : 0;
}

int orc_detect_hessian_affine(const float *img, int w, int h, const orc_hessaff_params *p, orc_affkey *out,
                              int max_out) {
  std::vector<AffKey> k;
  detect_hessian_affine(wrap(img, w, h), cvt(p), k);
  int n = (int)k.size();
  for (int i = 0; i < n && i < max_out; i++) {
    orc_affkey &o = out[i];
    o.x = k[i].x; o.y = k[i].y; o.s = k[i].s; o.a11 = k[i].a11; o.a12 = k[i].a12; o.a21 = k[i].a21; o.a22 = k[i].a22;
    o.response = k[i].response; o.sub_type = k[i].sub_type; o.octave = k[i].octave; o.level = k[i].level;
    o.r0 = k[i].r0; o.c0 = k[i].c0; o.pad = 0;
  }
  return n;
}

// MSER regions of one image (MSER+ first, then MSER-; extremaRLERegions' order) with their run lists: rows of
// (thresh, margin, min_int, max_int, area, border, seed_x, seed_y, polarity, n_runs) in info10, the runs (line, col1, col2)
// concatenated in runs3, the ellipse (cx, cy, sxx, sxy, syy) in ell5.  Returns the number of regions; *n_runs_total = all runs.
int orc_mser_regions(const float *img, int w, int h, const orc_hessaff_params *p, int *info10, int max_regions, int *runs3,
                     int max_runs, double *ell5, int *n_runs_total) {
  HessAffParams q = cvt(p);
  std::vector<AffKey> k;
  std::vector<MserRegion> regs;
  q.mode = 0;
  detect_mser(wrap(img, w, h), q, 1.0, 1.0, k, &regs);
  int n_plus = 0;
  for (const AffKey &a : k) n_plus += a.sub_type == 21;
  int total = 0;
  for (size_t i = 0; i < regs.size(); i++) {
    const MserRegion &r = regs[i];
    if ((int)i < max_regions) {
      int *o = info10 + 10 * i;
      o[0] = r.thresh; o[1] = r.margin; o[2] = r.min_int; o[3] = r.max_int; o[4] = r.area; o[5] = r.border; o[6] = r.seed_x;
      o[7] = r.seed_y; o[8] = (int)i < n_plus ? 0 : 1; o[9] = (int)r.rle.size();
      double *e = ell5 + 5 * i;
      e[0] = r.cx; e[1] = r.cy; e[2] = r.sxx; e[3] = r.sxy; e[4] = r.syy;
    }
    for (const MserRun &u : r.rle) {
      if (total < max_runs) { runs3[3 * total] = u.line; runs3[3 * total + 1] = u.col1; runs3[3 * total + 2] = u.col2; }
      total++;
    }
  }
  if (n_runs_total) *n_runs_total = total;
  return (int)regs.size();
}
// DetectMSERs for a view with the SynthImage's tilt / zoom (the regionsNumber scaling of extrema.cpp:201-202)
int orc_detect_mser_view(const float *img, int w, int h, const orc_hessaff_params *p, double tilt, double zoom, orc_affkey *out, int max_out) {
  std::vector<AffKey> k;
  detect_mser(wrap(img, w, h), cvt(p), tilt, zoom, k);
  int n = (int)k.size();
  for (int i = 0; i < n && i < max_out; i++) {
    orc_affkey &o = out[i];
    o.x = k[i].x; o.y = k[i].y; o.s = k[i].s; o.a11 = k[i].a11; o.a12 = k[i].a12; o.a21 = k[i].a21; o.a22 = k[i].a22;
    o.response = k[i].response; o.sub_type = k[i].sub_type; o.octave = k[i].octave; o.level = k[i].level;
    o.r0 = k[i].r0; o.c0 = k[i].c0; o.pad = 0;
  }
  return n;
}

// ---- orientation + description -------------------------------------------------------------
struct orc_region {            // mirrors include/mods_hip.h: mods_region
  double x, y, s, a11, a12, a21, a22, response;
  int sub_type, id, parent, pad;
  uint8_t desc[128];
};

static void to_regions(const orc_region *in, int n, std::vector<Region> &v) {
  v.resize(n);
  for (int i = 0; i < n; i++) {
    Region &r = v[i]; const orc_region &o = in[i];
    r.x = o.x; r.y = o.y; r.s = o.s; r.a11 = o.a11; r.a12 = o.a12; r.a21 = o.a21; r.a22 = o.a22; r.response = o.response;
    r.sub_type = o.sub_type; r.id = o.id; r.parent = o.parent;
    std::memcpy(r.desc, o.desc, 128);
  }
}
static int from_regions(const std::vector<Region> &v, orc_region *out, int max_out) {
  int n = (int)v.size();
  for (int i = 0; i < n && i < max_out; i++) {
    const Region &r = v[i]; orc_region &o = out[i];
    o.x = r.x; o.y = r.y; o.s = r.s; o.a11 = r.a11; o.a12 = r.a12; o.a21 = r.a21; o.a22 = r.a22; o.response = r.response;
    o.sub_type = r.sub_type; o.id = r.id; o.parent = r.parent; o.pad = 0;
    std::memcpy(o.desc, r.desc, 128);
  }
  return n;
}

int orc_dominant_angle_half(const float *patch, int ps, double th, float *angle) {
  return dominant_angle(wrap(patch, ps, ps), th, angle, true) ? 1 : 0;
}
void orc_half_rootsift_patch(const float *patch, int ps, double maxBinValue, unsigned char *out128) {
  sift_patch_to_desc(wrap(patch, ps, ps), out128, true, maxBinValue, true);
}
int orc_dominant_angle(const float *patch, int ps, double th, float *angle) {
  return dominant_angle(wrap(patch, ps, ps), th, angle) ? 1 : 0;
}
void orc_sift_desc(const float *patch, int ps, int rootsift, double maxBinValue, uint8_t *out128) {
  sift_patch_to_desc(wrap(patch, ps, ps), out128, rootsift != 0, maxBinValue);
}
void orc_extract_desc_patch(const float *img, int w, int h, const orc_region *r, double mrSize, int patchSize,
                            int photoNorm, float *patch_out) {
  std::vector<Region> v; to_regions(r, 1, v);
  Img patch(patchSize, patchSize);
  extract_desc_patch(v[0], wrap(img, w, h), mrSize, patchSize, photoNorm != 0, patch);
  std::memcpy(patch_out, patch.d.data(), sizeof(float) * (size_t)patchSize * patchSize);
}
// ExtractPatchesColumn (non-fast, no photometric normalisation): n patches of patchSize x patchSize, fp32
void orc_extract_patches_column(const float *img, int w, int h, const orc_region *r, int n, double mrSize, int patchSize,
                                float *patches_out) {
  std::vector<Region> v; to_regions(r, n, v);
  Img im = wrap(img, w, h);
  Img patch(patchSize, patchSize);
  for (int i = 0; i < n; i++) {
    extract_desc_patch(v[i], im, mrSize, patchSize, false, patch, true);
    std::memcpy(patches_out + (size_t)i * patchSize * patchSize, patch.d.data(), sizeof(float) * (size_t)patchSize * patchSize);
  }
}
int orc_detect_orientation(const float *img, int w, int h, const orc_region *in, int n, double mrSize, int patchSize,
                           int maxAngles, double th, orc_region *out, int max_out) {
  std::vector<Region> v, o; to_regions(in, n, v);
  detect_orientation(v, o, wrap(img, w, h), mrSize, patchSize, maxAngles, th);
  return from_regions(o, out, max_out);
}
int orc_filter_centres_inside(orc_region *r, int n, int w, int h) {
  std::vector<Region> v; to_regions(r, n, v);
  filter_centres_inside(v, w, h);
  return from_regions(v, r, n);
}
int orc_affnet_apply(orc_region *r, int n, const float *a3, int w, int h, double mrSize) {
  std::vector<Region> v; to_regions(r, n, v);
  affnet_apply(v, a3, w, h, mrSize);
  return from_regions(v, r, n);
}
void orc_orinet_apply(orc_region *r, int n, const float *yx) {
  std::vector<Region> v; to_regions(r, n, v);
  orinet_apply(v, yx);
  from_regions(v, r, n);
}
int orc_filter_touch_boundary(orc_region *r, int n, int w, int h) {
  std::vector<Region> v; to_regions(r, n, v);
  filter_touch_boundary(v, w, h);
  return from_regions(v, r, n);
}
void orc_describe_rootsift(const float *img, int w, int h, orc_region *r, int n, double mrSize, int patchSize, int photoNorm) {
  std::vector<Region> v; to_regions(r, n, v);
  describe_rootsift(v, wrap(img, w, h), mrSize, patchSize, photoNorm != 0);
  from_regions(v, r, n);
}

// SynthDetectDescribeKeypoints for one identity view, HessianAffine + RootSIFT
// (imagerepresentation.cpp:686-1104): detect -> centres inside -> orientation -> touch-boundary
// filter -> RootSIFT.
// flags: bit 2 = [DominantOrientation] addUpRight; bit 0 = DetectOrientation in doHalfSIFT mode (what the reference does for every descriptor of a view as soon as one
// descriptor name of the step contains "Half", imagerepresentation.cpp:725-731, 909-943); bit 1 = also HalfRootSIFT:
// out_half (same regions, desc[0..63] = HalfRootSIFT, desc[64..127] = 0) is filled.
int orc_detect_describe_ex(const float *img, int w, int h, const orc_hessaff_params *p, double ori_mrSize,
                           int ori_patchSize, int maxAngles, double ori_th, double desc_mrSize, int desc_patchSize,
                           int photoNorm, int flags, orc_region *out, orc_region *out_half, int max_out, int *n_detected) {
  Img im = wrap(img, w, h);
  std::vector<AffKey> k;
  detect_hessian_affine(im, cvt(p), k);
  if (n_detected) *n_detected = (int)k.size();
  std::vector<Region> v(k.size()), o;
  for (size_t i = 0; i < k.size(); i++) {
    Region &r = v[i];
    r.x = k[i].x; r.y = k[i].y; r.s = k[i].s; r.a11 = k[i].a11; r.a12 = k[i].a12; r.a21 = k[i].a21; r.a22 = k[i].a22;
    r.response = k[i].response; r.sub_type = k[i].sub_type; r.id = (int)i; r.parent = (int)i;
    std::memset(r.desc, 0, 128);
  }
  filter_centres_inside(v, w, h);
  detect_orientation(v, o, im, ori_mrSize, ori_patchSize, maxAngles, ori_th, (flags & 1) != 0);
  if (flags & 4) {   // [DominantOrientation] addUpRight (imagerepresentation.cpp:915-930): the upright copies come first
    std::vector<Region> up;
    detect_orientation(v, up, im, ori_mrSize, ori_patchSize, 0, 1.0, false, true);
    up.insert(up.end(), o.begin(), o.end());
    o.swap(up);
  }
  filter_touch_boundary(o, w, h);
  if ((flags & 2) && out_half) {
    std::vector<Region> hv = o;
    describe_rootsift(hv, im, desc_mrSize, desc_patchSize, photoNorm != 0, true, (flags & 8) != 0);
    from_regions(hv, out_half, max_out);
  }
  describe_rootsift(o, im, desc_mrSize, desc_patchSize, photoNorm != 0, false, (flags & 8) != 0);   // bit 3: [SIFTDescriptor] FastPatchExtraction
  return from_regions(o, out, max_out);
}
int orc_detect_describe(const float *img, int w, int h, const orc_hessaff_params *p, double ori_mrSize,
                        int ori_patchSize, int maxAngles, double ori_th, double desc_mrSize, int desc_patchSize,
                        int photoNorm, orc_region *out, int max_out, int *n_detected) {
  return orc_detect_describe_ex(img, w, h, p, ori_mrSize, ori_patchSize, maxAngles, ori_th, desc_mrSize, desc_patchSize, photoNorm, 0, out,
                                nullptr, max_out, n_detected);
}

// ---- view synthesis -------------------------------------------------------------------------------
struct orc_view_geom {          // mirrors include/mods_hip.h: mods_view_geom
  int identity, w_rot, h_rot, w_new, h_new, ksize_x, ksize_y, pad;
  double rotation, tilt, zoom, sigma_x, sigma_y;
  double H[9], warpRot[6], warpTilt[6];
};
static void cvt_geom(const ViewGeom &g, orc_view_geom *o) {
  o->identity = g.identity ? 1 : 0; o->w_rot = g.w_rot; o->h_rot = g.h_rot; o->w_new = g.w_new; o->h_new = g.h_new;
  o->ksize_x = g.identity ? 0 : g.ksize_x; o->ksize_y = g.identity ? 0 : g.ksize_y; o->pad = 0;
  o->rotation = g.rotation; o->tilt = g.tilt; o->zoom = g.zoom;
  o->sigma_x = g.identity ? 0 : g.sigma_x; o->sigma_y = g.identity ? 0 : g.sigma_y;
  for (int i = 0; i < 9; i++) o->H[i] = g.H[i];
  for (int i = 0; i < 6; i++) { o->warpRot[i] = g.identity ? 0 : g.warpRot[i]; o->warpTilt[i] = g.identity ? 0 : g.warpTilt[i]; }
}
void orc_view_geometry(int w, int h, double tilt, double phi, double zoom, double initSigma, orc_view_geom *out) {
  ViewGeom g;
  view_geometry(w, h, tilt, phi, zoom, initSigma, &g);
  cvt_geom(g, out);
}
void orc_warp_affine(const float *src, int w, int h, const double *M, int dw, int dh, float cval, float *dst) {
  Img o;
  warp_affine(wrap(src, w, h), M, dw, dh, cval, o);
  std::memcpy(dst, o.d.data(), sizeof(float) * (size_t)dw * dh);
}
void orc_gauss_blur_xy(const float *src, int w, int h, int kx, int ky, double sx, double sy, float *dst) {
  Img o;
  gauss_blur_xy(wrap(src, w, h), o, kx, ky, sx, sy);
  std::memcpy(dst, o.d.data(), sizeof(float) * (size_t)w * h);
}
// dst must hold w_new*h_new floats of orc_view_geometry for the same arguments
void orc_synth_view(const float *src, int w, int h, double tilt, double phi, double zoom, double initSigma, int doBlur,
                    float *dst, orc_view_geom *geom) {
  ViewGeom g;
  Img o;
  generate_synth_view(wrap(src, w, h), tilt, phi, zoom, initSigma, doBlur, o, &g);
  std::memcpy(dst, o.d.data(), sizeof(float) * o.d.size());
  if (geom) cvt_geom(g, geom);
}

// SynthDetectDescribeKeypoints for one synthesised view (imagerepresentation.cpp:704-1099), HessianAffine +
// RootSIFT: detect on the view -> centres (in the original frame) inside -> orientation on the view ->
// ReprojectRegions -> RootSIFT on the view.  `out` carries reproj_kp (original frame) + descriptor;
// `out_det` (optional) the matching det_kp (view frame).
int orc_detect_describe_view_ex(const float *view, int vw, int vh, const double *H, int orig_w, int orig_h,
                                const orc_hessaff_params *p, double ori_mrSize, int ori_patchSize, int maxAngles, double ori_th,
                                double desc_mrSize, int desc_patchSize, int photoNorm, int flags, orc_region *out, orc_region *out_det,
                                orc_region *out_half, int max_out, int *n_detected) {
  Img im = wrap(view, vw, vh);
  std::vector<AffKey> k;
  detect_hessian_affine(im, cvt(p), k);
  if (n_detected) *n_detected = (int)k.size();
  std::vector<Region> v(k.size()), o, rep;
  for (size_t i = 0; i < k.size(); i++) {
    Region &r = v[i];
    r.x = k[i].x; r.y = k[i].y; r.s = k[i].s; r.a11 = k[i].a11; r.a12 = k[i].a12; r.a21 = k[i].a21; r.a22 = k[i].a22;
    r.response = k[i].response; r.sub_type = k[i].sub_type; r.id = (int)i; r.parent = (int)i;
    std::memset(r.desc, 0, 128);
  }
  filter_centres_inside_view(v, H, orig_w, orig_h);
  detect_orientation(v, o, im, ori_mrSize, ori_patchSize, maxAngles, ori_th, (flags & 1) != 0);
  reproject_regions_view(o, rep, H, orig_w, orig_h);
  if ((flags & 2) && out_half) {
    std::vector<Region> hv = o, hrep = rep;
    describe_rootsift(hv, im, desc_mrSize, desc_patchSize, photoNorm != 0, true);
    for (size_t i = 0; i < hv.size(); i++) std::memcpy(hrep[i].desc, hv[i].desc, 128);
    from_regions(hrep, out_half, max_out);
  }
  describe_rootsift(o, im, desc_mrSize, desc_patchSize, photoNorm != 0);
  for (size_t i = 0; i < o.size(); i++) std::memcpy(rep[i].desc, o[i].desc, 128);
  if (out_det) from_regions(o, out_det, max_out);
  return from_regions(rep, out, max_out);
}
int orc_detect_describe_view(const float *view, int vw, int vh, const double *H, int orig_w, int orig_h,
                             const orc_hessaff_params *p, double ori_mrSize, int ori_patchSize, int maxAngles, double ori_th,
                             double desc_mrSize, int desc_patchSize, int photoNorm, orc_region *out, orc_region *out_det,
                             int max_out, int *n_detected) {
  return orc_detect_describe_view_ex(view, vw, vh, H, orig_w, orig_h, p, ori_mrSize, ori_patchSize, maxAngles, ori_th, desc_mrSize,
                                     desc_patchSize, photoNorm, 0, out, out_det, nullptr, max_out, n_detected);
}

// ---- matching -----------------------------------------------------------------------------------
struct orc_tentative { int q, t, t_bad, t_2nd; float d1, d2, d2nd, pad; double ratio; };   // mirrors mods_tentative

int orc_match_fginn(const orc_region *q, int nq, const orc_region *t, int nt, double ratio, double contradDist, int nn,
                    orc_tentative *out, int max_out) {
  std::vector<Region> a, b; to_regions(q, nq, a); to_regions(t, nt, b);
  std::vector<Tentative> tc;
  match_fginn(a, b, tc, ratio, contradDist, nn);
  for (size_t i = 0; i < tc.size() && (int)i < max_out; i++) {
    orc_tentative &o = out[i];
    o.q = tc[i].q; o.t = tc[i].t; o.t_bad = tc[i].t_bad; o.t_2nd = tc[i].t_2nd; o.d1 = tc[i].d1; o.d2 = tc[i].d2;
    o.d2nd = tc[i].d2nd; o.pad = 0; o.ratio = tc[i].ratio;
  }
  return (int)tc.size();
}
int orc_match_distance(const orc_region *q, int nq, const orc_region *t, int nt, double threshold, orc_tentative *out, int max_out) {
  std::vector<Region> a, b; to_regions(q, nq, a); to_regions(t, nt, b);
  std::vector<Tentative> tc;
  match_distance(a, b, tc, threshold);
  for (size_t i = 0; i < tc.size() && (int)i < max_out; i++) {
    orc_tentative &o = out[i];
    o.q = tc[i].q; o.t = tc[i].t; o.t_bad = tc[i].t_bad; o.t_2nd = tc[i].t_2nd; o.d1 = tc[i].d1; o.d2 = tc[i].d2;
    o.d2nd = tc[i].d2nd; o.pad = 0; o.ratio = tc[i].ratio;
  }
  return (int)tc.size();
}
int orc_duplicate_filter(orc_tentative *tcs, int n, const orc_region *q, int nq, const orc_region *t, int nt, double r,
                         int mode) {
  std::vector<Region> a, b; to_regions(q, nq, a); to_regions(t, nt, b);
  std::vector<Tentative> tc(n);
  for (int i = 0; i < n; i++) {
    tc[i].q = tcs[i].q; tc[i].t = tcs[i].t; tc[i].t_bad = tcs[i].t_bad; tc[i].t_2nd = tcs[i].t_2nd; tc[i].d1 = tcs[i].d1;
    tc[i].d2 = tcs[i].d2; tc[i].d2nd = tcs[i].d2nd; tc[i].ratio = tcs[i].ratio;
  }
  duplicate_filter(tc, a, b, r, mode);
  for (size_t i = 0; i < tc.size(); i++) {
    orc_tentative &o = tcs[i];
    o.q = tc[i].q; o.t = tc[i].t; o.t_bad = tc[i].t_bad; o.t_2nd = tc[i].t_2nd; o.d1 = tc[i].d1; o.d2 = tc[i].d2;
    o.d2nd = tc[i].d2nd; o.pad = 0; o.ratio = tc[i].ratio;
  }
  return (int)tc.size();
}

}  // extern "C"
