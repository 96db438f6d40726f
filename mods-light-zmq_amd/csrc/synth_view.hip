// View synthesis: rotate -> anisotropic anti-aliasing blur -> tilt/zoom warp of a grey float image in
// HBM, and the detect/describe chain of one synthesised view.
//
// Reference behaviour: GenerateSynthImageCorr, synth-detection.cpp:324-518 (non-AREA_INTERP build),
// called per view from ImageRepresentation::SynthDetectDescribeKeypoints, imagerepresentation.cpp:712.
// Its pixel work is two cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT 128) calls and one
// cv::GaussianBlur(ksize 3 or 5, BORDER_REFLECT_101); OpenCV is not pinned by the reference, the
// semantics implemented are the ones written down in oracle/synth_view.cpp (fixed-point inverse map in
// 1/1024 px, rounded to 1/32 px, 32x32 table of bilinear weights; small symmetric filter orders).
//
// All three kernels are HBM-streaming (4 B read-ish + 4 B written per output pixel); the warp gathers
// its four taps through L2.  One thread per output pixel, 64x4 tiles so that a wave covers one row
// segment (coalesced stores; the rotation makes loads walk a slanted line, which L2 absorbs).
#include "common.hpp"
#include "detmath.hpp"
#include <cmath>

namespace mods {

struct WarpMat { double m[6]; };   // inverse map dst -> src, as OpenCV leaves it after inverting M

// cvRound(double): round half to even (default FP rounding mode)
__device__ __forceinline__ int cv_round_dev(double v) { return (int)rint(v); }

// grid = (ceil(dw/64), ceil(dh/4)), block = (64, 4)
__global__ void __launch_bounds__(256) warp_affine_kernel(const float *__restrict__ src, int sw, int sh, int sstride, WarpMat M,
                                                          float *__restrict__ dst, int dw, int dh, int dstride, float cval) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 1 << INTER_BITS;
  const int round_delta = AB_SCALE / TAB / 2;
  const int adelta = cv_round_dev(M.m[0] * x * AB_SCALE);
  const int bdelta = cv_round_dev(M.m[3] * x * AB_SCALE);
  const int X0 = cv_round_dev((M.m[1] * y + M.m[2]) * AB_SCALE) + round_delta;
  const int Y0 = cv_round_dev((M.m[4] * y + M.m[5]) * AB_SCALE) + round_delta;
  const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS);
  const int Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
  const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
  const float fx = (float)(X & (TAB - 1)) * (1.f / TAB), fy = (float)(Y & (TAB - 1)) * (1.f / TAB);
  const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
  float out;
  if ((unsigned)sx < (unsigned)max(sw - 1, 0) && (unsigned)sy < (unsigned)max(sh - 1, 0)) {
    const float *S = src + (size_t)sy * sstride + sx;
    out = S[0] * w0 + S[1] * w1 + S[sstride] * w2 + S[sstride + 1] * w3;
  } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
    out = cval;
  } else {
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
    const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    const float v0 = x0 && y0 ? src[(size_t)sy * sstride + sx] : cval;
    const float v1 = x1 && y0 ? src[(size_t)sy * sstride + sx + 1] : cval;
    const float v2 = x0 && y1 ? src[(size_t)(sy + 1) * sstride + sx] : cval;
    const float v3 = x1 && y1 ? src[(size_t)(sy + 1) * sstride + sx + 1] : cval;
    out = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
  }
  dst[(size_t)y * dstride + x] = out;
}

__device__ __forceinline__ int reflect101_dev(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

struct SmallTaps { float k[128]; int n; };   // n <= 127 taps (tilt 6 at zoom 0.25 with initSigma 0.8, [MSER1] of iters_MODS.ini: 59)

// Row pass of cv::GaussianBlur with BORDER_REFLECT_101: n <= 5 -> symmetric small filter order
// (centre, then pairs outwards); larger -> generic left-to-right order.  Fused multiply-adds as an FMA build of OpenCV
// has them (the contract shared with the oracle, see pyramid.hip).
__global__ void __launch_bounds__(256) blur_row_reflect_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h, SmallTaps t) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w || y >= h) return;
  const float *S = src + (size_t)y * w;
  const int r = t.n / 2;
  float s;
  if (t.n <= 5) {
    s = S[x] * t.k[r];
    for (int j = 1; j <= r; j++) s = fmaf(S[reflect101_dev(x - j, w)] + S[reflect101_dev(x + j, w)], t.k[r + j], s);
  } else {
    s = t.k[0] * S[reflect101_dev(x - r, w)];
    for (int j = 1; j < t.n; j++) s = fmaf(t.k[j], S[reflect101_dev(x - r + j, w)], s);
  }
  dst[(size_t)y * w + x] = s;
}

// Column pass: n == 3 -> fma(S0 + S2, f1, S1*f0) (small symmetric column filter); otherwise centre first, then pairs.
__global__ void __launch_bounds__(256) blur_col_reflect_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h, SmallTaps t) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w || y >= h) return;
  const int r = t.n / 2;
  float s;
  if (t.n == 3) {
    s = fmaf(src[(size_t)reflect101_dev(y - 1, h) * w + x] + src[(size_t)reflect101_dev(y + 1, h) * w + x], t.k[2], src[(size_t)y * w + x] * t.k[1]);
  } else {
    s = t.k[r] * src[(size_t)y * w + x];
    for (int j = 1; j <= r; j++) s = fmaf(t.k[r + j], src[(size_t)reflect101_dev(y + j, h) * w + x] + src[(size_t)reflect101_dev(y - j, h) * w + x], s);
  }
  dst[(size_t)y * w + x] = s;
}

// inverse of the 2x3 map the way cv::warpAffine does it (imgwarp: D = 1/det, adjugate, back-substituted shift)
static WarpMat invert_affine(const double *Mf) {
  WarpMat W;
  double *M = W.m;
  for (int i = 0; i < 6; i++) M[i] = Mf[i];
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D;
  M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  return W;
}

int launch_warp_affine(mods_ctx *c, const float *src, int sw, int sh, int sstride, const double *M, float *dst, int dw, int dh,
                       int dstride, float cval) {
  if (dw <= 0 || dh <= 0) return MODS_OK;
  const WarpMat W = invert_affine(M);
  hipLaunchKernelGGL(warp_affine_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(64, 4), 0, c->stream, src, sw, sh, sstride, W, dst, dw,
                     dh, dstride, cval);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

int launch_blur_xy_reflect(mods_ctx *c, const float *src, float *tmp, float *dst, int w, int h, int kx, int ky, double sx, double sy) {
  if (kx > 127 || ky > 127 || !(kx & 1) || !(ky & 1)) { set_error("anti-aliasing kernel size %dx%d not supported", kx, ky); return MODS_E_ARG; }
  SmallTaps tx, ty;
  tx.n = kx; ty.n = ky;
  gauss_kernel_host(kx, sx, tx.k);
  gauss_kernel_host(ky, sy, ty.k);
  const dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
  hipLaunchKernelGGL(blur_row_reflect_kernel, grid, block, 0, c->stream, src, tmp, w, h, tx);
  hipLaunchKernelGGL(blur_col_reflect_kernel, grid, block, 0, c->stream, tmp, dst, w, h, ty);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods

using namespace mods;

extern "C" {

// Everything GenerateSynthImageCorr derives from (w, h, tilt, phi, zoom, InitSigma) before it touches
// pixels, synth-detection.cpp:336-469.  Host-side double arithmetic + libm, line by line.
int mods_view_geometry(int w, int h, double tilt, double phi, double zoom, double InitSigma, mods_view_geom *g) {
  if (!g || w <= 0 || h <= 0) { set_error("view_geometry: bad arguments"); return MODS_E_ARG; }
  memset(g, 0, sizeof(*g));
  int zoomed = 0;
  bool vertical_tilt = false;
  if (tilt < 0) { tilt = -tilt; vertical_tilt = true; }
  if (fabs(zoom - 1.0f) >= 0.05) zoomed = 1;
  const int wS1 = (int)(w * zoom), hS1 = (int)(h * zoom);
  if ((fabs(tilt - 1.) <= 0.1) && (fabs(phi) <= 0.2) && (fabs(zoom - 1.) <= 0.1)) {   // original image
    g->identity = 1; g->rotation = 0.0; g->tilt = 1.0; g->zoom = 1.0;
    g->H[0] = 1.0; g->H[4] = 1.0; g->H[8] = 1.0;
    g->w_new = g->w_rot = w; g->h_new = g->h_rot = h;
    return MODS_OK;
  }
  g->rotation = phi * 180 / M_PI; g->tilt = tilt; g->zoom = zoom;
  double kV = 1., kH = 1.;
  if (zoomed) { kV = (double)w / (double)wS1; kH = (double)h / (double)hS1; }
  double cp, sp;
  det_sincos(phi, &sp, &cp);   // the contract's cos / sin (detmath.hpp): libm's sincos() and cos()/sin() disagree in the last bit
  const bool first = (phi >= 0) && (phi < M_PI / 2);
  // the tilt compresses x (division by tilt*kH) unless it is "vertical", in which case it compresses y
  const double fx = vertical_tilt ? kH : tilt * kH;
  const double fy = vertical_tilt ? tilt * kV : kV;
  double w_new, h_new;
  double *H = g->H;
  H[0] = cp / fx; H[1] = sp / fx;
  H[3] = -sp / fy; H[4] = cp / fy;
  if (first) {
    w_new = floor((0.5 + cp * w + sp * h) / fx);
    h_new = floor((0.5 + sp * w + cp * h) / fy);
    H[2] = 0;
    H[5] = floor(0.5 + sp * w / fy);
  } else {
    w_new = floor((0.5 - cp * w + sp * h) / fx);
    h_new = floor((0.5 + sp * w - cp * h) / fy);
    H[2] = -floor(cp * w / fx);
    H[5] = floor(0.5 + (sp * w - cp * h) / fy);
  }
  H[6] = 0; H[7] = 0; H[8] = 1;
  g->w_new = (int)w_new; g->h_new = (int)h_new;
  const double sigma_aa_2 = zoomed ? InitSigma / (4.0 * zoom) : InitSigma / 2.0;
  const double sigma_aa = InitSigma * tilt / (2.0 * zoom);
  g->sigma_x = vertical_tilt ? sigma_aa_2 : sigma_aa;
  g->sigma_y = vertical_tilt ? sigma_aa : sigma_aa_2;
  double *R = g->warpRot;
  R[0] = cp; R[1] = sp; R[3] = -sp; R[4] = cp;
  if (first) {
    g->w_rot = (int)floor((0.5 + cp * w + sp * h));
    g->h_rot = (int)floor((0.5 + sp * w + cp * h));
    R[2] = 0; R[5] = floor(0.5 + sp * w);
  } else {
    g->w_rot = (int)floor((0.5 - cp * w + sp * h));
    g->h_rot = (int)floor((0.5 + sp * w - cp * h));
    R[2] = -floor(cp * w); R[5] = floor(0.5 + (sp * w - cp * h));
  }
  int kx = (int)floor(2.0 * 3.0 * g->sigma_x + 1.0);
  if (kx % 2 == 0) kx++;
  if (kx < 3) kx = 3;
  int ky = (int)floor(2.0 * 3.0 * g->sigma_y + 1.0);
  if (ky % 2 == 0) ky++;
  if (ky < 3) ky = 3;
  g->ksize_x = kx; g->ksize_y = ky;
  double *T = g->warpTilt;
  T[0] = 1.0 / fx; T[4] = 1.0 / fy;
  return MODS_OK;
}

// Pixels of one view.  src_dev: w x h grey float image in HBM (row stride `stride` floats); dst_dev: g->w_new x
// g->h_new, dense.  The context needs room for the rotated intermediate (g->w_rot * g->h_rot <= max_w * max_h).
int mods_synth_view_dev(mods_ctx *c, const float *src_dev, int w, int h, int stride, const mods_view_geom *g, int doBlur, float *dst_dev) {
  if (!c || !src_dev || !g || !dst_dev) { set_error("synth_view: null argument"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  if (g->identity) {
    MODS_HIP_CHECK(hipMemcpy2DAsync(dst_dev, sizeof(float) * w, src_dev, sizeof(float) * stride, sizeof(float) * w, h, hipMemcpyDeviceToDevice, c->stream));
    return MODS_OK;
  }
  const size_t cap = (size_t)c->max_w * c->max_h * c->batch;
  if ((size_t)g->w_rot * g->h_rot > cap) { set_error("rotated view %dx%d larger than the context", g->w_rot, g->h_rot); return MODS_E_ARG; }
  StageScope ts(c, MODS_STAGE_SYNTH, 4.0 * ((double)w * h + 5.0 * g->w_rot * g->h_rot + (double)g->w_new * g->h_new));
  int rc;
  float *rot = c->tmp_dev, *scratch = c->input_dev;
  if ((rc = launch_warp_affine(c, src_dev, w, h, stride, g->warpRot, rot, g->w_rot, g->h_rot, g->w_rot, 128.f))) return rc;
  if (doBlur) {
    if ((rc = launch_blur_xy_reflect(c, rot, scratch, rot, g->w_rot, g->h_rot, g->ksize_x, g->ksize_y, g->sigma_x, g->sigma_y))) return rc;
  }
  return launch_warp_affine(c, rot, g->w_rot, g->h_rot, g->w_rot, g->warpTilt, dst_dev, g->w_new, g->h_new, g->w_new, 128.f);
}

// host-buffer primitives for the parity tests
int mods_warp_affine(mods_ctx *c, const float *src, int w, int h, const double *M, int dw, int dh, float cval, float *dst) {
  if (!c || !src || !M || !dst) { set_error("warp_affine: null argument"); return MODS_E_ARG; }
  const size_t cap = (size_t)c->max_w * c->max_h * c->batch;
  if ((size_t)w * h > cap || (size_t)dw * dh > cap) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, src, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice, c->stream));
  int rc = launch_warp_affine(c, c->input_dev, w, h, w, M, c->tmp_dev, dw, dh, dw, cval);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(dst, c->tmp_dev, sizeof(float) * (size_t)dw * dh, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

int mods_gauss_blur_xy(mods_ctx *c, const float *src, int w, int h, int kx, int ky, double sx, double sy, float *dst) {
  if (!c || !src || !dst) { set_error("gauss_blur_xy: null argument"); return MODS_E_ARG; }
  if ((size_t)w * h > (size_t)c->max_w * c->max_h) { set_error("image larger than the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  if (!c->view_dev) MODS_HIP_CHECK(hipMalloc(&c->view_dev, sizeof(float) * (size_t)c->max_w * c->max_h * c->batch));
  MODS_HIP_CHECK(hipMemcpyAsync(c->input_dev, src, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice, c->stream));
  int rc = launch_blur_xy_reflect(c, c->input_dev, c->tmp_dev, c->view_dev, w, h, kx, ky, sx, sy);
  if (rc) return rc;
  MODS_HIP_CHECK(hipMemcpyAsync(dst, c->view_dev, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

// n_src = 1 or 2 images of one size seen through the SAME view (both images of a pair have the same view schedule): one
// chain of launches for both - a view of a hard pair is ~100 launches of a few microseconds, so the count is the cost.
static int view_detect_describe(mods_ctx *c, const float *const *src_dev, int n_src, int w, int h, int stride, double tilt, double phi,
                                double zoom, double initSigma, int doBlur, const mods_hessaff_params *det, const mods_describe_params *desc,
                                mods_view_geom *geom_out, int *n_detected, int *n_regions) {
  if (!c || !src_dev || !det || !desc || n_src < 1 || n_src > c->batch) { set_error("detect_describe_view: bad argument"); return MODS_E_ARG; }
  for (int i = 0; i < n_src; i++) if (!src_dev[i]) { set_error("detect_describe_view: null image"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  mods_view_geom g;
  int rc = mods_view_geometry(w, h, tilt, phi, zoom, initSigma, &g);
  if (rc) return rc;
  if (geom_out) *geom_out = g;
  if ((size_t)g.w_new * g.h_new > (size_t)c->max_w * c->max_h) { set_error("view %dx%d larger than the context", g.w_new, g.h_new); return MODS_E_ARG; }
  if (g.w_new < 16 || g.h_new < 16) {   // nothing to detect on a sliver
    for (int i = 0; i < n_src; i++) { if (n_detected) n_detected[i] = 0; if (n_regions) n_regions[i] = 0; }
    MODS_HIP_CHECK(hipMemsetAsync(c->region_count, 0, sizeof(int) * n_src, c->stream));
    c->last_region_counts.assign(n_src, 0);
    c->last_inside_counts.assign(n_src, 0);
    return MODS_OK;
  }
  if (!c->view_dev) MODS_HIP_CHECK(hipMalloc(&c->view_dev, sizeof(float) * (size_t)c->max_w * c->max_h * c->batch));
  const size_t vpx = (size_t)g.w_new * g.h_new;
  for (int i = 0; i < n_src; i++)
    if ((rc = mods_synth_view_dev(c, src_dev[i], w, h, stride, &g, doBlur, c->view_dev + i * vpx))) return rc;
  // DetectAffineKeypoints / DetectMSERs scale regionsNumber by the SynthImage fields |tilt|, zoom (scale-space-detector.cpp:20-21, extrema.cpp:201-202)
  if ((rc = detect_any(c, c->view_dev, n_src, g.w_new, g.h_new, g.w_new, det, g.tilt, g.zoom))) return rc;
  if ((rc = describe_run_view(c, c->view_dev, n_src, g.w_new, g.h_new, desc, g.H, w, h, nullptr))) return rc;
  // every count and the error flag of the view in ONE round trip (pinned host words; three synchronisations before)
  int *hc = c->host_counts;
  MODS_HIP_CHECK(hipMemcpyAsync(hc, c->cand_count, sizeof(int) * 3 * c->batch, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 3 * c->batch, c->region_count, sizeof(int) * c->batch, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 4 * c->batch, c->inside_count, sizeof(int) * n_src, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(hipMemcpyAsync(hc + 5 * c->batch, c->desc_err_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  c->last_region_counts.assign(n_src, 0);
  c->last_inside_counts.assign(n_src, 0);
  for (int i = 0; i < n_src; i++) {
    if (hc[i] > c->max_cand) { set_error("NMS hit list overflow: %d > %d", hc[i], c->max_cand); return MODS_E_CAPACITY; }
    if (n_detected) n_detected[i] = hc[2 * c->batch + i];
    const int nr = hc[3 * c->batch + i];
    if (nr > (c->max_cand < (1 << 17) ? c->max_cand : (1 << 17))) { set_error("region list overflow: %d", nr); return MODS_E_CAPACITY; }
    if (n_regions) n_regions[i] = nr;
    c->last_region_counts[i] = nr;
    c->last_inside_counts[i] = hc[4 * c->batch + i];
  }
  if (hc[5 * c->batch]) {
    MODS_HIP_CHECK(hipMemsetAsync(c->desc_err_dev, 0, sizeof(int), c->stream));
    set_error("measurement region larger than the descriptor scratch");
    return MODS_E_CAPACITY;
  }
  return MODS_OK;
}

// One view of SynthDetectDescribeKeypoints (imagerepresentation.cpp:704-1099) for HessianAffine + RootSIFT:
// synthesise -> detect on the view -> centres (original frame) inside -> orientation on the view ->
// ReprojectRegions -> RootSIFT on the view.  Leaves the regions (reproj_kp + descriptor) in the context
// (mods_regions_fetch(ctx, 0, ...) / mods_regions_dev).  geom_out (optional) receives the geometry.
int mods_detect_describe_view_dev(mods_ctx *c, const float *src_dev, int w, int h, int stride, double tilt, double phi, double zoom,
                                  double initSigma, int doBlur, const mods_hessaff_params *det, const mods_describe_params *desc,
                                  mods_view_geom *geom_out, int *n_detected, int *n_regions) {
  const float *srcs[1] = {src_dev};
  return view_detect_describe(c, srcs, 1, w, h, stride, tilt, phi, zoom, initSigma, doBlur, det, desc, geom_out, n_detected, n_regions);
}

// The same view of two images of one size in one chain of launches (context batch >= 2): regions of image i in slot i.
int mods_detect_describe_view2_dev(mods_ctx *c, const float *src1_dev, const float *src2_dev, int w, int h, int stride, double tilt,
                                   double phi, double zoom, double initSigma, int doBlur, const mods_hessaff_params *det,
                                   const mods_describe_params *desc, mods_view_geom *geom_out, int *n_detected2, int *n_regions2) {
  const float *srcs[2] = {src1_dev, src2_dev};
  return view_detect_describe(c, srcs, 2, w, h, stride, tilt, phi, zoom, initSigma, doBlur, det, desc, geom_out, n_detected2, n_regions2);
}

int mods_view_fetch(mods_ctx *c, const mods_view_geom *g, float *dst_host) {
  if (!c || !g || !dst_host || !c->view_dev) { set_error("view_fetch: no view"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(dst_host, c->view_dev, sizeof(float) * (size_t)g->w_new * g->h_new, hipMemcpyDeviceToHost, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

int mods_regions_copy_dev(mods_ctx *c, int img, mods_region *dst_dev, int n) {
  if (!c || !dst_dev || img < 0 || img >= c->batch || n < 0 || n > c->max_cand) { set_error("regions_copy_dev: bad arguments"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(dst_dev, c->regions_dev + (size_t)img * c->max_cand, sizeof(mods_region) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

// the HalfRootSIFT twins of the regions of slot `img` (describe stage with halfDesc), same order
int mods_regions_half_copy_dev(mods_ctx *c, int img, mods_region *dst_dev, int n) {
  if (!c || !dst_dev || img < 0 || img >= c->batch || n < 0 || n > c->max_cand) { set_error("regions_half_copy_dev: bad arguments"); return MODS_E_ARG; }
  if (!c->have_half || !c->regions_half_dev) { set_error("regions_half_copy_dev: no HalfRootSIFT descriptors in the context"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipSetDevice(c->device));
  MODS_HIP_CHECK(hipMemcpyAsync(dst_dev, c->regions_half_dev + (size_t)img * c->max_cand, sizeof(mods_region) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  MODS_HIP_CHECK(mods::stream_wait(c->stream));
  return MODS_OK;
}

const mods_region *mods_regions_dev(mods_ctx *c, int img) { return c->regions_dev + (size_t)img * c->max_cand; }
const float *mods_view_pixels_dev(mods_ctx *c) { return c->view_dev; }

}  // extern "C"
