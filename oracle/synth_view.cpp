// ORACLE — TEST INFRASTRUCTURE ONLY.  View synthesis of the reference restated on a plain float
// image: GenerateSynthImageCorr, synth-detection.cpp:324-518 (rotate -> anisotropic anti-aliasing
// blur -> tilt/zoom warp), and the two OpenCV calls it is made of.
//
// Parity status: "parity unpinned" for the OpenCV parts (no OpenCV in the image, the reference pins no
// version).  cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT) is restated as OpenCV's fixed-point path:
// the inverse map is evaluated in 1/1024 px integers, rounded to 1/32 px, and the four taps are
// combined with the 32x32 table of bilinear weights in fp32; cv::GaussianBlur(ksize 3 or 5,
// BORDER_REFLECT_101) as sepFilter2D with OpenCV's small symmetric row/column filter orders.
// The view geometry (sizes, H, warp matrices, sigmas) is plain double arithmetic + libm and follows
// the reference line by line.
#include "orc.h"
#include "detmath.h"
#include <cmath>

namespace orc {

static int cv_round(double v) {   // saturate_cast<int>(double) = lrint: round half to even
  double fl = std::floor(v);
  double diff = v - fl;
  if (diff > 0.5) return (int)fl + 1;
  if (diff < 0.5) return (int)fl;
  return (((long long)fl) & 1LL) ? (int)fl + 1 : (int)fl;
}

// cv::warpAffine(src, dst, M, Size(dw,dh), INTER_LINEAR, BORDER_CONSTANT, Scalar(cval)), M maps src -> dst
void warp_affine(const Img &src, const double Mfwd[6], int dw, int dh, float cval, Img &dst) {
  double M[6];
  for (int i = 0; i < 6; i++) M[i] = Mfwd[i];
  {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
  }
  const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 1 << INTER_BITS;
  const int round_delta = AB_SCALE / TAB / 2;
  Img out(dw, dh);
  std::vector<int> adelta(dw), bdelta(dw);
  for (int x = 0; x < dw; x++) {
    adelta[x] = cv_round(M[0] * x * AB_SCALE);
    bdelta[x] = cv_round(M[3] * x * AB_SCALE);
  }
  const int sw = src.w, sh = src.h;
  for (int y = 0; y < dh; y++) {
    const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    float *D = out.row(y);
    for (int x = 0; x < dw; x++) {
      const int X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);
      const int Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS);
      const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
      const float fx = (float)(X & (TAB - 1)) * (1.f / TAB), fy = (float)(Y & (TAB - 1)) * (1.f / TAB);
      const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
      if ((unsigned)sx < (unsigned)(sw - 1 > 0 ? sw - 1 : 0) && (unsigned)sy < (unsigned)(sh - 1 > 0 ? sh - 1 : 0)) {
        const float *S = src.row(sy) + sx;
        D[x] = S[0] * w0 + S[1] * w1 + S[sw] * w2 + S[sw + 1] * w3;
      } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        D[x] = cval;
      } else {
        const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
        const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
        const float v0 = x0 && y0 ? src.at(sy, sx) : cval;
        const float v1 = x1 && y0 ? src.at(sy, sx + 1) : cval;
        const float v2 = x0 && y1 ? src.at(sy + 1, sx) : cval;
        const float v3 = x1 && y1 ? src.at(sy + 1, sx + 1) : cval;
        D[x] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
      }
    }
  }
  dst = out;
}

static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

// cv::GaussianBlur(img, img, Size(kx,ky), sx, sy) with the default BORDER_REFLECT_101, kx, ky in {3, 5}
// (synth-detection.cpp:499), fused like gauss_blur() in image_ops.cpp (an FMA build of OpenCV).  Row pass:
// SymmRowSmallFilter  s = S[0]*k0; s = fma(S[-1]+S[1], k1, s) [; s = fma(S[-2]+S[2], k2, s)];
// column pass: ksize 3 -> SymmColumnSmallFilter  s = fma(S0+S2, f1, S1*f0);  ksize 5 -> SymmColumnFilter
// s = f0*S[0]; s = fma(f1, S[1]+S[-1], s); s = fma(f2, S[2]+S[-2], s).  Larger (odd) sizes follow the generic
// orders used by gauss_blur().
void gauss_blur_xy(const Img &src, Img &dst, int kx, int ky, double sx, double sy) {
  const std::vector<float> kxv = gauss_kernel(kx, sx), kyv = gauss_kernel(ky, sy);
  const int w = src.w, h = src.h, rx = kx / 2, ry = ky / 2;
  Img tmp(w, h);
  for (int y = 0; y < h; y++) {
    const float *S = src.row(y);
    float *T = tmp.row(y);
    for (int x = 0; x < w; x++) {
      float s;
      if (kx <= 5) {
        s = S[x] * kxv[rx];
        for (int j = 1; j <= rx; j++) s = std::fmaf(S[reflect101(x - j, w)] + S[reflect101(x + j, w)], kxv[rx + j], s);
      } else {
        s = kxv[0] * S[reflect101(x - rx, w)];
        for (int j = 1; j < kx; j++) s = std::fmaf(kxv[j], S[reflect101(x - rx + j, w)], s);
      }
      T[x] = s;
    }
  }
  Img out(w, h);
  for (int y = 0; y < h; y++) {
    float *D = out.row(y);
    for (int x = 0; x < w; x++) {
      float s;
      if (ky == 3) {
        s = std::fmaf(tmp.at(reflect101(y - 1, h), x) + tmp.at(reflect101(y + 1, h), x), kyv[2], tmp.at(y, x) * kyv[1]);
      } else {
        s = kyv[ry] * tmp.at(y, x);
        for (int j = 1; j <= ry; j++) s = std::fmaf(kyv[ry + j], tmp.at(reflect101(y + j, h), x) + tmp.at(reflect101(y - j, h), x), s);
      }
      D[x] = s;
    }
  }
  dst = out;
}

// Geometry of one synthesised view, synth-detection.cpp:336-469: everything GenerateSynthImageCorr
// derives from (w, h, tilt, phi, zoom, InitSigma) before it touches pixels.
bool view_geometry(int w, int h, double tilt, double phi, double zoom, double InitSigma, ViewGeom *g) {
  int zoomed = 0;
  bool vertical_tilt = false;
  if (tilt < 0) { tilt = -tilt; vertical_tilt = true; }
  if (std::fabs(zoom - 1.0f) >= 0.05) zoomed = 1;
  const int wS1 = (int)(w * zoom), hS1 = (int)(h * zoom);
  g->identity = (std::fabs(tilt - 1.) <= 0.1) && (std::fabs(phi) <= 0.2) && (std::fabs(zoom - 1.) <= 0.1);
  g->rotation = phi * 180 / M_PI; g->tilt = tilt; g->zoom = zoom;
  if (g->identity) {
    g->rotation = 0; g->tilt = 1; g->zoom = 1;
    for (int i = 0; i < 9; i++) g->H[i] = (i % 4 == 0) ? 1.0 : 0.0;
    g->w_new = w; g->h_new = h; g->w_rot = w; g->h_rot = h;
    return true;
  }
  double d, d2, w_new, h_new;
  // cos(phi) / sin(phi): one evaluation each by the fixed sequences of detmath.h (the reference calls libm per use; glibc's
  // sincos() and cos()/sin() differ in the last bit for ~1e-3 of the arguments and compilers pick between them freely, so a
  // contract on top of libm cannot be bit exact)
  double cp, sp;
  det_sincos(phi, &sp, &cp);
  double kV = 1., kH = 1.;
  if (zoomed) { kV = (double)w / (double)wS1; kH = (double)h / (double)hS1; }
  double *H = g->H;
  const bool first = (phi >= 0) && (phi < M_PI / 2);
  if (vertical_tilt) {
    if (first) {
      w_new = std::floor((0.5 + cp * w + sp * h) / (kH));
      h_new = std::floor((0.5 + sp * w + cp * h) / (tilt * kV));
      H[0] = cp / kH; H[1] = sp / kH; H[2] = 0;
      H[3] = -sp / (tilt * kV); H[4] = cp / (tilt * kV); H[5] = std::floor(0.5 + sp * w / (tilt * kV));
    } else {
      w_new = std::floor((0.5 - cp * w + sp * h) / (kH));
      h_new = std::floor((0.5 + sp * w - cp * h) / (tilt * kV));
      d = -std::floor(cp * w / kH);
      d2 = std::floor(0.5 + (sp * w - cp * h) / (tilt * kV));
      H[0] = cp / kH; H[1] = sp / kH; H[2] = d;
      H[3] = -sp / (tilt * kV); H[4] = cp / (tilt * kV); H[5] = d2;
    }
  } else {
    if (first) {
      w_new = std::floor((0.5 + cp * w + sp * h) / (tilt * kH));
      h_new = std::floor((0.5 + sp * w + cp * h) / (kV));
      H[0] = cp / (tilt * kH); H[1] = sp / (tilt * kH); H[2] = 0;
      H[3] = -sp / kV; H[4] = cp / kV; H[5] = std::floor(0.5 + sp * w / kV);
    } else {
      w_new = std::floor((0.5 - cp * w + sp * h) / (tilt * kH));
      h_new = std::floor((0.5 + sp * w - cp * h) / (kV));
      d = -std::floor(cp * w / (tilt * kH));
      d2 = std::floor(0.5 + (sp * w - cp * h) / kV);
      H[0] = cp / (tilt * kH); H[1] = sp / (tilt * kH); H[2] = d;
      H[3] = -sp / kV; H[4] = cp / kV; H[5] = d2;
    }
  }
  H[6] = 0; H[7] = 0; H[8] = 1;
  g->w_new = (int)w_new; g->h_new = (int)h_new;      // cv::Size(w_new, h_new): double -> int
  // anti-aliasing
  const double sigma_aa_2 = zoomed ? InitSigma / (4.0 * zoom) : InitSigma / 2.0;
  const double sigma_aa = InitSigma * tilt / (2.0 * zoom);
  if (vertical_tilt) { g->sigma_x = sigma_aa_2; g->sigma_y = sigma_aa; }
  else { g->sigma_x = sigma_aa; g->sigma_y = sigma_aa_2; }
  double *R = g->warpRot;
  if (first) {
    g->w_rot = (int)std::floor((0.5 + cp * w + sp * h));
    g->h_rot = (int)std::floor((0.5 + sp * w + cp * h));
    R[0] = cp; R[1] = sp; R[2] = 0;
    R[3] = -sp; R[4] = cp; R[5] = std::floor(0.5 + sp * w);
  } else {
    g->w_rot = (int)std::floor((0.5 - cp * w + sp * h));
    g->h_rot = (int)std::floor((0.5 + sp * w - cp * h));
    d = -std::floor(cp * w);
    d2 = std::floor(0.5 + (sp * w - cp * h));
    R[0] = cp; R[1] = sp; R[2] = d;
    R[3] = -sp; R[4] = cp; R[5] = d2;
  }
  int kx = (int)std::floor(2.0 * 3.0 * g->sigma_x + 1.0);
  if (kx % 2 == 0) kx++;
  if (kx < 3) kx = 3;
  int ky = (int)std::floor(2.0 * 3.0 * g->sigma_y + 1.0);
  if (ky % 2 == 0) ky++;
  if (ky < 3) ky = 3;
  g->ksize_x = kx; g->ksize_y = ky;
  double *T = g->warpTilt;
  if (vertical_tilt) { T[0] = 1.0 / kH; T[1] = 0; T[2] = 0; T[3] = 0; T[4] = 1.0 / (tilt * kV); T[5] = 0; }
  else { T[0] = 1.0 / (tilt * kH); T[1] = 0; T[2] = 0; T[3] = 0; T[4] = 1.0 / kV; T[5] = 0; }
  return true;
}

// GenerateSynthImageCorr on a grey float image (the non-AREA_INTERP build, synth-detection.cpp:471-517)
void generate_synth_view(const Img &in, double tilt, double phi, double zoom, double InitSigma, int doBlur, Img &out, ViewGeom *g) {
  view_geometry(in.w, in.h, tilt, phi, zoom, InitSigma, g);
  if (g->identity) { out = in; return; }
  Img temp;
  warp_affine(in, g->warpRot, g->w_rot, g->h_rot, 128.f, temp);
  if (doBlur) gauss_blur_xy(temp, temp, g->ksize_x, g->ksize_y, g->sigma_x, g->sigma_y);
  warp_affine(temp, g->warpTilt, g->w_new, g->h_new, 128.f, out);
}

}  // namespace orc
