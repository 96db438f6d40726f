// ORACLE — TEST INFRASTRUCTURE ONLY.  Dominant orientation, patch extraction and
// SIFT / RootSIFT description restated from synth-detection.{hpp,cpp} and
// matching/siftdesc.cpp for the identity view (H = I, reproj_kp == det_kp).
#include "orc.h"
#include "detmath.h"
#include <algorithm>
#include <cmath>

namespace orc {

// synth-detection.cpp:21  const double k_sigma = 2 * 3.0 * sqrt(3.0)
static double k_sigma_synth() { return 2 * 3.0 * std::sqrt(3.0); }

// ReprojectRegionsAndRemoveTouchBoundary(..., dontRemove = true), synth-detection.cpp:151-190,
// as called at imagerepresentation.cpp:867: with H = I only the centre test remains.
void filter_centres_inside(std::vector<Region> &r, int w, int h) {
  std::vector<Region> keep;
  keep.reserve(r.size());
  for (size_t i = 0; i < r.size(); i++)
    if ((r[i].x < w) && (r[i].y < h) && (r[i].x > 0) && (r[i].y > 0)) keep.push_back(r[i]);
  r.swap(keep);
}

// ReprojectRegions, synth-detection.cpp:631-706 with H = I.
void filter_touch_boundary(std::vector<Region> &r, int w, int h) {
  const double ks = k_sigma_synth();
  std::vector<Region> keep;
  keep.reserve(r.size());
  for (size_t i = 0; i < r.size(); i++) {
    const Region &p = r[i];
    if ((p.x < w) && (p.y < h) && (p.x > 0) && (p.y > 0)) {
      if (!interpolate_check_borders(w, h, (float)p.x, (float)p.y, (float)p.a11, (float)p.a12, (float)p.a21,
                                     (float)p.a22, (int)(ks * p.s), (int)(ks * p.s)))
        keep.push_back(p);
    }
  }
  r.swap(keep);
}

// AffNet branch of the external affine adaptation, imagerepresentation.cpp:798-842: the network's (a11, a21, a22) per keypoint
// (a12 := 0), rectifyAffineTransformationUpIsUp (helpers.cpp:401-410), getEigenvalues (helpers.cpp:504-515) with the
// anisotropy limit 6, interpolateCheckBorders on the mrSize * s box.  a3: n x 3 floats.
void affnet_apply(std::vector<Region> &r, const float *a3, int w, int h, double mrSize) {
  std::vector<Region> keep;
  keep.reserve(r.size());
  for (size_t i = 0; i < r.size(); i++) {
    Region t = r[i];
    t.a11 = a3[3 * i]; t.a12 = 0; t.a21 = a3[3 * i + 1]; t.a22 = a3[3 * i + 2];
    {
      double a = t.a11, b = t.a12, c = t.a21, d = t.a22;
      double det = std::sqrt(std::fabs(a * d - b * c));
      double b2a2 = std::sqrt(b * b + a * a);
      t.a11 = b2a2 / det; t.a12 = 0; t.a21 = (d * b + c * a) / (b2a2 * det); t.a22 = det / b2a2;
    }
    float l1 = 1.0f, l2 = 1.0f;
    {
      const float a = (float)t.a11, b = (float)t.a12, c = (float)t.a21, d = (float)t.a22;
      float trace = a + d;
      float delta1 = (trace * trace - 4 * (a * d - b * c));
      if (delta1 < 0) continue;
      float delta = std::sqrt(delta1);
      l1 = (trace + delta) / 2.0f;
      l2 = (trace - delta) / 2.0f;
    }
    if ((l1 / l2 > 6) || (l2 / l1 > 6)) continue;
    if (interpolate_check_borders(w, h, (float)t.x, (float)t.y, (float)t.a11, (float)t.a12, (float)t.a21, (float)t.a22,
                                  (int)(mrSize * t.s), (int)(mrSize * t.s)))
      continue;
    keep.push_back(t);
  }
  r.swap(keep);
}

// OriNet branch of the orientation estimate, imagerepresentation.cpp:877-899: angle = atan2(y, x) of the network's two values
// per keypoint (float arguments: the float overload, host libm), frame rotated as DetectOrientation does.  yx: n x 2 floats.
void orinet_apply(std::vector<Region> &r, const float *yx) {
  for (size_t i = 0; i < r.size(); i++) {
    const Region c = r[i];
    double angle = std::atan2(yx[2 * i], yx[2 * i + 1]);
    double ci, si;
    det_sincos(angle, &si, &ci);      // cos / sin by the fixed sequences of detmath.h (glibc's sincos() and cos()/sin() disagree
                                      // in the last bit for ~1e-3 of the arguments, and compilers choose between them freely)
    r[i].a11 = c.a11 * ci - c.a12 * si;
    r[i].a12 = c.a11 * si + c.a12 * ci;
    r[i].a21 = c.a21 * ci - c.a22 * si;
    r[i].a22 = c.a21 * si + c.a22 * ci;
  }
}

// cv::invert(H, Hinv, DECOMP_LU) for a 3x3 CV_64F matrix: OpenCV's closed form (determinant by the first
// row, cofactors times 1/det; all zeros when det == 0).  Parity unpinned, benign.
void invert3(const double *S, double *t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.) { for (int i = 0; i < 9; i++) t[i] = 0; return; }
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// HIsEye, synth-detection.cpp:144-149 (eps1 = 0.01, :22)
bool h_is_eye(const double *H) {
  return (std::fabs(H[0] - 1.0) + std::fabs(H[1]) + std::fabs(H[2]) + std::fabs(H[3]) + std::fabs(H[4] - 1.0) + std::fabs(H[5]) +
          std::fabs(H[6]) + std::fabs(H[7]) + std::fabs(H[8] - 1.0) < 0.01);
}

// ReprojectByH, synth-detection.cpp:578-587: centre and frame through the affine part of Hinv; s is kept
static Region reproject_by_h(const Region &in, const double *H) {
  Region o = in;
  o.x = (H[0] * in.x + H[1] * in.y + H[2]);
  o.y = (H[3] * in.x + H[4] * in.y + H[5]);
  o.a11 = (H[0] * in.a11 + H[1] * in.a21);
  o.a12 = (H[0] * in.a12 + H[1] * in.a22);
  o.a21 = (H[3] * in.a11 + H[4] * in.a21);
  o.a22 = (H[3] * in.a12 + H[4] * in.a22);
  return o;
}

// ReprojectRegionsAndRemoveTouchBoundary(kps, H, orig_w, orig_h, mrSize, dontRemove = true),
// synth-detection.cpp:151-190, for a synthesised view: H maps original -> view; a region stays when its
// centre, taken back to the original image, lies strictly inside it.
void filter_centres_inside_view(std::vector<Region> &det, const double *H, int orig_w, int orig_h) {
  double Hinv[9];
  invert3(H, Hinv);
  const bool eye = h_is_eye(H);
  std::vector<Region> keep;
  keep.reserve(det.size());
  for (size_t i = 0; i < det.size(); i++) {
    const Region rp = eye ? det[i] : reproject_by_h(det[i], Hinv);
    if ((rp.x < orig_w) && (rp.y < orig_h) && (rp.x > 0) && (rp.y > 0)) keep.push_back(det[i]);
  }
  det.swap(keep);
}

// ReprojectRegions(kps, H, orig_w, orig_h), synth-detection.cpp:631-706: det (view frame) and rep (original
// frame) of the regions whose reprojected measurement box stays inside the original image.
void reproject_regions_view(std::vector<Region> &det, std::vector<Region> &rep, const double *H, int orig_w, int orig_h) {
  const double ks = k_sigma_synth();
  double Hinv[9];
  invert3(H, Hinv);
  const bool eye = h_is_eye(H);
  std::vector<Region> kd, kr;
  kd.reserve(det.size()); kr.reserve(det.size());
  for (size_t i = 0; i < det.size(); i++) {
    const Region p = eye ? det[i] : reproject_by_h(det[i], Hinv);
    if ((p.x < orig_w) && (p.y < orig_h) && (p.x > 0) && (p.y > 0)) {
      if (!interpolate_check_borders(orig_w, orig_h, (float)p.x, (float)p.y, (float)p.a11, (float)p.a12, (float)p.a21,
                                     (float)p.a22, (int)(ks * p.s), (int)(ks * p.s))) {
        kd.push_back(det[i]);
        kr.push_back(p);
      }
    }
  }
  det.swap(kd);
  rep.swap(kr);
}

// smoothCircularBuffer<36>, synth-detection.cpp:811-822
static void smooth_circular(float *hist, int bins) {
  float first = hist[0], prev = hist[bins - 1];
  for (int i = 0; i < bins - 1; i++) {
    float cur = hist[i];
    hist[i] = prev + cur + hist[i + 1];
    prev = cur;
  }
  hist[bins - 1] = prev + hist[bins - 1] + first;
}

// EstimateDominantAnglesFunctor::operator() (synth-detection.cpp:836-929): the local maxima >= max_th * max in bin order
// 0..35, with parabolic refinement; the first maxAngles of them are returned (the reference sorts a copy of the peak values
// and never uses it: the order stays the bin order; maxAngles = -1: all).
// half = doHalfSIFT: after the threshold has been taken from the full histogram, bins i and i + 18 are added into
// bin i (orientation modulo pi) and the upper half is cleared (:891-898).
int dominant_angles(const Img &img, double max_th, int maxAngles, std::vector<float> &angles, bool half) {
  angles.clear();
  if (maxAngles == 0) return 0;
  const int pS = img.w;
  const int bins = 36;
  const float PIf = float(M_PI);
  static thread_local Img orimask;
  if (orimask.w != pS) { orimask = Img(pS, pS); compute_circular_gauss_mask(orimask, pS / 3.0f); }
  Img gmag(pS, pS), gori(pS, pS);
  // computeGradientMagnitudeAndOrientation, helpers.cpp:840-862 (interior only)
  for (int r = 1; r < pS - 1; ++r)
    for (int c = 1; c < pS - 1; ++c) {
      float xgrad = img.at(r, c + 1) - img.at(r, c - 1);
      float ygrad = img.at(r + 1, c) - img.at(r - 1, c);
      gmag.at(r, c) = std::sqrt(xgrad * xgrad + ygrad * ygrad);
      gori.at(r, c) = atan2_lut_ff(ygrad, xgrad);
    }
  float hist[bins + 1];
  for (int i = 0; i < bins; i++) hist[i] = 0.0f;
  hist[bins] = 0.0f;   // the reference leaves hist[36] uninitialised; it is write-only
  const float *maskptr = orimask.row(1);
  const float *pmag = gmag.row(1), *pori = gori.row(1);
  const int maskPixels = pS * (pS - 2);
  for (int i = 0; i < maskPixels; ++i) {
    if (maskptr[i] > 0 && pmag[i] > 1.0) {
      int bin = (int)(bins * (pori[i] / PIf + 1.0f) / 2.0f);
      hist[bin] += pmag[i] * maskptr[i];
    }
  }
  for (int i = 0; i < 6; i++) smooth_circular(hist, bins);
  float thresh = 0.0;
  for (int i = 0; i < bins; i++)
    if (hist[i] > thresh) thresh = hist[i];
  thresh = (float)(thresh * max_th);
  if (half) {
    const int halfbins = bins / 2;
    for (int i = 0; i < halfbins; i++) { hist[i] += hist[i + halfbins]; hist[i + halfbins] = 0; }
  }
  if (maxAngles < 0) maxAngles = 100000000;
  for (int k = 0; k < bins && (int)angles.size() < maxAngles; k++) {
    int b = k, a = (k == 0) ? bins - 1 : k - 1, c = (k == bins - 1) ? 0 : k + 1;
    if (hist[b] >= thresh && hist[b] > hist[a] && hist[b] > hist[c]) {
      float pp = (hist[a] - hist[c]) / (hist[a] - 2.0f * hist[b] + hist[c]) / 2.0f;
      angles.push_back(2.0f * PIf * (b + 0.5f + pp) / bins - PIf);
    }
  }
  return (int)angles.size();
}
bool dominant_angle(const Img &img, double max_th, float *angle_out, bool half) {   // maxAngles = 1
  std::vector<float> a;
  if (!dominant_angles(img, max_th, 1, a, half)) return false;
  *angle_out = a[0];
  return true;
}

// DetectOrientation, synth-detection.cpp:1039-1149: up to maxAngles oriented copies per keypoint, in the order of the angles.
int detect_orientation(const std::vector<Region> &in, std::vector<Region> &out, const Img &img, double mrSize,
                       int patchSize, int maxAngles, double th, bool half, bool add_upright) {
  const double ks = k_sigma_synth();
  std::vector<Region> tmp;
  tmp.reserve(in.size());
  const int patchImageSize = 2 * int(mrSize) + 1;
  const double imageToPatchScale = double(patchImageSize) / (double)patchSize;
  std::vector<std::vector<Region>> slot(in.size());
  std::vector<char> ok(in.size(), 0);
#pragma omp parallel num_threads(g_threads)
  {
  Img patch(patchSize, patchSize);
#pragma omp for schedule(dynamic, 16)
  for (long i = 0; i < (long)in.size(); i++) {
    const Region &k = in[i];
    float curr_sc = (float)(imageToPatchScale * k.s);
    if (interpolate_check_borders(img.w, img.h, (float)k.x, (float)k.y, (float)k.a11, (float)k.a12, (float)k.a21,
                                  (float)k.a22, (int)(ks * k.s), (int)(ks * k.s)))
      continue;
    if (maxAngles > 0) {
      interpolate(img, (float)k.x, (float)k.y, (float)k.a11 * curr_sc, (float)k.a12 * curr_sc,
                  (float)k.a21 * curr_sc, (float)k.a22 * curr_sc, patch);
      std::vector<float> angs;
      dominant_angles(patch, th, maxAngles, angs, half);
      for (float ang : angs) {
        double si, ci;
        det_sincos(-(double)ang, &si, &ci);
        Region t = k;
        t.a11 = k.a11 * ci - k.a12 * si;
        t.a12 = k.a11 * si + k.a12 * ci;
        t.a21 = k.a21 * ci - k.a22 * si;
        t.a22 = k.a21 * si + k.a22 * ci;
        t.parent = (int)i;
        slot[i].push_back(t); ok[i] |= 1;
      }
    }
    if (add_upright) ok[i] |= 2;     // addUpRight (:1140-1142): the unrotated region itself, after its oriented copy
  }
  }   // omp parallel
  for (size_t i = 0; i < in.size(); i++) {
    if (ok[i] & 1) tmp.insert(tmp.end(), slot[i].begin(), slot[i].end());
    if (ok[i] & 2) { Region t = in[i]; t.parent = (int)i; tmp.push_back(t); }
  }
  out.swap(tmp);
  return (int)out.size();
}

// ---------------------------------------------------------------------------------------
// SIFT (matching/siftdesc.cpp), patchSize 41, spatialBins 4, orientationBins 8, maxBinValue 0.2
// ---------------------------------------------------------------------------------------
struct SiftTables {
  int ps;
  Img mask;
  std::vector<int> bin0, bin1;
  std::vector<double> w0, w1;
  explicit SiftTables(int patchSize) : ps(patchSize), mask(patchSize, patchSize) {
    compute_circular_gauss_mask(mask, 0);
    // precomputeBinsAndWeights, siftdesc.cpp:22-71
    const int spatialBins = 4, orientationBins = 8;
    int halfSize = ps >> 1;
    float step = float(spatialBins + 1) / (2 * halfSize);
    bin0.resize(ps); bin1.resize(ps); w0.resize(ps); w1.resize(ps);
    for (int i = 0; i < ps; i++) {
      float x = step * i;
      int xi = (int)(x);
      bin0[i] = xi - 1;
      bin1[i] = xi;
      w1[i] = x - xi;
      w0[i] = 1.0f - w1[i];
      if (bin0[i] < 0) { bin0[i] = 0; w0[i] = 0; }
      if (bin0[i] >= spatialBins) { bin0[i] = spatialBins - 1; w0[i] = 0; }
      if (bin1[i] < 0) { bin1[i] = 0; w1[i] = 0; }
      if (bin1[i] >= spatialBins) { bin1[i] = spatialBins - 1; w1[i] = 0; }
      bin0[i] *= orientationBins;
      bin1[i] *= orientationBins;
    }
  }
};

static double normalize_d(std::vector<double> &v) {   // siftdesc.cpp:133-158 (size % 4 == 0)
  double len = 0.0;
  for (size_t i = 0; i < v.size(); i += 4) {
    const double sq0 = v[i] * v[i], sq1 = v[i + 1] * v[i + 1], sq2 = v[i + 2] * v[i + 2], sq3 = v[i + 3] * v[i + 3];
    len += sq0 + sq1 + sq2 + sq3;
  }
  len = std::sqrt(len);
  const double fac = 1.0 / len;
  for (size_t i = 0; i < v.size(); i++) v[i] *= fac;
  return len;
}

// computeRootSiftDescriptor / computeSiftDescriptor + (Root)SIFTnorm(double),
// siftdesc.cpp:346-400, 288-345, 73-131, 199-222, 248-263.
// half = doHalfSIFT (siftdesc.cpp:401-436): the raw histogram (no normalisation) is folded, half[i*4 + j] =
// vec[i*8 + j] + vec[i*8 + j + 4], and the 64 values go through the same normalisation; out[64..127] = 0.
void sift_patch_to_desc(const Img &patch, uint8_t out[128], bool rootsift, double maxBinValue, bool half) {
  const int ps = patch.w;
  static thread_local SiftTables *T = nullptr;
  if (!T || T->ps != ps) { delete T; T = new SiftTables(ps); }
  const int spatialBins = 4, orientationBins = 8;
  Img grad(ps, ps), ori(ps, ps);
  for (int r = 0; r < ps; ++r)
    for (int c = 0; c < ps; ++c) {
      float xgrad, ygrad;
      if (c == 0) xgrad = patch.at(r, c + 1) - patch.at(r, c);
      else if (c == ps - 1) xgrad = patch.at(r, c) - patch.at(r, c - 1);
      else xgrad = patch.at(r, c + 1) - patch.at(r, c - 1);
      if (r == 0) ygrad = patch.at(r + 1, c) - patch.at(r, c);
      else if (r == ps - 1) ygrad = patch.at(r, c) - patch.at(r - 1, c);
      else ygrad = patch.at(r + 1, c) - patch.at(r - 1, c);
      grad.at(r, c) = std::sqrt(xgrad * xgrad + ygrad * ygrad);
      ori.at(r, c) = atan2_lut_ff(ygrad, xgrad);
    }
  std::vector<double> vec(spatialBins * spatialBins * orientationBins, 0.0);
  // samplePatch, siftdesc.cpp:73-131 (magnLess = false)
  const double M_PI_DOUBLED = 6.28318530718;
  for (int r = 0; r < ps; ++r) {
    const int br0 = spatialBins * T->bin0[r];
    const float wr0 = (float)T->w0[r];
    const int br1 = spatialBins * T->bin1[r];
    const float wr1 = (float)T->w1[r];
    for (int c = 0; c < ps; ++c) {
      float val = (float)(float(false) * 1.0 + (1.0 - float(false)) * T->mask.at(r, c) * grad.at(r, c));
      const int bc0 = T->bin0[c];
      const float wc0 = (float)(T->w0[c] * val);
      const int bc1 = T->bin1[c];
      const float wc1 = (float)(T->w1[c] * val);
      const float o = (float)(float(orientationBins) * (ori.at(r, c) + M_PI_DOUBLED) / M_PI_DOUBLED);
      int bo0 = (int)o;
      const float wo1 = o - bo0;
      bo0 %= orientationBins;
      int bo1 = (bo0 + 1) % orientationBins;
      const float wo0 = 1.0f - wo1;
      val = wr0 * wc0;
      if (val > 0) { vec[br0 + bc0 + bo0] += val * wo0; vec[br0 + bc0 + bo1] += val * wo1; }
      val = wr0 * wc1;
      if (val > 0) { vec[br0 + bc1 + bo0] += val * wo0; vec[br0 + bc1 + bo1] += val * wo1; }
      val = wr1 * wc0;
      if (val > 0) { vec[br1 + bc0 + bo0] += val * wo0; vec[br1 + bc0 + bo1] += val * wo1; }
      val = wr1 * wc1;
      if (val > 0) { vec[br1 + bc1 + bo0] += val * wo0; vec[br1 + bc1 + bo1] += val * wo1; }
    }
  }
  if (half) {
    std::vector<double> hv(spatialBins * spatialBins * orientationBins / 2);
    int bin1 = 0;
    for (int i = 0; i < spatialBins * spatialBins; i++)
      for (int j = 0; j < orientationBins / 2; j++) hv[bin1++] = vec[i * orientationBins + j] + vec[i * orientationBins + j + orientationBins / 2];
    vec.swap(hv);
    std::memset(out, 0, 128);
  }
  normalize_d(vec);
  bool changed = false;
  for (size_t i = 0; i < vec.size(); i++)
    if (vec[i] > maxBinValue) { vec[i] = maxBinValue; changed = true; }
  if (changed) normalize_d(vec);
  if (rootsift) {
    double sum = 0.;
    for (size_t i = 0; i < vec.size(); i++) sum += std::fabs(vec[i]);
    for (size_t i = 0; i < vec.size(); i++) vec[i] = std::sqrt(vec[i] / sum);
    for (size_t i = 0; i < vec.size(); i++) {
      int b = std::max(0, std::min((int)(512.0 * vec[i] + 0.5), 255));
      out[i] = (uint8_t)b;
    }
  } else {
    for (size_t i = 0; i < vec.size(); i++) {
      int b = std::max(0, std::min((int)(512.0f * vec[i] + 0.5), 255));
      out[i] = (uint8_t)b;
    }
  }
}

// Non-fast branch of DescribeRegions<>, synth-detection.hpp:186-231; with column_rule the same code as it appears in
// ExtractPatchesColumn, synth-detection.cpp:52-102 (the sampled region is 2*ceil(s*mr) wide for even patch sizes).
void extract_desc_patch(const Region &k, const Img &img, double mrSize, int patchSize, bool photoNorm, Img &patch, bool column_rule, bool fast) {
  static thread_local Img mask;
  if (mask.w != patchSize) { mask = Img(patchSize, patchSize); compute_circular_gauss_mask(mask, 0); }
  if (patch.w != patchSize || patch.h != patchSize) patch = Img(patchSize, patchSize);
  if (fast) {   // the fast_extraction branch of DescribeRegions<>, synth-detection.hpp:232-253
    double mrScale = (double)mrSize * k.s;
    int patchImageSize = 2 * int(mrScale) + 1;
    double imageToPatchScale = double(patchImageSize) / (double)patchSize;
    float curr_sc = imageToPatchScale;
    interpolate(img, (float)k.x, (float)k.y, (float)k.a11 * curr_sc, (float)k.a12 * curr_sc, (float)k.a21 * curr_sc,
                (float)k.a22 * curr_sc, patch);
    if (photoNorm) {
      float mean, var;
      photometrically_normalize(patch, mask, mean, var);
    }
    return;
  }
  float mrScale = (float)std::ceil(k.s * mrSize);
  int patchImageSize = (!column_rule || patchSize % 2 != 0) ? 2 * int(mrScale) + 1 : 2 * int(mrScale);
  float imageToPatchScale = float(patchImageSize) / float(patchSize);
  if (imageToPatchScale > 0.4) {
    patchImageSize += 2;
    Img smoothed(patchImageSize, patchImageSize);
    interpolate(img, (float)k.x, (float)k.y, (float)k.a11, (float)k.a12, (float)k.a21, (float)k.a22, smoothed);
    gauss_blur(smoothed, smoothed, 1.5f * imageToPatchScale);
    interpolate(smoothed, (float)(patchImageSize >> 1), (float)(patchImageSize >> 1), imageToPatchScale, 0, 0,
                imageToPatchScale, patch);
  } else {
    interpolate(img, (float)k.x, (float)k.y, (float)k.a11 * imageToPatchScale, (float)k.a12 * imageToPatchScale,
                (float)k.a21 * imageToPatchScale, (float)k.a22 * imageToPatchScale, patch);
  }
  if (photoNorm) {
    float mean, var;
    photometrically_normalize(patch, mask, mean, var);
  }
}

void describe_rootsift(std::vector<Region> &r, const Img &img, double mrSize, int patchSize, bool photoNorm, bool half, bool fast) {
#pragma omp parallel num_threads(g_threads)
  {
    Img patch(patchSize, patchSize);
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < (long)r.size(); i++) {
      extract_desc_patch(r[i], img, mrSize, patchSize, photoNorm, patch, false, fast);
      sift_patch_to_desc(patch, r[i].desc, true, 0.2, half);   // [SIFTDescriptor] maxBinValue = 0.2 (io_mods.cpp:427)
    }
  }
}

}  // namespace orc
