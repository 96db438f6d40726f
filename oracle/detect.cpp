// ORACLE — TEST INFRASTRUCTURE ONLY.  Hessian scale-space pyramid, 3x3x3 NMS, sub-pixel
// localisation and Baumberg affine-shape iteration, restated from
// detectors/affinedetectors/{pyramid.h,pyramid.cpp,affine.cpp,scale-space-detector.hpp}.
#include "orc.h"
#include "detmath.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace orc {

// ScaleSpaceDetector::Response, pyramid.cpp:126-163.  Hessian and Harris have no iiDoG form (the branches are commented
// out and fall through to the end of a non-void function: undefined in the reference, rejected by the front ends here).
static void response(const Img &in, Img &out, float norm, const HessAffParams &p) {
  if (p.detector_type == 1) dog_response(in, out, norm, p.ii_dog != 0);
  else if (p.detector_type == 2) harris_response(in, out, norm);
  else hessian_response(in, out, norm);
}

// detectPyramidKeypoints + detectOctaveKeypoints (pyramid.cpp:496-529, 428-494), blur /
// response planes only.  upscaleInputImage = 0 (structures.hpp:137).
void build_pyramid(const Img &image, const HessAffParams &p, Pyramid &pyr) {
  pyr.oct.clear();
  float curSigma0 = 0.5f;
  float pixelDistance = 1.0f;
  Img first = image;
  if (p.initialSigma > curSigma0) {
    float sigma = std::sqrt(p.initialSigma * p.initialSigma - curSigma0 * curSigma0);
    gauss_blur(first, first, sigma);
  }
  const int minSize = 2 * p.border + 2;
  while (first.h > minSize && first.w > minSize) {
    Pyramid::Oct o;
    o.w = first.w; o.h = first.h; o.pixelDistance = pixelDistance;
    float sigmaStep = det_pow2f(1.0f / (float)p.numberOfScales);   // pow(2.0f, 1/nScales)
    float curSigma = p.initialSigma;
    o.blur.push_back(first);
    o.sigma.push_back(curSigma);
    Img r;
    response(first, r, curSigma * curSigma, p);
    o.resp.push_back(r);
    Img next;
    for (int i = 1; i < p.numberOfScales + 2; i++) {
      float sigma = curSigma * std::sqrt(sigmaStep * sigmaStep - 1.0f);
      Img nb;
      gauss_blur(o.blur.back(), nb, sigma);
      sigma = curSigma * sigmaStep;
      Img rr;
      response(nb, rr, sigma * sigma, p);
      if (i == p.numberOfScales) resize_half(nb, next);
      o.blur.push_back(nb);
      o.resp.push_back(rr);
      curSigma *= sigmaStep;
      o.sigma.push_back(curSigma);
    }
    pyr.oct.push_back(o);
    pixelDistance *= 2.0f;
    first = next;
  }
}

static bool is_max(float val, const Img &pix, int row, int col) {   // pyramid.cpp:41-51
  for (int r = row - 1; r <= row + 1; r++) {
    const float *p = pix.row(r);
    for (int c = col - 1; c <= col + 1; c++)
      if (p[c] > val) return false;
  }
  return true;
}
static bool is_min(float val, const Img &pix, int row, int col) {   // pyramid.cpp:53-63
  for (int r = row - 1; r <= row + 1; r++) {
    const float *p = pix.row(r);
    for (int c = col - 1; c <= col + 1; c++)
      if (p[c] < val) return false;
  }
  return true;
}

// ScaleSpaceDetector ctor thresholds, pyramid.h:46-66 (FIXED_TH, DET_HESSIAN): members are
// initialised in declaration order edgeScoreThreshold, finalThreshold, positiveThreshold,
// negativeThreshold => positive = 0.8 * threshold (un-squared), final = threshold^2.
struct Thresholds {
  double edgeScoreThreshold;
  float finalThreshold, positiveThreshold, negativeThreshold;
  explicit Thresholds(const HessAffParams &p) {
    edgeScoreThreshold = (p.edgeEigenValueRatio + 1.0f) * (p.edgeEigenValueRatio + 1.0f) / p.edgeEigenValueRatio;
    finalThreshold = p.threshold;
    positiveThreshold = (float)(0.8 * finalThreshold);
    negativeThreshold = -positiveThreshold;
    if (p.detector_type == 0) finalThreshold = p.threshold * p.threshold;   // pyramid.h:55-56: DET_HESSIAN only
    if (p.mode != 0) finalThreshold = positiveThreshold = negativeThreshold = 0.0f;   // pyramid.h:58-59: every mode but FIXED_TH
  }
};

// localizeKeypoint, pyramid.cpp:281-403.  Returns true and fills `out` when the point is
// accepted; octaveMap handling stays with the caller's processing order.
static bool localize(const Img &low, const Img &cur, const Img &high, const Img &blur,
                     std::vector<unsigned char> &octaveMap, int r, int c, float curScale,
                     float pixelDistance, const HessAffParams &p, const Thresholds &th, Candidate &out) {
  const int cols = cur.w, rows = cur.h;
  float b[3] = {0, 0, 0};
  float val = 0;
  int nr = r, nc = c;
  for (int iter = 0; iter < 5; iter++) {
    r = nr; c = nc;
    const float *cur0 = cur.row(r - 1), *cur1 = cur.row(r), *cur2 = cur.row(r + 1);
    const float *low0 = low.row(r - 1), *low1 = low.row(r), *low2 = low.row(r + 1);
    const float *high0 = high.row(r - 1), *high1 = high.row(r), *high2 = high.row(r + 1);
    float dxx = cur1[c - 1] - 2.0f * cur1[c] + cur1[c + 1];
    float dyy = cur0[c] - 2.0f * cur1[c] + cur2[c];
    float dss = low1[c] - 2.0f * cur1[c] + high1[c];
    float dxy = 0.25f * (cur2[c + 1] - cur2[c - 1] - cur0[c + 1] + cur0[c - 1]);
    if (0 == iter) {
      float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
      if (edgeScore >= th.edgeScoreThreshold || edgeScore < 0) return false;
    }
    float dxs = 0.25f * (high1[c + 1] - high1[c - 1] - low1[c + 1] + low1[c - 1]);
    float dys = 0.25f * (high2[c] - high0[c] - low2[c] + low0[c]);
    float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
    float dx = 0.5f * (cur1[c + 1] - cur1[c - 1]);
    float dy = 0.5f * (cur2[c] - cur0[c]);
    float ds = 0.5f * (high1[c] - low1[c]);
    b[0] = -dx; b[1] = -dy; b[2] = -ds;
    solve_linear_3x3(A, b);
    if (std::isnan(b[0]) || std::isnan(b[1]) || std::isnan(b[2])) return false;
    val = cur1[c] + 0.5f * (dx * b[0] + dy * b[1] + ds * b[2]);
    // MAX_SUBPIXEL_SHIFT 0.6 and POINT_SAFETY_BORDER 3 (pyramid.cpp:26,29); 0.6 is a double.
    if (b[0] > 0.6) { if (c < cols - 3) nc++; else return false; }
    if (b[1] > 0.6) { if (r < rows - 3) nr++; else return false; }
    if (b[0] < -0.6) { if (c > 3) nc--; else return false; }
    if (b[1] < -0.6) { if (r > 3) nr--; else return false; }
    if (nr == r && nc == c) break;
  }
  if (std::fabs(b[0]) > 1.5 || std::fabs(b[1]) > 1.5 || std::fabs(b[2]) > 1.5 ||
      std::fabs(val) < th.finalThreshold || octaveMap[(size_t)r * cols + c] > 0)
    return false;
  octaveMap[(size_t)r * cols + c] = 1;
  float scale = curScale * det_pow2f(b[2] / p.numberOfScales);
  // getPointType, pyramid.cpp:65-82 (HESSIAN_DARK 0, BRIGHT 1, SADDLE 2)
  int type;
  if (p.detector_type == 1) type = val < 0 ? 11 : 10;        // DOG_BRIGHT : DOG_DARK, pyramid.cpp:91-99
  else if (p.detector_type == 2) type = val < 0 ? 31 : 30;   // HARRIS_BRIGHT : HARRIS_DARK, :100-107
  else if (val < 0) type = 2;
  else {
    const float *ptr = blur.row(r) + c;
    float Lxx = (ptr[-1] - 2 * ptr[0] + ptr[1]);
    type = (Lxx < 0) ? 0 : 1;
  }
  out.r = r; out.c = c;
  out.x = pixelDistance * (c + b[0]);
  out.y = pixelDistance * (r + b[1]);
  out.s = pixelDistance * scale;
  out.pixelDistance = pixelDistance;
  out.type = type;
  out.response = val;
  return true;
}

// findLevelKeypoints (pyramid.cpp:405-425) for the three detection levels of every octave in
// the reference's processing order: octave, then level 1..S, then raster.
// nms_raw (optional) receives (octave, level, r, c) of every NMS hit before localisation.
void find_candidates(const Pyramid &pyr, const HessAffParams &p, std::vector<Candidate> &out,
                     std::vector<int> *nms_raw) {
  Thresholds th(p);
  out.clear();
  for (size_t oi = 0; oi < pyr.oct.size(); oi++) {
    const Pyramid::Oct &o = pyr.oct[oi];
    std::vector<unsigned char> octaveMap((size_t)o.w * o.h, 0);
    for (int lv = 1; lv <= p.numberOfScales; lv++) {
      const Img &low = o.resp[lv - 1], &cur = o.resp[lv], &high = o.resp[lv + 1];
      const Img &blur = o.blur[lv];
      const float curSigma = o.sigma[lv];
      for (int r = p.border; r < (o.h - p.border); r++) {
        const float *curPtr = cur.row(r);
        for (int c = p.border; c < (o.w - p.border); c++) {
          const float val = curPtr[c];
          if ((val > th.positiveThreshold && (is_max(val, cur, r, c) && is_max(val, low, r, c) && is_max(val, high, r, c))) ||
              (val < th.negativeThreshold && (is_min(val, cur, r, c) && is_min(val, low, r, c) && is_min(val, high, r, c)))) {
            if (nms_raw) { nms_raw->push_back((int)oi); nms_raw->push_back(lv); nms_raw->push_back(r); nms_raw->push_back(c); }
            Candidate cd;
            cd.octave = (int)oi; cd.level = lv; cd.r0 = r; cd.c0 = c;
            if (localize(low, cur, high, blur, octaveMap, r, c, curSigma, o.pixelDistance, p, th, cd))
              out.push_back(cd);
          }
        }
      }
    }
  }
}

// ---- the Hessian form of the Baumberg iteration (affBmbrgMethod = 1, affine.cpp:92-128) ------------------------------------
// PARITY UNPINNED beyond the reference's own lines: the step leans on cv::SVD::compute and on cv::Mat products of 2x2 float
// matrices, i.e. on OpenCV (4.x, absent here), restated below from its published sources: modules/core/src/lapack.cpp
// JacobiSVDImpl_<float> behind _SVDcompute (one-sided Jacobi on the rows of A^T, double accumulators, eps = 2 FLT_EPSILON,
// singular values sorted descending, U = normalised rows) and the len == 2 fast path of cv::gemm (matmul: fp32 products and sums,
// no accumulation in double).  A build of OpenCV that routes hal::SVD32f to LAPACK (sgesdd) may differ in the last bits and in
// the signs of U / V; no shipped configuration sets the key.
void svd2x2_f32(const float A[4], float d[2], float U[4], float Vt[4], bool *degenerate) {
  const int m = 2, n = 2;
  const float eps = FLT_EPSILON * 2;
  const double minval = FLT_MIN;
  float At[2][2] = {{A[0], A[2]}, {A[1], A[3]}};      // transpose(src, temp_a): row i = column i of A
  float V[2][2] = {{1, 0}, {0, 1}};
  double W[2];
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sd;
  }
  const int max_iter = 30;                              // std::max(m, 30)
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    {                                                   // the one pair (i, j) = (0, 1)
      float *Ai = At[0], *Aj = At[1];
      double a = W[0], p = 0, b = W[1];
      for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
      if (!(std::abs(p) <= eps * std::sqrt((double)a * b))) {
        p *= 2;
        const double beta = a - b, gamma = hypot((double)p, beta);
        float c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = (float)std::sqrt(delta / gamma);
          c = (float)(p / (gamma * s * 2));
        } else {
          c = (float)std::sqrt((gamma + beta) / (gamma * 2));
          s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const float t0 = c * Ai[k] + s * Aj[k];
          const float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[0] = a; W[1] = b;
        changed = true;
        for (int k = 0; k < n; k++) {
          const float t0 = c * V[0][k] + s * V[1][k];
          const float t1 = -s * V[0][k] + c * V[1][k];
          V[0][k] = t0; V[1][k] = t1;
        }
      }
    }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = std::sqrt(sd);
  }
  if (W[0] < W[1]) {
    std::swap(W[0], W[1]);
    for (int k = 0; k < m; k++) std::swap(At[0][k], At[1][k]);
    for (int k = 0; k < n; k++) std::swap(V[0][k], V[1][k]);
  }
  d[0] = (float)W[0]; d[1] = (float)W[1];
  // a singular value <= FLT_MIN: OpenCV builds the missing left vector from its random generator; findAffineShape then divides
  // by det = 0 and carries non-finite shapes into the next interpolate() (undefined there).  Reported, and the caller drops
  // the point.
  *degenerate = W[0] <= minval || W[1] <= minval;
  for (int i = 0; i < n; i++) {
    const float s = (float)(W[i] > minval ? 1 / W[i] : 0.);
    for (int k = 0; k < m; k++) At[i][k] *= s;
  }
  U[0] = At[0][0]; U[1] = At[1][0]; U[2] = At[0][1]; U[3] = At[1][1];      // transpose(temp_u, _u)
  Vt[0] = V[0][0]; Vt[1] = V[0][1]; Vt[2] = V[1][0]; Vt[3] = V[1][1];
}
// cv::gemm, 2x2 fp32 operands (the len == 2 path): every product and sum rounded to fp32
static void mul2x2_f32(const float a[4], const float b[4], float d[4]) {
  const float t0 = a[0] * b[0] + a[1] * b[2], t1 = a[0] * b[1] + a[1] * b[3];
  const float t2 = a[2] * b[0] + a[3] * b[2], t3 = a[2] * b[1] + a[3] * b[3];
  d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
}
// one step of affine.cpp:92-128: updates u, returns the two singular values' ratio term; false = degenerate Hessian
static bool hessian_shape_step(const Img &blur, float lx, float ly, float affRatio, float u[4], float *eigen_ratio_act) {
  Img h3(3, 3);
  interpolate(blur, lx, ly, u[0] * affRatio, u[1] * affRatio, u[2] * affRatio, u[3] * affRatio, h3);
  const float *q = h3.d.data();
  const float Dxx = (q[0] - 2.f * q[1] + q[2] + 2.f * q[3] - 4.f * q[4] + 2.f * q[5] + q[6] - 2.f * q[7] + q[8]);
  const float Dyy = (q[0] + 2.f * q[1] + q[2] - 2.f * q[3] - 4.f * q[4] - 2.f * q[5] + q[6] + 2.f * q[7] + q[8]);
  const float Dxy = (q[0] - q[2] - q[6] + q[8]);
  const float Au0[4] = {Dxx, Dxy, Dxy, Dyy};
  float d[2], U[4], Vt[4];
  bool degenerate;
  svd2x2_f32(Au0, d, U, Vt, &degenerate);
  if (degenerate) return false;
  float l1 = d[0], l2 = d[1];
  *eigen_ratio_act = (float)(1.0 - std::abs(l2) / std::abs(l1));
  const float det = std::sqrt(std::abs(l1 * l2));
  l2 = std::sqrt(std::sqrt(std::abs(l1) / det));
  l1 = (float)(1. / l2);
  const float D[4] = {l1, 0, 0, l2};
  float UD[4], Au[4], T[4], Ap[4];
  mul2x2_f32(U, D, UD);
  mul2x2_f32(UD, Vt, Au);        // Au = U * D * V (V is what SVD::compute returns as vt)
  mul2x2_f32(Au, u, T);
  mul2x2_f32(T, Au, Ap);         // Ap = Au * Ap * Au
  u[0] = Ap[0]; u[1] = Ap[1]; u[2] = Ap[2]; u[3] = Ap[3];
  return true;
}

// AffineShape::findAffineShape (affine.cpp:26-158): the SMM branch and, with p.aff_bmbrg_method = 1, the Hessian branch.
bool find_affine_shape(const Img &blur, float x, float y, float s, float pixelDistance,
                       const HessAffParams &p, const Img &mask, float a_out[4], int *iters) {
  float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;
  float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
  float lx = x / pixelDistance, ly = y / pixelDistance;
  float ratio = s / (p.initialSigma * pixelDistance);
  const int W = p.smmWindowSize;
  if (!p.doBaumberg) {
    a_out[0] = u11; a_out[1] = u12; a_out[2] = u21; a_out[3] = u22;
    if (iters) *iters = 0;
    return true;
  }
  const int maskPixels = W * W;
  Img img(W, W), fx(W, W), fy(W, W);
  if (p.aff_bmbrg_method == 1) {
    const float affRatio = s * p.aff_meas_region / pixelDistance;
    float u[4] = {u11, u12, u21, u22};
    for (int l = 0; l < p.maxIterations; l++) {
      eigen_ratio_bef = eigen_ratio_act;
      if (!hessian_shape_step(blur, lx, ly, affRatio, u, &eigen_ratio_act)) break;
      if (!get_eigenvalues(u[0], u[1], u[2], u[3], l1, l2)) break;
      if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
      if (eigen_ratio_act < p.convergenceThreshold && eigen_ratio_bef < p.convergenceThreshold) {
        a_out[0] = u[0]; a_out[1] = u[1]; a_out[2] = u[2]; a_out[3] = u[3];
        if (iters) *iters = l;
        return true;
      }
    }
    return false;
  }
  for (int l = 0; l < p.maxIterations; l++) {
    float a = 0, b = 0, c = 0;
    interpolate(blur, lx, ly, u11 * ratio, u12 * ratio, u21 * ratio, u22 * ratio, img);
    compute_gradient(img, fx, fy);
    const float *maskptr = mask.d.data();
    const float *pfx = fx.d.data(), *pfy = fy.d.data();
    for (int i = 0; i < maskPixels; ++i) {
      const float v = maskptr[i];
      const float gxx = pfx[i];
      const float gyy = pfy[i];
      const float gxy = gxx * gyy;
      a += gxx * gxx * v;
      b += gxy * v;
      c += gyy * gyy * v;
    }
    a /= maskPixels;
    b /= maskPixels;
    c /= maskPixels;
    inv_sqrt(a, b, c, l1, l2);
    if ((a != a) || (b != b) || (c != c)) break;
    eigen_ratio_bef = eigen_ratio_act;
    eigen_ratio_act = (float)(1.0 - l2 / l1);
    float u11t = u11, u12t = u12;
    u11 = a * u11t + b * u21;
    u12 = a * u12t + b * u22;
    u21 = b * u11t + c * u21;
    u22 = b * u12t + c * u22;
    if (!get_eigenvalues(u11, u12, u21, u22, l1, l2)) break;
    if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
    if (eigen_ratio_act < p.convergenceThreshold && eigen_ratio_bef < p.convergenceThreshold) {
      a_out[0] = u11; a_out[1] = u12; a_out[2] = u21; a_out[3] = u22;
      if (iters) *iters = l;
      return true;
    }
  }
  return false;
}

// synth-detection.cpp:134-143
void rectify_transformation(double &a11, double &a12, double &a21, double &a22) {
  double a = a11, b = a12, c = a21, d = a22;
  double det = std::sqrt(std::fabs(a * d - b * c));
  double b2a2 = std::sqrt(b * b + a * a);
  a11 = b2a2 / det;
  a12 = 0;
  a21 = (d * b + c * a) / (b2a2 * det);
  a22 = det / b2a2;
}

// DetectAffineKeypoints (scale-space-detector.cpp:13-32) -> onKeypointDetected ->
// findAffineShape on prevBlur (= blur level-1 of the detection level, pyramid.cpp:402,478) ->
// exportKeypoints: sort by |response| descending (scale-space-detector.hpp:120-131; std::sort is
// unstable on ties, fixed here as processing order) -> DetectAffineRegions
// (synth-detection.hpp:79-112): s *= sqrt|det A|, A -> lower-triangular det 1.
void detect_hessian_affine(const Img &image, const HessAffParams &p, std::vector<AffKey> &out) {
  if (p.detector_type == 3) { detect_mser(image, p, 1.0, 1.0, out); return; }   // DET_MSER (imagerepresentation.cpp:780-783)
  Pyramid pyr;
  build_pyramid(image, p, pyr);
  std::vector<Candidate> cand;
  find_candidates(pyr, p, cand);
  Img mask(p.smmWindowSize, p.smmWindowSize);
  compute_gauss_mask(mask);
  std::vector<AffKey> keys;
  std::vector<AffKey> slot(cand.size());
  std::vector<char> ok(cand.size(), 0);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 16)
  for (long i = 0; i < (long)cand.size(); i++) {
    const Candidate &cd = cand[i];
    const Img &prevBlur = pyr.oct[cd.octave].blur[cd.level - 1];
    float a[4]; int it;
    // onKeypointDetected, scale-space-detector.hpp:47-55: sampleFromImage -> findAffineShape(image, x, y, s, 1.0, ...)
    if (!(p.sample_from_image ? find_affine_shape(image, cd.x, cd.y, cd.s, 1.0f, p, mask, a, &it)
                              : find_affine_shape(prevBlur, cd.x, cd.y, cd.s, cd.pixelDistance, p, mask, a, &it))) continue;
    AffKey k;
    k.x = cd.x; k.y = cd.y; k.s = cd.s;
    k.a11 = a[0]; k.a12 = a[1]; k.a21 = a[2]; k.a22 = a[3];
    k.response = cd.response; k.sub_type = cd.type;
    k.octave = cd.octave; k.level = cd.level; k.r0 = cd.r0; k.c0 = cd.c0;
    slot[i] = k; ok[i] = 1;
  }
  for (size_t i = 0; i < cand.size(); i++)
    if (ok[i]) keys.push_back(slot[i]);
  std::stable_sort(keys.begin(), keys.end(),
                   [](const AffKey &k1, const AffKey &k2) { return std::fabs(k1.response) > std::fabs(k2.response); });
  // AffineDetector::prepareKeysForExport, scale-space-detector.hpp:126-198: the sorted list is cut as the mode says
  if (p.mode != 0 && !keys.empty()) {
    const int regNumber = (int)keys.size();
    auto above = [&](double thr) {      // std::lower_bound(keys, tempKey, responseCompareInvOrder): #{|response| > |thr|}
      int m = 0;
      while (m < regNumber && std::fabs(keys[m].response) > std::fabs(thr)) m++;
      return m;
    };
    int keep = regNumber;
    switch (p.mode) {
      case 1: {                                                       // RELATIVE_TH
        const float effectiveThreshold = (float)(std::fabs(keys[0].response) * p.rel_threshold);   // float member, pyramid.h:74
        keep = above((double)effectiveThreshold);
        break;
      }
      case 2: keep = std::min(regNumber, p.reg_number); break;       // FIXED_REG_NUMBER (the 3x Baumberg margin is cut again at :193-194)
      case 3: keep = (int)std::floor(p.rel_reg_number * (double)regNumber); break;   // RELATIVE_REG_NUMBER
      case 4: {                                                       // NOT_LESS_THAN_REGIONS (threshold un-squared, as written)
        const int fixTh = above((double)p.threshold);
        keep = fixTh < p.reg_number ? std::min(p.reg_number, regNumber) : std::min(fixTh, regNumber);
        break;
      }
    }
    if (keep < 0) keep = 0;
    if (keep < regNumber) keys.resize(keep);
  }
  out.clear();
  out.reserve(keys.size());
  for (size_t i = 0; i < keys.size(); i++) {
    AffKey k = keys[i];
    k.s = k.s * std::sqrt(std::fabs(k.a11 * k.a22 - k.a12 * k.a21));
    rectify_transformation(k.a11, k.a12, k.a21, k.a22);
    out.push_back(k);
  }
}

}  // namespace orc
