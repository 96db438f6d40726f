// ORACLE — TEST INFRASTRUCTURE ONLY.  Image primitives of the reference restated on a plain
// float image.  Build with -ffp-contract=off: every "a*b+c" below is two roundings.
#include "orc.h"
#include "detmath.h"
#include "atan_lut_data.h"
#include <cmath>
#include <cstdlib>

namespace orc {

static const double g_atan_lut[256] = ORC_ATAN_LUT_INIT;
const double *atan_lut() { return g_atan_lut; }

// Rounding variants of the OpenCV-dependent primitives (tools/readme_count_hunt.py).  The defaults (orc.h) are the
// contract stated in DESIGN.md - the reading that reproduces the reference's README counts; the other values are
// alternative readings of what an OpenCV build may do, kept so that the script can reproduce its table.
Variant g_variant;
int g_libm_variant = 0;
int g_threads = 1;

// detectors/helpers.cpp:720-721 / 728-729
int gauss_ksize(float sigma) {
  int size = (int)(2.0 * 3.0 * sigma + 1.0);
  if (size % 2 == 0) size++;
  return size;
}

// OpenCV getGaussianKernel(n, sigma, CV_32F) (imgproc/smooth): t = exp(-x^2/(2 sigma^2)) in
// double, stored as float, normalised by the double sum of the float taps.  exp -> det_exp.
std::vector<float> gauss_kernel(int n, double sigma) {
  std::vector<float> k(n);
  std::vector<double> kd(n);
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    double t = g_variant.libm ? std::exp(scale2X * x * x) : det_exp(scale2X * x * x);
    if (g_variant.kernel == 0) { k[i] = (float)t; sum += k[i]; }
    else { kd[i] = t; sum += t; }
  }
  if (g_variant.kernel == 2) { for (int i = 0; i < n; i++) k[i] = (float)(kd[i] / sum); return k; }   // one division per tap
  sum = 1. / sum;
  for (int i = 0; i < n; i++) k[i] = g_variant.kernel == 0 ? (float)(k[i] * sum) : (float)(kd[i] * sum);
  return k;
}

// cv::GaussianBlur(src, dst, Size(n,n), sigma, sigma, BORDER_REPLICATE) on CV_32F
// (helpers.cpp:717-731) = sepFilter2D with a float intermediate, evaluated as an FMA build of OpenCV does
// (v_muladd / _mm256_fmadd_ps in the row and column filters).  Of the readings tried by tools/readme_count_hunt.py
// this is the one that reproduces the reference's README counts exactly (2665/2331, 3287/2912); DESIGN.md has the table.
//   row pass, n > 5  : RowFilter order          s = k[0]*S[x-r]; s = fma(k[j], S[x-r+j], s), j = 1..n-1
//   row pass, n <= 5 : SymmRowSmallFilter order s = k[r]*S[x];   s = fma(S[x-j] + S[x+j], k[r+j], s), j = 1..r
//   column pass      : SymmColumnFilter order   s = k[r]*T[y];   s = fma(k[r+j], T[y+j] + T[y-j], s), j = 1..r
// fp32 throughout.  g_variant.row_fma / col_fma = 0 give the unfused (two roundings) form for the hunt script.
void gauss_blur(const Img &src, Img &dst, float sigma) {
  const int n = gauss_ksize(sigma);
  const int r = n / 2;
  const std::vector<float> k = gauss_kernel(n, (double)sigma);
  const int w = src.w, h = src.h;
  const bool rf = g_variant.row_fma != 0, cf = g_variant.col_fma != 0;
  auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
  Img tmp(w, h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int y = 0; y < h; y++) {
    const float *S = src.row(y);
    float *T = tmp.row(y);
    for (int x = 0; x < w; x++) {
      float s;
      if (n <= 5 && g_variant.small_row) {
        s = S[x] * k[r];
        for (int j = 1; j <= r; j++) {
          const float pr = S[cl(x - j, w - 1)] + S[cl(x + j, w - 1)];
          s = rf ? std::fmaf(pr, k[r + j], s) : s + pr * k[r + j];
        }
      } else {
        s = k[0] * S[cl(x - r, w - 1)];
        for (int j = 1; j < n; j++) {
          const float v = S[cl(x - r + j, w - 1)];
          s = rf ? std::fmaf(k[j], v, s) : s + k[j] * v;
        }
      }
      T[x] = s;
    }
  }
  Img out(w, h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int y = 0; y < h; y++) {
    float *D = out.row(y);
    for (int x = 0; x < w; x++) {
      float s = k[r] * tmp.at(y, x);
      for (int j = 1; j <= r; j++) {
        const float pr = tmp.at(cl(y + j, h - 1), x) + tmp.at(cl(y - j, h - 1), x);
        s = cf ? std::fmaf(k[r + j], pr, s) : s + k[r + j] * pr;
      }
      D[x] = s;
    }
  }
  dst = out;
}

static int cv_round_half_even(double v) {
  double fl = std::floor(v);
  double diff = v - fl;
  if (diff > 0.5) return (int)fl + 1;
  if (diff < 0.5) return (int)fl;
  return (((long long)fl) & 1LL) ? (int)fl + 1 : (int)fl;
}

// cv::resize(src, dst, Size(0,0), 0.5, 0.5, INTER_LINEAR), pyramid.cpp:476.  OpenCV maps the
// exact 2x decimation to its "area fast" path: dsize = cvRound(size*0.5) (half to even),
// every full 2x2 block -> ((a+b)+(c+d))*0.25f, blocks cut by the right/bottom edge -> running
// sum of the available pixels divided by their count.  Parity unpinned (no OpenCV here).
void resize_half(const Img &src, Img &dst) {
  const int w = src.w, h = src.h;
  const int dw = cv_round_half_even(w * 0.5), dh = cv_round_half_even(h * 0.5);
  const int dwidth1 = w / 2;
  Img out(dw, dh);
  for (int dy = 0; dy < dh; dy++) {
    float *D = out.row(dy);
    const int sy0 = dy * 2;
    if (sy0 >= h) { for (int dx = 0; dx < dw; dx++) D[dx] = 0; continue; }
    const int wfull = (sy0 + 2 <= h) ? dwidth1 : 0;
    int dx = 0;
    for (; dx < wfull; dx++) {
      const float *S0 = src.row(sy0) + 2 * dx;
      const float *S1 = src.row(sy0 + 1) + 2 * dx;
      // resize_tail = L: the last wfull % L outputs of a row come from the scalar loop (running sum of the block)
      if (g_variant.resize_tail > 0 && dx >= wfull - wfull % g_variant.resize_tail)
        D[dx] = (((S0[0] + S0[1]) + S1[0]) + S1[1]) * 0.25f;
      else
        D[dx] = ((S0[0] + S0[1]) + (S1[0] + S1[1])) * 0.25f;
    }
    for (; dx < dw; dx++) {
      const int sx0 = dx * 2;
      if (sx0 >= w) { D[dx] = 0; continue; }
      float sum = 0; int count = 0;
      for (int sy = 0; sy < 2; sy++) {
        if (sy0 + sy >= h) break;
        for (int sx = 0; sx < 2; sx++) {
          if (sx0 + sx >= w) break;
          sum += src.at(sy0 + sy, sx0 + sx);
          count++;
        }
      }
      D[dx] = sum / (float)count;
    }
  }
  dst = out;
}

// ScaleSpaceDetector::HessianResponse, pyramid.cpp:196-254.  The reference leaves the 1-px
// frame uninitialised (never read: border >= 2); it is 0 here.
void hessian_response(const Img &in, Img &out, float norm) {
  const int rows = in.h, cols = in.w;
  Img o(cols, rows);
  const float norm2 = norm * norm;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 1; r < rows - 1; ++r) {
    const float *p0 = in.row(r - 1), *p1 = in.row(r), *p2 = in.row(r + 1);
    float *q = o.row(r);
    for (int c = 1; c < cols - 1; ++c) {
      const float v11 = p0[c - 1], v12 = p0[c], v13 = p0[c + 1];
      const float v21 = p1[c - 1], v22 = p1[c], v23 = p1[c + 1];
      const float v31 = p2[c - 1], v32 = p2[c], v33 = p2[c + 1];
      float Lxx = (v21 - 2 * v22 + v23);
      float Lyy = (v12 - 2 * v22 + v32);
      float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
      q[c] = (Lxx * Lyy - Lxy * Lxy) * norm2;
    }
  }
  out = o;
}

// ScaleSpaceDetector::dogResponse / iidogResponse, pyramid.cpp:165-194.  `norm` (= sigma^2 of the level, pyramid.cpp:445,
// 458) is what the reference passes to gaussianBlur as the sigma.  Mat - Mat and Mat + Mat are one float operation per
// pixel; the iiDoG rescale is computed in double (255. / *SumPtr) and stored to float.
void dog_response(const Img &in, Img &out, float norm, bool ii) {
  Img nb;
  gauss_blur(in, nb, norm);
  Img o(in.w, in.h);
  const size_t n = (size_t)in.w * in.h;
  for (size_t i = 0; i < n; i++) {
    float v = in.d[i] - nb.d[i];
    if (ii) {
      const float sum = in.d[i] + nb.d[i];
      if (sum < 255.) v = (float)((double)v * (255. / (double)sum));
    }
    o.d[i] = v;
  }
  out = o;
}

// ScaleSpaceDetector::HarrisResponse, pyramid.cpp:256-278, with the OpenCV expression templates written out:
//   Lx.mul(Lx)                    cv::multiply, scale 1: one float product
//   sigmasq * gaussianBlur(..)    MatExpr scale -> convertTo(alpha): src * (float)alpha
//   dx2 + dy2                     cv::add
//   a.mul(b) - c.mul(c) - 0.04*s.mul(s)   = (fl(a*b) - fl(c*c)) - fl(fl(0.04f*s)*s): the two products are evaluated to
//                                 matrices, subtracted (MatOp::subtract -> cv::subtract), and cv::multiply with a scale
//                                 computes scale*src1*src2 left to right in float (arithm: mul_ / op_mul_scale)
void harris_response(const Img &in, Img &out, float norm) {
  const int rows = in.h, cols = in.w;
  const float sigmasq = (float)(0.6 * (double)norm);
  const float sigma = std::sqrt(sigmasq);
  Img Lx(cols, rows), Ly(cols, rows);
  compute_gradient(in, Lx, Ly);
  const size_t n = (size_t)cols * rows;
  Img xx(cols, rows), yy(cols, rows), xy(cols, rows), bxx, byy, bxy;
  for (size_t i = 0; i < n; i++) { xx.d[i] = Lx.d[i] * Lx.d[i]; yy.d[i] = Ly.d[i] * Ly.d[i]; xy.d[i] = Lx.d[i] * Ly.d[i]; }
  gauss_blur(xx, bxx, sigma); gauss_blur(yy, byy, sigma); gauss_blur(xy, bxy, sigma);
  Img o(cols, rows);
  const float k = (float)0.04;
  for (size_t i = 0; i < n; i++) {
    const float dx2 = bxx.d[i] * sigmasq, dy2 = byy.d[i] * sigmasq, dxdy = bxy.d[i] * sigmasq;
    const float sum = dx2 + dy2;
    const float t1 = dx2 * dy2, t2 = dxdy * dxdy;
    const float t3 = (k * sum) * sum;
    o.d[i] = (t1 - t2) - t3;
  }
  out = o;
}

// detectors/helpers.cpp:524-549
bool interpolate_check_borders(int orig_img_w, int orig_img_h, float ofsx, float ofsy, float a11,
                               float a12, float a21, float a22, int res_w, int res_h) {
  const int width = orig_img_w - 2;
  const int height = orig_img_h - 2;
  const float halfWidth = (float)std::ceil((float)res_w / 2.0);
  const float halfHeight = (float)std::ceil((float)res_h / 2.0);
  float x[4] = {-halfWidth, -halfWidth, +halfWidth, +halfWidth};
  float y[4] = {-halfHeight, +halfHeight, -halfHeight, +halfHeight};
  for (int i = 0; i < 4; i++) {
    float imx = ofsx + x[i] * a11 + y[i] * a12;
    float imy = ofsy + x[i] * a21 + y[i] * a22;
    if (std::floor(imx) <= 0 || std::floor(imy) <= 0 || std::ceil(imx) >= width || std::ceil(imy) >= height)
      return true;
  }
  return false;
}

// detectors/helpers.cpp:551-626.  WX/WY are *accumulated* in fp32 along the row (and rx/ry
// down the rows): the k-th sample coordinate is the result of k sequential float additions.
bool interpolate(const Img &im, float ofsx, float ofsy, float a11, float a12, float a21, float a22,
                 Img &res) {
  bool ret = false;
  const int width = im.w - 1;
  const int height = im.h - 1;
  const int halfWidth = res.w / 2;
  const int halfHeight = res.h / 2;
  float *out = res.d.data();
  float rx = ofsx - (float)halfHeight * a12;
  float ry = ofsy - (float)halfHeight * a22;
  bool touch = interpolate_check_borders(im.w, im.h, ofsx, ofsy, a11, a12, a21, a22, res.w, res.h);
  if (!touch) {
    for (int j = -halfHeight; j < res.h - halfHeight; ++j) {
      float WX = rx - (float)halfWidth * a11;
      float WY = ry - (float)halfWidth * a21;
      for (int i = -halfWidth; i < res.w - halfWidth; ++i) {
        const int x = (int)(WX);
        const int y = (int)(WY);
        const float wx = WX - (float)x;
        const float *Row0 = im.row(y);
        const float *Row1 = im.row(y + 1);
        const float I1 = wx * (Row0[x + 1] - Row0[x]) + Row0[x];
        *out++ = (WY - y) * (wx * (Row1[x + 1] - Row1[x]) + Row1[x] - I1) + I1;
        WX += a11;
        WY += a21;
      }
      rx += a12;
      ry += a22;
    }
  } else {
    for (int j = -halfHeight; j < res.h - halfHeight; ++j) {
      float WX = rx - halfWidth * a11;
      float WY = ry - halfWidth * a21;
      for (int i = -halfWidth; i < res.w - halfWidth; ++i) {
        const int x = (int)std::floor(WX);
        const int y = (int)std::floor(WY);
        if (WX >= 0 && WY >= 0 && x < width && y < height) {
          const float wx = WX - x;
          const float *Row0 = im.row(y);
          const float *Row1 = im.row(y + 1);
          const float I1 = wx * (Row0[x + 1] - Row0[x]) + Row0[x];
          *out++ = (WY - y) * (wx * (Row1[x + 1] - Row1[x]) + Row1[x] - I1) + I1;
        } else {
          *out++ = 0;
          ret = true;
        }
        WX += a11;
        WY += a21;
      }
      rx += a12;
      ry += a22;
    }
  }
  return ret;
}

// detectors/helpers.cpp:411-440 (exp(float) -> det_expf)
void compute_gauss_mask(Img &mask) {
  int size = mask.w;
  int halfSize = size >> 1;
  float scale = float(halfSize) / 3.0f;
  float scale2 = -2.0f * scale * scale;
  std::vector<float> tmp(halfSize + 1);
  for (int i = 0; i <= halfSize; i++) tmp[i] = det_expf((float(i * i) / scale2));
  int endSize = int(std::ceil(scale * 5.0f) - halfSize);
  for (int i = 1; i < endSize; i++)
    tmp[halfSize - i] += det_expf((float((i + halfSize) * (i + halfSize)) / scale2));
  for (int i = 0; i <= halfSize; i++)
    for (int j = 0; j <= halfSize; j++) {
      float v = tmp[i] * tmp[j];
      mask.at(i + halfSize, -j + halfSize) = v;
      mask.at(-i + halfSize, j + halfSize) = v;
      mask.at(i + halfSize, j + halfSize) = v;
      mask.at(-i + halfSize, -j + halfSize) = v;
    }
}

// detectors/helpers.cpp:442-461
void compute_circular_gauss_mask(Img &mask, float sigma) {
  int size = mask.w;
  int halfSize = size >> 1;
  float r2 = float(halfSize * halfSize);
  float sigma2;
  if (sigma == 0) sigma2 = 0.9f * r2;
  else sigma2 = 2 * sigma * sigma;
  float *mp = mask.d.data();
  for (int i = 0; i < mask.h; i++)
    for (int j = 0; j < mask.w; j++) {
      float disq = float((i - halfSize) * (i - halfSize) + (j - halfSize) * (j - halfSize));
      *mp++ = (disq < r2) ? det_expf(-disq / sigma2) : 0;
    }
}

// detectors/helpers.cpp:779-797
void compute_gradient(const Img &img, Img &gradx, Img &grady) {
  const int width = img.w, height = img.h;
  for (int r = 0; r < height; ++r)
    for (int c = 0; c < width; ++c) {
      float xgrad, ygrad;
      if (c == 0) xgrad = img.at(r, c + 1) - img.at(r, c);
      else if (c == width - 1) xgrad = img.at(r, c) - img.at(r, c - 1);
      else xgrad = img.at(r, c + 1) - img.at(r, c - 1);
      if (r == 0) ygrad = img.at(r + 1, c) - img.at(r, c);
      else if (r == height - 1) ygrad = img.at(r, c) - img.at(r - 1, c);
      else ygrad = img.at(r + 1, c) - img.at(r - 1, c);
      gradx.at(r, c) = xgrad;
      grady.at(r, c) = ygrad;
    }
}

static inline void swapf(float *a, float *b) { float t = *a; *a = *b; *b = t; }

// detectors/helpers.cpp:309-368
void solve_linear_3x3(float *A, float *b) {
  int i = 0;
  float *pr = A;
  float vp = std::fabs(A[0]);
  float tmp = std::fabs(A[3]);
  if (tmp > vp) { pr = A + 3; i = 1; vp = tmp; }
  if (std::fabs(A[6]) > vp) { pr = A + 6; i = 2; }
  if (pr != A) {
    swapf(pr, A); swapf(pr + 1, A + 1); swapf(pr + 2, A + 2); swapf(b + i, b);
  }
  vp = A[3] / A[0];
  A[4] -= vp * A[1]; A[5] -= vp * A[2]; b[1] -= vp * b[0];
  vp = A[6] / A[0];
  A[7] -= vp * A[1]; A[8] -= vp * A[2]; b[2] -= vp * b[0];
  if (std::fabs(A[4]) < std::fabs(A[7])) {
    swapf(A + 7, A + 4); swapf(A + 8, A + 5); swapf(b + 2, b + 1);
  }
  vp = A[7] / A[4];
  A[8] -= vp * A[5];
  b[2] -= vp * b[1];
  b[2] = (b[2]) / A[8];
  b[1] = (b[1] - A[5] * b[2]) / A[4];
  b[0] = (b[0] - A[2] * b[2] - A[1] * b[1]) / A[0];
}

// detectors/helpers.cpp:463-502 (all intermediate arithmetic in double)
void inv_sqrt(float &a, float &b, float &c, float &l1, float &l2) {
  double t, r;
  if (b != 0) {
    r = double(c - a) / (2 * b);
    if (r >= 0) t = 1.0 / (r + std::sqrt(1 + r * r));
    else t = -1.0 / (-r + std::sqrt(1 + r * r));
    r = 1.0 / std::sqrt(1 + t * t);
    t = t * r;
  } else {
    r = 1;
    t = 0;
  }
  double x, z, d;
  x = 1.0 / std::sqrt(r * r * a - 2 * r * t * b + t * t * c);
  z = 1.0 / std::sqrt(t * t * a + 2 * r * t * b + r * r * c);
  d = std::sqrt(x * z);
  x /= d;
  z /= d;
  if (x < z) { l1 = float(z); l2 = float(x); }
  else { l1 = float(x); l2 = float(z); }
  a = float(r * r * x + t * t * z);
  b = float(-r * t * x + t * r * z);
  c = float(t * t * x + r * r * z);
}

// detectors/helpers.cpp:504-515
bool get_eigenvalues(float a, float b, float c, float d, float &l1, float &l2) {
  float trace = a + d;
  float delta1 = (trace * trace - 4 * (a * d - b * c));
  if (delta1 < 0) return false;
  float delta = std::sqrt(delta1);
  l1 = (trace + delta) / 2.0f;
  l2 = (trace - delta) / 2.0f;
  return true;
}

// detectors/helpers.cpp:666-715
void photometrically_normalize(Img &image, const Img &binaryMask, float &sum, float &var) {
  const int width = image.w, height = image.h;
  sum = 0;
  float gsum = 0;
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++)
      if (binaryMask.at(j, i) > 0) { sum += image.at(j, i); gsum++; }
  sum = sum / gsum;
  var = 0;
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++)
      if (binaryMask.at(j, i) > 0) var += (sum - image.at(j, i)) * (sum - image.at(j, i));
  var = (float)std::sqrt((double)(var / gsum));
  if (var < 0.0001) return;
  float fac = 50.0f / var;
  for (int j = 0; j < height; j++) {
    float *imgRow = image.row(j);
    for (int i = 0; i < width; i++, imgRow++) {
      *imgRow = 128 + fac * (*imgRow - sum);
      if (*imgRow > 255) *imgRow = 255;
      if (*imgRow < 0) *imgRow = 0;
    }
  }
}

}  // namespace orc
