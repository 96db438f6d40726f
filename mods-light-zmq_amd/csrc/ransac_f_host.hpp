// Host-side building blocks of the DEGENSAC fundamental-matrix control loop: the 7-point solver,
// the least-squares F estimators of the local optimisation, and the plane-degeneracy machinery
// (sample check, homography LO, plane-and-parallax search).  Hypothesis scoring over all
// correspondences is done by the GPU (ransac_f.hip); everything here is O(1) or runs on the rare
// degenerate branch.
//
// Reference behaviour (file:line relative to the reference root, all under degensac/):
//   lin_fm / slcm / rroots3 / FDs / FDsSym / exFDs / exFDsSym / u2f / u2fw / singulF /
//   epipole / getorisig / all_ori_valid                      Ftools.c:14-452
//   lin_fmN, denormF                                         Ftools.c:246-276, utools.c:53-70
//   checksample / Hdetect / sortDs / dHDs / rFtH / innerFH / dual_sample / u2Fit / innerH
//                                                            DegUtils.c:42-744
//   iterH / inHrani (the non-"exp" homography LO innerH uses)   ranH.c:18-135
//   svduv (CCMATH, matutls/svduv.c), lap_SVD = LAPACK dgesvd_ (lapwrap.c:21-51)
// What the callers take from the SVD back ends:
//   - the rank-2 projection of a 3x3 matrix (singulF, LAPACK): one-sided Jacobi SVD here;
//   - the null direction of the 8x9 design matrix of eight correspondences (u2f/u2fw, len <= 8;
//     the ninth left vector of CCMATH's svduv): Householder QR here, the same reflections;
//   - "the third right singular vector" of a rank-2 3x3 matrix (Hdetect).  CCMATH does not sort
//     singular values, so WHICH vector comes third is part of the behaviour: svd_v_unsorted below
//     follows CCMATH's procedure step by step.
// Agreement with the reference: ~1e-12 relative or better, sign included where it matters.
// fp64, one rounding per operation (-ffp-contract=off).
#pragma once
#include <memory>
#include <utility>

#include "ransac_host.hpp"
#include "ransac_pool.hpp"
#include "ransac_simd.hpp"
#include <functional>
#include <ctime>

namespace mods {
namespace rs {

// ---- error functions (one correspondence) --------------------------------------------------------------
// FDs, Ftools.c:94-112: Sampson error; F indexed as stored (_f1.._f9 = F[0..8])
static inline double fds_point(const double *u, const double *F, double *wsum = nullptr) {
  const double rxc = F[0] * u[3] + F[3] * u[4] + F[6];
  const double ryc = F[1] * u[3] + F[4] * u[4] + F[7];
  const double rwc = F[2] * u[3] + F[5] * u[4] + F[8];
  const double r = (u[0] * rxc + u[1] * ryc + rwc);
  const double rx = F[0] * u[0] + F[1] * u[1] + F[2];
  const double ry = F[3] * u[0] + F[4] * u[1] + F[5];
  const double w = rxc * rxc + ryc * ryc + rx * rx + ry * ry;
  if (wsum) *wsum = w;
  return r * r / w;
}
// FDsSym, Ftools.c:114-135: symmetric epipolar distance; optionally returns a*b/(a+b) (exFDsSym :186-209)
static inline double fds_sym_point(const double *u, const double *F, double *wq = nullptr) {
  const double rxc = F[0] * u[3] + F[3] * u[4] + F[6];
  const double ryc = F[1] * u[3] + F[4] * u[4] + F[7];
  const double rwc = F[2] * u[3] + F[5] * u[4] + F[8];
  const double r = (u[0] * rxc + u[1] * ryc + rwc);
  const double rx = F[0] * u[0] + F[1] * u[1] + F[2];
  const double ry = F[3] * u[0] + F[4] * u[1] + F[5];
  const double a = rxc * rxc + ryc * ryc;
  const double b = rx * rx + ry * ry;
  if (wq) { *wq = (a * b) / (a + b); return r * r / *wq; }
  return r * r * (a + b) / (a * b);
}
static inline void FDs_all(const double *u, const double *F, double *p, int len) {
  for (int i = 0; i < len; i++) p[i] = fds_point(u + 6 * i, F);
}
static inline void FDsSym_all(const double *u, const double *F, double *p, int len) {
  for (int i = 0; i < len; i++) p[i] = fds_sym_point(u + 6 * i, F);
}
static inline void exFDs_all(const double *u, const double *F, double *p, double *w, int len) {   // Ftools.c:163-185
  for (int i = 0; i < len; i++) {
    double ws;
    p[i] = fds_point(u + 6 * i, F, &ws);
    w[i] = 1 / std::sqrt(ws);
  }
}
static inline void exFDsSym_all(const double *u, const double *F, double *p, double *w, int len) {   // Ftools.c:186-209
  for (int i = 0; i < len; i++) p[i] = fds_sym_point(u + 6 * i, F, &w[i]);
}

// ---- 7-point solver pieces ---------------------------------------------------------------------------------
// lin_fm, Ftools.c:14-36: 9 x len, row (3k+l) holds u'[k]*u[l], row stride len
static inline void lin_fm(const double *u, double *p, const int *inl, int len) {
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) p[(size_t)(3 * k + l) * len + i] = s[k + 3] * s[l];
  }
}

// slcm, Ftools.c:38-92: coefficients of det(A + (x - 1) B) = po[0] x^3 + po[1] x^2 + po[2] x + po[3];
// B is replaced by A - B as in the reference.  The cubic can be ill-conditioned (its roots decide which
// candidate matrices exist at all), so the coefficients are accumulated term by term in the reference's
// order: the determinant expansion below is that closed form, entry (r,c) of A/B written a<r><c>/b<r><c>.
static inline void slcm(const double *A, double *B, double *po) {
  const double a11 = A[0], a12 = A[1], a13 = A[2], a21 = A[3], a22 = A[4], a23 = A[5], a31 = A[6], a32 = A[7], a33 = A[8];
  {
    const double b11 = B[0], b12 = B[1], b13 = B[2], b21 = B[3], b22 = B[4], b23 = B[5], b31 = B[6], b32 = B[7], b33 = B[8];
    po[0] = -(b13 * b22 * b31) + b12 * b23 * b31 + b13 * b21 * b32 - b11 * b23 * b32 - b12 * b21 * b33 + b11 * b22 * b33;
    po[1] = -(a33 * b12 * b21) + a32 * b13 * b21 + a33 * b11 * b22 - a31 * b13 * b22 - a32 * b11 * b23 + a31 * b12 * b23 +
            a23 * b12 * b31 - a22 * b13 * b31 - a13 * b22 * b31 + 3 * b13 * b22 * b31 + a12 * b23 * b31 - 3 * b12 * b23 * b31 -
            a23 * b11 * b32 + a21 * b13 * b32 + a13 * b21 * b32 - 3 * b13 * b21 * b32 - a11 * b23 * b32 + 3 * b11 * b23 * b32 +
            (a22 * b11 - a21 * b12 - a12 * b21 + 3 * b12 * b21 + a11 * b22 - 3 * b11 * b22) * b33;
    po[2] = -(a21 * a33 * b12) + a21 * a32 * b13 + a13 * a32 * b21 - a12 * a33 * b21 + 2 * a33 * b12 * b21 - 2 * a32 * b13 * b21 -
            a13 * a31 * b22 + a11 * a33 * b22 - 2 * a33 * b11 * b22 + 2 * a31 * b13 * b22 + a12 * a31 * b23 - a11 * a32 * b23 +
            2 * a32 * b11 * b23 - 2 * a31 * b12 * b23 + 2 * a13 * b22 * b31 - 3 * b13 * b22 * b31 - 2 * a12 * b23 * b31 +
            3 * b12 * b23 * b31 + a13 * a21 * b32 - 2 * a21 * b13 * b32 - 2 * a13 * b21 * b32 + 3 * b13 * b21 * b32 +
            2 * a11 * b23 * b32 - 3 * b11 * b23 * b32 +
            a23 * (-(a32 * b11) + a31 * b12 + a12 * b31 - 2 * b12 * b31 - a11 * b32 + 2 * b11 * b32) +
            (-(a12 * a21) + 2 * a21 * b12 + 2 * a12 * b21 - 3 * b12 * b21 - 2 * a11 * b22 + 3 * b11 * b22) * b33 +
            a22 * (a33 * b11 - a31 * b13 - a13 * b31 + 2 * b13 * b31 + a11 * b33 - 2 * b11 * b33);
  }
  for (int i = 0; i < 9; i++) B[i] = A[i] - B[i];
  {
    const double b11 = B[0], b12 = B[1], b13 = B[2], b21 = B[3], b22 = B[4], b23 = B[5], b31 = B[6], b32 = B[7], b33 = B[8];
    po[3] = -(b13 * b22 * b31) + b12 * b23 * b31 + b13 * b21 * b32 - b11 * b23 * b32 - b12 * b21 * b33 + b11 * b22 * b33;
  }
}

// rroots3, Ftools.c:211-254: real roots of the cubic (Cardano / trigonometric form), in the
// reference's order
static inline int rroots3(const double *po, double *r) {
  const double b = po[1] / po[0];
  const double c = po[2] / po[0];
  const double b2 = b * b;
  const double bt = b / 3;
  const double p = (3 * c - b2) / 9;
  const double q = ((2 * b2 * b) / 27 - b * c / 3 + po[3] / po[0]) / 2;
  const double D = q * q + p * p * p;
  if (D > 0) {
    const double A = std::sqrt(D) - q;
    if (A > 0) {
      const double v = std::pow(A, 1.0 / 3);
      r[0] = v - p / v - bt;
    } else {
      const double v = std::pow(-A, 1.0 / 3);
      r[0] = p / v - v - bt;
    }
    return 1;
  }
  const double e = q > 0 ? 1 : -1;
  const double R = e * std::sqrt(-p);
  const double R2 = R * 2;
  double cosphi = q / (R * R * R);
  if (cosphi > 1) cosphi = 1;
  else if (cosphi < -1) cosphi = -1;
  const double phit = std::acos(cosphi) / 3;
  const double pit = 3.14159265358979 / 3;
  r[0] = -R2 * std::cos(phit) - bt;
  r[1] = R2 * std::cos(pit - phit) - bt;
  r[2] = R2 * std::cos(pit + phit) - bt;
  return 3;
}

// epipole / getorisig / all_ori_valid, Ftools.c:408-443
static inline void epipole(double *ec, const double *F) {
  crossprod(ec, F, F + 6);
  for (int i = 0; i < 3; i++)
    if ((ec[i] > 1.9984e-15) || (ec[i] < -1.9984e-15)) return;
  crossprod(ec, F + 3, F + 6);
}
static inline double getorisig(const double *F, const double *ec, const double *u) {
  const double s1 = F[0] * u[3] + F[3] * u[4] + F[6] * u[5];
  const double s2 = ec[1] * u[2] - ec[2] * u[1];
  return s1 * s2;
}
static inline int all_ori_valid(const double *F, const double *us, const int *idx, int N) {
  double ec[3];
  epipole(ec, F);
  const double sig1 = getorisig(F, ec, us + 6 * idx[0]);
  for (int i = 1; i < N; i++) {
    const double sig = getorisig(F, ec, us + 6 * idx[i]);
    if (sig1 * sig < 0) return 0;
  }
  return 1;
}

// ---- small SVD replacements ---------------------------------------------------------------------------------
// One-sided Jacobi on a 3x3 row-major matrix: A = W V^T with mutually orthogonal columns of W
// (norm of column j = singular value j, unsorted).
static inline void jacobi_svd3(const double *A, double *W, double *V) {
  for (int i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; k++) { alpha += W[3 * k + p] * W[3 * k + p]; beta += W[3 * k + q] * W[3 * k + q]; gamma += W[3 * k + p] * W[3 * k + q]; }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          const double wp = W[3 * k + p], wq = W[3 * k + q];
          W[3 * k + p] = c * wp - s * wq;
          W[3 * k + q] = s * wp + c * wq;
          const double vp = V[3 * k + p], vq = V[3 * k + q];
          V[3 * k + p] = c * vp - s * vq;
          V[3 * k + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
}
static inline int smallest_column(const double *W) {
  int jmin = 0;
  double nmin = 0;
  for (int j = 0; j < 3; j++) {
    double n = 0;
    for (int k = 0; k < 3; k++) n += W[3 * k + j] * W[3 * k + j];
    if (j == 0 || n < nmin) { nmin = n; jmin = j; }
  }
  return jmin;
}
// singulF, Ftools.c:279-299: closest rank-2 matrix (smallest singular component removed)
static inline void singulF(double *F) {
  double W[9], V[9];
  jacobi_svd3(F, W, V);
  const int jmin = smallest_column(W);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int j = 0; j < 3; j++)
        if (j != jmin) s += W[3 * r + j] * V[3 * c + j];
      F[3 * r + c] = s;
    }
}
// Right singular vectors of a small row-major m x n matrix (m >= n <= 9) the way CCMATH's svduv
// produces them (matutls/svduv.c + ldvmat.c + qrbdv.c): Householder bidiagonalisation, then implicit-
// shift QR sweeps on the bidiagonal with deflation from the bottom.  The singular values are NOT
// sorted; Hdetect takes "the third column of V" as the epipole (DegUtils.c:110-111), and for a
// rank-2 matrix the zero singular value does not always end up third, so the order in which this
// procedure deflates is part of the reference's behaviour and is followed step by step.  Only V and
// the singular values are produced (the left vectors do not feed back into them).
static inline void svd_v_unsorted(const double *Ain, int m, int n, double *d, double *V) {
  double a[81], e[10], w[9];
  for (int i = 0; i < m * n; i++) a[i] = Ain[i];
  for (int i = 0; i < 10; i++) e[i] = 0;
  // -- bidiagonalisation: column reflections leave d, row reflections leave e; the scaled reflection
  //    vectors stay in `a` (first component replaced by the factor sv)
  for (int i = 0; i < n; i++) {
    const int mm = m - i, nm = n - 1 - i;
    if (mm > 1) {
      double sv = 0., h = 0., s = 0.;
      for (int j = 0; j < mm; j++) { w[j] = a[(i + j) * n + i]; s += w[j] * w[j]; }
      if (s > 0.) {
        const double piv = a[i * n + i];
        h = std::sqrt(s);
        if (piv < 0.) h = -h;
        s += piv * h; s = 1. / s;
        w[0] += h;
        const double t = 1. / w[0];
        sv = 1. + std::fabs(piv / h);
        for (int k = 1; k < n - i; k++) {
          double r = 0.;
          for (int j = 0; j < mm; j++) r += w[j] * a[(i + j) * n + i + k];
          r *= s;
          for (int j = 0; j < mm; j++) a[(i + j) * n + i + k] -= r * w[j];
        }
        for (int j = 1; j < mm; j++) a[(i + j) * n + i] = t * w[j];
      }
      a[i * n + i] = sv; d[i] = -h;
    }
    if (mm == 1) d[i] = a[i * n + i];
    if (nm > 1) {
      double sv = 0., h = 0., s = 0.;
      double *row = &a[i * n + i + 1];
      for (int j = 0; j < nm; j++) s += row[j] * row[j];
      if (s > 0.) {
        h = std::sqrt(s);
        if (row[0] < 0.) h = -h;
        sv = 1. + std::fabs(row[0] / h);
        s += row[0] * h; s = 1. / s;
        row[0] += h;
        const double t = 1. / row[0];
        for (int r2 = i + 1; r2 < m; r2++) {
          double *other = &a[r2 * n + i + 1];
          double r = 0.;
          for (int j = 0; j < nm; j++) r += row[j] * other[j];
          r *= s;
          for (int j = 0; j < nm; j++) other[j] -= r * row[j];
        }
        for (int j = 1; j < nm; j++) row[j] *= t;
      }
      row[0] = sv; e[i] = -h;
    }
    if (nm == 1) e[i] = a[i * n + i + 1];
  }
  // -- V from the stored row reflections
  for (int i = 0; i < n * n; i++) V[i] = 0.;
  V[0] = 1.; V[n * n - 1] = 1.;
  for (int i = n - 2; i > 0; i--) {
    const int mm = n - 1 - i;
    const double sv = a[(i - 1) * n + i];
    const double *tail = &a[(i - 1) * n + i + 1];
    if (sv != 0.) {
      V[i * n + i] = 1. - sv;
      for (int j = 0; j < mm; j++) V[(i + 1 + j) * n + i] = -sv * tail[j];
      for (int k = i + 1; k < n; k++) {
        double s = 0.;
        for (int j = 0; j < mm; j++) s += V[(i + 1 + j) * n + k] * tail[j];
        s *= sv;
        for (int j = 0; j < mm; j++) V[(i + 1 + j) * n + k] -= s * tail[j];
        V[i * n + k] = -s;
      }
    } else {
      V[i * n + i] = 1.;
      for (int j = 0; j < mm; j++) { V[i * n + i + 1 + j] = 0.; V[(i + 1 + j) * n + i] = 0.; }
    }
  }
  // -- QR sweeps on the bidiagonal (d, e), rotations accumulated into V
  double t = std::fabs(d[0]);
  for (int j = 1; j < n; j++) { const double s = std::fabs(d[j]) + std::fabs(e[j - 1]); if (s > t) t = s; }
  t *= 1.e-15;
  int mw = n;
  const int maxit = 100 * n;
  for (int it = 0; mw > 1 && it < maxit; ++it) {
    int k;
    for (k = mw - 1; k > 0; --k) {
      if (std::fabs(e[k - 1]) < t) break;
      if (std::fabs(d[k - 1]) < t) {
        double s = 1., c = 0.;
        for (int i = k; i < mw; ++i) {
          const double aa = s * e[i - 1], bb = d[i];
          e[i - 1] *= c;
          const double uu = std::sqrt(aa * aa + bb * bb);
          d[i] = uu; s = -aa / uu; c = bb / uu;
        }
        break;
      }
    }
    double y = d[k], x = d[mw - 1], u = e[mw - 2];
    double aa = (y + x) * (y - x) - u * u, s = y * e[k], bb = s + s;
    u = std::sqrt(aa * aa + bb * bb);
    if (u != 0.) {
      double c = std::sqrt((u + aa) / (u + u));
      if (c != 0.) s /= (c * u); else s = 1.;
      for (int i = k; i < mw - 1; ++i) {
        bb = e[i];
        if (i > k) {
          aa = s * e[i]; bb *= c;
          e[i - 1] = u = std::sqrt(x * x + aa * aa);
          c = x / u; s = aa / u;
        }
        aa = c * y + s * bb; bb = c * bb - s * y;
        for (int r = 0; r < n; r++) {
          double *pv = &V[r * n + i];
          const double ww = c * pv[0] + s * pv[1];
          pv[1] = c * pv[1] - s * pv[0];
          pv[0] = ww;
        }
        s *= d[i + 1]; d[i] = u = std::sqrt(aa * aa + s * s);
        y = c * d[i + 1]; c = aa / u; s /= u;
        x = c * bb + s * y; y = c * y - s * bb;
      }
    }
    e[mw - 2] = x; d[mw - 1] = y;
    if (std::fabs(x) < t) --mw;
    if (mw == k + 1) --mw;
  }
  for (int i = 0; i < n; ++i)
    if (d[i] < 0.) {
      d[i] = -d[i];
      for (int r = 0; r < n; r++) V[r * n + i] = -V[r * n + i];
    }
}
// Hdetect's epipole: third column of V of svduv(F as stored) (DegUtils.c:108-111)
static inline void right_null3(const double *M, double *v) {
  double d[3], V[9];
  svd_v_unsorted(M, 3, 3, d, V);
  for (int k = 0; k < 3; k++) v[k] = V[3 * k + 2];
}
// Null direction of eight equations in nine unknowns, given as the 9 x 8 row-major matrix of
// lin_fm (column i = equation i): the last column of Q in the Householder QR of that matrix, i.e. the
// unit vector orthogonal to all eight columns (u2f's "V + 8" of svduv(D,Z,V,9,U,8), Ftools.c:322-333).
static inline void left_null_9x8(const double *Z, double *f) {
  double A[9 * 8];
  std::memcpy(A, Z, sizeof(A));
  double vs[8][9];
  double betas[8];
  for (int j = 0; j < 8; j++) {
    double norm2 = 0;
    for (int r = j; r < 9; r++) norm2 += A[r * 8 + j] * A[r * 8 + j];
    double *v = vs[j];
    for (int r = 0; r < 9; r++) v[r] = 0;
    if (norm2 == 0.0) { betas[j] = 0; continue; }
    const double nrm = std::sqrt(norm2);
    const double alpha = A[j * 8 + j] >= 0 ? -nrm : nrm;
    for (int r = j; r < 9; r++) v[r] = A[r * 8 + j];
    v[j] -= alpha;
    double vv = 0;
    for (int r = j; r < 9; r++) vv += v[r] * v[r];
    betas[j] = vv > 0 ? 2.0 / vv : 0.0;
    for (int c = j; c < 8; c++) {
      double dot = 0;
      for (int r = j; r < 9; r++) dot += v[r] * A[r * 8 + c];
      dot *= betas[j];
      for (int r = j; r < 9; r++) A[r * 8 + c] -= dot * v[r];
    }
  }
  for (int r = 0; r < 9; r++) f[r] = r == 8 ? 1.0 : 0.0;
  for (int j = 7; j >= 0; j--) {
    const double *v = vs[j];
    double dot = 0;
    for (int r = j; r < 9; r++) dot += v[r] * f[r];
    dot *= betas[j];
    for (int r = j; r < 9; r++) f[r] -= dot * v[r];
  }
}

// ---- least-squares F ---------------------------------------------------------------------------------------
// lin_fmN, Ftools.c:246-276: len x 9 row-major, entry [k*3+l] = a[l]*b[k] of the normalised points
static inline void lin_fmN(const double *u, double *p, const int *inl, int len, const double *A1, const double *A2) {
  double a[3], b[3];
  a[2] = 1; b[2] = 1;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    a[0] = s[0] * A1[0] + A1[1];
    a[1] = s[1] * A1[0] + A1[2];
    b[0] = s[3] * A2[0] + A2[1];
    b[1] = s[4] * A2[0] + A2[2];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) *p++ = a[l] * b[k];
  }
}
// lin_fmN (+ the optional row weights of u2fw) + cov_mat without the len x 9 matrix: the nine entries of a row are made in
// registers and go straight into the 45 running sums - the same products and the same additions in the same row order
// lanes < 0: the widest host SIMD form of this CPU (rs::SimdOps::cov_fm_all: the 45 sums side by side in vector lanes, same
// additions per sum); 0: the scalar loop below; 1 / 4 / 8: that lane count (self-test)
static inline void cov_fmN(const double *u, const int *inl, const double *w, int len, const double *A1, const double *A2, double *Cv, int lanes = -1) {
  double acc[45];
  if (lanes != 0) {
    const SimdOps *ops = lanes < 0 ? simd_ops() : simd_ops_lanes(lanes);
    if (ops) {
      ops->cov_fm_all(u, inl, w, len, A1, A2, acc);
      int q = 0;
      for (int r = 0; r < 9; r++)
        for (int c = 0; c <= r; c++) { Cv[9 * r + c] = acc[q]; Cv[r + 9 * c] = acc[q]; q++; }
      return;
    }
  }
  for (int q = 0; q < 45; q++) acc[q] = 0;
  double a[3], b[3], z[9];
  a[2] = 1; b[2] = 1;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    a[0] = s[0] * A1[0] + A1[1];
    a[1] = s[1] * A1[0] + A1[2];
    b[0] = s[3] * A2[0] + A2[1];
    b[1] = s[4] * A2[0] + A2[2];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) z[3 * k + l] = a[l] * b[k];
    if (w) {
      const double m = w[inl[i]];
      for (int k = 0; k < 9; k++) z[k] *= m;
    }
    int q = 0;
    for (int r = 0; r < 9; r++)
      for (int c = 0; c <= r; c++) acc[q++] += z[r] * z[c];
  }
  int q = 0;
  for (int r = 0; r < 9; r++)
    for (int c = 0; c <= r; c++) { Cv[9 * r + c] = acc[q]; Cv[r + 9 * c] = acc[q]; q++; }
}
// denormF, utools.c:53-70
static inline void denormF(double *F, const double *A1, const double *A2) {
  double r = A2[0], x = A2[1], y = A2[2];
  F[6] += x * F[0] + y * F[3];
  F[7] += x * F[1] + y * F[4];
  F[8] += x * F[2] + y * F[5];
  F[0] *= r; F[1] *= r; F[2] *= r;
  F[3] *= r; F[4] *= r; F[5] *= r;
  r = A1[0]; x = A1[1]; y = A1[2];
  F[2] += x * F[0] + y * F[1];
  F[5] += x * F[3] + y * F[4];
  F[8] += x * F[6] + y * F[7];
  F[0] *= r; F[3] *= r; F[6] *= r;
  F[1] *= r; F[4] *= r; F[7] *= r;
}

// u2f / u2fw, Ftools.c:302-405.  `w` == nullptr: unweighted.  `buffer` holds >= max(9*len, 81) doubles.
// len > 8: normalised 8-point algorithm through the 9x9 moment matrix; len <= 8: null direction of the
// unnormalised design matrix.  For len < 8 the reference runs its 9x8 SVD over a 9 x len buffer, i.e. on
// stale memory; that input is not defined and is not reproduced (callers guard it, see ransac_f.hip).
static inline void u2fw(const double *u, const int *inl, const double *w, int len, double *F, double *buffer, bool reference_form = false) {
  double A1[3], A2[3];
  double V[9 * 9], D[9];
  double *Z = buffer;
  if (len > 8) {
    normu(u, inl, len, A1, A2);
    if (reference_form) {   // the matrix written out, as Ftools.c:302-405 has it (kept for the self-test that pins cov_fmN to it)
      lin_fmN(u, Z, inl, len, A1, A2);
      if (w)
        for (int i = 0; i < len; i++) {
          const double m = w[inl[i]];
          for (int k = 0; k < 9; k++) Z[9 * i + k] *= m;
        }
      cov_mat(V, Z, len, 9);
    } else cov_fmN(u, inl, w, len, A1, A2, V);
    sym_eig(V, D, 9);                       // ascending: vector 0 belongs to the smallest eigenvalue
    for (int i = 0; i < 9; i++) F[i] = V[i];
  } else {
    double Z8[9 * 9 + 8];
    std::memset(Z8, 0, sizeof(Z8));
    if (len == 8) {
      lin_fm(u, Z8, inl, 8);
      if (w)                                 // scalmul(Z+i, w[j], 9, 9) over a stride-8 layout, Ftools.c:382-386
        for (int i = 0; i < len; i++) {
          const double m = w[inl[i]];
          for (int k = 0; k < 9; k++) Z8[i + 9 * k] *= m;
        }
    } else {
      lin_fm(u, Z8, inl, len);               // undefined input in the reference; a fixed, documented stand-in
      double T[9 * 8];
      std::memset(T, 0, sizeof(T));
      for (int k = 0; k < 9; k++)
        for (int i = 0; i < len; i++) T[k * 8 + i] = Z8[k * len + i];
      std::memcpy(Z8, T, sizeof(T));
    }
    left_null_9x8(Z8, F);
  }
  singulF(F);
  if (len > 8) denormF(F, A1, A2);
}
static inline void u2f(const double *u, const int *inl, int len, double *F, double *buffer) { u2fw(u, inl, nullptr, len, F, buffer); }

// ---- CCMATH helpers used by the degeneracy code (row-major) -------------------------------------------------
static inline void mat3_mul(double *c, const double *a, const double *b) {   // mmul, matutls/mmul.c
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.;
      for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
      c[3 * i + j] = s;
    }
}
static inline void mat3_tr(double *a, const double *b) {   // mattr(a,b,3,3)
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) a[3 * i + j] = b[3 * j + i];
}
static inline void skew_sym(const double *a, double *ax) {   // DegUtils.c:209-219
  ax[0] = 0; ax[1] = -a[2]; ax[2] = a[1];
  ax[3] = a[2]; ax[4] = 0; ax[5] = -a[0];
  ax[6] = -a[1]; ax[7] = a[0]; ax[8] = 0;
}

// dHDs, DegUtils.c:178-206: Sampson homography error of every correspondence
static inline void dHDs(const double *H, const double *u, unsigned len, double *Ds) {
  for (unsigned i = 0; i < len; i++) Ds[i] = hds_point(u + 6 * i, H);
}

// Evaluates an error function over ALL correspondences of a run.  The default works on the host;
// ransac_f.hip substitutes a GPU evaluation for long lists (same operations, same bits), which is what the
// O(len) passes of the local optimisations and of the degenerate branch cost on large inputs.
struct PointEval {
  const double *u = nullptr;
  int len = 0;
  PointEval(const double *u_, int len_) : u(u_), len(len_) {}
  virtual ~PointEval() {}
  virtual void hds(const double *H, double *out) { dHDs(H, u, (unsigned)len, out); }
  virtual void fds(const double *F, double *out) { FDs_all(u, F, out, len); }
  virtual void fds_sym(const double *F, double *out) { FDsSym_all(u, F, out, len); }
  virtual void exfds(const double *F, double *p, double *w) { exFDs_all(u, F, p, w, len); }
  virtual void exfds_sym(const double *F, double *p, double *w) { exFDsSym_all(u, F, p, w, len); }
  virtual bool concurrent() const { return true; }   // may several threads evaluate through this object at once?
  // optional: counts[j] = #{i : FDs(u_i, F_j) < th} for k matrices (k x 9) in one go - set by a caller that can count many models
  // at once cheaper than one evaluation each (ransac_f.hip: one launch over the resident correspondences); called from the thread
  // that owns the run only.  Unset: every model is evaluated through fds().
  std::function<void(const double *Fs, int k, double th, unsigned *counts)> count_fds;
};

// Hdetect, DegUtils.c:93-156: homography compatible with F through three correspondences
// (Hartley & Zisserman, "scene planes and homographies"); H stored column-wise like every H here
static inline void Hdetect(const double *F, const double *u7, const unsigned char *idx3, double *H) {
  double ec[3], Ex[9], A[9], Ft[9], u3a[9], u3b[9], Au3b[9], p1[9], p2[9], b[3];
  mat3_tr(Ft, F);
  right_null3(F, ec);
  skew_sym(ec, Ex);
  mat3_mul(A, Ex, Ft);
  for (int i = 0; i < 3; i++)          // columns = the three points
    for (int j = 0; j < 3; j++) { u3a[i + j * 3] = u7[idx3[i] * 6 + j]; u3b[i + j * 3] = u7[idx3[i] * 6 + j + 3]; }
  mat3_mul(Au3b, A, u3b);
  for (int c = 0; c < 3; c++) {        // p1(:,c) = u3a(:,c) x (A u3b)(:,c)
    const double x[3] = {u3a[c], u3a[3 + c], u3a[6 + c]}, y[3] = {Au3b[c], Au3b[3 + c], Au3b[6 + c]};
    p1[c] = x[1] * y[2] - x[2] * y[1];
    p1[3 + c] = x[2] * y[0] - x[0] * y[2];
    p1[6 + c] = x[0] * y[1] - x[1] * y[0];
  }
  for (int i = 0; i < 9; ++i) Ex[i] *= -1;
  mat3_mul(p2, Ex, u3a);
  for (int c = 0; c < 3; c++)
    b[c] = (p1[c] * p2[c] + p1[3 + c] * p2[3 + c] + p1[6 + c] * p2[6 + c]) / (p2[c] * p2[c] + p2[3 + c] * p2[3 + c] + p2[6 + c] * p2[6 + c]);
  double M[9];
  mat3_tr(M, u3b);                     // rows = points of image 2
  const int sing = minv3(M);
  double mb[3];
  for (int j = 0; j < 3; j++) {
    double z = 0.;
    for (int k = 0; k < 3; k++) z += M[3 * j + k] * b[k];
    mb[j] = z;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double z = 0.;
      z += ec[i] * mb[j];
      H[i + j * 3] = A[i * 3 + j] - z;
    }
  if (std::isnan(*H) || std::isinf(*H) || sing) {
    H[1] = H[2] = H[3] = H[5] = H[6] = H[7] = 0;
    H[0] = H[4] = H[8] = 1;
  }
}

// checksample, DegUtils.c:42-82: is the 7-point sample dominated by a plane (>= 5 of 7 within th of
// a homography)?  On success H holds that homography.
static inline int checksample(const double *F, const double *u7, double th, double *H) {
  static const unsigned char IDXS[5][3] = {{0, 1, 2}, {3, 4, 5}, {0, 1, 6}, {3, 4, 6}, {2, 5, 6}};
  double Ds[7], sDs[7], buffer[5 * 18];
  unsigned char idx[7];
  int inl[7];
  for (int i = 0; i < 5; ++i) {
    Hdetect(F, u7, IDXS[i], H);
    dHDs(H, u7, 7, Ds);
    std::memcpy(sDs, Ds, sizeof(sDs));            // sortDs, DegUtils.c:159-175 (exchange sort, strict <)
    for (int a = 0; a < 7; ++a) idx[a] = (unsigned char)a;
    for (int a = 0; a < 7; ++a)
      for (int c = a + 1; c < 7; ++c)
        if (sDs[c] < sDs[a]) {
          const double t = sDs[c]; sDs[c] = sDs[a]; sDs[a] = t;
          const unsigned char q = idx[c]; idx[c] = idx[a]; idx[a] = q;
        }
    for (int j = 0; j < 5; ++j) inl[j] = idx[j];
    u2h(u7, inl, 5, H, buffer);
    dHDs(H, u7, 7, Ds);
    int cnt = 0;
    for (int j = 0; j < 7; ++j)
      if (Ds[j] < th) ++cnt;
    if (cnt > 4) return 1;
  }
  return 0;
}

// ---- homography LO of the degenerate branch: iterH / inHrani (ranH.c:18-135) and innerH (DegUtils.c:699-735)
struct HLo {
  const double *u; int len;
  GlibcRand *rng;
  double *errs[5];
  double *buffer;
  PointEval *ev;
  int n_pad;          // stride of the error buffers: len rounded up to SIMD_PAD, so that the lanes-wide gain pass may read whole vectors
  double *gains;      // n_pad doubles
};
// inlidxs (rtools.c:155-166) over one error vector for the two thresholds iterH asks about back to back:
//   S  = inlidxs(err, th)   - count and MSAC sum (the sum is a serial chain of len additions: the gains are made lanes-wide
//                             first, SimdOps::gains_all = trunc_quad per lane, and added in index order);
//   Ss = inlidxs(err, ths)  - iterH reads only its count and its index list, so its sum is not formed.
// ths >= th.  `inl` receives the list of `list_ths ? ths : th`; returns S, *n_ths = Ss.I.
static inline Score inlidxs2(const HLo &L, const double *err, double th, double ths, bool list_ths, int *inl, unsigned *n_ths) {
  const int len = L.len;
  const SimdOps *ops = th != 0 ? simd_ops() : nullptr;
  Score s = {0, 0};
  unsigned n2 = 0;
  if (ops) ops->gains_all(err, L.n_pad, th * 9 / 4, L.gains);
  const double *g = L.gains;
  if (list_ths) {
    for (int i = 0; i < len; ++i) {
      const double e = err[i];
      s.J += ops ? g[i] : trunc_quad(e, th);
      if (e <= th) ++s.I;
      if (e <= ths) inl[n2++] = i;
    }
  } else {
    for (int i = 0; i < len; ++i) {
      const double e = err[i];
      s.J += ops ? g[i] : trunc_quad(e, th);
      if (e <= th) { inl[s.I] = i; ++s.I; }
      if (e <= ths) ++n2;
    }
  }
  if (n_ths) *n_ths = n2;
  return s;
}
static inline Score iterH(HLo &L, int *inliers, double th, double ths, double *H, unsigned inlLimit) {
  double *d = L.errs[1];
  double h[9];
  Score S = {0, 0}, maxS;
  unsigned SsI = 0;
  const double dth = (ths - th) / 4;
  auto lsq = [&](unsigned n) {
    if (n <= inlLimit) u2h(L.u, inliers, (int)n, h, L.buffer);
    else {
      int *sub = randsubset(*L.rng, inliers, (int)n, (int)inlLimit);
      u2h(L.u, sub, (int)inlLimit, h, L.buffer);
    }
  };
  maxS = inlidxs2(L, L.errs[4], th, th, false, inliers, nullptr);
  if (maxS.I < 4) return S;
  lsq(maxS.I);
  for (int it = 0; it < 4; ++it) {
    L.ev->hds(h, d);
    S = inlidxs2(L, d, th, ths, true, inliers, &SsI);   // S = inlidxs(d, th); Ss = inlidxs(d, ths): the list of Ss is the one that stays
    if (score_less(maxS, S)) {
      maxS = S;
      L.errs[1] = L.errs[0];
      L.errs[0] = d;
      d = L.errs[1];
      std::memcpy(H, h, 9 * sizeof(double));
    }
    if (SsI < 4) return maxS;
    lsq(SsI);
    ths -= dth;
  }
  L.ev->hds(h, d);
  S = inlidxs2(L, d, th, th, false, inliers, nullptr);
  if (score_less(maxS, S)) {
    maxS = S;
    L.errs[1] = L.errs[0];
    L.errs[0] = d;
    std::memcpy(H, h, 9 * sizeof(double));
  }
  return maxS;
}
static inline Score inHrani(HLo &L, int *inliers, int ninl, double th, double *H, unsigned inlLimit) {
  Score S, maxS = {0, 0};
  double *d, h[9];
  if (ninl < 8) return maxS;
  std::vector<int> intbuff(L.len);
  int ssiz = ninl / 2;
  if (ssiz > 12) ssiz = 12;
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  for (int i = 0; i < 10; ++i) {
    int *sample = randsubset(*L.rng, inliers, ninl, ssiz);
    u2h(L.u, sample, ssiz, h, L.buffer);
    L.ev->hds(h, L.errs[0]);
    L.errs[4] = L.errs[0];
    S = iterH(L, intbuff.data(), th, 4 * th, h, inlLimit);
    if (score_less(maxS, S)) {
      maxS = S;
      d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
      std::memcpy(H, h, 9 * sizeof(double));
    }
  }
  d = L.errs[2]; L.errs[2] = L.errs[0]; L.errs[0] = d;
  return maxS;
}
// innerH: note that the reference passes its `iters` argument on as the inlier limit of the LSQ steps.
// The ten repetitions of inHrani are independent but for two things they hand on: the generator (a repetition draws its
// 12-sample and then inlLimit numbers per least-squares step of iterH) and the order of the sampling pool, which the draws
// permute.  Neither depends on what a repetition computes as long as iterH takes all five of its least-squares steps on more
// than inlLimit inliers - the normal course.  So the samples of all repetitions are drawn first with the generator stepped
// over the draws iterH is expected to make, the repetitions run side by side (ransac_pool.hpp) on their own error buffers,
// and each checks that its generator ended where the next one was started from; if one did not (an early return inside
// iterH), the whole call is redone by the one-thread loop.  The results are folded in repetition order.
static thread_local int g_innerh_path = 0;   // how the last innerH of this thread ran: 0 one-thread loop, 1 side by side, 2 side by side, then redone
static inline unsigned innerH(double *H, const double *u, unsigned len, double th, unsigned iters, unsigned char *inl, GlibcRand &rng, double *buffer,
                              PointEval *ev = nullptr) {
  const int REPS = 10;
  g_innerh_path = 0;
  const size_t n_pad = ((size_t)len + SIMD_PAD - 1) / SIMD_PAD * SIMD_PAD;
  PointEval host_ev(u, (int)len);
  HLo L;
  L.u = u; L.len = (int)len; L.rng = &rng; L.buffer = buffer; L.ev = ev ? ev : &host_ev;
  const bool par = L.ev->concurrent() && TaskPool::get().threads() > 1;
  // scratch that lives across calls (see u2Fit): 5 buffers for the one-thread form + 3 per repetition, the index lists
  static thread_local std::vector<double> err;
  static thread_local std::vector<int> idx;
  const size_t need = n_pad * (5 + (par ? 3 * REPS : 0));
  if (err.size() < need) err.resize(need);
  if (idx.size() < (size_t)len * (2 + (par ? REPS : 0))) idx.resize((size_t)len * (2 + (par ? REPS : 0)));
  for (size_t i = len; i < n_pad; i++)
    for (size_t k = 0; k < need / n_pad; k++) err[k * n_pad + i] = 0;   // the tail the lanes-wide gain pass reads
  double *const err_base = err.data();   // (a lambda that runs on a pool thread would see THAT thread's err / idx)
  int *const idx_base = idx.data();
  int *inliers = idx_base;
  L.n_pad = (int)n_pad; L.gains = err_base + 4 * n_pad;
  for (int i = 0; i < 4; i++) L.errs[i] = err_base + (size_t)i * n_pad;
  L.errs[4] = nullptr;
  double *d = L.errs[0];
  L.ev->hds(H, d);
  Score S = inlidxs2(L, d, th, th, false, inliers, nullptr);
  const int ninl = (int)S.I;
  bool done = false;
  if (par && ninl >= 8) {
    struct Rep { GlibcRand rng, rng_end; int sample[12]; double h[9]; Score S; double *best; };
    Rep reps[REPS];
    const GlibcRand rng0 = rng;
    int *pool0 = idx_base + len;
    std::memcpy(pool0, inliers, sizeof(int) * ninl);
    int ssiz = ninl / 2;
    if (ssiz > 12) ssiz = 12;
    for (int r = 0; r < REPS; r++) {
      const int *sample = randsubset(rng, inliers, ninl, ssiz);
      std::memcpy(reps[r].sample, sample, sizeof(int) * ssiz);
      reps[r].rng = rng;
      for (unsigned k = 0; k < 5 * iters; k++) (void)rng.next();   // five randsubset(.., inlLimit) of iterH
      reps[r].rng_end = rng;
    }
    TaskPool::get().run(REPS, [&](int r) {
      Rep &R = reps[r];
      HLo T = L;
      double *base = err_base + (5 + 3 * (size_t)r) * n_pad;
      T.errs[0] = base; T.errs[1] = base + n_pad; T.errs[2] = T.errs[3] = nullptr; T.gains = base + 2 * n_pad;
      T.rng = &R.rng;
      double small[96];
      T.buffer = small;
      u2h(T.u, R.sample, ssiz, R.h, small);
      T.ev->hds(R.h, T.errs[0]);
      T.errs[4] = T.errs[0];
      R.S = iterH(T, idx_base + (size_t)len * (2 + r), th, 4 * th, R.h, iters);
      R.best = T.errs[0];
    });
    done = true;
    for (int r = 0; r < REPS; r++)
      if (std::memcmp(&reps[r].rng, &reps[r].rng_end, sizeof(GlibcRand)) != 0) done = false;
    g_innerh_path = done ? 1 : 2;
    if (done) {
      Score maxS = {0, 0};
      for (int r = 0; r < REPS; r++)
        if (score_less(maxS, reps[r].S)) {
          maxS = reps[r].S;
          d = reps[r].best;
          std::memcpy(H, reps[r].h, 9 * sizeof(double));
        }
    } else {
      rng = rng0;
      std::memcpy(inliers, pool0, sizeof(int) * ninl);
    }
  }
  if (!done) {
    inHrani(L, inliers, ninl, th, H, iters);
    d = L.errs[0];
  }
  unsigned I = 0;
  for (unsigned j = 0; j < len; j++) {
    if (d[j] <= th) { ++I; inl[j] = 1; }
    else inl[j] = 0;
  }
  return I;
}

// development aid (MODS_RANSAC_PROFILE): [0] rFtH candidate loops ms, [1] counting calls ms, [2] blocks, [3] off-plane points,
// [4] innerFH ms, [5] least-squares fits inside u2Fit ms, [6] innerFH calls, [7] u2Fit fits, [8] inner estimations that were run
// ahead and dropped (their trigger lay beyond the budget an earlier one set), [9] rounds of triggers
static thread_local double g_rfth_prof[16] = {0};   // [10] set-up before the search, [11] trigger handling (drawing the inner samples), [12] folding, [13] launching the counts, [14] / [15] the sample-evaluation and refinement stages of innerFH
static inline double rfth_prof_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

// ---- plane-and-parallax search ----------------------------------------------------------------------------
// u2Fit, DegUtils.c:629-697
// prof (optional): [0] += ms inside the least-squares fits, [1] += their number
static inline unsigned u2Fit(PointEval &ev, double *F, unsigned char *inl, double th, double ths, unsigned iters, double *prof = nullptr) {
  const double *u = ev.u;
  const unsigned len = (unsigned)ev.len;
  const double dth = (ths - th) / (iters - 1);
  // per-thread scratch that lives across calls: these are hundreds of kilobytes, and a fresh allocation of that size is a
  // fresh mapping whose page faults serialise the threads of ransac_pool.hpp on the process's memory map
  static thread_local std::vector<int> inlI;
  static thread_local std::vector<double> Ds;
  if (inlI.size() < len) { inlI.resize(len); Ds.resize(len); }
  double buffer[96];   // u2f's moment-matrix form needs no len x 9 matrix
  unsigned no_i;
  for (unsigned iter = 0; iter < iters; ++iter) {
    ev.fds(F, Ds.data());
    no_i = 0;
    for (unsigned i = 0; i < len; ++i) {
      if (Ds[i] < ths) { inl[i] = 1; ++no_i; }
      else inl[i] = 0;
    }
    if (no_i < 8) return no_i;
    no_i = 0;
    for (unsigned i = 0; i < len; ++i)
      if (inl[i]) inlI[no_i++] = (int)i;
    if (prof) { const double t_ = rfth_prof_now(); u2f(u, inlI.data(), (int)no_i, F, buffer); prof[0] += rfth_prof_now() - t_; prof[1] += 1; }
    else u2f(u, inlI.data(), (int)no_i, F, buffer);
    ths -= dth;
  }
  ev.fds(F, Ds.data());
  no_i = 0;
  for (unsigned i = 0; i < len; ++i) {
    if (Ds[i] < th) { inl[i] = 1; ++no_i; }
    else inl[i] = 0;
  }
  return no_i;
}

// dual_sample, DegUtils.c:587-626: sA correspondences of uA and sB of uB (shuffles with repetition of
// positions, as the reference does)
static inline void dual_sample(GlibcRand &rng, const double *uA, unsigned lenA, unsigned sA, const double *uB, unsigned lenB, unsigned sB, double *usam) {
  // the reference starts every call from two identity permutations of lenA and lenB entries and swaps sA + sB positions in them:
  // the identity lives across calls per thread (a call of innerFH makes 15 samples out of ~22 000 on-plane correspondences - 1.4 MB
  // of index writes for 150 swaps), and only the touched entries are put back
  static thread_local std::vector<unsigned> ident;
  const unsigned need = lenA > lenB ? lenA : lenB;
  if (ident.size() < need) { const size_t old_n = ident.size(); ident.resize(need); for (size_t i = old_n; i < need; ++i) ident[i] = (unsigned)i; }
  unsigned *ptr = ident.data();
  unsigned touched[64];
  auto draw_from = [&](const double *src, unsigned len, unsigned n, double *dst) {
    unsigned nt = 0;
    for (unsigned pos = 0; pos < n; ++pos) {
      const unsigned idx = (unsigned)rng.next() % len;
      const unsigned t = ptr[pos]; ptr[pos] = ptr[idx]; ptr[idx] = t;
      if (nt + 2 <= 64) { touched[nt++] = pos; touched[nt++] = idx; }
    }
    for (unsigned i = 0; i < n; ++i) std::memcpy(dst + 6 * i, src + 6 * ptr[i], 6 * sizeof(double));
    if (n <= 32) for (unsigned i = 0; i < nt; ++i) ptr[touched[i]] = touched[i];
    else for (unsigned i = 0; i < len; ++i) ptr[i] = i;
  };
  draw_from(uA, lenA, sA, usam);
  draw_from(uB, lenB, sB, usam + 6 * sA);
}

// innerFH, DegUtils.c:476-584: F from sam_sizH on-plane + sam_sizO off-plane correspondences, repCount times.
// The reference's loop is  sample -> 10-point F -> count -> (if the count beats every earlier sample's count) u2Fit -> keep the
// best.  Only the sampling touches the generator, and whether a repetition runs u2Fit depends on the COUNTS of the samples
// alone (max_s), never on what an earlier u2Fit returned.  So a call is cut into: draw() the repCount samples in order;
// eval(r) for every repetition, independent of each other; mark() the record-setting ones; refine(t) those, independent of
// each other; fold() everything in repetition order with the reference's comparisons.  Same operations on the same inputs
// as the one-thread loop, hence the same F and mask; eval / refine of one or of several calls run side by side
// (ransac_pool.hpp).
struct InnerFHJob {
  struct Rep {
    std::vector<double> usam;
    std::vector<unsigned char> v, v2;   // mask of the sample's F, mask after u2Fit
    double aF[9], aF2[9];
    unsigned no_i = 0, no_2 = 0;
    bool refine = false;
    double prof[2] = {0, 0};
  };
  PointEval *ev = nullptr;
  double th = 0;
  unsigned ns = 0, len = 0;
  std::vector<Rep> reps;
  std::vector<int> todo;
  bool want_prof = false;

  void draw(GlibcRand &rng, const double *uH, unsigned lenH, const double *uO, unsigned lenO, PointEval &ev_, double th_, unsigned repCount,
            unsigned sam_sizH, unsigned sam_sizO) {
    ev = &ev_; th = th_; ns = sam_sizH + sam_sizO; len = (unsigned)ev_.len;
    reps.assign(repCount, Rep());
    for (Rep &R : reps) {
      R.usam.resize((size_t)6 * ns);
      dual_sample(rng, uH, lenH, sam_sizH, uO, lenO, sam_sizO, R.usam.data());
    }
  }
  int n_eval() const { return (int)reps.size(); }
  // eval(r) = fit(r) + mask(r); with the counts of all samples made elsewhere (PointEval::count_fds) a repetition's mask is only made
  // when it is refined - fold() reads the mask of a repetition only if its count beat every earlier one's, which is mark()'s rule
  void fit(int r) {
    Rep &R = reps[r];
    std::vector<double> buffer((size_t)9 * ns + 96);
    std::vector<int> allInl(ns);
    for (unsigned i = 0; i < ns; ++i) allInl[i] = (int)i;
    u2f(R.usam.data(), allInl.data(), (int)ns, R.aF, buffer.data());
  }
  void mask(int r) {
    Rep &R = reps[r];
    static thread_local std::vector<double> Ds;   // (see u2Fit)
    if (Ds.size() < len) Ds.resize(len);
    ev->fds(R.aF, Ds.data());
    R.v.resize(len);
    unsigned no_i = 0;
    for (unsigned i = 0; i < len; ++i) {
      const unsigned char in = Ds[i] < th ? 1 : 0;
      R.v[i] = in; no_i += in;
    }
    R.no_i = no_i;
  }
  void eval(int r) { fit(r); mask(r); }
  void mark() {
    unsigned max_s = 0;
    todo.clear();
    for (size_t r = 0; r < reps.size(); ++r)
      if (reps[r].no_i > max_s) { max_s = reps[r].no_i; reps[r].refine = true; todo.push_back((int)r); }
  }
  int n_refine() const { return (int)todo.size(); }
  void refine(int t) {
    Rep &R = reps[todo[t]];
    if (R.v.empty()) mask(todo[t]);                 // (counted elsewhere: same count, now with the mask)
    R.v2 = R.v;
    std::memcpy(R.aF2, R.aF, sizeof(R.aF));
    R.no_2 = u2Fit(*ev, R.aF2, R.v2.data(), th, th * 3, 4, want_prof ? R.prof : nullptr);
  }
  void fold(double *F, unsigned char *inl, double *fit_prof) const {
    for (int i = 0; i < 9; ++i) F[i] = 1;
    for (unsigned i = 0; i < len; ++i) inl[i] = 0;
    unsigned max_i = 0;
    for (const Rep &R : reps) {
      if (max_i < R.no_i) {
        std::memcpy(inl, R.v.data(), len);
        std::memcpy(F, R.aF, sizeof(R.aF));
        max_i = R.no_i;
      }
      if (R.refine) {
        if (max_i < R.no_2) {
          std::memcpy(inl, R.v2.data(), len);
          std::memcpy(F, R.aF2, sizeof(R.aF2));
          max_i = R.no_2;
        }
        if (fit_prof) { fit_prof[0] += R.prof[0]; fit_prof[1] += R.prof[1]; }
      }
    }
  }
};
// runs the eval and refine stages of n drawn jobs: all evaluations side by side, then all refinements side by side
static inline void run_innerFH_jobs(InnerFHJob *const *jobs, int n, bool parallel) {
  auto for_each = [&](int m, const std::function<void(int)> &fn) {
    if (parallel) TaskPool::get().run(m, fn);
    else for (int i = 0; i < m; i++) fn(i);
  };
  std::vector<std::pair<int, int>> work;
  for (int j = 0; j < n; j++)
    for (int r = 0; r < jobs[j]->n_eval(); r++) work.push_back({j, r});
  const double t0_ = rfth_prof_now();
  PointEval *ev0 = n ? jobs[0]->ev : nullptr;
  bool counted_elsewhere = ev0 && (bool)ev0->count_fds;
  for (int j = 1; j < n && counted_elsewhere; j++) counted_elsewhere = jobs[j]->ev == ev0 && jobs[j]->th == jobs[0]->th;
  if (counted_elsewhere) {
    for_each((int)work.size(), [&](int i) { jobs[work[i].first]->fit(work[i].second); });
    std::vector<double> Fs((size_t)9 * work.size());
    std::vector<unsigned> cnt(work.size());
    for (size_t i = 0; i < work.size(); i++) std::memcpy(&Fs[9 * i], jobs[work[i].first]->reps[work[i].second].aF, 9 * sizeof(double));
    ev0->count_fds(Fs.data(), (int)work.size(), jobs[0]->th, cnt.data());
    for (size_t i = 0; i < work.size(); i++) jobs[work[i].first]->reps[work[i].second].no_i = cnt[i];
  } else
    for_each((int)work.size(), [&](int i) { jobs[work[i].first]->eval(work[i].second); });
  const double t1_ = rfth_prof_now();
  g_rfth_prof[14] += t1_ - t0_;
  work.clear();
  for (int j = 0; j < n; j++) {
    jobs[j]->mark();
    for (int t = 0; t < jobs[j]->n_refine(); t++) work.push_back({j, t});
  }
  for_each((int)work.size(), [&](int i) { jobs[work[i].first]->refine(work[i].second); });
  g_rfth_prof[15] += rfth_prof_now() - t1_;
}
static inline void innerFH(GlibcRand &rng, const double *uH, unsigned lenH, const double *uO, unsigned lenO, PointEval &ev,
                           double th, unsigned repCount, unsigned sam_sizH, unsigned sam_sizO, double *F, unsigned char *inl,
                           double *fit_prof = nullptr) {
  InnerFHJob job, *jp = &job;
  job.want_prof = fit_prof != nullptr;
  job.draw(rng, uH, lenH, uO, lenO, ev, th, repCount, sam_sizH, sam_sizO);
  run_innerFH_jobs(&jp, 1, ev.concurrent());
  job.fold(F, inl, fit_prof);
}

// Counts, for k candidate matrices (k x 9), the off-plane correspondences with FDs < limit.
// Supplied by the caller so that the GPU can do it; a null function means "count on the host".
// by_matrix: the candidates as k x 9 matrices made on the host; by_index (preferred when set): as k index pairs (p0, p1) into the
// off-plane set, the matrix F = ([e]x H^T)^T made where the count runs (Ht = H transposed; `candidate` below is its definition).
// begin / end (both or neither): the count of a block of index pairs as two calls with a buffer slot 0 / 1 between them, so that the
// next block can be drawn (and its count started) while this one is counted.
struct PairCounter {
  std::function<void(const double *Fs, int k, unsigned *counts)> by_matrix;
  std::function<void(const unsigned *pairs, int k, const double *Ht, unsigned *counts)> by_index;
  std::function<void(int slot, const unsigned *pairs, int k, const double *Ht)> begin;
  std::function<void(int slot, int k, unsigned *counts)> end;
  explicit operator bool() const { return (bool)by_matrix || (bool)by_index || (bool)begin; }
};

// rFtH, DegUtils.c:233-444: F from the plane homography H plus two off-plane correspondences.
// hinl marks the on-plane correspondences; returns the best inlier count (3 when nothing was found,
// 0 when there is not enough data) and writes F only on improvement.
// `upload_offplane(uN, n)` is called once with the off-plane set before `count` is used.
static inline unsigned rFtH(GlibcRand &rng, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F,
                            const std::function<void(const double *uN, const double *us, unsigned n)> &upload_offplane, const PairCounter &count,
                            PointEval *ev_in = nullptr) {
  PointEval host_ev(u, (int)len);
  PointEval &ev = ev_in ? *ev_in : host_ev;
  const unsigned MAX_SAM = 10000;
  const double conf = .999;
  const unsigned sam_sizH = 6, sam_sizO = 4;
  const double t_setup = rfth_prof_now();
  std::vector<unsigned char> nhinl(len), inl(len);
  unsigned nN = 0, nH = 0;
  {
    std::vector<double> Ds(len);
    ev.hds(H, Ds.data());
    for (unsigned i = 0; i < len; ++i) {
      if (Ds[i] > 100 * th) { nhinl[i] = 1; ++nN; }
      else nhinl[i] = 0;
      if (hinl[i]) ++nH;
    }
  }
  std::vector<double> uN((size_t)6 * nN + 6), us((size_t)6 * nN + 6), uV((size_t)6 * nN + 6), uH((size_t)6 * nH + 6), Ds(nN + 1);
  std::vector<unsigned char> v(nN + 1);
  nN = 0; nH = 0;
  for (unsigned i = 0; i < len; ++i) {
    if (nhinl[i]) {
      std::memcpy(&uN[6 * nN], u + 6 * i, 6 * sizeof(double));
      std::memcpy(&us[6 * nN], u + 6 * i, 3 * sizeof(double));
      us[6 * nN + 3] = H[0] * u[6 * i + 3] + H[3] * u[6 * i + 4] + H[6] * u[6 * i + 5];
      us[6 * nN + 4] = H[1] * u[6 * i + 3] + H[4] * u[6 * i + 4] + H[7] * u[6 * i + 5];
      us[6 * nN + 5] = H[2] * u[6 * i + 3] + H[5] * u[6 * i + 4] + H[8] * u[6 * i + 5];
      ++nN;
    }
    if (hinl[i]) { std::memcpy(&uH[6 * nH], u + 6 * i, 6 * sizeof(double)); ++nH; }
  }
  std::vector<unsigned> ptr(nN);
  for (unsigned i = 0; i < nN; ++i) ptr[i] = i;
  unsigned max_i = 3, m_i = sam_sizO, max_sam = MAX_SAM;
  if (nN < 4 || nH < 6) return 0;
  if (upload_offplane) upload_offplane(uN.data(), us.data(), nN);
  g_rfth_prof[3] = nN;
  g_rfth_prof[10] += rfth_prof_now() - t_setup;

  double Ht[9];
  mat3_tr(Ht, H);
  // candidate of one two-point sample: F = ([e]x H^T)^T with e through the two parallax lines
  auto candidate = [&](unsigned p0, unsigned p1, double *aFt) {
    double c1[3], c2[3], ec[3], sk[9], prod[9];
    crossprod(c1, &us[6 * p0], &us[6 * p0 + 3]);
    crossprod(c2, &us[6 * p1], &us[6 * p1 + 3]);
    crossprod(ec, c1, c2);
    const double ecNorm = std::sqrt(ec[0] * ec[0] + ec[1] * ec[1] + ec[2] * ec[2]);
    ec[0] = ec[0] / ecNorm; ec[1] = ec[1] / ecNorm; ec[2] = ec[2] / ecNorm;
    skew_sym(ec, sk);
    mat3_mul(prod, sk, Ht);
    mat3_tr(aFt, prod);
  };
  auto draw = [&](GlibcRand &g, std::vector<unsigned> &p) {
    for (unsigned pos = 0; pos < 2; ++pos) {
      const unsigned idx = pos + 1 + (unsigned)g.next() % (nN - pos - 1);
      const unsigned t = p[pos]; p[pos] = p[idx]; p[idx] = t;
    }
  };
  // the sample sequence only depends on the generator as long as no candidate triggers the inner
  // estimation (which draws from the same generator), so candidates are produced and counted in
  // blocks; at a trigger the block is rewound to that sample and the tail is redrawn
  const unsigned BLOCK = count ? 2048 : 1;
  std::vector<double> Fs((size_t)9 * BLOCK);
  std::vector<unsigned> cnt(BLOCK), pairs((size_t)2 * BLOCK);
  const bool by_index = (bool)count.by_index;
  // Between two inner estimations the loop only draws and counts, and what an inner estimation hands back to it is max_sam (the
  // budget, which can only shrink) - the generator leaves innerFH in a state that does not depend on its result (150 draws),
  // and m_i is set from the candidate's own count.  So the search runs AHEAD of the estimations, on the budget it knows: it
  // collects up to MAX_JOBS triggers (each with its samples drawn, i.e. with the generator advanced as innerFH advances it),
  // their estimations run side by side, and their results are folded in trigger order.  A trigger that lies beyond the budget
  // an earlier result has set never happened in the one-thread loop: it and everything after it are dropped, and the generator
  // is put where that loop leaves it (the state after the last estimation that did happen + two draws per remaining sample).
  struct LoopState { GlibcRand rng; std::vector<unsigned> ptr; unsigned no_sam, m_i; };
  struct Trigger { unsigned no_sam_at; InnerFHJob job; LoopState after; };
  const size_t MAX_JOBS = 8;
  LoopState done = {rng, ptr, 1, m_i};     // the state behind the last estimation that is known to have happened
  auto leave = [&]() {                       // the one-thread loop's exit from `done` with the current budget
    rng = done.rng;
    if (done.no_sam < 2 * max_sam)
      for (unsigned s = done.no_sam; s < 2 * max_sam; s++) { (void)rng.next(); (void)rng.next(); }
    return max_i;
  };
  const bool parallel = ev.concurrent();
  // With a begin / end counter the blocks are produced ONE AHEAD: while the device counts block A, the host draws block B from the
  // state behind A (as if A held no trigger) and starts its count; a trigger in A throws B away (its count is drained, not read).
  const bool async = (bool)count.begin && (bool)count.end;
  struct Blk { GlibcRand rng0, rng1; std::vector<unsigned> ptr0, ptr1; unsigned nb = 0; int slot = 0; };
  Blk blk[2];
  std::vector<unsigned> pairs2[2];
  if (async) { pairs2[0].resize((size_t)2 * BLOCK); pairs2[1].resize((size_t)2 * BLOCK); }
  std::vector<unsigned> cnt_unused(async ? BLOCK : 0);
  int pending[2] = {0, 0};                                     // blocks whose count has begun and not ended, per slot
  auto issue = [&](const GlibcRand &r0, const std::vector<unsigned> &p0, unsigned no_sam, int slot) {
    Blk &b = blk[slot];
    b.slot = slot; b.rng0 = r0; b.ptr0 = p0;
    b.nb = 2 * max_sam - no_sam;
    if (b.nb > BLOCK) b.nb = BLOCK;
    b.rng1 = r0; b.ptr1 = p0;
    const double t0_ = rfth_prof_now();
    for (unsigned s = 0; s < b.nb; s++) { draw(b.rng1, b.ptr1); pairs2[slot][2 * s] = b.ptr1[0]; pairs2[slot][2 * s + 1] = b.ptr1[1]; }
    g_rfth_prof[0] += rfth_prof_now() - t0_; g_rfth_prof[2] += 1;
    const double t1_ = rfth_prof_now();
    count.begin(slot, pairs2[slot].data(), (int)b.nb, Ht);
    g_rfth_prof[13] += rfth_prof_now() - t1_;
    pending[slot] = 1;
  };
  auto drain = [&]() {                                         // counts nobody will read: waited for, so that their buffers are free
    for (int q = 0; q < 2; q++) if (pending[q]) { count.end(q, (int)blk[q].nb, cnt_unused.data()); pending[q] = 0; }
  };
  while (done.no_sam < 2 * max_sam) {
    LoopState w = done;
    std::vector<std::unique_ptr<Trigger>> trig_list;
    int cur = 0;
    bool have_cur = false;
    while (trig_list.size() < MAX_JOBS && w.no_sam < 2 * max_sam) {
      unsigned nb = 2 * max_sam - w.no_sam;
      if (nb > BLOCK) nb = BLOCK;
      GlibcRand rng0 = w.rng;
      std::vector<unsigned> ptr0 = w.ptr;
      const double tq0 = rfth_prof_now();
      if (async) {
        if (!have_cur) issue(w.rng, w.ptr, w.no_sam, cur);
        Blk &A = blk[cur];
        nb = A.nb; rng0 = A.rng0; ptr0 = A.ptr0;
        bool have_next = false;
        if (w.no_sam + A.nb < 2 * max_sam) { issue(A.rng1, A.ptr1, w.no_sam + A.nb, cur ^ 1); have_next = true; }
        const double tq1 = rfth_prof_now();
        count.end(cur, (int)nb, cnt.data()); pending[cur] = 0;
        g_rfth_prof[1] += rfth_prof_now() - tq1;
        w.rng = A.rng1; w.ptr = A.ptr1;                        // the state behind the block (what the synchronous loop holds here)
        unsigned trig0 = nb;
        for (unsigned s = 0; s < nb; s++)
          if (cnt[s] > w.m_i) { trig0 = s; break; }
        if (trig0 == nb) { w.no_sam += nb; cur ^= 1; have_cur = have_next; continue; }
        drain(); have_cur = false;                             // the block drawn ahead starts behind a sample that is not the last one now
      } else {
      for (unsigned s = 0; s < nb; s++) {
        draw(w.rng, w.ptr);
        if (by_index) { pairs[2 * s] = w.ptr[0]; pairs[2 * s + 1] = w.ptr[1]; }
        else candidate(w.ptr[0], w.ptr[1], &Fs[9 * s]);
      }
      const double tq1 = rfth_prof_now();
      g_rfth_prof[0] += tq1 - tq0; g_rfth_prof[2] += 1;
      if (by_index) { count.by_index(pairs.data(), (int)nb, Ht, cnt.data()); g_rfth_prof[1] += rfth_prof_now() - tq1; }
      else if (count.by_matrix) { count.by_matrix(Fs.data(), (int)nb, cnt.data()); g_rfth_prof[1] += rfth_prof_now() - tq1; }
      else {
        FDs_all(uN.data(), Fs.data(), Ds.data(), (int)nN);
        unsigned c = 0;
        for (unsigned i = 0; i < nN; ++i) if (Ds[i] < th * 2) ++c;
        cnt[0] = c;
      }
      }
      unsigned trig = nb;
      for (unsigned s = 0; s < nb; s++)
        if (cnt[s] > w.m_i) { trig = s; break; }
      if (trig == nb) { w.no_sam += nb; continue; }
      // rewind to the state right after sample `trig`
      if (trig + 1 < nb) {
        w.rng = rng0; w.ptr = ptr0;
        for (unsigned s = 0; s <= trig; s++) draw(w.rng, w.ptr);
      }
      const double t_trig = rfth_prof_now();
      w.no_sam += trig;          // loop variable value while sample `trig` is processed
      double aFt[9];
      candidate(w.ptr[0], w.ptr[1], aFt);
      FDs_all(uN.data(), aFt, Ds.data(), (int)nN);
      unsigned no_i = 0;
      for (unsigned i = 0; i < nN; ++i)
        if (Ds[i] < th * 2) { std::memcpy(&uV[6 * no_i], &uN[6 * i], 6 * sizeof(double)); ++no_i; }
      w.m_i = no_i;
      std::unique_ptr<Trigger> t(new Trigger());
      t->no_sam_at = w.no_sam;
      t->job.want_prof = true;
      t->job.draw(w.rng, uH.data(), nH, uV.data(), no_i, ev, th, 15, sam_sizH, sam_sizO);
      w.no_sam += 1;             // ++no_sam of the for statement
      t->after = w;
      trig_list.push_back(std::move(t));
      g_rfth_prof[11] += rfth_prof_now() - t_trig;
    }
    drain();                                   // (a block drawn ahead of the last trigger or of the budget's end)
    if (trig_list.empty()) return leave();     // the budget ran out without another trigger
    {
      const double t_ = rfth_prof_now();
      std::vector<InnerFHJob *> jobs;
      for (auto &t : trig_list) jobs.push_back(&t->job);
      run_innerFH_jobs(jobs.data(), (int)jobs.size(), parallel);
      g_rfth_prof[4] += rfth_prof_now() - t_;
    }
    g_rfth_prof[9] += 1;
    const double t_fold = rfth_prof_now();
    struct FoldTime { double t0; ~FoldTime() { g_rfth_prof[12] += rfth_prof_now() - t0; } } fold_time{t_fold};
    for (size_t ti = 0; ti < trig_list.size(); ti++) {
      auto &t = trig_list[ti];
      if (!(t->no_sam_at < 2 * max_sam)) { g_rfth_prof[8] += (double)(trig_list.size() - ti); return leave(); }   // an earlier result had ended the loop before this sample
      double aF[9], fp[2] = {0, 0};
      t->job.fold(aF, inl.data(), fp);
      g_rfth_prof[6] += 1; g_rfth_prof[5] += fp[0]; g_rfth_prof[7] += fp[1];
      unsigned ninl = 0;
      for (unsigned i = 0; i < len; ++i) if (inl[i]) ++ninl;
      if (ninl > max_i) {
        max_i = ninl;
        std::memcpy(F, aF, sizeof(aF));
        unsigned maxni = 0;
        for (unsigned i = 0; i < len; ++i) if (inl[i] && nhinl[i]) ++maxni;
        const unsigned ns = (unsigned)nsamples((int)maxni, (int)nN, 2, conf);
        max_sam = max_sam > ns ? ns : max_sam;
      }
      done = std::move(t->after);
    }
  }
  return leave();
}

}  // namespace rs
}  // namespace mods
