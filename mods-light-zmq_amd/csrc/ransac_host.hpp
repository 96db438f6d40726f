// Host-side building blocks of the LO-RANSAC control loop (the GPU scores hypotheses; the
// sequential bookkeeping, the sampler and the local optimisation live here).
//
// Reference behaviour (file:line relative to the reference root, all under degensac/):
//   glibc srand/rand/random            used by rtools.c:12-39, exp_ranH.c:823,850,861-863
//   sample / randsubset / inlidxs / nsamples / truncQuad / scoreLess    rtools.c:12-257
//   lin_hg / lin_hgN / u2h / pinvJ / HDs / HDsSym / HDsSymMax / all_Hori_valid   Htools.c
//   normu / denormH / nullspace / cov_mat / det3     utools.c
//   SuperFastHash / ht*                              hash.c (exp_ranH.c:15 enables __HASHING__)
//   minv / trnm                                      matutls/minv.c, matutls/trnm.c (CCMATH)
//   lap_eig (LAPACK dsyev_)                          lapwrap.c:67-97 -> cyclic Jacobi here
// fp64, one rounding per operation (-ffp-contract=off), same operation order as the C code.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "ransac_simd.hpp"

namespace mods {
namespace rs {

// ---- glibc TYPE_3 additive-feedback generator (srandom_r / random_r), bit exact -------------
struct GlibcRand {
  int32_t r[31];
  int f, b;
  void seed(unsigned int s) {
    if (s == 0) s = 1;
    r[0] = (int32_t)s;
    for (int i = 1; i < 31; i++) {
      const long hi = r[i - 1] / 127773;
      const long lo = r[i - 1] % 127773;
      long word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r[i] = (int32_t)word;
    }
    f = 3; b = 0;
    for (int i = 0; i < 310; i++) (void)next();
  }
  int32_t next() {
    const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    const int32_t out = (int32_t)(v >> 1);
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    return out;
  }
};

struct Score { unsigned I; double J; };

static inline double trunc_quad(double epsilon, double thr) {   // rtools.c:228-236
  if (thr == 0) return 0;
  if (epsilon >= thr * 9 / 4) return 0;
  return 1 - (epsilon / (thr * 9 / 4));
}
static inline bool score_less(const Score &a, const Score &b) { return a.J < b.J; }   // __SCORE__ == SC_M

static inline Score inlidxs(const double *err, int len, double th, int *inl) {   // rtools.c:155-166
  Score s = {0, 0};
  for (int i = 0; i < len; ++i) {
    s.J += trunc_quad(err[i], th);
    if (err[i] <= th) { inl[s.I] = i; ++(s.I); }
  }
  return s;
}

static inline int nsamples(int ninl, int ptNum, int samsiz, double conf) {   // rtools.c:199-225
  double a = 1, b = 1;
  for (int i = 0; i < samsiz; i++) { a *= ninl - i; b *= ptNum - i; }
  a = a / b;
  if (a < 2.2204e-16) return 1000000;
  a = 1 - a;
  if (a < 2.2204e-16) return 1;
  b = std::log(1 - conf) / std::log(a);
  if (b > 1000000) return 1000000;
  return (int)std::ceil(b);
}

// randsubset, rtools.c:24-39: partial shuffle to the back of the pool; returns the offset of the subset
static inline int *randsubset(GlibcRand &g, int *pool, int max_sz, int siz) {
  for (int i = 0; i < siz; i++) {
    const int s = g.next() % (max_sz - i);
    const int j = max_sz - i - 1;
    const int q = pool[s];
    pool[s] = pool[j];
    pool[j] = q;
  }
  return pool + max_sz - siz;
}

// ---- small dense algebra -----------------------------------------------------------------------
static inline double det3(const double *A) {   // utools.c:196-202
  double r = (A[0] * A[4] * A[8] + A[2] * A[3] * A[7] + A[1] * A[5] * A[6]);
  r -= (A[2] * A[4] * A[6] + A[0] * A[5] * A[7] + A[1] * A[3] * A[8]);
  return r;
}

// Gauss-Jordan null space, utools.c:105-169 (row-major n x n, destroys `m`); returns #vectors
static inline int nullspace(double *m, double *ns, int n, int *buffer) {
  int *pnopivot = buffer, nonpivot = 0;
  int *ppivot = buffer + n;
  const double tol = 1e-12;
  int i = 0;
  for (int j = 0; j < n; j++) {
    double pivot = std::fabs(m[n * i + j]);
    int mx = i;
    for (int k = i + 1; k < n; k++) {
      const double t = std::fabs(m[n * k + j]);
      if (pivot < t) { pivot = t; mx = k; }
    }
    if (pivot < tol) {
      *(pnopivot++) = j; nonpivot++;
      for (int k = i; k < n; k++) m[n * k + j] = 0;
    } else {
      *(ppivot++) = j;
      for (int k = j; k < n; k++) { const double t = m[i * n + k]; m[i * n + k] = m[mx * n + k]; m[mx * n + k] = t; }
      pivot = m[i * n + j];
      for (int k = j; k < n; k++) m[i * n + k] /= pivot;
      for (int k = 0; k < i; k++) {
        pivot = -m[k * n + j];
        for (int l = j; l < n; l++) m[k * n + l] += pivot * m[i * n + l];
      }
      for (int k = i + 1; k < n; k++) {
        pivot = m[k * n + j];
        for (int l = j; l < n; l++) m[k * n + l] -= pivot * m[i * n + l];
      }
      i++;
    }
  }
  for (int k = 0; k < nonpivot; k++) {
    const int j = buffer[k];
    for (int l = 0; l < n - nonpivot; l++) ns[k * n + buffer[n + l]] = -m[l * n + j];
    for (int l = 0; l < nonpivot; l++) ns[k * n + buffer[l]] = (j == buffer[l]) ? 1 : 0;
  }
  return nonpivot;
}

static inline void trnm(double *a, int n) {   // in-place transpose, matutls/trnm.c
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++) { const double s = a[i * n + j]; a[i * n + j] = a[j * n + i]; a[j * n + i] = s; }
}

// CCMATH minv (matutls/minv.c) for n = 3: in-place inverse by Crout factorisation with row
// pivoting, same operation order.  Returns -1 on a singular matrix (a is left partially reduced,
// as in the original).
static inline int minv3(double *a) {
  const int n = 3;
  int le[3];
  double q0[3];
  double tq = 0., zr = 1.e-15;
  for (int j = 0; j < n; j++) {
    if (j > 0) {
      for (int i = 0; i < n; i++) q0[i] = a[i * n + j];
      for (int i = 1; i < n; i++) {
        const int lc = i < j ? i : j;
        double t = 0.;
        for (int k = 0; k < lc; k++) t += a[i * n + k] * q0[k];
        q0[i] -= t;
      }
      for (int i = 0; i < n; i++) a[i * n + j] = q0[i];
    }
    double s = std::fabs(a[j * n + j]);
    int lc = j;
    for (int k = j + 1; k < n; k++) {
      const double t = std::fabs(a[k * n + j]);
      if (t > s) { s = t; lc = k; }
    }
    tq = tq > s ? tq : s;
    if (s < zr * tq) return -1;
    le[j] = lc;
    if (lc != j)
      for (int k = 0; k < n; k++) { const double t = a[j * n + k]; a[j * n + k] = a[lc * n + k]; a[lc * n + k] = t; }
    const double t = 1. / a[j * n + j];
    for (int k = j + 1; k < n; k++) a[k * n + j] *= t;
    a[j * n + j] = t;
  }
  for (int j = 1; j < n; j++)
    for (int k = 0; k < j; k++) a[k * n + j] *= a[j * n + j];
  for (int j = 1; j < n; j++) {
    for (int i = 0; i < j; i++) q0[i] = a[i * n + j];
    for (int k = 0; k < j; k++) {
      double t = 0.;
      for (int i = k; i < j; i++) t -= a[k * n + i] * q0[i];
      q0[k] = t;
    }
    for (int i = 0; i < j; i++) a[i * n + j] = q0[i];
  }
  for (int j = n - 2; j >= 0; j--) {
    const int m = n - j - 1;
    for (int i = 0; i < m; i++) q0[i] = a[(j + 1 + i) * n + j];
    int mm = m;
    for (int k = n - 1; k > j; k--) {
      double t = -a[k * n + j];
      for (int i = j + 1; i < k; i++) t -= a[k * n + i] * q0[i - (j + 1)];
      q0[--mm] = t;
    }
    for (int i = 0; i < m; i++) a[(j + 1 + i) * n + j] = q0[i];
  }
  for (int k = 0; k < n - 1; k++) {
    for (int i = 0; i < n; i++) q0[i] = a[i * n + k];
    for (int j = 0; j < n; j++) {
      double t;
      int i;
      if (j > k) { t = 0.; i = j; }
      else { t = q0[j]; i = k + 1; }
      for (; i < n; i++) t += a[j * n + i] * q0[i];
      q0[j] = t;
    }
    for (int i = 0; i < n; i++) a[i * n + k] = q0[i];
  }
  for (int j = n - 2; j >= 0; j--) {
    const int c = le[j];
    for (int k = 0; k < n; k++) { const double t = a[k * n + j]; a[k * n + j] = a[k * n + c]; a[k * n + c] = t; }
  }
  return 0;
}

// Symmetric eigen-decomposition replacing LAPACK dsyev_ ("V","U"), lapwrap.c:67-97: cyclic Jacobi.
// On return `a` (n x n) holds the eigenvectors as dsyev leaves them in column-major storage:
// vector k occupies a[k*n .. k*n+n-1]; ev ascending.  The eigenvector of a simple eigenvalue is
// unique up to sign; sign and last-bit differences against MKL do not change the errors computed
// from it beyond 1e-12 relative.
static inline void sym_eig(double *a, double *ev, int n) {
  std::vector<double> V((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) off += a[p * n + q] * a[p * n + q];
    if (off < 1e-300) break;
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double app = a[p * n + p], aqq = a[q * n + q];
        if (std::fabs(apq) < 1e-18 * std::sqrt(std::fabs(app * aqq)) && sweep > 3) { a[p * n + q] = a[q * n + p] = 0; continue; }
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        rotated = true;
        for (int k = 0; k < n; k++) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
    if (!rotated) break;
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (a[order[j] * n + order[j]] < a[order[i] * n + order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  std::vector<double> out((size_t)n * n);
  for (int k = 0; k < n; k++) {
    ev[k] = a[order[k] * n + order[k]];
    for (int i = 0; i < n; i++) out[(size_t)k * n + i] = V[(size_t)i * n + order[k]];
  }
  std::memcpy(a, out.data(), sizeof(double) * n * n);
}

// ---- homography tools (Htools.c, utools.c) ---------------------------------------------------------
// lin_hg, Htools.c:19-58: 2len x 9 design matrix, column-major (column stride 2*len)
static inline void lin_hg(const double *u, double *dst, const int *inl, int len) {
  const int cs = 2 * len;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    double *p = dst + 2 * i;
    p[0 * cs] = s[3]; p[3 * cs] = s[4]; p[6 * cs] = s[5];
    p[1 * cs] = 0;    p[4 * cs] = 0;    p[7 * cs] = 0;
    p[2 * cs] = -s[0] * s[3]; p[5 * cs] = -s[0] * s[4]; p[8 * cs] = -s[0] * s[5];
    p = dst + 2 * i + 1;
    p[0 * cs] = 0;    p[3 * cs] = 0;    p[6 * cs] = 0;
    p[1 * cs] = s[3]; p[4 * cs] = s[4]; p[7 * cs] = s[5];
    p[2 * cs] = -s[1] * s[3]; p[5 * cs] = -s[1] * s[4]; p[8 * cs] = -s[1] * s[5];
  }
}

// normu, utools.c:7-55
static inline void normu(const double *u, const int *inl, int len, double *A1, double *A2) {
  for (int j = 0; j < 3; j++) { A1[j] = 0; A2[j] = 0; }
  for (int j = 0; j < len; j++) {
    const double *p = u + 6 * inl[j];
    A1[1] += p[0]; A1[2] += p[1];
    A2[1] += p[3]; A2[2] += p[4];
  }
  if (len > 0)
    for (int i = 1; i < 3; i++) { A1[i] /= len; A2[i] /= len; }
  for (int j = 0; j < len; j++) {
    const double *p = u + 6 * inl[j];
    double a = p[0] - A1[1], b = p[1] - A1[2];
    A1[0] += std::sqrt(a * a + b * b);
    a = p[3] - A2[1]; b = p[4] - A2[2];
    A2[0] += std::sqrt(a * a + b * b);
  }
  if (A1[0] != 0) A1[0] = len * std::sqrt(2) / A1[0];
  if (A2[0] != 0) A2[0] = len * std::sqrt(2) / A2[0];
  A1[1] *= -A1[0]; A1[2] *= -A1[0];
  A2[1] *= -A2[0]; A2[2] *= -A2[0];
}

// lin_hgN, Htools.c:60-98: normalised design matrix, row-major (2len x 9)
static inline void lin_hgN(const double *u, double *p, const int *inl, int len, const double *A1, const double *A2) {
  double a[3], b[3];
  a[2] = 1; b[2] = 1;
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    a[0] = s[0] * A1[0] + A1[1];
    a[1] = s[1] * A1[0] + A1[2];
    b[0] = s[3] * A2[0] + A2[1];
    b[1] = s[4] * A2[0] + A2[2];
    double *r0 = p + (size_t)(2 * i) * 9, *r1 = r0 + 9;
    for (int j = 0; j < 3; j++) {
      r0[3 * j] = b[j]; r0[3 * j + 1] = 0; r0[3 * j + 2] = -a[0] * b[j];
      r1[3 * j] = 0; r1[3 * j + 1] = b[j]; r1[3 * j + 2] = -a[1] * b[j];
    }
  }
}

// cov_mat, utools.c:172-185: Cv = Z^T Z; every entry is the sum over the rows IN ROW ORDER (that order is part
// of the result).  The reference walks the rows once per entry; here one walk over the rows feeds all
// siz*(siz+1)/2 running sums at once - the same additions in the same order per entry, 45x fewer passes
// over Z for siz = 9 (the least-squares steps on 10^4 inliers are dominated by this loop).
static inline void cov_mat(double *Cv, const double *Z, int len, int siz) {
  double acc[45];
  if (siz != 9) {     // generic form
    const int lenM = len * siz;
    for (int i = 0; i < siz; i++)
      for (int j = 0; j <= i; j++) {
        double val = 0;
        for (int k = 0; k < lenM; k += siz) val += Z[k + i] * Z[k + j];
        Cv[siz * i + j] = val;
        Cv[i + siz * j] = val;
      }
    return;
  }
  for (int q = 0; q < 45; q++) acc[q] = 0;
  for (int k = 0; k < len; k++) {
    const double *z = Z + (size_t)k * 9;
    int q = 0;
    for (int i = 0; i < 9; i++)
      for (int j = 0; j <= i; j++) acc[q++] += z[i] * z[j];
  }
  int q = 0;
  for (int i = 0; i < 9; i++)
    for (int j = 0; j <= i; j++) { Cv[9 * i + j] = acc[q]; Cv[i + 9 * j] = acc[q]; q++; }
}

// denormH, utools.c:76-98
static inline void denormH(double *F, const double *A1, const double *A2) {
  double r = A2[0], x = A2[1], y = A2[2];
  F[6] += x * F[0] + y * F[3];
  F[7] += x * F[1] + y * F[4];
  F[8] += x * F[2] + y * F[5];
  F[0] *= r; F[1] *= r; F[2] *= r;
  F[3] *= r; F[4] *= r; F[5] *= r;
  r = 1 / A1[0]; x = -A1[1] * r; y = -A1[2] * r;
  for (int i = 0; i < 9; i += 3) {
    F[i] = r * F[i] + x * F[i + 2];
    F[i + 1] = r * F[i + 1] + y * F[i + 2];
  }
}

// Z^T Z of the normalised design matrix without the matrix (lin_hgN + cov_mat, Htools.c:60-98, utools.c:172-185).
// The two rows of correspondence i are
//   r0 = [b0 0 c0  b1 0 c1  b2 0 c2],  r1 = [0 b0 d0  0 b1 d1  0 b2 d2],   c_j = -a0 b_j, d_j = -a1 b_j, b2 = 1
// and cov_mat adds z_p z_q over the rows in order r0(0) r1(0) r0(1) ...  A product with a structural zero is +-0, and
// adding +-0 never changes a running sum that started at +0, so every entry is one of 30 ordered sums:
//   sum b_j b_k (entries (3j,3k) and (3j+1,3k+1)),  sum b_j c_k,  sum b_j d_k,  sum (c_j c_k then d_j d_k),  or exactly +0.
// Same additions in the same order per entry as the reference; 36 products per correspondence instead of 90, no 2len x 9
// buffer.
// sums: bb[6] | cd[6] | bc[9] | bd[9]
static inline void cov_hgN_unfold(const double *sums, double *Cv) {
  const double *bb = sums, *cd = sums + 6, *bc = sums + 12, *bd = sums + 21;
  auto tri = [](int j, int k) { return j >= k ? j * (j + 1) / 2 + k : k * (k + 1) / 2 + j; };
  for (int i = 0; i < 9; i++)
    for (int q = 0; q <= i; q++) {
      const int j = i / 3, al = i % 3, k = q / 3, be = q % 3;
      double v;
      if (al == 2 && be == 2) v = cd[tri(j, k)];
      else if (al == 2) v = be == 0 ? bc[3 * k + j] : bd[3 * k + j];
      else if (be == 2) v = al == 0 ? bc[3 * j + k] : bd[3 * j + k];
      else v = al == be ? bb[tri(j, k)] : 0.0;
      Cv[9 * i + q] = v; Cv[i + 9 * q] = v;
    }
}
// through the host SIMD table (ransac_simd.inc: cov_hg_all - at 8 lanes the 30 sums run side by side in four vectors)
static inline void cov_hgN(const double *u, const int *inl, int len, const double *A1, const double *A2, double *Cv) {
  double sums[30];
  simd_ops()->cov_hg_all(u, inl, len, A1, A2, sums);
  cov_hgN_unfold(sums, Cv);
}
static inline void cov_hgN_scalar(const double *u, const int *inl, int len, const double *A1, const double *A2, double *Cv) {
  double bb[6] = {0, 0, 0, 0, 0, 0}, bc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cd[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < len; i++) {
    const double *s = u + 6 * inl[i];
    const double a0 = s[0] * A1[0] + A1[1], a1 = s[1] * A1[0] + A1[2];
    double b[3], c[3], d[3];
    b[0] = s[3] * A2[0] + A2[1]; b[1] = s[4] * A2[0] + A2[2]; b[2] = 1;
    for (int j = 0; j < 3; j++) { c[j] = -a0 * b[j]; d[j] = -a1 * b[j]; }
    int q = 0;
    for (int j = 0; j < 3; j++)
      for (int k = 0; k <= j; k++, q++) { bb[q] += b[j] * b[k]; cd[q] += c[j] * c[k]; cd[q] += d[j] * d[k]; }
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) { bc[3 * j + k] += b[j] * c[k]; bd[3 * j + k] += b[j] * d[k]; }
  }
  auto tri = [](int j, int k) { return j >= k ? j * (j + 1) / 2 + k : k * (k + 1) / 2 + j; };
  for (int i = 0; i < 9; i++)
    for (int q = 0; q <= i; q++) {
      const int j = i / 3, al = i % 3, k = q / 3, be = q % 3;
      double v;
      if (al == 2 && be == 2) v = cd[tri(j, k)];
      else if (al == 2) v = be == 0 ? bc[3 * k + j] : bd[3 * k + j];
      else if (be == 2) v = al == 0 ? bc[3 * j + k] : bd[3 * j + k];
      else v = al == be ? bb[tri(j, k)] : 0.0;
      Cv[9 * i + q] = v; Cv[i + 9 * q] = v;
    }
}

// u2h, Htools.c:100-132.  `reference_form` runs lin_hgN + cov_mat as written there (`buffer` then holds at least
// 18*len doubles; kept for the self-test that pins cov_hgN to it), otherwise `buffer` is unused.
static inline void u2h(const double *u, const int *inl, int len, double *H, double *buffer, bool reference_form = false) {
  double A1[3], A2[3];
  double V[9 * 9], D[9];
  int nb[2 * 9];
  if (len < 4) return;
  if (len == 4) {
    // The reference fills an 8x9 column-major block (stride 8) and then transposes the buffer as
    // if it were 9x9 (Htools.c:109-110); reproduced on the flat array.
    double Z2[9 * 9];
    std::memset(Z2, 0, sizeof(Z2));
    lin_hg(u, Z2, inl, len);
    trnm(Z2, 9);
    for (int i = 9 * 8; i < 9 * 9; ++i) Z2[i] = 0.0;
    std::memset(V, 0, sizeof(V));
    nullspace(Z2, V, 9, nb);
    std::memcpy(H, V, 9 * sizeof(double));
  } else {
    normu(u, inl, len, A1, A2);
    if (reference_form) {
      lin_hgN(u, buffer, inl, len, A1, A2);
      cov_mat(V, buffer, 2 * len, 9);
    } else cov_hgN(u, inl, len, A1, A2, V);
    sym_eig(V, D, 9);
    std::memcpy(H, V, 9 * sizeof(double));
    denormH(H, A1, A2);
  }
}

static inline void pinvJ(double a, double b, double c, double d, double e, double *pJ) {   // Htools.c:134-158
  const double a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d, e2 = e * e;
  const double c2pd2 = c2 + d2, ab = a * b, de = d * e;
  const double Q = c * (c2pd2 + e2);
  pJ[0] = -b * de + a * (c2 + e2);
  pJ[1] = b * c2pd2 - a * de;
  pJ[2] = Q;
  pJ[3] = -c * (a * d + b * e);
  pJ[4] = d * (b2 + c2) - ab * e;
  pJ[5] = -ab * d + e * (a2 + c2);
  pJ[6] = pJ[3];
  pJ[7] = c * (a2 + b2 + c2);
  const double N = a * pJ[0] + b * pJ[1] + c * pJ[2];
  for (int i = 0; i < 8; i++) pJ[i] /= N;
}

// Sampson error of one correspondence, Htools.c:160-199 (the design-matrix row is rebuilt from u)
static inline double hds_point(const double *u, const double *H) {
  const double z0[9] = {u[3], 0, -u[0] * u[3], u[4], 0, -u[0] * u[4], u[5], 0, -u[0] * u[5]};
  const double z1[9] = {0, u[3], -u[1] * u[3], 0, u[4], -u[1] * u[4], 0, u[5], -u[1] * u[5]};
  double r1 = 0, r2 = 0;
  for (int j = 0; j < 9; j++) { r1 += H[j] * z0[j]; r2 += H[j] * z1[j]; }
  double a = H[0] - H[2] * u[0];
  const double b = H[3] - H[5] * u[0];
  const double c = -H[8] - H[2] * u[3] - H[5] * u[4];
  const double d = H[1] - H[2] * u[1];
  const double e = H[4] - H[5] * u[1];
  double pJ[8];
  pinvJ(a, b, c, d, e, pJ);
  double p = 0;
  for (int j = 0; j < 4; j++) {
    a = pJ[j] * r1 + pJ[j + 4] * r2;
    p += a * a;
  }
  return p;
}

struct SymH { double Hinv[9], H1[9]; };   // Hinv = H^T (as stored), H1 = minv(Hinv)
static inline void sym_prepare(const double *H, SymH *s) {   // Htools.c:206-221
  s->Hinv[0] = H[0]; s->Hinv[1] = H[3]; s->Hinv[2] = H[6];
  s->Hinv[3] = H[1]; s->Hinv[4] = H[4]; s->Hinv[5] = H[7];
  s->Hinv[6] = H[2]; s->Hinv[7] = H[5]; s->Hinv[8] = H[8];
  for (int i = 0; i < 9; i++) s->H1[i] = s->Hinv[i];
  minv3(s->H1);
}
// symmetric transfer error parts d1, d2 of one correspondence, Htools.c:223-240
static inline void hsym_point(const double *u, const SymH *s, double *d1, double *d2) {
  const double *H1 = s->H1, *Hinv = s->Hinv;
  const double a = H1[6] * u[0] + H1[7] * u[1] + H1[8];
  const double b = Hinv[6] * u[3] + Hinv[7] * u[4] + Hinv[8];
  double xa = (H1[0] * u[0] + H1[1] * u[1] + H1[2]) / a;
  double ya = (H1[3] * u[0] + H1[4] * u[1] + H1[5]) / a;
  double xdiff = u[3] - xa, ydiff = u[4] - ya;
  *d1 = xdiff * xdiff + ydiff * ydiff;
  xa = (Hinv[0] * u[3] + Hinv[1] * u[4] + Hinv[2]) / b;
  ya = (Hinv[3] * u[3] + Hinv[4] * u[4] + Hinv[5]) / b;
  xdiff = u[0] - xa; ydiff = u[1] - ya;
  *d2 = xdiff * xdiff + ydiff * ydiff;
}

static inline void crossprod(double *out, const double *a, const double *b) {   // utools.c:187-193, st = 1
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}
static inline int all_Hori_valid(const double *us, const int *idx) {   // Htools.c:545-572
  double p[3], q[3];
  const double *a = us + 6 * idx[0], *b = us + 6 * idx[1], *c = us + 6 * idx[2], *d = us + 6 * idx[3];
  crossprod(p, a, b);
  crossprod(q, a + 3, b + 3);
  if ((p[0] * c[0] + p[1] * c[1] + p[2] * c[2]) * (q[0] * c[3] + q[1] * c[4] + q[2] * c[5]) < 0) return 0;
  if ((p[0] * d[0] + p[1] * d[1] + p[2] * d[2]) * (q[0] * d[3] + q[1] * d[4] + q[2] * d[5]) < 0) return 0;
  crossprod(p, c, d);
  crossprod(q, c + 3, d + 3);
  if ((p[0] * a[0] + p[1] * a[1] + p[2] * a[2]) * (q[0] * a[3] + q[1] * a[4] + q[2] * a[5]) < 0) return 0;
  if ((p[0] * b[0] + p[1] * b[1] + p[2] * b[2]) * (q[0] * b[3] + q[1] * b[4] + q[2] * b[5]) < 0) return 0;
  return 1;
}

// ---- hash of already-optimised inlier sets (hash.c) --------------------------------------------------
static inline uint32_t super_fast_hash(const char *data, int len) {
  auto get16 = [](const char *d) { return (uint32_t)(((uint32_t)((const uint8_t *)d)[1]) << 8) + (uint32_t)((const uint8_t *)d)[0]; };
  uint32_t hash = (uint32_t)len, tmp;
  if (len <= 0 || data == 0) return 0;
  const int rem = len & 3;
  len >>= 2;
  for (; len > 0; len--) {
    hash += get16(data);
    tmp = (get16(data + 2) << 11) ^ hash;
    hash = (hash << 16) ^ tmp;
    data += 4;
    hash += hash >> 11;
  }
  switch (rem) {
    case 3: hash += get16(data); hash ^= hash << 16; hash ^= (uint32_t)(((signed char)data[2]) << 18); hash += hash >> 11; break;
    case 2: hash += get16(data); hash ^= hash << 11; hash += hash >> 17; break;
    case 1: hash += (uint32_t)(signed char)*data; hash ^= hash << 10; hash += hash >> 1;
  }
  hash ^= hash << 3;
  hash += hash >> 5;
  hash ^= hash << 4;
  hash += hash >> 17;
  hash ^= hash << 25;
  hash += hash >> 6;
  return hash;
}

struct HashTable {   // 64 buckets, newest first (hash.c:69-99)
  struct Field { uint32_t hash; int length; int iterID; };
  std::vector<Field> bucket[64];
  void insert(uint32_t hash, int length, int iterID) { bucket[hash % 64].push_back({hash, length, iterID}); }
  int contains(uint32_t hash, int length, int iterID) const {
    const std::vector<Field> &b = bucket[hash % 64];
    for (size_t i = b.size(); i-- > 0;)
      if (b[i].hash == hash && b[i].length == length && b[i].iterID == iterID) return iterID;
    for (size_t i = b.size(); i-- > 0;)
      if (b[i].hash == hash && b[i].length == length) return b[i].iterID;
    return -1;
  }
};

}  // namespace rs
}  // namespace mods
