// Host SIMD forms of the LO step's error functions: one table per lane count, chosen once from the CPU's features.
#pragma once
namespace mods { namespace rs {
struct SimdOps {
  int lanes;
  // soa = {u0, u1, u3, u4, u5}, each n_pad doubles (n_pad a multiple of `lanes`, tail filled with any valid point)
  void (*hds_all)(const double *const *soa, int n_pad, const double *H, double *d);
  void (*hsym_all)(const double *const *soa, int n_pad, const double *H1, const double *Hinv, int mode, double *d);
  void (*gains_all)(const double *err, int n_pad, double lim, double *g);
  // FDs (mode 0), FDsSym (1), exFDs (2: + weight 1/sqrt(w)), exFDsSym (3: + weight a b/(a+b)); w may be null for modes 0, 1
  void (*fds_all)(const double *const *soa, int n_pad, const double *F, int mode, double *p, double *w);
  // the 45 ordered sums of the least-squares F's moment matrix (lower triangle, row-major); w may be null
  void (*cov_fm_all)(const double *u, const int *inl, const double *w, int len, const double *A1, const double *A2, double *acc45);
  // homography LO step: the 30 ordered sums of the moment matrix, bb[6] | cd[6] | bc[9] | bd[9] (rs::cov_hgN)
  void (*cov_hg_all)(const double *u, const int *inl, int len, const double *A1, const double *A2, double *sums30);
  // d1 and d2 of hsym_point, not combined; soa = {x1, y1, x2, y2}
  void (*hsym_both_all)(const double *const *soa, int n_pad, const double *H1, const double *Hinv, double *d1, double *d2);
};
const SimdOps *simd_ops();                 // widest table this CPU runs
const SimdOps *simd_ops_lanes(int lanes);  // 1, 4 (AVX2) or 8 (AVX-512F); nullptr when the CPU lacks it
enum { SIMD_PAD = 8 };                     // buffers are padded to a multiple of this many doubles
}}
