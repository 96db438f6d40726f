#include "ransac_simd.hpp"
namespace mods { namespace rs {
#define DECL(NS) namespace NS { \
  void hds_all(const double *const *, int, const double *, double *); \
  void hsym_all(const double *const *, int, const double *, const double *, int, double *); \
  void gains_all(const double *, int, double, double *); \
  void fds_all(const double *const *, int, const double *, int, double *, double *); \
  void cov_fm_all(const double *, const int *, const double *, int, const double *, const double *, double *); \
  void cov_hg_all(const double *, const int *, int, const double *, const double *, double *); \
  void hsym_both_all(const double *const *, int, const double *, const double *, double *, double *); }
DECL(simd1) DECL(simd4) DECL(simd8)
#undef DECL
static const SimdOps k_ops1 = {1, simd1::hds_all, simd1::hsym_all, simd1::gains_all, simd1::fds_all, simd1::cov_fm_all, simd1::cov_hg_all, simd1::hsym_both_all};
static const SimdOps k_ops4 = {4, simd4::hds_all, simd4::hsym_all, simd4::gains_all, simd4::fds_all, simd4::cov_fm_all, simd1::cov_hg_all, simd4::hsym_both_all};   // (the scalar form of cov_hg_all is slower when built with -mavx2: the baseline build serves 4 lanes too)
static const SimdOps k_ops8 = {8, simd8::hds_all, simd8::hsym_all, simd8::gains_all, simd8::fds_all, simd8::cov_fm_all, simd8::cov_hg_all, simd8::hsym_both_all};
const SimdOps *simd_ops_lanes(int lanes) {
  if (lanes == 1) return &k_ops1;
  if (lanes == 4) return __builtin_cpu_supports("avx2") ? &k_ops4 : nullptr;
  if (lanes == 8) return __builtin_cpu_supports("avx512f") ? &k_ops8 : nullptr;
  return nullptr;
}
const SimdOps *simd_ops() {
  static const SimdOps *best = [] {
    const SimdOps *o = simd_ops_lanes(8);
    if (!o) o = simd_ops_lanes(4);
    return o ? o : &k_ops1;
  }();
  return best;
}
}}
