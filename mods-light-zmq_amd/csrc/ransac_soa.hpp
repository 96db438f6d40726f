// Structure-of-arrays copy of a correspondence list for the host SIMD error functions (ransac_simd.hpp), shared by the
// homography and the fundamental-matrix control loops.
#pragma once
#include "ransac_simd.hpp"
#include <vector>

namespace mods {

// every per-point buffer of a run is padded to n_pad so that whole vectors can be read and written
struct PointsSoA {
  const rs::SimdOps *ops = rs::simd_ops();
  int len = 0, n_pad = 0;
  std::vector<double> store, gains;
  const double *col[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void build(const double *u, int n) {
    len = n; n_pad = (n + rs::SIMD_PAD - 1) / rs::SIMD_PAD * rs::SIMD_PAD;
    store.resize((size_t)5 * n_pad); gains.assign(n_pad, 0.0);
    const int comp[5] = {0, 1, 3, 4, 5};
    for (int c = 0; c < 5; c++) {
      double *dst = store.data() + (size_t)c * n_pad;
      for (int i = 0; i < n_pad; i++) dst[i] = u[(size_t)6 * (i < n ? i : n - 1) + comp[c]];
      col[c] = dst;
    }
  }
};

}  // namespace mods
