// Deterministic scalar math shared by host and device code of libmodsgpu.
//
// The reference evaluates expf / powf / cos / sin through libm at four places on the hot
// path (Gaussian taps and masks, keypoint scale pyramid.cpp:392, orientation rotation
// synth-detection.cpp:1095).  libm is host dependent in the last bit and does not exist on
// the device, so these are evaluated with fixed sequences of IEEE +,-,*,/ (compile with
// -ffp-contract=off; fp32/fp64 divide and sqrt are correctly rounded on gfx950 with hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).  Results are bit-identical on host and
// device and agree with a correctly rounded libm to <1e-15 relative.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define MODS_HD __host__ __device__ __forceinline__

namespace mods {

// 2^k * p for |k| < 1000 using exact power-of-two multiplications.
MODS_HD double scale_pow2(double p, int k) {
  // build 2^k from the exponent field; k in [-1022, 1023]
  long long bits = (long long)(k + 1023) << 52;
  double f;
  memcpy(&f, &bits, sizeof(f));
  return p * f;
}

// exp(x): x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor in Horner form.
MODS_HD double det_exp(double x) {
  const double INV_LN2 = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double kf = floor(x * INV_LN2 + 0.5);
  double r = (x - kf * LN2_HI) - kf * LN2_LO;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return scale_pow2(p, (int)kf);
}

MODS_HD float det_expf(float x) { return (float)det_exp((double)x); }

// powf(2.0f, x)
MODS_HD float det_pow2f(float x) {
  const double LN2 = 0.69314718055994530942;
  return (float)det_exp((double)x * LN2);
}

// sin, cos: a = k pi/2 + r, |r| <= pi/4, Taylor degree 17 / 16.
MODS_HD void det_sincos(double a, double *s, double *c) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  double kf = floor(a * TWO_OVER_PI + 0.5);
  double r = (a - kf * PIO2_HI) - kf * PIO2_LO;
  double r2 = r * r;
  double ps = -1.0 / 355687428096000.0;
  ps = ps * r2 + 1.0 / 1307674368000.0;
  ps = ps * r2 - 1.0 / 6227020800.0;
  ps = ps * r2 + 1.0 / 39916800.0;
  ps = ps * r2 - 1.0 / 362880.0;
  ps = ps * r2 + 1.0 / 5040.0;
  ps = ps * r2 - 1.0 / 120.0;
  ps = ps * r2 + 1.0 / 6.0;
  double sr = r - r * r2 * ps;
  double pc = 1.0 / 20922789888000.0;
  pc = pc * r2 - 1.0 / 87178291200.0;
  pc = pc * r2 + 1.0 / 479001600.0;
  pc = pc * r2 - 1.0 / 3628800.0;
  pc = pc * r2 + 1.0 / 40320.0;
  pc = pc * r2 - 1.0 / 720.0;
  pc = pc * r2 + 1.0 / 24.0;
  pc = pc * r2 - 0.5;
  double cr = 1.0 + r2 * pc;
  int q = (int)((long long)kf & 3LL);
  if (q == 0) { *s = sr; *c = cr; }
  else if (q == 1) { *s = cr; *c = -sr; }
  else if (q == 2) { *s = -sr; *c = -cr; }
  else { *s = -cr; *c = sr; }
}

}  // namespace mods
