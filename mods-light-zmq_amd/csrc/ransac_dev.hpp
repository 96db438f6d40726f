// Device-side error functions of the verification stage, shared by ransac.hip (homography) and
// ransac_f.hip (fundamental matrix; its plane-degeneracy branch scores homographies too).
// fp64, one rounding per operation, same operation order as the C code they restate:
//   pinvJ, HDs            degensac/Htools.c:134-199
//   HDsSym parts          degensac/Htools.c:206-240
//   truncQuad             degensac/rtools.c:228-236
//   FDs / FDsSym / exFDs  degensac/Ftools.c:94-209
#pragma once
#include <hip/hip_runtime.h>

namespace mods {

__device__ __forceinline__ void pinvJ_dev(double a, double b, double c, double d, double e, double *pJ) {
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
#pragma unroll
  for (int i = 0; i < 8; i++) pJ[i] /= N;
}

__device__ __forceinline__ double hds_dev(const double *u, const double *H) {   // HDs, Htools.c:160-199
  const double z0[9] = {u[3], 0, -u[0] * u[3], u[4], 0, -u[0] * u[4], u[5], 0, -u[0] * u[5]};
  const double z1[9] = {0, u[3], -u[1] * u[3], 0, u[4], -u[1] * u[4], 0, u[5], -u[1] * u[5]};
  double r1 = 0, r2 = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) { r1 += H[j] * z0[j]; r2 += H[j] * z1[j]; }
  double a = H[0] - H[2] * u[0];
  const double b = H[3] - H[5] * u[0];
  const double c = -H[8] - H[2] * u[3] - H[5] * u[4];
  const double d = H[1] - H[2] * u[1];
  const double e = H[4] - H[5] * u[1];
  double pJ[8];
  pinvJ_dev(a, b, c, d, e, pJ);
  double p = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    a = pJ[j] * r1 + pJ[j + 4] * r2;
    p += a * a;
  }
  return p;
}

__device__ __forceinline__ void hsym_dev(const double *u, const double *Hinv, const double *H1, double *d1, double *d2) {
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

__device__ __forceinline__ double trunc_quad_dev(double epsilon, double thr) {
  if (thr == 0) return 0;
  if (epsilon >= thr * 9 / 4) return 0;
  return 1 - (epsilon / (thr * 9 / 4));
}


struct FTerms { double r, a, b; };
__device__ __forceinline__ FTerms f_terms(const double *u, const double *F) {   // common part of FDs / FDsSym
  const double rxc = F[0] * u[3] + F[3] * u[4] + F[6];
  const double ryc = F[1] * u[3] + F[4] * u[4] + F[7];
  const double rwc = F[2] * u[3] + F[5] * u[4] + F[8];
  const double r = (u[0] * rxc + u[1] * ryc + rwc);
  const double rx = F[0] * u[0] + F[1] * u[1] + F[2];
  const double ry = F[3] * u[0] + F[4] * u[1] + F[5];
  FTerms t;
  t.r = r;
  t.a = rxc * rxc + ryc * ryc;
  t.b = rx * rx + ry * ry;
  return t;
}
// Sampson error; *wsum (optional) = the denominator (exFDs derives its weight 1/sqrt from it)
__device__ __forceinline__ double fds_from(const double *u, const double *F, double *wsum = nullptr) {
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

}  // namespace mods
