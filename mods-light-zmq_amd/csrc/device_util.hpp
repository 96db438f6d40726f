// Device-side restatements of the small pixel primitives of detectors/helpers.cpp that
// several kernels share.  fp32, one rounding per operation (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include "atan_lut_data.h"

namespace mods {

// interpolateCheckBorders, helpers.cpp:527-549
__host__ __device__ __forceinline__ bool check_borders(int img_w, int img_h, float ofsx, float ofsy, float a11, float a12,
                                              float a21, float a22, int res_w, int res_h) {
  const int width = img_w - 2, height = img_h - 2;
  const float halfWidth = (float)ceil((double)((float)res_w) / 2.0);
  const float halfHeight = (float)ceil((double)((float)res_h) / 2.0);
  const float xs[4] = {-halfWidth, -halfWidth, +halfWidth, +halfWidth};
  const float ys[4] = {-halfHeight, +halfHeight, -halfHeight, +halfHeight};
  bool touch = false;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float imx = ofsx + xs[i] * a11 + ys[i] * a12;
    float imy = ofsy + xs[i] * a21 + ys[i] * a22;
    if (floorf(imx) <= 0 || floorf(imy) <= 0 || ceilf(imx) >= width || ceilf(imy) >= height) touch = true;
  }
  return touch;
}

// A pointer read out of a structure in memory (the pyramid's plane table) has no known address space, so the compiler
// addresses through it with FLAT instructions: slower than global ones, counted on both wait counters, and every load is
// followed by a full s_waitcnt, which serialises the gathers of the keypoint kernels (round 3: 16 flat loads per Baumberg
// iteration, 45 in the NMS kernel).  The round trip through the global address space lets InferAddressSpaces settle it.
template <class T> __device__ __forceinline__ T *as_global(T *p) {
  typedef __attribute__((address_space(1))) T GT;
  GT *g = (GT *)(unsigned long long)p;     // through an integer: a pointer-to-pointer cast pair is folded away before the pass sees it
  return (T *)g;
}

// LDS hand-over between the lanes of ONE wave (LDS executes a wave's instructions in order; this only stops the
// compiler from moving LDS accesses across)
// value of the neighbouring lane (lane - 1 / lane + 1; the first / last lane keeps its own value, as __shfl_up / __shfl_down
// by 1 do) as ONE DPP move (wave_shr:1 / wave_shl:1): the shuffle builtins go through ds_bpermute, i.e. address arithmetic and
// an LDS round trip per value
__device__ __forceinline__ float lane_up1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_down1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sqrtf(x) without the denormal rescue.  hipcc expands a correctly rounded fp32 square root (the default,
// -fhip-fp32-correctly-rounded-divide-sqrt) into: scale x by 2^32 when x < 2^-96, v_sqrt_f32 (1 ulp), one ulp down / up decided by
// two FMA residuals, scale back, and a class test that passes zero and infinity through - 16 vector instructions, 8 of them half
// rate.  This is the SAME sequence without the scaling and the class test (9 instructions): bit-identical to sqrtf for x = +0, every
// x >= 2^-96, +infinity and NaN (the residual tests are false for NaN operands, so 0 stays 0 and infinity stays infinity); for
// 0 < x < 2^-96 the result is some value below 2^-47.  The operand must not be negative (callers pass sums of squares): the
// instruction reads a negative denormal as -0.  tests/test_gpu_describe.py::test_fast_sqrt_is_sqrtf runs all 2^32 operands against
// sqrtf.  Use it where operands below 2^-96 cannot matter (gradient magnitudes that are compared with 1 before use);
// fast_sqrtf_any is exact for every non-negative operand.
__device__ __forceinline__ float fast_sqrtf(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
  const float vp = fmaf(-sd, s, x), vs = fmaf(-su, s, x);
  s = vp <= 0.f ? sd : s;
  s = vs > 0.f ? su : s;
  return s;
}
// exact for every operand: the lanes of a wave take the compiler's full expansion together when one of them holds a positive
// operand below 2^-96 (two more vector instructions and a scalar branch that is never taken on image data)
__device__ __forceinline__ float fast_sqrtf_any(float x) {
  const bool tiny = (unsigned)(__float_as_int(x) - 1) < (unsigned)(0x0f800000 - 1);     // 0 < x < 2^-96 (+0 wraps to 0xffffffff)
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny) != 0ull, 0)) return sqrtf(x);
  return fast_sqrtf(x);
}

// two horizontally adjacent pixels by one 8-byte load (only 4-byte aligned): a gather costs the memory pipeline per lane
// and per instruction, so the four pixels of a bilinear tap are fetched with two loads
struct __attribute__((packed, aligned(4))) PixPair { float a, b; };

// The four pixels of one bilinear tap, loaded without combining them (lets a caller issue the loads
// of several taps before the first use).  valid = false: the tap is 0 (checked branch, outside).
struct TapLoads { float r00, r01, r10, r11, wx, wy; bool valid; };
__device__ __forceinline__ TapLoads tap_load(const float *__restrict__ im, int w, int h, float WX, float WY, bool touch) {
  TapLoads t;
  int x, y;
  if (!touch) { x = (int)WX; y = (int)WY; t.valid = true; }
  else { x = (int)floorf(WX); y = (int)floorf(WY); t.valid = WX >= 0 && WY >= 0 && x < w - 1 && y < h - 1; }
  t.wx = WX - (float)x;
  t.wy = WY - (float)y;
  if (t.valid) {
    const float *Row0 = im + (size_t)y * w + x;
    const PixPair p0 = *(const PixPair *)Row0, p1 = *(const PixPair *)(Row0 + w);
    t.r00 = p0.a; t.r01 = p0.b; t.r10 = p1.a; t.r11 = p1.b;
  } else { t.r00 = t.r01 = t.r10 = t.r11 = 0.f; }
  return t;
}
// Branch-free form of tap_load for the keypoint kernels: the loads are issued whatever the tap is (indices clamped into the
// image), and the checked branch's "outside -> 0" becomes a select in tap_combine.  With a branch per tap the compiler waits for
// every tap's loads before it starts the next one (s_waitcnt vmcnt(0) in front of each conditional block: 25-35 thousand cycles
// per Baumberg iteration went into 12 serial round trips).  floorf == the unchecked branch's (int) cast there: coordinates of
// an untouched window are >= 1.
__device__ __forceinline__ TapLoads tap_load_bf(const float *__restrict__ im, int w, int h, float WX, float WY, bool touch) {
  TapLoads t;
  const int x = (int)floorf(WX), y = (int)floorf(WY);
  t.valid = !touch || (WX >= 0 && WY >= 0 && x < w - 1 && y < h - 1);
  t.wx = WX - (float)x;
  t.wy = WY - (float)y;
  const int xc = min(max(x, 0), w - 2), yc = min(max(y, 0), h - 2);
  const float *Row0 = im + (unsigned)(__mul24(yc, w) + xc);
  const PixPair p0 = *(const PixPair *)Row0, p1 = *(const PixPair *)(Row0 + w);
  t.r00 = p0.a; t.r01 = p0.b; t.r10 = p1.a; t.r11 = p1.b;
  return t;
}
__device__ __forceinline__ float tap_combine(const TapLoads &t) {
  if (!t.valid) return 0.f;
  const float I1 = t.wx * (t.r01 - t.r00) + t.r00;
  return t.wy * (t.wx * (t.r11 - t.r10) + t.r10 - I1) + I1;
}

// A tap of a window that interpolateCheckBorders vouches for (`touch` false: the unchecked branch of interpolate(): (int) casts,
// no border test, no clamp).  That covers the window's own samples only: a lane of a partly filled tile whose sample lies beyond
// the window's edge (`inwin` false; its value is never stored) reads pixel 0 instead.  A third of the VALU work of the checked form
// (tap_load_bf: two floors, four clamps, four compares - all half-rate instructions on gfx950 - go away).
__device__ __forceinline__ TapLoads tap_load_inside(const float *__restrict__ im, int w, float WX, float WY, bool inwin) {
  TapLoads t;
  const int x = (int)WX, y = (int)WY;
  t.valid = true;
  t.wx = WX - (float)x;
  t.wy = WY - (float)y;
  const float *Row0 = im + (inwin ? (unsigned)(__mul24(y, w) + x) : 0u);
  const PixPair p0 = *(const PixPair *)Row0, p1 = *(const PixPair *)(Row0 + w);
  t.r00 = p0.a; t.r01 = p0.b; t.r10 = p1.a; t.r11 = p1.b;
  return t;
}
template <bool TOUCH>
__device__ __forceinline__ TapLoads tap_load_t(const float *__restrict__ im, int w, int h, float WX, float WY, bool inwin) {
  if (TOUCH) return tap_load_bf(im, w, h, WX, WY, true);
  return tap_load_inside(im, w, WX, WY, inwin);
}
template <bool TOUCH> __device__ __forceinline__ float tap_combine_t(const TapLoads &t) {
  const float I1 = t.wx * (t.r01 - t.r00) + t.r00;
  const float v = t.wy * (t.wx * (t.r11 - t.r10) + t.r10 - I1) + I1;
  return (!TOUCH || t.valid) ? v : 0.f;
}

// interpolate(img, x, y, A) over an n x n window (helpers.cpp:551-626), tile by tile: the 64 lanes of a wave sit on an 8 x 8
// block of neighbouring samples, wave `wv` of `nw` takes the tile rows wv, wv + nw, ...  A gather costs the memory pipeline
// per cache line that its lanes touch (tools/ubench/gather.hip: ~4 cycles per line, 266 cycles when every lane has its own
// line - which is what a contiguous run of samples per lane gives - 42 for an 8 x 8 pixel block), and the 64 taps of a tile lie
// on about nine image rows.  The reference's coordinates are sequential fp32 sums (row starts: += a12 / a22 per row, then
// += a11 / a21 per column): a lane walks its rows column by column, eight additions between two of its taps.
// store(row, col, value) is called for every sample of the window.
// The rows [row_begin, row_end) of the window are sampled (all columns).  TOUCH = whether the window touches the image border
// (interpolateCheckBorders): the same for every lane, so the two forms of the tap are two instantiations behind one scalar branch.
template <bool WIDE, bool TOUCH, class Store>
__device__ __forceinline__ void sample_tiles_rows_t(const float *__restrict__ img, int w, int h, float fx, float fy, float a11, float a12,
                                                    float a21, float a22, int n, int row_begin, int row_end, int wv, int nw, Store store) {
  // tile = TC columns x TR rows of neighbouring samples (round 6 measured 4 x 16 tiles - half the column steps between a lane's taps,
  // twice the image rows per gather: no change in any kernel, profiles/r06_describe_experiments.log)
  constexpr int TC = 8, TR = 64 / TC;
  const int half = n / 2;
  const int lane = threadIdx.x & 63, tcol = lane & (TC - 1), trow = lane / TC;
  float rx = fx - (float)half * a12;
  float ry = fy - (float)half * a22;
  // the row steps all lanes share as a scalar-counted loop, the lane's own 0..TR-1 as selects (a loop with a per-lane trip count costs
  // five vector instructions per step)
  for (int q = __builtin_amdgcn_readfirstlane(row_begin + wv * TR); q > 0; q--) { rx += a12; ry += a22; }
#pragma unroll
  for (int q = 0; q < TR - 1; q++) { const bool m = q < trow; const float nx = rx + a12, ny = ry + a22; rx = m ? nx : rx; ry = m ? ny : ry; }
  for (int r0 = row_begin + wv * TR; r0 < row_end; r0 += nw * TR) {
    const int row = r0 + trow;
    float WX = rx - (float)half * a11;
    float WY = ry - (float)half * a21;
#pragma unroll
    for (int q = 0; q < TC - 1; q++) { const bool m = q < tcol; const float nx = WX + a11, ny = WY + a21; WX = m ? nx : WX; WY = m ? ny : WY; }
    // column tiles in batches: eight per batch while at least eight remain (wide windows only), four while three or more
    // remain, then two: a wave waits for every batch, so the number of batches per row of tiles is what its time follows
    int c0 = 0;
    for (; WIDE && c0 + 8 * TC <= n; c0 += 8 * TC) {
      TapLoads t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        t[u] = tap_load_t<TOUCH>(img, w, h, WX, WY, row < row_end);
#pragma unroll
        for (int q = 0; q < TC; q++) { WX += a11; WY += a21; }
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (row < row_end) store(row, c0 + TC * u + tcol, tap_combine_t<TOUCH>(t[u]));
    }
    for (; c0 + 2 * TC < n; c0 += 4 * TC) {              // four column tiles per batch while three or more remain (8 loads in flight)
      TapLoads t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        t[u] = tap_load_t<TOUCH>(img, w, h, WX, WY, row < row_end && c0 + TC * u + tcol < n);
#pragma unroll
        for (int q = 0; q < TC; q++) { WX += a11; WY += a21; }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int col = c0 + TC * u + tcol;
        if (row < row_end && col < n) store(row, col, tap_combine_t<TOUCH>(t[u]));
      }
    }
    for (; c0 < n; c0 += 2 * TC) {
      TapLoads t[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        t[u] = tap_load_t<TOUCH>(img, w, h, WX, WY, row < row_end && c0 + TC * u + tcol < n);
#pragma unroll
        for (int q = 0; q < TC; q++) { WX += a11; WY += a21; }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int col = c0 + TC * u + tcol;
        if (row < row_end && col < n) store(row, col, tap_combine_t<TOUCH>(t[u]));
      }
    }
    if (r0 + nw * TR >= row_end) break;          // (no walk behind the wave's last tile row)
    for (int q = nw * TR; q > 0; q--) { rx += a12; ry += a22; }
  }
}
template <bool WIDE, class Store>
__device__ __forceinline__ void sample_tiles_rows(const float *__restrict__ img, int w, int h, float fx, float fy, float a11, float a12,
                                                  float a21, float a22, int n, int row_begin, int row_end, int wv, int nw, Store store) {
  // (every lane evaluates the same expression: the first lane's value makes the branch a scalar one)
  const bool touch = __builtin_amdgcn_readfirstlane((int)check_borders(w, h, fx, fy, a11, a12, a21, a22, n, n)) != 0;
  if (touch) sample_tiles_rows_t<WIDE, true>(img, w, h, fx, fy, a11, a12, a21, a22, n, row_begin, row_end, wv, nw, store);
  else sample_tiles_rows_t<WIDE, false>(img, w, h, fx, fy, a11, a12, a21, a22, n, row_begin, row_end, wv, nw, store);
}
template <class Store>
__device__ __forceinline__ void sample_tiles(const float *__restrict__ img, int w, int h, float fx, float fy, float a11, float a12,
                                             float a21, float a22, int n, int wv, int nw, Store store) {
  sample_tiles_rows<false>(img, w, h, fx, fy, a11, a12, a21, a22, n, 0, n, wv, nw, store);
}

// atan2LUTff, helpers.cpp:160-207.  The octant constants are float, the table double: each
// +/- is a double operation rounded to float on return.
__device__ const double g_atan_lut[256] = MODS_ATAN_LUT_INIT;

// Branch-free form of atan2LUTff (helpers.cpp:160-207).  Every branch of the reference returns
// (float)(B + s * L[(int)(255.f * num / den)]) with (num, den) = (min, max) of |x|, |y| chosen by |x| > |y|, and (B, s) fixed by
// the signs and that comparison: 8 cases ("octants") x 256 table entries.  atan2_lut_sel picks the case and the index;
// atan2_lut_case evaluates a case exactly as its branch does (used to tabulate functions of the result per (case, index)).
struct AtanSel { int oct, idx; bool zero; };   // zero: the reference's `x == 0` exit (returns 0.f)
__device__ __forceinline__ AtanSel atan2_lut_sel(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const bool xp = x > 0.f, yp = y > 0.f, first = ax > ay;
  const float num = first ? ay : ax, den = first ? ax : ay;
  AtanSel r;
  r.zero = !xp && !yp && !first && x == 0.f;
  r.oct = (xp ? 0 : 4) | (yp ? 0 : 2) | (first ? 0 : 1);
  const int idx = (int)(255.f * num / den);
  r.idx = r.zero ? 0 : idx;
  return r;
}
__device__ __forceinline__ float atan2_lut_case(int oct, double L) {
  const float PI_2f = 1.57079632679489661923f;
  const float PIf = 3.14159265358979323846f;
  switch (oct) {
    case 0: return (float)L;                              // x > 0, y > 0, x > y
    case 1: return (float)((double)PI_2f - L);
    case 2: return (float)(-L);                           // x > 0, y <= 0, x > |y|
    case 3: return (float)((double)(-PI_2f) + L);
    case 4: return (float)((double)PIf - L);              // x <= 0, y > 0, |x| > y
    case 5: return (float)((double)PI_2f + L);
    case 6: return (float)((double)(-PIf) + L);           // x <= 0, y <= 0, |x| > |y|
    default: return (float)((double)(-PI_2f) - L);
  }
}

// table pointer variant (e.g. an LDS copy of the table)
__device__ __forceinline__ float atan2_lut_ff_t(float y, float x, const double *__restrict__ L) {
  const float PI_2f = 1.57079632679489661923f;
  const float PIf = 3.14159265358979323846f;
  if (x > 0.f) {
    if (y > 0.f) {
      if (x > y) return (float)L[(int)(255.f * y / x)];
      return (float)((double)PI_2f - L[(int)(255 * x / y)]);
    } else {
      float absy = fabsf(y);
      if (x > absy) return (float)(-L[(int)(255.f * absy / x)]);
      return (float)((double)(-PI_2f) + L[(int)(255.f * x / absy)]);
    }
  } else if (y > 0.f) {
    float absx = fabsf(x);
    if (absx > y) return (float)((double)PIf - L[(int)(255.f * y / absx)]);
    return (float)((double)PI_2f + L[(int)(255.f * absx / y)]);
  } else {
    float absx = fabsf(x), absy = fabsf(y);
    if (absx > absy) return (float)((double)(-PIf) + L[(int)(255.f * absy / absx)]);
    if (x == 0.f) return 0.f;
    return (float)((double)(-PI_2f) - L[(int)(255.f * absx / absy)]);
  }
}

}  // namespace mods
