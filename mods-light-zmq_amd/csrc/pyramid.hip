// Hessian scale-space pyramid on gfx950: separable Gaussian blur (row + column pass fused
// through an LDS tile), 3x3 Hessian response, 2x decimation.
//
// Reference behaviour (file:line relative to the reference root):
//   gaussianBlur / gaussianBlurInplace   detectors/helpers.cpp:717-731 (cv::GaussianBlur, REPLICATE)
//   HessianResponse                      detectors/affinedetectors/pyramid.cpp:196-254
//   octave loop, sigma schedule          pyramid.cpp:428-529
//   cv::resize(.., 0.5, 0.5, LINEAR)     pyramid.cpp:476
// Arithmetic contract (shared with the CPU oracle): fp32, built with -ffp-contract=off so that every fusion is
// written out.  cv::GaussianBlur as an FMA build of OpenCV evaluates it (the variant that reproduces the reference's
// README counts exactly, tools/readme_count_hunt.py): row pass s = k[0]*S[0]; s = fma(k[j], S[j], s) left to right
// (ksize <= 5, which the pyramid never uses: centre tap, then s = fma(S[-j] + S[j], k[r+j], s)),
// column pass s = k[r]*T[y]; s = fma(k[r+j], T[y+j] + T[y-j], s) for j = 1..r.
#include "common.hpp"
#include "detmath.hpp"
#include "device_util.hpp"
#include <algorithm>
#include <cmath>

namespace mods {

// ---------------------------------------------------------------------------------------
// host-side table builders
// ---------------------------------------------------------------------------------------
int gauss_ksize(float sigma) {                    // helpers.cpp:720-721
  int size = (int)(2.0 * 3.0 * sigma + 1.0);
  if (size % 2 == 0) size++;
  return size;
}

// OpenCV getGaussianKernel(n, sigma, CV_32F): double exp, float taps, normalised by the double
// sum of the float taps.
void gauss_kernel_host(int n, double sigma, float *out) {
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    out[i] = (float)det_exp(scale2X * x * x);
    sum += out[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) out[i] = (float)(out[i] * sum);
}

void gauss_mask_host(int size, float *mask) {     // computeGaussMask, helpers.cpp:411-440
  int halfSize = size >> 1;
  float scale = float(halfSize) / 3.0f;
  float scale2 = -2.0f * scale * scale;
  std::vector<float> tmp(halfSize + 1);
  for (int i = 0; i <= halfSize; i++) tmp[i] = det_expf(float(i * i) / scale2);
  int endSize = int(ceil(scale * 5.0f) - halfSize);
  for (int i = 1; i < endSize; i++)
    tmp[halfSize - i] += det_expf(float((i + halfSize) * (i + halfSize)) / scale2);
  for (int i = 0; i <= halfSize; i++)
    for (int j = 0; j <= halfSize; j++) {
      float v = tmp[i] * tmp[j];
      mask[(i + halfSize) * size + (-j + halfSize)] = v;
      mask[(-i + halfSize) * size + (j + halfSize)] = v;
      mask[(i + halfSize) * size + (j + halfSize)] = v;
      mask[(-i + halfSize) * size + (-j + halfSize)] = v;
    }
}

void circular_gauss_mask_host(int size, float sigma, float *mask) {   // helpers.cpp:442-461
  int halfSize = size >> 1;
  float r2 = float(halfSize * halfSize);
  float sigma2 = (sigma == 0) ? 0.9f * r2 : 2 * sigma * sigma;
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) {
      float disq = float((i - halfSize) * (i - halfSize) + (j - halfSize) * (j - halfSize));
      mask[i * size + j] = (disq < r2) ? det_expf(-disq / sigma2) : 0;
    }
}

void resize_half_dims(int w, int h, int *dw, int *dh) {   // cvRound(size * 0.5): half to even
  auto rnd = [](double v) {
    double fl = floor(v), d = v - fl;
    if (d > 0.5) return (int)fl + 1;
    if (d < 0.5) return (int)fl;
    return (((long long)fl) & 1LL) ? (int)fl + 1 : (int)fl;
  };
  *dw = rnd(w * 0.5);
  *dh = rnd(h * 0.5);
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
constexpr int BLUR_TW = 64;   // output tile width  (one wave-wide row)
constexpr int BLUR_TH = 32;   // output tile height

// Fused separable blur.  grid = (ceil(w/64), ceil(h/32), n_img), block = 256.
// LDS: input tile (TH+2r) x (TW+2r), row-pass tile (TH+2r) x TW, taps.
__global__ __launch_bounds__(256) void gauss_blur_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                         int w, int h, const float *__restrict__ taps, int n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int r = n >> 1;
  const int IW = BLUR_TW + 2 * r;
  const int IH = BLUR_TH + 2 * r;
  float *s_in = smem;
  float *s_row = smem + IH * IW;
  float *s_tap = s_row + IH * BLUR_TW;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)w * h;
  src += plane * blockIdx.z;
  dst += plane * blockIdx.z;
  const int x0 = blockIdx.x * BLUR_TW;
  const int y0 = blockIdx.y * BLUR_TH;

  if (tid < n) s_tap[tid] = taps[tid];
  // stage the input tile, BORDER_REPLICATE = clamp
  for (int ly = tid / 64; ly < IH; ly += 4) {
    int gy = y0 - r + ly;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    const float *row = src + (size_t)gy * w;
    for (int lx = tid & 63; lx < IW; lx += 64) {
      int gx = x0 - r + lx;
      gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
      s_in[ly * IW + lx] = row[gx];
    }
  }
  __syncthreads();
  // row pass: lane = column, taps left to right
  {
    const int lx = tid & 63;
    for (int ly = tid / 64; ly < IH; ly += 4) {
      const float *p = s_in + ly * IW + lx;
      float s;
      if (n <= 5) {             // SymmRowSmallFilter order (ksize <= 5)
        s = p[r] * s_tap[r];
        for (int j = 1; j <= r; j++) s = fmaf(p[r - j] + p[r + j], s_tap[r + j], s);
      } else {
        s = s_tap[0] * p[0];
        for (int j = 1; j < n; j++) s = fmaf(s_tap[j], p[j], s);
      }
      s_row[ly * BLUR_TW + lx] = s;
    }
  }
  __syncthreads();
  // column pass: lane = column, each thread walks 8 rows
  {
    const int lx = tid & 63;
    const int gx = x0 + lx;
    const int ty = tid / 64;
    if (gx < w) {
      for (int k = 0; k < BLUR_TH / 4; k++) {
        const int ly = ty * (BLUR_TH / 4) + k;
        const int gy = y0 + ly;
        if (gy >= h) break;
        const float *p = s_row + (ly + r) * BLUR_TW + lx;
        float s = s_tap[r] * p[0];
        for (int j = 1; j <= r; j++) s = fmaf(s_tap[r + j], p[j * BLUR_TW] + p[-j * BLUR_TW], s);
        dst[(size_t)gy * w + gx] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Register-blocked blur for radii 1..8 (every blur of the pyramid: ksize 9..15).  Tile = 128 x 64
// outputs per 256-thread block.  Row pass: a thread produces 4 adjacent outputs from a window of
// aligned float4 global loads held in registers (taps are kernel arguments -> SGPRs, loops fully
// unrolled), results go to LDS as float4.  Column pass: float4 LDS reads, centre tap then symmetric
// pairs.  Same arithmetic order as gauss_blur_kernel.
// ---------------------------------------------------------------------------------------
struct BlurTaps { float t[17]; };
// overlapping tiles of the fused blur + response kernel: origins step by `step`, a tile blurs step + lap pixels
// (the tile that owns response row / column n - 1, the last of the frame, is the last one: origins up to n - 2)
__host__ __device__ static inline int blur_resp_tiles(int n, int step, int lap) { (void)lap; return n > 2 ? (n - 1 + step - 1) / step : 1; }
constexpr int FB_TW = 128;   // tile width; the tile height is a template parameter (64: large planes, 16: small planes,
                             // where a short per-thread row chain matters more than halo reuse)

// RESP = true additionally writes the Hessian response of the blurred plane (pyramid.cpp:196-254) to `resp`: the blurred tile
// goes back to LDS after the column pass and a third phase makes the 3x3 stencil from it, so the blurred plane is not read
// again from HBM (12 B/px per level instead of 8 + 8).  The stencil needs a 1-px ring of blurred values around the pixels a
// workgroup answers for, so RESP tiles overlap: tile origins step by (FB_TW - 4, FB_TH - 2), a workgroup still blurs
// FB_TW x FB_TH pixels (the overlap is blurred, and stored, twice with identical values) and owns the response of the
// (FB_TW - 4) x (FB_TH - 2) pixels starting at (x0 + 1, y0 + 1).
// One 16-byte store per lane, at any 4-byte aligned address.  hipcc turns `if (whole) *(float4 *)d = s; else <guarded scalar
// stores>` into a 12-byte + a 4-byte store (it sinks the common scalar stores of the two branches), and a float4 of alignment 4
// likewise; the fused kernel's tail is store-issue bound (switching its stores off takes it from 0.75 to 0.40 ms per batch), so
// the instruction is written out.  gfx950 global memory accesses only need dword alignment.
// The `s_nop 1` belongs to the store: a VMEM store of more than 8 bytes reads its data registers late, and a VALU write to them
// within the next 2 wait states corrupts the last dword(s) (the compiler inserts this wait for its own stores, it does not look
// inside inline assembly; without it the .w of a float4 was occasionally the next iteration's value).
// The address is a wave-uniform base (`base`: an SGPR pair) + the lane's BYTE offset in one VGPR (no 64-bit address arithmetic,
// half the address registers).  NT = the streaming (non-temporal) form.
template <bool NT>
__device__ __forceinline__ void store_f4_at(float *base, unsigned byte_off, float x, float y, float z, float w) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f q = {x, y, z, w};
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(byte_off), "v"(q), "s"(base) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(byte_off), "v"(q), "s"(base) : "memory");
}

// Workgroups per CU the register allocation aims at (second launch bound).  The kernel overlaps its phases - window loads, row
// pass, column pass out of LDS, stores - only across workgroups (barriers separate them inside one).  Round 5 (profiles/
// r05_blur_variants.log): 32-bit offsets against the plane pointer instead of 64-bit addresses took 9-10 registers off every
// instantiation (R = 7: 98 -> 88); holding the large radii to 80 registers (6 workgroups) makes the compiler spill and costs 35 %
// (107 against 80 us), so they get the budget of 5.
#ifndef BLUR_OCC_LO
#define BLUR_OCC_LO 7    // R <= 4
#endif
#ifndef BLUR_OCC_HI
#define BLUR_OCC_HI 5    // R >= 5
#endif
#ifndef BLUR_SWZ
#define BLUR_SWZ 1
#endif
#ifndef BLUR_COLJ
#define BLUR_COLJ 1
#endif
#ifndef BLUR_COLJ_MIN_R
#define BLUR_COLJ_MIN_R 1
#endif
#ifndef BLUR_COLJ_MIN_TH
#define BLUR_COLJ_MIN_TH 32   // the outward column pass pays with 4 outputs per thread; the 16-row tiles (2 outputs) read their 2R + 2 rows up front
#endif
template <int R, int FB_TH, int OV, bool RESP>
__global__ __launch_bounds__(256, (R <= 4 ? BLUR_OCC_LO : BLUR_OCC_HI)) void gauss_blur_fast_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                              BlurTaps taps, float *__restrict__ resp, float norm2) {
  constexpr int N = 2 * R + 1;
  constexpr int R4 = (R + 3) / 4;            // float4s on each side of the outputs
  constexpr int NV = 2 * R4 + OV;            // float4s in the register window
  constexpr int NO = 4 * OV;                 // adjacent outputs per thread and strip row
  constexpr int D = 4 * R4 - R;              // window offset of tap 0 for output 0
  constexpr int ROWS = FB_TH + 2 * R;
  constexpr int TPR = FB_TW / NO;            // threads per strip row
  constexpr int RPS = 256 / TPR;             // strip rows per step
  extern __shared__ __attribute__((aligned(16))) float smem[];   // ROWS x FB_TW row-pass results
  const int tid = threadIdx.x;
  const size_t plane = (size_t)w * h;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2), so every XCD gets a
  // contiguous band of tiles in (image, tile row, tile column) order and the halo rows shared by
  // vertically adjacent tiles are served by one L2 instead of being fetched by two.
  constexpr int SX = RESP ? FB_TW - 4 : FB_TW, SY = RESP ? FB_TH - 2 : FB_TH;   // tile origin steps
  const int tiles_x = RESP ? blur_resp_tiles(w, SX, 4) : (w + FB_TW - 1) / FB_TW;
  const int tiles_y = RESP ? blur_resp_tiles(h, SY, 2) : (h + FB_TH - 1) / FB_TH;
  const int nwg = gridDim.x;
  const int q8 = nwg / 8, r8 = nwg % 8;
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int img = wg / (tiles_x * tiles_y);
  const int trem = wg - img * tiles_x * tiles_y;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  src += plane * img;
  dst += plane * img;
  const int x0 = txi * SX, y0 = tyi * SY;
  // LDS column swizzle (16-byte units of a strip row).  At OV = 2 a thread writes the units 2 rc and 2 rc + 1: the eight lanes that
  // a ds_write_b128 services together would hit units 0, 2, .. 14 = only 4 of the 8 unit slots of the 32 banks a write sees (2-way
  // conflict: a quarter of the kernel's LDS cycles in round 4's counters).  Unit f lives at f ^ ((f >> 3) & 1): lanes 0-3 then write
  // the even slots and lanes 4-7 the odd ones, and every 16-lane group of the ds_read_b128 readers (units tc, tc + 1) still covers
  // 16 different units.
  auto swz = [](int f) { return (OV == 2 && BLUR_SWZ) ? (f ^ ((f >> 3) & 1)) : f; };
  // Row pass.  The memory pipeline takes about one clock per lane and load whatever the width, so the window of a thread is
  // kept wide: NO outputs from NV aligned float4 loads (1.25 loads per 4 outputs at OV = 1, R = 5..8; 0.75 at OV = 2).
  const int rc = tid % TPR;                   // output group within the row
  const int xg = x0 + NO * rc;
  // Addresses are 32-bit byte offsets from the image's plane (uniform per workgroup: an SGPR pair), see blur_with_slot's size check.
  // 16-byte global loads need dword alignment only, so rows of any width take them.
  const bool fast_x = (xg - 4 * R4 >= 0) && (xg + NO - 1 + 4 * R4 <= w - 1);   // whole window inside the row
  const char *srcb = (const char *)src;
  auto load_window = [&](int ly, float *win) {
    int gy = y0 - R + ly;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    const unsigned rowb = (unsigned)gy * (unsigned)w * 4u;
    if (fast_x) {
      const unsigned b0 = rowb + (unsigned)(xg - 4 * R4) * 4u;
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const float4 q = *(const float4 *)(srcb + (b0 + 16u * v));
        win[4 * v] = q.x; win[4 * v + 1] = q.y; win[4 * v + 2] = q.z; win[4 * v + 3] = q.w;
      }
    } else {
      // BORDER_REPLICATE.  Window float4s start at multiples of 4 pixels: one is wholly left of the row (column 0 four times),
      // inside, or reaches sh = 1.. pixels past the last column w - 1: the float4 that ends at w - 1 is loaded instead and moved
      // down by sh, its last pixel filling the rest.
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const int gx = xg - 4 * R4 + 4 * v;
        const int cx = gx < 0 ? 0 : (gx > w - 4 ? w - 4 : gx);
        const float4 q = *(const float4 *)(srcb + (rowb + (unsigned)cx * 4u));
        const int sh = gx - cx;     // < 0: left of the row; 0: inside; 1, 2, 3, >= 4: past the end
        float4 o;
        o.x = sh <= 0 ? q.x : (sh == 1 ? q.y : (sh == 2 ? q.z : q.w));
        o.y = sh < 0 ? q.x : (sh == 0 ? q.y : (sh == 1 ? q.z : q.w));
        o.z = sh < 0 ? q.x : (sh == 0 ? q.z : q.w);
        o.w = sh < 0 ? q.x : q.w;
        win[4 * v] = o.x; win[4 * v + 1] = o.y; win[4 * v + 2] = o.z; win[4 * v + 3] = o.w;
      }
    }
  };
  // A thread owns NO columns and every RPS-th row of the (TH + 2R)-row strip; the next row window is in flight while the
  // current one is used (a ring of register windows, statically indexed after unrolling).
  constexpr int NI = (ROWS + RPS - 1) / RPS;
  constexpr int DEPTH = 1;   // measured: 2 or 4 windows in flight are slower (22-29 us vs 13 us per 1080p-pair plane)
  float win[DEPTH][4 * NV];
  const int ly0 = tid / TPR;
#pragma unroll
  for (int i = 0; i < DEPTH; i++)
    if (ly0 + RPS * i < ROWS) load_window(ly0 + RPS * i, win[i]);
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int ly = ly0 + RPS * i;
    if (ly < ROWS) {
      const float *wv = win[i % DEPTH];
      float o[NO];
#pragma unroll
      for (int u = 0; u < NO; u++) {
        float s;
        if constexpr (R <= 2) {   // ksize <= 5: cv::GaussianBlur's SymmRowSmallFilter, centre tap then the symmetric pairs
          s = wv[D + u + R] * taps.t[R];
#pragma unroll
          for (int j = 1; j <= R; j++) s = fmaf(wv[D + u + R - j] + wv[D + u + R + j], taps.t[R + j], s);
        } else {
          s = taps.t[0] * wv[D + u];
#pragma unroll
          for (int j = 1; j < N; j++) s = fmaf(taps.t[j], wv[D + u + j], s);
        }
        o[u] = s;
      }
#pragma unroll
      for (int v = 0; v < OV; v++)
        *(float4 *)(smem + ly * FB_TW + 4 * swz(OV * rc + v)) = make_float4(o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]);
    }
    if (i + DEPTH < NI && ly + RPS * DEPTH < ROWS) load_window(ly + RPS * DEPTH, win[i % DEPTH]);
  }
  __syncthreads();
  // Column pass.  A thread makes RPT vertically adjacent outputs of its 4 columns.  The taps are walked OUTWARDS for all RPT
  // outputs together: step j needs the strip rows R + j .. R + j + RPT - 1 ("up") and R - j .. R - j + RPT - 1 ("dn"), of which all
  // but one each were the previous step's, so a step costs two 16-byte LDS reads and the thread holds 2 RPT + 2 strip rows at a
  // time instead of all 2R + RPT (every output still gets: centre tap, then fma(tap j, row(+j) + row(-j), s), j = 1..R).
  const int tc = tid & 31;                    // 4-pixel column group
  const int x4 = x0 + 4 * tc;
  const int tcs = swz(tc), tcs1 = swz(tc + 1 < 32 ? tc + 1 : tc);
  float4 bl[RESP ? FB_TH / 8 : 1];            // RESP: this thread's blurred outputs, for the response phase
#pragma unroll
  for (int k = 0; k < (RESP ? FB_TH / 8 : 1); k++) bl[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x4 < w) {
    constexpr int RPT = FB_TH / 8;
    const int lyb = (tid >> 5) * RPT;
    const float *colp = smem + lyb * FB_TW + 4 * tcs;
    float4 acc[RPT];
    if constexpr (BLUR_COLJ && FB_TH >= BLUR_COLJ_MIN_TH && R >= BLUR_COLJ_MIN_R) {
    float4 up[RPT], dn[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const float4 c = *(const float4 *)(colp + (k + R) * FB_TW);
      acc[k] = make_float4(taps.t[R] * c.x, taps.t[R] * c.y, taps.t[R] * c.z, taps.t[R] * c.w);
      up[k] = c; dn[k] = c;
    }
    // (the two reads of step j + 1 are issued ahead of step j's arithmetic; the scheduling barrier keeps the compiler from hoisting
    // the reads of ALL steps to the top, which would bring the 2R + RPT rows back into registers)
    float4 nu = *(const float4 *)(colp + (RPT + R) * FB_TW), nd = *(const float4 *)(colp + (R - 1) * FB_TW);
#pragma unroll
    for (int j = 1; j <= R; j++) {
#pragma unroll
      for (int k = 0; k + 1 < RPT; k++) up[k] = up[k + 1];
      up[RPT - 1] = nu;
#pragma unroll
      for (int k = RPT - 1; k > 0; k--) dn[k] = dn[k - 1];
      dn[0] = nd;
      if (j < R) {
        nu = *(const float4 *)(colp + (RPT + R + j) * FB_TW);
        nd = *(const float4 *)(colp + (R - j - 1) * FB_TW);
      }
      const float t = taps.t[R + j];
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const float4 a = up[k], b = dn[k];
        acc[k].x = fmaf(t, a.x + b.x, acc[k].x); acc[k].y = fmaf(t, a.y + b.y, acc[k].y);
        acc[k].z = fmaf(t, a.z + b.z, acc[k].z); acc[k].w = fmaf(t, a.w + b.w, acc[k].w);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    } else {
    float4 col[2 * R + RPT];
#pragma unroll
    for (int q = 0; q < 2 * R + RPT; q++) col[q] = *(const float4 *)(colp + q * FB_TW);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const float4 c = col[k + R];
      float4 s = make_float4(taps.t[R] * c.x, taps.t[R] * c.y, taps.t[R] * c.z, taps.t[R] * c.w);
#pragma unroll
      for (int j = 1; j <= R; j++) {
        const float4 a = col[k + R + j], b = col[k + R - j];
        const float t = taps.t[R + j];
        s.x = fmaf(t, a.x + b.x, s.x); s.y = fmaf(t, a.y + b.y, s.y); s.z = fmaf(t, a.z + b.z, s.z); s.w = fmaf(t, a.w + b.w, s.w);
      }
      acc[k] = s;
    }
    }
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int gy = y0 + lyb + k;
      if (gy < h) {
        const float4 s = acc[k];
        const unsigned db = ((unsigned)gy * (unsigned)w + (unsigned)x4) * 4u;
        float *d = (float *)((char *)dst + db);
        if (x4 + 3 < w) store_f4_at<false>(dst, db, s.x, s.y, s.z, s.w);
        else {
          d[0] = s.x;
          if (x4 + 1 < w) d[1] = s.y;
          if (x4 + 2 < w) d[2] = s.z;
          if (x4 + 3 < w) d[3] = s.w;
        }
        if constexpr (RESP) bl[k] = s;
      }
    }
  }
  if constexpr (RESP) {
    // blurred tile -> LDS (the strip is dead once every thread holds its columns in registers)
    __syncthreads();
    {
      constexpr int RPT = FB_TH / 8;
      const int lyb = (tid >> 5) * RPT;
#pragma unroll
      for (int k = 0; k < RPT; k++) *(float4 *)(smem + (lyb + k) * FB_TW + 4 * tcs) = bl[k];
    }
    __syncthreads();
    // Response phase: thread = 4 columns x RQ rows of the owned region; rows / columns of the image frame and pixels whose
    // stencil would leave the image hold 0 (the reference leaves the frame undefined and never reads it).
    resp += plane * img;
    constexpr int RQ = (SY + 7) / 8;
    const int rg = tid >> 5;
    if (tc < SX / 4) {
      const int xr = x0 + 1 + 4 * tc;                     // first response column of this thread
      float v[RQ + 2][6];
#pragma unroll
      for (int q = 0; q < RQ + 2; q++) {
        const int ly = RQ * rg + q;                       // tile row of stencil row q (owned rows start at tile row 1)
        if (ly < FB_TH) {
          const float4 a = *(const float4 *)(smem + ly * FB_TW + 4 * tcs);
          const float4 b = *(const float4 *)(smem + ly * FB_TW + 4 * tcs1);     // tc < 31: inside the row (a 16-byte read: an
          v[q][0] = a.x; v[q][1] = a.y; v[q][2] = a.z; v[q][3] = a.w; v[q][4] = b.x; v[q][5] = b.y;   // 8-byte one at this stride is a 2-way bank conflict)
        } else {
#pragma unroll
          for (int e = 0; e < 6; e++) v[q][e] = 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < RQ; k++) {
        const int ly = RQ * rg + k + 1;                   // tile row of the output
        const int y = y0 + ly;
        if (ly <= SY && y < h) {
          float o[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int x = xr + u;
            float out = 0.f;
            if (x < w - 1 && y >= 1 && y < h - 1) {        // x >= 1 by construction
              const float v11 = v[k][u], v12 = v[k][u + 1], v13 = v[k][u + 2];
              const float v21 = v[k + 1][u], v22 = v[k + 1][u + 1], v23 = v[k + 1][u + 2];
              const float v31 = v[k + 2][u], v32 = v[k + 2][u + 1], v33 = v[k + 2][u + 2];
              float Lxx = (v21 - 2 * v22 + v23);
              float Lyy = (v12 - 2 * v22 + v32);
              float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
              out = (Lxx * Lyy - Lxy * Lxy) * norm2;
            }
            o[u] = out;
          }
          const unsigned db = ((unsigned)y * (unsigned)w + (unsigned)xr) * 4u;
          float *d = (float *)((char *)resp + db);
          if (xr + 3 < w) store_f4_at<true>(resp, db, o[0], o[1], o[2], o[3]);
          else {
            if (xr < w) d[0] = o[0];
            if (xr + 1 < w) d[1] = o[1];
            if (xr + 2 < w) d[2] = o[2];
          }
          if (xr == 1) d[-1] = 0.f;                       // column 0 of the frame
        }
      }
      if (y0 == 0 && rg == 0) {                           // row 0 of the frame
        float *d = resp + xr;
#pragma unroll
        for (int u = 0; u < 4; u++) if (xr + u < w) d[u] = 0.f;
        if (xr == 1) d[-1] = 0.f;
      }
    }
  }
}

// 3x3 Hessian determinant response.  grid = (ceil(w/64), ceil(h/4), n_img), block = 256.
__global__ __launch_bounds__(256) void hessian_response_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               int w, int h, float norm2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const size_t plane = (size_t)w * h;
  src += plane * blockIdx.z;
  dst += plane * blockIdx.z;
  float out = 0.f;   // the reference leaves the 1-px frame undefined; it is never read
  if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
    const float *p0 = src + (size_t)(y - 1) * w + x;
    const float *p1 = p0 + w;
    const float *p2 = p1 + w;
    const float v11 = p0[-1], v12 = p0[0], v13 = p0[1];
    const float v21 = p1[-1], v22 = p1[0], v23 = p1[1];
    const float v31 = p2[-1], v32 = p2[0], v33 = p2[1];
    float Lxx = (v21 - 2 * v22 + v23);
    float Lyy = (v12 - 2 * v22 + v32);
    float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
    out = (Lxx * Lyy - Lxy * Lxy) * norm2;
  }
  dst[(size_t)y * w + x] = out;
}

// The same response for planes whose width is a multiple of 4: a thread makes a 4 x 2 block of outputs from 4 rows x
// (one aligned float4 + the two pixels beside it): 12 loads per 8 pixels instead of 72.
// grid = (ceil(w/256), ceil(h/8), n_img), block = 256 (64 column groups x 4 row pairs).
__global__ __launch_bounds__(256) void hessian_response4_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                int w, int h, float norm2) {
  const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
  if (x4 >= w || y0 >= h) return;
  const size_t plane = (size_t)w * h;
  src += plane * blockIdx.z;
  dst += plane * blockIdx.z;
  float v[4][6];   // rows y0-1 .. y0+2 (clamped into the image: clamped rows only feed frame outputs, which are 0), columns x4-1 .. x4+4
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int y = y0 - 1 + q;
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const float *row = src + (size_t)y * w + x4;
    const float4 c = *(const float4 *)row;
    v[q][0] = x4 > 0 ? row[-1] : 0.f;
    v[q][1] = c.x; v[q][2] = c.y; v[q][3] = c.z; v[q][4] = c.w;
    v[q][5] = x4 + 4 < w ? row[4] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int y = y0 + k;
    if (y >= h) break;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int x = x4 + u;
      float out = 0.f;   // the reference leaves the 1-px frame undefined; it is never read
      if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        const float v11 = v[k][u], v12 = v[k][u + 1], v13 = v[k][u + 2];
        const float v21 = v[k + 1][u], v22 = v[k + 1][u + 1], v23 = v[k + 1][u + 2];
        const float v31 = v[k + 2][u], v32 = v[k + 2][u + 1], v33 = v[k + 2][u + 2];
        float Lxx = (v21 - 2 * v22 + v23);
        float Lyy = (v12 - 2 * v22 + v32);
        float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
        out = (Lxx * Lyy - Lxy * Lxy) * norm2;
      }
      o[u] = out;
    }
    *(float4 *)(dst + (size_t)y * w + x4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// cv::resize 0.5x: full 2x2 blocks ((a+b)+(c+d))*0.25f, edge blocks running sum / count.
__device__ __forceinline__ float decimated_at(const float *__restrict__ src, int w, int h, int dx, int dy) {
  const int sy0 = dy * 2, sx0 = dx * 2;
  if (sy0 >= h) return 0.f;
  const int wfull = (sy0 + 2 <= h) ? (w / 2) : 0;
  if (dx < wfull) {
    const float *S0 = src + (size_t)sy0 * w + sx0;
    const float *S1 = S0 + w;
    return ((S0[0] + S0[1]) + (S1[0] + S1[1])) * 0.25f;
  }
  if (sx0 >= w) return 0.f;
  float sum = 0; int count = 0;
  for (int sy = 0; sy < 2; sy++) {
    if (sy0 + sy >= h) break;
    for (int sx = 0; sx < 2; sx++) {
      if (sx0 + sx >= w) break;
      sum += src[(size_t)(sy0 + sy) * w + sx0 + sx];
      count++;
    }
  }
  return sum / (float)count;
}
__global__ __launch_bounds__(256) void resize_half_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          int w, int h, int dw, int dh) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= dw || dy >= dh) return;
  src += (size_t)w * h * blockIdx.z;
  dst += (size_t)dw * dh * blockIdx.z;
  dst[(size_t)dy * dw + dx] = decimated_at(src, w, h, dx, dy);
}

// The same decimation, and the Hessian response of the decimated plane (the first level of the next octave) from the block's
// tile in LDS: the 64 x 4 outputs of a block and the ring of decimated values around them (recomputed, 396 instead of 256
// values per block).  One launch instead of two per octave - the response launches of the small octaves cost 7-8 us each for
// a few microseconds of work.  Arithmetic of hessian_response4_kernel.
__global__ __launch_bounds__(256) void resize_half_resp_kernel(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ resp,
                                                               int w, int h, int dw, int dh, float norm2) {
  __shared__ float tile[6][66 + 2];
  src += (size_t)w * h * blockIdx.z;
  dst += (size_t)dw * dh * blockIdx.z;
  resp += (size_t)dw * dh * blockIdx.z;
  const int bx = blockIdx.x * 64 - 1, by = blockIdx.y * 4 - 1;
  for (int i = threadIdx.x; i < 6 * 66; i += 256) {
    const int ty = i / 66, tx = i - ty * 66;
    const int dx = bx + tx, dy = by + ty;
    tile[ty][tx] = (dx >= 0 && dx < dw && dy >= 0 && dy < dh) ? decimated_at(src, w, h, dx, dy) : 0.f;
  }
  __syncthreads();
  const int tx = (threadIdx.x & 63) + 1, ty = (threadIdx.x >> 6) + 1;
  const int x = bx + tx, y = by + ty;
  if (x >= dw || y >= dh) return;
  dst[(size_t)y * dw + x] = tile[ty][tx];
  float out = 0.f;   // the reference leaves the 1-px frame undefined; it is never read
  if (x >= 1 && x < dw - 1 && y >= 1 && y < dh - 1) {
    const float v11 = tile[ty - 1][tx - 1], v12 = tile[ty - 1][tx], v13 = tile[ty - 1][tx + 1];
    const float v21 = tile[ty][tx - 1], v22 = tile[ty][tx], v23 = tile[ty][tx + 1];
    const float v31 = tile[ty + 1][tx - 1], v32 = tile[ty + 1][tx], v33 = tile[ty + 1][tx + 1];
    float Lxx = (v21 - 2 * v22 + v23);
    float Lyy = (v12 - 2 * v22 + v32);
    float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
    out = (Lxx * Lyy - Lxy * Lxy) * norm2;
  }
  resp[(size_t)y * dw + x] = out;
}


// The tail of the pyramid in ONE launch: the octaves whose planes fit LDS (<= LDSP_CAP pixels: 120 x 68 and smaller for a 1080p
// image) are built level by level by one 1024-thread workgroup per image - plane A (the current level), plane B (the row pass)
// and the next octave's first plane N stay in LDS, only the results go to HBM.  As separate launches these octaves are ~6
// launches each of 6-8 us for a few microseconds of work (18 launches for a 1080p batch).  Same arithmetic as the tiled
// kernels: row pass left to right (centre first for ksize <= 5), column pass centre then symmetric pairs (below + above),
// BORDER_REPLICATE, response stencil of hessian_response4_kernel, decimation of resize_half_kernel.
constexpr int LDSP_CAP = 9216;
struct LdsPyramidPlan {
  int first, n_oct, n_levels, S;
  int ntap[kMaxLevels];
  BlurTaps taps[kMaxLevels];
  float norm2[kMaxLevels];     // response norm of level l (l >= 1), squared as the kernels take it
  float norm2_first;           // ... of a decimated first level
};
__device__ __forceinline__ float lds_response(const float *A, int w, int h, int x, int y, float norm2) {
  if (!(x >= 1 && x < w - 1 && y >= 1 && y < h - 1)) return 0.f;
  const float *r0 = A + (y - 1) * w + x, *r1 = r0 + w, *r2 = r1 + w;
  const float v11 = r0[-1], v12 = r0[0], v13 = r0[1], v21 = r1[-1], v22 = r1[0], v23 = r1[1], v31 = r2[-1], v32 = r2[0], v33 = r2[1];
  float Lxx = (v21 - 2 * v22 + v23);
  float Lyy = (v12 - 2 * v22 + v32);
  float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
  return (Lxx * Lyy - Lxy * Lxy) * norm2;
}
template <int N>
__device__ __forceinline__ float lds_row_taps(const float *pp, const float *t) {        // pp = &row[x - R], whole window inside the row
  float s = t[0] * pp[0];
#pragma unroll
  for (int j = 1; j < N; j++) s = fmaf(t[j], pp[j], s);
  return s;
}
__device__ __forceinline__ float lds_row_px(const float *row, int x, int w, int n, const float *t) {
  const int R = n >> 1;
  if (n > 5 && x >= R && x + R < w) {
    const float *pp = row + x - R;
    switch (n) {
      case 7: return lds_row_taps<7>(pp, t);
      case 9: return lds_row_taps<9>(pp, t);
      case 11: return lds_row_taps<11>(pp, t);
      case 13: return lds_row_taps<13>(pp, t);
      case 15: return lds_row_taps<15>(pp, t);
      case 17: return lds_row_taps<17>(pp, t);
    }
  }
  float s;
  if (n <= 5) {
    s = row[x] * t[R];
    for (int j = 1; j <= R; j++) { const int xa = x - j < 0 ? 0 : x - j, xb = x + j > w - 1 ? w - 1 : x + j; s = fmaf(row[xa] + row[xb], t[R + j], s); }
  } else {
    int x0 = x - R; x0 = x0 < 0 ? 0 : x0;
    s = t[0] * row[x0];
    for (int j = 1; j < n; j++) { int xx = x - R + j; xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx); s = fmaf(t[j], row[xx], s); }
  }
  return s;
}
template <int R>
__device__ __forceinline__ float lds_col_taps(const float *pc, int w, const float *t) {   // pc = &B[y * w + x], rows y - R .. y + R inside
  float s = t[R] * pc[0];
#pragma unroll
  for (int j = 1; j <= R; j++) s = fmaf(t[R + j], pc[j * w] + pc[-j * w], s);
  return s;
}
__device__ __forceinline__ float lds_col_px(const float *B, int x, int y, int w, int h, int n, const float *t) {
  const int R = n >> 1;
  const float *pc = B + y * w + x;
  if (y >= R && y + R < h) {
    switch (R) {
      case 1: return lds_col_taps<1>(pc, w, t);
      case 2: return lds_col_taps<2>(pc, w, t);
      case 3: return lds_col_taps<3>(pc, w, t);
      case 4: return lds_col_taps<4>(pc, w, t);
      case 5: return lds_col_taps<5>(pc, w, t);
      case 6: return lds_col_taps<6>(pc, w, t);
      case 7: return lds_col_taps<7>(pc, w, t);
      case 8: return lds_col_taps<8>(pc, w, t);
    }
  }
  float s = t[R] * pc[0];
  for (int j = 1; j <= R; j++) {
    const int ya = y + j > h - 1 ? h - 1 : y + j, yb = y - j < 0 ? 0 : y - j;
    s = fmaf(t[R + j], B[ya * w + x] + B[yb * w + x], s);
  }
  return s;
}
// p / w for p < 2^32 / w as one multiply-high (w = 1: the magic is 0 and p itself is returned)
__device__ __forceinline__ unsigned lds_div_magic(int w) { return w > 1 ? 0xFFFFFFFFu / (unsigned)w + 1u : 0u; }
__device__ __forceinline__ int lds_div(int p, unsigned magic) { return magic ? (int)__umulhi((unsigned)p, magic) : p; }
__global__ __launch_bounds__(1024) void pyramid_lds_kernel(const PyramidDev *__restrict__ P, LdsPyramidPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *A = sm, *B = sm + LDSP_CAP, *N = B + LDSP_CAP;
  const int tid = threadIdx.x, b = blockIdx.x;
  int w = P->oct[pl.first].w, h = P->oct[pl.first].h;
  {
    const float *src = as_global(P->oct[pl.first].blur[0]) + (size_t)w * h * b;
    for (int p = tid; p < w * h; p += 1024) A[p] = src[p];
  }
  __syncthreads();
  for (int oi = pl.first; oi < pl.n_oct; oi++) {
    const OctaveDev &o = P->oct[oi];
    const size_t plane = (size_t)w * h * b;
    for (int l = 1; l < pl.n_levels; l++) {
      const int n = pl.ntap[l];
      const float *t = pl.taps[l].t;
      // (one workgroup per image = one CU: the passes are bound by instruction issue, so the tap loops are instantiated for the
      // kernel size - immediate LDS offsets, no address arithmetic or clamping per tap away from the plane's edge)
      // (p / w as one multiply-high: exact for p < 2^32 / w, the planes here hold <= LDSP_CAP pixels)
      const int npx = w * h;
      const unsigned inv_w = lds_div_magic(w);
      for (int p = tid; p < npx; p += 1024) {              // row pass A -> B
        const int y = lds_div(p, inv_w), x = p - y * w;
        B[p] = lds_row_px(A + y * w, x, w, n, t);
      }
      __syncthreads();
      float *blur = as_global(o.blur[l]) + plane;
      for (int p = tid; p < npx; p += 1024) {              // column pass B -> A (and the level's blurred plane)
        const int y = lds_div(p, inv_w), x = p - y * w;
        const float s = lds_col_px(B, x, y, w, h, n, t);
        A[p] = s;
        blur[p] = s;
      }
      __syncthreads();
      float *resp = as_global(o.resp[l]) + plane;
      for (int p = tid; p < npx; p += 1024) { const int y = lds_div(p, inv_w), x = p - y * w; resp[p] = lds_response(A, w, h, x, y, pl.norm2[l]); }
      if (l == pl.S && oi + 1 < pl.n_oct) {                // first level of the next octave: decimation + its response
        const OctaveDev &nx = P->oct[oi + 1];
        const int dw = nx.w, dh = nx.h;
        const unsigned inv_dw = lds_div_magic(dw);
        for (int q = tid; q < dw * dh; q += 1024) { const int dy = lds_div(q, inv_dw), dx = q - dy * dw; N[q] = decimated_at(A, w, h, dx, dy); }
        __syncthreads();
        float *nb = as_global(nx.blur[0]) + (size_t)dw * dh * b, *nr = as_global(nx.resp[0]) + (size_t)dw * dh * b;
        for (int q = tid; q < dw * dh; q += 1024) { const int dy = lds_div(q, inv_dw), dx = q - dy * dw; nb[q] = N[q]; nr[q] = lds_response(N, dw, dh, dx, dy, pl.norm2_first); }
      }
      __syncthreads();
    }
    if (oi + 1 < pl.n_oct) {
      const int dw = P->oct[oi + 1].w, dh = P->oct[oi + 1].h;
      for (int q = tid; q < dw * dh; q += 1024) A[q] = N[q];
      w = dw; h = dh;
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// DoG and Harris responses (ScaleSpaceDetector::dogResponse / iidogResponse / HarrisResponse, pyramid.cpp:165-194, 256-278).
// Their own Gaussian blurs are wide (DoG passes sigma^2 of the level as the sigma: ksize up to 99 with the default
// schedule), so they run as two plain passes over global memory with the taps in a device table: row pass as OpenCV's
// RowFilter (first tap, then fused multiply-adds left to right; SymmRowSmall order for ksize <= 5), column pass as
// SymmColumnFilter (centre tap, then the symmetric pairs), BORDER_REPLICATE - the arithmetic of gauss_blur_kernel.
// grid = (ceil(w/256), h, n_img), block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_blur_row_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                            const float *__restrict__ taps, int n) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t plane = (size_t)w * h * blockIdx.z;
  const float *row = src + plane + (size_t)y * w;
  const int r = n >> 1;
  auto at = [&](int j) { int gx = x - r + j; gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx); return row[gx]; };
  float s;
  if (n <= 5) {
    s = at(r) * taps[r];
    for (int j = 1; j <= r; j++) s = fmaf(at(r - j) + at(r + j), taps[r + j], s);
  } else {
    s = taps[0] * at(0);
    for (int j = 1; j < n; j++) s = fmaf(taps[j], at(j), s);
  }
  dst[plane + (size_t)y * w + x] = s;
}
__global__ __launch_bounds__(256) void wide_blur_col_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                            const float *__restrict__ taps, int n) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t plane = (size_t)w * h * blockIdx.z;
  const float *col = src + plane + x;
  const int r = n >> 1;
  auto at = [&](int gy) { gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy); return col[(size_t)gy * w]; };
  float s = taps[r] * at(y);
  for (int j = 1; j <= r; j++) s = fmaf(taps[r + j], at(y + j) + at(y - j), s);
  dst[plane + (size_t)y * w + x] = s;
}
// DoG = level - blur(level); iiDoG: where level + blur < 255, the value is rescaled by 255 / (level + blur) in double
__global__ __launch_bounds__(256) void dog_combine_kernel(const float *__restrict__ in, const float *__restrict__ nb, float *__restrict__ out,
                                                          size_t n, int ii) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = in[i], b = nb[i];
  float v = a - b;
  if (ii) {
    const float sum = a + b;
    if ((double)sum < 255.) v = (float)((double)v * (255. / (double)sum));
  }
  out[i] = v;
}
// computeGradient (helpers.cpp:779-797: central differences, one-sided at the frame) and the three products
__global__ __launch_bounds__(256) void harris_products_kernel(const float *__restrict__ src, float *__restrict__ xx, float *__restrict__ yy,
                                                              float *__restrict__ xy, int w, int h) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t plane = (size_t)w * h * blockIdx.z;
  const float *p = src + plane + (size_t)y * w + x;
  float gx, gy;
  if (x == 0) gx = p[1] - p[0]; else if (x == w - 1) gx = p[0] - p[-1]; else gx = p[1] - p[-1];
  if (y == 0) gy = p[w] - p[0]; else if (y == h - 1) gy = p[0] - p[-w]; else gy = p[w] - p[-w];
  const size_t o = plane + (size_t)y * w + x;
  xx[o] = gx * gx; yy[o] = gy * gy; xy[o] = gx * gy;
}
// sigmasq * blurred products, then dx2*dy2 - dxdy^2 - 0.04 (dx2 + dy2)^2 in OpenCV's evaluation order:
// (fl(dx2*dy2) - fl(dxdy*dxdy)) - fl(fl(0.04f * sum) * sum)
__global__ __launch_bounds__(256) void harris_combine_kernel(const float *__restrict__ bxx, const float *__restrict__ byy,
                                                             const float *__restrict__ bxy, float *__restrict__ out, size_t n, float sigmasq) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dx2 = bxx[i] * sigmasq, dy2 = byy[i] * sigmasq, dxdy = bxy[i] * sigmasq;
  const float sum = dx2 + dy2;
  const float t1 = dx2 * dy2, t2 = dxdy * dxdy;
  const float t3 = ((float)0.04 * sum) * sum;
  out[i] = (t1 - t2) - t3;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// Tap tables live in 16 device slots of 64 floats; a slot is re-uploaded (synchronously, so
// the host buffer is never read late) only when its sigma changes: steady state = no copies.
// both streams of the context drained: a tap slot may be read by launches of either
static int wait_both_streams(mods_ctx *ctx) {
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
  if (ctx->stream2 && ctx->stream2 != ctx->stream) MODS_HIP_CHECK(mods::stream_wait(ctx->stream2));
  return MODS_OK;
}
static int upload_taps(mods_ctx *ctx, int slot, float sigma, int *n_out) {
  const int n = gauss_ksize(sigma);
  if (n / 2 > kMaxBlurRadius) { set_error("gaussian kernel too wide: sigma=%g ksize=%d", (double)sigma, n); return MODS_E_ARG; }
  *n_out = n;
  if (ctx->taps_sigma[slot] == sigma) return MODS_OK;
  float taps[2 * kMaxBlurRadius + 1];
  gauss_kernel_host(n, (double)sigma, taps);
  for (int i = 0; i < n; i++) ctx->taps_host[slot][i] = taps[i];
  ctx->taps_host_n[slot] = n;
  { const int wrc = wait_both_streams(ctx); if (wrc) return wrc; }   // earlier launches may still read the slot
  mods::dev_state_changed(ctx);
  MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->gauss_taps_dev + slot * 64, taps, sizeof(float) * n, hipMemcpyHostToDevice));
  ctx->taps_sigma[slot] = sigma;
  return MODS_OK;
}

template <int R>
static void launch_fast_blur(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, const BlurTaps &taps, float *resp,
                             float norm2) {
#ifndef BLUR_OV_BIG
#define BLUR_OV_BIG 2   // float4 output groups per thread and strip row on the large planes
#endif
#ifndef BLUR_TH_BIG
#define BLUR_TH_BIG 32   // measured on 1080p pairs: 32-row tiles (4 workgroups per CU, phases of different tiles overlap) beat 64-row tiles by ~15 %
#endif
  const int tiles64 = ((w + FB_TW - 1) / FB_TW) * ((h + 63) / 64) * n_img;
  if (tiles64 >= 384) {   // at least ~1.5 tiles per CU: large planes: 32-row tiles
    const size_t lds = sizeof(float) * (size_t)(BLUR_TH_BIG + 2 * R) * FB_TW;
    if (resp) {
      const int tilesB = blur_resp_tiles(w, FB_TW - 4, 4) * blur_resp_tiles(h, BLUR_TH_BIG - 2, 2) * n_img;
      hipLaunchKernelGGL((gauss_blur_fast_kernel<R, BLUR_TH_BIG, BLUR_OV_BIG, true>), dim3(tilesB), dim3(256), lds, ctx->stream, src, dst, w, h, taps, resp, norm2);
    } else {
      const int tilesB = ((w + FB_TW - 1) / FB_TW) * ((h + BLUR_TH_BIG - 1) / BLUR_TH_BIG) * n_img;
      hipLaunchKernelGGL((gauss_blur_fast_kernel<R, BLUR_TH_BIG, BLUR_OV_BIG, false>), dim3(tilesB), dim3(256), lds, ctx->stream, src, dst, w, h, taps, nullptr, 0.f);
    }
  } else {                // small planes: short tiles, more workgroups, shorter per-thread row chains
    const size_t lds = sizeof(float) * (size_t)(16 + 2 * R) * FB_TW;
    if (resp) {
      const int tiles16 = blur_resp_tiles(w, FB_TW - 4, 4) * blur_resp_tiles(h, 16 - 2, 2) * n_img;
      hipLaunchKernelGGL((gauss_blur_fast_kernel<R, 16, 1, true>), dim3(tiles16), dim3(256), lds, ctx->stream, src, dst, w, h, taps, resp, norm2);
    } else {
      const int tiles16 = ((w + FB_TW - 1) / FB_TW) * ((h + 15) / 16) * n_img;
      hipLaunchKernelGGL((gauss_blur_fast_kernel<R, 16, 1, false>), dim3(tiles16), dim3(256), lds, ctx->stream, src, dst, w, h, taps, nullptr, 0.f);
    }
  }
}

// blur of one level; resp != nullptr: the Hessian response of the blurred plane (norm = sigma^2 of the level) comes out of the
// same launch (register-blocked kernel only; the generic kernel falls back to a separate response launch)
static int blur_with_slot(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, int slot, int n, float *resp = nullptr,
                          float norm = 0.f) {
  const int r = n / 2;
  if (r >= 1 && r <= 8 && ctx->taps_host_n[slot] == n && w >= 4 && (size_t)w * h < ((size_t)1 << 29)) {   // (16-byte row loads; 32-bit byte offsets inside a plane)
    BlurTaps taps;
    for (int i = 0; i < 17; i++) taps.t[i] = i < n ? ctx->taps_host[slot][i] : 0.f;
    // (the two tile instantiations are timed separately: launches of the small planes are launch-size bound)
    const bool big_tiles = ((w + FB_TW - 1) / FB_TW) * ((h + 63) / 64) * n_img >= 384;
    // algorithmic bytes of the launch by SURVEY 8d's unfused model: 8 B/px for the blur (+ 8 B/px for the response)
    StageScope ts(ctx, big_tiles ? MODS_STAGE_BLUR : MODS_STAGE_BLUR_SMALL, (resp ? 16.0 : 8.0) * w * h * n_img);
    const float norm2 = norm * norm;
    switch (r) {
      case 1: launch_fast_blur<1>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 2: launch_fast_blur<2>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 3: launch_fast_blur<3>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 4: launch_fast_blur<4>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 5: launch_fast_blur<5>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 6: launch_fast_blur<6>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      case 7: launch_fast_blur<7>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
      default: launch_fast_blur<8>(ctx, src, dst, w, h, n_img, taps, resp, norm2); break;
    }
    MODS_HIP_CHECK(hipGetLastError());
    return MODS_OK;
  }
  const size_t lds = sizeof(float) * ((size_t)(BLUR_TH + 2 * r) * (BLUR_TW + 2 * r) + (size_t)(BLUR_TH + 2 * r) * BLUR_TW + 64);
  dim3 grid((w + BLUR_TW - 1) / BLUR_TW, (h + BLUR_TH - 1) / BLUR_TH, n_img);
  {
    StageScope ts(ctx, MODS_STAGE_BLUR, 8.0 * w * h * n_img);
    hipLaunchKernelGGL(gauss_blur_kernel, grid, dim3(256), lds, ctx->stream, src, dst, w, h, ctx->gauss_taps_dev + slot * 64, n);
    MODS_HIP_CHECK(hipGetLastError());
  }
  if (resp) return launch_hessian_response(ctx, dst, resp, w, h, n_img, norm);
  return MODS_OK;
}

int launch_gauss_blur(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, float sigma) {
  int n;
  int rc = upload_taps(ctx, 15, sigma, &n);
  if (rc) return rc;
  return blur_with_slot(ctx, src, dst, w, h, n_img, 15, n);
}

int launch_hessian_response(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, float norm) {
  dim3 grid((w + 63) / 64, (h + 3) / 4, n_img);
  StageScope ts(ctx, MODS_STAGE_RESPONSE, 8.0 * w * h * n_img);
  if ((w & 3) == 0)
    hipLaunchKernelGGL(hessian_response4_kernel, dim3((w + 255) / 256, (h + 7) / 8, n_img), dim3(256), 0, ctx->stream, src, dst, w, h,
                       norm * norm);
  else
    hipLaunchKernelGGL(hessian_response_kernel, grid, dim3(256), 0, ctx->stream, src, dst, w, h, norm * norm);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// sigma of the blur inside the response of level l (pyramid.cpp:165-168: DoG blurs with norm = sigma_l^2 as the sigma;
// :260-261: Harris with sqrt(0.6 * sigma_l^2))
static float alt_response_sigma(const mods_hessaff_params &par, float level_sigma) {
  const float norm = level_sigma * level_sigma;
  if (par.detectorType == MODS_DET_DOG) return norm;
  const float sigmasq = (float)(0.6 * (double)norm);
  return sqrtf(sigmasq);
}

static int wide_blur(mods_ctx *ctx, const float *src, float *dst, float *tmp, int w, int h, int n_img, int level) {
  const float *taps = ctx->alt_taps_dev + (size_t)level * kAltTapStride;
  const int n = ctx->alt_ntap[level];
  dim3 grid((w + 255) / 256, h, n_img);
  StageScope ts(ctx, MODS_STAGE_RESPONSE, 16.0 * w * h * n_img);
  hipLaunchKernelGGL(wide_blur_row_kernel, grid, dim3(256), 0, ctx->stream, src, tmp, w, h, taps, n);
  hipLaunchKernelGGL(wide_blur_col_kernel, grid, dim3(256), 0, ctx->stream, tmp, dst, w, h, taps, n);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// Response() of one level for DET_DOG / DET_HARRIS, whole batch
static int launch_alt_response(mods_ctx *ctx, const float *blur, float *resp, int w, int h, int n_img, int level, float level_sigma) {
  const size_t n = (size_t)w * h * n_img;
  float *p0 = ctx->alt_planes, *p1 = p0 + ctx->alt_plane_elems, *p2 = p1 + ctx->alt_plane_elems, *p3 = p2 + ctx->alt_plane_elems;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  int rc;
  if (ctx->par.detectorType == MODS_DET_DOG) {
    if ((rc = wide_blur(ctx, blur, p0, p1, w, h, n_img, level))) return rc;
    hipLaunchKernelGGL(dog_combine_kernel, dim3(blocks), dim3(256), 0, ctx->stream, blur, p0, resp, n, ctx->par.iiDoGMode);
  } else {
    const float norm = level_sigma * level_sigma;
    const float sigmasq = (float)(0.6 * (double)norm);
    hipLaunchKernelGGL(harris_products_kernel, dim3((w + 255) / 256, h, n_img), dim3(256), 0, ctx->stream, blur, p0, p1, p2, w, h);
    if ((rc = wide_blur(ctx, p0, p0, p3, w, h, n_img, level))) return rc;
    if ((rc = wide_blur(ctx, p1, p1, p3, w, h, n_img, level))) return rc;
    if ((rc = wide_blur(ctx, p2, p2, p3, w, h, n_img, level))) return rc;
    hipLaunchKernelGGL(harris_combine_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p0, p1, p2, resp, n, sigmasq);
  }
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// decimation + Hessian response of the result (norm = sigma^2 of the next octave's first level, squared for the kernel as in launch_hessian_response)
static int launch_resize_half_resp(mods_ctx *ctx, const float *src, float *dst, float *resp, int w, int h, int dw, int dh, int n_img, float norm) {
  dim3 grid((dw + 63) / 64, (dh + 3) / 4, n_img);
  StageScope ts(ctx, MODS_STAGE_RESIZE, 5.0 * w * h * n_img + 4.0 * dw * dh * n_img);
  hipLaunchKernelGGL(resize_half_resp_kernel, grid, dim3(256), 0, ctx->stream, src, dst, resp, w, h, dw, dh, norm * norm);   // as launch_hessian_response
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

int launch_resize_half(mods_ctx *ctx, const float *src, float *dst, int w, int h, int dw, int dh, int n_img) {
  dim3 grid((dw + 63) / 64, (dh + 3) / 4, n_img);
  StageScope ts(ctx, MODS_STAGE_RESIZE, 5.0 * w * h * n_img);
  hipLaunchKernelGGL(resize_half_kernel, grid, dim3(256), 0, ctx->stream, src, dst, w, h, dw, dh);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// ---------------------------------------------------------------------------------------
// scale-space layout + build
// ---------------------------------------------------------------------------------------
// Computes the octave ladder (pyramid.cpp:520-528) and carves the planes out of the pools.
int pyramid_configure(mods_ctx *ctx, int w, int h, int n_img, const mods_hessaff_params *par) {
  if (w <= 0 || h <= 0 || n_img <= 0 || n_img > ctx->batch) { set_error("bad image batch %dx%d x%d (ctx batch %d)", w, h, n_img, ctx->batch); return MODS_E_ARG; }
  // every per-image scratch plane (input_dev, tmp_dev, the pyramid pool's first octave) is sized max_w * max_h
  if ((size_t)w * h > (size_t)ctx->max_w * ctx->max_h) { set_error("image %dx%d larger than the context (%dx%d)", w, h, ctx->max_w, ctx->max_h); return MODS_E_ARG; }
  const int n_levels = par->numberOfScales + 2;
  if (par->numberOfScales < 1 || n_levels > kMaxLevels) { set_error("numberOfScales %d unsupported", par->numberOfScales); return MODS_E_ARG; }
  if (par->border < 2) { set_error("border must be >= 2"); return MODS_E_ARG; }
  if (par->detectorType < MODS_DET_HESSIAN || par->detectorType > MODS_DET_HARRIS) { set_error("unknown detector type %d", par->detectorType); return MODS_E_ARG; }
  // Response() has no iiDoG form for Hessian / Harris: those branches are commented out in the reference and run off the end
  // of a non-void function (pyramid.cpp:130-136, 148-156)
  if (par->iiDoGMode && par->detectorType != MODS_DET_DOG) { set_error("iiDoGMode exists for the DoG detector only"); return MODS_E_ARG; }
  PyramidDev &P = ctx->pyr;
  P.n_levels = n_levels;
  P.n_oct = 0;
  const int minSize = 2 * par->border + 2;
  int cw = w, ch = h;
  float pd = 1.0f;
  size_t plane_elems = 0, omap_elems = 0;
  const float sigmaStep = det_pow2f(1.0f / (float)par->numberOfScales);
  while (ch > minSize && cw > minSize) {
    if (P.n_oct >= kMaxOctaves) break;
    OctaveDev &o = P.oct[P.n_oct++];
    o.w = cw; o.h = ch; o.pixelDistance = pd;
    float cur = par->initialSigma;
    for (int l = 0; l < n_levels; l++) { o.sigma[l] = cur; cur *= sigmaStep; }
    plane_elems += (size_t)cw * ch * n_img * 2 * n_levels;
    omap_elems += (size_t)cw * ch * n_img;
    int nw, nh;
    resize_half_dims(cw, ch, &nw, &nh);
    cw = nw; ch = nh;
    pd *= 2.0f;
  }
  if (plane_elems > ctx->plane_pool_elems) {
    if (ctx->plane_pool) MODS_HIP_CHECK(hipFree(ctx->plane_pool));
    ctx->plane_pool = nullptr;
    mods::dev_pool_reallocated(ctx); MODS_HIP_CHECK(hipMalloc(&ctx->plane_pool, plane_elems * sizeof(float)));
    ctx->plane_pool_elems = plane_elems;
  }
  if (omap_elems > ctx->omap_pool_elems) {
    if (ctx->omap_pool) MODS_HIP_CHECK(hipFree(ctx->omap_pool));
    ctx->omap_pool = nullptr;
    mods::dev_pool_reallocated(ctx); MODS_HIP_CHECK(hipMalloc(&ctx->omap_pool, omap_elems * sizeof(unsigned int)));
    ctx->omap_pool_elems = omap_elems;
    ctx->omap_dirty = true;
  }
  float *pp = ctx->plane_pool;
  unsigned int *mp = ctx->omap_pool;
  for (int oi = 0; oi < P.n_oct; oi++) {
    OctaveDev &o = P.oct[oi];
    const size_t n = (size_t)o.w * o.h * n_img;
    for (int l = 0; l < n_levels; l++) { o.blur[l] = pp; pp += n; o.resp[l] = pp; pp += n; }
    o.omap = mp; mp += n;
  }
  // the device copy of the table is refreshed only when the table changed (a batch of the same geometry leaves it alone: no
  // host-to-device copy in the steady state of a worker, and none inside a recorded graph - mods_ctx_graphs)
  if (!ctx->pyr_dev_valid || memcmp(&ctx->pyr_dev_image, &P, sizeof(PyramidDev)) != 0) {
    ctx->pyr_dev_valid = false;
    mods::dev_state_changed(ctx);     // (device state changed from the host: the next call is not a repeat - capi.hip: dd_run)
    MODS_HIP_CHECK(hipMemcpyAsync(ctx->pyr_dev, &P, sizeof(PyramidDev), hipMemcpyHostToDevice, ctx->stream));   // (stream ordered behind the readers of the old table)
    memcpy(&ctx->pyr_dev_image, &P, sizeof(PyramidDev));
    ctx->pyr_dev_valid = true;
  }
  if (par->detectorType != MODS_DET_HESSIAN) {
    if (!ctx->alt_taps_dev) MODS_HIP_CHECK(hipMalloc(&ctx->alt_taps_dev, sizeof(float) * kMaxLevels * kAltTapStride));
    const size_t need = (size_t)w * h * n_img;
    if (need > ctx->alt_plane_elems) {
      if (ctx->alt_planes) MODS_HIP_CHECK(hipFree(ctx->alt_planes));
      ctx->alt_planes = nullptr; ctx->alt_plane_elems = 0;
      mods::dev_pool_reallocated(ctx); MODS_HIP_CHECK(hipMalloc(&ctx->alt_planes, 4 * need * sizeof(float)));
      ctx->alt_plane_elems = need;
    }
    for (int l = 0; l < n_levels; l++) {
      const float sigma = alt_response_sigma(*par, P.oct[0].sigma[l]);
      if (ctx->alt_ntap[l] && ctx->alt_sigma[l] == sigma) continue;
      const int n = gauss_ksize(sigma);
      if (n > kAltTapStride - 1) { set_error("response blur too wide: sigma=%g ksize=%d", (double)sigma, n); return MODS_E_ARG; }
      std::vector<float> taps(n);
      gauss_kernel_host(n, (double)sigma, taps.data());
      { const int wrc = wait_both_streams(ctx); if (wrc) return wrc; }
      mods::dev_state_changed(ctx);
      MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->alt_taps_dev + (size_t)l * kAltTapStride, taps.data(), sizeof(float) * n, hipMemcpyHostToDevice));
      ctx->alt_ntap[l] = n; ctx->alt_sigma[l] = sigma;
    }
  }
  ctx->par = *par;
  ctx->reg_number_eff = par->regionsNumber;   // identity view; mods_detect_describe_view_dev rescales it
  ctx->last_w = w; ctx->last_h = h; ctx->last_n_img = n_img;
  // SMM mask for Baumberg
  if (ctx->smm_mask_size != par->smmWindowSize) {
    if (par->smmWindowSize < 3 || par->smmWindowSize > 31 || !(par->smmWindowSize & 1)) { set_error("smmWindowSize %d unsupported", par->smmWindowSize); return MODS_E_ARG; }
    std::vector<float> m((size_t)par->smmWindowSize * par->smmWindowSize);
    gauss_mask_host(par->smmWindowSize, m.data());
    mods::dev_state_changed(ctx);
    MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->smm_mask_dev, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->smm_mask_size = par->smmWindowSize;
  }
  return MODS_OK;
}

// detectPyramidKeypoints / detectOctaveKeypoints (pyramid.cpp:496-529, 428-494): every blur and
// response plane of every octave, for the whole batch.  `img_dev`: [n_img][h][stride].
// Joins the side stream of a forked pyramid into the main stream (no-op without a pending fork).  detect_run does this on
// its way to the compaction; every error exit between the fork and that point, and a call that finds the flag still set,
// come through here, so side-stream work never outlives the call that started it unjoined.
int pyramid_join_side(mods_ctx *ctx) {
  if (!ctx->pyr_side) return MODS_OK;
  ctx->pyr_side = false;
  MODS_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->stream2));
  MODS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return MODS_OK;
}

static int pyramid_build_levels(mods_ctx *ctx, const float *img_dev, int stride);
int pyramid_build(mods_ctx *ctx, const float *img_dev, int stride) {
  int rc = pyramid_join_side(ctx);          // a previous call that failed behind its fork
  if (rc) return rc;
  rc = pyramid_build_levels(ctx, img_dev, stride);
  if (rc) (void)pyramid_join_side(ctx);     // (the stream swap is undone by the build's own guard before this runs)
  return rc;
}

static int pyramid_build_levels(mods_ctx *ctx, const float *img_dev, int stride) {
  ctx->last_img_dev = img_dev; ctx->last_stride = stride;
  ctx->pyr_scope_begin = nullptr;
  if ((ctx->timing_mask >> MODS_STAGE_PYRAMID) & 1) {
    StageTimer &t = ctx->timers[MODS_STAGE_PYRAMID];
    if (!t.pool.empty()) { ctx->pyr_scope_begin = t.pool.back(); t.pool.pop_back(); }
    else MODS_HIP_CHECK(hipEventCreate(&ctx->pyr_scope_begin));
    MODS_HIP_CHECK(hipEventRecord(ctx->pyr_scope_begin, ctx->stream));
  }
  PyramidDev &P = ctx->pyr;
  const mods_hessaff_params &par = ctx->par;
  const int n_img = ctx->last_n_img, w = ctx->last_w, h = ctx->last_h;
  const int S = par.numberOfScales;
  if (P.n_oct == 0) return MODS_OK;
  int rc;
  // tap tables: slot 0 = initial blur, slot l = increment to level l
  int ntap[kMaxLevels + 1];
  const float sigmaStep = det_pow2f(1.0f / (float)S);
  const float curSigma0 = 0.5f;
  bool initial_blur = par.initialSigma > curSigma0;
  if (initial_blur) {
    float sigma = sqrtf(par.initialSigma * par.initialSigma - curSigma0 * curSigma0);
    if ((rc = upload_taps(ctx, 0, sigma, &ntap[0]))) return rc;
  }
  for (int l = 1; l < P.n_levels; l++) {
    float sigma = P.oct[0].sigma[l - 1] * sqrtf(sigmaStep * sigmaStep - 1.0f);
    if ((rc = upload_taps(ctx, l, sigma, &ntap[l]))) return rc;
  }
  // first level of octave 0
  const float *src0 = img_dev;
  float *packed = nullptr;
  if (stride != w) {   // repack to contiguous planes
    packed = ctx->tmp_dev;
    MODS_HIP_CHECK(hipMemcpy2DAsync(packed, sizeof(float) * w, img_dev, sizeof(float) * stride, sizeof(float) * w,
                                    (size_t)h * n_img, hipMemcpyDeviceToDevice, ctx->stream));
    src0 = packed;
  }
  if (par.detectorType != MODS_DET_HESSIAN) {   // DoG / Harris: plain blurs, the response of every level from its own launches
    if (initial_blur) {
      if ((rc = blur_with_slot(ctx, src0, P.oct[0].blur[0], w, h, n_img, 0, ntap[0]))) return rc;
    } else {
      MODS_HIP_CHECK(hipMemcpyAsync(P.oct[0].blur[0], src0, sizeof(float) * (size_t)w * h * n_img, hipMemcpyDeviceToDevice, ctx->stream));
    }
    for (int oi = 0; oi < P.n_oct; oi++) {
      OctaveDev &o = P.oct[oi];
      if ((rc = launch_alt_response(ctx, o.blur[0], o.resp[0], o.w, o.h, n_img, 0, o.sigma[0]))) return rc;
      for (int l = 1; l < P.n_levels; l++) {
        if ((rc = blur_with_slot(ctx, o.blur[l - 1], o.blur[l], o.w, o.h, n_img, l, ntap[l]))) return rc;
        if ((rc = launch_alt_response(ctx, o.blur[l], o.resp[l], o.w, o.h, n_img, l, o.sigma[l - 1] * sigmaStep))) return rc;
        if (l == S && oi + 1 < P.n_oct) {
          OctaveDev &nx = P.oct[oi + 1];
          if ((rc = launch_resize_half(ctx, o.blur[l], nx.blur[0], o.w, o.h, nx.w, nx.h, n_img))) return rc;
        }
      }
    }
    return MODS_OK;
  }
  // every blur launch also writes the Hessian response of its output (fused kernel): the separate response launch is left for
  // the first level of the octaves that start from a decimated plane
  if (initial_blur) {
    if ((rc = blur_with_slot(ctx, src0, P.oct[0].blur[0], w, h, n_img, 0, ntap[0], P.oct[0].resp[0], P.oct[0].sigma[0] * P.oct[0].sigma[0]))) return rc;
  } else {
    MODS_HIP_CHECK(hipMemcpyAsync(P.oct[0].blur[0], src0, sizeof(float) * (size_t)w * h * n_img, hipMemcpyDeviceToDevice, ctx->stream));
  }
  // Octave k + 1 needs only level S of octave k.  With a batch of large planes the octaves from the third on (6 % of the pixels, but
  // a serial chain of ~12 short launches and the one-workgroup-per-image LDS kernel: 0.25 ms of the 1.2 ms a 16-image 1080p batch
  // takes) go to a side stream as soon as octaves 0 and 1 have produced their level S - those two octaves' levels up to S are issued
  // first - next to the remaining levels of octaves 0 and 1 and their NMS (the bulk of both); detect_run launches the small
  // octaves' NMS on the side stream too and joins before the compaction.  (Round 3 tried a fork with every octave's NMS after the
  // join: the side chain then only had one blur launch to hide under, 1.436 against 1.416 ms; with octave 1 - all of it, or its
  // last level and its NMS - on the side stream as well the side chain is the longer one: 1.05-1.06 against 0.99-1.01 ms.)  mods_ctx_pyramid_streams(ctx, 1)
  // keeps one stream.
  const bool fork = ctx->pyr_streams >= 2 && P.n_oct >= 2 && S < P.n_levels - 1 && (size_t)w * h * n_img >= ((size_t)4 << 20);
  ctx->pyr_side = false;
  ctx->pyr_forked = false;
  // the octaves from `first_lds` on fit LDS and are built by pyramid_lds_kernel in one launch
  int first_lds = P.n_oct;
  {
    bool taps_ok = P.n_levels <= kMaxLevels;
    for (int l = 1; l < P.n_levels; l++) taps_ok = taps_ok && ntap[l] >= 3 && ntap[l] <= 17 && ctx->taps_host_n[l] == ntap[l];
    if (taps_ok)
      for (int oi = P.n_oct - 1; oi >= 1 && P.oct[oi].w * P.oct[oi].h <= LDSP_CAP; oi--) {
        // (the decimated plane of an octave goes to the N plane: LDSP_CAP / 4 + 256 floats, short of it only for absurd aspect ratios)
        if (oi + 1 < P.n_oct && P.oct[oi + 1].w * P.oct[oi + 1].h > LDSP_CAP / 4 + 256) break;
        first_lds = oi;
      }
  }
  hipStream_t main_stream = ctx->stream;
  struct StreamRestore { mods_ctx *c; hipStream_t s; ~StreamRestore() { c->stream = s; } } restore{ctx, main_stream};
  // levels [l0, l1) of octave oi on ctx->stream (+ the decimation into the next octave behind level S)
  auto build_levels = [&](int oi, int l0, int l1) -> int {
    OctaveDev &o = P.oct[oi];
    for (int l = l0; l < l1; l++) {
      const float sigma = o.sigma[l - 1] * sigmaStep;   // pyramid.cpp:455-458
      if ((rc = blur_with_slot(ctx, o.blur[l - 1], o.blur[l], o.w, o.h, n_img, l, ntap[l], o.resp[l], sigma * sigma))) return rc;
      if (l == S && oi + 1 < P.n_oct) {
        OctaveDev &nx = P.oct[oi + 1];
        if ((rc = launch_resize_half_resp(ctx, o.blur[l], nx.blur[0], nx.resp[0], o.w, o.h, nx.w, nx.h, n_img, nx.sigma[0] * nx.sigma[0]))) return rc;
      }
    }
    return MODS_OK;
  };
  if (!initial_blur && first_lds > 0)      // (the first level of a later octave gets its response from the decimation launch)
    if ((rc = launch_hessian_response(ctx, P.oct[0].blur[0], P.oct[0].resp[0], P.oct[0].w, P.oct[0].h, n_img, P.oct[0].sigma[0] * P.oct[0].sigma[0]))) return rc;
  // octaves [0, n_main) stay on this stream; with the fork their levels up to S (what the next octave needs) come first
  const int n_main = fork ? std::min(2, first_lds) : first_lds;
  ctx->pyr_side_first = n_main;
  for (int oi = 0; oi < n_main; oi++)
    if ((rc = build_levels(oi, 1, fork ? S + 1 : P.n_levels))) return rc;
  if (fork && n_main < P.n_oct) {                                     // everything the later octaves need exists behind this event
    if (!ctx->stream2) {
      // the side chain is the scale space's critical path (every octave waits for the one before it; tools/trace_pyr.sh: its small
      // launches took twice their own time next to the large octaves' NMS): it runs at the highest stream priority
#ifndef PYR_SIDE_PRIO
#define PYR_SIDE_PRIO 1
#endif
      int prio_low = 0, prio_high = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
      MODS_HIP_CHECK(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, PYR_SIDE_PRIO ? prio_high : prio_low));
      MODS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      MODS_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    MODS_HIP_CHECK(hipEventRecord(ctx->ev_fork, main_stream));
    MODS_HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    ctx->pyr_side = true;
    ctx->pyr_forked = true;
  }
  if (fork)
    for (int oi = 0; oi < n_main; oi++)
      if ((rc = build_levels(oi, S + 1, P.n_levels))) return rc;
  if (ctx->pyr_side) ctx->stream = ctx->stream2;                      // the launch helpers issue on ctx->stream
  for (int oi = n_main; oi < first_lds; oi++)
    if ((rc = build_levels(oi, 1, P.n_levels))) return rc;
  if (first_lds < P.n_oct) {
    LdsPyramidPlan pl;
    pl.first = first_lds; pl.n_oct = P.n_oct; pl.n_levels = P.n_levels; pl.S = S;
    const OctaveDev &o = P.oct[first_lds];
    for (int l = 0; l < kMaxLevels; l++) { pl.ntap[l] = 0; pl.norm2[l] = 0.f; for (int i = 0; i < 17; i++) pl.taps[l].t[i] = 0.f; }
    for (int l = 1; l < P.n_levels; l++) {
      pl.ntap[l] = ntap[l];
      for (int i = 0; i < ntap[l]; i++) pl.taps[l].t[i] = ctx->taps_host[l][i];
      const float sigma = o.sigma[l - 1] * sigmaStep;        // pyramid.cpp:455-458 (the same for every octave)
      const float norm = sigma * sigma;
      pl.norm2[l] = norm * norm;
    }
    { const float norm = o.sigma[0] * o.sigma[0]; pl.norm2_first = norm * norm; }
    const size_t lds = sizeof(float) * (2 * LDSP_CAP + LDSP_CAP / 4 + 256);
    static DynLdsOnce once;
    MODS_HIP_CHECK(dyn_lds_once(once, (const void *)pyramid_lds_kernel, (int)lds, ctx->device));
    StageScope ts(ctx, MODS_STAGE_BLUR_SMALL, 0.0);
    hipLaunchKernelGGL(pyramid_lds_kernel, dim3(n_img), dim3(1024), lds, ctx->stream, ctx->pyr_dev, pl);
    MODS_HIP_CHECK(hipGetLastError());
  }
  return MODS_OK;
}

}  // namespace mods
