// Measurement-region extraction (DescribeRegions<>, non-fast branch) and SIFT / RootSIFT.
//
// Reference behaviour (file:line relative to the reference root):
//   DescribeRegions<SIFTDescriptor>           synth-detection.hpp:170-263
//   interpolate / gaussianBlurInplace         detectors/helpers.cpp:551-626, 726-731
//   photometricallyNormalize                  detectors/helpers.cpp:666-715
//   SIFTDescriptor                            matching/siftdesc.cpp:22-131, 133-158, 199-263, 346-400
//
// Two kernels per batch:
//   extract : one workgroup per region -> the ps x ps patch (before photometric normalisation),
//             written to HBM.  The reference samples a P2 x P2 region (P2 = 2*ceil(s*mrSize)+3), blurs
//             all of it and resamples 41 x 41 points on an axis-aligned grid.  Windows with P2 <= 80 (fewer
//             columns than the 2 ps grid lines) are blurred in full inside LDS (extract_small_kernel).  Of a
//             larger window only the blurred values at the <= 2ps grid rows x 2ps grid columns are ever read,
//             so its row pass runs on every row but only the needed columns and its column pass only on the
//             needed (row, column) pairs - each value computed exactly as the full blur would (big_* kernels).
//   sift    : one workgroup per region: photometric normalisation, gradients, 4x4x8 histogram,
//             normalisation, quantisation.
// Every floating-point accumulation the reference performs sequentially keeps its order: sample
// coordinates (each thread replays the additions up to its first sample), histogram bins (one lane per
// bin, pixels in raster order), photometric sums and descriptor norms (one lane).
#include "describe_common.hpp"
#include "detmath.hpp"
#include "device_util.hpp"

namespace mods {

constexpr int SMALL_CAP = 80;   // P2 limit of the LDS-resident extraction tier
#ifndef EXTRACT_T_LO
#define EXTRACT_T_LO 48     // size classes of the LDS tier: P2 <= 48, <= 64, <= SMALL_CAP
#endif
#ifndef EXTRACT_T_MID
#define EXTRACT_T_MID 64
#endif
#ifndef ES_MINB
#define ES_MINB 5                // workgroups of extract_small_kernel per CU the register budget is set for
#endif

// 8-byte aligned place at or behind p inside the (16-byte aligned) dynamic LDS block `base`, by pointer arithmetic: rounding the
// address as an integer would make the result a FLAT pointer (its accesses wait on both counters)
__device__ __forceinline__ double *lds_doubles(float *base, const void *p) {
  const int off = (int)((const float *)p - base);
  return (double *)(base + ((off + 1) & ~1));
}

struct RegionGeom {             // per-region constants of DescribeRegions
  int P2;                       // 0: direct branch (imageToPatchScale <= 0.4)
  float scale;                  // imageToPatchScale
  float fx, fy, f11, f12, f21, f22;
};

__device__ __forceinline__ RegionGeom region_geom(const mods_region &r, double desc_mr, int ps, int patch_rule) {
  RegionGeom g;
  g.fx = (float)r.x; g.fy = (float)r.y;
  g.f11 = (float)r.a11; g.f12 = (float)r.a12; g.f21 = (float)r.a21; g.f22 = (float)r.a22;
  if (patch_rule == 2) {   // fast branch of DescribeRegions, synth-detection.hpp:232-253: always the direct interpolation
    const double mrs = desc_mr * r.s;
    g.scale = (float)(double(2 * int(mrs) + 1) / (double)ps);
    g.P2 = 0;
    return g;
  }
  const float mrScale = (float)ceil(r.s * desc_mr);
  const int P = (patch_rule == 0 || (ps & 1)) ? 2 * int(mrScale) + 1 : 2 * int(mrScale);
  g.scale = float(P) / float(ps);
  g.P2 = ((double)g.scale > 0.4) ? P + 2 : 0;
  return g;
}

// Gaussian taps for sigma (getGaussianKernel CV_32F) into s_tap; resampling sequence X_i (= Y_i) of
// interpolate(smoothed, c0, c0, scale, 0, 0, scale) into s_seq and the two source indices of every
// grid line into s_cidx.  Needs 2 doubles of scratch (s_red).  Ends with a barrier.
__device__ void blur_setup(int P2, float scale, int ps, int n_tap, float *s_tap, float *s_seq, int *s_cidx, double *s_red) {
  const int tid = threadIdx.x;
  const double sig = (double)(1.5f * scale);
  const double scale2X = -0.5 / (sig * sig);
  for (int i = tid; i < n_tap; i += 256) {
    const double x = i - (n_tap - 1) * 0.5;
    s_tap[i] = (float)det_exp(scale2X * x * x);
  }
  if (tid == 64) {   // a different wave than the tap normalisation below: the sequential sum only, the indices follow in parallel
    const float c0 = (float)(P2 >> 1);
    float v = c0 - (float)(ps / 2) * scale;
    for (int i = 0; i < ps; i++) { s_seq[i] = v; v += scale; }
  }
  __syncthreads();
  if (tid == 0) {
    double sum = 0;
    for (int i = 0; i < n_tap; i++) sum += s_tap[i];
    s_red[0] = 1. / sum;
  }
  if (tid >= 64 && tid < 64 + ps) {
    const int i = tid - 64;
    const int fl = (int)floorf(s_seq[i]);
    int i0 = fl, i1 = fl + 1;
    i0 = i0 < 0 ? 0 : (i0 > P2 - 1 ? P2 - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 > P2 - 1 ? P2 - 1 : i1);
    s_cidx[2 * i] = i0;
    s_cidx[2 * i + 1] = i1;
  }
  __syncthreads();
  for (int i = tid; i < n_tap; i += 256) s_tap[i] = (float)(s_tap[i] * s_red[0]);
  __syncthreads();
}

// Everything blur_setup makes is a function of (P2, ps): scale = (P2 - 2) / ps (region_geom), so the LDS tier reads it from a table
// with one entry per P2, written by blur_setup itself (bit-identical by construction): round 5 measured the per-region set-up -
// 41 sequential additions, a sequential tap sum, double-precision exponentials, three barriers - at 15 % of extract_small_kernel's
// time (EXTRACT_PROF).  Entry layout (floats): taps [0, 32) | seq [32, 96) | cidx (ints) [96, 224).
constexpr int BT_ENTRY = 256, BT_TAP = 0, BT_SEQ = 32, BT_CIDX = 96;
__global__ __launch_bounds__(256) void blur_table_kernel(int ps, float *__restrict__ table) {
  __shared__ float s_tap[32];
  __shared__ float s_seq[64];
  __shared__ int s_cidx[128];
  __shared__ double s_red[2];
  const int P2 = blockIdx.x;
  if (P2 < 3) return;
  const float scale = float(P2 - 2) / float(ps);              // region_geom: g.scale = float(P) / float(ps), P2 = P + 2
  const int n_tap = ((int)(2.0 * 3.0 * (1.5f * scale) + 1.0)) | 1;
  if (n_tap > 31) return;                                     // (not reached for P2 <= SMALL_CAP and the patch sizes this tier takes)
  blur_setup(P2, scale, ps, n_tap, s_tap, s_seq, s_cidx, s_red);
  float *e = table + (size_t)P2 * BT_ENTRY;
  const int t = threadIdx.x;
  if (t < 32) e[BT_TAP + t] = t < n_tap ? s_tap[t] : 0.f;
  if (t >= 64 && t < 64 + ps) e[BT_SEQ + (t - 64)] = s_seq[t - 64];
  if (t >= 128 && t < 128 + 2 * ps && t - 128 < 128) ((int *)e)[BT_CIDX + (t - 128)] = s_cidx[t - 128];
}
// (re)built on the context's stream when the patch size of the extraction changes: stream-ordered behind the launches that read the
// previous table
int launch_blur_table(mods_ctx *ctx, int ps) {
  if (ctx->blur_table_ps == ps && ctx->blur_table_dev) return MODS_OK;
  if (!ctx->blur_table_dev) MODS_HIP_CHECK(hipMalloc(&ctx->blur_table_dev, sizeof(float) * (SMALL_CAP + 1) * BT_ENTRY));
  mods::dev_state_changed(ctx);      // (device state changes: the next detect + describe call is not a repeat - capi.hip: dd_run)
  hipLaunchKernelGGL(blur_table_kernel, dim3(SMALL_CAP + 1), dim3(256), 0, ctx->stream, ps, ctx->blur_table_dev);
  MODS_HIP_CHECK(hipGetLastError());
  ctx->blur_table_ps = ps;
  return MODS_OK;
}

// row stride of the row-pass strip T (2 * ps columns, padded so that two adjacent column pairs are one aligned float4)
__host__ __device__ __forceinline__ int t_stride(int ps) { return (2 * ps + 3) & ~3; }

// column pass value at (needed row y, strip column q): centre tap first, symmetric pairs
__device__ __forceinline__ float col_value(const float *T, int P2, int ps2, int y, int q, int r_tap, const float *s_tap) {
  float s = s_tap[r_tap] * T[(size_t)y * ps2 + q];
  if (y - r_tap >= 0 && y + r_tap <= P2 - 1) {
    const float *c = T + (size_t)y * ps2 + q;
    int j = 1;
    for (; j + 1 <= r_tap; j += 2) {
      const float p0 = c[(size_t)j * ps2], m0 = c[-(ptrdiff_t)j * ps2];
      const float p1 = c[(size_t)(j + 1) * ps2], m1 = c[-(ptrdiff_t)(j + 1) * ps2];
      s = fmaf(s_tap[r_tap + j], p0 + m0, s);
      s = fmaf(s_tap[r_tap + j + 1], p1 + m1, s);
    }
    for (; j <= r_tap; j++) s = fmaf(s_tap[r_tap + j], (c[(size_t)j * ps2] + c[-(ptrdiff_t)j * ps2]), s);
  } else {
    for (int j = 1; j <= r_tap; j++) {
      int yp = y + j; yp = yp > P2 - 1 ? P2 - 1 : yp;
      int ym = y - j; ym = ym < 0 ? 0 : ym;
      s = fmaf(s_tap[r_tap + j], (T[(size_t)yp * ps2 + q] + T[(size_t)ym * ps2 + q]), s);
    }
  }
  return s;
}

// column pass fused with interpolate(smoothed, c0, c0, scale, 0, 0, scale): every blurred value is used by exactly one output
// pixel, so it is produced where it is consumed.  One call makes the two horizontally adjacent output pixels (j, 2m) and
// (j, 2m + 1): their four strip columns 4m .. 4m + 3 are one float4 per strip row.  For an output row between the grid rows y0
// and y1 = y0 + 1 the row windows of the two share all but one row (also when rows are clamped at the edge of the strip):
// rows are loaded once and feed all sums, each in the reference's order (centre tap, then symmetric pairs).
__device__ __forceinline__ void col_resample_pair(const float *T, int ts, int P2, int ps, int r_tap, bool touch2, const float *s_tap,
                                                  const float *s_seq, const int *s_cidx, int j, int m, float *out0, float *out1) {
  const int i0 = 2 * m, i1 = min(2 * m + 1, ps - 1);
  const float WY = s_seq[j], WX0 = s_seq[i0], WX1 = s_seq[i1];
  const int y = touch2 ? (int)floorf(WY) : (int)WY;
  const int x0 = touch2 ? (int)floorf(WX0) : (int)WX0, x1 = touch2 ? (int)floorf(WX1) : (int)WX1;
  const int y0 = s_cidx[2 * j], y1 = s_cidx[2 * j + 1];
  float4 r0, r1;   // blurred values at grid rows y0 / y1, strip columns 4m .. 4m+3
  if (y1 == y0 + 1) {
    const float *c = T + 4 * m;
    const float4 m0 = *(const float4 *)(c + y0 * ts), m1 = *(const float4 *)(c + y1 * ts);
    const float tc = s_tap[r_tap];
    r0 = make_float4(tc * m0.x, tc * m0.y, tc * m0.z, tc * m0.w);
    r1 = make_float4(tc * m1.x, tc * m1.y, tc * m1.z, tc * m1.w);
    float4 up = m1;                                 // row y0 + jj       (jj = 1)
    float4 dn_prev = m0;                            // row y1 - jj       (jj = 1)
    int jj = 1;
    for (; jj + 3 <= r_tap; jj += 4) {              // 8 loads in flight, then the sums in tap order
      float4 u[4], d[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int yu = min(y1 + jj + q, P2 - 1), yd = max(y0 - jj - q, 0);
        u[q] = *(const float4 *)(c + yu * ts);      // row y1 + jj  (= y0 + jj + 1)
        d[q] = *(const float4 *)(c + yd * ts);      // row y0 - jj
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float t = s_tap[r_tap + jj + q];
        r0.x = fmaf(t, up.x + d[q].x, r0.x); r0.y = fmaf(t, up.y + d[q].y, r0.y); r0.z = fmaf(t, up.z + d[q].z, r0.z); r0.w = fmaf(t, up.w + d[q].w, r0.w);
        r1.x = fmaf(t, u[q].x + dn_prev.x, r1.x); r1.y = fmaf(t, u[q].y + dn_prev.y, r1.y); r1.z = fmaf(t, u[q].z + dn_prev.z, r1.z); r1.w = fmaf(t, u[q].w + dn_prev.w, r1.w);
        up = u[q]; dn_prev = d[q];
      }
    }
    for (; jj <= r_tap; jj++) {
      const int yu = min(y1 + jj, P2 - 1), yd = max(y0 - jj, 0);
      const float4 u = *(const float4 *)(c + yu * ts), d = *(const float4 *)(c + yd * ts);
      const float t = s_tap[r_tap + jj];
      r0.x = fmaf(t, up.x + d.x, r0.x); r0.y = fmaf(t, up.y + d.y, r0.y); r0.z = fmaf(t, up.z + d.z, r0.z); r0.w = fmaf(t, up.w + d.w, r0.w);
      r1.x = fmaf(t, u.x + dn_prev.x, r1.x); r1.y = fmaf(t, u.y + dn_prev.y, r1.y); r1.z = fmaf(t, u.z + dn_prev.z, r1.z); r1.w = fmaf(t, u.w + dn_prev.w, r1.w);
      up = u; dn_prev = d;
    }
  } else {   // both grid rows clamped onto one strip row (output rows outside the strip)
    r0.x = col_value(T, P2, ts, y0, 4 * m, r_tap, s_tap);     r0.y = col_value(T, P2, ts, y0, 4 * m + 1, r_tap, s_tap);
    r0.z = col_value(T, P2, ts, y0, 4 * m + 2, r_tap, s_tap); r0.w = col_value(T, P2, ts, y0, 4 * m + 3, r_tap, s_tap);
    r1.x = col_value(T, P2, ts, y1, 4 * m, r_tap, s_tap);     r1.y = col_value(T, P2, ts, y1, 4 * m + 1, r_tap, s_tap);
    r1.z = col_value(T, P2, ts, y1, 4 * m + 2, r_tap, s_tap); r1.w = col_value(T, P2, ts, y1, 4 * m + 3, r_tap, s_tap);
  }
  const bool oky = !touch2 || (WY >= 0 && y < P2 - 1);
  {
    const float wx = WX0 - (float)x0;
    const float I1 = wx * (r0.y - r0.x) + r0.x;
    const float v = (WY - y) * (wx * (r1.y - r1.x) + r1.x - I1) + I1;
    *out0 = (oky && (!touch2 || (WX0 >= 0 && x0 < P2 - 1))) ? v : 0.f;
  }
  {
    const float wx = WX1 - (float)x1;
    const float I1 = wx * (r0.w - r0.z) + r0.z;
    const float v = (WY - y) * (wx * (r1.w - r1.z) + r1.z - I1) + I1;
    *out1 = (oky && (!touch2 || (WX1 >= 0 && x1 < P2 - 1))) ? v : 0.f;
  }
}

// ---- the LDS tier's blur as a FULL separable blur of the window (P2 < 2 ps: fewer blurred values than the 4 per output pixel) ----
// A window of the LDS tier has P2 <= 80 < 2 * 41 columns, so the "needed columns" of the strip form are all of them, several times
// over (82 strip columns for 19..80 window columns), and its column pass makes four blurred values per OUTPUT pixel where the
// window has only P2 * P2 of them (P2 = 30: 900 against 6 724).  Here every blurred value is made once - each by the same
// operations in the same order as the strip form (and as cv::GaussianBlur: row pass left to right, 5 taps centre then pairs;
// column pass centre then symmetric pairs, REPLICATE borders) - and the 41 x 41 resampling reads the blurred window.
// column stride of the transposed window S: a multiple of 4 whose quarter is odd - the float4 reads of the row pass (lanes on
// consecutive columns) and the tile-wise writes of the sampler (8 columns x 8 rows per wave) then spread over all banks
__host__ __device__ __forceinline__ int odd4(int n) { const int q = (n + 3) >> 2; return 4 * (q | 1); }
// e / d for 0 <= e < 2^16, 0 < d <= 256 without the integer division's expansion (rcp = 1.f / d): (e + 0.5) / d is at least
// 0.5 / d away from an integer, far more than the product's rounding error
__device__ __forceinline__ int small_div(int e, float rcp) { return (int)(((float)e + 0.5f) * rcp); }

// row pass, all columns: T[y][x] (row-major, row stride ts) from the transposed window St[x][y] (column stride st)
__device__ __forceinline__ void row_pass_full(const float *St, float *T, int P2, int st, int ts, int n_tap, const float *s_tap) {
  const int r_tap = n_tap >> 1, nq = (P2 + 3) >> 2;
  const float rcp = 1.0f / (float)P2;
  for (int e = threadIdx.x; e < nq * P2; e += 256) {
    const int yq = small_div(e, rcp), x = e - yq * P2, y = 4 * yq;
    const float *p = St + y;
    float4 s;
    if (n_tap == 5) {   // SymmRowSmallFilter: centre tap, then the symmetric pairs
      float4 v[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        int xb = x - 2 + u; xb = xb < 0 ? 0 : (xb > P2 - 1 ? P2 - 1 : xb);
        v[u] = *(const float4 *)(p + xb * st);
      }
      const float k0 = s_tap[2], k1 = s_tap[3], k2 = s_tap[4];
      s = make_float4(v[2].x * k0, v[2].y * k0, v[2].z * k0, v[2].w * k0);
      s.x = fmaf(v[1].x + v[3].x, k1, s.x); s.y = fmaf(v[1].y + v[3].y, k1, s.y); s.z = fmaf(v[1].z + v[3].z, k1, s.z); s.w = fmaf(v[1].w + v[3].w, k1, s.w);
      s.x = fmaf(v[0].x + v[4].x, k2, s.x); s.y = fmaf(v[0].y + v[4].y, k2, s.y); s.z = fmaf(v[0].z + v[4].z, k2, s.z); s.w = fmaf(v[0].w + v[4].w, k2, s.w);
    } else {
      int xa = x - r_tap; xa = xa < 0 ? 0 : xa;
      float4 c = *(const float4 *)(p + xa * st);
      float t = s_tap[0];
      s = make_float4(t * c.x, t * c.y, t * c.z, t * c.w);
      int j = 1;
      for (; j + 1 < n_tap; j += 2) {
        int xb = x - r_tap + j; xb = xb < 0 ? 0 : (xb > P2 - 1 ? P2 - 1 : xb);
        int xc = x - r_tap + j + 1; xc = xc < 0 ? 0 : (xc > P2 - 1 ? P2 - 1 : xc);
        const float4 c0 = *(const float4 *)(p + xb * st), c1 = *(const float4 *)(p + xc * st);
        t = s_tap[j];
        s.x = fmaf(t, c0.x, s.x); s.y = fmaf(t, c0.y, s.y); s.z = fmaf(t, c0.z, s.z); s.w = fmaf(t, c0.w, s.w);
        t = s_tap[j + 1];
        s.x = fmaf(t, c1.x, s.x); s.y = fmaf(t, c1.y, s.y); s.z = fmaf(t, c1.z, s.z); s.w = fmaf(t, c1.w, s.w);
      }
      for (; j < n_tap; j++) {
        int xb = x - r_tap + j; xb = xb < 0 ? 0 : (xb > P2 - 1 ? P2 - 1 : xb);
        c = *(const float4 *)(p + xb * st);
        t = s_tap[j];
        s.x = fmaf(t, c.x, s.x); s.y = fmaf(t, c.y, s.y); s.z = fmaf(t, c.z, s.z); s.w = fmaf(t, c.w, s.w);
      }
    }
    float *o = T + y * ts + x;
    o[0] = s.x;
    if (y + 1 < P2) o[ts] = s.y;
    if (y + 2 < P2) o[2 * ts] = s.z;
    if (y + 3 < P2) o[3 * ts] = s.w;
  }
}
// column pass, all rows: B[y][x] from T[y][x], four adjacent columns per item (centre tap, then the symmetric pairs)
__device__ __forceinline__ void col_pass_full(const float *T, float *B, int P2, int ts, int n_tap, const float *s_tap) {
  const int r_tap = n_tap >> 1, nx = ts >> 2;
  const float rcp = 1.0f / (float)nx;
  for (int e = threadIdx.x; e < nx * P2; e += 256) {
    const int y = small_div(e, rcp), xq = e - y * nx;
    const float *c = T + 4 * xq;
    const float4 m = *(const float4 *)(c + y * ts);
    const float tc = s_tap[r_tap];
    float4 s = make_float4(tc * m.x, tc * m.y, tc * m.z, tc * m.w);
    int j = 1;
    for (; j + 1 <= r_tap; j += 2) {
      const int yu0 = min(y + j, P2 - 1), yd0 = max(y - j, 0), yu1 = min(y + j + 1, P2 - 1), yd1 = max(y - j - 1, 0);
      const float4 u0 = *(const float4 *)(c + yu0 * ts), d0 = *(const float4 *)(c + yd0 * ts);
      const float4 u1 = *(const float4 *)(c + yu1 * ts), d1 = *(const float4 *)(c + yd1 * ts);
      float t = s_tap[r_tap + j];
      s.x = fmaf(t, u0.x + d0.x, s.x); s.y = fmaf(t, u0.y + d0.y, s.y); s.z = fmaf(t, u0.z + d0.z, s.z); s.w = fmaf(t, u0.w + d0.w, s.w);
      t = s_tap[r_tap + j + 1];
      s.x = fmaf(t, u1.x + d1.x, s.x); s.y = fmaf(t, u1.y + d1.y, s.y); s.z = fmaf(t, u1.z + d1.z, s.z); s.w = fmaf(t, u1.w + d1.w, s.w);
    }
    for (; j <= r_tap; j++) {
      const int yu = min(y + j, P2 - 1), yd = max(y - j, 0);
      const float4 u = *(const float4 *)(c + yu * ts), d = *(const float4 *)(c + yd * ts);
      const float t = s_tap[r_tap + j];
      s.x = fmaf(t, u.x + d.x, s.x); s.y = fmaf(t, u.y + d.y, s.y); s.z = fmaf(t, u.z + d.z, s.z); s.w = fmaf(t, u.w + d.w, s.w);
    }
    *(float4 *)(B + y * ts + 4 * xq) = s;
  }
}
// interpolate(smoothed, c0, c0, scale, 0, 0, scale) -> ps x ps from the blurred window B (the arithmetic of col_resample_pair)
__device__ __forceinline__ void resample_full(const float *B, int P2, int ts, int ps, float scale, const float *s_seq, const int *s_cidx,
                                              float *__restrict__ out) {
  const float c0 = (float)(P2 >> 1);
  const bool touch2 = check_borders(P2, P2, c0, c0, scale, 0.f, 0.f, scale, ps, ps);
  const float rcp = 1.0f / (float)ps;
  for (int e = threadIdx.x; e < ps * ps; e += 256) {
    const int j = small_div(e, rcp), i = e - j * ps;
    const float WY = s_seq[j], WX = s_seq[i];
    const int2 yy = *(const int2 *)(s_cidx + 2 * j), xx = *(const int2 *)(s_cidx + 2 * i);
    const int y = touch2 ? (int)floorf(WY) : (int)WY, x = touch2 ? (int)floorf(WX) : (int)WX;
    const float *b0 = B + yy.x * ts, *b1 = B + yy.y * ts;
    const float r00 = b0[xx.x], r01 = b0[xx.y], r10 = b1[xx.x], r11 = b1[xx.y];
    const float wx = WX - (float)x;
    const float I1 = wx * (r01 - r00) + r00;
    const float v = (WY - y) * (wx * (r11 - r10) + r10 - I1) + I1;
    const bool ok = !touch2 || (WY >= 0 && y < P2 - 1 && WX >= 0 && x < P2 - 1);
    out[e] = ok ? v : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// extract, LDS tier: direct branch and P2 <= SMALL_CAP.  grid = (N, n_img), block = 256.
// dynamic LDS: S cap*odd4(cap) (transposed window, then the blurred window) | T cap*cap4 (row pass) | seq 2ps | cidx 2ps | taps 32 | red 2 doubles
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, ES_MINB) void extract_small_kernel(const float *__restrict__ img_all, DescConst k,
                                                            const mods_region *__restrict__ reg_all, const int *__restrict__ items,
                                                            const int *__restrict__ n_items_dev, int items_cap,
                                                            float *__restrict__ patches, const float *__restrict__ blur_table) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.desc_ps, pp = ps * ps, ps2 = 2 * ps;
  const int cap = k.p2_hi > 4 ? k.p2_hi : 4;       // (the direct branch writes to the patch store, not to LDS)
  float *s_S = smem;
  float *s_T = s_S + cap * odd4(cap);
  float *s_seq = s_T + cap * ((cap + 3) & ~3);
  int *s_cidx = (int *)(s_seq + ps2);
  float *s_tap = (float *)(s_cidx + ps2);
  double *s_red = lds_doubles(smem, s_tap + 32);
  const int n = min(*n_items_dev, items_cap);
#ifdef EXTRACT_PROF
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pl = __builtin_amdgcn_s_memtime();
  int pn = 0;
#define PROF_MARK(i) { __syncthreads(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pl; pl = t_; }
#else
#define PROF_MARK(i)
#endif
  for (int it = blockIdx.x; it < n; it += gridDim.x) {
    const int code = items[it], b = code >> 17, ri = code & 0x1ffff;
    const float *img = img_all + (size_t)k.w * k.h * b;
    const RegionGeom g = region_geom(reg_all[(size_t)b * k.max_reg + ri], k.desc_mr, ps, k.patch_rule);
    float *out = patches + ((size_t)b * k.reg_cap + ri) * pp;
    __syncthreads();
    PROF_MARK(0)
    if (g.P2 > 0) {
      const int n_tap = ((int)(2.0 * 3.0 * (1.5f * g.scale) + 1.0)) | 1;
      const int stride = odd4(g.P2);
      sample_tiles(img, k.w, k.h, g.fx, g.fy, g.f11, g.f12, g.f21, g.f22, g.P2, threadIdx.x >> 6, 4,
                   [&](int row, int col, float v) { s_S[col * stride + row] = v; });   // transposed: see row_pass_full
      PROF_MARK(1)
      if (blur_table && n_tap <= 31 && ps <= 63) {     // taps, resampling sequence and source indices of this P2 from the table
        const float *e = blur_table + (size_t)g.P2 * BT_ENTRY;
        const int t = threadIdx.x;
        if (t < 32) s_tap[t] = e[BT_TAP + t];
        if (t >= 64 && t < 64 + ps) s_seq[t - 64] = e[BT_SEQ + (t - 64)];
        if (t >= 128 && t < 128 + 2 * ps && t - 128 < 128) s_cidx[t - 128] = ((const int *)e)[BT_CIDX + (t - 128)];
        __syncthreads();
      } else blur_setup(g.P2, g.scale, ps, n_tap, s_tap, s_seq, s_cidx, s_red);
      PROF_MARK(2)
      const int ts = (g.P2 + 3) & ~3;
      row_pass_full(s_S, s_T, g.P2, stride, ts, n_tap, s_tap);
      __syncthreads();
      PROF_MARK(3)
      col_pass_full(s_T, s_S, g.P2, ts, n_tap, s_tap);        // the blurred window takes the sampled one's place
      __syncthreads();
      resample_full(s_S, g.P2, ts, ps, g.scale, s_seq, s_cidx, out);
      PROF_MARK(4)
#ifdef EXTRACT_PROF
      pn++;
#endif
    } else {
      // direct branch: interpolate(img, x, y, A*scale) -> ps x ps
      sample_tiles(img, k.w, k.h, g.fx, g.fy, g.f11 * g.scale, g.f12 * g.scale, g.f21 * g.scale, g.f22 * g.scale, ps, threadIdx.x >> 6, 4,
                   [&](int row, int col, float v) { out[row * ps + col] = v; });
    }
  }
#ifdef EXTRACT_PROF
  if (threadIdx.x == 0 && (blockIdx.x % 512) == 7)
    printf("extract_small prof: block %d tier %d regions %d cycles: skip %llu sample %llu setup %llu rowpass %llu colres %llu\n", blockIdx.x,
           k.p2_hi, pn, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
}

// ---------------------------------------------------------------------------------------
// extract, HBM tier (P2 > k.p2_hi): phase kernels over device-built work lists, one WAVE per work item, no
// barriers and no LDS, so that the loads of many items are in flight per CU.  Per region a slab is reserved:
//   header (taps n_tap | seq ps | cidx 2ps ints, padded) | St (P2 columns x P2r) | T (P2 x 2ps)
// St is S = interpolate(img, x, y, A) stored TRANSPOSED (column-major, column stride P2r = P2 rounded up to 4):
// the row pass then reads 4 consecutive rows of a column as one float4, contiguous across the lanes.
// ---------------------------------------------------------------------------------------
constexpr int BIG_SROWS = 16;         // rows per sample item
constexpr int BIG_RROWS = 64;         // rows per row-pass item
constexpr int BIG_RLOADS = 128;       // float4 loads per lane a row-pass item aims at
struct BigRegion { int img, ri, P2, n_tap; unsigned long long slab; float scale; int P2r; };   // slab: float offset in the pool
struct BigLists {                     // device-resident bookkeeping, zeroed before every batch
  int n_regions, n_sitems, n_ritems, n_fitems;
  unsigned long long pool_used;
  int n_small[3];                   // regions of the three LDS tiers (P2 <= t_lo, <= t_mid, <= p2_hi), over all images
  int sift_next;                    // next work unit of sift_wave2_kernel
};
// regions up to this size take the fused sample + row-pass kernel (S stays in LDS, no S slab); larger ones the phase kernels
#ifndef BIG_FUSE_P2_MAX
#define BIG_FUSE_P2_MAX 1024   // measured (round 4, profiles/r04_extract_variants.log): 256 -> 11.01, 512 -> 10.67, 1024 -> 10.43 ms per 16-image describe leg
#endif
constexpr int BIG_FUSE_P2 = BIG_FUSE_P2_MAX;
constexpr int BIG_FUSE_TAPS = 320;     // taps the fused kernel stages in LDS
#ifndef BIG_FUSE_KB
#define BIG_FUSE_KB 32
#endif
// rows per fused item: the largest power of two (<= 64) with rows * P2 floats <= BIG_FUSE_KB KB (P2 <= 1024)
__device__ __forceinline__ int big_fuse_rows(int P2) {
  const int budget = BIG_FUSE_KB * 256;        // floats
  int r = 64;
  while (r > 4 && r * P2 > budget) r >>= 1;
  return r;
}
__device__ __forceinline__ bool big_is_fused(int P2, int n_tap) { return P2 <= BIG_FUSE_P2 && P2 <= 1024 && n_tap <= BIG_FUSE_TAPS; }

__device__ __forceinline__ int big_hdr_floats(int n_tap, int ps) { return (n_tap + 3 * ps + 8 + 3) & ~3; }
__device__ __forceinline__ unsigned long long big_s_floats(int P2, int P2r, int n_tap) { return big_is_fused(P2, n_tap) ? 0ull : (unsigned long long)P2 * P2r; }
// a row-pass item covers `steps` groups of 4 column pairs
__device__ __forceinline__ int big_rsteps(int n_tap) { const int v = BIG_RLOADS / (n_tap + 1); return v < 1 ? 1 : v; }

// grid = (CLASSIFY_BLOCKS, n_img), block 1024, grid-stride over the image's regions: reserve slabs, emit the work items.
// One returning atomic per WORKGROUP and counter: the amounts of a workgroup's regions are scanned inside each wave, the waves'
// totals meet in LDS, eight threads reserve the workgroup's ranges and every region takes its share.  (Until round 6 every wave
// with a region queued up to eight returning atomics on the same eight words of BigLists - ~20 000 per 16-image batch, serialised
// in the L2 at ~12 ns each - and the grid had a workgroup per 256 slots of the region CAPACITY: 0.17 ms per batch for a kernel
// that moves 2 MB.)
constexpr int CLASSIFY_BLOCKS = 16;
__global__ __launch_bounds__(1024) void big_classify_kernel(DescConst k, const mods_region *__restrict__ reg_all,
                                                           const int *__restrict__ reg_count, BigLists *__restrict__ bl,
                                                           BigRegion *__restrict__ regions, int2 *__restrict__ sitems,
                                                           int2 *__restrict__ ritems, int2 *__restrict__ fitems, int *__restrict__ small_items, int small_cap_items, int t_lo, int t_mid,
                                                           int max_regions, int max_items,
                                                           unsigned long long pool_elems, int *__restrict__ err_flag) {
  __shared__ int s_w[16][8];                       // per wave: totals of the seven integer amounts, then their bases
  __shared__ unsigned long long s_need[16];
  const int b = blockIdx.y;
  int n = reg_count[b];
  if (n > k.reg_cap) n = k.reg_cap;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int ri0 = blockIdx.x * 1024; ri0 < n; ri0 += gridDim.x * 1024) {
    const int ri = ri0 + threadIdx.x;
    const bool have = ri < n;
    RegionGeom g;
    g.P2 = 0;
    if (have) g = region_geom(reg_all[(size_t)b * k.max_reg + ri], k.desc_mr, k.desc_ps, k.patch_rule);
    // LDS tier: work lists of the three size classes (image << 17 | region); a blur wider than the 32 taps the LDS tier stages
    // (patch sizes below ~24 only) sends a region to the HBM tier whatever its size
    const int n_tap = have && g.P2 > 0 ? ((int)(2.0 * 3.0 * (1.5f * g.scale) + 1.0)) | 1 : 0;
    const bool lds_tier = have && g.P2 <= k.p2_hi && n_tap <= 32;
    const int tier = !lds_tier ? -1 : (g.P2 <= t_lo ? 0 : (g.P2 <= t_mid ? 1 : 2));
    const bool big = have && !lds_tier;
    const int P2r = (g.P2 + 3) & ~3;
    const unsigned long long need = ((unsigned long long)big_hdr_floats(n_tap, k.desc_ps) + big_s_floats(g.P2, P2r, n_tap) +
                                     (unsigned long long)g.P2 * t_stride(k.desc_ps) + 3ull) & ~3ull;
    const bool fused = big_is_fused(g.P2, n_tap);
    const int f_rows = big_fuse_rows(g.P2);
    const int f_chunks = fused ? (g.P2 + f_rows - 1) / f_rows : 0;
    const int s_chunks = fused ? 0 : (g.P2 + BIG_SROWS - 1) / BIG_SROWS;
    const int r_chunks = fused ? 0 : (g.P2 + BIG_RROWS - 1) / BIG_RROWS;
    const int steps = (k.desc_ps + 3) / 4, per = big_rsteps(n_tap), r_parts = (steps + per - 1) / per;
    // amounts: [0..2] one slot of a small list, [3] a region, [4] sample items, [5] row-pass items, [6] fused items; + pool floats
    int a[7] = {tier == 0, tier == 1, tier == 2, big ? 1 : 0, big ? s_chunks : 0, big ? r_chunks * r_parts : 0, big ? f_chunks : 0};
    const unsigned long long a_need = big ? need : 0ull;
    int p[7];
    // the three list slots by ballots, the four item amounts and the pool share by inclusive scans over the wave
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const unsigned long long m = __ballot(a[t] != 0);
      p[t] = __popcll(m & ((2ull << lane) - 1ull));
    }
#pragma unroll
    for (int t = 3; t < 7; t++) p[t] = a[t];
    unsigned long long p_need = a_need;
    for (int d = 1; d < 64; d <<= 1) {
      int u[4];
#pragma unroll
      for (int t = 0; t < 4; t++) u[t] = __shfl_up(p[3 + t], d);
      const unsigned lo = __shfl_up((unsigned)p_need, d), hi = __shfl_up((unsigned)(p_need >> 32), d);
      if (lane >= d) {
#pragma unroll
        for (int t = 0; t < 4; t++) p[3 + t] += u[t];
        p_need += ((unsigned long long)hi << 32) | lo;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int t = 0; t < 7; t++) s_w[wv][t] = p[t];
      s_need[wv] = p_need;
    }
    __syncthreads();
    if (threadIdx.x < 8) {             // thread t: counter t of the workgroup (7 = the pool)
      const int t = threadIdx.x;
      if (t < 7) {
        int tot = 0;
        for (int q = 0; q < 16; q++) tot += s_w[q][t];
        int *ctr = t < 3 ? &bl->n_small[t] : (t == 3 ? &bl->n_regions : (t == 4 ? &bl->n_sitems : (t == 5 ? &bl->n_ritems : &bl->n_fitems)));
        int base = tot ? atomicAdd(ctr, tot) : 0;
        for (int q = 0; q < 16; q++) { const int c = s_w[q][t]; s_w[q][t] = base; base += c; }
      } else {
        unsigned long long tot = 0;
        for (int q = 0; q < 16; q++) tot += s_need[q];
        unsigned long long base = tot ? atomicAdd(&bl->pool_used, tot) : 0ull;
        for (int q = 0; q < 16; q++) { const unsigned long long c = s_need[q]; s_need[q] = base; base += c; }
      }
    }
    __syncthreads();
    if (tier >= 0) {
      const int pos = s_w[wv][tier] + p[tier] - 1;
      if (pos < small_cap_items) small_items[(size_t)tier * small_cap_items + pos] = (b << 17) | ri;
    }
    if (big) {
      const int li = s_w[wv][3] + p[3] - a[3], s0 = s_w[wv][4] + p[4] - a[4], r0 = s_w[wv][5] + p[5] - a[5], f0 = s_w[wv][6] + p[6] - a[6];
      const unsigned long long off = s_need[wv] + p_need - a_need;
      if (li >= max_regions || off + need > pool_elems || s0 + s_chunks > max_items || r0 + r_chunks * r_parts > max_items ||
          f0 + f_chunks > max_items || g.P2 >= 65536 || n_tap > k.tap_cap) {
        atomicExch(err_flag, 1);
      } else {
        BigRegion br;
        br.img = b; br.ri = ri; br.P2 = g.P2; br.n_tap = n_tap; br.slab = off; br.scale = g.scale; br.P2r = P2r;
        regions[li] = br;
        for (int c = 0; c < f_chunks; c++) fitems[f0 + c] = make_int2(li, c * f_rows);
        for (int c = 0; c < s_chunks; c++) sitems[s0 + c] = make_int2(li, c * BIG_SROWS);
        for (int c = 0; c < r_chunks; c++)
          for (int q = 0; q < r_parts; q++) ritems[r0 + c * r_parts + q] = make_int2(li, c * BIG_RROWS | (q * per << 16));
      }
    }
    __syncthreads();                   // s_w / s_need are rewritten by the next round
  }
}

// grid-stride over the big regions, block 256: taps / resampling sequence into the slab header
__global__ __launch_bounds__(256) void big_setup_kernel(DescConst k, const BigLists *__restrict__ bl, const BigRegion *__restrict__ regions,
                                                        int max_regions, float *__restrict__ pool, const int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // taps tap_cap | seq ps | cidx 2ps | red
  if (*err_flag) return;   // a list or the pool overflowed in big_classify_kernel: the host reports it
  const int ps = k.desc_ps, ps2 = 2 * ps;
  float *s_tap = smem;
  float *s_seq = s_tap + k.tap_cap;
  int *s_cidx = (int *)(s_seq + ps2);
  double *s_red = lds_doubles(smem, s_cidx + ps2);
  const int n = min(bl->n_regions, max_regions);
  for (int li = blockIdx.x; li < n; li += gridDim.x) {
    const BigRegion br = regions[li];
    __syncthreads();
    blur_setup(br.P2, br.scale, ps, br.n_tap, s_tap, s_seq, s_cidx, s_red);
    float *hdr = pool + br.slab;
    for (int i = threadIdx.x; i < br.n_tap; i += 256) hdr[i] = s_tap[i];
    for (int i = threadIdx.x; i < ps; i += 256) hdr[br.n_tap + i] = s_seq[i];
    for (int i = threadIdx.x; i < ps2; i += 256) ((int *)(hdr + br.n_tap + ps))[i] = s_cidx[i];
  }
}

// the wave's work item (uniform): grid-stride in units of waves
#define BIG_WAVE_LOOP(n_items, it) \
  for (int it = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))); it < (n_items); it += (int)gridDim.x * 4)

// wave per sample item = 16 rows of St.  Lane (r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4) owns row r0 + 4q + r and
// the columns s, s + 4, s + 8, ...: at any time every group of 16 lanes (the unit in which the L1 looks up cache lines)
// samples a 4 x 4 block of the grid, i.e. a compact block of the image with few cache lines per gather.  The running coordinates of interpolate() (helpers.cpp:551-626) are a sequential fp32
// recurrence: a lane replays the row steps up to its row and then advances its own copy four column steps per sample -
// the same additions in the same order as the reference.
__global__ __launch_bounds__(256) void big_sample_kernel(const float *__restrict__ img_all, DescConst k, const BigLists *__restrict__ bl,
                                                         const BigRegion *__restrict__ regions, const int2 *__restrict__ sitems,
                                                         int max_items, const mods_region *__restrict__ reg_all,
                                                         float *__restrict__ pool, const int *__restrict__ err_flag) {
  if (*err_flag) return;
  const int n_items = min(bl->n_sitems, max_items);
  const int lane = threadIdx.x & 63;
  BIG_WAVE_LOOP(n_items, it) {
    const int2 item = sitems[it];
    const BigRegion br = regions[item.x];
    const RegionGeom g = region_geom(reg_all[(size_t)br.img * k.max_reg + br.ri], k.desc_mr, k.desc_ps, k.patch_rule);
    const float *img = img_all + (size_t)k.w * k.h * br.img;
    const int P2 = br.P2, P2r = br.P2r, half = P2 / 2, w = k.w, h = k.h;
    const int row = item.y + 4 * (lane >> 4) + (lane & 3), s = (lane >> 2) & 3;
    if (row >= P2) continue;
    const bool touch = check_borders(w, h, g.fx, g.fy, g.f11, g.f12, g.f21, g.f22, P2, P2);
    float rx = g.fx - (float)half * g.f12;
    float ry = g.fy - (float)half * g.f22;
    for (int q = 0; q < row; q++) { rx += g.f12; ry += g.f22; }
    float WX = rx - (float)half * g.f11;
    float WY = ry - (float)half * g.f21;
    for (int q = 0; q < s; q++) { WX += g.f11; WY += g.f21; }
    float *dst = pool + br.slab + big_hdr_floats(br.n_tap, k.desc_ps) + (size_t)s * P2r + row;
    for (int c = s; c < P2; c += 32, dst += (size_t)32 * P2r) {
      PixPair t0[8], t1[8];
      float wx[8], wy[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        ok[u] = false;
        if (c + 4 * u < P2) {
          int x, y;
          if (!touch) { x = (int)WX; y = (int)WY; ok[u] = true; }
          else { x = (int)floorf(WX); y = (int)floorf(WY); ok[u] = WX >= 0 && WY >= 0 && x < w - 1 && y < h - 1; }
          wx[u] = WX - (float)x;
          wy[u] = WY - (float)y;
          if (ok[u]) {
            const float *Row0 = img + (size_t)y * w + x;
            t0[u] = *(const PixPair *)Row0;
            t1[u] = *(const PixPair *)(Row0 + w);
          }
          WX += g.f11; WY += g.f21; WX += g.f11; WY += g.f21; WX += g.f11; WY += g.f21; WX += g.f11; WY += g.f21;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (c + 4 * u < P2) {
          float v = 0.f;
          if (ok[u]) {
            const float I1 = wx[u] * (t0[u].b - t0[u].a) + t0[u].a;
            v = wy[u] * (wx[u] * (t1[u].b - t1[u].a) + t1[u].a - I1) + I1;
          }
          dst[(size_t)4 * u * P2r] = v;
        }
    }
  }
}

// wave per row-pass item = 64 rows x `steps` groups of 4 column pairs.  Lane (yq = lane & 15, pg = lane >> 4) owns
// rows y0 + 4 yq .. + 3 and one column pair of the group:
//   T[y][q] = sum_j tap[j] * S[y][clamp(cidx[q] - r + j)], taps left to right.
// The two columns of a pair are x0 and x1 = x0 + 1 (their windows overlap in all but one sample, also after clamping)
// or, for a grid line clamped at the edge, x1 = x0 (both sums are the same).  Taps are wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void big_rowpass_kernel(DescConst k, const BigLists *__restrict__ bl, const BigRegion *__restrict__ regions,
                                                          const int2 *__restrict__ ritems, int max_items, float *__restrict__ pool,
                                                          const int *__restrict__ err_flag) {
  // the item's Gaussian taps, per wave: read through `pool` (which this kernel also writes) every tap would be a per-lane
  // global load with a full wait in front of its use (see big_fused_kernel); regions with more taps than fit keep that path
  constexpr int TAPS_LDS = 1024;
  __shared__ float s_wtap[4][TAPS_LDS];
  if (*err_flag) return;
  const int ps = k.desc_ps, ps2 = t_stride(ps);   // (row stride of T)
  const int n_items = min(bl->n_ritems, max_items);
  const int lane = threadIdx.x & 63;
  BIG_WAVE_LOOP(n_items, it) {
    const int2 item = ritems[it];
    const BigRegion br = regions[item.x];
    const int P2 = br.P2, P2r = br.P2r, n_tap = br.n_tap, r_tap = n_tap >> 1;
    const bool taps_in_lds = n_tap <= TAPS_LDS;
    wave_sync();
    if (taps_in_lds)
      for (int i = lane; i < n_tap; i += 64) s_wtap[threadIdx.x >> 6][i] = (pool + br.slab)[i];
    wave_sync();
    const int y = (item.y & 0xffff) + 4 * (lane & 15), pg = lane >> 4;
    const int step0 = item.y >> 16, step1 = min((ps + 3) / 4, step0 + big_rsteps(n_tap));
    const float *tap = taps_in_lds ? (const float *)s_wtap[threadIdx.x >> 6] : (const float *)(pool + br.slab);
    const int *cidx = (const int *)(pool + br.slab + n_tap + ps);
    const float *St = pool + br.slab + big_hdr_floats(n_tap, ps);
    float *T = (float *)St + big_s_floats(P2, P2r, n_tap);
    for (int step = step0; step < step1; step++) {
      const int pi = step * 4 + pg;
      const bool live = y < P2 && pi < ps;
      const int x0 = live ? cidx[2 * pi] : r_tap, x1 = live ? cidx[2 * pi + 1] : r_tap + 1;
      const bool interior = x0 - r_tap >= 0 && x0 + 1 + r_tap <= P2 - 1;
      const int yy = live ? y : 0;
      float4 s0, s1;
      if (__all(interior)) {
        const float *p = St + (size_t)(x0 - r_tap) * P2r + yy;
        float4 prev = *(const float4 *)p;
        float t = tap[0];
        s0 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
        prev = *(const float4 *)(p + P2r);
        s1 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
        int j = 1;
        for (; j + 7 < n_tap; j += 8) {
          float4 c[8];
#pragma unroll
          for (int u = 0; u < 8; u++) c[u] = *(const float4 *)(p + (size_t)(j + 1 + u) * P2r);
#pragma unroll
          for (int u = 0; u < 8; u++) {
            t = tap[j + u];
            s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
            s1.x = fmaf(t, c[u].x, s1.x); s1.y = fmaf(t, c[u].y, s1.y); s1.z = fmaf(t, c[u].z, s1.z); s1.w = fmaf(t, c[u].w, s1.w);
            prev = c[u];
          }
        }
        for (; j < n_tap; j++) {
          const float4 c = *(const float4 *)(p + (size_t)(j + 1) * P2r);
          t = tap[j];
          s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
          s1.x = fmaf(t, c.x, s1.x); s1.y = fmaf(t, c.y, s1.y); s1.z = fmaf(t, c.z, s1.z); s1.w = fmaf(t, c.w, s1.w);
          prev = c;
        }
      } else {
        // windows reaching over the edge: every sample index is clamped (a_j = S[clamp(x0 - r + j)]; the second
        // column uses a_{j+1})
        const float *p = St + yy;
        int xa = x0 - r_tap; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
        float4 prev = *(const float4 *)(p + (size_t)xa * P2r);
        float t = tap[0];
        s0 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
        xa = x0 - r_tap + 1; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
        prev = *(const float4 *)(p + (size_t)xa * P2r);
        s1 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
        for (int j = 1; j < n_tap; j++) {
          xa = x0 - r_tap + j + 1; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
          const float4 c = *(const float4 *)(p + (size_t)xa * P2r);
          t = tap[j];
          s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
          s1.x = fmaf(t, c.x, s1.x); s1.y = fmaf(t, c.y, s1.y); s1.z = fmaf(t, c.z, s1.z); s1.w = fmaf(t, c.w, s1.w);
          prev = c;
        }
      }
      if (live) {
        if (x1 == x0) s1 = s0;
        float *o = T + (size_t)y * ps2 + 2 * pi;
        *(float2 *)o = make_float2(s0.x, s1.x);
        if (y + 1 < P2) *(float2 *)(o + ps2) = make_float2(s0.y, s1.y);
        if (y + 2 < P2) *(float2 *)(o + 2 * ps2) = make_float2(s0.z, s1.z);
        if (y + 3 < P2) *(float2 *)(o + 3 * ps2) = make_float2(s0.w, s1.w);
      }
    }
  }
}

// block per fused item = R rows (64 for P2 <= 128, else 32) of a region with P2 <= BIG_FUSE_P2: the rows are sampled into
// LDS (transposed, St[col][R]) and the row pass reads them from there, so S never leaves the CU.
//   phase 1  thread (row = tid % R, phase = tid / R) walks the coordinate recurrence of its row and samples the columns
//            phase, phase + nph, ... (nph = 256 / R column steps between two of its samples)
//   phase 2  as big_rowpass_kernel, float4 = 4 rows from LDS; wave w takes the pair groups w, w + 4, ...
__global__ __launch_bounds__(256) void big_fused_kernel(const float *__restrict__ img_all, DescConst k, const BigLists *__restrict__ bl,
                                                        const BigRegion *__restrict__ regions, const int2 *__restrict__ fitems,
                                                        int max_items, const mods_region *__restrict__ reg_all,
                                                        float *__restrict__ pool, const int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) float s_St[];
  // the region's Gaussian taps, copied from its slab: read through `pool` (which this kernel also writes) they would be
  // per-lane global loads with a full wait in front of every use - one L2 round trip per tap in the clamped branch
  // (round 3: 36 of the 66 thousand cycles of an item).  P2 <= BIG_FUSE_P2 = 256 and patch sizes >= 8 give at most
  // (int)(9 * 254 / 8 + 1) | 1 = 287 taps (57 with the 41-pixel patch of the .ini).
  __shared__ float s_ftap[BIG_FUSE_TAPS];
  if (*err_flag) return;
  const int ps = k.desc_ps, ps2 = t_stride(ps);   // (row stride of T)
  const int n_items = min(bl->n_fitems, max_items);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef FUSED_PROF
  unsigned long long pt[3] = {0, 0, 0}, pl = __builtin_amdgcn_s_memtime();
  int pn = 0, prows = 0;
#define FPROF(i) { __syncthreads(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pl; pl = t_; }
#else
#define FPROF(i)
#endif
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int2 item = fitems[it];
    const BigRegion br = regions[item.x];
    const RegionGeom g = region_geom(reg_all[(size_t)br.img * k.max_reg + br.ri], k.desc_mr, ps, k.patch_rule);
    const float *img = img_all + (size_t)k.w * k.h * br.img;
    const int P2 = br.P2, w = k.w, h = k.h;
    const int R = big_fuse_rows(P2), r0 = item.y;
    __syncthreads();   // the previous item's row pass is done with the tile
    for (int i = tid; i < br.n_tap && i < BIG_FUSE_TAPS; i += 256) s_ftap[i] = (pool + br.slab)[i];
    FPROF(0)
    // phase 1: the item's rows, tile by tile (device_util.hpp: sample_tiles), stored transposed St[col][R]
    sample_tiles_rows<true>(img, w, h, g.fx, g.fy, g.f11, g.f12, g.f21, g.f22, P2, r0, min(P2, r0 + R), wv, 4,
                      [&](int row, int col, float v) { s_St[col * R + (row - r0)] = v; });
    __syncthreads();
    FPROF(1)
#ifdef FUSED_PROF
    pn++; prows += P2;
#endif
    {
      const int n_tap = br.n_tap, r_tap = n_tap >> 1;
      const int nq = R / 4, PS = 64 / nq;          // row quads, column pairs per step
      const int yq = lane % nq, pg = lane / nq;
      const int y = r0 + 4 * yq;
      const float *tap = s_ftap;
      const int *cidx = (const int *)(pool + br.slab + n_tap + ps);
      float *T = pool + br.slab + big_hdr_floats(n_tap, ps);
      const float *Sl = s_St + 4 * yq;
      const int steps = (ps + PS - 1) / PS;
      for (int step = wv; step < steps; step += 4) {
        const int pi = step * PS + pg;
        const bool live = y < P2 && pi < ps;
        const int x0 = live ? cidx[2 * pi] : r_tap, x1 = live ? cidx[2 * pi + 1] : r_tap + 1;
        const bool interior = x0 - r_tap >= 0 && x0 + 1 + r_tap <= P2 - 1;
        float4 s0, s1;
        if (__all(interior)) {
          const float *p = Sl + (x0 - r_tap) * R;
          float4 prev = *(const float4 *)p;
          float t = tap[0];
          s0 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
          prev = *(const float4 *)(p + R);
          s1 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
          int j = 1;
          for (; j + 7 < n_tap; j += 8) {
            float4 c[8];
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] = *(const float4 *)(p + (j + 1 + u) * R);
#pragma unroll
            for (int u = 0; u < 8; u++) {
              t = tap[j + u];
              s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
              s1.x = fmaf(t, c[u].x, s1.x); s1.y = fmaf(t, c[u].y, s1.y); s1.z = fmaf(t, c[u].z, s1.z); s1.w = fmaf(t, c[u].w, s1.w);
              prev = c[u];
            }
          }
          for (; j < n_tap; j++) {
            const float4 c = *(const float4 *)(p + (j + 1) * R);
            t = tap[j];
            s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
            s1.x = fmaf(t, c.x, s1.x); s1.y = fmaf(t, c.y, s1.y); s1.z = fmaf(t, c.z, s1.z); s1.w = fmaf(t, c.w, s1.w);
            prev = c;
          }
        } else {
          int xa = x0 - r_tap; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
          float4 prev = *(const float4 *)(Sl + xa * R);
          float t = tap[0];
          s0 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
          xa = x0 - r_tap + 1; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
          prev = *(const float4 *)(Sl + xa * R);
          s1 = make_float4(t * prev.x, t * prev.y, t * prev.z, t * prev.w);
          for (int j = 1; j < n_tap; j++) {
            xa = x0 - r_tap + j + 1; xa = xa < 0 ? 0 : (xa > P2 - 1 ? P2 - 1 : xa);
            const float4 c = *(const float4 *)(Sl + xa * R);
            t = tap[j];
            s0.x = fmaf(t, prev.x, s0.x); s0.y = fmaf(t, prev.y, s0.y); s0.z = fmaf(t, prev.z, s0.z); s0.w = fmaf(t, prev.w, s0.w);
            s1.x = fmaf(t, c.x, s1.x); s1.y = fmaf(t, c.y, s1.y); s1.z = fmaf(t, c.z, s1.z); s1.w = fmaf(t, c.w, s1.w);
            prev = c;
          }
        }
        if (live) {
          if (x1 == x0) s1 = s0;
          float *o = T + (size_t)y * ps2 + 2 * pi;
          *(float2 *)o = make_float2(s0.x, s1.x);
          if (y + 1 < P2) *(float2 *)(o + ps2) = make_float2(s0.y, s1.y);
          if (y + 2 < P2) *(float2 *)(o + 2 * ps2) = make_float2(s0.z, s1.z);
          if (y + 3 < P2) *(float2 *)(o + 3 * ps2) = make_float2(s0.w, s1.w);
        }
      }
    }
    FPROF(2)
  }
#ifdef FUSED_PROF
  if (tid == 0 && (blockIdx.x % 1024) == 9)
    printf("big_fused prof: block %d items %d mean P2 %d cycles per item: head %llu sample %llu rowpass %llu\n", blockIdx.x, pn, prows / max(pn, 1),
           pt[0] / max(pn, 1), pt[1] / max(pn, 1), pt[2] / max(pn, 1));
#endif
}

// wave per 64 pairs of adjacent output pixels of a region: column pass + resampling -> patch.  (Round 6 measured this kernel at 0.87 of
// the texture addresser and 27 % of the vector ALU - every strip value is fetched ~9 times through the vector cache - and built the
// obvious remedy, the strip of a region staged once in LDS by a workgroup per region: 0.607 against 0.550 ms per batch, three
// workgroups of 53 KB per CU load, then compute, and hide less than 24 independent waves do; left as it is.)
__global__ __launch_bounds__(256) void big_colres_kernel(DescConst k, const BigLists *__restrict__ bl, const BigRegion *__restrict__ regions,
                                                         int max_regions, const float *__restrict__ pool, float *__restrict__ patches,
                                                         const int *__restrict__ err_flag) {
  if (*err_flag) return;
  const int ps = k.desc_ps, pp = ps * ps, np = (ps + 1) >> 1, tasks = ps * np, chunks = (tasks + 63) / 64;
  const int n_items = min(bl->n_regions, max_regions) * chunks;
  const int lane = threadIdx.x & 63;
  BIG_WAVE_LOOP(n_items, it) {
    const int li = it / chunks, e = (it - li * chunks) * 64 + lane;
    const BigRegion br = regions[li];
    if (e >= tasks) continue;
    const float *tap = pool + br.slab;
    const float *seq = tap + br.n_tap;
    const int *cidx = (const int *)(seq + ps);
    const float *T = tap + big_hdr_floats(br.n_tap, ps) + big_s_floats(br.P2, br.P2r, br.n_tap);
    const float c0 = (float)(br.P2 >> 1);
    const bool touch2 = check_borders(br.P2, br.P2, c0, c0, br.scale, 0.f, 0.f, br.scale, ps, ps);
    const int j = e / np, m = e - j * np;
    float v0, v1;
    col_resample_pair(T, t_stride(ps), br.P2, ps, br.n_tap >> 1, touch2, tap, seq, cidx, j, m, &v0, &v1);
    float *o = patches + ((size_t)br.img * k.reg_cap + br.ri) * pp + j * ps + 2 * m;
    o[0] = v0;
    if (2 * m + 1 < ps) o[1] = v1;
  }
}

// ---------------------------------------------------------------------------------------
// photometric normalisation + SIFT on a patch held in LDS (block = 256)
// ---------------------------------------------------------------------------------------
// one lane: sequential fp32 sum of n contiguous LDS floats, loads batched eight at a time
__device__ __forceinline__ float ordered_sum(const float *v, int n) {
  float s = 0;
  int q = 0;
  if (n >= 8) {
    float4 a = *(const float4 *)(v), b = *(const float4 *)(v + 4);
    for (q = 8; q + 7 < n; q += 8) {   // the next 8 values are in flight while these 8 are added
      const float4 na = *(const float4 *)(v + q), nb = *(const float4 *)(v + q + 4);
      s += a.x; s += a.y; s += a.z; s += a.w; s += b.x; s += b.y; s += b.z; s += b.w;
      a = na; b = nb;
    }
    s += a.x; s += a.y; s += a.z; s += a.w; s += b.x; s += b.y; s += b.z; s += b.w;
  }
  for (; q < n; q++) s += v[q];
  return s;
}

// photometricallyNormalize, helpers.cpp:666-715.  s_midx: masked pixels in raster order; s_g: scratch
// (16-byte aligned, >= n_mask floats).
__device__ void photonorm_patch(float *s_patch, const unsigned short *s_midx, int n_mask, float *s_g, int pp, float *s_red) {
  const int tid = threadIdx.x;
  for (int q = tid; q < n_mask; q += 256) s_g[q] = s_patch[s_midx[q]];
  __syncthreads();
  if (tid == 0) s_red[0] = ordered_sum(s_g, n_mask) / (float)n_mask;   // gsum counts the masked pixels exactly
  __syncthreads();
  const float sum = s_red[0];
  for (int q = tid; q < n_mask; q += 256) { const float d = sum - s_g[q]; s_g[q] = d * d; }
  __syncthreads();
  if (tid == 0) s_red[1] = sqrtf(ordered_sum(s_g, n_mask) / (float)n_mask);
  __syncthreads();
  const float var = s_red[1];
  if (!((double)var < 0.0001)) {
    const float fac = 50.0f / var;
    for (int p = tid; p < pp; p += 256) {
      float v = 128 + fac * (s_patch[p] - sum);
      if (v > 255) v = 255;
      if (v < 0) v = 0;
      s_patch[p] = v;
    }
  }
  __syncthreads();
}

// computeRootSiftDescriptor / computeSiftDescriptor + norms.  s_px: {mask*grad, wo1} per pixel,
// s_bo: first orientation bin per pixel, s_wr / s_wc: per spatial bin effective row / column weights
// [4][ps] (built per workgroup by sift_tables), s_vec: 128 doubles, s_red: 2 doubles.
__device__ void sift_tables(const SiftTab *__restrict__ tab, int ps, float *s_wr) {
  // weight of pixel line i for spatial bin b: the reference visits (bin0, w0) and (bin1, w1); when
  // both name the same bin one of the weights is 0 (clamped), so the sum is the surviving weight.
  for (int e = threadIdx.x; e < 4 * ps; e += 256) {
    const int b8 = (e / ps) * 8, i = e - (e / ps) * ps;
    float w = 0.f;
    if (tab->bin0[i] == b8) w += tab->w0[i];
    if (tab->bin1[i] == b8) w += tab->w1[i];
    s_wr[e] = w;
  }
}

__device__ void sift_from_patch(float *s_patch, const float *__restrict__ mask, const float *s_w, const double *s_lut, int ps,
                                bool rootsift, double max_bin, float2 *s_px, unsigned char *s_bo, double *s_vec, double *s_red,
                                uint8_t *out, bool half = false) {
  double *s_sq = (double *)(((uintptr_t)s_patch + 7) & ~(uintptr_t)7);   // the patch is dead once the gradients are taken
  const int tid = threadIdx.x;
  const int pp = ps * ps;
  const double M_PI_DOUBLED = 6.28318530718;
#pragma unroll 1
  for (int p = tid; p < pp; p += 256) {
    const int r = p / ps, c = p - r * ps;
    float xgrad, ygrad;
    if (c == 0) xgrad = s_patch[p + 1] - s_patch[p];
    else if (c == ps - 1) xgrad = s_patch[p] - s_patch[p - 1];
    else xgrad = s_patch[p + 1] - s_patch[p - 1];
    if (r == 0) ygrad = s_patch[p + ps] - s_patch[p];
    else if (r == ps - 1) ygrad = s_patch[p] - s_patch[p - ps];
    else ygrad = s_patch[p + ps] - s_patch[p - ps];
    const float grad = sqrtf(xgrad * xgrad + ygrad * ygrad);
    const float ori = atan2_lut_ff_t(ygrad, xgrad, s_lut);
    const float o = (float)(8.0f * ((double)ori + M_PI_DOUBLED) / M_PI_DOUBLED);
    const int bo0 = (int)o;
    s_px[p] = make_float2(mask[p] * grad, o - bo0);
    s_bo[p] = (unsigned char)(bo0 % 8);
  }
  __syncthreads();
  // samplePatch (siftdesc.cpp:73-131): thread t < 128 owns vec[t], t = br*32 + bc*8 + bo, and visits
  // its block of pixels in raster order.  Per pixel exactly one (row weight, column weight) pair is
  // non-zero for this bin; pixels whose two orientation bins miss `bo` add +0.0 (no effect).
  // fl32(w (double) * v) of the reference = the fp32 product: the exact product of two floats fits a
  // double, so both round the same real number once.
  if (tid < 128) {
    const int br = tid >> 5, bc = (tid >> 3) & 3, bo = tid & 7;
    const float *wr = s_w + br * ps, *wc = s_w + bc * ps;
    int rlo = ps, rhi = 0, clo = ps;
    for (int i = 0; i < ps; i++) {
      if (wr[i] > 0) { rlo = min(rlo, i); rhi = i + 1; }
      if (wc[i] > 0) clo = min(clo, i);
    }
    // fixed 18-pixel window per row (spatial bins span <= 18 pixel columns for ps <= 45): entries past
    // the bin's columns carry weight 0 and add +0.0; their loads may run into the next row / the
    // following LDS words, whose value is irrelevant.  s_wpad: the bin's column weights, zero padded.
    constexpr int CW = 18;
    double acc = 0.0;
    for (int r = rlo; r < rhi; r++) {
      const float wrr = wr[r];
      const float2 *px = s_px + r * ps + clo;
      const unsigned char *bop = s_bo + r * ps + clo;
#pragma unroll 6
      for (int q = 0; q < CW; q++) {
        const float2 pv = px[q];
        const int bo0 = bop[q];
        const float wcq = (clo + q < ps) ? wc[clo + q] : 0.f;
        const float val = wrr * (wcq * pv.x);
        const bool m0 = bo0 == bo, m1 = ((bo0 + 1) & 7) == bo;
        const float wo = m0 ? (1.0f - pv.y) : pv.y;
        const float contrib = ((m0 || m1) && val > 0) ? val * wo : 0.0f;
        acc += (double)contrib;
      }
    }
    s_vec[tid] = acc;
  }
  __syncthreads();
  // doHalfSIFT (siftdesc.cpp:411-436): the raw histogram is folded, half[i*4 + j] = vec[i*8 + j] + vec[i*8 + j + 4], and the
  // 64 values take the place of the 128 below (the upper half is zero: it adds +0.0 to the ordered sums and quantises to 0)
  if (half) {
    double hv = 0.0;
    if (tid < 64) hv = s_vec[(tid >> 2) * 8 + (tid & 3)] + s_vec[(tid >> 2) * 8 + (tid & 3) + 4];
    __syncthreads();
    if (tid < 128) s_vec[tid] = tid < 64 ? hv : 0.0;
    __syncthreads();
  }
  // normalize (siftdesc.cpp:133-158) / clip / renormalise (:199-210, :248-257)
  for (int pass = 0; pass < 2; pass++) {
    // squares in parallel (into s_red[2..129] would alias nothing: use the idle patch buffer instead)
    if (tid < 128) ((double *)s_sq)[tid] = s_vec[tid] * s_vec[tid];
    __syncthreads();
    if (tid == 0) {
      const double *sq = (const double *)s_sq;
      double len = 0.0;
#pragma unroll 1
      for (int i = 0; i < 128; i += 16) {   // 16 loads in flight, then the reference's 4-term groups in order
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = sq[i + q];
#pragma unroll
        for (int q = 0; q < 16; q += 4) len += v[q] + v[q + 1] + v[q + 2] + v[q + 3];
      }
      len = sqrt(len);
      s_red[0] = 1.0 / len;
    }
    __syncthreads();
    bool changed = false;
    if (tid < 128) {
      double v = s_vec[tid] * s_red[0];
      if (pass == 0 && v > max_bin) { v = max_bin; changed = true; }
      s_vec[tid] = v;
    }
    const int any = __syncthreads_or(changed ? 1 : 0);
    if (!any) break;
  }
  if (rootsift) {
    if (tid == 0) {
      double sum = 0.;
#pragma unroll 1
      for (int i = 0; i < 128; i += 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = s_vec[i + q];
#pragma unroll
        for (int q = 0; q < 16; q++) sum += fabs(v[q]);
      }
      s_red[1] = sum;
    }
    __syncthreads();
    if (tid < 128) {
      const double v = sqrt(s_vec[tid] / s_red[1]);
      int bq = (int)(512.0 * v + 0.5);
      bq = bq < 255 ? bq : 255;
      bq = bq > 0 ? bq : 0;
      out[tid] = (uint8_t)bq;
    }
  } else if (tid < 128) {
    int bq = (int)(512.0 * s_vec[tid] + 0.5);
    bq = bq < 255 ? bq : 255;
    bq = bq > 0 ? bq : 0;
    out[tid] = (uint8_t)bq;
  }
  __syncthreads();
}

struct SiftLds {   // carve of the dynamic LDS of the SIFT kernels
  float *patch, *g, *w; float2 *px; unsigned char *bo; unsigned short *midx; double *vec, *red, *lut; float *redf;
  __device__ SiftLds(float *base, int ps) {
    const int pp = ps * ps, ppa = (pp + 3) & ~3;
    patch = base;
    px = (float2 *)(patch + ppa);
    g = (float *)px;                       // photometric normalisation scratch: dead before the gradients are written
    w = (float *)(px + ppa);
    vec = (double *)(((uintptr_t)(w + 4 * ps) + 7) & ~(uintptr_t)7);
    red = vec + 128;
    lut = red + 2;
    redf = (float *)(lut + 256);
    midx = (unsigned short *)(redf + 4);
    bo = (unsigned char *)(midx + ppa);
  }
};
static size_t sift_lds_bytes(int ps) {
  const size_t ppa = ((size_t)ps * ps + 3) & ~(size_t)3;
  return sizeof(float) * (ppa + 2 * ppa + 4 * ps + 4) + sizeof(double) * (130 + 256) + sizeof(unsigned short) * ppa + ppa + 32;
}

// grid = (N, n_img), block = 256: patches -> descriptors
__global__ __launch_bounds__(256, 5) void sift_kernel(DescConst k, const float *__restrict__ patches, mods_region *__restrict__ reg_all,
                                                   const int *__restrict__ reg_count, const float *__restrict__ mask,
                                                   const SiftTab *__restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.desc_ps, pp = ps * ps;
  SiftLds L(smem, ps);
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  sift_tables(tab, ps, L.w);
  for (int i = tid; i < 256; i += 256) L.lut[i] = g_atan_lut[i];
  if (tid < 64) {   // raster-ordered list of the masked pixels, one wave, ballot compaction
    int c = 0;
    for (int base = 0; base < pp; base += 64) {
      const int p = base + tid;
      const bool m = p < pp && mask[p] > 0;
      const unsigned long long bm = __ballot(m);
      if (m) L.midx[c + __popcll(bm & ((1ull << tid) - 1ull))] = (unsigned short)p;
      c += __popcll(bm);
    }
    if (tid == 0) ((int *)L.redf)[2] = c;
  }
  __syncthreads();
  const int n_mask = ((int *)L.redf)[2];
  mods_region *reg = reg_all + (size_t)b * k.max_reg;
  int n = reg_count[b];
  if (n > k.reg_cap) n = k.reg_cap;
  for (int ri = blockIdx.x; ri < n; ri += gridDim.x) {
    const float *src = patches + ((size_t)b * k.reg_cap + ri) * pp;
    __syncthreads();
    for (int p = tid; p < pp; p += 256) L.patch[p] = src[p];
    __syncthreads();
    if (k.photo) photonorm_patch(L.patch, L.midx, n_mask, L.g, pp, L.redf);
    sift_from_patch(L.patch, mask, L.w, L.lut, ps, k.root != 0, k.max_bin, L.px, L.bo, L.vec, L.red, reg[ri].desc, k.half_desc != 0);
  }
}

// ---------------------------------------------------------------------------------------
// SIFT, one WAVE per 8 regions.  The ordered sums (photometric mean / variance, the histogram bins, the norms) are
// sequential chains whose cost on a SIMD is the same for 1 or 64 active lanes, so the lanes of a wave carry the chains of
// 8 regions at once:
//   photometric normalisation  lanes 0..7 = the 8 regions, masked values staged through LDS in 64-value chunks
//   per pixel row r            64 lanes: gradient / orientation of row r -> (mask*grad, o); lane (region, which of the two row
//                              bins of r, column bin) walks its <= 18 columns in order and adds the two orientation
//                              contributions to the LDS accumulators with ds_add_f64 (LDS executes a wave's instructions in
//                              order, so every bin receives its terms in raster order; lanes of one instruction never share a bin)
//   norms                      lanes 0..7 run the 128-term sums, all lanes scale / clip / quantise
// Around the arithmetic (profiles/r03_sift_*; the round-2 form of this kernel is docs/history/r05_removed_paths.patch):
//   * persistent waves: one workgroup of 12 waves per CU pulls work units (8 regions of one image) from a device counter, so
//     the tables are set up once per workgroup and the images of a batch balance; no workgroup barrier after the set-up;
//   * the histogram bins live at  bo * 128 + (g >> 2) * 64 + br * 16 + (g & 3) * 4 + bc  (doubles): the 32 lanes of a
//     half-wave - 4 regions x 2 row bins x 4 column bins - always hit 32 different bank pairs whatever the orientation
//     bins are, so ds_add_f64 runs at its conflict-free rate (10 instead of 23 cycles per wave instruction, tools/ubench);
//   * the three pixel rows a gradient needs stay in registers (vertical neighbours = the lane's own values of the previous /
//     next row, horizontal neighbours = one DPP move from the adjacent lane), the patch rows are fetched three rows ahead;
//     photometric mean / factor sit in registers per pixel slot;
//   * 11 KB of LDS per wave and <= 168 VGPRs: 12 waves per CU.
// LDS: o table 8 KB | row / column weights | work prefix | masked-pixel list || per wave: bins 8 KB | pixel row (x, o) + pad
// (the photometric chunk aliases it) | mean, factor, flag.
// ---------------------------------------------------------------------------------------
constexpr int SW_R = 8;          // regions per wave
constexpr int SW_G = 68;         // photometric chunk stride in floats
constexpr int SW_CW = 18;        // pixel columns a spatial bin spans at most (patch sizes <= 45)
constexpr int S2_MAX_IMG = 64;   // images of one launch (work prefix table in LDS)
constexpr int S2_WAVES = 12;               // one workgroup per CU: 3 waves per SIMD (<= 168 VGPRs), 145 KB of LDS
constexpr int S2_BINS = 1024;                 // doubles per wave
__device__ __forceinline__ int s2_bin(int g, int br, int bc) { return (g >> 2) * 64 + br * 16 + (g & 3) * 4 + bc; }   // + bo * 128
__device__ __forceinline__ int s2_vec(int g, int i) { return (i & 7) * 128 + s2_bin(g, i >> 5, (i >> 3) & 3); }          // vec[i], i = (br*4 + bc)*8 + bo

static size_t sift_wave2_wave_bytes(int ps) {
  return sizeof(double) * S2_BINS + sizeof(float2) * ((size_t)SW_R * ps + SW_CW + 2) + sizeof(float) * 3 * SW_R;
}
static size_t sift_wave2_lds_bytes(int ps) {
  const size_t ppa = ((size_t)ps * ps + 7) & ~(size_t)7;
  return sizeof(float) * 2048 + sizeof(float) * (4 * ps + 4) + sizeof(int) * 72 + sizeof(unsigned short) * ppa + S2_WAVES * sift_wave2_wave_bytes(ps) + 64;
}

// NC: pixel columns a histogram lane walks per row = the widest span of positive column weights of a spatial bin, rounded up to
// an even number (15 -> 16 for the 41-pixel patch; SW_CW = 18 covers every patch size this kernel takes)
template <int NC>
__global__ __launch_bounds__(64 * S2_WAVES, 3) void sift_wave2_kernel(DescConst k, const float *__restrict__ patches, mods_region *__restrict__ reg_all,
                                                                        const int *__restrict__ reg_count, const float *__restrict__ mask,
                                                                        const SiftTab *__restrict__ tab, int n_img, int *__restrict__ next_unit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.desc_ps, pp = ps * ps, ppa = (pp + 7) & ~7;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float *s_ot = smem;                               // [8][256]: o of computeSiftDescriptor per atan2LUTff (case, table index)
  float *s_w = s_ot + 2048;                         // [4][ps] spatial weights
  int *s_pref = (int *)(s_w + 4 * ps + 4);          // [n_img + 1] work units before image b; [68]: n_mask
  unsigned short *s_midx = (unsigned short *)(s_pref + 72);
  unsigned char *wbase = (unsigned char *)(s_midx + ppa) + (size_t)wv * (sizeof(double) * S2_BINS + sizeof(float2) * (SW_R * ps + SW_CW + 2) + sizeof(float) * 3 * SW_R);
  double *acc = (double *)wbase;
  float2 *pxrow = (float2 *)(acc + S2_BINS);        // [8 * ps + pad]: (mask * |grad|, o) of the current pixel row
  float *s_mean = (float *)(pxrow + SW_R * ps + SW_CW + 2), *s_fac = s_mean + SW_R;
  int *s_flag = (int *)(s_fac + SW_R);
  float *g = (float *)pxrow;                        // photometric chunk [8][SW_G], dead before the first pixel row is written
  const int rowf = SW_R * ps;

  const double M_PI_DOUBLED = 6.28318530718;
  sift_tables(tab, ps, s_w);
  for (int i = tid; i < 2048; i += 64 * S2_WAVES) {   // o = (float)(8.0f * ((double)ori + 2 pi) / 2 pi), siftdesc.cpp:181, for every value ori can take
    const float ori = atan2_lut_case(i >> 8, g_atan_lut[i & 255]);
    s_ot[i] = (float)(8.0f * ((double)ori + M_PI_DOUBLED) / M_PI_DOUBLED);
  }
  if (tid < 64) {   // raster-ordered list of the masked pixels, one wave, ballot compaction
    int c = 0;
    for (int base = 0; base < pp; base += 64) {
      const int p = base + tid;
      const bool m = p < pp && mask[p] > 0;
      const unsigned long long bm = __ballot(m);
      if (m) s_midx[c + __popcll(bm & ((1ull << tid) - 1ull))] = (unsigned short)p;
      c += __popcll(bm);
    }
    if (tid == 0) s_pref[68] = c;
  }
  if (tid == 64) {  // work units (groups of 8 regions) before every image
    int cum = 0;
    for (int b = 0; b < n_img; b++) { s_pref[b] = cum; int n = reg_count[b]; if (n > k.reg_cap) n = k.reg_cap; cum += (n + SW_R - 1) / SW_R; }
    s_pref[n_img] = cum;
  }
  __syncthreads();
  const int n_mask = s_pref[68];
  const int total_units = s_pref[n_img];
  const float o_zero = (float)(8.0f * ((double)0.f + M_PI_DOUBLED) / M_PI_DOUBLED);

  // histogram lane: region hr, row-bin selector hsel, column bin hbc; its column window and weights
  const int hr = lane >> 3, hsel = (lane >> 2) & 1, hbc = lane & 3;
  int clo = ps;
  for (int i = ps - 1; i >= 0; i--) if (s_w[hbc * ps + i] > 0) clo = i;
  float wcw[NC];
#pragma unroll
  for (int q = 0; q < NC; q++) wcw[q] = (clo + q < ps) ? s_w[hbc * ps + clo + q] : 0.f;
  // pixel slots of a lane: e = lane + 64 u -> region e / ps, column e % ps (the same for every row and every unit)
  constexpr int RL = 6;                               // >= ceil(SW_R * ps / 64) for ps <= 48 (this kernel runs for ps <= 45)
  int s_rr[RL], s_c[RL];
#pragma unroll
  for (int u = 0; u < RL; u++) { const int e = lane + 64 * u; s_rr[u] = e < rowf ? e / ps : SW_R - 1; s_c[u] = e < rowf ? e - (e / ps) * ps : 0; }

#ifdef SIFT_PROF
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pl = __builtin_amdgcn_s_memtime();
  int pn = 0;
#define S2PROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pl; pl = t_; }
#else
#define S2PROF(i)
#endif
  for (;;) {
    int unit = 0;
    if (lane == 0) unit = atomicAdd(next_unit, 1);
    unit = __builtin_amdgcn_readfirstlane(unit);
    if (unit >= total_units) break;
    int b = 0;
    while (b + 1 < n_img && s_pref[b + 1] <= unit) b++;
    const int ri0 = (unit - s_pref[b]) * SW_R;
    mods_region *reg = reg_all + (size_t)b * k.max_reg;
    int n = reg_count[b];
    if (n > k.reg_cap) n = k.reg_cap;
    const float *pbase = patches + (size_t)b * k.reg_cap * pp;
    // patch of region slot rr (slots past the end repeat the last region; their results are not stored)
    auto patch_of = [&](int rr) { return pbase + (size_t)min(ri0 + rr, n - 1) * pp; };
    wave_sync();
    S2PROF(0)
#pragma unroll
    for (int i = 0; i < S2_BINS / 64; i++) acc[lane + 64 * i] = 0.0;
    if (lane < SW_R) { s_flag[lane] = 0; s_mean[lane] = 0.f; s_fac[lane] = 0.f; }
    // ---- photometricallyNormalize, helpers.cpp:666-715: mean, then deviation, both as sequential sums over the masked pixels
    if (k.photo) {
      for (int pass = 0; pass < 2; pass++) {
        float sum = 0.f;
        // the values of chunk q0 + 64 are in flight while lanes 0..7 add chunk q0
        float v[SW_R];
        {
          const int idx = s_midx[min(lane, n_mask - 1)];
#pragma unroll
          for (int rr = 0; rr < SW_R; rr++) v[rr] = patch_of(rr)[idx];
        }
        for (int q0 = 0; q0 < n_mask; q0 += 64) {
          const int cnt = min(64, n_mask - q0);
          wave_sync();
#pragma unroll
          for (int rr = 0; rr < SW_R; rr++) {
            float x = v[rr];
            if (pass) { const float d = s_mean[rr] - x; x = d * d; }
            g[rr * SW_G + lane] = x;
          }
          if (q0 + 64 < n_mask) {
            const int idx = s_midx[min(q0 + 64 + lane, n_mask - 1)];
#pragma unroll
            for (int rr = 0; rr < SW_R; rr++) v[rr] = patch_of(rr)[idx];
          }
          wave_sync();
          if (lane < SW_R) {
            const float *gl = g + lane * SW_G;
            int q = 0;
            for (; q + 15 < cnt; q += 16) {     // 4 reads in flight, then their 16 terms in order
              const float4 a0 = *(const float4 *)(gl + q), a1 = *(const float4 *)(gl + q + 4), a2 = *(const float4 *)(gl + q + 8), a3 = *(const float4 *)(gl + q + 12);
              sum += a0.x; sum += a0.y; sum += a0.z; sum += a0.w; sum += a1.x; sum += a1.y; sum += a1.z; sum += a1.w;
              sum += a2.x; sum += a2.y; sum += a2.z; sum += a2.w; sum += a3.x; sum += a3.y; sum += a3.z; sum += a3.w;
            }
            for (; q + 3 < cnt; q += 4) {
              const float4 a = *(const float4 *)(gl + q);
              sum += a.x; sum += a.y; sum += a.z; sum += a.w;
            }
            for (; q < cnt; q++) sum += gl[q];
          }
        }
        if (lane < SW_R) {
          if (pass == 0) s_mean[lane] = sum / (float)n_mask;
          else {
            const float var = sqrtf(sum / (float)n_mask);
            if (!((double)var < 0.0001)) { s_flag[lane] = 1; s_fac[lane] = 50.0f / var; }
          }
        }
        wave_sync();
      }
    }
    wave_sync();
    S2PROF(1)
    // per pixel slot: byte offset of its column in the patch store (32 bits from the image's first patch: scalar base + lane
    // offset loads; slots past the 8 * ps pixels of a row read the last pixel again and store nothing), and the normalisation
    // of its region (flag off: factor 1, mean 0 never used - the raw value is selected)
    unsigned sbyte[RL], cbyte[RL];
    float sm[RL], sf[RL];
    unsigned nflag = 0;
#pragma unroll
    for (int u = 0; u < RL; u++) {
      sbyte[u] = ((unsigned)min(ri0 + s_rr[u], n - 1) * (unsigned)pp + (unsigned)s_c[u]) * 4u;
      cbyte[u] = (unsigned)s_c[u] * 4u;
      sm[u] = s_mean[s_rr[u]]; sf[u] = s_fac[s_rr[u]];
      nflag |= (s_flag[s_rr[u]] ? 1u : 0u) << u;
    }
    auto fetch_row = [&](int r, float *dst) {
      const char *rowp = (const char *)(pbase + r * ps);           // uniform: scalar base + 32-bit lane offset
#pragma unroll
      for (int u = 0; u < RL; u++) dst[u] = *(const float *)(rowp + sbyte[u]);
    };
    auto norm_row = [&](const float *raw, float *dst) {
#pragma unroll
      for (int u = 0; u < RL; u++) {
        // 128 + fac * (v - mean) clamped to [0, 255] (helpers.cpp:705-712; the value is finite, so the median of (v, 0, 255)
        // is the two comparisons of the reference)
        const float t = __builtin_amdgcn_fmed3f(128 + sf[u] * (raw[u] - sm[u]), 0.f, 255.f);
        dst[u] = ((nflag >> u) & 1) ? t : raw[u];
      }
    };
    auto fetch_mask = [&](int r, float *dst) {
      const char *mrowp = (const char *)(mask + r * ps);
#pragma unroll
      for (int u = 0; u < RL; u++) dst[u] = *(const float *)(mrowp + cbyte[u]);
    };
    // rows r - 1, r, r + 1 normalised (nm, n0, np); row r + 2 as fetched (ra), row r + 3 in flight (rb)
    float nm[RL], n0[RL], np[RL], ra[RL], rb[RL], mcur[RL];
    fetch_row(0, ra); norm_row(ra, n0);
    fetch_row(min(1, ps - 1), ra); norm_row(ra, np);
    fetch_row(min(2, ps - 1), ra);
    fetch_row(min(3, ps - 1), rb);
    fetch_mask(0, mcur);
#pragma unroll
    for (int u = 0; u < RL; u++) nm[u] = n0[u];
    // ---- computeSiftDescriptor / samplePatch (siftdesc.cpp:73-131, 160-198), one pixel row at a time
#pragma unroll 1
    for (int r = 0; r < ps; r++) {
      {
        // branch-free: the one-sided differences at the patch border are the same subtraction with one operand at the pixel itself.
        // Vertically that needs nothing: the rows behind the last one are fetched clamped (np = the row itself at r = ps - 1) and
        // nm starts as row 0.  Horizontally the column of a slot decides (cbyte = 4 * column: a compare + a select per side).
        // The square root is the compiler's correctly rounded expansion without its rescue of operands below 2^-96 (fast_sqrtf_any,
        // device_util.hpp: the full expansion when a lane holds such an operand - never on image data).
        const unsigned last_col = 4u * (unsigned)(ps - 1);
#pragma unroll
        for (int u = 0; u < RL; u++) {
          const int e = lane + 64 * u;
          // horizontal neighbours: the adjacent lanes of this slot; across the slot boundary lane 63 <-> lane 0 of the next slot
          float right = lane_down1(n0[u]), left = lane_up1(n0[u]);
          if (u + 1 < RL) { const float nx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(n0[u + 1 < RL ? u + 1 : u]), 0)); right = lane == 63 ? nx : right; }
          if (u > 0) { const float pl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(n0[u > 0 ? u - 1 : u]), 63)); left = lane == 0 ? pl : left; }
          const float xa = cbyte[u] < last_col ? right : n0[u];
          const float xb = cbyte[u] > 0u ? left : n0[u];
          const float xgrad = xa - xb, ygrad = np[u] - nm[u];
          const float grad = fast_sqrtf_any(xgrad * xgrad + ygrad * ygrad);
          const AtanSel as = atan2_lut_sel(ygrad, xgrad);
          const float ot = s_ot[as.oct * 256 + as.idx];
          const float o = as.zero ? o_zero : ot;
          if (e < rowf) pxrow[e] = make_float2(mcur[u] * grad, o);
        }
      }
      fetch_mask(min(r + 1, ps - 1), mcur);     // in flight during the histogram pass
      wave_sync();
      S2PROF(2)
      {
        // the row's two spatial row bins (bin0 / bin1 and their weights; already multiplied by 8 = orientation bins)
        const int rb8 = hsel ? tab->bin1[r] : tab->bin0[r];
        const float wrr = hsel ? tab->w1[r] : tab->w0[r];
        if (wrr > 0) {
          double *abin = acc + s2_bin(hr, rb8 >> 3, hbc);
          const float2 *px = pxrow + hr * ps + clo;
#pragma unroll
          for (int q0 = 0; q0 < NC; q0 += NC / 2) {
            float2 pv[NC / 2];
#pragma unroll
            for (int q = 0; q < NC / 2; q++) pv[q] = px[q0 + q];
#pragma unroll
            for (int q = 0; q < NC / 2; q++) {
              // a pixel that the reference skips (val <= 0; also the NaN that a zero weight makes of the idle words behind
              // the row) adds +0.0, which leaves a bin as it is: no branch per pixel
              const float val = wrr * (wcw[q0 + q] * pv[q].x);
              const bool on = val > 0;
              const int bo0 = (int)pv[q].y;
              const float y = pv[q].y - bo0;
              const float c0 = on ? val * (1.0f - y) : 0.f, c1 = on ? val * y : 0.f;
              __hip_atomic_fetch_add(abin + (bo0 & 7) * 128, (double)c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              __hip_atomic_fetch_add(abin + ((bo0 + 1) & 7) * 128, (double)c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
        }
      }
      wave_sync();
      S2PROF(3)
      // next row: r + 1 becomes the centre, the row fetched two iterations ago is normalised, row r + 4 starts its way
#pragma unroll
      for (int u = 0; u < RL; u++) { nm[u] = n0[u]; n0[u] = np[u]; }
      norm_row(ra, np);
#pragma unroll
      for (int u = 0; u < RL; u++) ra[u] = rb[u];
      fetch_row(min(r + 4, ps - 1), rb);
      S2PROF(4)
    }
    wave_sync();
    // ---- normalize / clip / renormalise (siftdesc.cpp:133-158, 199-210, 248-257) and the RootSIFT mapping
    double *tmp = (double *)pxrow;      // 8 doubles of hand-over; the pixel row is idle here
    unsigned long long redo = ~0ull;   // bit 8 rr.. : region rr still takes part
    for (int pass = 0; pass < 2; pass++) {
      if (lane < SW_R && ((redo >> (8 * lane)) & 1)) {
        double len = 0.0;
#pragma unroll 1
        for (int i = 0; i < 128; i += 8) {
          double x[8];
#pragma unroll
          for (int q = 0; q < 8; q++) x[q] = acc[s2_vec(lane, i + q)];
#pragma unroll
          for (int q = 0; q < 8; q++) x[q] = x[q] * x[q];
          len += x[0] + x[1] + x[2] + x[3];
          len += x[4] + x[5] + x[6] + x[7];
        }
        len = sqrt(len);
        tmp[lane] = 1.0 / len;
      }
      wave_sync();
      bool changed = false;
      {
        const int rr = lane >> 3;
        if ((redo >> (8 * rr)) & 1) {
          const double inv = tmp[rr];
          for (int i = lane & 7; i < 128; i += 8) {
            double *vp = acc + s2_vec(rr, i);
            double x = *vp * inv;
            if (pass == 0 && x > k.max_bin) { x = k.max_bin; changed = true; }
            *vp = x;
          }
        }
      }
      const unsigned long long ch = __ballot(changed);
      unsigned long long next = 0;
      for (int rr = 0; rr < SW_R; rr++)
        if ((ch >> (8 * rr)) & 0xffull) next |= 0xffull << (8 * rr);
      redo = next;
      wave_sync();
      if (!redo) break;
    }
    if (k.root) {
      if (lane < SW_R) {
        double sum = 0.;
#pragma unroll 1
        for (int i = 0; i < 128; i += 8) {
          double x[8];
#pragma unroll
          for (int q = 0; q < 8; q++) x[q] = acc[s2_vec(lane, i + q)];
#pragma unroll
          for (int q = 0; q < 8; q++) sum += fabs(x[q]);
        }
        tmp[lane] = sum;
      }
      wave_sync();
    }
    {
      const int rr = lane >> 3;
      if (ri0 + rr < n) {
        const double rs = k.root ? tmp[rr] : 1.0;
        uint32_t *out = (uint32_t *)reg[ri0 + rr].desc;
        for (int w4 = lane & 7; w4 < 32; w4 += 8) {
          uint32_t word = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double x = acc[s2_vec(rr, 4 * w4 + q)];
            const double y = k.root ? sqrt(x / rs) : x;
            int bq = (int)(512.0 * y + 0.5);
            bq = bq < 255 ? bq : 255;
            bq = bq > 0 ? bq : 0;
            word |= (uint32_t)bq << (8 * q);
          }
          out[w4] = word;
        }
      }
    }
    S2PROF(5)
#ifdef SIFT_PROF
    pn++;
#endif
  }
#ifdef SIFT_PROF
  if (lane == 0 && (blockIdx.x % 64) == 3 && wv == 1)
    printf("sift_wave2 prof: block %d units %d cycles per unit: other %llu photo %llu grad %llu hist %llu shift+norm %llu norms %llu\n", blockIdx.x, pn,
           pt[0] / max(pn, 1), pt[1] / max(pn, 1), pt[2] / max(pn, 1), pt[3] / max(pn, 1), pt[4] / max(pn, 1), pt[5] / max(pn, 1));
#endif
}

__global__ __launch_bounds__(256) void sift_patch_test_kernel(const float *__restrict__ patch, int ps, int root, double max_bin,
                                                              const float *__restrict__ mask, const SiftTab *__restrict__ tab,
                                                              uint8_t *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  SiftLds L(smem, ps);
  sift_tables(tab, ps, L.w);
  for (int i = threadIdx.x; i < 256; i += 256) L.lut[i] = g_atan_lut[i];
  for (int p = threadIdx.x; p < ps * ps; p += 256) L.patch[p] = patch[p];
  __syncthreads();
  sift_from_patch(L.patch, mask, L.w, L.lut, ps, root != 0, max_bin, L.px, L.bo, L.vec, L.red, out);
}

// ---------------------------------------------------------------------------------------
int launch_extract_and_sift(mods_ctx *ctx, const float *img_dev, int n_img, DescConst k, const float *dmask, const SiftTab *tab, bool run_sift) {
  StageScope ts(ctx, MODS_STAGE_DESCRIBE);
  const int ps = k.desc_ps, ps2 = 2 * ps, pp = ps * ps;
  // HBM layout of the description scratch: patch store [n_img][reg_cap][ps*ps] | big-tier bookkeeping | slab pool
  const size_t patch_elems = (size_t)n_img * k.reg_cap * pp;
  const int max_big = 1 << 17, max_items = 1 << 20;
  const int small_cap_items = n_img * k.reg_cap;
  const size_t book_elems = (sizeof(BigLists) + sizeof(BigRegion) * max_big + 3 * sizeof(int2) * max_items + 3 * sizeof(int) * (size_t)small_cap_items + 15) / 4;
  // slab pool: ~25 M floats per 1080p image in practice (row-pass strips, S only above 256 px); 64 M per image of the batch
  // and per 2 Mpx of image area, at least 1 GiB
  const unsigned long long area_units = std::max<unsigned long long>(1, ((unsigned long long)k.w * k.h + (1ull << 21) - 1) >> 21);
  const unsigned long long pool_elems = std::max<unsigned long long>(256ull << 20, (unsigned long long)n_img * area_units * (64ull << 20));
  const size_t need = patch_elems + book_elems + pool_elems;
  if (need > ctx->desc_scratch_elems) {
    MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
    if (ctx->desc_scratch) MODS_HIP_CHECK(hipFree(ctx->desc_scratch));
    ctx->desc_scratch = nullptr;
    mods::dev_pool_reallocated(ctx); MODS_HIP_CHECK(hipMalloc(&ctx->desc_scratch, need * sizeof(float)));
    ctx->desc_scratch_elems = need;
  }
  float *patches = ctx->desc_scratch;
  BigLists *bl = (BigLists *)(ctx->desc_scratch + patch_elems);
  BigRegion *bregs = (BigRegion *)(bl + 1);
  int2 *sitems = (int2 *)(bregs + max_big);
  int2 *ritems = sitems + max_items;
  int2 *fitems = ritems + max_items;
  int *small_items = (int *)(fitems + max_items);
  float *pool = ctx->desc_scratch + patch_elems + book_elems;
  MODS_HIP_CHECK(hipMemsetAsync(bl, 0, sizeof(BigLists), ctx->stream));
  { const int trc = launch_blur_table(ctx, ps); if (trc) return trc; }
  {
  StageScope ts_extract(ctx, MODS_STAGE_EXTRACT);
  k.tap_cap = 4096;
  const int small_cap = SMALL_CAP;   // P2 limit of the LDS tier
  k.p2_hi = small_cap;
  hipLaunchKernelGGL(big_classify_kernel, dim3(CLASSIFY_BLOCKS, n_img), dim3(1024), 0, ctx->stream, k, ctx->regions_dev,
                     ctx->region_count, bl, bregs, sitems, ritems, fitems, small_items, small_cap_items, std::min(small_cap, EXTRACT_T_LO), std::min(small_cap, EXTRACT_T_MID), max_big, max_items,
                     pool_elems, ctx->desc_err_dev);
  // LDS tier in three launches over the work lists of big_classify_kernel: a workgroup's LDS follows the largest window of its class
  // (20 / 34 / 53 KB: 8 / 4 / 3 workgroups per CU by LDS), and the 48 < P2 <= 80 regions of a 1080p image are as
  // much sampling work as the 6 800 smaller ones; the HBM tier takes P2 > small_cap
  {
    const int tiers[4] = {-1, std::min(small_cap, EXTRACT_T_LO), std::min(small_cap, EXTRACT_T_MID), small_cap};
    for (int t = 0; t < 3; t++) {
      if (tiers[t + 1] <= tiers[t]) continue;
      DescConst kt = k;
      kt.p2_lo = tiers[t]; kt.p2_hi = tiers[t + 1];
      const size_t capS = kt.p2_hi > 4 ? kt.p2_hi : 4;
      const size_t ldsS = sizeof(float) * (capS * odd4((int)capS) + capS * ((capS + 3) & ~(size_t)3) + 2 * ps2 + 32) + 32;
      hipLaunchKernelGGL(extract_small_kernel, dim3(4096), dim3(256), ldsS, ctx->stream, img_dev, kt, ctx->regions_dev,
                         small_items + (size_t)t * small_cap_items, &bl->n_small[t], small_cap_items, patches,
                         (const float *)ctx->blur_table_dev);
    }
  }
  const size_t ldsH = sizeof(float) * (k.tap_cap + 4 * ps2) + 32;
  hipLaunchKernelGGL(big_setup_kernel, dim3(1024), dim3(256), ldsH, ctx->stream, k, bl, bregs, max_big, pool, ctx->desc_err_dev);
  hipLaunchKernelGGL(big_fused_kernel, dim3(8192), dim3(256), BIG_FUSE_KB * 1024, ctx->stream, img_dev, k, bl, bregs, fitems, max_items,
                     ctx->regions_dev, pool, ctx->desc_err_dev);
  hipLaunchKernelGGL(big_sample_kernel, dim3(4096), dim3(256), 0, ctx->stream, img_dev, k, bl, bregs, sitems, max_items,
                     ctx->regions_dev, pool, ctx->desc_err_dev);
  hipLaunchKernelGGL(big_rowpass_kernel, dim3(4096), dim3(256), 0, ctx->stream, k, bl, bregs, ritems, max_items, pool,
                     ctx->desc_err_dev);
  hipLaunchKernelGGL(big_colres_kernel, dim3(4096), dim3(256), 0, ctx->stream, k, bl, bregs, max_big, pool, patches, ctx->desc_err_dev);
  }
  StageScope ts_sift(ctx, MODS_STAGE_SIFT);
  if (run_sift && ps > 45)   // the block-per-region form: patch sizes the wave form's register rows do not cover
    hipLaunchKernelGGL(sift_kernel, dim3(2048, n_img), dim3(256), sift_lds_bytes(ps), ctx->stream, k, patches, ctx->regions_dev,
                       ctx->region_count, dmask, tab);
  else if (run_sift) {
    static DynLdsOnce once16, once18;
    MODS_HIP_CHECK(dyn_lds_once(once16, (const void *)sift_wave2_kernel<16>, 160 * 1024, ctx->device));
    MODS_HIP_CHECK(dyn_lds_once(once18, (const void *)sift_wave2_kernel<SW_CW>, 160 * 1024, ctx->device));
    // widest span of positive column weights of a spatial bin (siftdesc.cpp:22-71: step = 5 / (2 * (ps / 2)), bins x - 1 and x)
    int span = 0;
    {
      const float step = 5.0f / (float)(2 * (ps >> 1));
      for (int bc = 0; bc < 4; bc++) {
        int lo = ps, hi = -1;
        for (int i = 0; i < ps; i++) {
          const float x = step * i; const int xi = (int)x; const float w1 = x - xi, w0 = 1.0f - w1;
          const bool on = (xi - 1 == bc && w0 > 0) || (xi == bc && w1 > 0);
          if (on) { lo = std::min(lo, i); hi = i; }
        }
        if (hi >= lo) span = std::max(span, hi - lo + 1);
      }
    }
    // persistent: one workgroup of 12 waves per CU, work units pulled from bl->sift_next (zeroed with bl above; batches of
    // more than S2_MAX_IMG images take one launch per S2_MAX_IMG images)
    for (int b0 = 0; b0 < n_img; b0 += S2_MAX_IMG) {
      const int nb = std::min(S2_MAX_IMG, n_img - b0);
      if (b0) MODS_HIP_CHECK(hipMemsetAsync(&bl->sift_next, 0, sizeof(int), ctx->stream));
      const float *pch = patches + (size_t)b0 * k.reg_cap * pp;
      mods_region *regs = ctx->regions_dev + (size_t)b0 * k.max_reg;
      if (span <= 16)
        hipLaunchKernelGGL(sift_wave2_kernel<16>, dim3(ctx->n_cu), dim3(64 * S2_WAVES), sift_wave2_lds_bytes(ps), ctx->stream, k, pch,
                           regs, ctx->region_count + b0, dmask, tab, nb, &bl->sift_next);
      else
        hipLaunchKernelGGL(sift_wave2_kernel<SW_CW>, dim3(ctx->n_cu), dim3(64 * S2_WAVES), sift_wave2_lds_bytes(ps), ctx->stream, k, pch,
                           regs, ctx->region_count + b0, dmask, tab, nb, &bl->sift_next);
    }
  }
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

int launch_half_sift(mods_ctx *ctx, int n_img, DescConst k, const float *dmask, const SiftTab *tab) {
  StageScope ts(ctx, MODS_STAGE_DESCRIBE);
  if (!ctx->regions_half_dev) MODS_HIP_CHECK(hipMalloc(&ctx->regions_half_dev, sizeof(mods_region) * (size_t)ctx->max_cand * ctx->batch));
  MODS_HIP_CHECK(hipMemcpyAsync(ctx->regions_half_dev, ctx->regions_dev, sizeof(mods_region) * (size_t)ctx->max_cand * n_img,
                                hipMemcpyDeviceToDevice, ctx->stream));
  k.half_desc = 1;
  const int ps = k.desc_ps;
  const float *patches = ctx->desc_scratch;      // the patch store of launch_extract_and_sift: [n_img][reg_cap][ps*ps]
  hipLaunchKernelGGL(sift_kernel, dim3(2048, n_img), dim3(256), sift_lds_bytes(ps), ctx->stream, k, patches, ctx->regions_half_dev,
                     ctx->region_count, dmask, tab);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

int launch_sift_patch_test(mods_ctx *ctx, const float *patch_dev, int ps, int root, double max_bin, uint8_t *out_dev) {
  hipLaunchKernelGGL(sift_patch_test_kernel, dim3(1), dim3(256), sift_lds_bytes(ps), ctx->stream, patch_dev, ps, root, max_bin,
                     ctx->desc_tables_dev + kTabDescMask, (const SiftTab *)(ctx->desc_tables_dev + kTabSift), out_dev);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods
