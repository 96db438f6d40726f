// Dominant orientation, measurement-region extraction and SIFT / RootSIFT description.
//
// Reference behaviour (file:line relative to the reference root), identity view (H = I):
//   ReprojectRegionsAndRemoveTouchBoundary(dontRemove)   synth-detection.cpp:151-190
//   DetectOrientation / EstimateDominantAnglesFunctor    synth-detection.cpp:1039-1149, 836-929
//   ReprojectRegions                                     synth-detection.cpp:631-706
//   DescribeRegions<SIFTDescriptor> (non-fast branch)    synth-detection.hpp:170-263
//   photometricallyNormalize                             detectors/helpers.cpp:666-715
//   SIFTDescriptor (gradients, samplePatch, norms)       matching/siftdesc.cpp:22-131, 133-158, 199-263, 346-400
//
// Every accumulation the reference performs sequentially in floating point (sample
// coordinates, histogram bins, photometric sums, descriptor norms) is evaluated in the same
// order here: coordinates by one lane per patch row, histogram bins by one lane per bin that
// scans the pixels in raster order, scalar sums by a single lane.
#include "common.hpp"
#include "detmath.hpp"
#include "device_util.hpp"

namespace mods {

struct DescConst {
  int w, h;
  int max_cand, max_reg;
  double ks;               // synth-detection.cpp:21  k_sigma = 2*3*sqrt(3)
  int ori_ps;
  double ori_i2p;          // imageToPatchScale of DetectOrientation = (2*int(mrSize)+1)/patchSize
  int max_angles;
  double ori_th;
  double desc_mr;
  int desc_ps;
  int photo, root;
  double max_bin;
  int p2_lo, p2_hi;        // size tier handled by this launch: p2_lo < P2 <= p2_hi (P2 = 0: direct branch)
  size_t scratch_stride;   // floats per block
  int tap_cap;
};

struct OriOut { double a11, a12, a21, a22; int alive; int pad; };

// ---------------------------------------------------------------------------------------
// EstimateDominantAnglesFunctor for maxAngles = 1 on a ps x ps patch held in LDS.
// Block = 64 lanes.  s_val/s_bin: ps*(ps-2) entries, s_hist: 40 floats.  Returns found/angle
// (uniform across lanes).
// ---------------------------------------------------------------------------------------
__device__ bool dominant_angle_wave(const float *s_patch, const float *__restrict__ orimask, int ps, double th,
                                    float *s_val, int *s_bin, float *s_hist, float *angle_out) {
  const int lane = threadIdx.x;
  const int bins = 36;
  const float PIf = 3.14159265358979323846f;
  const int n = ps * (ps - 2);
  for (int p = lane; p < n; p += 64) {
    const int r = 1 + p / ps, c = p - (r - 1) * ps;
    float mag = 0.f, ori = 0.f;
    if (c >= 1 && c < ps - 1) {
      const float xgrad = s_patch[r * ps + c + 1] - s_patch[r * ps + c - 1];
      const float ygrad = s_patch[(r + 1) * ps + c] - s_patch[(r - 1) * ps + c];
      mag = sqrtf(xgrad * xgrad + ygrad * ygrad);
      ori = atan2_lut_ff(ygrad, xgrad);
    }
    const float m = orimask[r * ps + c];
    int bin = -1;
    float v = 0.f;
    if (m > 0 && (double)mag > 1.0) {
      bin = (int)(bins * (ori / PIf + 1.0f) / 2.0f);
      v = mag * m;
    }
    s_val[p] = v;
    s_bin[p] = bin;
  }
  __syncthreads();
  // one lane per bin, pixels in raster order (bin 36 is write-only in the reference)
  if (lane < bins) {
    float acc = 0.f;
    for (int p = 0; p < n; p++)
      if (s_bin[p] == lane) acc += s_val[p];
    s_hist[lane] = acc;
  }
  __syncthreads();
  // smoothCircularBuffer x6: new[i] = (old[i-1] + old[i]) + old[i+1], circular
  for (int it = 0; it < 6; it++) {
    float nv = 0.f;
    if (lane < bins) {
      const int a = lane == 0 ? bins - 1 : lane - 1, c = lane == bins - 1 ? 0 : lane + 1;
      nv = s_hist[a] + s_hist[lane] + s_hist[c];
    }
    __syncthreads();
    if (lane < bins) s_hist[lane] = nv;
    __syncthreads();
  }
  float thresh = 0.0f;
  for (int i = 0; i < bins; i++)
    if (s_hist[i] > thresh) thresh = s_hist[i];
  thresh = (float)((double)thresh * th);
  bool peak = false;
  if (lane < bins) {
    const int a = lane == 0 ? bins - 1 : lane - 1, c = lane == bins - 1 ? 0 : lane + 1;
    peak = s_hist[lane] >= thresh && s_hist[lane] > s_hist[a] && s_hist[lane] > s_hist[c];
  }
  const unsigned long long m = __ballot(peak);
  if (m == 0) return false;
  const int b = __ffsll((long long)m) - 1;
  const int a = b == 0 ? bins - 1 : b - 1, c = b == bins - 1 ? 0 : b + 1;
  const float ha = s_hist[a], hb = s_hist[b], hc = s_hist[c];
  const float pp = (ha - hc) / (ha - 2.0f * hb + hc) / 2.0f;
  *angle_out = 2.0f * PIf * (b + 0.5f + pp) / bins - PIf;
  return true;
}

// grid = (N, n_img), block = 64.  One wave per detected keypoint.
// dynamic LDS: ps*ps*(2 coords + patch) + ps*(ps-2)*(val + bin) + 40
__global__ __launch_bounds__(64) void orient_kernel(const float *__restrict__ img_all, DescConst k,
                                                    const mods_affkey *__restrict__ keys_all,
                                                    const int *__restrict__ key_count, const float *__restrict__ orimask,
                                                    OriOut *__restrict__ ori_all) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.ori_ps, pp2 = ps * ps;
  const int cst = ps + 1;   // padded row stride of the coordinate tiles (bank-conflict free column writes)
  float *s_cx = smem, *s_cy = s_cx + ps * cst, *s_patch = s_cy + ps * cst, *s_val = s_patch + pp2;
  int *s_bin = (int *)(s_val + ps * (ps - 2));
  float *s_hist = (float *)(s_bin + ps * (ps - 2));
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const float *img = img_all + (size_t)k.w * k.h * b;
  const mods_affkey *keys = keys_all + (size_t)b * k.max_cand;
  OriOut *ori = ori_all + (size_t)b * k.max_cand;
  int n = key_count[b];
  if (n > k.max_cand) n = k.max_cand;
  const int half = ps / 2;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const mods_affkey kp = keys[i];
    bool alive = (kp.x < k.w) && (kp.y < k.h) && (kp.x > 0) && (kp.y > 0);
    const float fx = (float)kp.x, fy = (float)kp.y;
    const float f11 = (float)kp.a11, f12 = (float)kp.a12, f21 = (float)kp.a21, f22 = (float)kp.a22;
    const int box = (int)(k.ks * kp.s);
    if (alive && check_borders(k.w, k.h, fx, fy, f11, f12, f21, f22, box, box)) alive = false;
    double n11 = kp.a11, n12 = kp.a12, n21 = kp.a21, n22 = kp.a22;
    if (alive && k.max_angles <= 0) alive = false;   // DetectOrientation pushes nothing (addUpRight = false)
    if (alive) {
      const float curr_sc = (float)(k.ori_i2p * kp.s);
      const float a11 = f11 * curr_sc, a12 = f12 * curr_sc, a21 = f21 * curr_sc, a22 = f22 * curr_sc;
      const bool touch = check_borders(k.w, k.h, fx, fy, a11, a12, a21, a22, ps, ps);
      __syncthreads();
      if (lane < ps) {
        float rx = fx - (float)half * a12;
        float ry = fy - (float)half * a22;
        for (int q = 0; q < lane; q++) { rx += a12; ry += a22; }
        float WX = rx - (float)half * a11;
        float WY = ry - (float)half * a21;
        for (int c = 0; c < ps; c++) {
          s_cx[lane * cst + c] = WX;
          s_cy[lane * cst + c] = WY;
          WX += a11;
          WY += a21;
        }
      }
      __syncthreads();
      for (int p = lane; p < pp2; p += 64) {
        const int q = (p / ps) * cst + (p % ps);
        s_patch[p] = bilinear_tap(img, k.w, k.h, s_cx[q], s_cy[q], touch);
      }
      __syncthreads();
      float ang = 0.f;
      const bool found = dominant_angle_wave(s_patch, orimask, ps, k.ori_th, s_val, s_bin, s_hist, &ang);
      if (!found) alive = false;
      else {
        double si, ci;
        det_sincos(-(double)ang, &si, &ci);
        n11 = kp.a11 * ci - kp.a12 * si;
        n12 = kp.a11 * si + kp.a12 * ci;
        n21 = kp.a21 * ci - kp.a22 * si;
        n22 = kp.a21 * si + kp.a22 * ci;
        // ReprojectRegions (H = I): same centre, rotated frame
        if (check_borders(k.w, k.h, fx, fy, (float)n11, (float)n12, (float)n21, (float)n22, box, box)) alive = false;
      }
    }
    if (lane == 0) {
      OriOut o;
      o.a11 = n11; o.a12 = n12; o.a21 = n21; o.a22 = n22; o.alive = alive ? 1 : 0; o.pad = 0;
      ori[i] = o;
    }
  }
}

// Order-preserving compaction of the surviving keypoints into the region list.
// grid = (1, n_img), block = 1024.
__global__ __launch_bounds__(1024) void compact_regions_kernel(DescConst k, const mods_affkey *__restrict__ keys_all,
                                                               const int *__restrict__ key_count,
                                                               const OriOut *__restrict__ ori_all,
                                                               mods_region *__restrict__ reg_all, int *__restrict__ reg_count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const mods_affkey *keys = keys_all + (size_t)b * k.max_cand;
  const OriOut *ori = ori_all + (size_t)b * k.max_cand;
  mods_region *reg = reg_all + (size_t)b * k.max_reg;
  int n = key_count[b];
  if (n > k.max_cand) n = k.max_cand;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const bool alive = i < n && ori[i].alive;
    const unsigned long long m = __ballot(alive);
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int q = 0; q < wv; q++) off += s_wave[q];
    if (alive) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < k.max_reg) {
        const mods_affkey kp = keys[i];
        const OriOut o = ori[i];
        mods_region r;
        r.x = kp.x; r.y = kp.y; r.s = kp.s; r.a11 = o.a11; r.a12 = o.a12; r.a21 = o.a21; r.a22 = o.a22;
        r.response = kp.response; r.sub_type = kp.sub_type; r.id = slot; r.parent = i; r.pad = 0;
        // descriptor bytes are written by describe_kernel; copy the POD head only
        memcpy(&reg[slot], &r, offsetof(mods_region, desc));
      }
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int q = 0; q < 16; q++) t += s_wave[q]; s_base += t; }
    __syncthreads();
  }
  if (tid == 0) reg_count[b] = s_base;
}

// ---------------------------------------------------------------------------------------
// SIFT on a ps x ps patch in LDS (block = 256).  Tables: bin0/bin1 (already *8), w0/w1.
// ---------------------------------------------------------------------------------------
struct SiftTab {          // precomputeBinsAndWeights, siftdesc.cpp:22-71 (host built, patchSize <= 64)
  int bin0[64], bin1[64];
  float w0[64], w1[64];   // stored as double in the reference but float valued
};

__device__ void sift_from_patch(const float *s_patch, const float *__restrict__ mask, const SiftTab *__restrict__ tab,
                                int ps, bool rootsift, double max_bin, float *s_val, int *s_bo0, float *s_wo1,
                                double *s_vec, double *s_red, uint8_t *out) {
  const int tid = threadIdx.x;
  const int pp = ps * ps;
  const double M_PI_DOUBLED = 6.28318530718;
  // gradients (siftdesc.cpp:352-373) + per-pixel weight/orientation split (:88-104)
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
    const float ori = atan2_lut_ff(ygrad, xgrad);
    s_val[p] = mask[p] * grad;
    const float o = (float)(8.0f * ((double)ori + M_PI_DOUBLED) / M_PI_DOUBLED);
    int bo0 = (int)o;
    s_wo1[p] = o - bo0;
    s_bo0[p] = bo0 % 8;
  }
  __syncthreads();
  // samplePatch: thread t < 128 owns vec[t], t = br*32 + bc*8 + bo; pixels visited in raster order
  if (tid < 128) {
    const int br8 = (tid >> 5) * 8, bc8 = ((tid >> 3) & 3) * 8, bo = tid & 7;
    int rlo = ps, rhi = 0, clo = ps, chi = 0;
    for (int i = 0; i < ps; i++) {
      if (tab->bin0[i] == br8 || tab->bin1[i] == br8) { rlo = min(rlo, i); rhi = max(rhi, i + 1); }
      if (tab->bin0[i] == bc8 || tab->bin1[i] == bc8) { clo = min(clo, i); chi = max(chi, i + 1); }
    }
    double acc = 0.0;
    for (int r = rlo; r < rhi; r++) {
      const bool r0m = tab->bin0[r] == br8, r1m = tab->bin1[r] == br8;
      const float wr0 = tab->w0[r], wr1 = tab->w1[r];
      for (int c = clo; c < chi; c++) {
        const bool c0m = tab->bin0[c] == bc8, c1m = tab->bin1[c] == bc8;
        if (!(c0m || c1m)) continue;
        const int p = r * ps + c;
        const float pv = s_val[p];
        const float wc0 = (float)((double)tab->w0[c] * pv);
        const float wc1 = (float)((double)tab->w1[c] * pv);
        const int bo0 = s_bo0[p];
        const int bo1 = (bo0 + 1) % 8;
        if (bo0 != bo && bo1 != bo) continue;
        const float wo1 = s_wo1[p];
        const float wo0 = 1.0f - wo1;
        const float wo = (bo0 == bo) ? wo0 : wo1;
        float val;
        if (r0m && c0m) { val = wr0 * wc0; if (val > 0) acc += (double)(val * wo); }
        if (r0m && c1m) { val = wr0 * wc1; if (val > 0) acc += (double)(val * wo); }
        if (r1m && c0m) { val = wr1 * wc0; if (val > 0) acc += (double)(val * wo); }
        if (r1m && c1m) { val = wr1 * wc1; if (val > 0) acc += (double)(val * wo); }
      }
    }
    s_vec[tid] = acc;
  }
  __syncthreads();
  // normalize (siftdesc.cpp:133-158) / clip / renormalise (:199-210, :248-257)
  for (int pass = 0; pass < 2; pass++) {
    if (tid == 0) {
      double len = 0.0;
      for (int i = 0; i < 128; i += 4) {
        const double sq0 = s_vec[i] * s_vec[i], sq1 = s_vec[i + 1] * s_vec[i + 1];
        const double sq2 = s_vec[i + 2] * s_vec[i + 2], sq3 = s_vec[i + 3] * s_vec[i + 3];
        len += sq0 + sq1 + sq2 + sq3;
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
      for (int i = 0; i < 128; i++) sum += fabs(s_vec[i]);
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

// photometricallyNormalize, helpers.cpp:666-715, on an LDS patch (block = 256).
__device__ void photonorm_patch(float *s_patch, const float *__restrict__ mask, int pp, float *s_red) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    float sum = 0, gsum = 0;
    for (int p = 0; p < pp; p++)
      if (mask[p] > 0) { sum += s_patch[p]; gsum++; }
    sum = sum / gsum;
    float var = 0;
    for (int p = 0; p < pp; p++)
      if (mask[p] > 0) var += (sum - s_patch[p]) * (sum - s_patch[p]);
    var = sqrtf(var / gsum);
    s_red[0] = sum;
    s_red[1] = var;
  }
  __syncthreads();
  const float sum = s_red[0], var = s_red[1];
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

// grid = (N, n_img), block = 256.  One block per region; regions outside this launch's size
// tier are skipped.  Global scratch per block: S (P2 x P2) then the row-pass strip (P2 x 2ps).
// dynamic LDS layout: patch pp floats | union { out (2ps)^2 ; val,wo1,bo0 (3 pp) ; cx,cy (2 pp) } |
//   seq 2*ps floats | cidx 2*ps ints | vec 128 doubles | red 8 doubles | taps tap_cap floats
// (the resampling grid, the coordinate tiles and the SIFT scratch are never live together)
__global__ __launch_bounds__(256) void describe_kernel(const float *__restrict__ img_all, DescConst k, int n_img,
                                                       mods_region *__restrict__ reg_all, const int *__restrict__ reg_count,
                                                       const float *__restrict__ mask, const SiftTab *__restrict__ tab,
                                                       float *__restrict__ scratch, int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.desc_ps, pp = ps * ps, ps2 = 2 * ps;
  float *s_patch = smem;
  float *s_un = s_patch + pp;
  float *s_out = s_un;
  float *s_val = s_un;
  float *s_wo1 = s_val + pp;
  int *s_bo0 = (int *)(s_wo1 + pp);
  float *s_cx = s_un;
  float *s_cy = s_cx + pp;
  float *s_seq = s_un + ps2 * ps2;
  int *s_cidx = (int *)(s_seq + ps2);
  double *s_vec = (double *)(((uintptr_t)(s_cidx + ps2) + 7) & ~(uintptr_t)7);
  double *s_red = s_vec + 128;
  float *s_tap = (float *)(s_red + 8);
  const int tid = threadIdx.x;
  float *S = scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * k.scratch_stride;
  for (int b = blockIdx.y; b < n_img; b += gridDim.y) {
  const float *img = img_all + (size_t)k.w * k.h * b;
  mods_region *reg = reg_all + (size_t)b * k.max_reg;
  int n = reg_count[b];
  if (n > k.max_reg) n = k.max_reg;
  for (int ri = blockIdx.x; ri < n; ri += gridDim.x) {
    const double rs = reg[ri].s;
    const float mrScale = (float)ceil(rs * k.desc_mr);
    int P = 2 * int(mrScale) + 1;
    const float scale = float(P) / float(ps);
    const bool big = (double)scale > 0.4;
    const int P2 = big ? P + 2 : 0;
    if (!(P2 > k.p2_lo && P2 <= k.p2_hi)) continue;
    const float fx = (float)reg[ri].x, fy = (float)reg[ri].y;
    const float f11 = (float)reg[ri].a11, f12 = (float)reg[ri].a12, f21 = (float)reg[ri].a21, f22 = (float)reg[ri].a22;
    __syncthreads();
    if (big) {
      const int n_tap = ((int)(2.0 * 3.0 * (1.5f * scale) + 1.0)) | 1;
      if ((size_t)P2 * P2 + (size_t)P2 * ps2 > k.scratch_stride || n_tap > k.tap_cap) {
        if (tid == 0) atomicExch(err_flag, 1);
        continue;
      }
      float *T = S + (size_t)P2 * P2;
      // 1. S = interpolate(img, x, y, A) on P2 x P2: one thread per patch row, sequential walk
      {
        const bool touch = check_borders(k.w, k.h, fx, fy, f11, f12, f21, f22, P2, P2);
        const int half = P2 / 2;
        float rx = fx - (float)half * f12;
        float ry = fy - (float)half * f22;
        int row_at = 0;
        for (int row = tid; row < P2; row += 256) {
          for (; row_at < row; row_at++) { rx += f12; ry += f22; }
          float WX = rx - (float)half * f11;
          float WY = ry - (float)half * f21;
          float *dst = S + (size_t)row * P2;
          for (int c = 0; c < P2; c++) {
            dst[c] = bilinear_tap(img, k.w, k.h, WX, WY, touch);
            WX += f11;
            WY += f21;
          }
        }
      }
      // 2. Gaussian taps for sigma = 1.5f * scale (getGaussianKernel, CV_32F) and the resampling
      //    coordinate sequence X_i = Y_i of interpolate(smoothed, c0, c0, scale, 0, 0, scale)
      const float sigma = 1.5f * scale;
      const int r_tap = n_tap >> 1;
      {
        const double sig = (double)sigma;
        const double scale2X = -0.5 / (sig * sig);
        for (int i = tid; i < n_tap; i += 256) {
          const double x = i - (n_tap - 1) * 0.5;
          s_tap[i] = (float)det_exp(scale2X * x * x);
        }
      }
      const float c0 = (float)(P2 >> 1);
      if (tid == 0) {
        const int halfp = ps / 2;
        float v = c0 - (float)halfp * scale;   // rx - halfWidth*a11 with rx = c0 - halfHeight*0
        for (int i = 0; i < ps; i++) {
          s_seq[i] = v;
          const int fl = (int)floorf(v);
          int i0 = fl, i1 = fl + 1;
          i0 = i0 < 0 ? 0 : (i0 > P2 - 1 ? P2 - 1 : i0);
          i1 = i1 < 0 ? 0 : (i1 > P2 - 1 ? P2 - 1 : i1);
          s_cidx[2 * i] = i0;
          s_cidx[2 * i + 1] = i1;
          v += scale;
        }
      }
      __syncthreads();
      if (tid == 0) {
        double sum = 0;
        for (int i = 0; i < n_tap; i++) sum += s_tap[i];
        s_red[0] = 1. / sum;
      }
      __syncthreads();
      for (int i = tid; i < n_tap; i += 256) s_tap[i] = (float)(s_tap[i] * s_red[0]);
      __syncthreads();
      // 3. row pass on the 2*ps needed columns, every row (taps left to right, REPLICATE)
      for (int e = tid; e < P2 * ps2; e += 256) {
        const int y = e / ps2, q = e - y * ps2;
        const int x = s_cidx[q];
        const float *row = S + (size_t)y * P2;
        int x0 = x - r_tap; x0 = x0 < 0 ? 0 : x0;
        float s = s_tap[0] * row[x0];
        for (int j = 1; j < n_tap; j++) {
          int xx = x - r_tap + j;
          xx = xx < 0 ? 0 : (xx > P2 - 1 ? P2 - 1 : xx);
          s += s_tap[j] * row[xx];
        }
        T[e] = s;
      }
      __syncthreads();
      // 4. column pass on the 2*ps needed rows (centre tap first, symmetric pairs)
      for (int e = tid; e < ps2 * ps2; e += 256) {
        const int yq = e / ps2, q = e - yq * ps2;
        const int y = s_cidx[yq];
        float s = s_tap[r_tap] * T[(size_t)y * ps2 + q];
        for (int j = 1; j <= r_tap; j++) {
          int yp = y + j; yp = yp > P2 - 1 ? P2 - 1 : yp;
          int ym = y - j; ym = ym < 0 ? 0 : ym;
          s += s_tap[r_tap + j] * (T[(size_t)yp * ps2 + q] + T[(size_t)ym * ps2 + q]);
        }
        s_out[e] = s;
      }
      __syncthreads();
      // 5. patch = interpolate(smoothed, c0, c0, scale, 0, 0, scale) through the compact 2ps x 2ps grid
      {
        const bool touch2 = check_borders(P2, P2, c0, c0, scale, 0.f, 0.f, scale, ps, ps);
        for (int p = tid; p < pp; p += 256) {
          const int j = p / ps, i = p - j * ps;
          const float WX = s_seq[i], WY = s_seq[j];
          float v;
          const int x = touch2 ? (int)floorf(WX) : (int)WX;
          const int y = touch2 ? (int)floorf(WY) : (int)WY;
          if (!touch2 || (WX >= 0 && WY >= 0 && x < P2 - 1 && y < P2 - 1)) {
            const float wx = WX - (float)x;
            const float *R0 = s_out + (2 * j) * ps2 + 2 * i;
            const float *R1 = R0 + ps2;
            const float I1 = wx * (R0[1] - R0[0]) + R0[0];
            v = (WY - y) * (wx * (R1[1] - R1[0]) + R1[0] - I1) + I1;
          } else v = 0.f;
          s_patch[p] = v;
        }
      }
    } else {
      // direct branch: interpolate(img, x, y, A*scale) -> ps x ps
      const float a11 = f11 * scale, a12 = f12 * scale, a21 = f21 * scale, a22 = f22 * scale;
      const bool touch = check_borders(k.w, k.h, fx, fy, a11, a12, a21, a22, ps, ps);
      const int half = ps / 2;
      if (tid < ps) {
        float rx = fx - (float)half * a12;
        float ry = fy - (float)half * a22;
        for (int q = 0; q < tid; q++) { rx += a12; ry += a22; }
        float WX = rx - (float)half * a11;
        float WY = ry - (float)half * a21;
        for (int c = 0; c < ps; c++) {
          s_cx[tid * ps + c] = WX;
          s_cy[tid * ps + c] = WY;
          WX += a11;
          WY += a21;
        }
      }
      __syncthreads();
      for (int p = tid; p < pp; p += 256) s_patch[p] = bilinear_tap(img, k.w, k.h, s_cx[p], s_cy[p], touch);
    }
    __syncthreads();
    if (k.photo) photonorm_patch(s_patch, mask, pp, (float *)s_red);
    sift_from_patch(s_patch, mask, tab, ps, k.root != 0, k.max_bin, s_val, s_bo0, s_wo1, s_vec, s_red, reg[ri].desc);
  }
  }
}

// single-patch entry points for the parity tests
__global__ __launch_bounds__(64) void dominant_angle_test_kernel(const float *__restrict__ patch, int ps, double th,
                                                                 const float *__restrict__ orimask, float *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_patch = smem, *s_val = s_patch + ps * ps;
  int *s_bin = (int *)(s_val + ps * (ps - 2));
  float *s_hist = (float *)(s_bin + ps * (ps - 2));
  for (int p = threadIdx.x; p < ps * ps; p += 64) s_patch[p] = patch[p];
  __syncthreads();
  float ang = 0.f;
  const bool f = dominant_angle_wave(s_patch, orimask, ps, th, s_val, s_bin, s_hist, &ang);
  if (threadIdx.x == 0) { out[0] = f ? 1.f : 0.f; out[1] = ang; }
}

__global__ __launch_bounds__(256) void sift_patch_test_kernel(const float *__restrict__ patch, int ps, int root, double max_bin,
                                                              const float *__restrict__ mask, const SiftTab *__restrict__ tab,
                                                              uint8_t *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pp = ps * ps;
  float *s_patch = smem, *s_val = s_patch + pp, *s_wo1 = s_val + pp;
  int *s_bo0 = (int *)(s_wo1 + pp);
  double *s_vec = (double *)(((uintptr_t)(s_bo0 + pp) + 7) & ~(uintptr_t)7);
  double *s_red = s_vec + 128;
  for (int p = threadIdx.x; p < pp; p += 256) s_patch[p] = patch[p];
  __syncthreads();
  sift_from_patch(s_patch, mask, tab, ps, root != 0, max_bin, s_val, s_bo0, s_wo1, s_vec, s_red, out);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static void build_sift_tab(int ps, SiftTab *t) {   // siftdesc.cpp:22-71, spatialBins 4, orientationBins 8
  const int spatialBins = 4, orientationBins = 8;
  int halfSize = ps >> 1;
  float step = float(spatialBins + 1) / (2 * halfSize);
  for (int i = 0; i < 64; i++) { t->bin0[i] = t->bin1[i] = -1; t->w0[i] = t->w1[i] = 0; }
  for (int i = 0; i < ps; i++) {
    float x = step * i;
    int xi = (int)(x);
    int b0 = xi - 1, b1 = xi;
    float w1 = x - xi;
    float w0 = 1.0f - w1;
    if (b0 < 0) { b0 = 0; w0 = 0; }
    if (b0 >= spatialBins) { b0 = spatialBins - 1; w0 = 0; }
    if (b1 < 0) { b1 = 0; w1 = 0; }
    if (b1 >= spatialBins) { b1 = spatialBins - 1; w1 = 0; }
    t->bin0[i] = b0 * orientationBins; t->bin1[i] = b1 * orientationBins;
    t->w0[i] = w0; t->w1[i] = w1;
  }
}

int describe_configure(mods_ctx *ctx, const mods_describe_params *par) {
  if (par->ori_patchSize < 8 || par->ori_patchSize > 48 || par->desc_patchSize < 9 || par->desc_patchSize > 63 ||
      !(par->desc_patchSize & 1)) { set_error("unsupported patch sizes (ori %d, desc %d)", par->ori_patchSize, par->desc_patchSize); return MODS_E_ARG; }
  if (par->ori_maxAngles > 1) { set_error("maxAngles > 1 is not supported"); return MODS_E_ARG; }
  if (!ctx->desc_tables_dev) {
    MODS_HIP_CHECK(hipMalloc(&ctx->desc_tables_dev, sizeof(float) * (64 * 64 * 2) + sizeof(SiftTab)));
    MODS_HIP_CHECK(hipMalloc(&ctx->desc_err_dev, sizeof(int)));
    MODS_HIP_CHECK(hipMemsetAsync(ctx->desc_err_dev, 0, sizeof(int), ctx->stream));
  }
  if (ctx->desc_ori_ps != par->ori_patchSize || ctx->desc_ps != par->desc_patchSize) {
    std::vector<float> m1((size_t)64 * 64, 0.f), m2((size_t)64 * 64, 0.f);
    circular_gauss_mask_host(par->ori_patchSize, par->ori_patchSize / 3.0f, m1.data());   // synth-detection.cpp:852
    circular_gauss_mask_host(par->desc_patchSize, 0.f, m2.data());                         // synth-detection.hpp:181, siftdesc.h:83
    SiftTab tab;
    build_sift_tab(par->desc_patchSize, &tab);
    MODS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MODS_HIP_CHECK(hipMemcpy(ctx->desc_tables_dev, m1.data(), sizeof(float) * 4096, hipMemcpyHostToDevice));
    MODS_HIP_CHECK(hipMemcpy(ctx->desc_tables_dev + 4096, m2.data(), sizeof(float) * 4096, hipMemcpyHostToDevice));
    MODS_HIP_CHECK(hipMemcpy(ctx->desc_tables_dev + 8192, &tab, sizeof(SiftTab), hipMemcpyHostToDevice));
    ctx->desc_ori_ps = par->ori_patchSize; ctx->desc_ps = par->desc_patchSize;
  }
  return MODS_OK;
}

static size_t desc_lds_bytes(int ps, int tap_cap) {
  const size_t pp = (size_t)ps * ps, ps2 = 2 * (size_t)ps;
  return sizeof(float) * (pp + ps2 * ps2 + 2 * ps2 + tap_cap) + sizeof(double) * (128 + 8) + 16;   // 3*pp <= (2ps)^2
}

// Orientation + compaction + description of the keypoints in ctx->keys_dev (as left by detect_run
// or uploaded by mods_orient_describe) for images `img_dev` [n_img][h][w].
int describe_run(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, const mods_describe_params *par) {
  int rc = describe_configure(ctx, par);
  if (rc) return rc;
  DescConst k;
  k.w = w; k.h = h; k.max_cand = ctx->max_cand; k.max_reg = ctx->max_cand;
  k.ks = 2 * 3.0 * sqrt(3.0);
  k.ori_ps = par->ori_patchSize;
  k.ori_i2p = double(2 * int(par->ori_mrSize) + 1) / (double)par->ori_patchSize;
  k.max_angles = par->ori_maxAngles;
  k.ori_th = par->ori_threshold;
  k.desc_mr = par->desc_mrSize; k.desc_ps = par->desc_patchSize; k.photo = par->photoNorm; k.root = par->rootSift;
  k.max_bin = par->maxBinValue;
  int *key_count = ctx->cand_count + 2 * ctx->batch;
  const float *orimask = ctx->desc_tables_dev, *dmask = ctx->desc_tables_dev + 4096;
  const SiftTab *tab = (const SiftTab *)(ctx->desc_tables_dev + 8192);
  {
    StageScope ts(ctx, MODS_STAGE_ORIENT);
    const int ps = k.ori_ps;
    const size_t lds = sizeof(float) * (2 * (size_t)ps * (ps + 1) + (size_t)ps * ps + 2 * (size_t)ps * (ps - 2) + 48);
    hipLaunchKernelGGL(orient_kernel, dim3(8192, n_img), dim3(64), lds, ctx->stream, img_dev, k, ctx->keys_dev, key_count,
                       orimask, (OriOut *)ctx->ori_dev);
    hipLaunchKernelGGL(compact_regions_kernel, dim3(1, n_img), dim3(1024), 0, ctx->stream, k, ctx->keys_dev, key_count,
                       (const OriOut *)ctx->ori_dev, ctx->regions_dev, ctx->region_count);
    MODS_HIP_CHECK(hipGetLastError());
  }
  {
    StageScope ts(ctx, MODS_STAGE_DESCRIBE);
    // tier A: direct branch and P2 <= 160 (one slab of 160*160 + 160*2ps floats per block)
    const int capA = 160;
    const int blocksA = 1024;
    const size_t strideA = (size_t)capA * capA + (size_t)capA * 2 * k.desc_ps;
    // tier B: everything larger, few blocks with slabs sized for P2 <= 3*max(w,h)
    const int capB = 3 * std::max(w, h) + 8;
    const int blocksB = 8;
    const size_t strideB = (size_t)capB * capB + (size_t)capB * 2 * k.desc_ps;
    const size_t need = (size_t)blocksA * strideA * n_img + (size_t)blocksB * strideB;
    if (need > ctx->desc_scratch_elems) {
      MODS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (ctx->desc_scratch) MODS_HIP_CHECK(hipFree(ctx->desc_scratch));
      ctx->desc_scratch = nullptr;
      MODS_HIP_CHECK(hipMalloc(&ctx->desc_scratch, need * sizeof(float)));
      ctx->desc_scratch_elems = need;
    }
    k.p2_lo = -1; k.p2_hi = capA; k.scratch_stride = strideA; k.tap_cap = 64;
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
      MODS_HIP_CHECK(hipFuncSetAttribute((const void *)describe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      lds_attr_set = true;
    }
    hipLaunchKernelGGL(describe_kernel, dim3(blocksA, n_img), dim3(256), desc_lds_bytes(k.desc_ps, k.tap_cap), ctx->stream,
                       img_dev, k, n_img, ctx->regions_dev, ctx->region_count, dmask, tab, ctx->desc_scratch, ctx->desc_err_dev);
    k.p2_lo = capA; k.p2_hi = 1 << 30; k.scratch_stride = strideB; k.tap_cap = 4096;
    hipLaunchKernelGGL(describe_kernel, dim3(blocksB, 1), dim3(256), desc_lds_bytes(k.desc_ps, k.tap_cap), ctx->stream,
                       img_dev, k, n_img, ctx->regions_dev, ctx->region_count, dmask, tab,
                       ctx->desc_scratch + (size_t)blocksA * strideA * n_img, ctx->desc_err_dev);
    MODS_HIP_CHECK(hipGetLastError());
  }
  return MODS_OK;
}

int launch_dominant_angle_test(mods_ctx *ctx, const float *patch_dev, int ps, double th, float *out_dev) {
  const size_t lds = sizeof(float) * ((size_t)ps * ps + 2 * (size_t)ps * (ps - 2) + 48);
  hipLaunchKernelGGL(dominant_angle_test_kernel, dim3(1), dim3(64), lds, ctx->stream, patch_dev, ps, th,
                     ctx->desc_tables_dev, out_dev);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

int launch_sift_patch_test(mods_ctx *ctx, const float *patch_dev, int ps, int root, double max_bin, uint8_t *out_dev) {
  const size_t lds = sizeof(float) * (4 * (size_t)ps * ps) + sizeof(double) * 136 + 16;
  hipLaunchKernelGGL(sift_patch_test_kernel, dim3(1), dim3(256), lds, ctx->stream, patch_dev, ps, root, max_bin,
                     ctx->desc_tables_dev + 4096, (const SiftTab *)(ctx->desc_tables_dev + 8192), out_dev);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods
