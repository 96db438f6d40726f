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
#include "describe_common.hpp"
#include "detmath.hpp"
#include "device_util.hpp"

namespace mods {

struct OriOut { double a11, a12, a21, a22; int alive; int pad; };

// Histogram bin of EstimateDominantAnglesFunctor, (int)(36 * (ori / pi + 1) / 2) (synth-detection.cpp:870), for every
// value ori = atan2LUTff(.) can take: [8 cases][256 table entries], see atan2_lut_sel.  Rewritten (same values) by
// ori_bin_table_kernel at the start of every orientation pass.
__device__ unsigned char g_ori_bin[2048];
__global__ __launch_bounds__(256) void ori_bin_table_kernel() {
  const float PIf = 3.14159265358979323846f;
  const int bins = 36;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const float ori = atan2_lut_case(e >> 8, g_atan_lut[e & 255]);
  g_ori_bin[e] = (unsigned char)(int)(bins * (ori / PIf + 1.0f) / 2.0f);
}

// ---------------------------------------------------------------------------------------
// EstimateDominantAnglesFunctor for maxAngles = 1 on a ps x ps patch held in LDS.
// Block = 64 lanes.  s_val/s_bin: ps*(ps-2) entries, s_hist: 40 floats.  Returns found/angle
// (uniform across lanes).
// ---------------------------------------------------------------------------------------
// addPeakAngle (synth-detection.cpp:824-834): parabolic refinement of the peak in bin b of the smoothed histogram
__device__ __forceinline__ float peak_angle(const float *s_hist, int b) {
  const int bins = 36;
  const float PIf = 3.14159265358979323846f;
  const int a = b == 0 ? bins - 1 : b - 1, c = b == bins - 1 ? 0 : b + 1;
  const float ha = s_hist[a], hb = s_hist[b], hc = s_hist[c];
  const float pp = (ha - hc) / (ha - 2.0f * hb + hc) / 2.0f;
  return 2.0f * PIf * (b + 0.5f + pp) / bins - PIf;
}
// The histogram is an ORDERED sum per bin (votes in raster order).  Round 3 let 36 lanes scan every vote (three vector instructions
// per vote and keypoint: half of this kernel's instructions, and the kernel is VALU-issue bound); since round 4 every vote is filed
// into its bin's list first and a lane adds its own bin's ~20 votes only:
//   pass 1, per 64 raster-consecutive pixels: the lanes that vote for the same bin find each other, a lane's place in its bin's list
//           = the bin's count so far + its rank among those lanes; the last lane of a group writes the new count (one writer per
//           bin: no conflict, and the wave's LDS accesses execute in program order);
//   then    the bins' list offsets (a prefix over 36 counts), the values scattered into the lists (the patch array is dead by then
//           and holds them), and lane b adds list b front to back - the same additions in the same order as the scan made (the scan
//           added +0.0f for every vote of another bin, which leaves a non-negative running sum unchanged).
// Round 5, by instruction count (tools/isa_stats.py: the loop over the pixels was 158 vector instructions per 64 pixels, half of the
// kernel's):
//   * "the lanes with my bin" comes out of LDS: every voting lane ORs its lane bit into its bin's 64-bit word (ds_or_b64; lanes of
//     one instruction that share a word are serialised by the LDS, the words of a wave's next read are complete because a wave's
//     LDS instructions execute in order), reads the word back and the group's last lane clears it - 3 LDS + 6 vector instructions
//     in the place of six ballots folded into a per-lane 64-bit mask (54 vector instructions);
//   * the pixel of vote p is patch index p + ps (row 1 + p / ps, column p % ps): no division; `votemask[p]` is the orientation mask
//     of that pixel, 0 in the first and last column (describe_configure), so the column test and the clamped neighbour loads go too
//     (a border column's gradient reads its in-row neighbours of the adjacent rows - finite values - and its vote is dropped);
//   * the square root without the denormal rescue of the compiler's expansion (fast_sqrtf, device_util.hpp).
// s_key: one word per pixel (bin | place << 6; bin 63 = no vote) - 16 bits while every place fits 10 bits (n <= 1024 votes: the
// default 32 x 32 patch has 960), 32 bits for larger orientation patches (ori_key_bytes); s_cnt / s_off: 64 ints each behind the
// histogram, then the 64 lane-set words.
__host__ __device__ static inline int ori_key_bytes(int ps) { return ps * (ps - 2) > 1024 ? 4 : 2; }   // bytes of a vote's (bin, place) word
__device__ bool dominant_angle_wave(float *s_patch, const float *__restrict__ votemask, int ps, double th,
                                    float *s_val, unsigned char *s_bin, float *s_hist, float *angle_out, int half = 0,
                                    unsigned long long *peaks_out = nullptr) {
  const int lane = threadIdx.x;
  const int bins = 36;
  const float PIf = 3.14159265358979323846f;
  const int n = ps * (ps - 2);
  unsigned short *s_key = (unsigned short *)s_bin;
  unsigned int *s_key32 = (unsigned int *)s_bin;
  const bool wide = n > 1024;             // a place can exceed 10 bits: 32-bit keys (uniform over the launch)
  int *s_cnt = (int *)(s_hist + 48), *s_off = s_cnt + 64;
  unsigned long long *s_set = (unsigned long long *)(s_off + 64);
  s_cnt[lane] = 0;
  s_set[lane] = 0ull;
  __syncthreads();
  const unsigned long long lanebit = 1ull << lane;
  const int zero_bin = (int)(bins * (0.f / PIf + 1.0f) / 2.0f);
  for (int p0 = 0; p0 < n; p0 += 64) {
    const int p = p0 + lane;
    int bin = 63;
    float v = 0.f;
    const float m = p < n ? votemask[(unsigned)p] : 0.f;
    if (m > 0) {
      const float *c = s_patch + p + ps;                 // the pixel; m > 0 only in the columns 1 .. ps - 2
      const float xgrad = c[1] - c[-1];
      const float ygrad = c[ps] - c[-ps];
      const AtanSel as = atan2_lut_sel(ygrad, xgrad);
      const int tbin = g_ori_bin[(unsigned)(as.oct * 256 + as.idx)];
      const float mag = fast_sqrtf(xgrad * xgrad + ygrad * ygrad);
      const int obin = as.zero ? zero_bin : tbin;
      if ((double)mag > 1.0 && obin < bins) {      // (bin 36 is write-only in the reference)
        bin = obin;
        v = mag * m;
      }
    }
    // the lanes of this round that vote for my bin (lanes without a vote: nobody)
    unsigned long long same = 0ull;
    if (bin < bins) __hip_atomic_fetch_or(s_set + bin, lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    wave_sync();
    if (bin < bins) same = s_set[bin];
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
    const int group = __popcll(same);
    const bool last = bin < bins && rank == group - 1;
    const int before = s_cnt[bin];
    if (p < n) {
      s_val[p] = v;
      const unsigned key = (unsigned)bin | ((unsigned)(before + rank) << 6);
      if (wide) s_key32[p] = key; else s_key[p] = (unsigned short)key;
    }
    if (last) { s_cnt[bin] = before + group; s_set[bin] = 0ull; }
    wave_sync();
  }
  __syncthreads();
  const int mine = lane < bins ? s_cnt[lane] : 0;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  s_off[lane] = incl - mine;
  __syncthreads();
  for (int p = lane; p < n; p += 64) {
    const unsigned key = wide ? s_key32[p] : (unsigned)s_key[p];
    const int b = (int)(key & 63u);
    if (b < bins) s_patch[s_off[b] + (int)(key >> 6)] = s_val[p];
  }
  __syncthreads();
  if (lane < bins) {
    float acc = 0.f;
    const float *list = s_patch + (incl - mine);
    for (int i = 0; i < mine; i++) acc += list[i];
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
  if (half) {   // doHalfSIFT (synth-detection.cpp:891-898): bins i and i + 18 folded AFTER the threshold was taken
    float nv = 0.f;
    if (lane < bins / 2) nv = s_hist[lane] + s_hist[lane + bins / 2];
    __syncthreads();
    if (lane < bins) s_hist[lane] = lane < bins / 2 ? nv : 0.f;
    __syncthreads();
  }
  bool peak = false;
  if (lane < bins) {
    const int a = lane == 0 ? bins - 1 : lane - 1, c = lane == bins - 1 ? 0 : lane + 1;
    peak = s_hist[lane] >= thresh && s_hist[lane] > s_hist[a] && s_hist[lane] > s_hist[c];
  }
  const unsigned long long m = __ballot(peak);
  if (peaks_out) *peaks_out = m;          // every local maximum above the threshold, bit = bin (addPeakAngle in bin order)
  if (m == 0) return false;
  *angle_out = peak_angle(s_hist, __ffsll((long long)m) - 1);
  return true;
}

// grid = (N, n_img), block = 64.  One wave per detected keypoint.
// dynamic LDS: patch ps*ps | vote values ps*(ps-2) (padded to 16) | vote bins (bytes) | hist 40
#ifndef ORIENT_WAVES
#define ORIENT_WAVES 4
#endif
__global__ __launch_bounds__(64, ORIENT_WAVES) void orient_kernel(const float *__restrict__ img_all, DescConst k,
                                                    const mods_affkey *__restrict__ keys_all,
                                                    const int *__restrict__ key_count, const float *__restrict__ orimask,
                                                    OriOut *__restrict__ ori_all, OriOut *__restrict__ ori_multi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ps = k.ori_ps, pp2 = ps * ps;
  const int nv = (ps * (ps - 2) + 15) & ~15;
  float *s_patch = smem;
  float *s_val = s_patch + ((pp2 + 3) & ~3);
  unsigned char *s_bin = (unsigned char *)(s_val + nv);
  float *s_hist = (float *)(s_bin + ori_key_bytes(ps) * nv);
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const float *img = img_all + (size_t)k.w * k.h * b;
  const mods_affkey *keys = keys_all + (size_t)b * k.max_cand;
  OriOut *ori = ori_all + (size_t)b * k.max_cand;
  int n = key_count[b];
  if (n > k.max_cand) n = k.max_cand;
#ifdef ORIENT_PROF
  unsigned long long pt[4] = {0, 0, 0, 0}, pl = __builtin_amdgcn_s_memtime();
#define OPROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pl; pl = t_; }
#else
#define OPROF(i)
#endif
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const mods_affkey kp = keys[i];
    // ReprojectRegionsAndRemoveTouchBoundary(dontRemove): centre, in the original frame, strictly inside
    bool alive;
    if (k.view) {
      const double rx = (k.Hinv[0] * kp.x + k.Hinv[1] * kp.y + k.Hinv[2]);
      const double ry = (k.Hinv[3] * kp.x + k.Hinv[4] * kp.y + k.Hinv[5]);
      alive = (rx < k.ow) && (ry < k.oh) && (rx > 0) && (ry > 0);
    } else alive = (kp.x < k.w) && (kp.y < k.h) && (kp.x > 0) && (kp.y > 0);
    const bool inside = alive;   // member of the unoriented ("None") region list of the reference
    const float fx = (float)kp.x, fy = (float)kp.y;
    const float f11 = (float)kp.a11, f12 = (float)kp.a12, f21 = (float)kp.a21, f22 = (float)kp.a22;
    const int box = (int)(k.ks * kp.s);
    if (alive && check_borders(k.w, k.h, fx, fy, f11, f12, f21, f22, box, box)) alive = false;
    double n11 = kp.a11, n12 = kp.a12, n21 = kp.a21, n22 = kp.a22;
    // addUpRight: DetectOrientation(maxAngNum = 0, addUpRight) keeps the unrotated region when the border test above passes;
    // ReprojectRegions then repeats that test on the reprojected frame (for H = I the same test: nothing more to check)
    bool upright = alive && k.add_upright;
    if (upright && k.view) {
      const double rx = (k.Hinv[0] * kp.x + k.Hinv[1] * kp.y + k.Hinv[2]);
      const double ry = (k.Hinv[3] * kp.x + k.Hinv[4] * kp.y + k.Hinv[5]);
      const double r11 = (k.Hinv[0] * n11 + k.Hinv[1] * n21), r12 = (k.Hinv[0] * n12 + k.Hinv[1] * n22);
      const double r21 = (k.Hinv[3] * n11 + k.Hinv[4] * n21), r22 = (k.Hinv[3] * n12 + k.Hinv[4] * n22);
      if (check_borders(k.ow, k.oh, (float)rx, (float)ry, (float)r11, (float)r12, (float)r21, (float)r22, box, box)) upright = false;
    }
    if (alive && k.max_angles <= 0) alive = false;   // DetectOrientation pushes no oriented copy
    if (alive) {
      const float curr_sc = (float)(k.ori_i2p * kp.s);
      const float a11 = f11 * curr_sc, a12 = f12 * curr_sc, a21 = f21 * curr_sc, a22 = f22 * curr_sc;
      __syncthreads();
      OPROF(0)
      // the wave samples the ps x ps patch tile by tile (device_util.hpp: sample_tiles)
      sample_tiles(img, k.w, k.h, fx, fy, a11, a12, a21, a22, ps, 0, 1, [&](int row, int col, float v) { s_patch[row * ps + col] = v; });
      __syncthreads();
      OPROF(1)
      float ang = 0.f;
      unsigned long long peaks = 0;
      const bool found = dominant_angle_wave(s_patch, orimask, ps, k.ori_th, s_val, s_bin, s_hist, &ang, k.ori_half, &peaks);
      OPROF(2)
      // the oriented copy for angle `a`: rotated frame, then ReprojectRegions' tests; false when it is dropped
      auto oriented = [&](float a, double &o11, double &o12, double &o21, double &o22) {
        double si, ci;
        det_sincos(-(double)a, &si, &ci);
        o11 = kp.a11 * ci - kp.a12 * si;
        o12 = kp.a11 * si + kp.a12 * ci;
        o21 = kp.a21 * ci - kp.a22 * si;
        o22 = kp.a21 * si + kp.a22 * ci;
        if (k.view) {
          // ReprojectRegions: ReprojectByH (synth-detection.cpp:578-587) then the centre and box tests in the original frame
          const double rx = (k.Hinv[0] * kp.x + k.Hinv[1] * kp.y + k.Hinv[2]);
          const double ry = (k.Hinv[3] * kp.x + k.Hinv[4] * kp.y + k.Hinv[5]);
          const double r11 = (k.Hinv[0] * o11 + k.Hinv[1] * o21), r12 = (k.Hinv[0] * o12 + k.Hinv[1] * o22);
          const double r21 = (k.Hinv[3] * o11 + k.Hinv[4] * o21), r22 = (k.Hinv[3] * o12 + k.Hinv[4] * o22);
          if (!((rx < k.ow) && (ry < k.oh) && (rx > 0) && (ry > 0))) return false;
          return !check_borders(k.ow, k.oh, (float)rx, (float)ry, (float)r11, (float)r12, (float)r21, (float)r22, box, box);
        }
        // ReprojectRegions (H = I): same centre, rotated frame
        return !check_borders(k.w, k.h, fx, fy, (float)o11, (float)o12, (float)o21, (float)o22, box, box);
      };
      if (k.ori_cap > 1) {
        // maxAngles > 1: the first ori_cap peaks in bin order, one oriented copy each (EstimateDominantAnglesFunctor,
        // synth-detection.cpp:900-927, DetectOrientation :1095-1106); slot j of the keypoint's ori_cap entries
        OriOut *om = ori_multi + ((size_t)b * k.max_cand + i) * k.ori_cap;
        int j = 0;
        for (unsigned long long pm = found ? peaks : 0ull; pm && j < k.ori_cap; pm &= pm - 1, j++) {
          const float a = peak_angle(s_hist, __ffsll((long long)pm) - 1);
          double o11, o12, o21, o22;
          const bool ok = oriented(a, o11, o12, o21, o22);
          if (lane == 0) { OriOut o; o.a11 = o11; o.a12 = o12; o.a21 = o21; o.a22 = o22; o.alive = ok ? 1 : 0; o.pad = 0; om[j] = o; }
        }
        if (lane == 0) for (; j < k.ori_cap; j++) { OriOut o; o.a11 = o.a12 = o.a21 = o.a22 = 0; o.alive = 0; o.pad = 0; om[j] = o; }
        alive = false;                   // the single-copy record below only carries the inside / upright bits
      } else if (!found) alive = false;
      else if (!oriented(ang, n11, n12, n21, n22)) alive = false;
    } else if (k.ori_cap > 1 && lane == 0) {
      OriOut *om = ori_multi + ((size_t)b * k.max_cand + i) * k.ori_cap;
      for (int j = 0; j < k.ori_cap; j++) { OriOut o; o.a11 = o.a12 = o.a21 = o.a22 = 0; o.alive = 0; o.pad = 0; om[j] = o; }
    }
    if (lane == 0) {
      OriOut o;
      o.a11 = n11; o.a12 = n12; o.a21 = n21; o.a22 = n22; o.alive = alive ? 1 : 0; o.pad = (inside ? 1 : 0) | (upright ? 2 : 0);
      ori[i] = o;
    }
    OPROF(3)
  }
#ifdef ORIENT_PROF
  if (lane == 0 && b == 0 && (blockIdx.x % 1024) == 5)
    printf("orient prof: block %d cycles: head %llu sample %llu angle %llu tail %llu\n", blockIdx.x, pt[0], pt[1], pt[2], pt[3]);
#endif
}

// Order-preserving compaction of the surviving keypoints into the region list.
// grid = (ceil(max_cand / 1024), n_img), block = 1024.
__global__ __launch_bounds__(1024) void compact_regions_kernel(DescConst k, const mods_affkey *__restrict__ keys_all,
                                                               const int *__restrict__ key_count,
                                                               const OriOut *__restrict__ ori_all,
                                                               mods_region *__restrict__ reg_all, int *__restrict__ reg_count,
                                                               int *__restrict__ inside_count) {
  // Round 6: a workgroup per 1024 keypoints instead of one per image walking its ~10 chunks with three barriers each (88 us per
  // batch on 16 CUs).  Every workgroup counts for itself what lies before its chunk - the flags of the whole list are 80 KB -,
  // so no workgroup waits for another: upright copies first (addUpRight), the oriented regions behind them, both in list order.
  __shared__ int s_red[16][5];
  __shared__ int s_wave[16][2];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const mods_affkey *keys = keys_all + (size_t)b * k.max_cand;
  const OriOut *ori = ori_all + (size_t)b * k.max_cand;
  mods_region *reg = reg_all + (size_t)b * k.max_reg;
  int n = key_count[b];
  if (n > k.max_cand) n = k.max_cand;
  const int base = blockIdx.x * 1024;
  if (base >= n && blockIdx.x != 0) return;            // (workgroup 0 writes the counts of an empty list)
  const bool last = base + 1024 >= n;
  // [0] upright copies before the chunk, [1] oriented regions before it, [2] upright copies in all, [3] oriented in all, [4] inside
  int c[5] = {0, 0, 0, 0, 0};
  for (int i = tid; i < n; i += 1024) {
    const int al = ori[i].alive != 0 ? 1 : 0, pad = ori[i].pad;
    const int up = (k.add_upright && (pad & 2)) ? 1 : 0;
    if (i < base) { c[0] += up; c[1] += al; }
    c[2] += up; c[3] += al; c[4] += pad & 1;
  }
#pragma unroll
  for (int q = 0; q < 5; q++) {
    for (int off = 32; off > 0; off >>= 1) c[q] += __shfl_xor(c[q], off);
    if (lane == 0) s_red[wv][q] = c[q];
  }
  const int i = base + tid;
  const bool have = i < n;
  const OriOut o = ori[have ? i : 0];
  const bool a_up = have && k.add_upright && (o.pad & 2) != 0, a_al = have && o.alive != 0;
  const unsigned long long m_up = __ballot(a_up), m_al = __ballot(a_al);
  if (lane == 0) { s_wave[wv][0] = __popcll(m_up); s_wave[wv][1] = __popcll(m_al); }
  __syncthreads();
  int tot[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 5; q++)
    for (int w = 0; w < 16; w++) tot[q] += s_red[w][q];
  int off_up = tot[0], off_al = tot[2] + tot[1];
  for (int w = 0; w < wv; w++) { off_up += s_wave[w][0]; off_al += s_wave[w][1]; }
  for (int pass = 0; pass < 2; pass++) {
    const bool alive = pass ? a_al : a_up;
    if (!alive) continue;
    const int slot = pass ? off_al + __popcll(m_al & ((1ull << lane) - 1ull)) : off_up + __popcll(m_up & ((1ull << lane) - 1ull));
    if (slot < k.reg_cap) {
      const mods_affkey kp = keys[i];
      mods_region r;
      r.x = kp.x; r.y = kp.y; r.s = kp.s;
      if (pass) { r.a11 = o.a11; r.a12 = o.a12; r.a21 = o.a21; r.a22 = o.a22; }
      else { r.a11 = kp.a11; r.a12 = kp.a12; r.a21 = kp.a21; r.a22 = kp.a22; }
      r.response = kp.response; r.sub_type = kp.sub_type; r.id = slot; r.parent = i; r.pad = 0;
      // descriptor bytes are written by describe_kernel; copy the POD head only
      memcpy(&reg[slot], &r, offsetof(mods_region, desc));
    }
  }
  if (last && tid == 0) { reg_count[b] = tot[2] + tot[3]; inside_count[b] = tot[4]; }
}

// The same compaction with up to k.ori_cap oriented copies per keypoint (maxAngles > 1): keypoint i contributes the alive ones
// of its ori_cap entries, in angle order, behind the copies of the keypoints before it (DetectOrientation pushes them in that
// order, synth-detection.cpp:1095-1106).  grid = (1, n_img), block = 1024.
__global__ __launch_bounds__(1024) void compact_regions_multi_kernel(DescConst k, const mods_affkey *__restrict__ keys_all,
                                                                     const int *__restrict__ key_count, const OriOut *__restrict__ ori_all,
                                                                     const OriOut *__restrict__ ori_multi_all, mods_region *__restrict__ reg_all,
                                                                     int *__restrict__ reg_count, int *__restrict__ inside_count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  __shared__ int s_inside;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const mods_affkey *keys = keys_all + (size_t)b * k.max_cand;
  const OriOut *ori = ori_all + (size_t)b * k.max_cand;
  const OriOut *om = ori_multi_all + (size_t)b * k.max_cand * k.ori_cap;
  mods_region *reg = reg_all + (size_t)b * k.max_reg;
  int n = key_count[b];
  if (n > k.max_cand) n = k.max_cand;
  if (tid == 0) { s_base = 0; s_inside = 0; }
  __syncthreads();
  for (int pass = k.add_upright ? 0 : 1; pass < 2; pass++) {
    for (int base = 0; base < n; base += 1024) {
      const int i = base + tid;
      int cnt = 0;
      if (i < n) {
        if (pass == 0) cnt = (ori[i].pad & 2) ? 1 : 0;
        else for (int j = 0; j < k.ori_cap; j++) cnt += om[(size_t)i * k.ori_cap + j].alive != 0;
      }
      const unsigned long long mi = __ballot(pass == 1 && i < n && (ori[i].pad & 1));
      int inc = cnt;
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
      if (lane == 63) s_wave[wv] = inc;
      if (lane == 0 && mi) atomicAdd(&s_inside, __popcll(mi));
      __syncthreads();
      int slot = s_base + inc - cnt;
      for (int q = 0; q < wv; q++) slot += s_wave[q];
      if (cnt) {
        const mods_affkey kp = keys[i];
        for (int j = 0; j < (pass ? k.ori_cap : 1); j++) {
          mods_region r;
          if (pass) {
            const OriOut o = om[(size_t)i * k.ori_cap + j];
            if (!o.alive) continue;
            r.a11 = o.a11; r.a12 = o.a12; r.a21 = o.a21; r.a22 = o.a22;
          } else { r.a11 = kp.a11; r.a12 = kp.a12; r.a21 = kp.a21; r.a22 = kp.a22; }
          if (slot < k.reg_cap) {
            r.x = kp.x; r.y = kp.y; r.s = kp.s;
            r.response = kp.response; r.sub_type = kp.sub_type; r.id = slot; r.parent = i; r.pad = 0;
            memcpy(&reg[slot], &r, offsetof(mods_region, desc));
          }
          slot++;
        }
      }
      __syncthreads();
      if (tid == 0) { int t = 0; for (int q = 0; q < 16; q++) t += s_wave[q]; s_base += t; }
      __syncthreads();
    }
  }
  if (tid == 0) { reg_count[b] = s_base; inside_count[b] = s_inside; }
}

// det_kp -> reproj_kp of the described regions of a synthesised view, in place (ReprojectByH: centre and
// frame through the affine part of inv(H); s, response, descriptor unchanged).  grid = (N, n_img), grid-stride per image.
__global__ __launch_bounds__(256) void reproject_regions_kernel(DescConst k, mods_region *__restrict__ reg, const int *__restrict__ reg_count) {
  const int b = blockIdx.y;
  int n = reg_count[b];
  if (n > k.max_reg) n = k.max_reg;
  reg += (size_t)b * k.max_reg;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    mods_region *r = reg + i;
    const double x = r->x, y = r->y, a11 = r->a11, a12 = r->a12, a21 = r->a21, a22 = r->a22;
    r->x = (k.Hinv[0] * x + k.Hinv[1] * y + k.Hinv[2]);
    r->y = (k.Hinv[3] * x + k.Hinv[4] * y + k.Hinv[5]);
    r->a11 = (k.Hinv[0] * a11 + k.Hinv[1] * a21);
    r->a12 = (k.Hinv[0] * a12 + k.Hinv[1] * a22);
    r->a21 = (k.Hinv[3] * a11 + k.Hinv[4] * a21);
    r->a22 = (k.Hinv[3] * a12 + k.Hinv[4] * a22);
  }
}

// fast_sqrtf / fast_sqrtf_any (device_util.hpp) against the compiler's correctly rounded sqrtf over ALL 2^32 operands.
// out[0]: operands in fast_sqrtf's domain (+0, x >= 2^-96, +infinity, NaN) where it differs from sqrtf (bit patterns; any two NaNs
// count as equal), out[1]: operands 0 < x < 2^-96 where it differs (allowed), out[2]: NON-NEGATIVE operands (and NaN) where
// fast_sqrtf_any differs, out[3]: operands visited, out[4]: negative operands where either differs (the callers' operands are sums
// of squares; v_sqrt_f32 takes a negative denormal for -0 where sqrtf says NaN).  grid = 4096 x 256 threads, 4096 operands each.
__global__ __launch_bounds__(256) void fast_sqrt_selftest_kernel(unsigned long long *out) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  unsigned long long bad_dom = 0, bad_tiny = 0, bad_any = 0, bad_neg = 0, seen = 0;
  for (unsigned i = 0; i < 4096u; i++) {
    const unsigned bits = i * (4096u * 256u) + tid;       // consecutive lanes = consecutive operands
    const float x = __uint_as_float(bits);
    const float want = sqrtf(x), a = fast_sqrtf(x), b = fast_sqrtf_any(x);
    const bool same_a = (__float_as_uint(a) == __float_as_uint(want)) || (a != a && want != want);
    const bool same_b = (__float_as_uint(b) == __float_as_uint(want)) || (b != b && want != want);
    const bool negative = (bits >> 31) != 0 && x == x;    // sign bit set, not a NaN
    const bool tiny = x > 0.f && x < 0x1.0p-96f;
    if (negative) { if (!same_a || !same_b) bad_neg++; }
    else {
      if (!same_a) { if (tiny) bad_tiny++; else bad_dom++; }
      if (!same_b) bad_any++;
    }
    seen++;
  }
  atomicAdd(out + 0, bad_dom); atomicAdd(out + 1, bad_tiny); atomicAdd(out + 2, bad_any); atomicAdd(out + 3, seen); atomicAdd(out + 4, bad_neg);
}
int launch_fast_sqrt_selftest(mods_ctx *ctx, unsigned long long *out5_host) {
  unsigned long long *dev = (unsigned long long *)ctx->tmp_dev;
  MODS_HIP_CHECK(hipMemsetAsync(dev, 0, 5 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(fast_sqrt_selftest_kernel, dim3(4096), dim3(256), 0, ctx->stream, dev);
  MODS_HIP_CHECK(hipGetLastError());
  MODS_HIP_CHECK(hipMemcpyAsync(out5_host, dev, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
  return MODS_OK;
}

// single-patch entry points for the parity tests
__global__ __launch_bounds__(64) void dominant_angle_test_kernel(const float *__restrict__ patch, int ps, double th,
                                                                 const float *__restrict__ orimask, float *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nv = (ps * (ps - 2) + 15) & ~15;
  float *s_patch = smem;
  float *s_val = s_patch + ((ps * ps + 3) & ~3);
  unsigned char *s_bin = (unsigned char *)(s_val + nv);
  float *s_hist = (float *)(s_bin + ori_key_bytes(ps) * nv);
  for (int p = threadIdx.x; p < ps * ps; p += 64) s_patch[p] = patch[p];
  __syncthreads();
  float ang = 0.f;
  const bool f = dominant_angle_wave(s_patch, orimask, ps, th, s_val, s_bin, s_hist, &ang);
  if (threadIdx.x == 0) { out[0] = f ? 1.f : 0.f; out[1] = ang; }
}

static size_t orient_lds_bytes(int ps) {   // patch | vote values | vote keys (16 bits) | histogram | list counts, offsets
  const size_t nv = ((size_t)ps * (ps - 2) + 15) & ~(size_t)15;
  return sizeof(float) * ((((size_t)ps * ps + 3) & ~(size_t)3) + nv + 48 + 128 + 128) + (size_t)ori_key_bytes(ps) * nv;   // (+ counts and offsets of the bins' lists, the lane-set words)
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
  if (ctx->ext_fn && (ctx->ext_ps < 8 || ctx->ext_ps > 63)) { set_error("external descriptor: patch size %d unsupported", ctx->ext_ps); return MODS_E_ARG; }
  if (par->ori_patchSize < 8 || par->ori_patchSize > 48 || par->desc_patchSize < 9 || par->desc_patchSize > 63 ||
      !(par->desc_patchSize & 1)) { set_error("unsupported patch sizes (ori %d, desc %d)", par->ori_patchSize, par->desc_patchSize); return MODS_E_ARG; }
  if (!ctx->desc_tables_dev) {
    MODS_HIP_CHECK(hipMalloc(&ctx->desc_tables_dev, sizeof(float) * kTabSift + sizeof(SiftTab)));
    MODS_HIP_CHECK(hipMalloc(&ctx->desc_err_dev, sizeof(int)));
    MODS_HIP_CHECK(hipMemsetAsync(ctx->desc_err_dev, 0, sizeof(int), ctx->stream));
  }
  if (ctx->desc_ori_ps != par->ori_patchSize || ctx->desc_ps != par->desc_patchSize) {
    std::vector<float> m1((size_t)64 * 64, 0.f), m2((size_t)64 * 64, 0.f);
    circular_gauss_mask_host(par->ori_patchSize, par->ori_patchSize / 3.0f, m1.data());   // synth-detection.cpp:852
    circular_gauss_mask_host(par->desc_patchSize, 0.f, m2.data());                         // synth-detection.hpp:181, siftdesc.h:83
    SiftTab tab;
    build_sift_tab(par->desc_patchSize, &tab);
    mods::dev_state_changed(ctx);     // (device tables change: the next detect + describe call is not a repeat - capi.hip: dd_run)
    MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
    MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->desc_tables_dev, m1.data(), sizeof(float) * 4096, hipMemcpyHostToDevice));
    MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->desc_tables_dev + kTabDescMask, m2.data(), sizeof(float) * 4096, hipMemcpyHostToDevice));
    // the orientation mask as the vote loop of dominant_angle_wave reads it: entry p = the mask of pixel (1 + p / ps, p % ps), i.e.
    // of patch index p + ps, and 0 in the first and the last column, where EstimateDominantAnglesFunctor computes no gradient
    {
      const int ps = par->ori_patchSize;
      std::vector<float> vm((size_t)64 * 64, 0.f);
      for (int p = 0; p < ps * (ps - 2); p++) { const int c = p % ps; vm[p] = (c >= 1 && c < ps - 1) ? m1[p + ps] : 0.f; }
      MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->desc_tables_dev + kTabVoteMask, vm.data(), sizeof(float) * 4096, hipMemcpyHostToDevice));
    }
    MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->desc_tables_dev + kTabSift, &tab, sizeof(SiftTab), hipMemcpyHostToDevice));
    ctx->desc_ori_ps = par->ori_patchSize; ctx->desc_ps = par->desc_patchSize;
  }
  return MODS_OK;
}

// Orientation + compaction + description of the keypoints in ctx->keys_dev (as left by detect_run
// or uploaded by mods_orient_describe) for images `img_dev` [n_img][h][w].
int describe_run(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, const mods_describe_params *par) {
  return describe_run_view(ctx, img_dev, n_img, w, h, par, nullptr, 0, 0, nullptr);
}

// Descriptors from outside the library (the ZMQ daemon of the reference's "ZMQ" descriptor, imagerepresentation.cpp:992-1006):
// the extracted patches of every image go to ctx->ext_fn, the 128 returned values per region (integer valued 0..255, as the
// HardNet daemon delivers them) become the region's descriptor bytes.
static int external_describe(mods_ctx *ctx, int n_img, const DescConst &k) {
  const int pp = k.desc_ps * k.desc_ps;
  std::vector<int> counts(n_img);
  MODS_HIP_CHECK(hipMemcpyAsync(counts.data(), ctx->region_count, sizeof(int) * n_img, hipMemcpyDeviceToHost, ctx->stream));
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
  std::vector<float> patches, out;
  std::vector<uint8_t> desc;
  for (int b = 0; b < n_img; b++) {
    const int n = std::min(counts[b], k.reg_cap);
    if (n <= 0) continue;
    patches.resize((size_t)n * pp); out.assign((size_t)n * 128, 0.f); desc.resize((size_t)n * 128);
    MODS_HIP_CHECK(mods::copy_wait(ctx->stream, patches.data(), ctx->desc_scratch + (size_t)b * k.reg_cap * pp, sizeof(float) * patches.size(), hipMemcpyDeviceToHost));
    int dim = 0;
    const int rc = ctx->ext_fn(ctx->ext_user, patches.data(), n, k.desc_ps, out.data(), out.size(), &dim);
    if (rc || dim != 128) { set_error("external descriptor failed (rc %d, %d values per patch; 128 expected)", rc, dim); return MODS_E_ARG; }
    for (size_t i = 0; i < desc.size(); i++) {
      const float v = out[i];
      desc[i] = !(v > 0.f) ? 0 : (v >= 255.f ? 255 : (uint8_t)(v + 0.5f));   // !(v > 0): also NaN replies -> 0
    }
    MODS_HIP_CHECK(hipMemcpy2D((uint8_t *)(ctx->regions_dev + (size_t)b * ctx->max_cand) + offsetof(mods_region, desc), sizeof(mods_region),
                               desc.data(), 128, 128, n, hipMemcpyHostToDevice));
  }
  return MODS_OK;
}

// ---------------------------------------------------------------------------------------
// AffNet / OriNet in the place of Baumberg / the dominant orientation (imagerepresentation.cpp:786-856, 874-900).  The
// networks live behind a callback (the ZMQ client); the patches come from the GPU extraction kernels, the per-keypoint
// arithmetic around the callback (a few thousand keypoints, double precision, libm) runs on the host as in the reference.
// ---------------------------------------------------------------------------------------
// ExtractPatchesColumn patches (mr, ps) of host region lists, one list per image: staged through the region slots
static int net_patches(mods_ctx *ctx, const float *img_dev, int n_img, DescConst k, double mr, int ps,
                       const std::vector<std::vector<mods_region>> &regs, const float *dmask, const SiftTab *tab,
                       std::vector<std::vector<float>> *patches) {
  std::vector<int> counts(n_img);
  for (int b = 0; b < n_img; b++) {
    counts[b] = (int)regs[b].size();
    if (counts[b] > k.reg_cap) { set_error("external shape / orientation: more keypoints than the patch store holds"); return MODS_E_CAPACITY; }
    if (counts[b])
      MODS_HIP_CHECK(hipMemcpyAsync(ctx->regions_dev + (size_t)b * ctx->max_cand, regs[b].data(), sizeof(mods_region) * counts[b],
                                    hipMemcpyHostToDevice, ctx->stream));
  }
  MODS_HIP_CHECK(hipMemcpyAsync(ctx->region_count, counts.data(), sizeof(int) * n_img, hipMemcpyHostToDevice, ctx->stream));
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));   // (the host vectors above are pageable)
  k.desc_mr = mr; k.desc_ps = ps; k.patch_rule = 1; k.photo = 0;
  int rc = launch_extract_and_sift(ctx, img_dev, n_img, k, dmask, tab, false);
  if (rc) return rc;
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
  patches->assign(n_img, std::vector<float>());
  const size_t pp = (size_t)ps * ps;
  for (int b = 0; b < n_img; b++) {
    (*patches)[b].resize(pp * counts[b]);
    if (counts[b])
      MODS_HIP_CHECK(mods::copy_wait(ctx->stream, (*patches)[b].data(), ctx->desc_scratch + (size_t)b * k.reg_cap * pp, sizeof(float) * pp * counts[b], hipMemcpyDeviceToHost));
  }
  return MODS_OK;
}

static int fetch_keys(mods_ctx *ctx, int n_img, const int *key_count, std::vector<std::vector<mods_affkey>> *keys) {
  std::vector<int> counts(n_img);
  MODS_HIP_CHECK(hipMemcpyAsync(counts.data(), key_count, sizeof(int) * n_img, hipMemcpyDeviceToHost, ctx->stream));
  MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
  keys->assign(n_img, std::vector<mods_affkey>());
  for (int b = 0; b < n_img; b++) {
    const int n = std::min(counts[b], ctx->max_cand);
    (*keys)[b].resize(n);
    if (n) MODS_HIP_CHECK(mods::copy_wait(ctx->stream, (*keys)[b].data(), ctx->keys_dev + (size_t)b * ctx->max_cand, sizeof(mods_affkey) * n, hipMemcpyDeviceToHost));
  }
  return MODS_OK;
}

static mods_region region_of_key(const mods_affkey &kp, int id) {
  mods_region r;
  memset(&r, 0, sizeof(r));
  r.x = kp.x; r.y = kp.y; r.s = kp.s; r.a11 = kp.a11; r.a12 = kp.a12; r.a21 = kp.a21; r.a22 = kp.a22;
  r.response = kp.response; r.sub_type = kp.sub_type; r.id = id; r.parent = id;
  return r;
}

// imagerepresentation.cpp:798-842: every keypoint's frame becomes the network's (a11, 0, a21, a22), rectified; keypoints with
// complex or too anisotropic eigenvalues, or whose measurement region touches the image border, are dropped.
static int external_shape(mods_ctx *ctx, const float *img_dev, int n_img, const DescConst &k, int *key_count, const float *dmask,
                          const SiftTab *tab) {
  std::vector<std::vector<mods_affkey>> keys;
  int rc = fetch_keys(ctx, n_img, key_count, &keys);
  if (rc) return rc;
  std::vector<std::vector<mods_region>> regs(n_img);
  for (int b = 0; b < n_img; b++)
    for (size_t i = 0; i < keys[b].size(); i++) regs[b].push_back(region_of_key(keys[b][i], (int)i));
  std::vector<std::vector<float>> patches;
  if ((rc = net_patches(ctx, img_dev, n_img, k, ctx->shape_mr, ctx->shape_ps, regs, dmask, tab, &patches))) return rc;
  std::vector<int> counts(n_img, 0);
  std::vector<float> out;
  for (int b = 0; b < n_img; b++) {
    const int n = (int)keys[b].size();
    std::vector<mods_affkey> kept;
    if (n > 0) {
      out.assign((size_t)n * 3, 0.f);
      int dim = 0;
      const int frc = ctx->shape_fn(ctx->shape_user, patches[b].data(), n, ctx->shape_ps, out.data(), out.size(), &dim);
      if (frc || dim != 3) { set_error("external shape function failed (rc %d, %d values per patch; 3 expected)", frc, dim); return MODS_E_ARG; }
      for (int i = 0; i < n; i++) {
        mods_affkey kp = keys[b][i];
        const double a = out[3 * i], bb = 0, c = out[3 * i + 1], d = out[3 * i + 2];
        // rectifyAffineTransformationUpIsUp (double), helpers.cpp:401-410
        const double det = sqrt(fabs(a * d - bb * c));
        const double b2a2 = sqrt(bb * bb + a * a);
        kp.a11 = b2a2 / det; kp.a12 = 0; kp.a21 = (d * bb + c * a) / (b2a2 * det); kp.a22 = det / b2a2;
        // getEigenvalues, helpers.cpp:504-515
        const float fa = (float)kp.a11, fb = (float)kp.a12, fc = (float)kp.a21, fd = (float)kp.a22;
        const float trace = fa + fd;
        const float delta1 = (trace * trace - 4 * (fa * fd - fb * fc));
        if (delta1 < 0) continue;
        const float delta = sqrtf(delta1);
        const float l1 = (trace + delta) / 2.0f, l2 = (trace - delta) / 2.0f;
        if ((l1 / l2 > 6) || (l2 / l1 > 6)) continue;
        const int box = (int)(ctx->shape_mr * kp.s);
        if (check_borders(k.w, k.h, (float)kp.x, (float)kp.y, fa, fb, fc, fd, box, box)) continue;
        kept.push_back(kp);
      }
    }
    counts[b] = (int)kept.size();
    if (counts[b]) MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->keys_dev + (size_t)b * ctx->max_cand, kept.data(), sizeof(mods_affkey) * counts[b], hipMemcpyHostToDevice));
  }
  MODS_HIP_CHECK(mods::copy_wait(ctx->stream, key_count, counts.data(), sizeof(int) * n_img, hipMemcpyHostToDevice));
  return MODS_OK;
}

// imagerepresentation.cpp:866-900 and what follows it for every detector: centres inside the original image
// (ReprojectRegionsAndRemoveTouchBoundary, dontRemove), frame rotated by atan2(y, x) of the network's two values, then the
// tests of ReprojectRegions.  Fills the region slots, region counts and the count of the unoriented ("None") list.
static int external_orientation(mods_ctx *ctx, const float *img_dev, int n_img, const DescConst &k, const int *key_count,
                                const float *dmask, const SiftTab *tab) {
  std::vector<std::vector<mods_affkey>> keys;
  int rc = fetch_keys(ctx, n_img, key_count, &keys);
  if (rc) return rc;
  std::vector<std::vector<mods_region>> regs(n_img);
  std::vector<int> inside(n_img, 0);
  for (int b = 0; b < n_img; b++)
    for (size_t i = 0; i < keys[b].size(); i++) {
      const mods_affkey &kp = keys[b][i];
      bool alive;
      if (k.view) {
        const double rx = (k.Hinv[0] * kp.x + k.Hinv[1] * kp.y + k.Hinv[2]);
        const double ry = (k.Hinv[3] * kp.x + k.Hinv[4] * kp.y + k.Hinv[5]);
        alive = (rx < k.ow) && (ry < k.oh) && (rx > 0) && (ry > 0);
      } else alive = (kp.x < k.w) && (kp.y < k.h) && (kp.x > 0) && (kp.y > 0);
      if (alive) regs[b].push_back(region_of_key(kp, (int)i));
    }
  for (int b = 0; b < n_img; b++) inside[b] = (int)regs[b].size();
  std::vector<std::vector<float>> patches;
  if ((rc = net_patches(ctx, img_dev, n_img, k, ctx->ori_mr, ctx->ori_ps, regs, dmask, tab, &patches))) return rc;
  std::vector<int> counts(n_img, 0);
  std::vector<float> out;
  for (int b = 0; b < n_img; b++) {
    const int n = (int)regs[b].size();
    std::vector<mods_region> kept;
    if (n > 0) {
      out.assign((size_t)n * 2, 0.f);
      int dim = 0;
      const int frc = ctx->ori_fn(ctx->ori_user, patches[b].data(), n, ctx->ori_ps, out.data(), out.size(), &dim);
      if (frc || dim != 2) { set_error("external orientation function failed (rc %d, %d values per patch; 2 expected)", frc, dim); return MODS_E_ARG; }
      for (int i = 0; i < n; i++) {
        mods_region r = regs[b][i];
        const double angle = atan2f(out[2 * i], out[2 * i + 1]);   // atan2(float, float): the float overload
        double ci, si;
        det_sincos(angle, &si, &ci);   // the contract's cos / sin (detmath.hpp), as in the built-in orientation path
        const double a11 = r.a11, a12 = r.a12, a21 = r.a21, a22 = r.a22;
        r.a11 = a11 * ci - a12 * si;
        r.a12 = a11 * si + a12 * ci;
        r.a21 = a21 * ci - a22 * si;
        r.a22 = a21 * si + a22 * ci;
        const int box = (int)(k.ks * r.s);
        bool alive = true;
        if (k.view) {
          const double rx = (k.Hinv[0] * r.x + k.Hinv[1] * r.y + k.Hinv[2]);
          const double ry = (k.Hinv[3] * r.x + k.Hinv[4] * r.y + k.Hinv[5]);
          const double r11 = (k.Hinv[0] * r.a11 + k.Hinv[1] * r.a21), r12 = (k.Hinv[0] * r.a12 + k.Hinv[1] * r.a22);
          const double r21 = (k.Hinv[3] * r.a11 + k.Hinv[4] * r.a21), r22 = (k.Hinv[3] * r.a12 + k.Hinv[4] * r.a22);
          if (!((rx < k.ow) && (ry < k.oh) && (rx > 0) && (ry > 0))) alive = false;
          else if (check_borders(k.ow, k.oh, (float)rx, (float)ry, (float)r11, (float)r12, (float)r21, (float)r22, box, box)) alive = false;
        } else if (check_borders(k.w, k.h, (float)r.x, (float)r.y, (float)r.a11, (float)r.a12, (float)r.a21, (float)r.a22, box, box)) alive = false;
        if (!alive) continue;
        r.parent = r.id; r.id = (int)kept.size();
        kept.push_back(r);
      }
    }
    counts[b] = (int)kept.size();
    if (counts[b] > k.reg_cap) counts[b] = k.reg_cap;
    if (counts[b]) MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->regions_dev + (size_t)b * ctx->max_cand, kept.data(), sizeof(mods_region) * counts[b], hipMemcpyHostToDevice));
  }
  MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->region_count, counts.data(), sizeof(int) * n_img, hipMemcpyHostToDevice));
  MODS_HIP_CHECK(mods::copy_wait(ctx->stream, ctx->inside_count, inside.data(), sizeof(int) * n_img, hipMemcpyHostToDevice));
  return MODS_OK;
}

// cv::invert(H, Hinv, DECOMP_LU) for 3x3 doubles: OpenCV's closed form (all zeros when singular)
static void invert3_cv(const double *S, double *t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.) { for (int i = 0; i < 9; i++) t[i] = 0; return; }
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// The same for a synthesised view: H (original -> view, 9 doubles) and the original image size; the
// regions come out in the original frame (reproj_kp).  H == nullptr or HIsEye(H) (synth-detection.cpp:
// 144-149): identity view.  det_copy_dev (optional): receives the described regions in the view frame.
int describe_run_view(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, const mods_describe_params *par, const double *H,
                      int orig_w, int orig_h, mods_region *det_copy_dev) {
  int rc = describe_configure(ctx, par);
  if (rc) return rc;
  DescConst k;
  k.view = 0; k.ow = w; k.oh = h;
  for (int i = 0; i < 6; i++) k.Hinv[i] = (i == 0 || i == 4) ? 1.0 : 0.0;
  if (H) {
    const bool eye = (std::fabs(H[0] - 1.0) + std::fabs(H[1]) + std::fabs(H[2]) + std::fabs(H[3]) + std::fabs(H[4] - 1.0) + std::fabs(H[5]) +
                      std::fabs(H[6]) + std::fabs(H[7]) + std::fabs(H[8] - 1.0) < 0.01);
    // (n_img > 1: the same view of several images of one size - one H for all of them)
    k.ow = orig_w; k.oh = orig_h;
    if (!eye) {
      double Hi[9];
      invert3_cv(H, Hi);
      k.view = 1;
      for (int i = 0; i < 6; i++) k.Hinv[i] = Hi[i];
    } else if (orig_w != w || orig_h != h) k.view = 1;   // identity map onto a different canvas: the tests still use (ow, oh)
  }
  k.w = w; k.h = h; k.max_cand = ctx->max_cand; k.max_reg = ctx->max_cand;
  k.reg_cap = std::min(ctx->max_cand, 1 << 17);
  k.ks = 2 * 3.0 * sqrt(3.0);
  k.ori_ps = par->ori_patchSize;
  k.ori_i2p = double(2 * int(par->ori_mrSize) + 1) / (double)par->ori_patchSize;
  k.max_angles = par->ori_maxAngles;
  // a 36-bin histogram has at most 18 local maxima; maxAngles <= 0 (the functor's -1 = "all" included): DetectOrientation
  // estimates nothing (`if (maxAngNum > 0)`, synth-detection.cpp:1086)
  k.ori_cap = std::max(1, std::min(par->ori_maxAngles, 18));
  k.ori_th = par->ori_threshold;
  k.ori_half = par->ori_halfMode; k.half_desc = 0; k.add_upright = par->addUpRight;
  k.desc_mr = par->desc_mrSize; k.desc_ps = par->desc_patchSize; k.photo = par->photoNorm; k.root = par->rootSift;
  k.max_bin = par->maxBinValue;
  k.patch_rule = par->fastExtraction ? 2 : 0;
  const bool external = ctx->ext_fn != nullptr;
  if (external && par->fastExtraction) { set_error("FastPatchExtraction with an external descriptor is not supported"); return MODS_E_ARG; }
  if (external) { k.desc_mr = ctx->ext_mr; k.desc_ps = ctx->ext_ps; k.photo = 0; k.patch_rule = 1; }
  int *key_count = ctx->cand_count + 2 * ctx->batch;
  const float *orimask = ctx->desc_tables_dev + kTabVoteMask, *dmask = ctx->desc_tables_dev + kTabDescMask;
  const SiftTab *tab = (const SiftTab *)(ctx->desc_tables_dev + kTabSift);
  // doExternalAffineAdaptation is set in the HessianAffine branch of the detector dispatch only (imagerepresentation.cpp:733-737):
  // the regions of DoG / HarrisAffine / MSER views keep their own frames
  if (ctx->shape_fn && ctx->par.detectorType == MODS_DET_HESSIAN && (rc = external_shape(ctx, img_dev, n_img, k, key_count, dmask, tab))) return rc;
  if (ctx->ori_fn) {
    if ((rc = external_orientation(ctx, img_dev, n_img, k, key_count, dmask, tab))) return rc;
  } else {
    StageScope ts(ctx, MODS_STAGE_ORIENT);
    const size_t lds = orient_lds_bytes(k.ori_ps);
    hipLaunchKernelGGL(ori_bin_table_kernel, dim3(8), dim3(256), 0, ctx->stream);
    if (k.ori_cap > 1) {     // room for ori_cap oriented copies per keypoint
      const size_t need = sizeof(OriOut) * (size_t)k.ori_cap * ctx->max_cand * ctx->batch;
      if (need > ctx->ori_multi_bytes) {
        if (ctx->ori_multi_dev) MODS_HIP_CHECK(hipFree(ctx->ori_multi_dev));
        ctx->ori_multi_dev = nullptr; ctx->ori_multi_bytes = 0;
        mods::dev_pool_reallocated(ctx); MODS_HIP_CHECK(hipMalloc(&ctx->ori_multi_dev, need));
        ctx->ori_multi_bytes = need;
      }
    }
    hipLaunchKernelGGL(orient_kernel, dim3(8192, n_img), dim3(64), lds, ctx->stream, img_dev, k, ctx->keys_dev, key_count,
                       orimask, (OriOut *)ctx->ori_dev, (OriOut *)ctx->ori_multi_dev);
    if (k.ori_cap > 1)
      hipLaunchKernelGGL(compact_regions_multi_kernel, dim3(1, n_img), dim3(1024), 0, ctx->stream, k, ctx->keys_dev, key_count,
                         (const OriOut *)ctx->ori_dev, (const OriOut *)ctx->ori_multi_dev, ctx->regions_dev, ctx->region_count, ctx->inside_count);
    else
      hipLaunchKernelGGL(compact_regions_kernel, dim3((ctx->max_cand + 1023) / 1024, n_img), dim3(1024), 0, ctx->stream, k, ctx->keys_dev, key_count,
                         (const OriOut *)ctx->ori_dev, ctx->regions_dev, ctx->region_count, ctx->inside_count);
    MODS_HIP_CHECK(hipGetLastError());
  }
  rc = launch_extract_and_sift(ctx, img_dev, n_img, k, dmask, tab, !external);
  if (rc) return rc;
  ctx->have_half = false;
  if (par->halfDesc && !external) {
    if ((rc = launch_half_sift(ctx, n_img, k, dmask, tab))) return rc;
    ctx->have_half = true;
  }
  if (external && (rc = external_describe(ctx, n_img, k))) return rc;
  if (det_copy_dev)
    MODS_HIP_CHECK(hipMemcpyAsync(det_copy_dev, ctx->regions_dev, sizeof(mods_region) * (size_t)ctx->max_cand, hipMemcpyDeviceToDevice, ctx->stream));
  if (k.view) {
    hipLaunchKernelGGL(reproject_regions_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, k, ctx->regions_dev, ctx->region_count);
    if (ctx->have_half)
      hipLaunchKernelGGL(reproject_regions_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, k, ctx->regions_half_dev, ctx->region_count);
    MODS_HIP_CHECK(hipGetLastError());
  }
  return MODS_OK;
}

int launch_dominant_angle_test(mods_ctx *ctx, const float *patch_dev, int ps, double th, float *out_dev) {
  const size_t lds = orient_lds_bytes(ps);
  hipLaunchKernelGGL(ori_bin_table_kernel, dim3(8), dim3(256), 0, ctx->stream);
  hipLaunchKernelGGL(dominant_angle_test_kernel, dim3(1), dim3(64), lds, ctx->stream, patch_dev, ps, th,
                     ctx->desc_tables_dev + kTabVoteMask, out_dev);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods
