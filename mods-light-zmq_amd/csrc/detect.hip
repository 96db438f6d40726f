// Keypoint extraction on the scale space: 3x3x3 NMS, sub-pixel/scale localisation, octaveMap
// de-duplication, Baumberg affine-shape iteration, |response| ordering and export.
//
// Reference behaviour (file:line relative to the reference root):
//   findLevelKeypoints / isMax / isMin   detectors/affinedetectors/pyramid.cpp:405-425, 41-63
//   localizeKeypoint                     pyramid.cpp:281-403   (thresholds: pyramid.h:46-66)
//   findAffineShape (SMM branch)         detectors/affinedetectors/affine.cpp:26-158
//   interpolate / computeGradient / invSqrt / getEigenvalues / solveLinear3x3
//                                        detectors/helpers.cpp:551-626, 779-797, 463-515, 309-368
//   exportKeypoints + DetectAffineRegions scale-space-detector.hpp:89-131, synth-detection.hpp:79-112
//
// The reference walks NMS hits in raster order per level and lets the first accepted point
// claim its octaveMap cell.  Here all hits are processed in parallel and the claim is resolved
// with an atomicMin over the processing-order key (level, r0, c0) of the points that pass
// every other test: the winner is exactly the reference's first claimant.  The output order
// (|response| descending, ties in processing order) is produced by a rank sort at the end, so
// nothing upstream depends on the order hits were appended in.
#include "common.hpp"
#include "detmath.hpp"
#include "device_util.hpp"

namespace mods {

constexpr int ORDER_POS_BITS = 25;   // r0*w+c0 < 2^25 per octave

struct DetectConst {
  int border;
  int n_scales;
  float pos_th, neg_th, final_th;
  double edge_th;
  int max_cand;
  int smm;                // smmWindowSize
  int max_iter;
  float conv_th;
  float initial_sigma;
  int do_baumberg;
  int det_type;           // MODS_DET_HESSIAN / DOG / HARRIS
  const float *sfi_img;   // sampleFromImage: the batch's input images [n_img][sfi_h][sfi_w], else null
  int sfi_w, sfi_h;
  float aff_meas_region;  // AffineShapeParams::affMeasRegion = 0.5 (affine.h:64; no key reads it)
};

// ---------------------------------------------------------------------------------------
// NMS for all detection levels of one octave: a thread walks NMS_ROWS rows of one pixel column and, per
// pixel, the S levels.  Hits are appended with one atomic per wave (order is irrelevant, see header).
// grid = (ceil((w-2b)/64), ceil((h-2b)/(4*NMS_ROWS)), n_img), block = 256
// ---------------------------------------------------------------------------------------
constexpr int NMS_ROWS = 8;

// One launch covers every octave of a kind (the small octaves are dependent load -> test -> gather chains of ~20 us whatever
// their size: side by side they cost one chain, not one per octave).  blockIdx.x runs over the octaves' blocks, octave after octave.
struct NmsPlan {
  int n;
  int oi[kMaxOctaves];
  int blk_begin[kMaxOctaves + 1];
  int nbx[kMaxOctaves];                       // blocks per block-row of the octave
  int w[kMaxOctaves], h[kMaxOctaves], words[kMaxOctaves], wide[kMaxOctaves];
  unsigned long long mask_off[kMaxOctaves];   // where the octave's ballot words start
};
__device__ __forceinline__ int nms_plan_entry(const NmsPlan &pl, int blk) {
  int e = 0;
  while (e + 1 < pl.n && blk >= pl.blk_begin[e + 1]) e++;
  return e;
}

// low/high planes of a pixel that already is an in-plane extremum above the gate: all 18 loads issued
// together (no short-circuit chain of dependent loads)
typedef float nms_f3 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ bool nms_other_planes(const float *__restrict__ low, const float *__restrict__ high, int w, int r, int c,
                                                 float val, bool want_max) {
#ifdef NMS_EXP_NOFLUSH
  return true;
#endif
  // three 12-byte loads per plane (dword alignment is enough for global_load_dwordx3) instead of nine 4-byte ones: a third of
  // the vector-memory instructions over the same six cache lines
  nms_f3 v[6];
#pragma unroll
  for (int dr = -1; dr <= 1; dr++) {
    const size_t off = (size_t)(r + dr) * w + c - 1;
    v[(dr + 1) * 2] = *(const nms_f3 *)(low + off);
    v[(dr + 1) * 2 + 1] = *(const nms_f3 *)(high + off);
  }
  bool ok = true;
#pragma unroll
  for (int q = 0; q < 6; q++) {
    ok = ok && !(want_max ? (v[q].x > val) : (v[q].x < val));
    ok = ok && !(want_max ? (v[q].y > val) : (v[q].y < val));
    ok = ok && !(want_max ? (v[q].z > val) : (v[q].z < val));
  }
  return ok;
}

// Hits are written as ballot words (one u64 per wave-row per level: no atomics); nms_compact_kernel
// turns the words into the hit list with one atomic per block.
// A wave covers NMS_COLS = 62 pixel columns: lanes 1..62 answer for a column each, lanes 0 and 63 only carry the columns beside
// them, so that every lane loads ONE value per row and takes its left / right neighbours from the adjacent lanes (the
// earlier form loaded three values per row and lane: the kernel is bound by the number of lane-loads, not by bytes).
// mask layout per octave: [n_img][S][h - 2*border][words], words = ceil((w - 2*border) / 62), bit = lane
constexpr int NMS_COLS = 62;
// s_code / s_val: the calling wave's list (64 * NMS_ROWS entries), s_hit: its NMS_ROWS hit words
__device__ __forceinline__ void nms_narrow_body(const PyramidDev *__restrict__ P, const NmsPlan &pl, int pe, const DetectConst &k,
                                                unsigned long long *__restrict__ mask, unsigned int *s_code, float *s_val,
                                                unsigned long long *s_hit) {
  const OctaveDev &o = P->oct[pl.oi[pe]];
  const int bx = (blockIdx.x - pl.blk_begin[pe]) % pl.nbx[pe], by = (blockIdx.x - pl.blk_begin[pe]) / pl.nbx[pe];
  mask += pl.mask_off[pe];
  const int w = o.w, h = o.h;
  const int b = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int c = k.border + bx * NMS_COLS - 1 + lane;
  const int r_base = k.border + (by * 4 + (threadIdx.x >> 6)) * NMS_ROWS;
  const size_t plane = (size_t)w * h * b;
  const bool col_ok = lane >= 1 && lane <= NMS_COLS && c < w - k.border;
  const int cc = c < w - 1 ? c : w - 1;      // lanes past the row read a valid column (c >= border - 1 >= 1)
  if (r_base >= h - k.border) return;
  const int ih = h - 2 * k.border, words = pl.nbx[pe];
  // in-plane extrema above the gate are rare per lane but not per wave: they are collected into a wave-private LDS list and
  // the 18 loads of the other two planes run over the list with all lanes busy (instead of once per row for a few lanes)
  for (int lv = 1; lv <= k.n_scales; lv++) {
    const float *cur = as_global(o.resp[lv]) + plane;
    // the (NMS_ROWS+2)-row column of this lane is loaded up front (independent loads in flight), the columns beside it come
    // from the neighbour lanes
    float win[NMS_ROWS + 2][3];
#pragma unroll
    for (int rr = 0; rr < NMS_ROWS + 2; rr++) {
      int r = r_base - 1 + rr;
      r = r < h - 1 ? r : h - 1;             // rows past the image are never used (clamped to stay in bounds)
      win[rr][1] = cur[(size_t)r * w + cc];
    }
#pragma unroll
    for (int rr = 0; rr < NMS_ROWS + 2; rr++) {
      win[rr][0] = lane_up1(win[rr][1]);
      win[rr][2] = lane_down1(win[rr][1]);
    }
    wave_sync();                              // the previous level's list and hit words have been consumed
    if (lane < NMS_ROWS) s_hit[lane] = 0ull;
    int n_c = 0;
#pragma unroll
    for (int rr = 0; rr < NMS_ROWS; rr++) {
      const int r = r_base + rr;
      const bool row_ok = r < h - k.border;  // uniform per wave
      const float val = win[rr + 1][1];
      // isMax / isMin on the level's own plane (pyramid.cpp:41-63: a neighbour strictly beyond val rejects)
      bool mx = true, mn = true;
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 3; e++) {
          if (q == 1 && e == 1) continue;
          mx = mx && !(win[rr + q][e] > val);
          mn = mn && !(win[rr + q][e] < val);
        }
      const bool cmax = row_ok && col_ok && val > k.pos_th && mx;
      const bool cmin = row_ok && col_ok && !cmax && val < k.neg_th && mn;
      const bool cnd = cmax || cmin;
      const unsigned long long m = __ballot(cnd);
      if (cnd) {
        const int pos = n_c + __popcll(m & ((1ull << lane) - 1ull));
        s_code[pos] = ((unsigned int)rr << 7) | ((unsigned int)lane << 1) | (cmax ? 1u : 0u);
        s_val[pos] = val;
      }
      n_c += __popcll(m);
    }
    wave_sync();
    const float *low = as_global(o.resp[lv - 1]) + plane, *high = as_global(o.resp[lv + 1]) + plane;
    for (int t0 = 0; t0 < n_c; t0 += 64) {
      const int t = t0 + lane;
      if (t < n_c) {
        const unsigned int code = s_code[t];
        const int rr = code >> 7, ln = (code >> 1) & 63;
        if (nms_other_planes(low, high, w, r_base + rr, k.border + bx * NMS_COLS - 1 + ln, s_val[t], (code & 1u) != 0))
          atomicOr(&s_hit[rr], 1ull << ln);
      }
    }
    wave_sync();
    unsigned long long *mrow = mask + (((size_t)b * k.n_scales + (lv - 1)) * ih + (r_base - k.border)) * words + bx;
    if (lane < NMS_ROWS && r_base + lane < h - k.border) mrow[(size_t)lane * words] = s_hit[lane];
  }
}

__global__ __launch_bounds__(256) void nms_kernel(const PyramidDev *__restrict__ P, NmsPlan pl, DetectConst k,
                                                  unsigned long long *__restrict__ mask) {
  __shared__ unsigned int s_code[4][64 * NMS_ROWS];
  __shared__ float s_val[4][64 * NMS_ROWS];
  __shared__ unsigned long long s_hit[4][NMS_ROWS];
  const int wv = threadIdx.x >> 6;
  nms_narrow_body(P, pl, nms_plan_entry(pl, blockIdx.x), k, mask, s_code[wv], s_val[wv], s_hit[wv]);
}

// The same NMS for planes whose width is a multiple of 4: a lane answers for FOUR adjacent columns and loads them as one
// aligned 16-byte value per row (a quarter of the load instructions per byte; these kernels are bound by the number of
// vector-memory instructions, not by bytes), a wave covers 62 x 4 = 248 columns x NMS_ROWS rows (lanes 0 and 63 only carry the
// columns beside the block).  In-plane test: isMax <=> val equals the maximum of its 3x3 block (no neighbour strictly larger,
// pyramid.cpp:41-51), so the six column maxima of a lane's 3-row window are taken once (v_max3) and every pixel needs one more
// v_max3; likewise for isMin.
// mask layout per octave: [n_img][S][h - 2*border][words], words = 4 * ceil(w / 248): word (block, sub-column), bit = lane
constexpr int NMS4_COLS = 248;
constexpr int NMS4_CAP = 1024;     // in-plane extrema listed per wave before the other planes are consulted
__global__ __launch_bounds__(256) void nms4_kernel(const PyramidDev *__restrict__ P, NmsPlan pl, DetectConst k,
                                                   unsigned long long *__restrict__ mask) {
  __shared__ unsigned int s_code[4][NMS4_CAP];
  __shared__ float s_val[4][NMS4_CAP];
  __shared__ unsigned long long s_hit[4][NMS_ROWS][4];
  const int pe = nms_plan_entry(pl, blockIdx.x);
  if (!pl.wide[pe]) {     // a few small octaves whose rows are not float4-aligned ride along in the same launch (same LDS arrays)
    const int wq = threadIdx.x >> 6;
    nms_narrow_body(P, pl, pe, k, mask, s_code[wq], s_val[wq], &s_hit[wq][0][0]);
    return;
  }
  const OctaveDev &o = P->oct[pl.oi[pe]];
  const int bx = (blockIdx.x - pl.blk_begin[pe]) % pl.nbx[pe], by = (blockIdx.x - pl.blk_begin[pe]) / pl.nbx[pe];
  mask += pl.mask_off[pe];
  const int w = o.w, h = o.h;
  const int b = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c0 = bx * NMS4_COLS - 4 + 4 * lane;                       // first of the lane's four columns (multiple of 4)
  const int r_base = k.border + (by * 4 + wv) * NMS_ROWS;
  const size_t plane = (size_t)w * h * b;
  if (r_base >= h - k.border) return;
  const int ih = h - 2 * k.border, words = 4 * pl.nbx[pe];
  const int cl = c0 < 0 ? 0 : (c0 > w - 4 ? w - 4 : c0);              // lanes beside the row read a valid address (values unused)
  const bool lane_ok = lane >= 1 && lane <= 62;
#ifndef NMS4_PREFETCH
#define NMS4_PREFETCH 0
#endif
  // NMS4_PREFETCH = 1: the rows of a level's own plane are requested while the level before it is worked on (the kernel is bound by
  // the round trips of a wave - own rows, then the candidates' neighbour planes, three levels in sequence -, not by bytes: 2.4 TB/s).
  // Round 5 measured it both ways: the NMS stage with per-launch scopes 0.338 -> 0.295 ms, the kernel under rocprofv3 119 -> 132 us
  // per launch (162 registers: 3 waves per SIMD instead of 4), the whole scale space unchanged.  Left off.
  unsigned rowoff[NMS_ROWS + 2];
#pragma unroll
  for (int rr = 0; rr < NMS_ROWS + 2; rr++) {
    int r = r_base - 1 + rr;
    r = r < h - 1 ? r : h - 1;               // rows past the image are never used (clamped to stay in bounds)
    rowoff[rr] = (unsigned)r * (unsigned)w + (unsigned)cl;
  }
  float4 nxt[NMS_ROWS + 2];
  if (NMS4_PREFETCH) {
    const float *first = as_global(o.resp[1]) + plane;
#pragma unroll
    for (int rr = 0; rr < NMS_ROWS + 2; rr++) nxt[rr] = *(const float4 *)(first + rowoff[rr]);
  }
  for (int lv = 1; lv <= k.n_scales; lv++) {
    const float *cur = as_global(o.resp[lv]) + plane;
    const float *low = as_global(o.resp[lv - 1]) + plane, *high = as_global(o.resp[lv + 1]) + plane;
    float4 own[NMS_ROWS + 2];
    if (NMS4_PREFETCH) {
#pragma unroll
      for (int rr = 0; rr < NMS_ROWS + 2; rr++) own[rr] = nxt[rr];
      if (lv < k.n_scales) {
#pragma unroll
        for (int rr = 0; rr < NMS_ROWS + 2; rr++) nxt[rr] = *(const float4 *)(high + rowoff[rr]);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < NMS_ROWS + 2; rr++) own[rr] = *(const float4 *)(cur + rowoff[rr]);
    }
    wave_sync();                              // the previous level's list and hit words have been consumed
    if (lane < NMS_ROWS * 4) s_hit[wv][lane >> 2][lane & 3] = 0ull;
    wave_sync();
    int n_c = 0;
    auto flush = [&]() {                      // the listed in-plane extrema against the other two planes (18 loads each)
      wave_sync();
      for (int t0 = 0; t0 < n_c; t0 += 64) {
        const int t = t0 + lane;
        if (t < n_c) {
          const unsigned int code = s_code[wv][t];
          const int rr = code >> 9, ln = (code >> 3) & 63, sub = (code >> 1) & 3;
          const int c = bx * NMS4_COLS - 4 + 4 * ln + sub;
          if (nms_other_planes(low, high, w, r_base + rr, c, s_val[wv][t], (code & 1u) != 0)) atomicOr(&s_hit[wv][rr][sub], 1ull << ln);
        }
      }
      wave_sync();
      n_c = 0;
    };
#pragma unroll
    for (int rr = 0; rr < NMS_ROWS; rr++) {
      const int r = r_base + rr;
      const bool row_ok = r < h - k.border;  // uniform per wave
      if (n_c > NMS4_CAP - 4 * 64) flush();  // uniform per wave; keeps the list within its capacity whatever the plane holds
      // the lane's 3 x 6 window: its own four columns of rows rr .. rr+2 and the columns beside them from the neighbour lanes
      float cmx[6], cmn[6];
      {
        const float4 a = own[rr], m = own[rr + 1], z = own[rr + 2];
        cmx[1] = fmaxf(fmaxf(a.x, m.x), z.x); cmn[1] = fminf(fminf(a.x, m.x), z.x);
        cmx[2] = fmaxf(fmaxf(a.y, m.y), z.y); cmn[2] = fminf(fminf(a.y, m.y), z.y);
        cmx[3] = fmaxf(fmaxf(a.z, m.z), z.z); cmn[3] = fminf(fminf(a.z, m.z), z.z);
        cmx[4] = fmaxf(fmaxf(a.w, m.w), z.w); cmn[4] = fminf(fminf(a.w, m.w), z.w);
        cmx[0] = lane_up1(cmx[4]); cmn[0] = lane_up1(cmn[4]);
        cmx[5] = lane_down1(cmx[1]); cmn[5] = lane_down1(cmn[1]);
      }
      const float vals[4] = {own[rr + 1].x, own[rr + 1].y, own[rr + 1].z, own[rr + 1].w};
#pragma unroll
      for (int sub = 0; sub < 4; sub++) {
        const float val = vals[sub];
        const int c = c0 + sub;
        const bool col_ok = lane_ok && c >= k.border && c < w - k.border;
        const float m9 = fmaxf(fmaxf(cmx[sub], cmx[sub + 1]), cmx[sub + 2]);
        const float n9 = fminf(fminf(cmn[sub], cmn[sub + 1]), cmn[sub + 2]);
        const bool cmax = row_ok && col_ok && val > k.pos_th && !(m9 > val);
        const bool cmin = row_ok && col_ok && !cmax && val < k.neg_th && !(n9 < val);
        const bool cnd = cmax || cmin;
        const unsigned long long m = __ballot(cnd);
        if (cnd) {
          const int pos = n_c + __popcll(m & ((1ull << lane) - 1ull));
          s_code[wv][pos] = ((unsigned int)rr << 9) | ((unsigned int)lane << 3) | ((unsigned int)sub << 1) | (cmax ? 1u : 0u);
          s_val[wv][pos] = val;
        }
        n_c += __popcll(m);
      }
    }
    flush();
    unsigned long long *mrow = mask + (((size_t)b * k.n_scales + (lv - 1)) * ih + (r_base - k.border)) * words + 4 * bx;
    if (lane < NMS_ROWS * 4 && r_base + (lane >> 2) < h - k.border) mrow[(size_t)(lane >> 2) * words + (lane & 3)] = s_hit[wv][lane >> 2][lane & 3];
  }
}

// grid = (sum over octaves of ceil(total_words / (256 * NMS_CW)), n_img), block 256: ballot words -> hit records.  A thread
// takes NMS_CW words (coalesced, 256 apart), so a block owns 2048 words and draws its list slots with ONE returning atomic:
// at one word per thread the kernel was bound by ~540 serialised atomics per image counter (105 us per 16-image batch).
constexpr int NMS_CW = 8;
__global__ __launch_bounds__(256) void nms_compact_kernel(NmsPlan pl, DetectConst k, const unsigned long long *__restrict__ mask,
                                                          CandDev *__restrict__ cand, int *__restrict__ cand_count) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int pe = nms_plan_entry(pl, blockIdx.x);
  const int oi = pl.oi[pe], h = pl.h[pe], words = pl.words[pe], wide = pl.wide[pe];
  mask += pl.mask_off[pe];
  const int b = blockIdx.y;
  const int ih = h - 2 * k.border;
  const int total = k.n_scales * ih * words;
  const int idx0 = (blockIdx.x - pl.blk_begin[pe]) * 256 * NMS_CW + threadIdx.x;
  unsigned long long m[NMS_CW];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < NMS_CW; j++) {
    const int idx = idx0 + j * 256;
    m[j] = idx < total ? mask[(size_t)b * total + idx] : 0ull;
    cnt += __popcll(m[j]);
  }
  // block-wide exclusive prefix of cnt
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = cnt;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot ? atomicAdd(&cand_count[b], tot) : 0;
  }
  __syncthreads();
  if (cnt == 0) return;
  int slot = s_base + inc - cnt;
  for (int q = 0; q < wv; q++) slot += s_wave[q];
#pragma unroll
  for (int j = 0; j < NMS_CW; j++) {
    unsigned long long mm = m[j];
    if (!mm) continue;
    const int idx = idx0 + j * 256;
    const int lv = idx / (ih * words) + 1;
    const int rem = idx - (lv - 1) * ih * words;
    const int r = k.border + rem / words;
    // column of bit `bit`: nms_kernel: border + block * 62 - 1 + bit; nms4_kernel (wide): block * 248 - 4 + 4 * bit + sub-column
    const int wi = rem % words;
    const int c0 = wide ? (wi >> 2) * NMS4_COLS - 4 + (wi & 3) : k.border + wi * NMS_COLS - 1;
    const int cstep = wide ? 4 : 1;
    while (mm) {
      const int bit = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      if (slot < k.max_cand) {
        CandDev &cd = cand[(size_t)b * k.max_cand + slot];
        cd.octave = oi; cd.level = lv; cd.r0 = r; cd.c0 = c0 + cstep * bit; cd.state = 0;
      }
      slot++;
    }
  }
}

__device__ __forceinline__ void swapf(float &a, float &b) { float t = a; a = b; b = t; }

// solveLinear3x3, helpers.cpp:309-368 (partial pivoting, fp32)
__device__ void solve_linear_3x3(float *A, float *b) {
  int i = 0;
  int pr = 0;
  float vp = fabsf(A[0]);
  float tmp = fabsf(A[3]);
  if (tmp > vp) { pr = 3; i = 1; vp = tmp; }
  if (fabsf(A[6]) > vp) { pr = 6; i = 2; }
  if (pr != 0) {
    swapf(A[pr], A[0]); swapf(A[pr + 1], A[1]); swapf(A[pr + 2], A[2]); swapf(b[i], b[0]);
  }
  vp = A[3] / A[0];
  A[4] -= vp * A[1]; A[5] -= vp * A[2]; b[1] -= vp * b[0];
  vp = A[6] / A[0];
  A[7] -= vp * A[1]; A[8] -= vp * A[2]; b[2] -= vp * b[0];
  if (fabsf(A[4]) < fabsf(A[7])) { swapf(A[7], A[4]); swapf(A[8], A[5]); swapf(b[2], b[1]); }
  vp = A[7] / A[4];
  A[8] -= vp * A[5];
  b[2] -= vp * b[1];
  b[2] = (b[2]) / A[8];
  b[1] = (b[1] - A[5] * b[2]) / A[4];
  b[0] = (b[0] - A[2] * b[2] - A[1] * b[1]) / A[0];
}

// localizeKeypoint (pyramid.cpp:281-403): one thread per NMS hit, grid-stride over the hit list.
// Points that pass every test claim their cell with atomicMin(order key).
__global__ __launch_bounds__(256) void localize_kernel(const PyramidDev *__restrict__ P, DetectConst k,
                                                       CandDev *__restrict__ cand, const int *__restrict__ cand_count) {
  const int b = blockIdx.y;
  int n = cand_count[b];
  if (n > k.max_cand) n = k.max_cand;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    CandDev &cd = cand[(size_t)b * k.max_cand + i];
    const OctaveDev &o = P->oct[cd.octave];
    const int cols = o.w, rows = o.h;
    const size_t plane = (size_t)cols * rows * b;
    const float *low = as_global(o.resp[cd.level - 1]) + plane;
    const float *cur = as_global(o.resp[cd.level]) + plane;
    const float *high = as_global(o.resp[cd.level + 1]) + plane;
    int r = cd.r0, c = cd.c0;
    float bb[3] = {0, 0, 0};
    float val = 0;
    int nr = r, nc = c;
    bool ok = true;
    for (int iter = 0; iter < 5; iter++) {
      r = nr; c = nc;
      const float *cur0 = cur + (size_t)(r - 1) * cols, *cur1 = cur0 + cols, *cur2 = cur1 + cols;
      const float *low0 = low + (size_t)(r - 1) * cols, *low1 = low0 + cols, *low2 = low1 + cols;
      const float *high0 = high + (size_t)(r - 1) * cols, *high1 = high0 + cols, *high2 = high1 + cols;
      float dxx = cur1[c - 1] - 2.0f * cur1[c] + cur1[c + 1];
      float dyy = cur0[c] - 2.0f * cur1[c] + cur2[c];
      float dss = low1[c] - 2.0f * cur1[c] + high1[c];
      float dxy = 0.25f * (cur2[c + 1] - cur2[c - 1] - cur0[c + 1] + cur0[c - 1]);
      if (0 == iter) {
        float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
        if ((double)edgeScore >= k.edge_th || edgeScore < 0) { ok = false; break; }
      }
      float dxs = 0.25f * (high1[c + 1] - high1[c - 1] - low1[c + 1] + low1[c - 1]);
      float dys = 0.25f * (high2[c] - high0[c] - low2[c] + low0[c]);
      float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
      float dx = 0.5f * (cur1[c + 1] - cur1[c - 1]);
      float dy = 0.5f * (cur2[c] - cur0[c]);
      float ds = 0.5f * (high1[c] - low1[c]);
      bb[0] = -dx; bb[1] = -dy; bb[2] = -ds;
      solve_linear_3x3(A, bb);
      if (isnan(bb[0]) || isnan(bb[1]) || isnan(bb[2])) { ok = false; break; }
      val = cur1[c] + 0.5f * (dx * bb[0] + dy * bb[1] + ds * bb[2]);
      // MAX_SUBPIXEL_SHIFT is the double 0.6; POINT_SAFETY_BORDER 3
      if ((double)bb[0] > 0.6) { if (c < cols - 3) nc++; else { ok = false; break; } }
      if ((double)bb[1] > 0.6) { if (r < rows - 3) nr++; else { ok = false; break; } }
      if ((double)bb[0] < -0.6) { if (c > 3) nc--; else { ok = false; break; } }
      if ((double)bb[1] < -0.6) { if (r > 3) nr--; else { ok = false; break; } }
      if (nr == r && nc == c) break;
    }
    if (ok && (fabsf(bb[0]) > 1.5f || fabsf(bb[1]) > 1.5f || fabsf(bb[2]) > 1.5f || fabsf(val) < k.final_th)) ok = false;
    if (!ok) { cd.state = 0; continue; }
    const float scale = o.sigma[cd.level] * det_pow2f(bb[2] / k.n_scales);
    int type;   // getPointType, pyramid.cpp:65-124
    if (k.det_type == MODS_DET_DOG) type = val < 0 ? 11 : 10;
    else if (k.det_type == MODS_DET_HARRIS) type = val < 0 ? 31 : 30;
    else if (val < 0) type = 2;
    else {
      const float *ptr = as_global(o.blur[cd.level]) + plane + (size_t)r * cols + c;
      float Lxx = (ptr[-1] - 2 * ptr[0] + ptr[1]);
      type = (Lxx < 0) ? 0 : 1;
    }
    cd.r = r; cd.c = c;
    cd.x = o.pixelDistance * (c + bb[0]);
    cd.y = o.pixelDistance * (r + bb[1]);
    cd.s = o.pixelDistance * scale;
    cd.pixelDistance = o.pixelDistance;
    cd.type = type;
    cd.response = val;
    cd.state = 1;
    const unsigned int key = ((unsigned int)cd.level << ORDER_POS_BITS) | (unsigned int)(cd.r0 * cols + cd.c0);
    atomicMin(as_global(o.omap) + plane + (size_t)r * cols + c, key);
  }
}

// Second half of the octaveMap rule: a point survives iff it is the first claimant of its cell.
// Survivors are appended to the accepted list (order irrelevant).
__global__ __launch_bounds__(256) void accept_kernel(const PyramidDev *__restrict__ P, DetectConst k,
                                                     CandDev *__restrict__ cand, const int *__restrict__ cand_count,
                                                     int *__restrict__ acc_list, int *__restrict__ acc_count) {
  const int b = blockIdx.y;
  int n = cand_count[b];
  if (n > k.max_cand) n = k.max_cand;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    CandDev &cd = cand[(size_t)b * k.max_cand + i];
    if (cd.state != 1) continue;
    const OctaveDev &o = P->oct[cd.octave];
    const size_t plane = (size_t)o.w * o.h * b;
    const unsigned int key = ((unsigned int)cd.level << ORDER_POS_BITS) | (unsigned int)(cd.r0 * o.w + cd.c0);
    if (as_global(o.omap)[plane + (size_t)cd.r * o.w + cd.c] == key) {
      cd.state = 2;
      const int slot = atomicAdd(&acc_count[b], 1);
      acc_list[(size_t)b * k.max_cand + slot] = i;
    } else cd.state = 5;      // lost its octaveMap cell to an earlier point; r, c stay valid for omap_reset_kernel
  }
}

// The octaveMap cells this batch wrote (every localised point that passed the tests) go back to "empty", so that the map of
// one u32 per pixel and octave is cleared by ~10^5 stores instead of a 177 MB fill per 16-image batch.
__global__ __launch_bounds__(256) void omap_reset_kernel(const PyramidDev *__restrict__ P, DetectConst k, const CandDev *__restrict__ cand,
                                                         const int *__restrict__ cand_count) {
  const int b = blockIdx.y;
  int n = cand_count[b];
  if (n > k.max_cand) n = k.max_cand;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const CandDev &cd = cand[(size_t)b * k.max_cand + i];
    if (cd.state != 2 && cd.state != 5) continue;
    const OctaveDev &o = P->oct[cd.octave];
    as_global(o.omap)[(size_t)o.w * o.h * b + (size_t)cd.r * o.w + cd.c] = 0xFFFFFFFFu;
  }
}

// ---------------------------------------------------------------------------------------
// The Hessian form of the iteration (affBmbrgMethod = 1, affine.cpp:92-128): nine bilinear samples at s * affMeasRegion, the 3x3
// finite-difference Hessian of the warped neighbourhood, its SVD (OpenCV's one-sided Jacobi for fp32, restated in
// oracle/detect.cpp:svd2x2_f32 - parity unpinned, see there) and Ap <- Au Ap Au.  A few hundred instructions per step:
// one LANE per accepted point.  grid = (N, n_img) grid-stride, block 256.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void mul2x2_f32(const float a[4], const float b[4], float d[4]) {   // cv::gemm's 2x2 fp32 path
  const float t0 = a[0] * b[0] + a[1] * b[2], t1 = a[0] * b[1] + a[1] * b[3];
  const float t2 = a[2] * b[0] + a[3] * b[2], t3 = a[2] * b[1] + a[3] * b[3];
  d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
}
__device__ bool svd2x2_f32(const float A[4], float d[2], float U[4], float Vt[4]) {   // false: a singular value <= FLT_MIN
  const float eps = 1.1920929e-7f * 2;
  const double minval = 1.17549435e-38;
  float a0[2] = {A[0], A[2]}, a1[2] = {A[1], A[3]};       // rows of A^T
  float v0[2] = {1.f, 0.f}, v1[2] = {0.f, 1.f};
  double W0 = (double)a0[0] * a0[0] + (double)a0[1] * a0[1];
  double W1 = (double)a1[0] * a1[0] + (double)a1[1] * a1[1];
  for (int iter = 0; iter < 30; iter++) {
    double p = (double)a0[0] * a1[0];
    p += (double)a0[1] * a1[1];
    if (fabs(p) <= eps * sqrt(W0 * W1)) break;
    p *= 2;
    const double beta = W0 - W1, gamma = hypot(p, beta);
    float c, s;
    if (beta < 0) {
      const double delta = (gamma - beta) * 0.5;
      s = (float)sqrt(delta / gamma);
      c = (float)(p / (gamma * s * 2));
    } else {
      c = (float)sqrt((gamma + beta) / (gamma * 2));
      s = (float)(p / (gamma * c * 2));
    }
    W0 = 0; W1 = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const float t0 = c * a0[q] + s * a1[q];
      const float t1 = -s * a0[q] + c * a1[q];
      a0[q] = t0; a1[q] = t1;
      W0 += (double)t0 * t0; W1 += (double)t1 * t1;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const float t0 = c * v0[q] + s * v1[q];
      const float t1 = -s * v0[q] + c * v1[q];
      v0[q] = t0; v1[q] = t1;
    }
  }
  W0 = sqrt((double)a0[0] * a0[0] + (double)a0[1] * a0[1]);
  W1 = sqrt((double)a1[0] * a1[0] + (double)a1[1] * a1[1]);
  if (W0 < W1) {
    const double tw = W0; W0 = W1; W1 = tw;
#pragma unroll
    for (int q = 0; q < 2; q++) { float t = a0[q]; a0[q] = a1[q]; a1[q] = t; t = v0[q]; v0[q] = v1[q]; v1[q] = t; }
  }
  d[0] = (float)W0; d[1] = (float)W1;
  if (W0 <= minval || W1 <= minval) return false;
  const float s0 = (float)(1 / W0), s1 = (float)(1 / W1);
  a0[0] *= s0; a0[1] *= s0; a1[0] *= s1; a1[1] *= s1;
  U[0] = a0[0]; U[1] = a1[0]; U[2] = a0[1]; U[3] = a1[1];
  Vt[0] = v0[0]; Vt[1] = v0[1]; Vt[2] = v1[0]; Vt[3] = v1[1];
  return true;
}
__global__ __launch_bounds__(256) void baumberg_hessian_kernel(const PyramidDev *__restrict__ P, DetectConst k, CandDev *__restrict__ cand,
                                                               const int *__restrict__ acc_list, const int *__restrict__ acc_count,
                                                               unsigned long long *__restrict__ sort_keys, int *__restrict__ sort_idx,
                                                               int *__restrict__ key_count) {
  const int b = blockIdx.y;
  const int n_acc = acc_count[b];
  for (int slot = blockIdx.x * 256 + threadIdx.x; slot < n_acc; slot += gridDim.x * 256) {
    const int ci = acc_list[(size_t)b * k.max_cand + slot];
    CandDev &cd = cand[(size_t)b * k.max_cand + ci];
    const OctaveDev &o = P->oct[cd.octave];
    const bool sfi = k.sfi_img != nullptr;
    const int iw = sfi ? k.sfi_w : o.w, ih = sfi ? k.sfi_h : o.h;
    const float *im = sfi ? k.sfi_img + (size_t)iw * ih * b : as_global(o.blur[cd.level - 1]) + (size_t)iw * ih * b;
    const float pd = sfi ? 1.0f : cd.pixelDistance;
    float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;
    float u[4] = {1.0f, 0.0f, 0.0f, 1.0f};
    const float lx = cd.x / pd, ly = cd.y / pd;
    const float aff_ratio = cd.s * k.aff_meas_region / pd;
    bool converged = !k.do_baumberg;
    for (int l = 0; l < k.max_iter && k.do_baumberg; l++) {
      const float a11 = u[0] * aff_ratio, a12 = u[1] * aff_ratio, a21 = u[2] * aff_ratio, a22 = u[3] * aff_ratio;
      const bool touch = check_borders(iw, ih, lx, ly, a11, a12, a21, a22, 3, 3);
      float q[9];
      float rx = lx - 1.0f * a12, ry = ly - 1.0f * a22;         // interpolate(), helpers.cpp:551-626: sequential fp32 coordinates
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float WX = rx - 1.0f * a11, WY = ry - 1.0f * a21;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          q[j * 3 + i] = tap_combine(tap_load_bf(im, iw, ih, WX, WY, touch));
          WX += a11; WY += a21;
        }
        rx += a12; ry += a22;
      }
      const float Dxx = (q[0] - 2.f * q[1] + q[2] + 2.f * q[3] - 4.f * q[4] + 2.f * q[5] + q[6] - 2.f * q[7] + q[8]);
      const float Dyy = (q[0] + 2.f * q[1] + q[2] - 2.f * q[3] - 4.f * q[4] - 2.f * q[5] + q[6] + 2.f * q[7] + q[8]);
      const float Dxy = (q[0] - q[2] - q[6] + q[8]);
      const float Au0[4] = {Dxx, Dxy, Dxy, Dyy};
      float d[2], U[4], Vt[4];
      eigen_ratio_bef = eigen_ratio_act;
      if (!svd2x2_f32(Au0, d, U, Vt)) break;
      float l1 = d[0], l2 = d[1];
      eigen_ratio_act = (float)(1.0 - fabsf(l2) / fabsf(l1));
      const float det = sqrtf(fabsf(l1 * l2));
      l2 = sqrtf(sqrtf(fabsf(l1) / det));
      l1 = (float)(1. / l2);
      const float D[4] = {l1, 0.f, 0.f, l2};
      float UD[4], Au[4], T[4], Ap[4];
      mul2x2_f32(U, D, UD);
      mul2x2_f32(UD, Vt, Au);
      mul2x2_f32(Au, u, T);
      mul2x2_f32(T, Au, Ap);
      u[0] = Ap[0]; u[1] = Ap[1]; u[2] = Ap[2]; u[3] = Ap[3];
      // getEigenvalues, helpers.cpp:504-515
      const float trace = u[0] + u[3];
      const float delta1 = (trace * trace - 4 * (u[0] * u[3] - u[1] * u[2]));
      if (delta1 < 0) break;
      const float delta = sqrtf(delta1);
      l1 = (trace + delta) / 2.0f;
      l2 = (trace - delta) / 2.0f;
      if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
      if (eigen_ratio_act < k.conv_th && eigen_ratio_bef < k.conv_th) { converged = true; break; }
    }
    if (converged) {
      cd.a11 = u[0]; cd.a12 = u[1]; cd.a21 = u[2]; cd.a22 = u[3];
      cd.state = 3;
      const unsigned int absbits = __float_as_uint(fabsf(cd.response));
      const unsigned int order = ((unsigned int)cd.octave << 28) | ((unsigned int)cd.level << ORDER_POS_BITS) |
                                 (unsigned int)(cd.r0 * o.w + cd.c0);   // the octave's own width (iw may be the sampling image's: not unique there)
      const int sl2 = atomicAdd(&key_count[b], 1);
      sort_keys[(size_t)b * k.max_cand + sl2] = ((unsigned long long)(~absbits) << 32) | order;
      sort_idx[(size_t)b * k.max_cand + sl2] = ci;
    } else cd.state = 4;
  }
}

// ---------------------------------------------------------------------------------------
// Baumberg iteration: one wave per accepted point.
// ---------------------------------------------------------------------------------------
// invSqrt, helpers.cpp:463-502 (double inside)
__device__ void inv_sqrt(float &a, float &b, float &c, float &l1, float &l2) {
  double t, r;
  if (b != 0) {
    r = double(c - a) / (2 * b);
    if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
    else t = -1.0 / (-r + sqrt(1 + r * r));
    r = 1.0 / sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  double x, z, d;
  x = 1.0 / sqrt(r * r * a - 2 * r * t * b + t * t * c);
  z = 1.0 / sqrt(t * t * a + 2 * r * t * b + r * r * c);
  d = sqrt(x * z);
  x /= d;
  z /= d;
  if (x < z) { l1 = float(z); l2 = float(x); }
  else { l1 = float(x); l2 = float(z); }
  a = float(r * r * x + t * t * z);
  b = float(-r * t * x + t * r * z);
  c = float(t * t * x + r * r * z);
}

// The W x W window of one Baumberg iteration for windows of at most 24 columns (one batch of three column tiles per tile row), with
// the loads of tile row i + 1 in flight while tile row i is combined and stored: a wave otherwise waits one full round trip per
// tile row (five for the 19 x 19 window of the .ini), and the sample phase was 70-85 % of an iteration (BAUMBERG_PROF, round 6).
// Coordinates: the reference's sequential fp32 sums, as in the loop this replaces (row starts += a12 / a22 per row, taps += a11 /
// a21 per column); a lane's taps are issued in the same order, only their combination is delayed.
template <int TH, bool INSIDE>
__device__ __forceinline__ void baumberg_sample_piped(const float *__restrict__ im, int iw, int ih, float lx, float ly, float a11,
                                                      float a12, float a21, float a22, int W, int sl, bool touch, float *s_img) {
  const int half = W / 2;
  const int tcol = sl & 7, trow = sl >> 3;
  float rx = lx - (float)half * a12;
  float ry = ly - (float)half * a22;
#pragma unroll
  for (int q = 0; q < TH - 1; q++) { const bool m = q < trow; const float nx = rx + a12, ny = ry + a22; rx = m ? nx : rx; ry = m ? ny : ry; }
  auto issue = [&](TapLoads (&t)[3], int r0) {
    const int row = r0 + trow;
    float WX = rx - (float)half * a11;
    float WY = ry - (float)half * a21;
#pragma unroll
    for (int q = 0; q < 7; q++) { const bool m = q < tcol; const float nx = WX + a11, ny = WY + a21; WX = m ? nx : WX; WY = m ? ny : WY; }
#pragma unroll
    for (int u = 0; u < 3; u++) {
      if (INSIDE) t[u] = tap_load_inside(im, iw, WX, WY, row < W && 8 * u + tcol < W);
      else t[u] = tap_load_bf(im, iw, ih, WX, WY, touch);
#pragma unroll
      for (int q = 0; q < 8; q++) { WX += a11; WY += a21; }
    }
#pragma unroll
    for (int q = 0; q < TH; q++) { rx += a12; ry += a22; }
  };
  auto finish = [&](const TapLoads (&t)[3], int r0) {
    const int row = r0 + trow;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int col = 8 * u + tcol;
      if (row < W && col < W) s_img[row * W + col] = INSIDE ? tap_combine_t<false>(t[u]) : tap_combine(t[u]);
    }
  };
  TapLoads A[3], B[3];
  issue(A, 0);
  for (int r0 = 0; r0 < W; r0 += 2 * TH) {
    const bool haveB = r0 + TH < W;
    if (haveB) issue(B, r0 + TH);
    finish(A, r0);
    if (!haveB) break;
    if (r0 + 2 * TH < W) issue(A, r0 + 2 * TH);
    finish(B, r0 + TH);
  }
}

// grid = (N, n_img) grid-stride over the accepted list; block = 64 (one wave) = KP keypoints side by side, 64 / KP lanes each.
// Most of an iteration is work that costs the SIMD the same for one lane or for sixty-four (three 361-term ordered sums, the
// double-precision inverse square root of the 2x2 moment matrix), so several keypoints share the wave; a keypoint that has
// converged or failed idles until the others of its wave are done.
// dynamic LDS: mask WP | KP x (img, pa, pb[, pc]: 3 or 4 WP | 4 sums)
#ifndef BAUMBERG_WAVES
#define BAUMBERG_WAVES 4
#endif
template <int KP>
__global__ __launch_bounds__(64, BAUMBERG_WAVES) void baumberg_kernel(const PyramidDev *__restrict__ P, DetectConst k,
                                                      CandDev *__restrict__ cand, const int *__restrict__ acc_list,
                                                      const int *__restrict__ acc_count, const float *__restrict__ mask,
                                                      unsigned long long *__restrict__ sort_keys, int *__restrict__ sort_idx,
                                                      int *__restrict__ key_count, unsigned long long *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int G = 64 / KP;                                  // lanes per keypoint
  const int W = k.smm, WW = W * W, WP = (WW + 3) & ~3;
  const int lane = threadIdx.x, sub = lane / G, sl = lane % G;
  float *s_mask = smem;
  // a keypoint's third product array takes the place of its window once the gradient pass has read it (64 / KP >= W: see there):
  // 3 instead of 4 arrays per keypoint, 16 instead of 13 waves per CU by LDS - the kernel follows its occupancy (round 6:
  // 13 / 10 / 8 waves per CU = 1.47 / 1.67 / 1.97 ms per batch, profiles/r06_describe_experiments.log)
  const bool alias_pc = G >= W;
  const int KS = (alias_pc ? 3 : 4) * WP + 4;
  float *s_img = smem + WP + sub * KS, *s_pa = s_img + WP, *s_pb = s_pa + WP, *s_pc = alias_pc ? s_img : s_pb + WP, *s_sum = s_img + KS - 4;
  const int b = blockIdx.y;
  const int half = W / 2;
  for (int p = lane; p < WW; p += 64) s_mask[p] = mask[p];
  const int g_r0 = sl / W, g_c0 = sl - g_r0 * W, g_dr = G / W, g_dc = G - g_dr * W;   // pixel walk of the gradient pass
  const int n_acc = acc_count[b];
#ifdef BAUMBERG_PROF
  unsigned long long pt[5] = {0, 0, 0, 0, 0}, pl = __builtin_amdgcn_s_memtime();
  int pn = 0;
#define BPROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pl; pl = t_; }
#else
#define BPROF(i)
#endif
  for (int slot0 = blockIdx.x * KP; slot0 < n_acc; slot0 += gridDim.x * KP) {
    const int slot = slot0 + sub;
    const bool have = slot < n_acc;
    const int ci = acc_list[(size_t)b * k.max_cand + (have ? slot : slot0)];
    CandDev &cd = cand[(size_t)b * k.max_cand + ci];
    const OctaveDev &o = P->oct[cd.octave];
    // prevBlur (pyramid.cpp:402), or with sampleFromImage the input image at pixel distance 1 (scale-space-detector.hpp:47-55)
    const bool sfi = k.sfi_img != nullptr;
    const int iw = sfi ? k.sfi_w : o.w, ih = sfi ? k.sfi_h : o.h;
    const float *im = sfi ? k.sfi_img + (size_t)iw * ih * b : as_global(o.blur[cd.level - 1]) + (size_t)iw * ih * b;
    const float pd = sfi ? 1.0f : cd.pixelDistance;
    float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;
    float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
    const float lx = cd.x / pd, ly = cd.y / pd;
    const float ratio = cd.s / (k.initial_sigma * pd);
    bool converged = false;
    bool active = have && k.do_baumberg;       // still iterating (uniform within the keypoint's lanes)
    if (!k.do_baumberg) converged = true;
    int n_iter = 0;
    for (int l = 0; l < k.max_iter && __any(active); l++) {
      n_iter += active ? 1 : 0;
      const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
      wave_sync();   // previous iteration's readers are done with the tiles
      BPROF(0)
      // every lane samples a contiguous run of the W x W window; its first coordinates are rebuilt by
      // replaying the reference's sequential fp32 additions (row steps, then column steps)
      if (active) {
        const bool touch = check_borders(iw, ih, lx, ly, a11, a12, a21, a22, W, W);
        const bool all_inside = __all(!touch);      // over the lanes that are still iterating (one or two keypoints)
        // The lanes of a keypoint sample the window tile by tile (8 columns x G / 8 rows per step), lane = position inside
        // the tile: what a gather costs is the number of cache lines its 64 lanes touch (tools/ubench/gather.hip: 4 cycles
        // per line, 266 for 64 scattered lanes, 42 for an 8 x 8 pixel block), and a tile of neighbouring samples lies on a
        // few image rows.  A lane keeps the reference's sequential fp32 coordinate chains by walking ITS rows column by
        // column (8 additions between two of its taps) and from row to row (G / 8 additions).
        constexpr int TH = G / 8;                   // rows of a tile
#ifndef BAUMBERG_NO_PIPE
        if (W <= 24) {
          if (all_inside) baumberg_sample_piped<TH, true>(im, iw, ih, lx, ly, a11, a12, a21, a22, W, sl, touch, s_img);
          else baumberg_sample_piped<TH, false>(im, iw, ih, lx, ly, a11, a12, a21, a22, W, sl, touch, s_img);
        } else
#endif
        {
        const int tcol = sl & 7, trow = sl >> 3;
        float rx = lx - (float)half * a12;
        float ry = ly - (float)half * a22;
#pragma unroll
        for (int q = 0; q < TH - 1; q++) { const bool m = q < trow; const float nx = rx + a12, ny = ry + a22; rx = m ? nx : rx; ry = m ? ny : ry; }
        for (int r0 = 0; r0 < W; r0 += TH) {
          const int row = r0 + trow;
          float WX = rx - (float)half * a11;
          float WY = ry - (float)half * a21;
#pragma unroll
          for (int q = 0; q < 7; q++) { const bool m = q < tcol; const float nx = WX + a11, ny = WY + a21; WX = m ? nx : WX; WY = m ? ny : WY; }
          // column tiles of the row in batches of three (the 19-wide window of the .ini is one batch): 6 loads in flight.
          // When no window of the wave's keypoints touches the image border (nearly always) the taps take the unchecked form
          if (all_inside) {
            for (int c0 = 0; c0 < W; c0 += 24) {
              TapLoads t[3];
#pragma unroll
              for (int u = 0; u < 3; u++) {
                t[u] = tap_load_inside(im, iw, WX, WY, row < W && c0 + 8 * u + tcol < W);
#pragma unroll
                for (int q = 0; q < 8; q++) { WX += a11; WY += a21; }
              }
#pragma unroll
              for (int u = 0; u < 3; u++) {
                const int col = c0 + 8 * u + tcol;
                if (row < W && col < W) s_img[row * W + col] = tap_combine_t<false>(t[u]);
              }
            }
          } else {
            for (int c0 = 0; c0 < W; c0 += 24) {
              TapLoads t[3];
#pragma unroll
              for (int u = 0; u < 3; u++) {
                t[u] = tap_load_bf(im, iw, ih, WX, WY, touch);
#pragma unroll
                for (int q = 0; q < 8; q++) { WX += a11; WY += a21; }
              }
#pragma unroll
              for (int u = 0; u < 3; u++) {
                const int col = c0 + 8 * u + tcol;
                if (row < W && col < W) s_img[row * W + col] = tap_combine(t[u]);
              }
            }
          }
#pragma unroll
          for (int q = 0; q < TH; q++) { rx += a12; ry += a22; }
        }
        }
      }
      wave_sync();
      BPROF(1)
      // computeGradient (helpers.cpp:779-797) and the three SMM products
      // (the one-sided differences at the window border are the same subtraction with one operand at the pixel itself)
      // (row and column of a lane's pixels advance by G / W and G % W with one carry: no division per pixel - the compiler's
      // expansion of p / W was 16 of this loop's 58 vector instructions, and the loop a third of an iteration's)
      // The third product of a pixel is stored one round late, over the window itself: round j reads window values from G j - W
      // on, the store of round j - 1 covers [G (j - 1), G j) and is issued behind round j's reads (LDS executes a wave's
      // instructions in order), and no later round reads below G j when G >= W.
      if (active) {
        int r = g_r0, c = g_c0;
        float pend = 0.f;
        int pend_at = -1;
        for (int p = sl; p < WW; p += G) {
          const float xa = s_img[p + (c < W - 1 ? 1 : 0)], xb = s_img[p - (c > 0 ? 1 : 0)];
          const float ya = s_img[p + (r < W - 1 ? W : 0)], yb = s_img[p - (r > 0 ? W : 0)];
          const float xgrad = xa - xb, ygrad = ya - yb;
          const float v = s_mask[p];
          const float gxy = xgrad * ygrad;
          s_pa[p] = xgrad * xgrad * v;
          s_pb[p] = gxy * v;
          if (alias_pc) {
            if (pend_at >= 0) s_pc[pend_at] = pend;
            pend = ygrad * ygrad * v; pend_at = p;
          } else s_pc[p] = ygrad * ygrad * v;
          r += g_dr; c += g_dc;
          if (c >= W) { c -= W; r++; }
        }
        // the last round's products (a lane whose loop ended a round early holds a pixel of the round before): every read is done
        if (alias_pc && pend_at >= 0) s_pc[pend_at] = pend;
      }
      wave_sync();
      BPROF(2)
      // ordered accumulation (raster order, fp32): three lanes per keypoint, one sum each
      if (active && sl < 3) {
        const float *arr = sl == 0 ? s_pa : (sl == 1 ? s_pb : s_pc);
        float acc = 0;
        int i = 0;
        for (; i + 31 < WW; i += 32) {   // 8 LDS reads in flight, then their 32 terms in order
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = *(const float4 *)(arr + i + 4 * u);
#pragma unroll
          for (int u = 0; u < 8; u++) { acc += v[u].x; acc += v[u].y; acc += v[u].z; acc += v[u].w; }
        }
        for (; i + 3 < WW; i += 4) {
          const float4 v4 = *(const float4 *)(arr + i);
          acc += v4.x; acc += v4.y; acc += v4.z; acc += v4.w;
        }
        for (; i < WW; i++) acc += arr[i];
        s_sum[sl] = acc;
      }
      wave_sync();
      BPROF(3)
#ifdef BAUMBERG_PROF
      pn++;
#endif
      if (active) {
        float a = s_sum[0], bq = s_sum[1], c = s_sum[2];
        a /= WW; bq /= WW; c /= WW;
        inv_sqrt(a, bq, c, l1, l2);
        if ((a != a) || (bq != bq) || (c != c)) active = false;
        else {
          eigen_ratio_bef = eigen_ratio_act;
          eigen_ratio_act = (float)(1.0 - l2 / l1);
          const float u11t = u11, u12t = u12;
          u11 = a * u11t + bq * u21;
          u12 = a * u12t + bq * u22;
          u21 = bq * u11t + c * u21;
          u22 = bq * u12t + c * u22;
          // getEigenvalues, helpers.cpp:504-515
          const float trace = u11 + u22;
          const float delta1 = (trace * trace - 4 * (u11 * u22 - u12 * u21));
          if (delta1 < 0) active = false;
          else {
            const float delta = sqrtf(delta1);
            l1 = (trace + delta) / 2.0f;
            l2 = (trace - delta) / 2.0f;
            if ((l1 / l2 > 6) || (l2 / l1 > 6)) active = false;
            else if (eigen_ratio_act < k.conv_th && eigen_ratio_bef < k.conv_th) { converged = true; active = false; }
          }
        }
      }
    }
    if (have && sl == 0) {
      // (bench.py's per-keypoint figures; off unless mods_baumberg_stats_enable)
      if (stats && n_iter) {                     // 64 slots per image: the adds of a launch's 8192 workgroups do not queue on one word
        unsigned long long *st = stats + 2 * ((size_t)b * 64 + (blockIdx.x & 63));
        atomicAdd(&st[0], 1ull); atomicAdd(&st[1], (unsigned long long)n_iter);
      }
      if (converged) {
        cd.a11 = u11; cd.a12 = u12; cd.a21 = u21; cd.a22 = u22;
        cd.state = 3;
        // sort key: |response| descending, then processing order (octave, level, r0, c0)
        const unsigned int absbits = __float_as_uint(fabsf(cd.response));
        const unsigned int order = ((unsigned int)cd.octave << 28) | ((unsigned int)cd.level << ORDER_POS_BITS) |
                                   (unsigned int)(cd.r0 * o.w + cd.c0);   // the octave's own width (iw may be the sampling image's: not unique there)
        const int sl2 = atomicAdd(&key_count[b], 1);
        sort_keys[(size_t)b * k.max_cand + sl2] = ((unsigned long long)(~absbits) << 32) | order;
        sort_idx[(size_t)b * k.max_cand + sl2] = ci;
      } else cd.state = 4;   // accepted by the pyramid, dropped by the affine adaptation
    }
    BPROF(4)
  }
#ifdef BAUMBERG_PROF
  if (lane == 0 && b == 0 && (blockIdx.x % 512) == 5)
    printf("baumberg prof: block %d iterations %d cycles: invsqrt+update %llu sample %llu gradient %llu sums %llu tail %llu\n", blockIdx.x, pn, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
}

// ---------------------------------------------------------------------------------------
// Rank sort + export.  Keys are unique, so rank = #{j : key_j < key_i}.
// rank_count_kernel: grid = (blocks over i, splits over j, n_img): partial counts added atomically.
// export_kernel    : applies DetectAffineRegions: s *= sqrt|det A|, rectifyTransformation (fp64).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_count_kernel(DetectConst k, const unsigned long long *__restrict__ sort_keys,
                                                         const int *__restrict__ key_count, int *__restrict__ rank, int min_n) {
  __shared__ unsigned long long tile[1024];
  const int b = blockIdx.z;
  const int n = key_count[b];
  if (n <= min_n) return;                       // rank_sort_kernel's images
  const unsigned long long *keys = sort_keys + (size_t)b * k.max_cand;
  const int nblk = (n + 255) / 256;
  const int ntile = (n + 1023) / 1024;
  for (int t = blockIdx.y; t < ntile; t += gridDim.y) {
    const int t0 = t * 1024;
    __syncthreads();
    for (int q = threadIdx.x; q < 1024; q += 256) tile[q] = (t0 + q < n) ? keys[t0 + q] : ~0ull;
    __syncthreads();
    const int lim = min(1024, n - t0);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
      const int i = blk * 256 + threadIdx.x;
      if (i >= n) continue;
      const unsigned long long mine = keys[i];
      int cnt = 0;
      for (int q = 0; q < lim; q++) cnt += (tile[q] < mine) ? 1 : 0;
      if (cnt) atomicAdd(&rank[(size_t)b * k.max_cand + i], cnt);
    }
  }
}

// The same ranks by sorting: one 1024-thread workgroup per image holds the image's keys (64 bits) and their positions (16 bits)
// in LDS - 16 384 x 10 bytes = all 160 KB of a CU - and runs a bitonic network over them; rank[i] = where key i ends up.  Keys
// are unique, so the order is the one rank_count_kernel counts.  Images with more keys than fit (4096^2 images: 56 k) keep the
// counting kernel, which returns at once for the others.  10 k keys: 105 barrier-separated passes of 8 compare-exchanges per
// thread ~ 60 us per image against 310 us of counting for a batch (round 3).
constexpr int RANK_SORT_MAX = 16384;
#ifndef RANK_COUNT_MAX_IMG
#define RANK_COUNT_MAX_IMG 4      // calls with up to this many images rank by counting (latency), larger batches by sorting (throughput)
#endif
__global__ __launch_bounds__(1024) void rank_sort_kernel(DetectConst k, const unsigned long long *__restrict__ sort_keys,
                                                         const int *__restrict__ key_count, int *__restrict__ rank) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];
  const int b = blockIdx.x;
  const int n = key_count[b];
  if (n <= 0 || n > RANK_SORT_MAX) return;
  int npad = 2048;
  while (npad < n) npad <<= 1;
  unsigned short *s_idx = (unsigned short *)(s_key + npad);
  const unsigned long long *keys = sort_keys + (size_t)b * k.max_cand;
  for (int i = threadIdx.x; i < npad; i += 1024) { s_key[i] = i < n ? keys[i] : ~0ull; s_idx[i] = (unsigned short)i; }
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1)
    for (int lj = 31 - __clz(kk >> 1); lj >= 0; lj--) {
      const int j = 1 << lj;
      for (int t = threadIdx.x; t < (npad >> 1); t += 1024) {
        const int a = ((t >> lj) << (lj + 1)) + (t & (j - 1)), c = a + j;
        const unsigned long long ka = s_key[a], kc = s_key[c];
        const bool up = (a & kk) == 0;
        if ((ka > kc) == up) {
          s_key[a] = kc; s_key[c] = ka;
          const unsigned short ia = s_idx[a]; s_idx[a] = s_idx[c]; s_idx[c] = ia;
        }
      }
      __syncthreads();
    }
  int *r = rank + (size_t)b * k.max_cand;
  for (int i = threadIdx.x; i < n; i += 1024) r[s_idx[i]] = i;
}

__global__ __launch_bounds__(256) void export_kernel(DetectConst k, const CandDev *__restrict__ cand, const int *__restrict__ sort_idx,
                                                     const int *__restrict__ key_count, const int *__restrict__ rank,
                                                     mods_affkey *__restrict__ out) {
  const int b = blockIdx.y;
  const int n = key_count[b];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const CandDev &cd = cand[(size_t)b * k.max_cand + sort_idx[(size_t)b * k.max_cand + i]];
    mods_affkey o;
    double a = cd.a11, bb = cd.a12, c = cd.a21, d = cd.a22;
    o.x = cd.x; o.y = cd.y;
    o.s = (double)cd.s * sqrt(fabs(a * d - bb * c));
    const double det = sqrt(fabs(a * d - bb * c));
    const double b2a2 = sqrt(bb * bb + a * a);
    o.a11 = b2a2 / det;
    o.a12 = 0;
    o.a21 = (d * bb + c * a) / (b2a2 * det);
    o.a22 = det / b2a2;
    o.response = cd.response;
    o.sub_type = cd.type;
    o.octave = cd.octave; o.level = cd.level; o.r0 = cd.r0; o.c0 = cd.c0; o.pad = 0;
    out[(size_t)b * k.max_cand + rank[(size_t)b * k.max_cand + i]] = o;
  }
}

// AffineDetector::prepareKeysForExport (scale-space-detector.hpp:126-198) on the exported (sorted by |response|, descending)
// list: the number of keys that survive the cut of `mode`.  One block per image.  std::lower_bound with
// responseCompareInvOrder = #{|response| > |threshold|} on a sorted list, counted here without relying on the order.
__global__ __launch_bounds__(256) void select_keys_kernel(int max_cand, const mods_affkey *__restrict__ keys, int *__restrict__ key_count,
                                                          int mode, float rel_threshold, float threshold, int reg_number,
                                                          float rel_reg_number) {
  __shared__ int s_cnt;
  const int b = blockIdx.x;
  const int n = key_count[b];
  if (n <= 0) return;
  const mods_affkey *k = keys + (size_t)b * max_cand;
  double thr = 0;
  if (mode == MODS_DET_RELATIVE_TH) thr = (double)(float)(fabs(k[0].response) * rel_threshold);   // float effectiveThreshold, pyramid.h:74
  else if (mode == MODS_DET_NOT_LESS_THAN_REGIONS) thr = (double)threshold;                       // un-squared, as written (:173)
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  if (mode == MODS_DET_RELATIVE_TH || mode == MODS_DET_NOT_LESS_THAN_REGIONS) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += fabs(k[i].response) > fabs(thr) ? 1 : 0;
    if (c) atomicAdd(&s_cnt, c);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int keep = n;
    if (mode == MODS_DET_RELATIVE_TH) keep = s_cnt;
    else if (mode == MODS_DET_FIXED_REG_NUMBER) keep = min(n, reg_number);
    else if (mode == MODS_DET_RELATIVE_REG_NUMBER) keep = (int)floor((double)rel_reg_number * (double)n);
    else if (mode == MODS_DET_NOT_LESS_THAN_REGIONS) keep = s_cnt < reg_number ? min(reg_number, n) : min(s_cnt, n);
    key_count[b] = max(0, min(keep, n));
  }
}

// ---------------------------------------------------------------------------------------
static int detect_run_stages(mods_ctx *ctx);
int detect_run(mods_ctx *ctx) {
  const int rc = detect_run_stages(ctx);
  if (rc) (void)pyramid_join_side(ctx);   // an error exit ahead of the join: the side stream's launches must not outlive this call unjoined
  return rc;
}

static int detect_run_stages(mods_ctx *ctx) {
  const PyramidDev &P = ctx->pyr;
  const mods_hessaff_params &par = ctx->par;
  const int n_img = ctx->last_n_img;
  DetectConst k;
  k.border = par.border;
  k.n_scales = par.numberOfScales;
  // ScaleSpaceDetector ctor (pyramid.h:46-66): positive = 0.8*threshold (un-squared), final = threshold^2
  const double er = (double)par.edgeEigenValueRatio;
  k.edge_th = (er + 1.0f) * (er + 1.0f) / er;
  k.pos_th = (float)(0.8 * par.threshold);
  k.neg_th = -k.pos_th;
  k.final_th = par.detectorType == MODS_DET_HESSIAN ? par.threshold * par.threshold : par.threshold;   // pyramid.h:55-56
  k.det_type = par.detectorType;
  if (par.mode != MODS_DET_FIXED_TH) {   // pyramid.h:58-59: every extremum is a candidate, the sorted list is cut afterwards
    if (par.mode < 0 || par.mode > MODS_DET_NOT_LESS_THAN_REGIONS) { set_error("unknown detector mode %d", par.mode); return MODS_E_ARG; }
    // the reference resizes its list to these numbers unchecked (negative / > 1 values end in std::length_error or in
    // default-constructed keypoints): refuse them
    if ((par.mode == MODS_DET_FIXED_REG_NUMBER || par.mode == MODS_DET_NOT_LESS_THAN_REGIONS) && par.regionsNumber < 0) { set_error("regionsNumber must be >= 0 in this mode"); return MODS_E_ARG; }
    if (par.mode == MODS_DET_RELATIVE_REG_NUMBER && !(par.relativeRegionsNumber >= 0.f && par.relativeRegionsNumber <= 1.f)) { set_error("relativeRegionsNumber must be in [0, 1]"); return MODS_E_ARG; }
    if (par.mode == MODS_DET_RELATIVE_TH && !(par.relativeThreshold >= 0.f)) { set_error("relativeThreshold must be >= 0"); return MODS_E_ARG; }
    k.pos_th = k.neg_th = k.final_th = 0.f;
  }
  k.max_cand = ctx->max_cand;
  k.smm = par.smmWindowSize;
  k.max_iter = par.maxIterations;
  k.conv_th = par.convergenceThreshold;
  k.initial_sigma = par.initialSigma;
  k.do_baumberg = par.doBaumberg;
  k.aff_meas_region = 0.5f;
  k.sfi_img = par.sampleFromImage ? ctx->last_img_dev : nullptr;
  k.sfi_w = ctx->last_w; k.sfi_h = ctx->last_h;
  if (par.affBmbrgMethod != 0 && par.affBmbrgMethod != 1) { set_error("affBmbrgMethod %d: 0 (second moment matrix) or 1 (Hessian)", par.affBmbrgMethod); return MODS_E_ARG; }
  if (par.sampleFromImage && (!ctx->last_img_dev || ctx->last_stride != ctx->last_w)) { set_error("sampleFromImage needs the dense input image of the batch"); return MODS_E_ARG; }
  for (int oi = 0; oi < P.n_oct; oi++)
    if ((size_t)P.oct[oi].w * P.oct[oi].h >= (1u << ORDER_POS_BITS)) { set_error("octave too large for the order key"); return MODS_E_ARG; }

  // (with a forked pyramid these fills go to the side stream: off the longer chain, and joined before their first reader)
  hipStream_t fill_stream = ctx->pyr_side ? ctx->stream2 : ctx->stream;
  MODS_HIP_CHECK(hipMemsetAsync(ctx->cand_count, 0, sizeof(int) * 3 * ctx->batch, fill_stream));   // cand/acc/key counts
  if (ctx->omap_dirty) {     // first use of the pool, or a call that did not reach omap_reset_kernel: fill it once
    MODS_HIP_CHECK(hipMemsetAsync(ctx->omap_pool, 0xFF, ctx->omap_pool_elems * sizeof(unsigned int), fill_stream));
  }
  ctx->omap_dirty = true;    // until this call has put the cells it claims back
  int *acc_count = ctx->cand_count + ctx->batch;
  int *key_count = ctx->cand_count + 2 * ctx->batch;
  {
    StageScope ts(ctx, MODS_STAGE_NMS);
    // three launches for the whole pyramid: the four-columns-per-lane kernel over every octave whose rows are 16-byte aligned
    // float4s, the one-column kernel over the others, the compaction of all ballot words
    NmsPlan wide_pl, narrow_pl, comp_pl;
    wide_pl.n = narrow_pl.n = comp_pl.n = 0;
    wide_pl.blk_begin[0] = narrow_pl.blk_begin[0] = comp_pl.blk_begin[0] = 0;
    size_t mask_words = 0;
    for (int oi = 0; oi < P.n_oct; oi++) {
      const OctaveDev &o = P.oct[oi];
      const int iw = o.w - 2 * par.border, ih = o.h - 2 * par.border;
      if (iw <= 0 || ih <= 0) continue;
      const bool wide = (o.w & 3) == 0 && o.w >= 8;
      const int nblk = wide ? (o.w + NMS4_COLS - 1) / NMS4_COLS : (iw + NMS_COLS - 1) / NMS_COLS;
      const int words = wide ? 4 * nblk : nblk;
      const int nby = (ih + 4 * NMS_ROWS - 1) / (4 * NMS_ROWS);
      NmsPlan &pl = wide ? wide_pl : narrow_pl;
      pl.oi[pl.n] = oi; pl.nbx[pl.n] = nblk; pl.w[pl.n] = o.w; pl.h[pl.n] = o.h; pl.words[pl.n] = words; pl.wide[pl.n] = wide ? 1 : 0;
      pl.mask_off[pl.n] = mask_words;
      pl.blk_begin[pl.n + 1] = pl.blk_begin[pl.n] + nblk * nby;
      pl.n++;
      const int total = par.numberOfScales * ih * words;
      comp_pl.oi[comp_pl.n] = oi; comp_pl.nbx[comp_pl.n] = nblk; comp_pl.w[comp_pl.n] = o.w; comp_pl.h[comp_pl.n] = o.h;
      comp_pl.words[comp_pl.n] = words; comp_pl.wide[comp_pl.n] = wide ? 1 : 0; comp_pl.mask_off[comp_pl.n] = mask_words;
      comp_pl.blk_begin[comp_pl.n + 1] = comp_pl.blk_begin[comp_pl.n] + (total + 256 * NMS_CW - 1) / (256 * NMS_CW);
      comp_pl.n++;
      mask_words += (size_t)n_img * total;
    }
    if (mask_words > ctx->nms_mask_words) { set_error("nms mask buffer too small"); return MODS_E_CAPACITY; }
    unsigned long long *mask = (unsigned long long *)ctx->nms_mask;
    if (wide_pl.n && narrow_pl.n && narrow_pl.blk_begin[narrow_pl.n] <= 64 && wide_pl.n + narrow_pl.n <= kMaxOctaves) {
      for (int e = 0; e < narrow_pl.n; e++) {      // the tail octaves (30 x 17 pixels ...): not worth a launch of their own
        const int d = wide_pl.n;
        wide_pl.oi[d] = narrow_pl.oi[e]; wide_pl.nbx[d] = narrow_pl.nbx[e]; wide_pl.w[d] = narrow_pl.w[e]; wide_pl.h[d] = narrow_pl.h[e];
        wide_pl.words[d] = narrow_pl.words[e]; wide_pl.wide[d] = 0; wide_pl.mask_off[d] = narrow_pl.mask_off[e];
        wide_pl.blk_begin[d + 1] = wide_pl.blk_begin[d] + (narrow_pl.blk_begin[e + 1] - narrow_pl.blk_begin[e]);
        wide_pl.n++;
      }
      narrow_pl.n = 0;
    }
    if (ctx->pyr_side) {
      // the large octaves' planes are on this stream, the others' on the side stream (pyramid_build): each part's NMS follows its
      // planes, the compaction waits for both
      auto split = [&](const NmsPlan &pl, NmsPlan &first, NmsPlan &rest) {
        first.n = rest.n = 0; first.blk_begin[0] = rest.blk_begin[0] = 0;
        for (int e = 0; e < pl.n; e++) {
          NmsPlan &d = pl.oi[e] < ctx->pyr_side_first ? first : rest;
          const int i = d.n;
          d.oi[i] = pl.oi[e]; d.nbx[i] = pl.nbx[e]; d.w[i] = pl.w[e]; d.h[i] = pl.h[e]; d.words[i] = pl.words[e]; d.wide[i] = pl.wide[e];
          d.mask_off[i] = pl.mask_off[e];
          d.blk_begin[i + 1] = d.blk_begin[i] + (pl.blk_begin[e + 1] - pl.blk_begin[e]);
          d.n++;
        }
      };
      NmsPlan w0, w1, n0, n1;
      split(wide_pl, w0, w1); split(narrow_pl, n0, n1);
      if (w0.n) hipLaunchKernelGGL(nms4_kernel, dim3(w0.blk_begin[w0.n], 1, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, w0, k, mask);
      if (n0.n) hipLaunchKernelGGL(nms_kernel, dim3(n0.blk_begin[n0.n], 1, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, n0, k, mask);
      if (w1.n) hipLaunchKernelGGL(nms4_kernel, dim3(w1.blk_begin[w1.n], 1, n_img), dim3(256), 0, ctx->stream2, ctx->pyr_dev, w1, k, mask);
      if (n1.n) hipLaunchKernelGGL(nms_kernel, dim3(n1.blk_begin[n1.n], 1, n_img), dim3(256), 0, ctx->stream2, ctx->pyr_dev, n1, k, mask);
      { const int jrc = pyramid_join_side(ctx); if (jrc) return jrc; }
    } else {
      if (wide_pl.n) hipLaunchKernelGGL(nms4_kernel, dim3(wide_pl.blk_begin[wide_pl.n], 1, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, wide_pl, k, mask);
      if (narrow_pl.n) hipLaunchKernelGGL(nms_kernel, dim3(narrow_pl.blk_begin[narrow_pl.n], 1, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, narrow_pl, k, mask);
    }
    if (comp_pl.n) hipLaunchKernelGGL(nms_compact_kernel, dim3(comp_pl.blk_begin[comp_pl.n], n_img), dim3(256), 0, ctx->stream, comp_pl, k, mask, ctx->cand,
                                      ctx->cand_count);
    MODS_HIP_CHECK(hipGetLastError());
  }
  if (ctx->pyr_scope_begin) {      // MODS_STAGE_PYRAMID: from the first blur launch to here
    StageTimer &t = ctx->timers[MODS_STAGE_PYRAMID];
    hipEvent_t e1;
    if (!t.pool.empty()) { e1 = t.pool.back(); t.pool.pop_back(); }
    else MODS_HIP_CHECK(hipEventCreate(&e1));
    MODS_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    t.pending.emplace_back(ctx->pyr_scope_begin, e1);
    ctx->pyr_scope_begin = nullptr;
  }
  {
    StageScope ts(ctx, MODS_STAGE_LOCALIZE);
    hipLaunchKernelGGL(localize_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, k, ctx->cand, ctx->cand_count);
    hipLaunchKernelGGL(accept_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, k, ctx->cand, ctx->cand_count,
                       ctx->sort_idx + (size_t)ctx->batch * ctx->max_cand, acc_count);
    hipLaunchKernelGGL(omap_reset_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, k, ctx->cand, ctx->cand_count);
    MODS_HIP_CHECK(hipGetLastError());
    ctx->omap_dirty = false;
  }
  {
    StageScope ts(ctx, MODS_STAGE_BAUMBERG);
#ifndef BAUMBERG_KP
#define BAUMBERG_KP 2
#endif
    const size_t wp = (((size_t)par.smmWindowSize * par.smmWindowSize) + 3) & ~(size_t)3;
    // (three arrays per keypoint when the third product can take the window's place: baumberg_kernel, alias_pc)
    const size_t lds = sizeof(float) * (wp + BAUMBERG_KP * ((64 / BAUMBERG_KP >= par.smmWindowSize ? 3 : 4) * wp + 4));
    if (par.affBmbrgMethod == 1)
      hipLaunchKernelGGL(baumberg_hessian_kernel, dim3(64, n_img), dim3(256), 0, ctx->stream, ctx->pyr_dev, k, ctx->cand,
                         ctx->sort_idx + (size_t)ctx->batch * ctx->max_cand, acc_count, ctx->sort_keys, ctx->sort_idx, key_count);
    else
      hipLaunchKernelGGL(baumberg_kernel<BAUMBERG_KP>, dim3(8192, n_img), dim3(64), lds, ctx->stream, ctx->pyr_dev, k, ctx->cand,
                         ctx->sort_idx + (size_t)ctx->batch * ctx->max_cand, acc_count, ctx->smm_mask_dev, ctx->sort_keys,
                         ctx->sort_idx, key_count, ctx->baum_stats_dev);
    MODS_HIP_CHECK(hipGetLastError());
  }
  {
    StageScope ts(ctx, MODS_STAGE_SORT);
    // rank array: the raw-hit half of sort_idx's accept list is dead by now; use the dedicated buffer
    MODS_HIP_CHECK(hipMemsetAsync(ctx->rank_dev, 0, sizeof(int) * (size_t)ctx->max_cand * n_img, ctx->stream));
    // images of up to RANK_SORT_MAX keys are ranked by one workgroup out of LDS, larger ones by the O(n^2) count.  The sort is the
    // cheaper one for the GPU (one CU per image for ~0.2 ms: 105 barrier-separated passes) and the slower one for the caller: a call
    // with a few images - one pair, a view of the ladder - has the GPU to itself, where the count over 4 096 workgroups takes a few
    // microseconds; both give the same ranks (the keys are unique)
    const int sort_max = n_img <= RANK_COUNT_MAX_IMG ? 0 : RANK_SORT_MAX;
    if (sort_max > 0) {
      static DynLdsOnce once;
      MODS_HIP_CHECK(dyn_lds_once(once, (const void *)rank_sort_kernel, 160 * 1024, ctx->device));
      hipLaunchKernelGGL(rank_sort_kernel, dim3(n_img), dim3(1024), RANK_SORT_MAX * 10, ctx->stream, k, ctx->sort_keys, key_count, ctx->rank_dev);
    }
    hipLaunchKernelGGL(rank_count_kernel, dim3(64, 32, n_img), dim3(256), 0, ctx->stream, k, ctx->sort_keys, key_count, ctx->rank_dev, sort_max);
    hipLaunchKernelGGL(export_kernel, dim3(256, n_img), dim3(256), 0, ctx->stream, k, ctx->cand, ctx->sort_idx, key_count,
                       ctx->rank_dev, ctx->keys_dev);
    if (par.mode != MODS_DET_FIXED_TH)
      hipLaunchKernelGGL(select_keys_kernel, dim3(n_img), dim3(256), 0, ctx->stream, ctx->max_cand, ctx->keys_dev, key_count, par.mode,
                         par.relativeThreshold, par.threshold, ctx->reg_number_eff, par.relativeRegionsNumber);
    MODS_HIP_CHECK(hipGetLastError());
  }
  return MODS_OK;
}

}  // namespace mods
