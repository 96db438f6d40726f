// Brute-force 128-D nearest-neighbour matching with the first-geometrically-inconsistent
// (FGINN) ratio test on the gfx950 matrix cores.
//
// Reference behaviour: MatchFlannFGINN, matching/matching.cpp:356-460, with the exact linear
// index ([Matching] vector_matcher = linear, io_mods.cpp:389-390): for every query descriptor
// the k = 50 nearest train descriptors (squared L2, ties by lower index) are walked in order;
// neighbour j is accepted when d0/dj <= ratio^2, the walk stops at the first neighbour further
// than contradDist from the nearest one.
//
// The walk is restated as reductions that a tiled distance GEMM can fold into its epilogue
// (no N x M matrix, no top-k lists).  With (d0, i0) the nearest neighbour and D* the smallest
// integer distance whose ratio test passes (fl32(d0/d) <= ratio^2; the quotient is monotone):
//   reject  <=> some train t != i0 has d_t < D* and lies > contradDist from i0
//   else accept the first train in (d, t) order with d_t >= D*, provided at most nn-2 trains
//   (all consistent, all with d_t < D*) precede it.
// Pass 1 finds (d0, i0); pass 2 accumulates {any inconsistent below D*, count below D*,
// min key at/above D*, min key below D*} per query.
//
// Distances are exact integers: descriptors are offset to int8 (v - 128), the contraction runs
// on v_mfma_i32_32x32x32_i8 (i32 accumulate) and d = cq + ct - 2*dot with precombined norms.
// Trains are the MFMA rows (streamed), queries the columns (resident in registers), so every
// lane reduces its 16 results per tile into per-query running values.
#include "common.hpp"

namespace mods {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MatchConst {
  int n_q, n_t;
  int nn;
  double sqminratio, contr_sq;
  int tiles_per_split;
  int max_distance;       // >= 0: MatchFLANNDistance (Hamming) decisions in the emit stage; -1: FGINN
};

// Packs one region list for the matcher: int8 descriptors (v-128), c = sum v^2 - 256*sum(v-128),
// centre coordinates.  grid = ceil(n/4), block = 256 (one wave per region).
__global__ __launch_bounds__(256) void match_pack_kernel(const mods_region *__restrict__ reg, const int *__restrict__ count_ptr,
                                                         int count_fixed, int8_t *__restrict__ desc, int *__restrict__ cvec,
                                                         int *__restrict__ c2neg, unsigned int *__restrict__ parity,
                                                         double2 *__restrict__ xy, int max_n) {
  int n = count_ptr ? *count_ptr : count_fixed;
  if (n > max_n) n = max_n;
  const int lane = threadIdx.x & 63;
  // rows of the last tile beyond the list: a seed that can never be a tile maximum (stale or zero entries there
  // must not shadow the valid rows; -2 * seed still fits an int)
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    const int i = n + (int)threadIdx.x;
    if (i < ((n + 31) & ~31)) c2neg[i] = -(1 << 25);
  }
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
    const uint8_t *d = reg[i].desc;
    const int v0 = d[lane * 2], v1 = d[lane * 2 + 1];
    desc[(size_t)i * 128 + lane * 2] = (int8_t)(v0 - 128);
    desc[(size_t)i * 128 + lane * 2 + 1] = (int8_t)(v1 - 128);
    int n2 = v0 * v0 + v1 * v1;
    int s1 = (v0 - 128) + (v1 - 128);
    for (int off = 32; off > 0; off >>= 1) { n2 += __shfl_xor(n2, off); s1 += __shfl_xor(s1, off); }
    if (lane == 0) {
      const int c = n2 - 256 * s1;
      cvec[i] = c;
      c2neg[i] = -(c >> 1);       // accumulator seed of the distance tiles: acc = dot - floor(c/2)
      if (c & 1) atomicOr(&parity[i >> 5], 1u << (i & 31));   // c & 1 of the 32 rows of a tile in one word (zeroed by the caller)
      xy[i] = make_double2(reg[i].x, reg[i].y);
    }
  }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return ((unsigned long long)hi << 32) | lo;
}

// Epilogue design.  A 32x32x128 tile costs 4 MFMAs (~128 clk of the matrix pipe) and leaves 16 results per
// lane; building 64-bit (distance, index) keys for all of them costs several times that in VALU work.  So:
//   - the accumulators are seeded with -floor(ct/2) of their train rows (the seed loads land directly in the
//     C registers), which makes the MFMA result acc = dot - floor(ct/2) and the squared distance
//     d = cq + (ct & 1) - 2*acc: the nearest trains of a tile are simply its LARGEST accumulators;
//   - the fast path takes the maximum of the 16 accumulators (eight 3-input max) and compares -2*max with the
//     lane's threshold on d - cq (pass 1: best distance so far; pass 2: max(D*, best distance >= D* so far));
//     -2*acc <= d - cq, so a tile that cannot matter never passes, and only passing tiles take the exact path;
//   - train tiles stream in index order, so after the first few hundred trains a lane passes with probability
//     ~32/T per tile.
// A wave keeps QB query blocks resident (QB x 4 B-operand registers): every A tile fetched from L1/L2 feeds
// QB x 4 MFMAs.
constexpr int MATCH_QB = 2;        // pass 2 (its exact path is taken per wave: fewer queries per wave, fewer exact tiles)
#ifndef MATCH_QB_NN1
#define MATCH_QB_NN1 2   // measured: 4 blocks per wave (16 MFMAs per tile) run slower, 0.67 vs 0.39 ms on 56 k x 40 k
#endif
constexpr int MATCH_QB1 = MATCH_QB_NN1;   // pass 1: 16 MFMAs per train tile and wave

__device__ __forceinline__ int acc_max16(const v16i &acc) {
  int m = max(max(acc[0], acc[1]), acc[2]);
  m = max(max(m, acc[3]), acc[4]);
  m = max(max(m, acc[5]), acc[6]);
  m = max(max(m, acc[7]), acc[8]);
  m = max(max(m, acc[9]), acc[10]);
  m = max(max(m, acc[11]), acc[12]);
  m = max(max(m, acc[13]), acc[14]);
  return max(m, acc[15]);
}
// accumulator seed of one tile for this lane: rows (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ v16i acc_seed(const int *__restrict__ c2n, int tbase, int g) {
  v16i acc;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int4 c = *(const int4 *)(c2n + tbase + 8 * q + 4 * g);
    acc[4 * q + 0] = c.x; acc[4 * q + 1] = c.y; acc[4 * q + 2] = c.z; acc[4 * q + 3] = c.w;
  }
  return acc;
}

// LDS image of a train tile: 32 rows of 128 int8 at a padded stride (144 B: the 16-byte operand reads of 16 consecutive
// rows then fall into distinct banks), followed by the 32 accumulator seeds
constexpr int MT_ROW = 144;
constexpr int MT_BYTES = 32 * MT_ROW + 128;
struct TileRegs { v4i a; int s; };
__device__ __forceinline__ TileRegs tile_fetch(const int8_t *__restrict__ tdesc, const int *__restrict__ tc2n, int tt) {
  TileRegs r;
  r.a = *(const v4i *)(tdesc + (size_t)tt * 4096 + threadIdx.x * 16);
  r.s = threadIdx.x < 32 ? tc2n[tt * 32 + threadIdx.x] : 0;
  return r;
}
__device__ __forceinline__ void tile_store(char *buf, const TileRegs &r) {
  *(v4i *)(buf + (threadIdx.x >> 3) * MT_ROW + (threadIdx.x & 7) * 16) = r.a;
  if (threadIdx.x < 32) ((int *)(buf + 32 * MT_ROW))[threadIdx.x] = r.s;
}

// Pass 1: the two smallest keys (d << 32 | t) per query and train split.  grid = (ceil(n_q/(128*QB)), splits), block 256.
// The second key is what makes pass 2 rare: with (d1, t1) the smallest key over t != i0, a query whose d1 is >= D* has no
// train below D* at all and its first train at or above D* IS (d1, t1) - see match_mid_kernel.
// best2: [split][n_qpad][2], n_qpad = gridDim.x * 128 * QB (plain stores: every (split, query) has one owner).
//
// Exact path = branch-free register top-2 of packed 32-bit keys.  With acc = dot - floor(ct/2) the distance is
// d = cq + (ct & 1) - 2*acc, so k = 2*acc - (ct & 1) = -(d - cq) orders the trains of a query (larger = nearer).  A lane
// packs k4 = (k << 4) | (15 - r) (r = accumulator register = row order inside the lane) with ONE v_lshl_add_u32 per value:
// k4 = (acc << 5) + C[row], C[row] = (-(ct & 1) << 4) | (15 - r(row)) comes with the tile through LDS.  The two largest
// k4 of a lane are kept by v_med3_i32 + v_max_i32 per value (no compares, no exec masks, no LDS keys), and the tile
// they came from by five selects per tile: values of one tile are distinct (distinct r), equal k4 of different tiles
// keep the earlier tile = the lower train index, which is the (d, t) order.  |acc| < 2^23 (dot within +-2.1 M, ct/2 within
// 2.1 M), rows past the end of the list carry the seed -2^25: their k4 lie below every real one and do not wrap.
constexpr int MT2_BYTES = 32 * MT_ROW + 256;
#ifndef MATCH_WG_WAVES
#define MATCH_WG_WAVES 4   // waves of a pass-1 workgroup = the waves that wait for each other at the tile barrier
#endif
constexpr int NN1_WAVES = MATCH_WG_WAVES, NN1_THREADS = 64 * NN1_WAVES, NN1_LOADS = 256 / NN1_THREADS;   // 16-byte loads per thread and tile
struct TileRegs2 { v4i a[NN1_LOADS]; int s; };
__device__ __forceinline__ TileRegs2 tile_fetch2(const int8_t *__restrict__ tdesc, const int *__restrict__ tc2n, const int *__restrict__ tc, int tt) {
  TileRegs2 r;
#pragma unroll
  for (int i = 0; i < NN1_LOADS; i++) r.a[i] = *(const v4i *)(tdesc + (size_t)tt * 4096 + (threadIdx.x + i * NN1_THREADS) * 16);
  // threads 0..31: the seeds, 32..63: ct of the rows (turned into the key constants when the tile is stored: nothing
  // waits for this load before the MFMAs)
  r.s = threadIdx.x < 64 ? (threadIdx.x < 32 ? tc2n : tc - 32)[tt * 32 + threadIdx.x] : 0;
  return r;
}
__device__ __forceinline__ void tile_store2(char *buf, const TileRegs2 &r) {
#pragma unroll
  for (int i = 0; i < NN1_LOADS; i++) {
    const int t = threadIdx.x + i * NN1_THREADS;
    *(v4i *)(buf + (t >> 3) * MT_ROW + (t & 7) * 16) = r.a[i];
  }
  if (threadIdx.x < 64) {
    const int row = threadIdx.x & 31;
    const int ck = (-(r.s & 1) * 16) | (15 - ((row & 3) + 4 * (row >> 3)));
    ((int *)(buf + 32 * MT_ROW))[threadIdx.x] = threadIdx.x < 32 ? r.s : ck;
  }
}
// max of a value over the two half-waves (lane l and lane l ^ 32) in one VALU instruction: v_permlane32_swap exchanges the upper
// half of one copy with the lower half of the other, so both copies then hold both halves' values lane by lane (the shuffle
// builtin goes through ds_bpermute: eight address instructions and an LDS round trip inside the exact path)
__device__ __forceinline__ int max_halves(int x) {
#ifdef MATCH_NO_PERMLANE
  return max(x, __shfl_xor(x, 32));
#else
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  return max((int)r[0], (int)r[1]);
#endif
}
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
  int o;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}
#ifdef MATCH_STATS
__device__ unsigned long long g_match_stats[4];
#endif
#ifndef MATCH_XCH
#define MATCH_XCH 16   // tiles between two exchanges of the shared bound (a power of two)
#endif

// amdgpu_num_vgpr counts in pairs on gfx950 (the value is doubled for the unified register file): 62 = at most 124 VGPRs, i.e.
// v124..v127 of the 128 allocated (4 waves per SIMD) stay untouched.  With all 128 in use - `ds_read_b128 v[124:127]` fed the MFMAs -
// this kernel intermittently changed single results of kernels running next to it on OTHER streams (one keypoint's shape, one
// descriptor per ~10^3 pairs under the six-worker pipeline; never with one stream); every build that left the top registers
// unused, whatever else it changed, was clean.  tools/stress_match.py reproduces it, DESIGN.md "The matcher and its neighbours".
#ifndef MATCH_NN1_VGPRS
#define MATCH_NN1_VGPRS 62
#endif
__global__ __launch_bounds__(NN1_THREADS) __attribute__((amdgpu_num_vgpr(MATCH_NN1_VGPRS))) void match_nn1_kernel(MatchConst k, const int8_t *__restrict__ qdesc, const int *__restrict__ qc,
                                                        const int8_t *__restrict__ tdesc, const int *__restrict__ tc,
                                                        const int *__restrict__ tc2n, const unsigned int *__restrict__ tpar,
                                                        unsigned long long *__restrict__ best2, int *__restrict__ gthr) {
  constexpr int QB = MATCH_QB1;
  constexpr int NONE = (int)0x80000000;
  const int lane = threadIdx.x & 63, g = lane >> 5;
  const int jbase = (blockIdx.x * NN1_WAVES + (threadIdx.x >> 6)) * (32 * QB) + (lane & 31);
  v4i bq[QB][4];
  int M1[QB], M2[QB], T1[QB], T2[QB];   // two largest packed keys of this lane's rows and their tiles
  int bk[QB], alim[QB];                 // bound on k = -(d - cq): only k >= bk can matter; acc >= alim <=> 2*acc >= bk
#pragma unroll
  for (int b = 0; b < QB; b++) {
    const int j = jbase + 32 * b;
    const int jc = j < k.n_q ? j : k.n_q - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) bq[b][ks] = *(const v4i *)(qdesc + (size_t)jc * 128 + ks * 32 + g * 16);
    // gthr = (second smallest distance) - cq + 1 shared by the workgroups that scan other train ranges for the same queries
    // (starts at 0x7f7f7f7f = no bound): the global second key is at most any split's second key, so trains above it
    // cannot be among the two nearest.  d - cq < gthr  <=>  k >= 1 - gthr
    bk[b] = 1 - __hip_atomic_load(&gthr[jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    alim[b] = (bk[b] + 1) >> 1;
    M1[b] = NONE; M2[b] = NONE; T1[b] = 0; T2[b] = 0;
  }
  const int n_tiles = (k.n_t + 31) / 32;
  const int t0 = blockIdx.y * k.tiles_per_split;
  const int t1 = min(n_tiles, t0 + k.tiles_per_split);
  // The four waves of a workgroup walk the same train tiles: a tile (32 x 128 B), its 32 accumulator seeds and its 32 key
  // constants are fetched from global memory once per workgroup (one 16-byte load per thread), handed over through LDS
  // (double buffered, one barrier per tile) and read from there as MFMA operands; the next tile's loads are in flight
  // during the MFMAs.
#ifdef MATCH_STATS
  unsigned int st_exact = 0, st_all = 0;
#endif
  // the 8 MFMAs of one tile (LDS image `cur`) into acc
  auto mm = [&](const char *cur, v16i (&acc)[QB]) {
    v4i a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) a[ks] = *(const v4i *)(cur + (lane & 31) * MT_ROW + ks * 32 + g * 16);
    acc[0] = acc_seed((const int *)(cur + 32 * MT_ROW), 0, g);
#pragma unroll
    for (int b = 1; b < QB; b++) acc[b] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int b = 0; b < QB; b++) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], bq[b][ks], acc[b], 0, 0, 0);
  };
  // epilogue of tile tt (its LDS image `cur` still holds the key constants)
  auto epi = [&](const char *cur, v16i (&acc)[QB], int tt) {
    // key constants of this lane's 16 rows: read for every tile (they share the A operands' dead registers), so that the exact
    // path does not start with an LDS round trip
    int4 ck[4];
#pragma unroll
    for (int q = 0; q < 4; q++) ck[q] = ((const int4 *)(cur + 32 * MT_ROW + 128))[2 * q + g];
#pragma unroll
    for (int b = 0; b < QB; b++) {
#ifdef MATCH_STATS
      st_all++;
#endif
      // some train of this tile may be among the two nearest so far: a wave-level branch.  The 16 values of a lane are four
      // groups of four rows (one accumulator quad each); a group is inserted only when some lane of the wave holds a value of
      // that group at or above its bound - almost always ONE group of the four.  A lane inserts all four values of such a
      // group (harmless for the lanes that did not pass: the insertion is exact whatever the bound).  Per entry ~30 VALU
      // instructions instead of 63: the kernel issues about as many VALU cycles as MFMA cycles, so the count matters as much
      // as the latency.
      int gm[4];
#pragma unroll
      for (int q = 0; q < 4; q++) gm[q] = max(max(max(acc[b][4 * q + 0], acc[b][4 * q + 1]), acc[b][4 * q + 2]), acc[b][4 * q + 3]);
      const int tmax = max(max(max(gm[0], gm[1]), gm[2]), gm[3]);
      if (__any(tmax >= alim[b])) {
#ifdef MATCH_STATS
        st_exact++;
#endif
        const int m1o = M1[b], m2o = M2[b];
        int m1 = m1o, m2 = m2o;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (!__any(gm[q] >= alim[b])) continue;
          const int4 c = ck[q];   // rows 8q + 4g .. + 3
          const int x0 = (acc[b][4 * q + 0] << 5) + c.x, x1 = (acc[b][4 * q + 1] << 5) + c.y;
          const int x2 = (acc[b][4 * q + 2] << 5) + c.z, x3 = (acc[b][4 * q + 3] << 5) + c.w;
          m2 = med3_i32(m1, m2, x0); m1 = max(m1, x0);
          m2 = med3_i32(m1, m2, x1); m1 = max(m1, x1);
          m2 = med3_i32(m1, m2, x2); m1 = max(m1, x2);
          m2 = med3_i32(m1, m2, x3); m1 = max(m1, x3);
        }
        const bool c1 = m1 != m1o;
        T2[b] = (c1 && m2 == m1o) ? T1[b] : (m2 == m2o ? T2[b] : tt);
        T1[b] = c1 ? tt : T1[b];
        M1[b] = m1; M2[b] = m2;
        // the two half-waves hold the same 32 queries (different train rows): the second key of their union is at least the
        // larger of their second keys; an equal distance at a lower index still counts, hence k >= bound
        const int nb = max_halves(max(bk[b], m2 >> 4));
        bk[b] = nb; alim[b] = (nb + 1) >> 1;
      }
    }
    if (((tt - t0) & (MATCH_XCH - 1)) == MATCH_XCH - 1) {   // publish / pick up the bound every MATCH_XCH tiles
#pragma unroll
      for (int b = 0; b < QB; b++) {
        const int j = jbase + 32 * b;
        int nb = bk[b];
#ifndef MATCH_NO_ATOMIC
        if (g == 0 && j < k.n_q) nb = max(nb, 1 - atomicMin(&gthr[j], 1 - nb));
#endif
        nb = max_halves(nb);
        bk[b] = nb; alim[b] = (nb + 1) >> 1;
      }
    }
  };
  __shared__ __attribute__((aligned(16))) char s_tile[2 * MT2_BYTES];
  TileRegs2 nxt;
  if (t0 < t1) { nxt = tile_fetch2(tdesc, tc2n, tc, t0); tile_store2(s_tile, nxt); }
  __syncthreads();
  for (int tt = t0; tt < t1; tt++) {
    const char *cur = s_tile + ((tt - t0) & 1) * MT2_BYTES;
    if (tt + 1 < t1) nxt = tile_fetch2(tdesc, tc2n, tc, tt + 1);
    v16i acc[QB];
    mm(cur, acc);
    epi(cur, acc, tt);
    if (tt + 1 < t1) tile_store2(s_tile + (((tt - t0) & 1) ^ 1) * MT2_BYTES, nxt);
    __syncthreads();
  }
#ifdef MATCH_STATS
  if (lane == 0) { atomicAdd(&g_match_stats[0], (unsigned long long)st_all); atomicAdd(&g_match_stats[1], (unsigned long long)st_exact); }
#endif
  const size_t n_qpad = (size_t)gridDim.x * (32 * NN1_WAVES) * QB;
#pragma unroll
  for (int b = 0; b < QB; b++) {
    const int j = jbase + 32 * b;
    const int cqv = qc[j < k.n_q ? j : k.n_q - 1] - 4194304;
    unsigned long long mine_b = ~0ull, sec_b = ~0ull;
    {
      const int r = 15 - (M1[b] & 15), t = T1[b] * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (M1[b] != NONE && t < k.n_t) mine_b = ((unsigned long long)(unsigned int)(cqv - (M1[b] >> 4)) << 32) | (unsigned int)t;
    }
    {
      const int r = 15 - (M2[b] & 15), t = T2[b] * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (M2[b] != NONE && t < k.n_t) sec_b = ((unsigned long long)(unsigned int)(cqv - (M2[b] >> 4)) << 32) | (unsigned int)t;
    }
    const unsigned long long o1 = shfl_xor_u64(mine_b, 32), o2 = shfl_xor_u64(sec_b, 32);
    const unsigned long long m1 = mine_b < o1 ? mine_b : o1;
    const unsigned long long hi = mine_b < o1 ? o1 : mine_b;
    const unsigned long long lo2 = sec_b < o2 ? sec_b : o2;
    const unsigned long long m2 = hi < lo2 ? hi : lo2;
    if (g == 0 && j < k.n_q) {
#ifndef MATCH_NO_ATOMIC
      atomicMin(&gthr[j], 1 - bk[b]);
#endif
      unsigned long long *o = best2 + ((size_t)blockIdx.y * n_qpad + j) * 2;
      o[0] = m1; o[1] = m2;
    }
  }
}

struct QueryMid {        // per query state between the passes
  int i0, d0, dstar, pad;
  double x0, y0;
};

// fl32(d0/d) <= ratio^2, evaluated as the reference does (float quotient promoted to double)
__device__ __forceinline__ bool ratio_ok(int d0, int d, double sqmin) {
  const double ratio = (double)((float)d0 / (float)d);
  return ratio <= sqmin;
}

// grid = ceil(n_q/256), block 256.  Merges the top-2 keys of the train splits and settles every query that does not need
// pass 2.  With (d0, i0) the nearest train, D* the smallest distance passing the ratio test and (d1, t1) the smallest key over
// t != i0:
//   d1 >= D*                       no train lies below D*, and the first train at or above D* is (d1, t1): accepted with
//                                  key_ge = (d1, t1), n_lt = 0 (what pass 2 would have found)
//   d1 <  D*, t1 far from i0       a contradicting neighbour below D*: bad, rejected whatever the other trains are
//   d1 <  D*, t1 near i0           undecided: the query goes on the pass-2 list (descriptor, norm and state are copied to the
//                                  compact arrays by match_gather_kernel)
__global__ __launch_bounds__(256) void match_mid_kernel(MatchConst k, const unsigned long long *__restrict__ best2, int splits, size_t n_qpad,
                                                        const double2 *__restrict__ txy, QueryMid *__restrict__ mid,
                                                        unsigned long long *__restrict__ key_ge, unsigned long long *__restrict__ key_lt,
                                                        int *__restrict__ n_lt, int *__restrict__ bad, int *__restrict__ list2,
                                                        int *__restrict__ count2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k.n_q) return;
  unsigned long long m1 = ~0ull, m2 = ~0ull;
  for (int sp = 0; sp < splits; sp++) {
    const unsigned long long *p = best2 + ((size_t)sp * n_qpad + j) * 2;
    const unsigned long long a = p[0], b = p[1];
    if (a < m1) { m2 = m1 < b ? m1 : b; m1 = a; }      // (a <= b within a split)
    else if (a < m2) m2 = a;
  }
  QueryMid m;
  m.i0 = (int)(unsigned int)m1;
  m.d0 = (int)(m1 >> 32);
  m.pad = 0;
  int ds;
  if (m.d0 == 0) ds = 1;
  else {
    double e = ceil((double)m.d0 / k.sqminratio);
    if (e > 2.0e9) e = 2.0e9;
    ds = (int)e - 2;
    if (ds < 1) ds = 1;
    while (!ratio_ok(m.d0, ds, k.sqminratio) && ds < 2000000000) ds++;
    while (ds > 1 && ratio_ok(m.d0, ds - 1, k.sqminratio)) ds--;
  }
  m.dstar = ds;
  const double2 p = txy[m.i0];
  m.x0 = p.x; m.y0 = p.y;
  mid[j] = m;
  unsigned long long kg = ~0ull;
  int isbad = 0;
  if (m2 != ~0ull) {
    if ((int)(m2 >> 32) >= ds) kg = m2;
    else {
      const double2 q = txy[(unsigned int)m2];
      const double dx = m.x0 - q.x, dy = m.y0 - q.y;
      if (dx * dx + dy * dy > k.contr_sq) isbad = 1;
      else list2[atomicAdd(count2, 1)] = j;
    }
  }
  key_ge[j] = kg; key_lt[j] = ~0ull; n_lt[j] = 0; bad[j] = isbad;
}

// one wave per pass-2 query: descriptor row, norm and state into the compact arrays.  grid = ceil(n_q/4), block 256.
__global__ __launch_bounds__(256) void match_gather_kernel(const int *__restrict__ list2, const int *__restrict__ count2,
                                                           const int8_t *__restrict__ qdesc, const int *__restrict__ qc,
                                                           const QueryMid *__restrict__ mid, int8_t *__restrict__ qdesc2,
                                                           int *__restrict__ qc2, QueryMid *__restrict__ mid2) {
  const int n2 = *count2;
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n2; i += gridDim.x * 4) {
    const int j = list2[i];
    ((short *)(qdesc2 + (size_t)i * 128))[lane] = ((const short *)(qdesc + (size_t)j * 128))[lane];
    if (lane == 0) { qc2[i] = qc[j]; mid2[i] = mid[j]; }
  }
}

// Pass 2: FGINN reductions.  Same tiling and the same two-speed epilogue as pass 1: a tile is examined exactly
// only when one of its partial distances lies below max(D*, smallest distance >= D* found so far).
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(MATCH_NN1_VGPRS))) void match_fginn_kernel(MatchConst k, const int8_t *__restrict__ qdesc, const int *__restrict__ qc,
                                                          const int8_t *__restrict__ tdesc, const int *__restrict__ tc,
                                                          const int *__restrict__ tc2n, const unsigned int *__restrict__ tpar,
                                                          const double2 *__restrict__ txy, const QueryMid *__restrict__ mid,
                                                          unsigned long long *__restrict__ key_ge, unsigned long long *__restrict__ key_lt,
                                                          int *__restrict__ n_lt, int *__restrict__ bad,
                                                          const int *__restrict__ n_q_dev, const int *__restrict__ out_index) {
  constexpr int QB = MATCH_QB;
  // the query list of this pass is the compact list of match_mid_kernel: its length lives on the device, and results go to
  // the slots of the original queries
  // the grid is ONE row of workgroups: the few query blocks the compact list fills share it, each with as many train splits
  // as fit (a list of a few hundred queries would otherwise occupy a few workgroups for the length of the whole train list)
  k.n_q = *n_q_dev;
  const int n_qb = (k.n_q + 4 * 32 * QB - 1) / (4 * 32 * QB);
  if (n_qb == 0) return;
  const int n_tiles = (k.n_t + 31) / 32;
  int splits = max(1, min(n_tiles, (int)gridDim.x / n_qb));
  const int tps = (n_tiles + splits - 1) / splits;
  const int qb = blockIdx.x % n_qb, sp = blockIdx.x / n_qb;
  if (sp * tps >= n_tiles) return;
  const int lane = threadIdx.x & 63, g = lane >> 5;
  const int jbase = (qb * 4 + (threadIdx.x >> 6)) * (32 * QB) + (lane & 31);
  v4i bq[QB][4];
  int cq[QB], thr[QB], cnt[QB], isbad[QB], i0[QB], dstar[QB];
  double x0[QB], y0[QB];
  unsigned long long kge[QB], klt[QB];
#pragma unroll
  for (int b = 0; b < QB; b++) {
    const int j = jbase + 32 * b;
    const int jc = j < k.n_q ? j : k.n_q - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) bq[b][ks] = *(const v4i *)(qdesc + (size_t)jc * 128 + ks * 32 + g * 16);
    cq[b] = qc[jc] - 4194304;
    const QueryMid m = mid[jc];
    i0[b] = m.i0; dstar[b] = m.dstar; x0[b] = m.x0; y0[b] = m.y0;
    thr[b] = 0x7fffffff;          // until a distance >= D* has been seen every tile is examined
    kge[b] = ~0ull; klt[b] = ~0ull; cnt[b] = 0; isbad[b] = 0;
  }
  const int t0 = sp * tps;
  const int t1 = min(n_tiles, t0 + tps);
#ifdef MATCH_FGINN_SHARED
  // The four waves of a workgroup walk the same train tiles: a tile (32 x 128 B) and its 32 accumulator seeds are fetched
  // from global memory once per workgroup (one 16-byte load per thread), handed over through LDS (double buffered, one
  // barrier per tile) and read from there as MFMA operands; the next tile's loads are in flight during the MFMAs.
  __shared__ __attribute__((aligned(16))) char s_tile[2 * MT_BYTES];
  TileRegs nxt;
  if (t0 < t1) { nxt = tile_fetch(tdesc, tc2n, t0); tile_store(s_tile, nxt); }
  __syncthreads();
  for (int tt = t0; tt < t1; tt++) {
    const int tbase = tt * 32;
    const char *cur = s_tile + ((tt - t0) & 1) * MT_BYTES;
    if (tt + 1 < t1) nxt = tile_fetch(tdesc, tc2n, tt + 1);
    v4i a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) a[ks] = *(const v4i *)(cur + (lane & 31) * MT_ROW + ks * 32 + g * 16);
    const unsigned int par = tpar[tt];   // wave-uniform: ct & 1 of the 32 train rows
    v16i acc[QB];
    acc[0] = acc_seed((const int *)(cur + 32 * MT_ROW), 0, g);
#pragma unroll
    for (int b = 1; b < QB; b++) acc[b] = acc[0];
#else
  // (pass 2 takes the exact path on most tiles at these list sizes: its waves drift apart, a barrier per tile costs more than
  // the shared fetch saves, so every wave fetches its own operands)
  for (int tt = t0; tt < t1; tt++) {
    const int tbase = tt * 32;
    const int8_t *arow = tdesc + (size_t)(tbase + (lane & 31)) * 128 + g * 16;
    v4i a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) a[ks] = *(const v4i *)(arow + ks * 32);
    const unsigned int par = tpar[tt];   // wave-uniform: ct & 1 of the 32 train rows
    v16i acc[QB];
    acc[0] = acc_seed(tc2n, tbase, g);
#pragma unroll
    for (int b = 1; b < QB; b++) acc[b] = acc[0];
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int b = 0; b < QB; b++) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], bq[b][ks], acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < QB; b++) {
      if (-2 * acc_max16(acc[b]) < thr[b]) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          if (-2 * acc[b][r] >= thr[b]) continue;                 // d - cq >= -2*acc
          const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
          const int t = tbase + row;
          if (t >= k.n_t || t == i0[b]) continue;
          const int dpr = (int)((par >> row) & 1u) - 2 * acc[b][r];
          if (!(dpr < thr[b])) continue;
          const int d = dpr + cq[b];
          const unsigned long long key = ((unsigned long long)(unsigned int)d << 32) | (unsigned int)t;
          if (d >= dstar[b]) {
            if (key < kge[b]) kge[b] = key;
          } else {
            cnt[b]++;
            if (key < klt[b]) klt[b] = key;
            const double2 p = txy[t];
            const double dx = x0[b] - p.x, dy = y0[b] - p.y;
            if (dx * dx + dy * dy > k.contr_sq) isbad[b] = 1;
          }
        }
        // everything below D* always counts; at or above D* only a smaller distance than the best one matters
        // (an equal distance at a later index loses the (d, t) order)
        if (kge[b] != ~0ull) thr[b] = max(dstar[b], (int)(kge[b] >> 32)) - cq[b];
      }
      thr[b] = -max_halves(-thr[b]);   // min over the two half-waves (thr <= 0x7fffffff, so the negation is exact)
    }
#ifdef MATCH_FGINN_SHARED
    if (tt + 1 < t1) tile_store(s_tile + (((tt - t0) & 1) ^ 1) * MT_BYTES, nxt);
    __syncthreads();
#endif
  }
#pragma unroll
  for (int b = 0; b < QB; b++) {
    unsigned long long o = shfl_xor_u64(kge[b], 32); if (o < kge[b]) kge[b] = o;
    o = shfl_xor_u64(klt[b], 32); if (o < klt[b]) klt[b] = o;
    cnt[b] += __shfl_xor(cnt[b], 32);
    isbad[b] |= __shfl_xor(isbad[b], 32);
    const int jq = jbase + 32 * b;
    if (g == 0 && jq < k.n_q) {
      const int j = out_index[jq];
      if (kge[b] != ~0ull) atomicMin(&key_ge[j], kge[b]);
      if (klt[b] != ~0ull) atomicMin(&key_lt[j], klt[b]);
      if (cnt[b]) atomicAdd(&n_lt[j], cnt[b]);
      if (isbad[b]) atomicOr(&bad[j], 1);
    }
  }
}

// Decision + order-preserving compaction into the tentative list, in two launches over ceil(n_q/1024) blocks:
// the first counts the accepted queries of every block, the second adds up the counts of the blocks before it
// and writes its own tentatives at that offset (query order is the output order of the reference's loop).
__device__ __forceinline__ bool fginn_accept(const MatchConst &k, int j, const QueryMid *__restrict__ mid,
                                             const unsigned long long *__restrict__ key_ge, const unsigned long long *__restrict__ key_lt,
                                             const int *__restrict__ n_lt, const int *__restrict__ bad, mods_tentative *tc) {
  if (j >= k.n_q) return false;
  if (k.max_distance >= 0) {   // MatchFLANNDistance, matching.cpp:612-627: mid = nearest, key_ge = second nearest
    const QueryMid m = mid[j];
    if (m.d0 > k.max_distance) return false;
    const unsigned long long k2 = key_ge[j];
    tc->q = j; tc->t = m.i0; tc->t_bad = tc->t_2nd = (int)(unsigned int)k2;
    tc->d1 = (float)m.d0; tc->d2 = tc->d2nd = (float)(int)(k2 >> 32); tc->pad = 0;
    tc->ratio = (double)tc->d1 / (double)tc->d2;
    return true;
  }
  const int K = min(k.nn, k.n_t);
  const unsigned long long kg = key_ge[j];
  const int c = n_lt[j];
  if (bad[j] || kg == ~0ull || c + 1 > K - 1) return false;
  const QueryMid m = mid[j];
  const unsigned long long k2 = c > 0 ? key_lt[j] : kg;
  const int d2 = (int)(kg >> 32);
  tc->q = j; tc->t = m.i0; tc->t_bad = (int)(unsigned int)kg; tc->t_2nd = (int)(unsigned int)k2;
  tc->d1 = (float)m.d0; tc->d2 = (float)d2; tc->d2nd = (float)(int)(k2 >> 32); tc->pad = 0;
  tc->ratio = sqrt((double)((float)m.d0 / (float)d2));
  return true;
}

__global__ __launch_bounds__(1024) void match_emit_count_kernel(MatchConst k, const QueryMid *__restrict__ mid,
                                                                const unsigned long long *__restrict__ key_ge,
                                                                const unsigned long long *__restrict__ key_lt, const int *__restrict__ n_lt,
                                                                const int *__restrict__ bad, int *__restrict__ block_counts) {
  mods_tentative tc;
  const bool emit = fginn_accept(k, blockIdx.x * 1024 + threadIdx.x, mid, key_ge, key_lt, n_lt, bad, &tc);
  const int c = __syncthreads_count(emit ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}

__global__ __launch_bounds__(1024) void match_emit_kernel(MatchConst k, const QueryMid *__restrict__ mid,
                                                          const unsigned long long *__restrict__ key_ge,
                                                          const unsigned long long *__restrict__ key_lt, const int *__restrict__ n_lt,
                                                          const int *__restrict__ bad, const double2 *__restrict__ qxy,
                                                          const double2 *__restrict__ txy, const mods_region *__restrict__ qreg,
                                                          const mods_region *__restrict__ treg, const int *__restrict__ block_counts,
                                                          mods_tentative *__restrict__ out, int *__restrict__ out_count, int max_out) {
  __shared__ int s_wave[16], s_wtot[16];
  __shared__ int s_base, s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // offset of this block = accepted queries of all earlier blocks; the total fixes the packed layout of the output
  // (tentatives | correspondences | frames, see common.hpp)
  int part = 0, tot = 0;
  for (int q = tid; q < (int)gridDim.x; q += 1024) { const int c = block_counts[q]; tot += c; if (q < (int)blockIdx.x) part += c; }
  for (int off = 32; off > 0; off >>= 1) { part += __shfl_xor(part, off); tot += __shfl_xor(tot, off); }
  if (lane == 0) { s_wave[wv] = part; s_wtot[wv] = tot; }
  __syncthreads();
  if (tid == 0) { int t = 0, u = 0; for (int q = 0; q < 16; q++) { t += s_wave[q]; u += s_wtot[q]; } s_base = t; s_total = u; }
  __syncthreads();
  const int base = s_base;
  const size_t n_out = (size_t)min(s_total, max_out);
  double *u6 = (double *)((char *)out + tent_u6_off(n_out)), *laf = (double *)((char *)out + tent_laf_off(n_out));
  __syncthreads();
  mods_tentative tc;
  const bool emit = fginn_accept(k, blockIdx.x * 1024 + tid, mid, key_ge, key_lt, n_lt, bad, &tc);
  const unsigned long long mm = __ballot(emit);
  if (lane == 0) s_wave[wv] = __popcll(mm);
  __syncthreads();
  int off = base;
  for (int q = 0; q < wv; q++) off += s_wave[q];
  if (emit) {
    const int slot = off + __popcll(mm & ((1ull << lane) - 1ull));
    if (slot < max_out) {
      out[slot] = tc;
      // correspondence in the layout LORANSACFiltering hands to degensac (matching.cpp:691-713)
      const double2 a = qxy[tc.q], bpt = txy[tc.t];
      double *u = u6 + (size_t)slot * 6;
      u[0] = a.x; u[1] = a.y; u[2] = 1.; u[3] = bpt.x; u[4] = bpt.y; u[5] = 1.;
      // local affine frames of both regions for the LAF checks (matching.cpp:192-308)
      double *f = laf + (size_t)slot * 14;
      const mods_region &r1 = qreg[tc.q], &r2 = treg[tc.t];
      f[0] = r1.x; f[1] = r1.y; f[2] = r1.a11; f[3] = r1.a12; f[4] = r1.a21; f[5] = r1.a22; f[6] = r1.s;
      f[7] = r2.x; f[8] = r2.y; f[9] = r2.a11; f[10] = r2.a12; f[11] = r2.a21; f[12] = r2.a22; f[13] = r2.s;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {   // total = offset of the last block + its own count
    int t = base;
    for (int q = 0; q < 16; q++) t += s_wave[q];
    *out_count = t;
  }
}

// ---------------------------------------------------------------------------------------
static int match_target_blocks() {
  static const int v = getenv("MODS_MATCH_BLOCKS") ? std::max(1, atoi(getenv("MODS_MATCH_BLOCKS"))) : 2048;
  return v;
}
static size_t match_pad(const mods_ctx *ctx) { return ((size_t)ctx->max_cand + 127) & ~(size_t)63; }   // list stride, tile tail included

int match_ensure_buffers(mods_ctx *ctx) {
  if (ctx->m_desc) return MODS_OK;
  const size_t n = match_pad(ctx);
  MODS_HIP_CHECK(hipMalloc(&ctx->m_desc, 2 * n * 128));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_c, (4 * n + 2 * (n / 32 + 2)) * sizeof(int)));   // c of queries, trains; -floor(c/2) of both; parity words of both
  MODS_HIP_CHECK(hipMalloc(&ctx->m_regs, 2 * (size_t)ctx->max_cand * sizeof(mods_region)));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_xy, 2 * n * sizeof(double2)));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_u64, 3 * n * sizeof(unsigned long long)));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_int, (2 * n + n / 1024 + 2) * sizeof(int)));   // n_lt, bad, per-block counts of the compaction
  MODS_HIP_CHECK(hipMalloc(&ctx->m_mid, n * sizeof(QueryMid)));
  // pass-1 top-2 table (one pair of keys per query and train split; splits * query blocks <= target blocks + query blocks) and
  // the pass-2 subset: list | count | norms | descriptors | state
  ctx->m_best2_cap = (size_t)match_target_blocks() * 128 * MATCH_QB1 + n + 128 * MATCH_QB1;
  MODS_HIP_CHECK(hipMalloc(&ctx->m_p2, ctx->m_best2_cap * 16 + n * (3 * sizeof(int) + 128 + sizeof(QueryMid)) + 128));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_tent, tent_bytes(n) + 64));
  // the tentative count lives in pinned host memory: the emit kernel's single store lands there, the host reads it after a
  // stream synchronisation - no 4-byte copy launch per search
  MODS_HIP_CHECK(hipHostMalloc(&ctx->m_count, 64 * sizeof(int)));   // [0]: the last search; [i]: pair i of a batch (m_count_out)
  MODS_HIP_CHECK(hipMemsetAsync(ctx->m_desc, 0, 2 * n * 128, ctx->stream));
  MODS_HIP_CHECK(hipMemsetAsync(ctx->m_c, 0, 4 * n * sizeof(int), ctx->stream));
  return MODS_OK;
}

// One launch instead of four memsets per search: tentative counter, pass-2 list counter, the parity words the pack kernels OR
// into, and the shared bounds of pass 1 (0x7f7f7f7f = none).  grid = ceil(max(n_q, n_t) / 256), block 256
__global__ __launch_bounds__(256) void match_init_kernel(int n_q, int n_t, int *__restrict__ m_count, int *__restrict__ count2,
                                                         unsigned int *__restrict__ qpar, unsigned int *__restrict__ tpar,
                                                         int *__restrict__ gthr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { *m_count = 0; *count2 = 0; }
  if (i < n_q) gthr[i] = 0x7f7f7f7f;
  if (i <= n_q / 32) qpar[i] = 0u;
  if (i <= n_t / 32) tpar[i] = 0u;
}

// queries / trains: device region lists with host-known sizes n_q, n_t.
int match_run(mods_ctx *ctx, const mods_region *q_dev, int n_q, const mods_region *t_dev, int n_t, double ratio,
              double contradDist, int nn) {
  int rc = match_ensure_buffers(ctx);
  if (rc) return rc;
  if (n_q > ctx->max_cand || n_t > ctx->max_cand) { set_error("match: list larger than the context capacity"); return MODS_E_CAPACITY; }
  MatchConst k;
  k.n_q = n_q; k.n_t = n_t; k.nn = nn;
  k.max_distance = -1;
  k.sqminratio = ratio * ratio;
  k.contr_sq = contradDist * contradDist;
  if (!(k.sqminratio < 1.0)) { set_error("FGINN ratio >= 1 (all-neighbours mode) is not supported"); return MODS_E_ARG; }
  // where the packed list and its length go: the context's own buffer / counter, or what a batch of pairs set (capi.hip: match_pairs)
  mods_tentative *tent_out = ctx->m_tent_out ? ctx->m_tent_out : ctx->m_tent;
  int *count_out = ctx->m_count_out ? ctx->m_count_out : ctx->m_count;
  if (n_q == 0 || n_t == 0) { MODS_HIP_CHECK(hipMemsetAsync(count_out, 0, sizeof(int), ctx->stream)); return MODS_OK; }
  const size_t n = match_pad(ctx);
  int8_t *qd = ctx->m_desc, *td = ctx->m_desc + n * 128;
  int *qc = ctx->m_c, *tc = ctx->m_c + n, *qc2 = ctx->m_c + 2 * n, *tc2 = ctx->m_c + 3 * n;
  unsigned int *qpar = (unsigned int *)(ctx->m_c + 4 * n), *tpar = qpar + n / 32 + 2;
  double2 *qxy = (double2 *)ctx->m_xy, *txy = (double2 *)ctx->m_xy + n;
  unsigned long long *best = ctx->m_u64, *key_ge = ctx->m_u64 + n, *key_lt = ctx->m_u64 + 2 * n;
  int *n_lt = ctx->m_int, *bad = ctx->m_int + n;
  StageScope ts(ctx, MODS_STAGE_MATCH);
  const int n_tiles = (n_t + 31) / 32;
  const int target_blocks = match_target_blocks();
  // carve of m_p2 (every part 16-byte aligned)
  unsigned long long *best2 = (unsigned long long *)ctx->m_p2;
  QueryMid *mid2 = (QueryMid *)(best2 + 2 * ctx->m_best2_cap);
  int8_t *qd2 = (int8_t *)(mid2 + n);
  int *list2 = (int *)(qd2 + n * 128), *qcs = list2 + n, *count2 = qcs + n;
  int *gthr = count2 + 16;
  static const int dbg_mask = getenv("MODS_MATCH_MASK") ? atoi(getenv("MODS_MATCH_MASK")) : 63;   // DEBUG bisect
  if (dbg_mask & 1) hipLaunchKernelGGL(match_init_kernel, dim3((std::max(n_q, n_t) + 255) / 256), dim3(256), 0, ctx->stream, n_q, n_t, count_out, count2, qpar, tpar, gthr);
  if (dbg_mask & 1) hipLaunchKernelGGL(match_pack_kernel, dim3(std::min(2048, (n_q + 3) / 4)), dim3(256), 0, ctx->stream, q_dev, (const int *)nullptr, n_q, qd, qc, qc2, qpar, qxy, ctx->max_cand);
  if (dbg_mask & 1) hipLaunchKernelGGL(match_pack_kernel, dim3(std::min(2048, (n_t + 3) / 4)), dim3(256), 0, ctx->stream, t_dev, (const int *)nullptr, n_t, td, tc, tc2, tpar, txy, ctx->max_cand);
  (void)best;
  // pass 1: top-2 keys per query and train split
  constexpr int QPB = 32 * NN1_WAVES * MATCH_QB1;      // queries per pass-1 workgroup
  const int qblocks1 = (n_q + QPB - 1) / QPB;
  int splits1 = std::max(1, std::min(n_tiles, target_blocks * (4 / NN1_WAVES) / std::max(1, qblocks1)));
  MatchConst k1 = k;
  k1.tiles_per_split = (n_tiles + splits1 - 1) / splits1;
  splits1 = (n_tiles + k1.tiles_per_split - 1) / k1.tiles_per_split;
  const size_t n_qpad = (size_t)qblocks1 * QPB;
  if ((size_t)splits1 * n_qpad > ctx->m_best2_cap) { set_error("match: top-2 table too small"); return MODS_E_CAPACITY; }
  if (dbg_mask & 2) hipLaunchKernelGGL(match_nn1_kernel, dim3(qblocks1, splits1), dim3(NN1_THREADS), 0, ctx->stream, k1, qd, qc, td, tc, tc2, tpar, best2, gthr);
  if (dbg_mask & 4) hipLaunchKernelGGL(match_mid_kernel, dim3((n_q + 255) / 256), dim3(256), 0, ctx->stream, k, best2, splits1, n_qpad, txy, (QueryMid *)ctx->m_mid,
                     key_ge, key_lt, n_lt, bad, list2, count2);
  // pass 2 on the undecided queries only (their number stays on the device: the grid covers the worst case, idle blocks exit)
  if (dbg_mask & 4) hipLaunchKernelGGL(match_gather_kernel, dim3(std::min(1024, (n_q + 3) / 4)), dim3(256), 0, ctx->stream, list2, count2, qd, qc,
                     (const QueryMid *)ctx->m_mid, qd2, qcs, mid2);
  {
    const int qblocks = (n_q + 128 * MATCH_QB - 1) / (128 * MATCH_QB);
    if (dbg_mask & 8) hipLaunchKernelGGL(match_fginn_kernel, dim3(std::max(target_blocks, qblocks)), dim3(256), 0, ctx->stream, k, qd2, qcs, td, tc, tc2, tpar, txy,
                       (const QueryMid *)mid2, key_ge, key_lt, n_lt, bad, count2, list2);
  }
  const int eblocks = (n_q + 1023) / 1024;
  int *block_counts = (int *)(ctx->m_int + 2 * n);
  if (dbg_mask & 16) hipLaunchKernelGGL(match_emit_count_kernel, dim3(eblocks), dim3(1024), 0, ctx->stream, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, block_counts);
  if (dbg_mask & 16) hipLaunchKernelGGL(match_emit_kernel, dim3(eblocks), dim3(1024), 0, ctx->stream, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, qxy, txy, q_dev, t_dev, block_counts, tent_out, count_out, ctx->max_cand);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// ---------------------------------------------------------------------------------------
// MatchFLANNDistance (matching.cpp:572-633) with the default binary_dist = Hamming and an exact (linear) index: the two
// nearest trains of every query by Hamming distance over the 128 descriptor bytes, ascending distance then ascending index
// (FLANN's KNNSimpleResultSet over a linear scan).  The packed lists hold v - 128 per byte; the XOR of two such bytes is
// the XOR of the values.  Thread = query (32 dwords in registers), the block walks the trains in LDS tiles of 64.
// grid = ceil(n_q / 256), block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hamming_nn2_kernel(MatchConst k, const int8_t *__restrict__ qd, const int8_t *__restrict__ td,
                                                          QueryMid *__restrict__ mid, unsigned long long *__restrict__ key2) {
  __shared__ uint32_t s_t[64][33];   // 33: the per-thread walk over a row is conflict free, the staging is coalesced
  const int j = blockIdx.x * 256 + threadIdx.x;
  uint32_t q[32];
  const uint32_t *qp = (const uint32_t *)(qd + (size_t)min(j, k.n_q - 1) * 128);
#pragma unroll
  for (int e = 0; e < 32; e++) q[e] = qp[e];
  unsigned long long k1 = ~0ull, k2 = ~0ull;
  for (int t0 = 0; t0 < k.n_t; t0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
      const int r = e >> 5, c = e & 31;
      s_t[r][c] = t0 + r < k.n_t ? ((const uint32_t *)(td + (size_t)(t0 + r) * 128))[c] : 0u;
    }
    __syncthreads();
    const int nt = min(64, k.n_t - t0);
    for (int r = 0; r < nt; r++) {
      int d = 0;
#pragma unroll
      for (int e = 0; e < 32; e++) d += __popc(q[e] ^ s_t[r][e]);
      const unsigned long long key = ((unsigned long long)(unsigned int)d << 32) | (unsigned int)(t0 + r);
      if (key < k1) { k2 = k1; k1 = key; }
      else if (key < k2) k2 = key;
    }
  }
  if (j < k.n_q) {
    QueryMid m;
    m.i0 = (int)(unsigned int)k1; m.d0 = (int)(k1 >> 32); m.dstar = 0; m.pad = 0; m.x0 = 0; m.y0 = 0;
    mid[j] = m;
    key2[j] = k2 == ~0ull ? (((unsigned long long)2147483647u << 32) | 0xffffffffull) : k2;   // a single train: distance INT_MAX, index -1
  }
}

int match_run_distance(mods_ctx *ctx, const mods_region *q_dev, int n_q, const mods_region *t_dev, int n_t, double threshold) {
  int rc = match_ensure_buffers(ctx);
  if (rc) return rc;
  if (n_q > ctx->max_cand || n_t > ctx->max_cand) { set_error("match: list larger than the context capacity"); return MODS_E_CAPACITY; }
  MatchConst k;
  memset(&k, 0, sizeof(k));
  k.n_q = n_q; k.n_t = n_t; k.nn = 2;
  k.max_distance = (int)(float)threshold;                       // int max_distance = (int)float(par.matchDistanceThreshold)
  if (k.max_distance < 0) { set_error("match: negative distance threshold"); return MODS_E_ARG; }
  MODS_HIP_CHECK(hipMemsetAsync(ctx->m_count, 0, sizeof(int), ctx->stream));
  if (n_q == 0 || n_t == 0) return MODS_OK;
  const size_t n = match_pad(ctx);
  int8_t *qd = ctx->m_desc, *td = ctx->m_desc + n * 128;
  int *qc = ctx->m_c, *tc = ctx->m_c + n, *qc2 = ctx->m_c + 2 * n, *tc2 = ctx->m_c + 3 * n;
  unsigned int *qpar = (unsigned int *)(ctx->m_c + 4 * n), *tpar = qpar + n / 32 + 2;
  MODS_HIP_CHECK(hipMemsetAsync(qpar, 0, sizeof(unsigned int) * 2 * (n / 32 + 2), ctx->stream));
  double2 *qxy = (double2 *)ctx->m_xy, *txy = (double2 *)ctx->m_xy + n;
  unsigned long long *key_ge = ctx->m_u64 + n, *key_lt = ctx->m_u64 + 2 * n;
  int *n_lt = ctx->m_int, *bad = ctx->m_int + n;
  StageScope ts(ctx, MODS_STAGE_MATCH);
  hipLaunchKernelGGL(match_pack_kernel, dim3(std::min(2048, (n_q + 3) / 4)), dim3(256), 0, ctx->stream, q_dev, (const int *)nullptr, n_q, qd, qc, qc2, qpar, qxy, ctx->max_cand);
  hipLaunchKernelGGL(match_pack_kernel, dim3(std::min(2048, (n_t + 3) / 4)), dim3(256), 0, ctx->stream, t_dev, (const int *)nullptr, n_t, td, tc, tc2, tpar, txy, ctx->max_cand);
  hipLaunchKernelGGL(hamming_nn2_kernel, dim3((n_q + 255) / 256), dim3(256), 0, ctx->stream, k, qd, td, (QueryMid *)ctx->m_mid, key_ge);
  const int eblocks = (n_q + 1023) / 1024;
  int *block_counts = (int *)(ctx->m_int + 2 * n);
  hipLaunchKernelGGL(match_emit_count_kernel, dim3(eblocks), dim3(1024), 0, ctx->stream, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, block_counts);
  hipLaunchKernelGGL(match_emit_kernel, dim3(eblocks), dim3(1024), 0, ctx->stream, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, qxy, txy, q_dev, t_dev, block_counts, ctx->m_tent, ctx->m_count, ctx->max_cand);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods

#ifdef MATCH_STATS
// development aid: (query block, tile) pairs visited / taken through the exact path by match_nn1_kernel since the last call
extern "C" void mods_debug_match_stats(unsigned long long out[2]) {
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mods::g_match_stats), 16);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(mods::g_match_stats), z, 32);
}
#endif
