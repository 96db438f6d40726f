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
// Pass 1 finds the two nearest trains (d0, i0), (d1, t1) of every query - a branch-free MFMA sweep that keeps the three best
// half-tile maxima per query and train split, and an exact finish over the few half tiles that can hold them; most queries
// are settled by those two keys (match_mid_kernel), the rest go through pass 2, which accumulates {any inconsistent below D*,
// count below D*, min key at/above D*, min key below D*} per query.
//
// Distances are exact integers: descriptors are offset to int8 (v - 128), the contraction runs
// on v_mfma_i32_32x32x32_i8 (i32 accumulate) and d = cq + ct - 2*dot with precombined norms (the same integers whether the dot
// product comes from the matrix cores or from v_dot4 in the exact finish).
// Trains are the MFMA rows (streamed), queries the columns (resident in registers), so every
// lane reduces its 16 results per tile into per-query running values.
#include "common.hpp"
#include <type_traits>

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

// The searches of one grouped launch: blockIdx.y (blockIdx.z in the pack kernel) = search.  Every scratch buffer of the context
// exists once per search of a group ("set"): set j = set 0 + j * stride, so a kernel takes the pointers of set 0 and moves them.
// The pairs of a pipeline batch are matched in ONE set of launches (a 10 k x 9 k search fills 40 of the 256 CUs on its own, and
// every dispatch costs the host a completion interrupt); a single search is a group of one.
constexpr int MATCH_MAX_JOBS = 16;
struct MatchJobs {
  int n_jobs;
  int n_q[MATCH_MAX_JOBS], n_t[MATCH_MAX_JOBS];
  int tps[MATCH_MAX_JOBS], qblocks[MATCH_MAX_JOBS], splits[MATCH_MAX_JOBS];     // pass-1 geometry (nn1_grid)
  int eblocks[MATCH_MAX_JOBS];                                                   // blocks of the emit stage
  const mods_region *q_reg[MATCH_MAX_JOBS], *t_reg[MATCH_MAX_JOBS];
  mods_tentative *tent_out[MATCH_MAX_JOBS];
  int *count_out[MATCH_MAX_JOBS];
  size_t s_desc, s_p2;                       // set strides in bytes (m_desc, m_p2)
  size_t s_c, s_xy, s_u64, s_int, s_mid;     // set strides in elements of the buffer's type (m_c, m_xy, m_u64, m_int, m_mid)
};
template <class T> __device__ __forceinline__ T *set_el(T *p, int job, size_t stride) { return p + (size_t)job * stride; }
template <class T> __device__ __forceinline__ T *set_by(T *p, int job, size_t stride_bytes) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type C;
  return (T *)((C *)p + (size_t)job * stride_bytes);
}

// Accumulator seed of a train row: acc = dot - floor(ct/2) + MATCH_BIAS.  dot lies in [-2^21, 2^21] and ct in [2^21, 2^22], so the
// seeded accumulators of real rows lie in [1, 5*2^20 + 1] (23 bits, never negative); the rows past the end of the list inside the last
// tile carry zero descriptors and the seed 0: their accumulators are exactly 0, below every real one.
constexpr int MATCH_BIAS = (1 << 22) + 1;

// Packs one region list for the matcher: int8 descriptors (v-128), c = sum v^2 - 256*sum(v-128),
// centre coordinates.  grid = ceil(n/4), block = 256 (one wave per region).
__global__ __launch_bounds__(256) void match_pack_kernel(const mods_region *__restrict__ reg, const int *__restrict__ count_ptr,
                                                         int count_fixed, int8_t *__restrict__ desc, int *__restrict__ cvec,
                                                         int *__restrict__ c2neg, unsigned int *__restrict__ parity,
                                                         double2 *__restrict__ xy, int max_n) {
  int n = count_ptr ? *count_ptr : count_fixed;
  if (n > max_n) n = max_n;
  const int lane = threadIdx.x & 63;
  // rows of the last tile beyond the list: zero descriptor, zero seed (stale entries there must not shadow the valid rows)
  if (blockIdx.x == 0) {
    const int n32 = (n + 31) & ~31;
    for (int e = threadIdx.x; e < (n32 - n) * 32; e += 256) ((int *)(desc + (size_t)n * 128))[e] = 0;
    if (threadIdx.x < 32 && n + (int)threadIdx.x < n32) c2neg[n + threadIdx.x] = 0;
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
      c2neg[i] = MATCH_BIAS - (c >> 1);       // accumulator seed of the distance tiles: acc = dot - floor(c/2) + MATCH_BIAS
      if (c & 1) atomicOr(&parity[i >> 5], 1u << (i & 31));   // c & 1 of the 32 rows of a tile in one word (zeroed by the caller)
      xy[i] = make_double2(reg[i].x, reg[i].y);
    }
  }
}

// Both lists of a search in ONE launch (blockIdx.y = list), which also does what match_init_kernel did: a wave packs 32
// consecutive regions - one tile of the distance kernels, two lanes per region - and WRITES the tile's parity word (the single-list kernel ORs bits into
// words that a launch before it had to clear), block (0, 0) clears the two counters of the search.  Three dispatches fewer per search.
// grid = (ceil(max(n_q, n_t) / 32 / 4), 2, searches), block 256
struct PackList { const mods_region *reg; int n; int8_t *desc; int *cvec; int *c2neg; unsigned int *parity; double2 *xy; };
__global__ __launch_bounds__(256) void match_pack2_kernel(MatchJobs J, PackList lq, PackList lt, int max_n, int *__restrict__ count2) {
  const int job = blockIdx.z;
  PackList L = blockIdx.y == 0 ? lq : lt;       // (reg and n of the lists come from the job table)
  L.reg = blockIdx.y == 0 ? J.q_reg[job] : J.t_reg[job];
  L.n = blockIdx.y == 0 ? J.n_q[job] : J.n_t[job];
  L.desc = set_by(L.desc, job, J.s_desc); L.cvec = set_el(L.cvec, job, J.s_c); L.c2neg = set_el(L.c2neg, job, J.s_c);
  L.parity = set_el(L.parity, job, J.s_c); L.xy = set_el(L.xy, job, J.s_xy);
  const int n = min(L.n, max_n);
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { *J.count_out[job] = 0; *set_by(count2, job, J.s_p2) = 0; }
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile * 32 >= n) return;
  // two lanes per region (64 descriptor bytes each, four 16-byte loads), the 32 regions of the tile side by side: no loop over rows
  const int i = tile * 32 + (lane >> 1), half = lane & 1;
  unsigned int n2 = 0, sb = 0;
  uint4 *dst = (uint4 *)(L.desc + (size_t)i * 128 + 64 * half);
  if (i < n) {
    const uint4 *src = (const uint4 *)(L.reg[i].desc + 64 * half);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint4 v = src[q];
      n2 = __builtin_amdgcn_udot4(v.x, v.x, n2, false); n2 = __builtin_amdgcn_udot4(v.y, v.y, n2, false);
      n2 = __builtin_amdgcn_udot4(v.z, v.z, n2, false); n2 = __builtin_amdgcn_udot4(v.w, v.w, n2, false);
      sb = __builtin_amdgcn_udot4(v.x, 0x01010101u, sb, false); sb = __builtin_amdgcn_udot4(v.y, 0x01010101u, sb, false);
      sb = __builtin_amdgcn_udot4(v.z, 0x01010101u, sb, false); sb = __builtin_amdgcn_udot4(v.w, 0x01010101u, sb, false);
      dst[q] = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);   // v - 128 as int8
    }
  } else {
    // rows of the last tile beyond the list: zero descriptor, zero seed (stale entries there must not shadow the valid rows)
#pragma unroll
    for (int q = 0; q < 4; q++) dst[q] = make_uint4(0u, 0u, 0u, 0u);
    if (half == 0) L.c2neg[i] = 0;
  }
  n2 += __shfl_xor(n2, 1); sb += __shfl_xor(sb, 1);
  const int c = (int)n2 - 256 * ((int)sb - 128 * 128);        // sum v^2 - 256 * sum (v - 128)
  if (i < n && half == 0) {
    L.cvec[i] = c;
    L.c2neg[i] = MATCH_BIAS - (c >> 1);       // accumulator seed of the distance tiles: acc = dot - floor(c/2) + MATCH_BIAS
    L.xy[i] = make_double2(L.reg[i].x, L.reg[i].y);
  }
  // parity word of the tile: bit r = c & 1 of row r (the even lanes' votes, squeezed together)
  unsigned long long m = __ballot(i < n && half == 0 && (c & 1));
  m = (m | (m >> 1)) & 0x3333333333333333ull;
  m = (m | (m >> 2)) & 0x0f0f0f0f0f0f0f0full;
  m = (m | (m >> 4)) & 0x00ff00ff00ff00ffull;
  m = (m | (m >> 8)) & 0x0000ffff0000ffffull;
  m = (m | (m >> 16)) & 0x00000000ffffffffull;
  if (lane == 0) L.parity[tile] = (unsigned int)m;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return ((unsigned long long)hi << 32) | lo;
}

// Pass-1 design.  A 32x32x128 tile costs 4 MFMAs (~128 clk of the matrix pipe) and leaves 16 results per lane: the distances of
// ONE query (the lane's column) to 16 of the tile's 32 trains (a "half tile": rows (r&3) + 8*(r>>2) + 4*(lane>>5)).  With the seeded
// accumulators acc = dot - floor(ct/2) + MATCH_BIAS the squared distance is d = cq + (ct & 1) - 2*(acc - MATCH_BIAS): a larger
// accumulator is a strictly nearer train (the parity term only orders equal accumulators).  Nothing exact happens inside the tile
// loop: a lane keeps the THREE largest half-tile maxima it has seen, as packed 32-bit keys
//     key = max16(acc) << 9 | (255 - tile index inside the train split) << 1 | (1 - half)
// by v_max3 x 8, one v_lshl_or, v_med3 x 2 and v_max per query block and tile - no branch, no ballot, no exchange between waves or
// workgroups.  The two nearest trains of a query then lie in half tiles whose maximum is at least the second largest maximum A
// over all half tiles (two different half tiles hold an accumulator >= A, i.e. two trains strictly nearer than anything in a half
// tile whose maximum is below A), so match_fix_kernel recomputes exactly the half tiles with maximum >= A - two of them unless
// maxima tie - and a (query, split) stream whose THIRD key also reaches A may have dropped such a half tile: that query is
// rescanned over all trains (the exact fallback; three half tiles of one stream tying with the runner-up needs repeated descriptors).
// A wave keeps QB query blocks resident (QB x 4 B-operand registers): every A tile read from LDS feeds QB x 4 MFMAs.
constexpr int MATCH_QB = 2;        // pass 2 (its exact path is taken per wave: fewer queries per wave, fewer exact tiles)
#ifndef MATCH_QB_NN1
#define MATCH_QB_NN1 4   // measured (round 4, profiles/r04_match_variants.log): 2 blocks at 5 waves per SIMD 318 us, 3 blocks (spills) 349, 4 blocks at 3 waves 284
#endif
constexpr int MATCH_QB1 = MATCH_QB_NN1;   // pass 1: 16 MFMAs per train tile and wave (half the LDS reads and tile fetches per MFMA of 2 blocks)
constexpr int NN1_MAX_TPS = 256;          // tiles per train split: the tile index inside the split has 8 bits of the key

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
// max of a value over the two half-waves (lane l and lane l ^ 32) in one VALU instruction: v_permlane32_swap exchanges the upper
// half of one copy with the lower half of the other, so both copies then hold both halves' values lane by lane (the shuffle
// builtin goes through ds_bpermute: eight address instructions and an LDS round trip in every tile of pass 2)
__device__ __forceinline__ int max_halves(int x) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  return max((int)r[0], (int)r[1]);
}
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned b, unsigned c) {
  unsigned o;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}
// keeps (m1 >= m2 >= m3) the three largest keys seen
__device__ __forceinline__ void top3_insert(unsigned &m1, unsigned &m2, unsigned &m3, unsigned x) {
  m3 = med3_u32(m2, m3, x);
  m2 = med3_u32(m1, m2, x);
  m1 = max(m1, x);
}

// Co-residency rule (DESIGN.md "The matcher and its neighbours").  A wave that issues independent MFMAs (several accumulator
// chains) makes double-precision VALU results of OTHER waves on the same SIMD go wrong - waves of other streams and contexts
// included; tools/ubench/mfma_aggr.hip reproduces it with nothing but MFMAs next to the fp64 victim of tools/ubench/spin_victim.hip
// (10^5..10^9 wrong rounds per second, whatever the operand data; none with ONE dependent chain per wave, none when the MFMA kernel
// allocates the whole register file of its SIMDs).  So the two matrix-core kernels of this file never share a SIMD with foreign
// waves: a workgroup of either pass is four waves of 512 registers (one per SIMD, the whole file); the allocation is forced by touching the last register (nn1_own_simd / fginn_own_simd), and
// tests/test_cpu_host.py::test_matrix_core_kernels_own_their_simds reads it back from the built library.
__device__ __forceinline__ void own_simd_256() { asm volatile("v_mov_b32 v255, 0" ::: "v255"); }
__device__ __forceinline__ void own_simd_512() { asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255"); }

#ifndef NN1_NWAVES
#define NN1_NWAVES 8    // measured: 4 waves of 512 registers (one per SIMD) 310 us, 8 waves of 256 (two per SIMD) 270 us at 60 156 x 47 177
#endif
constexpr int NN1_WAVES = NN1_NWAVES, NN1_THREADS = 64 * NN1_WAVES;
// The four waves of a pass-1 workgroup - one per SIMD, each owning its SIMD - walk the same train tiles.  A tile (32 rows x 128 B)
// and its 32 accumulator seeds go from global memory straight into LDS (global_load_lds: no staging registers, no ds_write),
// NN1_PF tiles ahead of the MFMAs in a ring of NN1_PF + 1 images.
// LDS image of a tile: the DMA writes a wave's 64 x 16 bytes linearly (8 rows), so rows are 128 bytes apart and the bank
// conflicts of the operand reads (lane = row, same 16-byte column) are removed by a swizzle on the SOURCE side: the 16-byte
// chunk c of row r sits at chunk position c ^ ((r >> 1) & 7); the 16 rows of a ds_read_b128 lane group then fall into 16
// different 16-byte slots of the 256-byte bank row.
// With one wave per SIMD nothing else hides a wave's own latencies, so the tile loop is software-pipelined inside the wave: the 16
// MFMAs of tile i (4 query blocks x 4 K slices, into one of two accumulator sets) are interleaved with the epilogue of tile i - 1
// on the other set (3 VALU instructions per MFMA), the LDS reads of tile i + 1's operands (a K slice's registers are refilled
// once its four MFMAs have been issued) and the DMA of tile i + NN1_PF; one barrier per tile.
#ifndef NN1_PF
#define NN1_PF 3
#endif
constexpr int NN1_NBUF = NN1_PF + 1;
constexpr int NN1_IMG = 4096 + 128;       // rows | seeds
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ void nn1_dma_tile(const int8_t *__restrict__ tdesc, const int *__restrict__ tc2n, int tt, char *img, int w, int lane) {
  const int r = 8 * w + (lane >> 3);
  __builtin_amdgcn_global_load_lds((gptr_t)(tdesc + (size_t)tt * 4096 + r * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4)), (lptr_t)(img + w * 1024), 16, 0, 0);
  if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(tc2n + tt * 32 + w * 8 + lane), (lptr_t)(img + 4096 + w * 32), 4, 0, 0);
}
struct Nn1State {          // what a wave carries through the tile loop besides the accumulators
  v4i bq[MATCH_QB1][4];    // query operands (resident)
  v4i a[4];                // train operands of the tile being multiplied (refilled slice by slice for the next one)
  v16i seed;               // accumulator seeds of that tile
  unsigned M1[MATCH_QB1], M2[MATCH_QB1], M3[MATCH_QB1];
  int aoff[4];             // this lane's operand address inside an image: row lane & 31, chunk 2 * ks + g at its swizzled position
  int soff;                // ... and of its first seed
};
// one epilogue slice: three of the twelve VALU instructions that fold block b's accumulators of the previous tile into its keys
template <int PART>
__device__ __forceinline__ void nn1_epi_slice(const v16i &acc, int &m, unsigned &M1, unsigned &M2, unsigned &M3, unsigned idc) {
  if (PART == 0) { m = max(max(acc[0], acc[1]), acc[2]); m = max(max(m, acc[3]), acc[4]); m = max(max(m, acc[5]), acc[6]); }
  if (PART == 1) { m = max(max(m, acc[7]), acc[8]); m = max(max(m, acc[9]), acc[10]); m = max(max(m, acc[11]), acc[12]); }
  if (PART == 2) { m = max(max(m, acc[13]), acc[14]); m = max(m, acc[15]); const unsigned x = ((unsigned)m << 9) | idc; M3 = med3_u32(M2, M3, x); m = (int)x; }
  if (PART == 3) { const unsigned x = (unsigned)m; M2 = med3_u32(M1, M2, x); M1 = max(M1, x); }
}
// Tile step of tile i (image RING = i mod 4):
//   head   tile i + 1 has landed for everybody behind the barrier (a wave's two DMA instructions per tile complete in order: all but
//          the NN1_PF - 2 youngest tiles), and the image of tile i - 1 - whose last readers passed the previous barrier - takes tile
//          i + NN1_PF (GUARD: near the end of the split there is nothing left to fetch and the wait is for everything);
//   body   multiplies tile i into accW while (EPI) folding tile i - 1 from accR, and refills the operand registers from the image
//          of tile i + 1 (read whatever it holds after the last tile: unused).
template <bool EPI, int RING, bool GUARD>
__device__ __forceinline__ void nn1_step(Nn1State &st, v16i (&accW)[MATCH_QB1], const v16i (&accR)[MATCH_QB1], char *s_ring, const int8_t *__restrict__ tdesc,
                                         const int *__restrict__ tc2n, int tile, int i, int n, int w, int lane, unsigned idc_prev) {
  constexpr int QB = MATCH_QB1;
  if (!GUARD || i - 1 + NN1_PF < n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * (NN1_PF - 2)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (w < 4 && (!GUARD || i + NN1_PF < n)) nn1_dma_tile(tdesc, tc2n, tile + NN1_PF, s_ring + ((RING + NN1_PF) % NN1_NBUF) * NN1_IMG, w, lane);
  const char *nxt = s_ring + ((RING + 1) % NN1_NBUF) * NN1_IMG;
  int m[QB];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
#pragma unroll
    for (int b = 0; b < QB; b++) {
      accW[b] = ks == 0 ? __builtin_amdgcn_mfma_i32_32x32x32_i8(st.a[0], st.bq[b][0], st.seed, 0, 0, 0)
                        : __builtin_amdgcn_mfma_i32_32x32x32_i8(st.a[ks], st.bq[b][ks], accW[b], 0, 0, 0);
      if (EPI) {      // block ks of the previous tile, slice b
        if (b == 0) nn1_epi_slice<0>(accR[ks], m[ks], st.M1[ks], st.M2[ks], st.M3[ks], idc_prev);
        if (b == 1) nn1_epi_slice<1>(accR[ks], m[ks], st.M1[ks], st.M2[ks], st.M3[ks], idc_prev);
        if (b == 2) nn1_epi_slice<2>(accR[ks], m[ks], st.M1[ks], st.M2[ks], st.M3[ks], idc_prev);
        if (b == 3) nn1_epi_slice<3>(accR[ks], m[ks], st.M1[ks], st.M2[ks], st.M3[ks], idc_prev);
      }
    }
  }
  // all MFMAs of the tile have been issued: refill the operand registers for the next one.  (Refills between the rounds would
  // save nothing - the registers are not needed before the next step - and the compiler's wait for the LAST operand, which it
  // writes as lgkmcnt(0), would then also wait for the reads issued just before it: ~100 exposed cycles per tile.)  Their latency
  // runs under the tail of the matrix pipe, the barrier and the DMA issue of the next step.
#pragma unroll
  for (int ks = 0; ks < 4; ks++) st.a[ks] = *(const v4i *)(nxt + st.aoff[ks]);
  st.seed = acc_seed((const int *)(nxt + 4096 + st.soff), 0, 0);
  // the order above, pinned: per MFMA three VALU instructions of the epilogue, the LDS reads at the end
#pragma unroll
  for (int j = 0; j < 4 * QB; j++) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
}
// Pass 1.  grid = qblocks * splits workgroups of 256 threads; best3: [split][n_qpad] (M1, M2, M3, 0), n_qpad = qblocks * 128 * QB
// (plain 16-byte stores: every (split, query) has one owner).
// Workgroup -> (query block, split): with a split count that is a multiple of 8, the workgroups that the dispatcher places on one
// XCD (linear id mod 8) share that XCD's train splits (split mod 8 = XCD), so an XCD's L2 holds one eighth of the train list
// instead of all of it; any other placement is only slower.
__global__ __launch_bounds__(NN1_THREADS) __attribute__((amdgpu_waves_per_eu(NN1_WAVES / 4, NN1_WAVES / 4))) void match_nn1_kernel(MatchJobs J, MatchConst k,
                                                        const int8_t *__restrict__ qdesc, const int8_t *__restrict__ tdesc,
                                                        const int *__restrict__ tc2n, uint4 *__restrict__ best3) {
  constexpr int QB = MATCH_QB1;
  const int job = blockIdx.y;
  const int qblocks = J.qblocks[job], splits = J.splits[job];
  if ((int)blockIdx.x >= qblocks * splits) return;        // (the grid is sized for the largest search of the group)
  k.n_q = J.n_q[job]; k.n_t = J.n_t[job]; k.tiles_per_split = J.tps[job];
  qdesc = set_by(qdesc, job, J.s_desc); tdesc = set_by(tdesc, job, J.s_desc); tc2n = set_el(tc2n, job, J.s_c); best3 = set_by(best3, job, J.s_p2);
  static_assert(QB == 4 && NN1_NBUF == 4, "nn1_step is laid out for four query blocks and a ring of four images");
  if (NN1_WAVES == 4) own_simd_512(); else own_simd_256();
  const int lane = threadIdx.x & 63, g = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int qb, sp;
  if ((splits & 7) == 0) { const int x = blockIdx.x & 7, idx = blockIdx.x >> 3; sp = (idx / qblocks) * 8 + x; qb = idx % qblocks; }
  else { sp = blockIdx.x / qblocks; qb = blockIdx.x % qblocks; }
  const int n_tiles = (k.n_t + 31) / 32;
  const int t0 = sp * k.tiles_per_split;
  const int n = max(0, min(n_tiles, t0 + k.tiles_per_split) - t0);      // tiles of this workgroup
  __shared__ __attribute__((aligned(16))) char s_ring[NN1_NBUF * NN1_IMG];
#pragma unroll
  for (int d = 0; d < NN1_PF; d++)
    if (w < 4 && d < n) nn1_dma_tile(tdesc, tc2n, t0 + d, s_ring + d * NN1_IMG, w, lane);
  const int jbase = (qb * NN1_WAVES + w) * (32 * QB) + (lane & 31);
  Nn1State st;
#pragma unroll
  for (int b = 0; b < QB; b++) {
    const int j = jbase + 32 * b;
    const int jc = j < k.n_q ? j : k.n_q - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) st.bq[b][ks] = *(const v4i *)(qdesc + (size_t)jc * 128 + ks * 32 + g * 16);
    st.M1[b] = 0u; st.M2[b] = 0u; st.M3[b] = 0u;
  }
  // the query operands are complete before the tile loop starts (otherwise the compiler's waits for them sit in front of the
  // loop's MFMAs, where in every later iteration they would wait for the prefetch instead)
#pragma unroll
  for (int b = 0; b < QB; b++)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) asm volatile("" : "+v"(st.bq[b][ks]));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 4; ks++) st.aoff[ks] = (lane & 31) * 128 + (((2 * ks + g) ^ (((lane & 31) >> 1) & 7)) << 4);
  st.soff = 16 * g;
#pragma unroll
  for (int ks = 0; ks < 4; ks++) st.a[ks] = *(const v4i *)(s_ring + st.aoff[ks]);
  st.seed = acc_seed((const int *)(s_ring + 4096 + st.soff), 0, 0);
  v16i accA[QB], accB[QB];            // even tiles go into accA, odd ones into accB
  auto idc = [](int i) { return (unsigned)(NN1_MAX_TPS - 1 - i) << 1; };
#define NN1_STEP(EPI, RING, GUARD, W, R, I) nn1_step<EPI, RING, GUARD>(st, W, R, s_ring, tdesc, tc2n, t0 + (I), (I), n, w, lane, idc((I) - 1))
  if (n > 0) {
    NN1_STEP(false, 0, true, accA, accB, 0);
    int i = 1;
    if (i < n) { NN1_STEP(true, 1, true, accB, accA, i); i++; }
    if (i < n) { NN1_STEP(true, 2, true, accA, accB, i); i++; }
    if (i < n) { NN1_STEP(true, 3, true, accB, accA, i); i++; }
    // four tiles per iteration: image numbers and accumulator sets are compile-time constants, nothing is conditional
    for (; i + 3 + NN1_PF < n; i += 4) {
      NN1_STEP(true, 0, false, accA, accB, i);
      NN1_STEP(true, 1, false, accB, accA, i + 1);
      NN1_STEP(true, 2, false, accA, accB, i + 2);
      NN1_STEP(true, 3, false, accB, accA, i + 3);
    }
    for (; i < n; i += 4) {
      NN1_STEP(true, 0, true, accA, accB, i);
      if (i + 1 < n) NN1_STEP(true, 1, true, accB, accA, i + 1);
      if (i + 2 < n) NN1_STEP(true, 2, true, accA, accB, i + 2);
      if (i + 3 < n) NN1_STEP(true, 3, true, accB, accA, i + 3);
    }
    // the last tile's epilogue
    const v16i (&accL)[QB] = (n & 1) ? accA : accB;
#pragma unroll
    for (int b = 0; b < QB; b++) top3_insert(st.M1[b], st.M2[b], st.M3[b], ((unsigned)acc_max16(accL[b]) << 9) | idc(n - 1));
  }
#undef NN1_STEP
  const size_t n_qpad = (size_t)qblocks * (32 * NN1_WAVES) * QB;
#pragma unroll
  for (int b = 0; b < QB; b++) {
    const int j = jbase + 32 * b;
    // the two half-waves hold the same 32 queries (the two halves of every tile): the lower one merges both triples
    unsigned m1 = st.M1[b] | (unsigned)(1 - g), m2 = st.M2[b] | (unsigned)(1 - g), m3 = st.M3[b] | (unsigned)(1 - g);
    const unsigned o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32), o3 = __shfl_xor(m3, 32);
    top3_insert(m1, m2, m3, o1); top3_insert(m1, m2, m3, o2); top3_insert(m1, m2, m3, o3);
    if (g == 0 && j < k.n_q) best3[(size_t)sp * n_qpad + j] = make_uint4(m1, m2, m3, 0u);
  }
}

// Exact finish of pass 1: one half-wave (32 lanes) per query.  grid = ceil(n_q/8), block 256.
//   1. A = accumulator level of the second largest key over the (M1, M2) of all train splits (at least 1: rows past the end of the
//      list have the level 0);
//   2. candidates = every kept half tile at or above A; unsafe when some split's M3 is at or above A too (see above), or when the
//      candidates do not fit the list;
//   3. lane = one train row of a candidate (two candidates per round), exact d = cq + ct - 2*dot by v_dot4 over the packed
//      descriptors, the two smallest keys (d << 32 | t) by a half-wave reduction; an unsafe query walks all trains instead.
// best2: [n_q][2], the input of match_mid_kernel (one "split").
constexpr int FIX_MAXC = 32;
__device__ __forceinline__ int desc_dot(const int8_t *__restrict__ qrow, const int8_t *__restrict__ trow) {
  int dot = 0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int4 a = ((const int4 *)qrow)[e], b = ((const int4 *)trow)[e];
    dot = __builtin_amdgcn_sdot4(a.x, b.x, dot, false); dot = __builtin_amdgcn_sdot4(a.y, b.y, dot, false);
    dot = __builtin_amdgcn_sdot4(a.z, b.z, dot, false); dot = __builtin_amdgcn_sdot4(a.w, b.w, dot, false);
  }
  return dot;
}
__device__ __forceinline__ void top2_min_insert(unsigned long long &k1, unsigned long long &k2, unsigned long long key) {
  const bool first = key < k1;
  k2 = first ? k1 : (key < k2 ? key : k2);
  k1 = first ? key : k1;
}
__global__ __launch_bounds__(256) void match_fix_kernel(MatchJobs J, MatchConst k, const uint4 *__restrict__ best3,
                                                        const int8_t *__restrict__ qdesc, const int *__restrict__ qc,
                                                        const int8_t *__restrict__ tdesc, const int *__restrict__ tc,
                                                        unsigned long long *__restrict__ best2) {
  __shared__ int s_cand[8][FIX_MAXC];
  const int job = blockIdx.y;
  k.n_q = J.n_q[job]; k.n_t = J.n_t[job]; k.tiles_per_split = J.tps[job];
  if ((int)blockIdx.x * 8 >= k.n_q) return;
  const int splits = J.splits[job];
  const size_t n_qpad = (size_t)J.qblocks[job] * (32 * NN1_WAVES) * MATCH_QB1;
  best3 = set_by(best3, job, J.s_p2); best2 = set_by(best2, job, J.s_p2);
  qdesc = set_by(qdesc, job, J.s_desc); tdesc = set_by(tdesc, job, J.s_desc); qc = set_el(qc, job, J.s_c); tc = set_el(tc, job, J.s_c);
  const int lane = threadIdx.x & 63, hl = lane & 31, hw = threadIdx.x >> 5;
  const int hshift = lane & 32;             // this half-wave's bits of a ballot
  const int j = blockIdx.x * 8 + hw;
  const bool live = j < k.n_q;
  const int jc = live ? j : k.n_q - 1;
  // 1. the two largest (M1, M2) keys over the splits
  unsigned K1 = 0u, K2 = 0u;
  for (int s = hl; s < splits; s += 32) {
    const uint4 v = best3[(size_t)s * n_qpad + jc];
    K2 = max(min(K1, v.x), max(K2, v.y));      // v.x >= v.y
    K1 = max(K1, v.x);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const unsigned o1 = __shfl_xor(K1, off), o2 = __shfl_xor(K2, off);
    K2 = max(min(K1, o1), max(K2, o2));
    K1 = max(K1, o1);
  }
  const unsigned A = max(K2 >> 9, 1u);
  // 2. candidate half tiles (tile << 1 | half) of this query
  int nc = 0;
  bool unsafe = false;
  for (int s0 = 0; s0 < splits; s0 += 32) {
    const int s = s0 + hl;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (s < splits) v = best3[(size_t)s * n_qpad + jc];
    unsafe |= (v.z >> 9) >= A;
    const unsigned key[2] = {v.x, v.y};
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const bool c = (key[e] >> 9) >= A;
      const unsigned m = (unsigned)(__ballot(c) >> hshift);
      const int slot = nc + __popc(m & ((1u << hl) - 1u));
      if (c && slot < FIX_MAXC)
        s_cand[hw][slot] = ((s * k.tiles_per_split + (NN1_MAX_TPS - 1 - (int)((key[e] >> 1) & 255u))) << 1) | (int)(1u - (key[e] & 1u));
      nc += __popc(m);
    }
  }
  unsafe = ((unsigned)(__ballot(unsafe) >> hshift) != 0u) || nc > FIX_MAXC;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 3. exact distances
  const int8_t *qrow = qdesc + (size_t)jc * 128;
  const int cq = qc[jc] - 4194304;
  unsigned long long k1 = ~0ull, k2 = ~0ull;
  if (!unsafe) {
    const int r = hl & 15;
    for (int c0 = 0; c0 < nc; c0 += 2) {
      const int ci = c0 + (hl >> 4);
      if (ci < nc) {
        const int cd = s_cand[hw][ci];
        const int t = (cd >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (cd & 1);
        if (t < k.n_t) {
          const int d = cq + tc[t] - 2 * desc_dot(qrow, tdesc + (size_t)t * 128);
          top2_min_insert(k1, k2, ((unsigned long long)(unsigned int)d << 32) | (unsigned int)t);
        }
      }
    }
  } else {
    for (int t = hl; t < k.n_t; t += 32) {
      const int d = cq + tc[t] - 2 * desc_dot(qrow, tdesc + (size_t)t * 128);
      top2_min_insert(k1, k2, ((unsigned long long)(unsigned int)d << 32) | (unsigned int)t);
    }
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const unsigned long long o1 = shfl_xor_u64(k1, off), o2 = shfl_xor_u64(k2, off);
    const unsigned long long hi = k1 < o1 ? o1 : k1, lo2 = k2 < o2 ? k2 : o2;
    k1 = k1 < o1 ? k1 : o1;
    k2 = hi < lo2 ? hi : lo2;
  }
  if (hl == 0 && live) { best2[(size_t)j * 2] = k1; best2[(size_t)j * 2 + 1] = k2; }
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
//                                  compact arrays)
__global__ __launch_bounds__(256) void match_mid_kernel(MatchJobs J, MatchConst k, const unsigned long long *__restrict__ best2,
                                                        const double2 *__restrict__ txy, QueryMid *__restrict__ mid,
                                                        unsigned long long *__restrict__ key_ge, unsigned long long *__restrict__ key_lt,
                                                        int *__restrict__ n_lt, int *__restrict__ bad, int *__restrict__ list2,
                                                        int *__restrict__ count2, const int8_t *__restrict__ qdesc,
                                                        const int *__restrict__ qc, int8_t *__restrict__ qdesc2, int *__restrict__ qc2,
                                                        QueryMid *__restrict__ mid2) {
  const int job = blockIdx.y;
  k.n_q = J.n_q[job]; k.n_t = J.n_t[job];
  const int splits = 1; const size_t n_qpad = (size_t)k.n_q;     // (the finish kernel leaves one exact pair of keys per query)
  best2 = set_by(best2, job, J.s_p2); list2 = set_by(list2, job, J.s_p2); count2 = set_by(count2, job, J.s_p2);
  qdesc2 = set_by(qdesc2, job, J.s_p2); qc2 = set_by(qc2, job, J.s_p2); mid2 = set_by(mid2, job, J.s_p2);
  qdesc = set_by(qdesc, job, J.s_desc); qc = set_el(qc, job, J.s_c);
  txy = set_el(txy, job, J.s_xy); mid = set_el(mid, job, J.s_mid);
  key_ge = set_el(key_ge, job, J.s_u64); key_lt = set_el(key_lt, job, J.s_u64); n_lt = set_el(n_lt, job, J.s_int); bad = set_el(bad, job, J.s_int);
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
      else {
        // undecided: the query goes on the pass-2 list, and this thread leaves its descriptor row, norm and state in the compact
        // arrays (a few % of the queries: eight 16-byte copies each; a launch of its own for this was 5 us + a boundary, round 6)
        const int i = atomicAdd(count2, 1);
        list2[i] = j;
        const uint4 *src = (const uint4 *)(qdesc + (size_t)j * 128);
        uint4 *dst = (uint4 *)(qdesc2 + (size_t)i * 128);
        uint4 row[8];
#pragma unroll
        for (int q = 0; q < 8; q++) row[q] = src[q];
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = row[q];
        qc2[i] = qc[j]; mid2[i] = m;
      }
    }
  }
  key_ge[j] = kg; key_lt[j] = ~0ull; n_lt[j] = 0; bad[j] = isbad;
}

// Pass 2: FGINN reductions.  Same tiling and the same two-speed epilogue as pass 1: a tile is examined exactly
// only when one of its partial distances lies below max(D*, smallest distance >= D* found so far).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void match_fginn_kernel(MatchJobs J, MatchConst k, const int8_t *__restrict__ qdesc, const int *__restrict__ qc,
                                                          const int8_t *__restrict__ tdesc, const int *__restrict__ tc,
                                                          const int *__restrict__ tc2n, const unsigned int *__restrict__ tpar,
                                                          const double2 *__restrict__ txy, const QueryMid *__restrict__ mid,
                                                          unsigned long long *__restrict__ key_ge, unsigned long long *__restrict__ key_lt,
                                                          int *__restrict__ n_lt, int *__restrict__ bad,
                                                          const int *__restrict__ n_q_dev, const int *__restrict__ out_index) {
  constexpr int QB = MATCH_QB;
  own_simd_512();
  {
    const int job = blockIdx.y;
    k.n_t = J.n_t[job];
    qdesc = set_by(qdesc, job, J.s_p2); qc = set_by(qc, job, J.s_p2); mid = set_by(mid, job, J.s_p2);
    n_q_dev = set_by(n_q_dev, job, J.s_p2); out_index = set_by(out_index, job, J.s_p2);
    tdesc = set_by(tdesc, job, J.s_desc); tc = set_el(tc, job, J.s_c); tc2n = set_el(tc2n, job, J.s_c); tpar = set_el(tpar, job, J.s_c);
    txy = set_el(txy, job, J.s_xy);
    key_ge = set_el(key_ge, job, J.s_u64); key_lt = set_el(key_lt, job, J.s_u64); n_lt = set_el(n_lt, job, J.s_int); bad = set_el(bad, job, J.s_int);
  }
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
    cq[b] = qc[jc] - 4194304 + 2 * MATCH_BIAS;   // d = cq + (ct & 1) - 2 * (seeded accumulator)
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
  // round 6: the operands and seeds of tile tt + 1 are requested before tile tt is multiplied and examined (a wave fetched them at
  // the top of every step and waited a full round trip per tile: ~1 us x 29 tiles per workgroup at 60 156 x 47 177); the rows of
  // the step behind the split's last tile are read clamped and not used
  v4i an[4];
  v16i seedn;
  {
    const int8_t *arow = tdesc + (size_t)(t0 * 32 + (lane & 31)) * 128 + g * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) an[ks] = *(const v4i *)(arow + ks * 32);
    seedn = acc_seed(tc2n, t0 * 32, g);
  }
  for (int tt = t0; tt < t1; tt++) {
    const int tbase = tt * 32;
    v4i a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) a[ks] = an[ks];
    const unsigned int par = tpar[tt];   // wave-uniform: ct & 1 of the 32 train rows
    v16i acc[QB];
    acc[0] = seedn;
#pragma unroll
    for (int b = 1; b < QB; b++) acc[b] = acc[0];
    {
      const int tn = tt + 1 < t1 ? tt + 1 : tt;
      const int8_t *arow = tdesc + (size_t)(tn * 32 + (lane & 31)) * 128 + g * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) an[ks] = *(const v4i *)(arow + ks * 32);
      seedn = acc_seed(tc2n, tn * 32, g);
    }
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

// Decision + order-preserving compaction into the tentative list, in two launches over ceil(n_q / EMIT_T) blocks (EMIT_T = 256
// since round 6: at 1024 queries per block a 60 000-query search kept 59 CUs busy with chains of dependent loads - 24 + 5 us):
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

constexpr int EMIT_T = 256;        // queries (= threads) per block of the emit stage
__global__ __launch_bounds__(EMIT_T) void match_emit_count_kernel(MatchJobs J, MatchConst k, const QueryMid *__restrict__ mid,
                                                                const unsigned long long *__restrict__ key_ge,
                                                                const unsigned long long *__restrict__ key_lt, const int *__restrict__ n_lt,
                                                                const int *__restrict__ bad, int *__restrict__ block_counts) {
  const int job = blockIdx.y;
  if ((int)blockIdx.x >= J.eblocks[job]) return;
  k.n_q = J.n_q[job]; k.n_t = J.n_t[job];
  mid = set_el(mid, job, J.s_mid); key_ge = set_el(key_ge, job, J.s_u64); key_lt = set_el(key_lt, job, J.s_u64);
  n_lt = set_el(n_lt, job, J.s_int); bad = set_el(bad, job, J.s_int); block_counts = set_el(block_counts, job, J.s_int);
  mods_tentative tc;
  const bool emit = fginn_accept(k, blockIdx.x * EMIT_T + threadIdx.x, mid, key_ge, key_lt, n_lt, bad, &tc);
  const int c = __syncthreads_count(emit ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}

__global__ __launch_bounds__(EMIT_T) void match_emit_kernel(MatchJobs J, MatchConst k, const QueryMid *__restrict__ mid,
                                                          const unsigned long long *__restrict__ key_ge,
                                                          const unsigned long long *__restrict__ key_lt, const int *__restrict__ n_lt,
                                                          const int *__restrict__ bad, const double2 *__restrict__ qxy,
                                                          const double2 *__restrict__ txy, const int *__restrict__ block_counts, int max_out) {
  __shared__ int s_wave[16], s_wtot[16];
  __shared__ int s_base, s_total;
  const int job = blockIdx.y;
  const int n_blocks = J.eblocks[job];
  if ((int)blockIdx.x >= n_blocks) return;
  k.n_q = J.n_q[job]; k.n_t = J.n_t[job];
  mid = set_el(mid, job, J.s_mid); key_ge = set_el(key_ge, job, J.s_u64); key_lt = set_el(key_lt, job, J.s_u64);
  n_lt = set_el(n_lt, job, J.s_int); bad = set_el(bad, job, J.s_int); block_counts = set_el(block_counts, job, J.s_int);
  qxy = set_el(qxy, job, J.s_xy); txy = set_el(txy, job, J.s_xy);
  const mods_region *__restrict__ qreg = J.q_reg[job], *__restrict__ treg = J.t_reg[job];
  mods_tentative *__restrict__ out = J.tent_out[job];
  int *__restrict__ out_count = J.count_out[job];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // offset of this block = accepted queries of all earlier blocks; the total fixes the packed layout of the output
  // (tentatives | correspondences | frames, see common.hpp)
  int part = 0, tot = 0;
  for (int q = tid; q < n_blocks; q += EMIT_T) { const int c = block_counts[q]; tot += c; if (q < (int)blockIdx.x) part += c; }
  for (int off = 32; off > 0; off >>= 1) { part += __shfl_xor(part, off); tot += __shfl_xor(tot, off); }
  if (lane == 0) { s_wave[wv] = part; s_wtot[wv] = tot; }
  __syncthreads();
  if (tid == 0) { int t = 0, u = 0; for (int q = 0; q < EMIT_T / 64; q++) { t += s_wave[q]; u += s_wtot[q]; } s_base = t; s_total = u; }
  __syncthreads();
  const int base = s_base;
  const size_t n_out = (size_t)min(s_total, max_out);
  double *u6 = (double *)((char *)out + tent_u6_off(n_out)), *laf = (double *)((char *)out + tent_laf_off(n_out));
  __syncthreads();
  mods_tentative tc;
  const bool emit = fginn_accept(k, blockIdx.x * EMIT_T + tid, mid, key_ge, key_lt, n_lt, bad, &tc);
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
  if ((int)blockIdx.x == n_blocks - 1 && tid == 0) {   // total = offset of the last block + its own count
    int t = base;
    for (int q = 0; q < EMIT_T / 64; q++) t += s_wave[q];
    *out_count = t;
  }
}

// ---------------------------------------------------------------------------------------
#ifndef FGINN_BLOCKS_N
#define FGINN_BLOCKS_N 256    // measured (round 4): 1024 / 512 / 256 workgroups: 78 / 60 / 47 us at 60 156 x 47 177, 19 / 19 / 15 us on a 1080p pair
#endif
constexpr int FGINN_BLOCKS = FGINN_BLOCKS_N;     // pass 2: one workgroup per CU (it owns the register file), one round
static size_t match_pad(const mods_ctx *ctx) { return ((size_t)ctx->max_cand + 127) & ~(size_t)63; }   // list stride, tile tail included

// Train splits of pass 1.  A workgroup owns its compute unit (register allocation), so the launch runs in rounds of one workgroup
// per CU and its time is rounds x (tiles per split + the workgroup's fixed cost: 128 KB of query operands, the first tiles'
// latency, the final key merge - about six tiles' worth): the split count minimises that product, never more than NN1_MAX_TPS
// tiles each.
struct Nn1Grid { int qblocks, splits, tiles_per_split; size_t n_qpad; };
static Nn1Grid nn1_grid(int n_q, int n_t) {
  constexpr int QPB = 32 * NN1_WAVES * MATCH_QB1;      // queries per pass-1 workgroup
  constexpr int CUS = 256, FIXED = 6;
  Nn1Grid gr;
  const int n_tiles = (n_t + 31) / 32;
  gr.qblocks = (n_q + QPB - 1) / QPB;
  const int smin = std::max(1, (n_tiles + NN1_MAX_TPS - 1) / NN1_MAX_TPS);
  int splits = smin;
  {
    long best = -1;
    const int smax = std::max(smin, std::min(n_tiles, 4 * CUS / std::max(1, gr.qblocks) + 1));
    for (int sc = smin; sc <= smax; sc++) {
      const long rounds = ((long)gr.qblocks * sc + CUS - 1) / CUS, tps = (n_tiles + sc - 1) / sc;
      const long cost = rounds * (tps + FIXED);
      if (best < 0 || cost < best) { best = cost; splits = sc; }
    }
  }
  gr.tiles_per_split = (n_tiles + splits - 1) / splits;
  gr.splits = (n_tiles + gr.tiles_per_split - 1) / gr.tiles_per_split;
  gr.n_qpad = (size_t)gr.qblocks * QPB;
  return gr;
}

// set strides of the matcher's scratch buffers (MatchJobs)
static void match_strides(const mods_ctx *ctx, MatchJobs *J) {
  const size_t n = match_pad(ctx);
  J->s_desc = 2 * n * 128;
  J->s_c = 4 * n + 2 * (n / 32 + 2);
  J->s_xy = 2 * n;
  J->s_u64 = 3 * n;
  J->s_int = 2 * n + n / 256 + 2;         // n_lt | bad | the emit stage's block counts (EMIT_T = 256 queries per block)
  J->s_mid = n;
  J->s_p2 = (ctx->m_best2_cap * 16 + n * 16 + n * (3 * sizeof(int) + 128 + sizeof(QueryMid)) + 128 + 255) & ~(size_t)255;
}

// n_sets: searches a grouped launch may hold (the pairs of a pipeline batch); growing the sets reallocates, so the pipeline's warm-up
// asks for what its batches need
int match_ensure_buffers(mods_ctx *ctx, int n_sets) {
  if (n_sets < 1) n_sets = 1;
  if (n_sets > MATCH_MAX_JOBS) n_sets = MATCH_MAX_JOBS;
  if (ctx->m_desc && ctx->m_sets >= n_sets) return MODS_OK;
  const size_t n = match_pad(ctx);
  if (ctx->m_desc) {        // more sets than before: the per-search scratch is reallocated (nothing in it outlives a search)
    MODS_HIP_CHECK(mods::stream_wait(ctx->stream));
    MODS_HIP_CHECK(hipFree(ctx->m_desc)); MODS_HIP_CHECK(hipFree(ctx->m_c)); MODS_HIP_CHECK(hipFree(ctx->m_xy)); MODS_HIP_CHECK(hipFree(ctx->m_u64));
    MODS_HIP_CHECK(hipFree(ctx->m_int)); MODS_HIP_CHECK(hipFree(ctx->m_mid)); MODS_HIP_CHECK(hipFree(ctx->m_p2));
    ctx->m_desc = nullptr; ctx->m_c = nullptr; ctx->m_xy = nullptr; ctx->m_u64 = nullptr; ctx->m_int = nullptr; ctx->m_mid = nullptr; ctx->m_p2 = nullptr;
  }
  // pass-1 key table (one 16-byte triple per query and train split): the largest product splits * padded queries over the list
  // sizes this context admits (the split count grows with the train list, so only the query count is scanned)
  {
    size_t cap = 0;
    for (int nq = 1; nq < ctx->max_cand + 32 * NN1_WAVES * MATCH_QB1; nq += 32 * NN1_WAVES * MATCH_QB1) {
      const Nn1Grid gr = nn1_grid(std::min(nq, ctx->max_cand), ctx->max_cand);
      cap = std::max(cap, (size_t)gr.splits * gr.n_qpad);
    }
    ctx->m_best2_cap = cap;
  }
  MatchJobs st;
  match_strides(ctx, &st);
  const size_t S = (size_t)n_sets;
  MODS_HIP_CHECK(hipMalloc(&ctx->m_desc, S * st.s_desc));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_c, S * st.s_c * sizeof(int)));   // c of queries, trains; seeds of both; parity words of both
  MODS_HIP_CHECK(hipMalloc(&ctx->m_xy, S * st.s_xy * sizeof(double2)));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_u64, S * st.s_u64 * sizeof(unsigned long long)));
  MODS_HIP_CHECK(hipMalloc(&ctx->m_int, S * st.s_int * sizeof(int)));   // n_lt, bad, per-block counts of the compaction
  MODS_HIP_CHECK(hipMalloc(&ctx->m_mid, S * st.s_mid * sizeof(QueryMid)));
  // m_p2: key table | exact top-2 per query | pass-2 subset: state, descriptors, list, norms, count
  MODS_HIP_CHECK(hipMalloc(&ctx->m_p2, S * st.s_p2));
  if (!ctx->m_regs) MODS_HIP_CHECK(hipMalloc(&ctx->m_regs, 2 * (size_t)ctx->max_cand * sizeof(mods_region)));
  if (!ctx->m_tent) MODS_HIP_CHECK(hipMalloc(&ctx->m_tent, tent_bytes(n) + 64));
  // the tentative count lives in pinned host memory: the emit kernel's single store lands there, the host reads it after a
  // stream synchronisation - no 4-byte copy launch per search
  if (!ctx->m_count) MODS_HIP_CHECK(hipHostMalloc(&ctx->m_count, 192 * sizeof(int)));   // [0]: the last search; [i]: pair i of a batch (m_count_out)
  MODS_HIP_CHECK(hipMemsetAsync(ctx->m_desc, 0, S * st.s_desc, ctx->stream));
  MODS_HIP_CHECK(hipMemsetAsync(ctx->m_c, 0, S * st.s_c * sizeof(int), ctx->stream));
  ctx->m_sets = n_sets;
  return MODS_OK;
}

// One launch instead of three memsets per search: tentative counter, pass-2 list counter and the parity words the pack kernels
// OR into.  grid = ceil(max(n_q, n_t) / 32 / 256) + 1, block 256
__global__ __launch_bounds__(256) void match_init_kernel(int n_q, int n_t, int *__restrict__ m_count, int *__restrict__ count2,
                                                         unsigned int *__restrict__ qpar, unsigned int *__restrict__ tpar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { *m_count = 0; *count2 = 0; }
  if (i <= n_q / 32) qpar[i] = 0u;
  if (i <= n_t / 32) tpar[i] = 0u;
}

// n_jobs searches (queries / trains: device region lists with host-known sizes) in one set of launches; search j leaves its packed
// list at tent_out[j] and its length at *count_out[j] (pinned host memory).
int match_run_group(mods_ctx *ctx, int n_jobs, const mods_region *const *q_dev, const int *n_q, const mods_region *const *t_dev, const int *n_t,
                    mods_tentative *const *tent_out, int *const *count_out, double ratio, double contradDist, int nn) {
  if (n_jobs < 1 || n_jobs > MATCH_MAX_JOBS) { set_error("match: %d searches in a group", n_jobs); return MODS_E_ARG; }
  int rc = match_ensure_buffers(ctx, n_jobs);
  if (rc) return rc;
  MatchConst k;
  k.n_q = 0; k.n_t = 0; k.nn = nn;
  k.max_distance = -1;
  k.tiles_per_split = 0;
  k.sqminratio = ratio * ratio;
  k.contr_sq = contradDist * contradDist;
  if (!(k.sqminratio < 1.0)) { set_error("FGINN ratio >= 1 (all-neighbours mode) is not supported"); return MODS_E_ARG; }
  MatchJobs J;
  memset(&J, 0, sizeof(J));
  match_strides(ctx, &J);
  J.n_jobs = n_jobs;
  int max_q = 0, max_t = 0, max_nn1 = 0, max_eb = 0;
  for (int j = 0; j < n_jobs; j++) {
    if (n_q[j] > ctx->max_cand || n_t[j] > ctx->max_cand) { set_error("match: list larger than the context capacity"); return MODS_E_CAPACITY; }
    const bool empty = n_q[j] <= 0 || n_t[j] <= 0;      // an empty search: every kernel passes it by, the pack kernel leaves its count at 0
    J.n_q[j] = empty ? 0 : n_q[j]; J.n_t[j] = empty ? 0 : n_t[j];
    J.q_reg[j] = q_dev[j]; J.t_reg[j] = t_dev[j];
    J.tent_out[j] = tent_out[j]; J.count_out[j] = count_out[j];
    if (empty) continue;
    // pass 1: the three largest half-tile maxima per query and train split, then the exact two nearest trains per query
    const Nn1Grid gr = nn1_grid(n_q[j], n_t[j]);
    if ((size_t)gr.splits * gr.n_qpad > ctx->m_best2_cap) { set_error("match: pass-1 key table too small"); return MODS_E_CAPACITY; }
    J.tps[j] = gr.tiles_per_split; J.qblocks[j] = gr.qblocks; J.splits[j] = gr.splits;
    J.eblocks[j] = (n_q[j] + EMIT_T - 1) / EMIT_T;
    max_q = std::max(max_q, n_q[j]); max_t = std::max(max_t, n_t[j]);
    max_nn1 = std::max(max_nn1, gr.qblocks * gr.splits); max_eb = std::max(max_eb, J.eblocks[j]);
  }
  const unsigned G = (unsigned)n_jobs;
  const size_t n = match_pad(ctx);
  int8_t *qd = ctx->m_desc, *td = ctx->m_desc + n * 128;
  int *qc = ctx->m_c, *tc = ctx->m_c + n, *qc2 = ctx->m_c + 2 * n, *tc2 = ctx->m_c + 3 * n;
  unsigned int *qpar = (unsigned int *)(ctx->m_c + 4 * n), *tpar = qpar + n / 32 + 2;
  double2 *qxy = (double2 *)ctx->m_xy, *txy = (double2 *)ctx->m_xy + n;
  unsigned long long *key_ge = ctx->m_u64 + n, *key_lt = ctx->m_u64 + 2 * n;
  int *n_lt = ctx->m_int, *bad = ctx->m_int + n;
  StageScope ts(ctx, MODS_STAGE_MATCH);
  // carve of m_p2 (every part 16-byte aligned)
  uint4 *best3 = (uint4 *)ctx->m_p2;
  unsigned long long *best2 = (unsigned long long *)(best3 + ctx->m_best2_cap);
  QueryMid *mid2 = (QueryMid *)(best2 + 2 * n);
  int8_t *qd2 = (int8_t *)(mid2 + n);
  int *list2 = (int *)(qd2 + n * 128), *qcs = list2 + n, *count2 = qcs + n;
  {
    const PackList lq = {nullptr, 0, qd, qc, qc2, qpar, qxy}, lt = {nullptr, 0, td, tc, tc2, tpar, txy};
    hipLaunchKernelGGL(match_pack2_kernel, dim3((std::max(std::max(max_q, max_t), 1) + 127) / 128, 2, G), dim3(256), 0, ctx->stream, J, lq, lt, ctx->max_cand, count2);
  }
  if (max_q > 0) {
    {
      StageScope ts1(ctx, MODS_STAGE_MATCH_NN1);
      hipLaunchKernelGGL(match_nn1_kernel, dim3(max_nn1, G), dim3(NN1_THREADS), 0, ctx->stream, J, k, qd, td, tc2, best3);
    }
    hipLaunchKernelGGL(match_fix_kernel, dim3((max_q + 7) / 8, G), dim3(256), 0, ctx->stream, J, k, (const uint4 *)best3, qd, qc, td, tc, best2);
    hipLaunchKernelGGL(match_mid_kernel, dim3((max_q + 255) / 256, G), dim3(256), 0, ctx->stream, J, k, (const unsigned long long *)best2, txy,
                       (QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, list2, count2, qd, qc, qd2, qcs, mid2);
    // pass 2 on the undecided queries only (their number stays on the device: the grid covers the worst case, idle blocks exit)
    {
      const int qblocks = (max_q + 128 * MATCH_QB - 1) / (128 * MATCH_QB);
      // (a group shares the chip: a single search gets the whole row of workgroups, the searches of a group their share of it)
      const int row = std::max(std::max(FGINN_BLOCKS / n_jobs, 64), qblocks);
      hipLaunchKernelGGL(match_fginn_kernel, dim3(row, G), dim3(256), 0, ctx->stream, J, k, qd2, qcs, td, tc, tc2, tpar, txy,
                         (const QueryMid *)mid2, key_ge, key_lt, n_lt, bad, count2, list2);
    }
    int *block_counts = (int *)(ctx->m_int + 2 * n);
    hipLaunchKernelGGL(match_emit_count_kernel, dim3(max_eb, G), dim3(EMIT_T), 0, ctx->stream, J, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, block_counts);
    hipLaunchKernelGGL(match_emit_kernel, dim3(max_eb, G), dim3(EMIT_T), 0, ctx->stream, J, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, qxy, txy,
                       block_counts, ctx->max_cand);
  }
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

// queries / trains: device region lists with host-known sizes n_q, n_t.
int match_run(mods_ctx *ctx, const mods_region *q_dev, int n_q, const mods_region *t_dev, int n_t, double ratio,
              double contradDist, int nn) {
  int rc = match_ensure_buffers(ctx);
  if (rc) return rc;
  // where the packed list and its length go: the context's own buffer / counter, or what a batch of pairs set (capi.hip: match_pairs)
  mods_tentative *tent_out = ctx->m_tent_out ? ctx->m_tent_out : ctx->m_tent;
  int *count_out = ctx->m_count_out ? ctx->m_count_out : ctx->m_count;
  return match_run_group(ctx, 1, &q_dev, &n_q, &t_dev, &n_t, &tent_out, &count_out, ratio, contradDist, nn);
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
  const int eblocks = (n_q + EMIT_T - 1) / EMIT_T;
  int *block_counts = (int *)(ctx->m_int + 2 * n);
  MatchJobs J;
  memset(&J, 0, sizeof(J));
  match_strides(ctx, &J);
  J.n_jobs = 1; J.n_q[0] = n_q; J.n_t[0] = n_t; J.eblocks[0] = eblocks;
  J.q_reg[0] = q_dev; J.t_reg[0] = t_dev; J.tent_out[0] = ctx->m_tent; J.count_out[0] = ctx->m_count;
  hipLaunchKernelGGL(match_emit_count_kernel, dim3(eblocks), dim3(EMIT_T), 0, ctx->stream, J, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, block_counts);
  hipLaunchKernelGGL(match_emit_kernel, dim3(eblocks), dim3(EMIT_T), 0, ctx->stream, J, k, (const QueryMid *)ctx->m_mid, key_ge, key_lt, n_lt, bad, qxy, txy, block_counts, ctx->max_cand);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods

