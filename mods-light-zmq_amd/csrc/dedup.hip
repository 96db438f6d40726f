// DuplicateFiltering on the device (matching/matching.cpp:2615-2679 as mods_duplicate_filter, csrc/capi.hip, restates it on the
// host): the tentatives are put in the order of `whichCorrespondenceRemains` (a stable sort by FGINN ratio / distance / region
// scale; "random" keeps the list order), and a correspondence is dropped when an EARLIER KEPT one lies within `duplicateDist` of
// it in both images.
//
// The rule is sequential only through short chains, so it is restated as four launches over the packed lists the matcher leaves
// in HBM (mods_tentative[n] | u6[n][6] | laf[n][14], common.hpp) - the lists of ALL pairs of a pipeline batch in the same
// launches (blockIdx.z = pair):
//   dup_rank_kernel    rank = #{(key, index) pairs below mine}: block (i tile, j tile) counts a 256 x 256 block of the comparison
//                      matrix out of LDS and adds its part to rank[i] (keys are doubles compared as doubles, ties by list
//                      index: std::sort over (key, index) pairs on the host)
//   dup_scatter_kernel (x1 y1 x2 y2) and the list index to their sorted places
//   dup_near_kernel    for every sorted position p the earlier positions within the distance in both images, by brute force:
//                      block (p tile, earlier tile) tests 256 x 256 pairs (the same two fp64 tests in the same order as the
//                      host's hash walk; which earlier one is met first is irrelevant, only whether a kept one exists).  A
//                      1080p pair's 6 000 tentatives are 19 M tests, a 4096 x 4096 pair's 24 000 are 290 M
//   dup_resolve_kernel one workgroup per pair: kept[p] = no kept position in near(p), iterated to its fixed point (position p
//                      depends on positions below p only, so every round settles at least the lowest open one; chains are a
//                      few links long), then an order-preserving compaction into the packed output list
// A position with more than DUP_K near predecessors, or a list longer than the resolve kernel's LDS holds, raises `status`: the
// caller then filters that list on the host.  Integer / comparison work only on the keys; every distance is the host's fp64
// expression (no contraction), so the kept set is identical.
#include "common.hpp"
#include <algorithm>

namespace mods {

constexpr int DUP_K = 12;              // near predecessors kept per position
constexpr int DUP_MAX_N = 150 * 1024;  // one state byte per position in the LDS of the resolve kernel

struct DupBatch {
  int n_jobs, max_n, mode;
  double r_sq;
  char *scratch; size_t stride;        // per job: xy4[max_n] | order[max_n] | near[max_n * DUP_K]
  int *counters;                       // per job: rank[max_n] | near_cnt[max_n] - the only arrays that must start at zero, contiguous
  DupJob job[DUP_MAX_JOBS];            // over the jobs so that ONE fill of 8 B per position clears them
};
__device__ __forceinline__ double4 *dj_xy4(const DupBatch &b, int j) { return (double4 *)(b.scratch + b.stride * j); }
__device__ __forceinline__ int *dj_rank(const DupBatch &b, int j) { return b.counters + (size_t)2 * b.max_n * j; }
__device__ __forceinline__ int *dj_cnt(const DupBatch &b, int j) { return dj_rank(b, j) + b.max_n; }
__device__ __forceinline__ int *dj_order(const DupBatch &b, int j) { return (int *)(dj_xy4(b, j) + b.max_n); }
__device__ __forceinline__ int *dj_near(const DupBatch &b, int j) { return dj_order(b, j) + b.max_n; }

__device__ __forceinline__ double dup_key(const char *src, int n, int i, int mode) {
  const mods_tentative *t = (const mods_tentative *)src;
  if (mode == 1) return fabs(t[i].ratio);
  if (mode == 2) return fabs((double)t[i].d1);
  const double *laf = (const double *)(src + tent_laf_off((size_t)n));
  return fabs(laf[(size_t)i * 14 + 6]);
}

// grid = (tiles, tiles, jobs), block 256: block (x, y) = the entries x * 256 .. against the keys y * 256 ..
__global__ __launch_bounds__(256) void dup_rank_kernel(DupBatch b) {
  __shared__ double s_key[256];
  const DupJob &J = b.job[blockIdx.z];
  const int n = min(*J.n_src, b.max_n);
  const int i0 = blockIdx.x * 256, t0 = blockIdx.y * 256;
  if (i0 >= n || t0 >= n) return;
  const int i = i0 + threadIdx.x;
  if (t0 + (int)threadIdx.x < n) s_key[threadIdx.x] = dup_key(J.src, n, t0 + threadIdx.x, b.mode);
  __syncthreads();
  if (i >= n) return;
  const double mine = dup_key(J.src, n, i, b.mode);
  const int m = min(256, n - t0);
  int part = 0;
  for (int q = 0; q < m; q++) {
    const double kq = s_key[q];
    part += (kq < mine || (!(mine < kq) && t0 + q < i)) ? 1 : 0;     // (key, index) pairs in lexicographic order
  }
  if (part) atomicAdd(&dj_rank(b, blockIdx.z)[i], part);
}

// grid = (tiles, 1, jobs), block 256
__global__ __launch_bounds__(256) void dup_scatter_kernel(DupBatch b) {
  const DupJob &J = b.job[blockIdx.z];
  const int n = min(*J.n_src, b.max_n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int rank = (b.mode >= 1 && b.mode <= 3) ? dj_rank(b, blockIdx.z)[i] : i;
  const double *u = (const double *)(J.src + tent_u6_off((size_t)n)) + (size_t)i * 6;
  dj_order(b, blockIdx.z)[rank] = i;
  dj_xy4(b, blockIdx.z)[rank] = make_double4(u[0], u[1], u[3], u[4]);
}

// grid = (tiles, tiles, jobs), block 256: block (x, y <= x) = the sorted positions x * 256 .. against the earlier positions y * 256 ..
__global__ __launch_bounds__(256) void dup_near_kernel(DupBatch b) {
  __shared__ double4 s_xy[256];
  const DupJob &J = b.job[blockIdx.z];
  const int n = min(*J.n_src, b.max_n);
  const int p0 = blockIdx.x * 256, t0 = blockIdx.y * 256;
  if (p0 >= n || t0 > p0) return;
  const double4 *xy4 = dj_xy4(b, blockIdx.z);
  const int p = p0 + threadIdx.x;
  if (t0 + (int)threadIdx.x < n) s_xy[threadIdx.x] = xy4[t0 + threadIdx.x];
  __syncthreads();
  if (p >= n) return;
  const double4 me = xy4[p];
  const int m = min(256, p - t0);      // earlier positions only
  int *cnt = dj_cnt(b, blockIdx.z) + p, *near = dj_near(b, blockIdx.z) + (size_t)p * DUP_K;
  for (int q = 0; q < m; q++) {
    const double4 o = s_xy[q];
    double ex = o.x - me.x, ey = o.y - me.y;
    if (ex * ex + ey * ey > b.r_sq) continue;
    ex = o.z - me.z; ey = o.w - me.w;
    if (ex * ex + ey * ey <= b.r_sq) {
      const int slot = atomicAdd(cnt, 1);
      if (slot < DUP_K) near[slot] = t0 + q;
    }
  }
}

// grid = jobs, one workgroup of 1024 threads each; dynamic LDS: one state byte per position
__global__ __launch_bounds__(1024) void dup_resolve_kernel(DupBatch b, int lds_n) {
  extern __shared__ unsigned char s_state[];        // 0 open, 1 kept, 2 dropped
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const DupJob &J = b.job[blockIdx.x];
  int *n_dst = J.n_dst, *status = J.status;          // pinned host memory: the caller's copy
  int *res = dj_rank(b, blockIdx.x);                 // {kept, status} for dup_copy_kernel, in the (dead) rank array: device memory
  const int *near = dj_near(b, blockIdx.x), *near_cnt = dj_cnt(b, blockIdx.x);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = *J.n_src;
  if (n > b.max_n || n > lds_n) {                    // (uniform)
    if (tid == 0) { *status = 1; *n_dst = 0; res[0] = 0; res[1] = 1; }
    return;
  }
  int over = 0;
  for (int p = tid; p < n; p += 1024) {
    const int c = near_cnt[p];
    over |= c > DUP_K ? 1 : 0;
    s_state[p] = c == 0 ? 1 : 0;
  }
  if (__syncthreads_or(over)) {
    if (tid == 0) { *status = 1; *n_dst = 0; res[0] = 0; res[1] = 1; }
    return;
  }
  for (;;) {
    int changed = 0;
    for (int p = tid; p < n; p += 1024) {
      if (s_state[p] != 0) continue;
      const int c = near_cnt[p];
      bool any_kept = false, all_settled = true;
      for (int q = 0; q < c; q++) {
        const unsigned char s = s_state[near[(size_t)p * DUP_K + q]];
        any_kept |= s == 1;
        all_settled &= s != 0;
      }
      if (any_kept) { s_state[p] = 2; changed = 1; }
      else if (all_settled) { s_state[p] = 1; changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  // the kept positions' places in the packed output (sorted order), left in the near-count array - dead by now - for
  // dup_copy_kernel: one workgroup moving 200 bytes per kept correspondence was most of this kernel's time
  int *slot_of = dj_cnt(b, blockIdx.x);
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int p0 = 0; p0 < n; p0 += 1024) {
    const int p = p0 + tid;
    const bool keep = p < n && s_state[p] == 1;
    const unsigned long long mm = __ballot(keep);
    if (lane == 0) s_wave[wv] = __popcll(mm);
    __syncthreads();
    int off = s_base;
    for (int q = 0; q < wv; q++) off += s_wave[q];
    if (p < n) slot_of[p] = keep ? off + __popcll(mm & ((1ull << lane) - 1ull)) : -1;
    __syncthreads();
    if (tid == 0) { int t = s_base; for (int q = 0; q < 16; q++) t += s_wave[q]; s_base = t; }
    __syncthreads();
  }
  const int total = s_base;
  if (tid == 0) { *n_dst = total; *status = 0; res[0] = total; res[1] = 0; }
}

// grid = (tiles of 256 positions, jobs): the kept correspondences of a resolved list move to their places in the packed output
__global__ __launch_bounds__(256) void dup_copy_kernel(DupBatch b) {
  const DupJob &J = b.job[blockIdx.y];
  const int *res = dj_rank(b, blockIdx.y);           // left by dup_resolve_kernel in device memory (a read of the caller's pinned words
  if (res[1] != 0) return;                           //  would be a PCIe round trip per wave)
  const int n = *J.n_src, total = res[0];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int slot = dj_cnt(b, blockIdx.y)[p];
  if (slot < 0) return;
  const int i = dj_order(b, blockIdx.y)[p];
  const char *src = J.src; char *dst = J.dst;
  const mods_tentative *st = (const mods_tentative *)src;
  const double *su = (const double *)(src + tent_u6_off((size_t)n)), *sl = (const double *)(src + tent_laf_off((size_t)n));
  mods_tentative *dt = (mods_tentative *)dst;
  double *du = (double *)(dst + tent_u6_off((size_t)total)), *dl = (double *)(dst + tent_laf_off((size_t)total));
  dt[slot] = st[i];
#pragma unroll
  for (int q = 0; q < 6; q++) du[(size_t)slot * 6 + q] = su[(size_t)i * 6 + q];
#pragma unroll
  for (int q = 0; q < 14; q++) dl[(size_t)slot * 14 + q] = sl[(size_t)i * 14 + q];
}

// Queues the filter behind the searches that leave their packed lists at job[i].src and the lengths at *job[i].n_src
// (device-readable): the kept correspondences go to job[i].dst (room for the source list), their number to *n_dst and 0 / 1 (not
// filtered: filter on the host) to *status - both in pinned host memory, valid after the stream has been synchronised.
// grid_n: an upper bound of the list lengths known to the caller (sizes the launches; 0 = the context's capacity).
static size_t dup_job_stride(const mods_ctx *c) {
  const size_t n = (size_t)c->max_cand;
  return (n * sizeof(double4) + n * sizeof(int) * (1 + DUP_K) + 255) & ~(size_t)255;
}
// scratch for the lists of n_jobs pairs (mods_ctx_warmup calls this ahead of the pipeline: a hipMalloc synchronises the device)
int dup_filter_reserve(mods_ctx *c, int n_jobs) {
  if (n_jobs < 1 || n_jobs > DUP_MAX_JOBS) { set_error("duplicate filter: %d lists", n_jobs); return MODS_E_ARG; }
  if (c->dd_jobs >= n_jobs) return MODS_OK;
  if (c->dd_buf) { MODS_HIP_CHECK(mods::stream_wait(c->stream)); MODS_HIP_CHECK(hipFree(c->dd_buf)); c->dd_buf = nullptr; c->dd_jobs = 0; }
  MODS_HIP_CHECK(hipMalloc(&c->dd_buf, (dup_job_stride(c) + 2 * (size_t)c->max_cand * sizeof(int)) * n_jobs));
  c->dd_jobs = n_jobs;
  return MODS_OK;
}

int dup_filter_dev(mods_ctx *c, const DupJob *jobs, int n_jobs, int grid_n, double r, int mode) {
  { const int rrc = dup_filter_reserve(c, n_jobs); if (rrc) return rrc; }
  const size_t n = (size_t)c->max_cand;
  const size_t stride = dup_job_stride(c);
  const size_t ctr_job = 2 * n * sizeof(int);
  DupBatch b;
  b.n_jobs = n_jobs; b.max_n = c->max_cand; b.mode = mode; b.r_sq = r * r;   // the packed layout of a source list is that of min(*n_src, max_cand) entries (match_emit_kernel)
  b.scratch = (char *)c->dd_buf; b.stride = stride;
  b.counters = (int *)((char *)c->dd_buf + stride * c->dd_jobs);
  for (int i = 0; i < n_jobs; i++) b.job[i] = jobs[i];
  if (grid_n > c->max_cand || grid_n < 1) grid_n = c->max_cand;
  static DynLdsOnce once;
  MODS_HIP_CHECK(dyn_lds_once(once, (const void *)dup_resolve_kernel, DUP_MAX_N, c->device));
  const int tiles = (grid_n + 255) / 256;
  // rank and near-count arrays start at zero: one fill over the counters of the batch's jobs, whole (a list longer than the caller's
  // grid_n bound - which the resolve kernel hands to the host - must not meet the previous call's place numbers either)
  MODS_HIP_CHECK(hipMemsetAsync(b.counters, 0, ctr_job * n_jobs, c->stream));
  if (mode >= 1 && mode <= 3) hipLaunchKernelGGL(dup_rank_kernel, dim3(tiles, tiles, n_jobs), dim3(256), 0, c->stream, b);
  hipLaunchKernelGGL(dup_scatter_kernel, dim3(tiles, 1, n_jobs), dim3(256), 0, c->stream, b);
  hipLaunchKernelGGL(dup_near_kernel, dim3(tiles, tiles, n_jobs), dim3(256), 0, c->stream, b);
  const int lds = (int)std::min((size_t)DUP_MAX_N, ((size_t)grid_n + 63) & ~(size_t)63);
  hipLaunchKernelGGL(dup_resolve_kernel, dim3(n_jobs), dim3(1024), (size_t)lds, c->stream, b, lds);
  hipLaunchKernelGGL(dup_copy_kernel, dim3(tiles, n_jobs), dim3(256), 0, c->stream, b);
  MODS_HIP_CHECK(hipGetLastError());
  return MODS_OK;
}

}  // namespace mods
