// Internal declarations of libmodsgpu (context, device buffers, launch helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/mods_hip.h"

namespace mods {

void set_error(const char *fmt, ...);

#define MODS_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      mods::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MODS_E_HIP;                                                                  \
    }                                                                                     \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device: a function's attributes live with the device's
// code object, and a process may hold contexts on several GPUs (mods_multi, MODS_DEVICES).  `site` = a static per call site.
struct DynLdsOnce { std::atomic<unsigned> done{0}; };
inline hipError_t dyn_lds_once(DynLdsOnce &site, const void *fn, int bytes, int device) {
  const unsigned bit = 1u << (device & 31);
  if (site.done.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) site.done.fetch_or(bit, std::memory_order_release);
  return e;
}

// How the calling thread waits for a stream.  The runtime's hipStreamSynchronize spins on the completion signal: a pipeline
// worker that waits 40 ms for its batch of pairs then burns a whole core, and eight ranks of a node need the cores for the
// verification.  A thread that sets tl_wait_sleep_ns > 0 (the pipeline's workers do) polls hipStreamQuery instead: a short
// run of back-to-back queries for work that is about to finish, then a nanosleep between queries.  No interrupt path of the
// runtime is involved (hipDeviceScheduleBlockingSync hung on the test boxes).  Everything else keeps the runtime's wait.
inline thread_local long tl_wait_sleep_ns = 0;
inline thread_local long tl_wait_sleep_max_ns = 0;   // the sleep between two polls grows by half per poll up to this (a long wait costs few polls, a short one stays sharp)
inline thread_local int tl_wait_spin_polls = 8;
hipError_t stream_wait(hipStream_t s);
// hipMemcpy / hipMemset that wait, on a stream of the library instead of the legacy stream: an operation on the legacy stream waits
// for every blocking stream of the process and is REFUSED (hipErrorStreamCaptureImplicit) while another thread records a stream
// (mods_ctx_graphs: a pipeline worker recording its launch chain) - seen once as a failed workspace growth of a verify thread
hipError_t copy_wait(hipStream_t s, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t fill_wait(hipStream_t s, void *dst, int value, size_t bytes);
hipStream_t thread_stream(int dev = -1);   // a non-blocking stream of the calling thread on device `dev` (-1: the current one; entry points without a context)
int device_of_pointer(const void *p);      // the device a device pointer lives on, -1 when the runtime does not know it as one
void wait_mode_for_worker(long default_sleep_ns);   // MODS_SYNC=spin|sleep[:us] decides for the pipeline's threads

constexpr int kMaxOctaves = 16;
constexpr int kMaxLevels = 8;        // numberOfScales + 2 <= 8
constexpr int kMaxBlurRadius = 16;   // fused separable blur: ksize <= 33
constexpr int kAltTapStride = 512;   // widest kernel of the DoG / Harris response blurs (ksize <= 511)

// One octave of the scale space for a batch of images: plane(b) = base + b * w * h.
struct OctaveDev {
  int w, h;
  float pixelDistance;
  float sigma[kMaxLevels];
  float *blur[kMaxLevels];
  float *resp[kMaxLevels];
  unsigned int *omap;                // dedup map (pyramid.cpp octaveMap), one u32 per pixel
};

struct PyramidDev {
  int n_oct;
  int n_levels;                      // numberOfScales + 2
  OctaveDev oct[kMaxOctaves];
};

// raw NMS hit / localisation record, device side (AoS, 64 B)
struct CandDev {
  int octave, level, r0, c0;
  int r, c;
  float x, y, s, pixelDistance, response;
  int type;
  int state;                         // 0 rejected, 1 passed tests (pending dedup), 2 accepted
  float a11, a12, a21, a22;          // Baumberg result
};

struct StageTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<hipEvent_t> pool;
  double total_ms = 0;
  int launches = 0;
  double bytes = 0;
};

}  // namespace mods

__host__ __device__ inline size_t tent_u6_off(size_t n) { return (n * sizeof(mods_tentative) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t tent_laf_off(size_t n) { return tent_u6_off(n) + n * 6 * sizeof(double); }
__host__ __device__ inline size_t tent_bytes(size_t n) { return tent_laf_off(n) + n * 14 * sizeof(double); }

struct mods_ctx {
  int device = 0;
  int n_cu = 256;                    // compute units of the device (sizes the persistent grids)
  int max_w = 0, max_h = 0, batch = 1;
  hipStream_t stream = nullptr;
  hipEvent_t pyr_scope_begin = nullptr;   // MODS_STAGE_PYRAMID: opened by pyramid_build, closed by detect_run after the compaction
  // the octaves from the third on are built on a side stream next to the large octaves' last level and their NMS (pyramid_build
  // forks, detect_run joins before the compaction)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool pyr_side = false;
  int pyr_side_first = 0;                 // the first octave built on the side stream
  int pyr_streams = 2;                    // mods_ctx_pyramid_streams: 1 keeps the whole scale space on ctx->stream
  // scale space
  mods::PyramidDev pyr;              // host copy of the descriptor table
  mods::PyramidDev *pyr_dev = nullptr;
  mods::PyramidDev pyr_dev_image;    // what pyr_dev holds (pyramid_configure uploads the table only when it changed)
  bool pyr_dev_valid = false;
  float *plane_pool = nullptr;       // all blur/response planes
  size_t plane_pool_elems = 0;
  unsigned int *omap_pool = nullptr;
  size_t omap_pool_elems = 0;
  bool omap_dirty = true;            // the pool holds cells that are not 0xFFFFFFFF (detect_run fills it before use)
  float *input_dev = nullptr;        // staging for host-pointer entry points
  float *tmp_dev = nullptr;
  // DoG / Harris responses (pyramid.cpp:165-194, 256-278): per-level tap tables of the response's own blur and scratch planes
  float *alt_taps_dev = nullptr;           // [kMaxLevels][kAltTapStride]
  int alt_ntap[mods::kMaxLevels] = {0};
  float alt_sigma[mods::kMaxLevels] = {0};  // sigma the tables were built for
  float *alt_planes = nullptr; size_t alt_plane_elems = 0;   // 4 planes of the first octave's size
  float *view_dev = nullptr;         // pixels of the current synthesised view (allocated on first use)
  float *gauss_taps_dev = nullptr;   // [16 slots][64] Gaussian taps
  float taps_sigma[16] = {0};        // sigma currently held by each slot (0 = empty)
  float taps_host[16][2 * mods::kMaxBlurRadius + 1] = {{0}};
  int taps_host_n[16] = {0};
  float *smm_mask_dev = nullptr;     // computeGaussMask(smmWindowSize)
  unsigned long long *baum_stats_dev = nullptr;   // per image slot: {keypoints that entered the Baumberg iteration, iterations run} (mods_baumberg_stats)
  int smm_mask_size = 0;
  // candidates
  int max_cand = 0;                  // per image
  mods::CandDev *cand = nullptr;     // [batch][max_cand]
  int *cand_count = nullptr;         // [batch] raw NMS hits
  mods_affkey *keys_dev = nullptr;   // [batch][max_cand] sorted output
  unsigned long long *sort_keys = nullptr;
  int *sort_idx = nullptr;
  int *rank_dev = nullptr;           // [batch][max_cand]
  unsigned long long *nms_mask = nullptr;   // ballot words of one octave's NMS
  size_t nms_mask_words = 0;
  int *key_count = nullptr;          // [batch]
  int *host_counts = nullptr;        // pinned
  unsigned char *u8_stage_dev = nullptr;   // [batch][max_h][max_w] staging of 8-bit host images (pair pipeline), lazily allocated
  char *pin_arena = nullptr;         // pinned host staging of a batch's tentative lists (pair pipeline), lazily allocated
  size_t pin_arena_cap = 0;
  mods_hessaff_params par;
  int reg_number_eff = -1;           // par.regionsNumber after the tilt / zoom scaling of DetectAffineKeypoints (scale-space-detector.cpp:20-21)
  int last_w = 0, last_h = 0, last_n_img = 0;
  const float *last_img_dev = nullptr; int last_stride = 0;   // the batch the pyramid was built from (sampleFromImage)
  // orientation + description
  float *desc_tables_dev = nullptr;  // [orimask 64x64][desc mask 64x64][orientation vote mask 64x64][SiftTab], offsets kTab*
  int *desc_err_dev = nullptr;
  int desc_ori_ps = 0, desc_ps = 0;
  void *ori_dev = nullptr;           // [batch][max_cand] OriOut
  void *ori_multi_dev = nullptr; size_t ori_multi_bytes = 0;   // [batch][max_cand][ori_cap] OriOut (maxAngles > 1), allocated on first use
  mods_region *regions_dev = nullptr;  // [batch][max_cand]
  mods_region *regions_half_dev = nullptr;   // HalfRootSIFT twins (allocated on first use)
  bool have_half = false;
  int *region_count = nullptr;       // [batch]
  int *inside_count = nullptr;       // [batch] keypoints that pass the centre test (the reference's unoriented list)
  std::vector<int> last_inside_counts;
  float *desc_scratch = nullptr;
  int blur_table_ps = 0;
  float *blur_table_dev = nullptr;   // per-P2 taps / resampling sequence / source indices of the LDS extraction tier (sift.hip: blur_table_kernel)
  // external descriptor (e.g. a ZMQ daemon): when set, patches go to this function instead of the SIFT kernel
  mods_descriptor_fn ext_fn = nullptr;
  void *ext_user = nullptr;
  double ext_mr = 0;
  int ext_ps = 0;
  // AffNet / OriNet in the place of Baumberg / the dominant gradient orientation (imagerepresentation.cpp:786-856, 874-900)
  mods_descriptor_fn shape_fn = nullptr, ori_fn = nullptr;
  void *shape_user = nullptr, *ori_user = nullptr;
  double shape_mr = 0, ori_mr = 0;
  int shape_ps = 0, ori_ps = 0;
  size_t desc_scratch_elems = 0;
  std::vector<int> last_region_counts;
  // matching
  int8_t *m_desc = nullptr;          // [2][pad][128] int8 descriptors (query list, train list)
  int *m_c = nullptr;                // [2][pad] precombined norms
  void *m_xy = nullptr;              // [2][pad] double2 centres
  unsigned long long *m_u64 = nullptr;
  int *m_int = nullptr;
  void *m_mid = nullptr;
  void *m_p2 = nullptr;              // pass-1 top-2 keys per train split, pass-2 query subset (see match.hip)
  size_t m_best2_cap = 0;            // entries (pairs of keys) in the top-2 table
  int m_sets = 0;                    // searches the matcher's scratch buffers hold side by side (match_ensure_buffers)
  mods_tentative *m_tent = nullptr;
  // m_tent holds the n tentatives of the last search PACKED: mods_tentative[n] | (16-byte aligned) u6[n][6] = the correspondences
  // (x1 y1 1 x2 y2 1) | laf[n][14] = the frames (x y a11 a12 a21 a22 s) of both regions - one device-to-host copy of
  // tent_bytes(n) bytes brings all three (tent_u6_off / tent_laf_off give the parts)
  int *m_count = nullptr;            // PINNED HOST memory (192 ints): [0] tentatives of the last search, [1..63] of pair i of a batch (emit kernel);
                                     // [64 + i] / [128 + i] kept correspondences / status of the device duplicate filter (i = 0: the last search)
  mods_tentative *m_tent_out = nullptr; int *m_count_out = nullptr;   // set by a batch of pairs: where match_run leaves the packed list / its length
  char *m_tent_batch = nullptr; size_t m_tent_batch_cap = 0;          // the packed lists of a batch, one segment per pair
  mods_region *m_regs = nullptr;     // [2][max_cand] staging for host-side lists
  void *dd_buf = nullptr; int dd_jobs = 0;   // duplicate filter on the device (dedup.hip): per list of a batch sorted coordinates, ranks, near lists
  mods_tentative *m_tent2 = nullptr; // the filtered packed list of the last search (single-pair path)
  std::vector<mods_tentative> h_tent;  // host copies for the sequential stages
  std::vector<double> h_u6, h_laf;
  std::vector<unsigned char> h_mask;
  void *mser = nullptr;              // MserState (mser.hip): buffers of the MSER detector, allocated on first use
  // the step loop spreads the views of a step over a few more contexts of the same GPU (imgrep.hip: run_view_jobs)
  std::vector<mods_ctx *> helpers;
  std::atomic<bool> helpers_failed{false};   // a helper context could not be made (memory): reported once, not retried
  struct StageArena { mods_region *buf = nullptr; size_t cap = 0; };
  std::vector<StageArena> helper_stage;
  // timing
  int timing_mask = 0;
  // mods_ctx_graphs: the launches of a detect + describe call replayed as one hipGraph (capi.hip: mods_detect_describe_dev)
  bool dd_graphs = false;
  // `epoch` = dev_state_epoch when the call was made: every host-side change of device tables or pools bumps that counter
  // (mods::dev_state_changed / dev_pool_reallocated), so a call behind such a change is never taken for a repeat of the one before it
  struct DdKey { const float *img = nullptr; int n_img = 0, w = 0, h = 0, stride = 0; unsigned long long par_hash = 0, epoch = ~0ull;
                 bool operator==(const DdKey &o) const { return img == o.img && n_img == o.n_img && w == o.w && h == o.h && stride == o.stride && par_hash == o.par_hash && epoch == o.epoch; } };
  unsigned long long dev_state_epoch = 0;
  std::vector<std::pair<DdKey, hipGraphExec_t>> dd_cache;   // recorded calls (a worker sees a few batch sizes), oldest first
  bool dd_stale = false;                                    // a pool the recordings point into was reallocated: they are dropped
  DdKey dd_prev;                                            // arguments of the context's previous detect + describe call
  std::vector<DdKey> dd_linear;                             // arguments whose recording had no second branch: never replayed (see dd_run)
  bool pyr_forked = false;                                  // the last pyramid_build put octaves on the side stream
  long dd_replays = 0;
  mods::StageTimer timers[MODS_STAGE_COUNT];
};

namespace mods {

// float offsets of the tables in mods_ctx::desc_tables_dev
constexpr int kTabOriMask = 0, kTabDescMask = 4096, kTabVoteMask = 8192, kTabSift = 12288;

// A hipGraph replay of a detect + describe call is valid only while the device tables and pools it was recorded against are
// untouched: every upload, setter and (re)allocation that changes them from the host goes through one of these two
inline void dev_state_changed(mods_ctx *c) { c->dev_state_epoch++; }                          // tables / parameters refreshed from the host
inline void dev_pool_reallocated(mods_ctx *c) { c->dev_state_epoch++; c->dd_stale = true; }   // a pool moved: recordings are dropped
struct StageScope {                  // brackets launches of one stage with events when enabled
  mods_ctx *ctx; int stage; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
  StageScope(mods_ctx *c, int s, double bytes = 0);
  ~StageScope();
};

// pyramid.hip
int pyramid_configure(mods_ctx *ctx, int w, int h, int n_img, const mods_hessaff_params *par);
int pyramid_build(mods_ctx *ctx, const float *img_dev, int stride);
int pyramid_join_side(mods_ctx *ctx);   // joins a forked pyramid's side stream into ctx->stream (no-op without a pending fork)
int launch_gauss_blur(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, float sigma);
int launch_hessian_response(mods_ctx *ctx, const float *src, float *dst, int w, int h, int n_img, float norm);
int launch_resize_half(mods_ctx *ctx, const float *src, float *dst, int w, int h, int dw, int dh, int n_img);
void resize_half_dims(int w, int h, int *dw, int *dh);
int gauss_ksize(float sigma);
void gauss_kernel_host(int n, double sigma, float *out);
void gauss_mask_host(int size, float *out);
void circular_gauss_mask_host(int size, float sigma, float *out);

// detect.hip
int detect_run(mods_ctx *ctx);       // NMS -> localise -> dedup -> Baumberg -> sort, for the configured batch

// mser.hip
int detect_any(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, int stride, const mods_hessaff_params *par, double tilt,
               double zoom);         // the detector the parameter set names: scale space (pyramid + detect_run) or MSER
void mser_release(mods_ctx *ctx);

// match.hip
int match_run(mods_ctx *ctx, const mods_region *q_dev, int n_q, const mods_region *t_dev, int n_t, double ratio,
              double contradDist, int nn);
int match_run_distance(mods_ctx *ctx, const mods_region *q_dev, int n_q, const mods_region *t_dev, int n_t, double threshold);   // MatchFLANNDistance, Hamming
int match_ensure_buffers(mods_ctx *ctx, int n_sets = 1);
int match_run_group(mods_ctx *ctx, int n_jobs, const mods_region *const *q_dev, const int *n_q, const mods_region *const *t_dev, const int *n_t,
                    mods_tentative *const *tent_out, int *const *count_out, double ratio, double contradDist, int nn);   // <= 16 searches in one set of launches

// dedup.hip
constexpr int DUP_MAX_JOBS = 64;
struct DupJob { const char *src; char *dst; const int *n_src; int *n_dst; int *status; };
int dup_filter_dev(mods_ctx *c, const DupJob *jobs, int n_jobs, int grid_n, double r, int mode);
int dup_filter_reserve(mods_ctx *c, int n_jobs);
int launch_fast_sqrt_selftest(mods_ctx *ctx, unsigned long long *out5_host);   // describe.hip
int launch_blur_table(mods_ctx *ctx, int ps);   // sift.hip
bool ransac_profile_on();     // MODS_RANSAC_PROF: per-call breakdown of the verification on stderr (ransac.hip)
int ransac_profile_mode();    // 0 off, 1 wall time, 2 the calling thread's CPU time (MODS_RANSAC_PROF=cpu)

// describe.hip
int describe_run(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, const mods_describe_params *par);
int describe_run_view(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, const mods_describe_params *par, const double *H,
                      int orig_w, int orig_h, mods_region *det_copy_dev);
int describe_configure(mods_ctx *ctx, const mods_describe_params *par);
int launch_dominant_angle_test(mods_ctx *ctx, const float *patch_dev, int ps, double th, float *out_dev);
int launch_sift_patch_test(mods_ctx *ctx, const float *patch_dev, int ps, int root, double max_bin, uint8_t *out_dev);


}  // namespace mods

// capi.hip: the packed output of the last search in one device-to-host copy, split into the caller's arrays (synchronises the stream)
extern "C" int mods_match_copy_out(mods_ctx *c, int n, mods_tentative *tent, double *u6, double *laf);
