// Per-thread GPU workspace shared by the homography and the fundamental-matrix verification
// (the reference's C entry points carry no context argument, so the workspace is thread-local).
#pragma once
#include "common.hpp"
#include <mutex>

namespace mods {

struct RansacGpu {
  int device = -1;
  hipStream_t stream = nullptr;
  double *u_dev = nullptr; size_t u_cap = 0;
  void *hyp_dev = nullptr; void *hyp_host = nullptr; int hyp_cap = 0;   // hyp_cap slots of HYP_SLOT_BYTES
  double *d_dev = nullptr; double *gain_dev = nullptr; size_t dg_cap = 0;
  int *counts_dev = nullptr; double *J_dev = nullptr;
  int *counts_host = nullptr; double *J_host = nullptr;
  bool counts_dirty = false;                            // a scoring round did not complete: counts_dev is cleared before the next one
  double *row_host = nullptr; size_t row_cap = 0;
  double *aux_dev = nullptr; size_t aux_cap = 0;        // second point set (off-plane correspondences of rFtH)
  // two-point candidates of rFtH as index pairs and their counts: two slots of cand_cap candidates each (a block is counted while
  // the next one is drawn), an event per slot behind its count's copy back
  unsigned int *cand_dev = nullptr, *cand_host = nullptr;
  int *candc_dev = nullptr, *candc_host = nullptr; int cand_cap = 0;
  hipEvent_t cand_ev[2] = {nullptr, nullptr};
  // models counted over all correspondences in one launch (innerFH's samples): k x 9 doubles in, k x COUNT_PARTS partial counts out,
  // both in mapped host memory
  double *cntf_host = nullptr, *cntf_dev = nullptr; int *cntc_host = nullptr, *cntc_dev = nullptr; int cntf_cap = 0;
  double score_ms = 0; long launches = 0;
  ~RansacGpu();
};
enum { HYP_SLOT_BYTES = 27 * 8 };   // largest hypothesis record (homography + its two symmetric operands)

RansacGpu *ransac_gpu();                                  // nullptr + mods_last_error when no device
bool ransac_ws_reserve(RansacGpu *ws, int len, int n_hyp);
bool ransac_fetch_row(RansacGpu *ws, int len, int k, double *dst);
bool ransac_counts_begin(RansacGpu *ws);                 // start of a scoring round (clears counters a failed round left behind)
long ransac_pinned_seed();                                // >= 0: pinned (mods_ransac_pin_seed / MODS_RANSAC_SEED)

// lane k adds gain[i][k], i = 0..len-1, in correspondence order (the MSAC score is a sequential sum)
__global__ void ransac_gain_kernel(const double *__restrict__ gain, int len, int n_hyp, int kstride, int *__restrict__ counts,
                                   double *__restrict__ J_out, int *__restrict__ counts_out);

// A device failure inside the control loops unwinds to the extern "C" entry point (the reference's signatures have no
// error channel): the entry point returns "no model" and raises the calling thread's failure flag, which
// mods_loransac_h / mods_loransac_f turn into MODS_E_HIP; the message is in mods_last_error().
struct RansacDeviceError {};
[[noreturn]] inline void ransac_fail() { throw RansacDeviceError(); }
void ransac_set_failed(int failed);       // calling thread's flag
int ransac_failed();

#define RS_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return false; } } while (0)

}  // namespace mods
