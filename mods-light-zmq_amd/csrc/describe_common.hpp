// Shared declarations of the orientation / description stage (describe.hip, sift.hip).
#pragma once
#include "common.hpp"

namespace mods {

struct DescConst {
  int w, h;
  int max_cand, max_reg;   // strides of the key / region lists
  int reg_cap;             // regions that can be described per image (patch store capacity)
  double ks;               // synth-detection.cpp:21  k_sigma = 2*3*sqrt(3)
  int ori_ps;
  double ori_i2p;          // imageToPatchScale of DetectOrientation = (2*int(mrSize)+1)/patchSize
  int max_angles;
  int ori_cap;             // oriented copies kept per keypoint: 1, or min(maxAngles, 18) (18 = the most peaks a 36-bin histogram has)
  double ori_th;
  int ori_half;            // doHalfSIFT of EstimateDominantAnglesFunctor
  int add_upright;         // [DominantOrientation] addUpRight: unrotated copies, ahead of the oriented ones
  int half_desc;           // sift_kernel: HalfRootSIFT (64 values) instead of the 128-value descriptor
  double desc_mr;
  int desc_ps;
  int photo, root;
  double max_bin;
  int p2_lo, p2_hi;        // size tier handled by a launch: p2_lo < P2 <= p2_hi (P2 = 0: direct branch)
  size_t scratch_stride;   // floats per block
  int tap_cap;
  // synthesised view (H != I): keypoints live in the view frame; the inside / touch-boundary tests of
  // ReprojectRegions* run on their reprojection into the original image (ow x oh) through Hinv
  int patch_rule;          // size of the sampled region: 0 = DescribeRegions (2*ceil(s*mr)+1, synth-detection.hpp:189),
                           // 1 = ExtractPatchesColumn (the same for odd patch sizes, 2*ceil(s*mr) for even ones, synth-detection.cpp:57),
                           // 2 = the fast branch of DescribeRegions (direct interpolation at (2*int(mr*s)+1)/patchSize, :232-253)
  int view;                // 0: identity view (reproj_kp == det_kp)
  int ow, oh;
  double Hinv[6];          // affine part of inv(H), row-major 2x3
};

struct SiftTab {           // precomputeBinsAndWeights, siftdesc.cpp:22-71 (host built, patchSize <= 64)
  int bin0[64], bin1[64];  // already multiplied by orientationBins
  float w0[64], w1[64];    // stored as double in the reference but float valued
};

// sift.hip
int launch_extract_and_sift(mods_ctx *ctx, const float *img_dev, int n_img, DescConst k, const float *dmask, const SiftTab *tab,
                            bool run_sift = true);
// HalfRootSIFT twins of the described regions: copies regions_dev to regions_half_dev and overwrites the descriptors from the
// patch store left by launch_extract_and_sift (block-per-region SIFT kernel)
int launch_half_sift(mods_ctx *ctx, int n_img, DescConst k, const float *dmask, const SiftTab *tab);
int launch_sift_patch_test(mods_ctx *ctx, const float *patch_dev, int ps, int root, double max_bin, uint8_t *out_dev);

}  // namespace mods
