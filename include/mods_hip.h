/*
 * mods_hip.h — C ABI of libmodsgpu.so, the MI355X (gfx950) implementation of the
 * detect -> describe -> match -> verify hot path of MODS (ducha-aiki/mods-light-zmq).
 *
 * Plain pointers and sizes only.  Every entry point names the reference interface it
 * replaces (file:line relative to the reference root); INTEGRATION.md shows the binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - images: row-major fp32, single channel, `stride` in elements (the reference passes
 *     contiguous CV_32FC1 cv::Mat, imagerepresentation.cpp:293-302).
 *   - all functions return 0 on success, a negative MODS_E_* code otherwise; nothing is
 *     computed on the CPU when the GPU is missing (MODS_E_NODEVICE).
 *   - `*_dev` variants take device pointers (HBM resident inputs/outputs); the others take
 *     host pointers and stage through the context's pinned buffers.
 */
#ifndef MODS_HIP_H
#define MODS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODS_OK 0
#define MODS_E_NODEVICE (-1)   /* no HIP device / extension unusable: never falls back to CPU */
#define MODS_E_ARG (-2)
#define MODS_E_CAPACITY (-3)   /* an output or internal list overflowed its capacity */
#define MODS_E_HIP (-4)        /* a HIP runtime call failed; see mods_last_error() */

typedef struct mods_ctx mods_ctx;

/* [HessianAffine] keys of the .ini (io_mods.cpp:167-207) = PyramidParams + AffineShapeParams
 * (detectors/structures.hpp:114-150, detectors/affinedetectors/affine.h:26-68). */
typedef struct mods_hessaff_params {
  int numberOfScales;          /* 3 */
  float initialSigma;          /* 1.6 */
  float threshold;             /* 5.33 */
  float edgeEigenValueRatio;   /* 10 */
  int border;                  /* 5 */
  int maxIterations;           /* max_iter = 16 */
  float convergenceThreshold;  /* 0.05 */
  int smmWindowSize;           /* 19 */
  int doBaumberg;              /* 1 */
  /* keypoint selection, AffineDetector::prepareKeysForExport (scale-space-detector.hpp:126-198); in every mode but FixedTh the
   * detector's thresholds are 0 (pyramid.h:58-59): every 3x3x3 extremum is localised and adapted, then the sorted list is cut */
  int mode;                    /* [HessianAffine] mode: MODS_DET_* (FixedTh) */
  float relativeThreshold;     /* RelativeTh: keep |response| > relativeThreshold * max|response| */
  int regionsNumber;           /* FixedRegNumber: the strongest regionsNumber; NotLessThanRegions: at least that many */
  float relativeRegionsNumber; /* RelativeRegNumber: the strongest floor(relativeRegionsNumber * n) */
  /* which response the scale space is searched in: the [HessianAffine] / [DoG] / [HarrisAffine] sections of the .ini fill the
   * same parameter set and differ in DetectorType (io_mods.cpp:167-169, 208-210, 260-262).  ScaleSpaceDetector::Response,
   * pyramid.cpp:126-163; final threshold = threshold^2 for Hessian, threshold otherwise (pyramid.h:55-56); point types
   * pyramid.cpp:65-124 */
  int detectorType;            /* MODS_DET_HESSIAN */
  int iiDoGMode;               /* DoG only: illumination-invariant rescale of the response (pyramid.cpp:172-194) */
  int sampleFromImage;         /* AffineShapeParams::sampleFromImage (affine.h:47, io_mods.cpp:184): findAffineShape samples the input
                                * image at pixel distance 1 instead of the blur level below the detection level
                                * (scale-space-detector.hpp:47-55); dense (stride == w) input only */
  /* detectorType = MODS_DET_MSER: the [MSER] section (io_mods.cpp:101-123) = extrema::ExtremaParams
   * (detectors/mser/extrema/extremaParams.h:56-89); mode, relativeThreshold, regionsNumber, relativeRegionsNumber above are
   * shared (regionsNumber of a view is scaled by 2 * zoom / tilt, extrema.cpp:201-202), everything else above is unused.
   * Replaces DetectMSERs (detectors/mser/extrema/extrema.cpp:196-295) behind DetectAffineRegions<> (imagerepresentation.cpp:780-783):
   * MSER+ regions first (sub_type 21), then MSER- (20), response = the stability margin; in every mode but FixedTh the margin
   * bound is 1 and the list is sorted by margin and cut (prepareKeysForExport, extrema.cpp:31-90). */
  double mserMaxArea;          /* max_area = 0.01 (0.05 in config_affori_classic.ini): largest region as a fraction of the image */
  double mserMinMargin;        /* min_margin = 10 (8 in the shipped .ini): the stability bound of FixedTh */
  int mserMinSize;             /* min_size = 30 pixels */
  int affBmbrgMethod;          /* [HessianAffine] affBmbrgMethod (io_mods.cpp:193; AffineBaumbergMethod, affine.h:21-24): 0 = second moment
                                * matrix, 1 = the Hessian form of the iteration (affine.cpp:92-128: 3x3 samples at s * 0.5, SVD of the
                                * Hessian); HessianAffine / DoG / HarrisAffine */
} mods_hessaff_params;
enum { MODS_DET_FIXED_TH = 0, MODS_DET_RELATIVE_TH, MODS_DET_FIXED_REG_NUMBER, MODS_DET_RELATIVE_REG_NUMBER,
       MODS_DET_NOT_LESS_THAN_REGIONS };   /* detection_mode_t, detectors/structures.hpp:10-14 */
enum { MODS_DET_HESSIAN = 0, MODS_DET_DOG = 1, MODS_DET_HARRIS = 2, MODS_DET_MSER = 3 };   /* detector_type, detectors/structures.hpp:16-19 */

/* AffineKeypoint (detectors/structures.hpp:185-195) + provenance of the pyramid hit. */
typedef struct mods_affkey {
  double x, y, s, a11, a12, a21, a22, response;
  int sub_type;                /* Hessian: 0 dark, 1 bright, 2 saddle; DoG: 10 dark, 11 bright; Harris: 30 dark, 31 bright (pyramid.h:33-40);
                                * MSER: 21 MSER+, 20 MSER- (extrema.cpp:261, 286) */
  int octave, level, r0, c0;   /* NMS cell that produced the point (MSER: threshold, polarity, seed row, seed column) */
  int pad;
} mods_affkey;

/* pyramid localisation result before affine adaptation (pyramid.cpp:281-403) */
typedef struct mods_candidate {
  int octave, level, r0, c0, r, c;
  float x, y, s, pixelDistance, response;
  int type;
} mods_candidate;

/* AffineRegion subset that travels through orientation/description/matching
 * (detectors/structures.hpp:214-229): det_kp == reproj_kp for the identity view. */
typedef struct mods_region {
  double x, y, s, a11, a12, a21, a22, response;
  int sub_type;
  int id, parent;
  int pad;
  uint8_t desc[128];           /* RootSIFT, integer valued 0..255 (siftdesc.cpp:199-222) */
} mods_region;

/* TentativeCorrespExt subset (matching/matching.hpp:39-51) */
typedef struct mods_tentative {
  int q, t;                    /* first / second: indices into the query and train lists */
  int t_bad, t_2nd;            /* secondbad / secondbadby2ndcl */
  float d1, d2, d2nd;          /* squared L2 distances */
  float pad;
  double ratio;                /* sqrt(d1/d2) */
} mods_tentative;

/* ---- context ------------------------------------------------------------------------ */
const char *mods_last_error(void);
int mods_device_count(void);
/* One context = one GPU, one stream set, one pool of HBM scratch sized for images up to
 * max_w x max_h and `batch` images per call. */
int mods_ctx_create(int device, int max_w, int max_h, int batch, mods_ctx **out);
/* flags bit 0: non-blocking stream (no implicit ordering after the default stream: inputs must be complete
 * when a call is made); lets several contexts overlap on one GPU */
int mods_ctx_create_ex(int device, int max_w, int max_h, int batch, int flags, mods_ctx **out);
void mods_ctx_destroy(mods_ctx *ctx);
int mods_ctx_sync(mods_ctx *ctx);
void *mods_ctx_stream(mods_ctx *ctx);            /* hipStream_t the kernels run on */

/* per-kernel HIP-event timing (bench.py roofline leg).  When enabled every launch of the
 * named stage is bracketed by events on the context's stream. */
enum { MODS_STAGE_BLUR = 0, MODS_STAGE_RESPONSE, MODS_STAGE_RESIZE, MODS_STAGE_NMS, MODS_STAGE_LOCALIZE,
       MODS_STAGE_BAUMBERG, MODS_STAGE_SORT, MODS_STAGE_ORIENT, MODS_STAGE_DESCRIBE, MODS_STAGE_MATCH,
       MODS_STAGE_RANSAC_SCORE, MODS_STAGE_SYNTH,
       MODS_STAGE_BLUR_SMALL,   /* blur launches of the small planes (the 16-row tile instantiation); MODS_STAGE_BLUR: 32-row tiles */
       MODS_STAGE_PYRAMID,      /* the whole scale space of a batch in ONE scope on the context's stream: every blur, response and
                                 * resize launch of every octave + NMS + compaction (the small octaves run on a second stream
                                 * inside it, so the per-stage sums above overlap and add up to more than this) */
       MODS_STAGE_MATCH_NN1,    /* the matrix-core kernel of the search alone (match_nn1_kernel, i8 MFMA), inside MODS_STAGE_MATCH */
       MODS_STAGE_EXTRACT,      /* measurement-region extraction alone (classify .. column pass + resampling), inside MODS_STAGE_DESCRIBE */
       MODS_STAGE_SIFT,         /* the SIFT kernels alone, inside MODS_STAGE_DESCRIBE */
       MODS_STAGE_COUNT };
int mods_ctx_timing_enable(mods_ctx *ctx, int stage_mask);
/* on != 0: mods_detect_describe_dev (and what is built on it: the pair entry points, the pipeline's workers) records the ~70
   launches of a call into a hipGraph the second time it sees the same arguments (image pointer, sizes, parameters) and replays
   it from then on - one submission and one completion for the host instead of one per launch.  The results are the same
   launches' results.  Only calls whose scale space forks onto the context's side stream (images x batch of 4 megapixels or more,
   pyramid streams = 2) are recorded; others stay eager (a linear recording faults on replay with the ROCm 7.0.2 runtime unless
   DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is set before the runtime starts; csrc/capi.hip: dd_run).  Off by default for a context.
   The pair pipeline used it for its workers' contexts during round 5 and does so only with MODS_PIPELINE_STREAMS=2 now: with this
   runtime a graph launch with branches - like the side stream itself - keeps the HIP runtime's own thread spinning for its
   duration (tools/ubench/rt_thread_probe.hip), a core per process that the pipeline's one-stream workers do not cost.
   mods_ctx_graph_replays: calls served by a replay so far. */
int mods_ctx_graphs(mods_ctx *ctx, int on);
long mods_ctx_graph_replays(const mods_ctx *ctx);
/* Streams the Hessian scale space of a large batch is built on: 2 (default) = the octaves from the third on, and their non-maximum
   suppression, run on a side stream next to the large octaves' last level and NMS; 1 = everything on the context's stream (what
   per-launch timing wants).  The planes and the candidates are the same either way. */
int mods_ctx_pyramid_streams(mods_ctx *ctx, int n);
/* sums since the last reset; resolves pending events (synchronises the stream) */
int mods_ctx_timing_read(mods_ctx *ctx, int stage, double *total_ms, int *launches, double *bytes);
int mods_ctx_timing_reset(mods_ctx *ctx);

/* ---- B2: detector --------------------------------------------------------------------
 * Replaces  int DetectAffineKeypoints(cv::Mat &input, vector<AffineKeypoint> &out,
 *           ScaleSpaceDetectorParams, ScalePyramid&, double tilt, double zoom)
 * (detectors/affinedetectors/scale-space-detector.cpp:13-32) followed by the
 * s*=sqrt|det A| / rectifyTransformation loop of DetectAffineRegions<>
 * (synth-detection.hpp:79-112).  Output sorted by |response| descending, cut as par->mode says. */
int mods_detect_hessian_affine(mods_ctx *ctx, const float *img, int w, int h, int stride,
                               const mods_hessaff_params *par, mods_affkey *out, int max_out, int *n_out);
/* batch of `n_img` same-size device-resident images; out[i*max_out ...], n_out[i] */
int mods_detect_hessian_affine_dev(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, int stride,
                                   const mods_hessaff_params *par, mods_affkey *out_host, int max_out,
                                   int *n_out_host);

/* Introspection used by the parity tests (no reference counterpart): planes of the last
 * pyramid built by the context. kind 0 = blur, 1 = response. */
int mods_pyramid_octaves(mods_ctx *ctx);
int mods_pyramid_dims(mods_ctx *ctx, int octave, int *w, int *h);
int mods_pyramid_plane(mods_ctx *ctx, int img, int octave, int level, int kind, float *dst_host);
int mods_pyramid_candidates(mods_ctx *ctx, int img, mods_candidate *out, int max_out, int *n_out);

/* single primitives (parity tests / building blocks; references: helpers.cpp:717-731,
 * pyramid.cpp:196-254, pyramid.cpp:476, helpers.cpp:551-626) */
int mods_gauss_blur(mods_ctx *ctx, const float *src, int w, int h, float sigma, float *dst);
int mods_hessian_response(mods_ctx *ctx, const float *src, int w, int h, float norm, float *dst);
int mods_resize_half(mods_ctx *ctx, const float *src, int w, int h, float *dst, int *dw, int *dh);

/* ---- B3: orientation + description ---------------------------------------------------------
 * [DominantOrientation] and [SIFTDescriptor] keys (io_mods.cpp:731-740, 423-436). */
typedef struct mods_describe_params {
  double ori_mrSize;       /* 5.1962 */
  int ori_patchSize;       /* 32 */
  int ori_maxAngles;       /* 1; > 1: an oriented copy per histogram peak above the threshold, the first maxAngles in bin order
                              (synth-detection.cpp:900-927, 1095-1106); <= 0: no oriented copies (`if (maxAngNum > 0)`, :1086) */
  double ori_threshold;    /* (double)(float)0.8 */
  double desc_mrSize;      /* 5.1962 */
  int desc_patchSize;      /* 41 */
  int photoNorm;           /* 1 */
  int rootSift;            /* 1 = RootSIFT, 0 = SIFT */
  double maxBinValue;      /* 0.2 */
  /* Half descriptors (imagerepresentation.cpp:725-731, 909-943, 970-979; siftdesc.cpp:401-436).  As soon as a step's
   * descriptor list names a "Half*" descriptor the reference estimates the dominant orientation in doHalfSIFT mode
   * (orientation modulo pi: histogram bins i and i + 18 folded after the threshold is taken) for EVERY descriptor of the view. */
  int ori_halfMode;        /* 0 */
  int addUpRight;          /* [DominantOrientation] addUpRight (imagerepresentation.cpp:915-930): the unrotated copy of every region
                              that passes DetectOrientation's border test is described too; those copies come first in the list */
  int halfDesc;            /* 1: also HalfRootSIFT (64 values: orientation bins j and j + 4 of the raw histogram added, then the
                              RootSIFT normalisation) for the same regions, see mods_regions_half_dev */
  int fastExtraction;      /* [SIFTDescriptor] FastPatchExtraction: the fast branch of DescribeRegions (synth-detection.hpp:232-253): one
                              interpolate() of the image at imageToPatchScale = (2*int(mrSize*s)+1)/patchSize (double), no smoothing */
} mods_describe_params;

/* Replaces, for one identity view (H = I), the chain of imagerepresentation.cpp:867-968:
 *   ReprojectRegionsAndRemoveTouchBoundary(dontRemove)  synth-detection.cpp:151-190
 *   DetectOrientation(...)                               synth-detection.cpp:1039-1149
 *   ReprojectRegions(...)                                synth-detection.cpp:631-706
 *   DescribeRegions<SIFTDescriptor>(...)                 synth-detection.hpp:170-263, matching/siftdesc.cpp
 * Input: keypoints as produced by mods_detect_hessian_affine; output: oriented, described
 * regions in input order (dropped ones removed). */
int mods_orient_describe(mods_ctx *ctx, const float *img, int w, int h, int stride, const mods_affkey *keys,
                         int n_keys, const mods_describe_params *par, mods_region *out, int max_out, int *n_out);

/* detect + orient + describe for a batch of device-resident images; regions stay in HBM for the
 * matcher (mods_regions_fetch copies them out).  n_regions_host[i] = regions of image i. */
int mods_detect_describe_dev(mods_ctx *ctx, const float *img_dev, int n_img, int w, int h, int stride,
                             const mods_hessaff_params *det, const mods_describe_params *desc,
                             int *n_detected_host, int *n_regions_host);
/* External descriptors (reference: the "ZMQ" descriptor, DescribeWithZmq, imagerepresentation.cpp:21-103, 992-1006).  While
 * a function is set, the describe stage extracts ExtractPatchesColumn's patches (patchSize x patchSize at mrSize, fp32, no
 * photometric normalisation) and hands them to it; the function returns 128 values per patch (0..255, integer valued, as
 * the HardNet daemon delivers them), which become the descriptor bytes.  libmodszmq.so provides mods_zmq_descriptor_hook
 * (user = endpoint string) with this signature. */
typedef int (*mods_descriptor_fn)(void *user, const float *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim);
int mods_ctx_set_external_descriptor(mods_ctx *ctx, mods_descriptor_fn fn, void *user, double mrSize, int patchSize);
/* External affine shape and orientation (reference: [AffineAdaptation] useZMQ=1 with AffNet, [DominantOrientation] useZMQ=1 with
 * OriNet; imagerepresentation.cpp:786-856, 874-900).  While a shape function is set, the describe stage replaces the frame of
 * every detected keypoint (detect with doBaumberg = 0) by the function's (a11, a21, a22) for its ExtractPatchesColumn patch
 * (3 values per patch), rectified "up is up", and drops keypoints by the reference's eigenvalue-ratio and border tests.  While
 * an orientation function is set, the dominant-orientation estimate is replaced by atan2(y, x) of the function's 2 values per
 * patch.  Same callback type as the descriptor (libmodszmq's mods_zmq_descriptor_hook with the daemon's endpoint as user). */
int mods_ctx_set_external_shape(mods_ctx *ctx, mods_descriptor_fn fn, void *user, double mrSize, int patchSize);
int mods_ctx_set_external_orientation(mods_ctx *ctx, mods_descriptor_fn fn, void *user, double mrSize, int patchSize);
int mods_patches_fetch(mods_ctx *ctx, int img, int ps, float *out, int max_regions, int *n_out);   /* patches of the last describe call */
/* Baumberg work counters of image slot img (bench.py's per-keypoint figures): keypoints that entered the affine-shape iteration and
 * iterations run since mods_baumberg_stats_enable(ctx, 1); one iteration = smmWindowSize^2 bilinear taps (affine.cpp:26-158). */
int mods_baumberg_stats_enable(mods_ctx *ctx, int on);
int mods_baumberg_stats(mods_ctx *ctx, int img, unsigned long long *keypoints, unsigned long long *iterations);
int mods_unoriented_count(mods_ctx *ctx, int img);   /* |"None" region list| of slot img after the last describe call */
int mods_regions_fetch(mods_ctx *ctx, int img, mods_region *out, int max_out, int *n_out);

/* parity-test building blocks: one 32x32 orientation patch / one 41x41 descriptor patch */
int mods_dominant_angle(mods_ctx *ctx, const float *patch, int ps, double th, float *angle, int *found);
/* Self-test of the square root the gradient passes use (csrc/device_util.hpp: fast_sqrtf = the compiler's correctly rounded
 * expansion without its denormal rescue) against sqrtf over all 2^32 operands.  out5: {operands of its stated domain (+0, x >= 2^-96,
 * +infinity, NaN) where it differs (must be 0), operands 0 < x < 2^-96 where it differs (allowed), non-negative operands where the
 * everywhere-exact form differs (must be 0), operands visited (2^32), negative operands where either differs (outside both
 * domains: the callers' operands are sums of squares)}. */
int mods_selftest_fast_sqrt(mods_ctx *ctx, unsigned long long *out5);
int mods_sift_patch(mods_ctx *ctx, const float *patch, int ps, int rootsift, double maxBinValue, uint8_t *out128);

/* ---- view synthesis ---------------------------------------------------------------------------
 * Replaces GenerateSynthImageCorr (synth-detection.cpp:324-518; the caller hands over a grey float image,
 * i.e. the state after the (B+G+R)/3 conversion of :343-354) and the per-view body of
 * ImageRepresentation::SynthDetectDescribeKeypoints (imagerepresentation.cpp:704-1099) for
 * HessianAffine + RootSIFT.  tilt < 0 = vertical tilt, phi in radians, as in ViewSynthParameters. */
typedef struct mods_view_geom {
  int identity;              /* 1: the original image is the view (tilt ~ 1, phi ~ 0, zoom ~ 1) */
  int w_rot, h_rot;          /* size after the rotation */
  int w_new, h_new;          /* size of the view */
  int ksize_x, ksize_y, pad; /* anti-aliasing Gaussian */
  double rotation, tilt, zoom;   /* SynthImage fields: degrees, |tilt|, zoom */
  double sigma_x, sigma_y;
  double H[9];               /* original -> view (SynthImage::H), row-major */
  double warpRot[6], warpTilt[6];   /* the two cv::warpAffine matrices (src -> dst) */
} mods_view_geom;

int mods_view_geometry(int w, int h, double tilt, double phi, double zoom, double initSigma, mods_view_geom *out);
/* src_dev: w x h floats in HBM (row stride `stride` floats); dst_dev: out->w_new x out->h_new, dense.
 * The context must hold the rotated intermediate: w_rot * h_rot <= max_w * max_h * batch
 * (max_w = max_h = ceil(hypot(w, h)) is always enough). */
int mods_synth_view_dev(mods_ctx *ctx, const float *src_dev, int w, int h, int stride, const mods_view_geom *geom, int doBlur,
                        float *dst_dev);
/* synthesise + detect + orient + reproject + describe one view; regions (reproj_kp, original frame, with
 * descriptors) stay in the context: mods_regions_fetch(ctx, 0, ...) or mods_regions_dev(ctx, 0). */
int mods_detect_describe_view_dev(mods_ctx *ctx, const float *src_dev, int w, int h, int stride, double tilt, double phi,
                                  double zoom, double initSigma, int doBlur, const mods_hessaff_params *det,
                                  const mods_describe_params *desc, mods_view_geom *geom_out, int *n_detected, int *n_regions);
/* the same view of TWO images of one size (the two images of a pair share their view schedule) in one chain of launches; the
 * context needs batch >= 2; regions of image i stay in slot i (mods_regions_fetch(ctx, i, ...)), n_detected2 / n_regions2: [2] */
int mods_detect_describe_view2_dev(mods_ctx *ctx, const float *src1_dev, const float *src2_dev, int w, int h, int stride, double tilt,
                                   double phi, double zoom, double initSigma, int doBlur, const mods_hessaff_params *det,
                                   const mods_describe_params *desc, mods_view_geom *geom_out, int *n_detected2, int *n_regions2);
const mods_region *mods_regions_dev(mods_ctx *ctx, int img);     /* device pointer of the region list of image `img` */
/* the same regions with desc[0..63] = HalfRootSIFT, desc[64..127] = 0 (filled when mods_describe_params.halfDesc was set) */
const mods_region *mods_regions_half_dev(mods_ctx *ctx, int img);
int mods_regions_fetch_half(mods_ctx *ctx, int img, mods_region *out, int max_out, int *n_out);
int mods_regions_copy_dev(mods_ctx *ctx, int img, mods_region *dst_dev, int n);   /* D2D copy of its first n entries */
const float *mods_view_pixels_dev(mods_ctx *ctx);                /* pixels of the last synthesised view (w_new x h_new) */
int mods_view_fetch(mods_ctx *ctx, const mods_view_geom *geom, float *dst_host);   /* host copy of those pixels */
/* host-buffer primitives (parity tests): cv::warpAffine(LINEAR, BORDER_CONSTANT cval), M maps src -> dst;
 * cv::GaussianBlur(Size(kx,ky), sx, sy, BORDER_REFLECT_101) */
int mods_warp_affine(mods_ctx *ctx, const float *src, int w, int h, const double *M, int dw, int dh, float cval, float *dst);
int mods_gauss_blur_xy(mods_ctx *ctx, const float *src, int w, int h, int kx, int ky, double sx, double sy, float *dst);

/* ---- B4: matching ------------------------------------------------------------------------------
 * Replaces  int MatchFlannFGINN(const AffineRegionList &q, const AffineRegionList &t,
 *           TentativeCorrespListExt &out, const MatchPars &par, const int nn = 50)
 * (matching/matching.hpp:261-262, matching.cpp:356-460) for vector_matcher = linear, vector_dist = L2:
 * exact brute-force squared-L2 search (ties: lower train index first), FGINN walk over nn neighbours.
 * ratio = par.currMatchRatio (FGINNThreshold of the iters .ini), contradDist = [Matching] contradDist.
 * Tentatives come out in query order.  u6_out (optional, 6 doubles per tentative) receives the
 * correspondences as LORANSACFiltering lays them out for degensac: x1 y1 1 x2 y2 1; laf_out (optional,
 * 14 doubles per tentative) the two local affine frames x y a11 a12 a21 a22 s used by the LAF checks. */
int mods_match_fginn(mods_ctx *ctx, const mods_region *q, int n_q, const mods_region *t, int n_t, double ratio,
                     double contradDist, int nn, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);
/* same, on the HBM-resident region lists of images img_q / img_t left by mods_detect_describe_dev */
/* MatchFLANNDistance (matching/matching.cpp:572-633) with binary_dist = Hamming (io_mods.cpp:365) and an exact index: the
 * nearest train of every query by Hamming distance over the 128 descriptor bytes is a tentative when the distance is at most
 * (int)(float)threshold; d1, d2 = the two smallest distances (ties by train index), ratio = d1 / d2. */
int mods_match_distance(mods_ctx *ctx, const mods_region *q, int n_q, const mods_region *t, int n_t, double threshold,
                        mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);
int mods_match_dev(mods_ctx *ctx, int img_q, int img_t, double ratio, double contradDist, int nn, mods_tentative *out,
                   double *u6_out, double *laf_out, int max_out, int *n_out);

/* Replaces  void DuplicateFiltering(TentativeCorrespListExt &in, const double r, const int mode)
 * (matching.cpp:2615-2679).  Host-side, in place on (tent, u6); mode 0 = keep order (MODE_RANDOM),
 * 1 = best FGINN ratio first, 2 = best distance first (configuration.hpp:31-34); equal keys keep
 * list order. */
int mods_duplicate_filter(mods_tentative *tent, double *u6, double *laf, int n, double r, int mode, int *n_out);
/* The same filter on the context's GPU (csrc/dedup.hip: rank count for the order, brute-force near-predecessor lists, fixed-point
 * resolution of the keep / drop rule; the pair entry points run it behind the search when [DuplicateFiltering] doBeforeRANSAC = 1).
 * Host lists in and out as above, laf required; identical result.  *on_device (optional) = 0 when the list was handed to the
 * host filter instead (a correspondence with more than 12 near predecessors, or more than 150 k correspondences).
 * The call stages the list in the context's tentative buffer: it replaces the "last search" that mods_match_fetch_internal and
 * the fetch entry points of the matcher read (run it after those, or search again). */
int mods_duplicate_filter_gpu(mods_ctx *ctx, mods_tentative *tent, double *u6, double *laf, int n, double r, int mode, int *n_out,
                              int *on_device);

/* ---- B1: verification ---------------------------------------------------------------------------
 * The degensac C ABI itself is exported with the reference's exact signatures (degensac/exp_ranH.h:32-36,
 * degensac/Htools.h): exp_ransacHcustom, HDs, HDsSym, HDsSymMax, HDsi, HDsiSym, HDsiSymMax, HDsidx,
 * HDsSymidx, HDsSymidxMax - see include/mods_degensac.h.  Hypotheses are scored on the GPU.
 *
 * [RANSAC] keys (io_mods.cpp:437-455).  errorType: 0 Sampson, 1 SymmMax, 2 SymmSum. */
typedef struct mods_ransac_params {
  double err_threshold;     /* 4.0 px */
  double confidence;        /* 0.99 */
  int max_samples;          /* 1000000 */
  int localOptimization;    /* 1 */
  double LAFCoef;           /* 2 (F branch) */
  double HLAFCoef;          /* 12 */
  int errorType;            /* 0 */
  int doSymmCheck;          /* 1 */
  int useF;                 /* 0: homography (ver_type "Homog"), 1: epipolar geometry (DEGENSAC, ver_type "Epipolar") */
  /* ver_type 1 of the command line (GR_TRUTH, mods.cpp:290-320): verification against a known homography.
   * groundTruth 0: off; 1: HMatrixFiltering of the tentatives only; 2: doBothRANSACgroundTruth - LORANSACFiltering first, then
   * HMatrixFiltering of its inliers (the verified list), next to HMatrixFiltering of all tentatives for the log. */
  int groundTruth;
  int ransacForStopping;    /* [Matching] RANSACforStopping: in ground-truth mode the step loop stops on the RANSAC inlier count */
  double gtH[9];            /* the homography img1 -> img2, row-major, as the file of argv[Tmin+2] holds it (mods.cpp:92-98) */
} mods_ransac_params;

/* Replaces  int HMatrixFiltering(TentativeCorrespListExt &in, TentativeCorrespListExt &true_corresp, double *H, const int isExtended,
 *           const RANSACPars pars)  (matching/matching.cpp:917-1012): mask[i] = 1 where the error of correspondence i under H
 * (errorType: 0 HDs Sampson, 1 HDsSymMax, otherwise HDsSym; the point of the second image first, as the reference stacks it) is at
 * most (float)(err_threshold^2).  H_rowmajor: img1 -> img2.  Host-side. */
int mods_hmatrix_filter(const double *u6, int n, const double *H_rowmajor, const mods_ransac_params *par, unsigned char *mask, int *n_true);

/* Replaces  int LORANSACFiltering(TentativeCorrespListExt &in, TentativeCorrespListExt &out, double *H,
 *           const RANSACPars pars)  (matching/matching.hpp:267-269, matching.cpp:637-805) for useF = 0.
 * u6: n x 6 correspondences, laf: n x 14 frames (may be NULL: LAF check skipped).  mask[i] = 1 for the
 * correspondences kept after RANSAC + NaiveHCheck + H_LAF_check; H_out row-major img1 -> img2
 * (all -1 when no model).  stats3 (optional) = {samples drawn, LO runs, orientation rejects}. */
int mods_loransac_h(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask,
                    double *H_out, int *n_inliers, int *stats3);
/* The useF = 1 branch of the same function (matching.cpp:711-726, 804-816): exp_ransacFcustom (inlLimit 0,
 * error functions by errorType: 0 FDs/exFDs, otherwise FDsSym/exFDsSym) followed by F_LAF_check
 * (matching.cpp:192-249, bound LAFCoef * err_threshold).  F_out = the matrix as degensac stores it
 * (x2^T F x1 = 0, F_out[3*c + r] = entry (r, c)), as ransac_corresp.H receives it; all -1 when n < 8.
 * stats3 = {samples drawn, LO runs, plane consensus *Ih}. */
int mods_loransac_f(const double *u6, const double *laf, int n, const mods_ransac_params *par, unsigned char *mask,
                    double *F_out, int *n_inliers, int *stats3);
/* The verification half of one step of the reference's loop (mods.cpp:278-368), in place on (tent, u6, laf):
 *   par->dup_before_ransac = 1 ([DuplicateFiltering] doBeforeRANSAC): DuplicateFiltering (mods.cpp:283), then LORANSACFiltering;
 *   par->dup_before_ransac = 0: LORANSACFiltering on every tentative, then DuplicateFiltering on the verified list
 *   (mods.cpp:357-368; TrueMatch1st = the size of the de-duplicated list, which also drives the minMatches stop).
 * On return the first *n_verified entries of tent / u6 / laf are the verified correspondences in output order and *n_unique is
 * the size of the list the RANSAC ran on.  H_out, stats3 as mods_loransac_h / _f; ms_dup / ms_ransac (optional): wall clock. */
struct mods_pair_params;
int mods_verify_tentatives(int device, const struct mods_pair_params *par, mods_tentative *tent, double *u6, double *laf, int n,
                           int *n_unique, int *n_verified, double *H_out, int *stats3, double *ms_dup, double *ms_ransac);
/* the same with the three ground-truth counts of mods_ladder_result (gt3, may be NULL; zeros outside ground-truth mode) */
int mods_verify_tentatives_ex(int device, const struct mods_pair_params *par, mods_tentative *tent, double *u6, double *laf, int n,
                              int *n_unique, int *n_verified, double *H_out, int *stats3, int *gt3, double *ms_dup, double *ms_ransac);
/* GPU used by the degensac entry points of the calling thread (default 0). */
int mods_ransac_set_device(int device);
/* The reference seeds with srand(time(NULL)) (exp_ranH.c:823).  seed >= 0 makes every call behave as if
 * time(NULL) returned `seed`; seed < 0 restores the wall clock.  MODS_RANSAC_SEED in the environment
 * has the same effect when no seed is pinned. */
void mods_ransac_pin_seed(long seed);
/* self-test hook: first n values of srand(seed); rand(); ... from the restated glibc generator */
void mods_test_glibc_rand(unsigned seed, int n, int *out);
/* self-test hooks of the fast host forms of the homography LO step (no device needed): the error functions at a given
 * SIMD width (lanes 0 = the scalar HDs / HDsSym / HDsSymMax, 1 / 4 / 8 = vector forms; MODS_E_ARG when the CPU lacks the
 * width), u2h and its moment matrix with the design matrix written out (reference_form 1: lin_hgN + cov_mat,
 * Htools.c:60-132, utools.c:172-185) or folded into 30 ordered sums (0), and inlidxs (rtools.c:155-166; lanes < 0: the list-only
 * form of the wide-threshold sets - count and list, J = 0). */
int mods_test_host_errfn(int type, const double *u6, int len, const double *H, int lanes, double *out);
int mods_test_host_u2h(const double *u6, const int *inl, int n, int reference_form, double *H);
int mods_test_host_cov(const double *u6, const int *inl, int n, int reference_form, double *Cv);
int mods_test_host_inlidxs(const double *err, int len, double th, int lanes, int *inl, unsigned *I, double *J);
/* mods_test_host_cov also takes reference_form 2 (the 30 folded sums in scalar code) and 100 + lanes (the sums through the SIMD
 * table of that width).  The checks LORANSACFiltering runs behind the homography (H -> inv(H^T), NaiveHCheck, H_LAF_check;
 * matching.cpp:745-805, 1014-1043, 250-308) over the correspondences with inl[i] != 0: lanes 0 = the scalar statement, 1 / 4 / 8 =
 * lanes-wide; mask[n], H_out[9], returns the number of survivors (< 0: MODS_E_*). */
int mods_test_host_hchecks(const double *u6, const double *laf14, int n, const unsigned char *inl, const double *Hloran,
                           const mods_ransac_params *par, int lanes, unsigned char *mask, double *H_out);
/* the same for the epipolar error functions of the F-matrix path: mode 0 FDs, 1 FDsSym, 2 exFDs, 3 exFDsSym (Ftools.c:94-209) */
int mods_test_host_fds(int mode, const double *u6, int len, const double *F, int lanes, double *p, double *w);
/* self-test hooks of the host-side pieces of the F-matrix path (no device needed; they let the CPU
 * test-suite compare the restated solvers with the reference's own degensac build, oracle/_ref):
 *   seven_point : u7 = 7 correspondences (7 x 6) -> up to 3 matrices in F27, returns their number
 *                 (-1: null space not two-dimensional)           [exp_ranF.c:884-911]
 *   u2f         : least-squares F of n >= 8 correspondences idx[] of u, optional weights w[len]   [Ftools.c:302-405]
 *   checksample : plane-degeneracy test of a 7-point sample      [DegUtils.c:42-82]
 *   inner_h     : homography LO of the degenerate branch, generator seeded with `seed`   [DegUtils.c:699-735]
 *   rfth        : plane-and-parallax search, generator seeded with `seed`, candidates counted on the host  [DegUtils.c:233-444] */
int mods_test_seven_point(const double *u7, double *F27);
void mods_test_u2f(const double *u, const int *idx, int n, const double *w, double *F);
void mods_test_u2f_form(const double *u, const int *idx, int n, const double *w, int reference_form, double *F);   /* 1: matrix written out */
int mods_test_checksample(const double *F, const double *u7, double th, double *H);
unsigned mods_test_inner_h(unsigned seed, double *H, const double *u, unsigned len, double th, unsigned iters, unsigned char *inl);
unsigned mods_test_rfth(unsigned seed, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F);
/* self-test hook of the host half of the MSER detector (csrc/mser_host.hpp; no device needed): the grey-level growth of one
 * polarity (invert = 1: MSER-) of a w x h 8-bit image - GetExtrema + FastSetOptThresholds4StableRegion (getExtrema.cpp:385-437,
 * optThresh.cpp:73-165).  out5: rows seed_x, seed_y, threshold, margin, area in output order; tree_out (optional): 3 ints per
 * pixel of the (w + 2) x (h + 2) frame = the merge tree the kernels walk (slot at entry, parent slot or 0x7fffffff, merge level).
 * Returns the number of stable thresholds. */
int mods_test_mser_grow(const unsigned char *img8, int w, int h, int min_size, double max_area, double min_margin, int invert,
                        int *out5, int max_out, int *tree_out);
/* Host threads the degenerate branch of DEGENSAC spreads its independent pieces over (csrc/ransac_pool.hpp): MODS_RANSAC_THREADS, by
 * default the cores this process may use, at most 8; 1 = the one-thread loops.  Results do not depend on it. */
int mods_ransac_host_threads(void);
/* innerH through the host SIMD evaluation of the production path (simd = 0: scalar); *next_rand = the generator's next value after
 * the call; *path = 0 one-thread loop, 1 repetitions side by side on the pool's threads, 2 side by side, then redone by the loop */
unsigned mods_test_inner_h2(unsigned seed, double *H, const double *u, unsigned len, double th, unsigned iters, unsigned char *inl, int simd,
                            int *next_rand, int *path);
/* the same through the host SIMD evaluation of the production path (simd = 0: scalar); *next_rand = the generator's next value
 * after the call, prof10 (optional) = the call's breakdown (10 doubles, ransac_f_host.hpp: g_rfth_prof) */
unsigned mods_test_rfth2(unsigned seed, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F,
                         int simd, int *next_rand, double *prof10);

/* ---- whole hot path for one image pair ---------------------------------------------------------
 * The step loop body of mods.cpp:202-383 for one step of HessianAffine + RootSIFT on identity views:
 *   SynthDetectDescribeKeypoints x2 (mods.cpp:234-251) -> MatchImgReps (:270) -> DuplicateFiltering
 *   (:283) -> LORANSACFiltering (:325). */
typedef struct mods_pair_params {
  mods_hessaff_params det;
  mods_describe_params desc;
  double fginn_ratio;        /* iters .ini: FGINNThreshold = 0.8 */
  double contradDist;        /* [Matching] contradDist = 10 */
  int nn;                    /* 50 */
  int dup_before_ransac;     /* [DuplicateFiltering] doBeforeRANSAC = 1 */
  double dup_dist;           /* duplicateDist = 2.0 */
  int dup_mode;              /* whichCorrespondenceRemains: random 0, bestFGINN 1, bestDistance 2, biggerRegion 3 */
  mods_ransac_params ransac;
} mods_pair_params;

typedef struct mods_pair_result {
  int n_detected[2];         /* regions per image after affine adaptation */
  int n_described[2];        /* descriptors per image */
  int n_tentatives;          /* after FGINN matching */
  int n_unique;              /* after duplicate filtering */
  int n_inliers;             /* after RANSAC + NaiveHCheck + H_LAF_check */
  int ransac_samples, ransac_lo, ransac_rejects;
  double H[9];               /* row-major img1 -> img2, all -1 when verification failed; with ransac.useF: F as degensac stores it */
  /* wall-clock per stage in ms, the reference's TimeLog buckets (detectors/structures.hpp:33-56) */
  double ms_detect_describe, ms_match, ms_duplicates, ms_ransac;
} mods_pair_result;

/* img_dev: [2][h][stride] fp32 in HBM (image 1, image 2).  matches_out (optional): up to max_matches
 * rows x1 y1 x2 y2 of the verified correspondences (matchings.txt layout, matching.cpp:2610-2611). */
int mods_match_pair_dev(mods_ctx *ctx, const float *img_dev, int w, int h, int stride, const mods_pair_params *par,
                        mods_pair_result *res, double *matches_out, int max_matches);

/* ---- multi-view representation and the MODS step loop ---------------------------------------------------
 * mods_view_schedule   = SetVSPars for one detector (synth-detection.cpp:191-322): views of a step that no
 *                        earlier step produced; `prev`/`n_prev` is the history, extended in place.
 * mods_imgrep          = the accumulated (HessianAffine, RootSIFT) regions of one image,
 *                        ImageRepresentation::AddRegions (imagerepresentation.cpp:637-684); HBM resident.
 * mods_match_reps      = CorrespondenceBank::MatchImgReps, separate-detector branch (correspondencebank.cpp:288-340)
 *                        for a slice of the queries (the slice is what a rank of the multi-GPU path owns).
 * mods_match_ladder_dev= the step loop of mods.cpp:202-383 on one GPU: per step, new views of both images ->
 *                        match all accumulated regions -> duplicate filter -> LO-RANSAC; stops at min_matches. */
typedef struct mods_view_par { double zoom, tilt, phi; } mods_view_par;      /* ViewSynthParameters subset */
typedef struct mods_imgrep mods_imgrep;
typedef struct mods_ladder_step {     /* one [HessianAffine<i>] section of the iterations .ini (io_mods.cpp:457-492) */
  double scale_set[8]; int n_scales;  /* ScaleSet */
  double tilt_set[8]; int n_tilts;    /* TiltSet */
  double phi;                         /* Phi: rotation density in degrees */
  double initSigma;                   /* initSigma */
  int doBlur;                         /* 1 (io_mods.cpp:475) */
  double fginn_ratio;                 /* FGINNThreshold of RootSIFT (0: RootSIFT lists are not matched) */
  int half_orientation;               /* a descriptor of the step is a Half* one: orientation in doHalfSIFT mode for all of them */
  double fginn_ratio_half;            /* FGINNThreshold of HalfRootSIFT; 0: not described / not matched.  [Matching<i>]
                                         SeparateDescriptors = RootSIFT,HalfRootSIFT: both lists are matched and joined,
                                         HalfRootSIFT first (the bank's key order, correspondencebank.cpp:114-148, 288-340) */
  double dist_threshold;              /* DistanceThreshold of RootSIFT / of HalfRootSIFT: > 0 runs MatchFLANNDistance (nearest by */
  double dist_threshold_half;         /* Hamming distance within the threshold, matching.cpp:572-633), whose tentatives REPLACE the
                                         FGINN ones of that descriptor (it clears the list it appends to) */
} mods_ladder_step;
typedef struct mods_ladder_result {
  int steps_done, n_views;            /* steps executed; views synthesised (both images) */
  int n_detected[2], n_described[2];  /* summed over views / accumulated regions per image */
  int n_unoriented[2];                /* detections inside the image, summed over views (log: UnorientedReg) */
  int n_tentatives, n_unique, n_inliers;   /* of the last step */
  int ransac_samples, ransac_lo, ransac_rejects;
  double H[9];
  double ms_detect_describe, ms_match, ms_duplicates, ms_ransac;
  /* ground-truth mode (mods_ransac_params.groundTruth), of the last step: TrueMatch1st of HMatrixFiltering over all unique
   * tentatives, Tentatives1stRANSAC (LORANSAC inliers), TrueMatch1stRANSAC (of those, the ones the ground truth confirms);
   * n_inliers = the verified list that is written out */
  int gt_true, gt_ransac_inliers, gt_true_of_ransac;
} mods_ladder_result;

int mods_view_schedule(const double *scale_set, int n_scales, const double *tilt_set, int n_tilts, double phi_base,
                       mods_view_par *prev, int *n_prev, int prev_cap, mods_view_par *out, int max_out);
int mods_imgrep_create(mods_ctx *ctx, int capacity, mods_imgrep **out);
void mods_imgrep_destroy(mods_imgrep *rep);
int mods_imgrep_clear(mods_imgrep *rep);
int mods_imgrep_count(const mods_imgrep *rep);
const mods_region *mods_imgrep_regions_dev(const mods_imgrep *rep);
int mods_imgrep_append_ctx(mods_imgrep *rep, mods_ctx *ctx, int img);      /* regions the context holds for image slot img */
int mods_imgrep_append_ctx_half(mods_imgrep *rep, mods_ctx *ctx, int img); /* their HalfRootSIFT twins */
int mods_imgrep_append_dev(mods_imgrep *rep, const mods_region *src_dev, int n);
int mods_imgrep_append_host(mods_imgrep *rep, const mods_region *src, int n);
int mods_imgrep_fetch(mods_imgrep *rep, int begin, int count, mods_region *out);
int mods_match_reps(mods_ctx *ctx, const mods_imgrep *q, int q_begin, int q_end, const mods_imgrep *t, double ratio,
                    double contradDist, int nn, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);
/* img1_dev, img2_dev: dense fp32 images in HBM; the context needs max_w = max_h >= ceil(hypot(w, h)) of the larger image. */
int mods_match_ladder_dev(mods_ctx *ctx, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                          const mods_ladder_step *steps, int n_steps, int min_matches, const mods_pair_params *par,
                          mods_imgrep *rep1, mods_imgrep *rep2, mods_ladder_result *res, double *matches_out, int max_matches);

/* The same loop with several scale-space detectors per step (the [HessianAffine<i>] / [DoG<i>] / [HarrisAffine<i>] sections of the
 * iterations file next to each other, io_mods.cpp:457-492): steps[step * n_det + d] is detector d's section of that step
 * (n_tilts = n_scales = -1: none), dets[d] its parameter set, reps1[d] / reps2[d] its region banks.  Every detector has its own
 * view history and its own lists; a detector is matched in the steps that bring new views of it, lists that a step does not
 * touch keep their tentatives (fginn_ratio / fginn_ratio_half < 0: not named in [Matching<i>]; 0: named but not searched), and
 * the joint list is the bank's key order: HalfRootSIFT before RootSIFT, detectors in the order given - list them sorted by
 * name (CorrespondenceBank::MatchImgReps / GetCorresponcesVector, correspondencebank.cpp:114-148, 286-340). */
int mods_match_ladder_dets_dev(mods_ctx *ctx, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                               const mods_ladder_step *steps, const mods_hessaff_params *dets, int n_steps, int n_det, int min_matches,
                               const struct mods_pair_params *par, mods_imgrep **reps1, mods_imgrep **reps2, mods_ladder_result *res,
                               double *matches_out, int max_matches);

/* Grouped matching of a step ([Matching<i>] GroupDetectors / GroupDescriptors, CorrespondenceBank::MatchImgReps,
 * correspondencebank.cpp:245-285): the regions of the named detectors, joined in the order they are named, are searched as ONE
 * query and ONE train list per named descriptor, with the [Matching]-wide thresholds (matchRatio<Descriptor> /
 * matchDistance<Descriptor>, io_mods.cpp:448-452), in every step that names a group - whether or not the step brought new
 * views.  The result is the bank's list of the pseudo-detector "Group"; group_pos = the number of real detectors whose name
 * sorts before "Group" (its place in the joint list). */
typedef struct mods_ladder_group {
  int n_dets, dets[8];                /* GroupDetectors as indices into the detector array, in the order given; 0: no group */
  double fginn_ratio, fginn_ratio_half;        /* matchRatioRootSIFT / matchRatioHalfRootSIFT; < 0: descriptor not in GroupDescriptors */
  double dist_threshold, dist_threshold_half;  /* matchDistanceRootSIFT / matchDistanceHalfRootSIFT (> 0: MatchFLANNDistance replaces) */
} mods_ladder_group;
int mods_match_ladder_groups_dev(mods_ctx *ctx, const float *img1_dev, int w1, int h1, const float *img2_dev, int w2, int h2,
                                 const mods_ladder_step *steps, const mods_hessaff_params *dets, const mods_ladder_group *groups /* [n_steps] or NULL */,
                                 int group_pos, int n_steps, int n_det, int min_matches, const struct mods_pair_params *par,
                                 mods_imgrep **reps1, mods_imgrep **reps2, mods_ladder_result *res, double *matches_out, int max_matches);

/* ---- one hard pair on several GPUs of a node ------------------------------------------------------
 * The views of every step (ImageRepresentation::SynthDetectDescribeKeypoints' views loop, imagerepresentation.cpp:704-1099) are
 * sharded over the devices, largest first; the described regions travel in ONE all-gather per step (RCCL over xGMI, device
 * buffers; the one host process knows all counts, so nothing else is exchanged); every device searches its slice of the query
 * rows against all trains (CorrespondenceBank::MatchImgReps, correspondencebank.cpp:234-343) and the host of device 0 filters
 * duplicates and verifies.  Same result as mods_match_ladder_dev.  devices[]: HIP device ids; a device listed twice makes
 * the exchange use plain device copies (development on a one-GPU box).  The `mods` command line takes MODS_DEVICES=0,1,... */
typedef struct mods_multi mods_multi;
int mods_multi_create(const int *devices, int n, int w, int h, int rep_capacity, mods_multi **out);
void mods_multi_destroy(mods_multi *m);
int mods_multi_uses_rccl(const mods_multi *m);
mods_imgrep *mods_multi_bank(mods_multi *m, int image);   /* regions of image 1 (0) / 2 (1) after a run; owned by m */
mods_imgrep *mods_multi_bank_det(mods_multi *m, int image, int det);   /* the same for detector `det` of a several-detector run */
int mods_match_ladder_multi(mods_multi *m, const float *img1_host, int w1, int h1, const float *img2_host, int w2, int h2,
                            const mods_ladder_step *steps, int n_steps, int min_matches, const mods_pair_params *par,
                            mods_ladder_result *res, double *matches_out, int max_matches);
/* The whole step loop of mods_match_ladder_groups_dev on several GPUs: several detectors per step (steps[step * n_det + det],
 * dets[n_det]), HalfRootSIFT lists (a job's HalfRootSIFT twins travel behind its RootSIFT regions in the same all-gather), the
 * distance matcher, grouped matching (groups[n_steps] or NULL, group_pos).  Same result as on one GPU, field by field. */
int mods_match_ladder_groups_multi(mods_multi *m, const float *img1_host, int w1, int h1, const float *img2_host, int w2, int h2,
                                   const mods_ladder_step *steps, const mods_hessaff_params *dets, const mods_ladder_group *groups /* or NULL */,
                                   int group_pos, int n_steps, int n_det, int min_matches, const struct mods_pair_params *par,
                                   mods_ladder_result *res, double *matches_out, int max_matches);
/* queries [q_begin, q_end) of bank q against bank t with either matcher: distance > 0 runs MatchFLANNDistance (Hamming threshold)
 * instead of the FGINN search; tentative indices refer to the full lists */
int mods_match_reps_any(mods_ctx *ctx, const mods_imgrep *q, int q_begin, int q_end, const mods_imgrep *t, double ratio, double contradDist,
                        int nn, double distance, mods_tentative *out, double *u6_out, double *laf_out, int max_out, int *n_out);
int mods_regions_half_copy_dev(mods_ctx *ctx, int img, mods_region *dst_dev, int n);   /* HalfRootSIFT twins of slot img, D2D */
/* host logic of the sharding (no device needed): owner[i] = device of view job i with area areas[i] */
int mods_multi_assign(const double *areas, int n, int n_dev, int *owner);

/* Pre-extracted mode (mods.cpp:196-229, 288-383): the banks were filled by the caller (mods_imgrep_append_host, e.g. from the
 * k1 / k2 files of an earlier run); one matching + duplicate filtering + verification pass, no detection. */
int mods_match_verify_reps(mods_ctx *ctx, mods_imgrep *rep1, mods_imgrep *rep2, double fginn_ratio, const mods_pair_params *par,
                           mods_ladder_result *res, double *matches_out, int max_matches);
/* plain device-memory helpers for callers that do not link a HIP runtime themselves (the mods CLI) */
int mods_host_alloc(size_t bytes, void **out);     /* pinned host memory */
int mods_host_free(void *p);
int mods_dev_alloc(size_t bytes, void **out);
int mods_dev_free(void *p);
/* Both copies are complete when the call returns.  They run on a NON-BLOCKING stream of the calling thread on the device the buffer
 * lives on: they are ordered after nothing else - not after work in flight on a context's stream, on the legacy stream or on an
 * application's own streams.  A buffer that some stream is still writing (or reading, for an upload) has to be waited for first
 * (mods_ctx_sync, or the application's own synchronisation); every entry point of this library that produces a buffer has returned
 * only after its work was complete, unless its comment says otherwise. */
int mods_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int mods_dev_download(void *dst_host, const void *src_dev, size_t bytes);

/* ---- pair pipeline ----------------------------------------------------------------------------------
 * Throughput form of the same path: `gpu_workers` threads (one context each) run detect/describe/match
 * while `verify_workers` threads run duplicate filtering + LO-RANSAC of earlier pairs (mods.cpp overlaps
 * its two images with OpenMP tasks, mods.cpp:234-251; here the overlap is across pairs).  Results are
 * returned in submission order and are identical to mods_match_pair_dev's.
 * Host cost: ~1.0 ms of process CPU per 1080p pair: the pipeline's threads wait on recorded events with sleeping
 * polls (MODS_SYNC=spin | sleep:<us> changes that), and its workers' contexts keep ONE stream and eager launches
 * (MODS_PIPELINE_STREAMS=2: side stream for the small octaves + hipGraph replay per worker - same results, same rate, one more
 * busy core: the runtime's own thread spins while a dependency between two streams is pending). */
typedef struct mods_pipeline mods_pipeline;
int mods_pipeline_create(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                         mods_pipeline **out);
/* pairs_per_batch > 1: a GPU worker takes up to that many queued pairs through detect/describe as one batch of
 * launches (same results; larger launches, fewer of them per pair) */
int mods_pipeline_create_ex(int device, int w, int h, const mods_pair_params *par, int gpu_workers, int verify_workers,
                            int pairs_per_batch, mods_pipeline **out);
int mods_pipeline_capacity(const mods_pipeline *p);    /* pairs that may be in flight before submit blocks */
/* HIP-event timing of the workers' contexts (sums over them); enable/read while nothing is in flight */
int mods_pipeline_timing_enable(mods_pipeline *p, int stage_mask);
int mods_pipeline_timing_read(mods_pipeline *p, int stage, double *total_ms, int *launches, double *bytes);
long mods_pipeline_graph_replays(mods_pipeline *p);   /* batches whose detect + describe chain was a graph replay (mods_ctx_graphs) */
/* CPU seconds the GPU workers' / the verify workers' own threads have spent inside their stages since the last reset (thread clocks;
   the RANSAC task pool's helper threads are not in them).  What a pair costs the host: needed to size ranks per node. */
int mods_pipeline_cpu_seconds(mods_pipeline *p, double *gpu_workers_s, double *verify_workers_s, int reset);
int mods_pipeline_submit(mods_pipeline *p, const float *img_dev, long tag);
/* the pair in HOST memory, [2][h][w] fp32 or 8-bit grey (what cv::imread hands to the ImageRepresentation constructor,
 * mods.cpp:111-121, 184-185): uploaded on the worker's stream (asynchronously when the memory is pinned, see
 * mods_host_alloc); the memory must stay valid until the pair's result has been fetched */
int mods_pipeline_submit_host(mods_pipeline *p, const float *img_host, long tag);
int mods_pipeline_submit_host_u8(mods_pipeline *p, const unsigned char *img_host, long tag);
int mods_pipeline_next(mods_pipeline *p, mods_pair_result *res, long *tag);
/* the same, and the verified matches of the pair (rows x1 y1 x2 y2, the order of mods_match_pair_dev's matches_out): the first
 * min(res->n_inliers, max_matches) rows are written */
int mods_pipeline_next_matches(mods_pipeline *p, mods_pair_result *res, long *tag, double *matches_out, int max_matches);
void mods_pipeline_destroy(mods_pipeline *p);

#ifdef __cplusplus
}
#endif
#endif /* MODS_HIP_H */
