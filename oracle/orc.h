// ORACLE — TEST INFRASTRUCTURE ONLY (see detmath.h).  CPU restatement of the reference's
// detect -> describe -> match -> verify path.  Every function cites the reference
// file:line it follows (paths relative to the reference root).
//
// Parity status: the degensac part (RANSAC) is pinned against the reference's own C code
// compiled from /root/reference (oracle/_ref, see Makefile).  The OpenCV-dependent parts
// (GaussianBlur, resize, FLANN linear search) are restated from OpenCV's documented
// algorithms with a fixed summation order; OpenCV is not in the image and the reference
// pins no version => for those stages: "parity unpinned" (see DESIGN.md).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

struct Img {
  int w = 0, h = 0;
  std::vector<float> d;
  Img() {}
  Img(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, 0.f) {}
  float *row(int y) { return d.data() + (size_t)y * w; }
  const float *row(int y) const { return d.data() + (size_t)y * w; }
  float &at(int y, int x) { return d[(size_t)y * w + x]; }
  float at(int y, int x) const { return d[(size_t)y * w + x]; }
};

// experiment switches of the unpinned OpenCV arithmetic (the defaults are the contract; see image_ops.cpp)
struct Variant { int kernel = 0, row_fma = 1, col_fma = 1, small_row = 1, resize_tail = 0, libm = 0; };
extern Variant g_variant;

// Threads used inside one oracle call (OpenMP over image rows, keypoints, queries: the "row-parallel" CPU baseline of
// bench.py).  Default 1; every parallel loop writes per-item slots and keeps the serial order, so results do not depend on it.
extern int g_threads;

// ---- image primitives (image_ops.cpp) -------------------------------------------------
int gauss_ksize(float sigma);                                  // detectors/helpers.cpp:720-721
std::vector<float> gauss_kernel(int n, double sigma);          // OpenCV getGaussianKernel, CV_32F
void gauss_blur(const Img &src, Img &dst, float sigma);        // detectors/helpers.cpp:717-731
void resize_half(const Img &src, Img &dst);                    // pyramid.cpp:476 (cv::resize 0.5)
void hessian_response(const Img &in, Img &out, float norm);    // pyramid.cpp:196-254
void dog_response(const Img &in, Img &out, float norm, bool ii);   // pyramid.cpp:165-194
void harris_response(const Img &in, Img &out, float norm);     // pyramid.cpp:256-278
bool interpolate_check_borders(int img_w, int img_h, float ofsx, float ofsy, float a11,
                               float a12, float a21, float a22, int res_w, int res_h);
bool interpolate(const Img &im, float ofsx, float ofsy, float a11, float a12, float a21,
                 float a22, Img &res);                         // detectors/helpers.cpp:551-626
void compute_gauss_mask(Img &mask);                            // detectors/helpers.cpp:411-440
void compute_circular_gauss_mask(Img &mask, float sigma);      // detectors/helpers.cpp:442-461
void compute_gradient(const Img &img, Img &gx, Img &gy);       // detectors/helpers.cpp:779-797
void solve_linear_3x3(float *A, float *b);                     // detectors/helpers.cpp:309-368
void inv_sqrt(float &a, float &b, float &c, float &l1, float &l2);   // helpers.cpp:463-502
bool get_eigenvalues(float a, float b, float c, float d, float &l1, float &l2);  // :504-515
void photometrically_normalize(Img &image, const Img &mask, float &sum, float &var);  // :666-715

// ---- view synthesis (synth_view.cpp) ---------------------------------------------------
struct ViewGeom {               // what GenerateSynthImageCorr derives before touching pixels, synth-detection.cpp:336-469
  bool identity;
  double rotation, tilt, zoom;  // SynthImage fields (degrees, |tilt|, zoom)
  double H[9];                  // original -> view
  int w_rot, h_rot, w_new, h_new;
  double warpRot[6], warpTilt[6];
  double sigma_x, sigma_y;
  int ksize_x, ksize_y;
};
bool view_geometry(int w, int h, double tilt, double phi, double zoom, double InitSigma, ViewGeom *g);
void warp_affine(const Img &src, const double M[6], int dw, int dh, float cval, Img &dst);   // cv::warpAffine LINEAR/CONSTANT
void gauss_blur_xy(const Img &src, Img &dst, int kx, int ky, double sx, double sy);          // cv::GaussianBlur, REFLECT_101
void generate_synth_view(const Img &in, double tilt, double phi, double zoom, double InitSigma, int doBlur, Img &out,
                         ViewGeom *g);                                                      // synth-detection.cpp:324-518

// ---- detector (detect.cpp) ------------------------------------------------------------
struct HessAffParams {          // PyramidParams + AffineShapeParams, detectors/structures.hpp:114-150,
  int numberOfScales = 3;       // affine.h:26-68; defaults = build/config_affori_classic.ini
  float initialSigma = 1.6f;
  float threshold = 5.33f;
  double edgeEigenValueRatio = 10.0;
  int border = 5;
  int maxIterations = 16;
  float convergenceThreshold = 0.05f;
  int smmWindowSize = 19;
  int doBaumberg = 1;
  int mode = 0;                 // detection_mode_t (structures.hpp:10-14): 0 FIXED_TH, 1 RELATIVE_TH, 2 FIXED_REG_NUMBER,
  float rel_threshold = -1;     //   3 RELATIVE_REG_NUMBER, 4 NOT_LESS_THAN_REGIONS; defaults of PyramidParams (:138-150)
  int reg_number = -1;
  float rel_reg_number = -1;
  int detector_type = 0;        // detector_type (structures.hpp:16-18): 0 DET_HESSIAN, 1 DET_DOG, 2 DET_HARRIS
  int ii_dog = 0;               // iiDoGMode (only the DoG response has such a form, pyramid.cpp:126-161)
  int sample_from_image = 0;    // AffineShapeParams::sampleFromImage (affine.h:47): findAffineShape on the input image, pixel distance 1
  // detector_type 3 = DET_MSER: extrema::ExtremaParams (detectors/mser/extrema/extremaParams.h:56-89, [MSER] of the .ini,
  // io_mods.cpp:101-123); mode / rel_threshold / reg_number / rel_reg_number above are shared
  double mser_max_area = 0.01;
  double mser_min_margin = 10;
  int mser_min_size = 30;
  // AffineShapeParams::affBmbrgMethod / affMeasRegion (affine.h:21-24, 49-50, 63-64): 0 = SMM, 1 = Hessian form of the iteration
  int aff_bmbrg_method = 0;
  float aff_meas_region = 0.5f;
};

struct Candidate {              // one accepted pyramid keypoint before affine adaptation
  int octave, level, r0, c0;    // NMS position that spawned it (processing order key)
  int r, c;                     // final integer position after localisation
  float x, y, s, pixelDistance, response;
  int type;
};

struct AffKey {                 // AffineKeypoint subset, detectors/structures.hpp:185-195
  double x, y, s, a11, a12, a21, a22, response;
  int sub_type;
  int octave, level, r0, c0;    // provenance (not in the reference struct; for parity tests)
};

struct Pyramid {
  struct Oct { int w, h; float pixelDistance; std::vector<Img> blur, resp; std::vector<float> sigma; };
  std::vector<Oct> oct;
};

void build_pyramid(const Img &image, const HessAffParams &p, Pyramid &pyr);
void find_candidates(const Pyramid &pyr, const HessAffParams &p, std::vector<Candidate> &out,
                     std::vector<int> *nms_raw = nullptr);
// cv::SVD::compute of a 2x2 fp32 matrix as OpenCV's Jacobi routine does it (detect.cpp; parity unpinned): A = U diag(d) Vt
void svd2x2_f32(const float A[4], float d[2], float U[4], float Vt[4], bool *degenerate);
bool find_affine_shape(const Img &blur, float x, float y, float s, float pixelDistance,
                       const HessAffParams &p, const Img &mask, float a[4], int *iters);
// Full DetectAffineKeypoints + DetectAffineRegions (scale-space-detector.cpp:13-32,
// synth-detection.hpp:79-112): sorted by |response| desc, s*=sqrt|det|, A rectified.
void detect_hessian_affine(const Img &image, const HessAffParams &p, std::vector<AffKey> &out);
void rectify_transformation(double &a11, double &a12, double &a21, double &a22);   // synth-detection.cpp rectifyTransformation

// ---- MSER (mser.cpp; detectors/mser/extrema/) -------------------------------------------
struct MserParams { int min_size = 30; double max_area = 0.01, min_margin = 10; bool relative = false; };   // the fields the growth reads
struct MserRun { int line, col1, col2; };                       // RLEItem, libExtrema.h:35-39
struct MserRegion {                                             // RLERegion, libExtrema.h:42-81
  int thresh, margin, min_int, max_int, area, border, seed_x, seed_y;
  std::vector<MserRun> rle;
  double cx, cy, sxx, sxy, syy;
};
void mser_rle_to_ellipse(const std::vector<MserRun> &rle, double &barX, double &barY, double &sumX2, double &sumXY, double &sumY2);
void mser_regions(const unsigned char *img8, int w, int h, const MserParams &par, bool inverted, std::vector<MserRegion> &out);
// DetectMSERs + DetectAffineRegions (extrema.cpp:196-295, synth-detection.hpp:79-112); MSER+ first, then MSER-
void detect_mser(const Img &image, const HessAffParams &p, double tilt, double zoom, std::vector<AffKey> &out,
                 std::vector<MserRegion> *regions_out = nullptr);

// ---- orientation + descriptor (describe.cpp) -------------------------------------------
struct Region {                 // AffineRegion subset: det_kp == reproj_kp for the identity view
  double x, y, s, a11, a12, a21, a22, response;
  int sub_type;
  int id, parent;
  uint8_t desc[128];
};
// ReprojectRegionsAndRemoveTouchBoundary(dontRemove=true) for H=I: keep centres inside.
void filter_centres_inside(std::vector<Region> &r, int w, int h);           // synth-detection.cpp:151-190
int detect_orientation(const std::vector<Region> &in, std::vector<Region> &out, const Img &img,
                       double mrSize, int patchSize, int maxAngles, double th, bool half = false, bool add_upright = false);   // :1039-1149 (half: doHalfSIFT)
void filter_touch_boundary(std::vector<Region> &r, int w, int h);           // ReprojectRegions :631-706
void affnet_apply(std::vector<Region> &r, const float *a3, int w, int h, double mrSize);   // imagerepresentation.cpp:798-842
void orinet_apply(std::vector<Region> &r, const float *yx);                               // imagerepresentation.cpp:877-899
// the same two steps for a synthesised view (H: original -> view); det = view frame, rep = original frame
void invert3(const double *S, double *t);                                   // cv::invert 3x3
bool h_is_eye(const double *H);                                             // synth-detection.cpp:144-149
void filter_centres_inside_view(std::vector<Region> &det, const double *H, int orig_w, int orig_h);
void reproject_regions_view(std::vector<Region> &det, std::vector<Region> &rep, const double *H, int orig_w, int orig_h);
void describe_rootsift(std::vector<Region> &r, const Img &img, double mrSize, int patchSize,
                       bool photoNorm, bool half = false, bool fast = false);                  // synth-detection.hpp:170-263 (half: HalfRootSIFT, 64 values)
void sift_patch_to_desc(const Img &patch41, uint8_t out[128], bool rootsift, double maxBinValue = 0.2, bool half = false);   // matching/siftdesc.cpp
void extract_desc_patch(const Region &r, const Img &img, double mrSize, int patchSize, bool photoNorm,
                        Img &patch, bool column_rule = false, bool fast = false);
bool dominant_angle(const Img &patch, double th, float *angle, bool half = false);   // :836-929 (maxAngles=1)
int dominant_angles(const Img &patch, double th, int maxAngles, std::vector<float> &angles, bool half = false);   // any maxAngles (-1: all peaks)

// ---- matching (match.cpp) ------------------------------------------------------------------
struct Tentative {
  int q, t;            // first (query idx in list1), second (train idx in list2)
  int t_bad, t_2nd;    // secondbad, secondbadby2ndcl
  float d1, d2, d2nd;  // squared L2 (integer valued)
  double ratio;        // sqrt(d1/d2)
};
int match_fginn(const std::vector<Region> &q, const std::vector<Region> &t, std::vector<Tentative> &out,
                double ratio, double contradDist, int nn);                   // matching.cpp:356-460
int match_distance(const std::vector<Region> &q, const std::vector<Region> &t, std::vector<Tentative> &out, double matchDistanceThreshold);   // matching.cpp:572-633
void duplicate_filter(std::vector<Tentative> &tc, const std::vector<Region> &q,
                      const std::vector<Region> &t, double r, int mode);      // matching.cpp:2615-2679

}  // namespace orc
