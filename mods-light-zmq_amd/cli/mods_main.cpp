// mods - command-line front end of the MI355X matcher, argument- and file-compatible with the reference's
// `mods` binary for the part of the system that is in scope (HessianAffine + RootSIFT steps, LO-RANSAC
// homography or DEGENSAC epipolar verification).
//
// Reference behaviour:
//   argv layout        getCLIparam, io_mods.cpp:558-603 (Tmin = 9, io_mods.h:13):
//       mods img1 img2 out1 out2 k1 k2 matchings log [logOnly] [ver_type] [H/F file] [config.ini] [iters.ini]
//            [read_pre_extracted] [match_one_to_many]
//       ver_type 0 = LO-RANSAC homography, 1 = ground-truth homography (the H file is read), 2 = LO-RANSAC epipolar (3 = ORSA: not built)
//   configuration      the [HessianAffine], [DominantOrientation], [SIFTDescriptor], [Matching], [DuplicateFiltering],
//                      [RANSAC], [TextOutput], [Computing] keys of io_mods.cpp:160-207, 423-455, 605-740 and the
//                      [Iterations] / [HessianAffine<i>] sections of the iterations file (:457-492)
//   step loop          mods.cpp:202-383 (mods_match_ladder_dev)
//   outputs            matchings "x1 y1 x2 y2" (matching.cpp:2596-2613), log line (io_mods.cpp:10-66),
//                      keypoint files (imagerepresentation.cpp:198-204, 1219-1255), H/F file (matching.cpp:2681-2686),
//                      time.log (io_mods.cpp:67-99, mods.cpp:528-540); exit code 0 / 1
// Detectors: HessianAffine, DoG, HarrisAffine (one scale-space detector, three responses) and MSER; ORB / FAST / ... steps of an
// iterations file are skipped with a warning, match images (out1/out2) are not drawn.  Pre-extracted input
// (read_pre_extracted = 1) is read from .npz or text keypoint files.  The vector matcher is always the exact (linear) search;
// external (ZMQ) descriptor / AffNet / OriNet daemons are used when the configuration asks for them.
#include "../../include/mods_hip.h"
#include "../../include/mods_zmq.h"
#include "image_io.hpp"
#include "npz_io.hpp"
#include "ini_reader.hpp"
#include <chrono>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

using modscli::GreyImage;
using modscli::IniReader;

namespace {

const int Tmin = 9;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool ends_with(const std::string &s, const std::string &suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

struct Config {
  mods_pair_params pair;
  // the scale-space detectors that have views in some step, sorted by name (the reference's maps are keyed by name), their
  // parameter sets, and steps[step * n_det + d] = detector d's section of that step (n_tilts = -1: none)
  std::vector<std::string> det_names;
  std::vector<mods_hessaff_params> det_params;
  std::vector<mods_ladder_step> steps;
  // grouped matching ([Matching<i>] GroupDetectors / GroupDescriptors, correspondencebank.cpp:245-285): one entry per step when
  // any step names a group; group_pos = place of the bank's "Group" entry among the detector names
  std::vector<mods_ladder_group> groups;
  int group_pos = 0;
  int max_steps = 4, min_matches = 15;
  int load_color = 1;
  int verbose = 0, time_log = 1, write_keypoints = 1, write_matches = 1, output_h = 0;
  // [zmqDescriptor] (io_mods.cpp:395-407): used when a step asks for the "ZMQ" descriptor instead of RootSIFT
  bool use_zmq = false;
  std::string zmq_port = "tcp://localhost:5555";
  double zmq_mr = 3.0 * 1.7320508075688772;
  int zmq_ps = 32;
  // [AffineAdaptation] useZMQ + [AffNet], [DominantOrientation] useZMQ + [OriNet] (io_mods.cpp:124-134, 523-524, 727-736)
  bool aff_zmq = false, ori_zmq = false;
  std::string aff_port = "tcp://localhost:5556", ori_port = "tcp://localhost:5557";
  double aff_mr = 3.0 * 1.7320508075688772, ori_mr = 3.0 * 1.7320508075688772;
  int aff_ps = 32, ori_ps = 32;
};

int read_config(const std::string &config_fn, const std::string &iters_fn, int ver_type, Config *cfg) {
  IniReader ini(config_fn);
  if (ini.ParseError() < 0) { std::cerr << "Can't load " << config_fn << std::endl; return 1; }
  IniReader it(iters_fn);
  if (it.ParseError() < 0) { std::cerr << "Can't load  " << iters_fn << std::endl; return 1; }
  mods_pair_params &p = cfg->pair;
  cfg->verbose = (int)ini.GetInteger("TextOutput", "verbose", 0);     // first: the notes below depend on it
  // [HessianAffine] / [DoG] / [HarrisAffine], io_mods.cpp:160-207, 208-244, 260-296: one key set, DetectorType set by the
  // section; defaults = PyramidParams / AffineShapeParams constructors
  auto read_det = [&](const char *sec, int type) {
    mods_hessaff_params d;
    memset(&d, 0, sizeof(d));
    d.detectorType = type;
    d.threshold = (float)ini.GetDouble(sec, "threshold", 16.0 / 3.0);
    d.border = (int)ini.GetInteger(sec, "border", 5);
    d.numberOfScales = (int)ini.GetInteger(sec, "numberOfScales", 3);
    d.initialSigma = (float)ini.GetDouble(sec, "initialSigma", 1.6);
    d.edgeEigenValueRatio = (float)ini.GetDouble(sec, "edgeEigenValueRatio", 10.0);
    d.maxIterations = (int)ini.GetInteger(sec, "max_iter", 16);
    d.smmWindowSize = (int)ini.GetInteger(sec, "smmWindowSize", 19);
    d.convergenceThreshold = (float)ini.GetDouble(sec, "convergenceThreshold", 0.05);
    d.doBaumberg = (int)ini.GetInteger(sec, "doBaumberg", 1);
    d.iiDoGMode = ini.GetBoolean(sec, "iiDoGMode", false) ? 1 : 0;
    d.sampleFromImage = ini.GetBoolean(sec, "sampleFromImage", false) ? 1 : 0;     // io_mods.cpp:184, 222, 275
    // patch_size (io_mods.cpp:187): the side of the normalised patch that normalizeAffine renders for a callback which only
    // records the geometry (scale-space-detector.hpp:56-90): it never reaches a keypoint, here or in the reference
    if (ini.GetInteger(sec, "patch_size", 41) != 41 && cfg->verbose)
      std::cerr << "Note: [" << sec << "] patch_size has no effect on the keypoints (the reference renders that patch and drops it)" << std::endl;
    // keypoint selection, io_mods.cpp:170-173, 194-205 (defaults: PyramidParams, structures.hpp:138-150)
    d.relativeThreshold = (float)ini.GetDouble(sec, "relativeThreshold", -1.0);
    d.relativeRegionsNumber = (float)ini.GetDouble(sec, "relativeRegionsNumber", -1.0);
    d.regionsNumber = (int)ini.GetInteger(sec, "regionsNumber", -1);
    const std::string mode = ini.GetStringVector(sec, "mode")[0];
    d.mode = mode == "RelativeTh" ? MODS_DET_RELATIVE_TH : mode == "FixedRegNumber" ? MODS_DET_FIXED_REG_NUMBER
           : mode == "NotLessThanRegions" ? MODS_DET_NOT_LESS_THAN_REGIONS : mode == "RelativeRegNumber" ? MODS_DET_RELATIVE_REG_NUMBER
           : MODS_DET_FIXED_TH;
    return d;
  };
  // [MSER], GetMSERPars (io_mods.cpp:101-123); defaults = extrema::ExtremaParams (detectors/mser/extrema/extremaParams.h:72-88)
  auto read_mser = [&]() {
    mods_hessaff_params d;
    memset(&d, 0, sizeof(d));
    d.detectorType = MODS_DET_MSER;
    d.relativeThreshold = (float)ini.GetDouble("MSER", "relativeThreshold", -1.0);
    d.relativeRegionsNumber = (float)ini.GetDouble("MSER", "relativeRegionsNumber", -1.0);
    d.regionsNumber = (int)ini.GetInteger("MSER", "regionsNumber", -1);
    d.mserMaxArea = ini.GetDouble("MSER", "max_area", 0.01);
    d.mserMinSize = (int)ini.GetInteger("MSER", "min_size", 30);
    d.mserMinMargin = (double)ini.GetInteger("MSER", "min_margin", 10);     // read as an integer (io_mods.cpp:108)
    const std::string mode = ini.GetStringVector("MSER", "mode")[0];
    d.mode = mode == "RelativeTh" ? MODS_DET_RELATIVE_TH : mode == "FixedRegNumber" ? MODS_DET_FIXED_REG_NUMBER
           : mode == "NotLessThanRegions" ? MODS_DET_NOT_LESS_THAN_REGIONS : mode == "RelativeRegNumber" ? MODS_DET_RELATIVE_REG_NUMBER
           : MODS_DET_FIXED_TH;
    return d;
  };
  p.det = read_det("HessianAffine", MODS_DET_HESSIAN);
  // only this section reads the key (io_mods.cpp:193); AffineBaumbergMethod, affine.h:21-24
  p.det.affBmbrgMethod = (int)ini.GetInteger("HessianAffine", "affBmbrgMethod", 0);
  if (p.det.affBmbrgMethod != 0 && p.det.affBmbrgMethod != 1) { std::cerr << "[HessianAffine] affBmbrgMethod must be 0 (SMM) or 1 (Hessian)" << std::endl; return 1; }
  // [DominantOrientation] :731-740 and [SIFTDescriptor] :423-436
  mods_describe_params &q = p.desc;
  q.ori_mrSize = ini.GetDouble("DominantOrientation", "mrSize", 3.0 * std::sqrt(3.0));
  q.ori_patchSize = (int)ini.GetInteger("DominantOrientation", "patchSize", 32);
  q.ori_maxAngles = (int)ini.GetInteger("DominantOrientation", "maxAngles", 1);
  q.ori_threshold = (double)(float)ini.GetDouble("DominantOrientation", "threshold", 0.8);
  q.addUpRight = ini.GetBoolean("DominantOrientation", "addUpRight", false) ? 1 : 0;      // io_mods.cpp:732
  // (halfSIFTMode and addMirrored of the same section are parsed by the reference and used nowhere: io_mods.cpp:733-734)
  q.desc_mrSize = ini.GetDouble("SIFTDescriptor", "mrSize", 3.0 * std::sqrt(3.0));
  q.desc_patchSize = (int)ini.GetInteger("SIFTDescriptor", "patchSize", 41);
  q.photoNorm = ini.GetBoolean("SIFTDescriptor", "photoNorm", true) ? 1 : 0;
  q.rootSift = 1;
  q.maxBinValue = ini.GetDouble("SIFTDescriptor", "maxBinValue", 0.2);
  q.fastExtraction = ini.GetBoolean("SIFTDescriptor", "FastPatchExtraction", false) ? 1 : 0;
  if (ini.GetInteger("SIFTDescriptor", "spatialBins", 4) != 4 || ini.GetInteger("SIFTDescriptor", "orientationBins", 8) != 8) {
    std::cerr << "Only 4x4x8 SIFT is supported" << std::endl;
    return 1;
  }
  // [Matching]
  p.fginn_ratio = 0.8;
  p.contradDist = ini.GetDouble("Matching", "contradDist", 10.0);
  p.nn = 50;
  const std::string vm = ini.GetString("Matching", "vector_matcher", "");
  if (!vm.empty() && vm != "linear") std::cerr << "Note: vector_matcher=" << vm << ": this build always searches exactly (the linear index), the approximate indices of FLANN are not reproduced" << std::endl;
  // [DuplicateFiltering] :665-679
  p.dup_dist = ini.GetDouble("DuplicateFiltering", "duplicateDist", 3.0);
  p.dup_before_ransac = (int)ini.GetDouble("DuplicateFiltering", "doBeforeRANSAC", 1);
  const std::string fm = ini.GetString("DuplicateFiltering", "whichCorrespondenceRemains", "random");
  p.dup_mode = fm == "bestFGINN" ? 1 : fm == "bestDistance" ? 2 : fm == "biggerRegion" ? 3 : 0;
  // [RANSAC] :437-455
  mods_ransac_params &r = p.ransac;
  r.err_threshold = ini.GetDouble("RANSAC", "err_threshold", 2.0);
  r.confidence = ini.GetDouble("RANSAC", "confidence", 0.99);
  r.max_samples = (int)ini.GetInteger("RANSAC", "max_samples", 100000);
  r.localOptimization = (int)ini.GetInteger("RANSAC", "localOptimization", 1);
  r.LAFCoef = (double)ini.GetInteger("RANSAC", "LAFcoef", ini.GetInteger("Matching", "LAFcoef", 0));
  r.HLAFCoef = (double)ini.GetInteger("RANSAC", "HLAFcoef", 10);
  r.doSymmCheck = (int)ini.GetInteger("RANSAC", "doSymmCheck", 0);
  const std::string et = ini.GetStringVector("RANSAC", "ErrorType")[0];
  r.errorType = et == "Sampson" ? 0 : et == "SymmMax" ? 1 : 2;
  r.useF = ver_type == 2 ? 1 : 0;
  // GR_TRUTH (mods.cpp:85-87, 290-320, 381-383; io_mods.cpp:340-341): with doBothRANSACgroundTruth the verified list is the
  // LORANSAC inliers that the ground truth confirms, RANSACforStopping stops the step loop on the RANSAC inlier count
  r.groundTruth = ver_type == 1 ? (ini.GetInteger("Matching", "doBothRANSACgroundTruth", 1) ? 2 : 1) : 0;
  r.ransacForStopping = ini.GetInteger("Matching", "RANSACforStopping", 1) ? 1 : 0;
  // [TextOutput], [Computing]
  cfg->time_log = (int)ini.GetInteger("TextOutput", "timeLog", 0);
  cfg->write_keypoints = (int)ini.GetInteger("TextOutput", "writeKeypoints", 1);
  cfg->write_matches = (int)ini.GetInteger("TextOutput", "writeMatches", 1);
  cfg->output_h = (int)ini.GetInteger("TextOutput", "outputEstimatedHorF", 0);
  if (ini.GetInteger("TextOutput", "outputAllTentatives", 0)) std::cerr << "Warning: outputAllTentatives is not supported, only verified matches are written" << std::endl;
  cfg->load_color = (int)ini.GetInteger("Computing", "LoadColor", 1);
  if (ini.Has("zmqDescriptor", "port")) cfg->zmq_port = ini.GetString("zmqDescriptor", "port", "");
  cfg->zmq_mr = ini.GetDouble("zmqDescriptor", "mrSize", cfg->zmq_mr);
  cfg->zmq_ps = (int)ini.GetInteger("zmqDescriptor", "patchSize", cfg->zmq_ps);
  cfg->aff_zmq = ini.GetBoolean("AffineAdaptation", "useZMQ", false);
  if (ini.Has("AffNet", "port")) cfg->aff_port = ini.GetString("AffNet", "port", "");
  cfg->aff_mr = ini.GetDouble("AffNet", "mrSize", cfg->aff_mr);
  cfg->aff_ps = (int)ini.GetInteger("AffNet", "patchSize", cfg->aff_ps);
  cfg->ori_zmq = ini.GetBoolean("DominantOrientation", "useZMQ", false);
  if (ini.Has("OriNet", "port")) cfg->ori_port = ini.GetString("OriNet", "port", "");
  cfg->ori_mr = ini.GetDouble("OriNet", "mrSize", cfg->ori_mr);
  cfg->ori_ps = (int)ini.GetInteger("OriNet", "patchSize", cfg->ori_ps);
  // iterations file :457-492
  cfg->max_steps = (int)it.GetInteger("Iterations", "Steps", 4);
  cfg->min_matches = (int)it.GetInteger("Iterations", "minMatches", 15);
  static const char *other_detectors[] = {"ORB", "FAST", "ReadAffs", "STAR", "BRISK", "SURF", "SIFT", "TILDE", "FOCI"};
  // the detectors of this build, in name order (the order of the reference's region / correspondence maps)
  static const struct { const char *name; int type; } ss_detectors[] = {{"DoG", MODS_DET_DOG}, {"HarrisAffine", MODS_DET_HARRIS}, {"HessianAffine", MODS_DET_HESSIAN},
                                                                        {"MSER", MODS_DET_MSER}};
  const int kDets = 4;
  std::vector<mods_ladder_step> all[4];
  for (int i = 0; i < cfg->max_steps; i++) {
    for (const char *od : other_detectors)
      if (it.Has(od + std::to_string(i), "TiltSet") || it.Has(od + std::to_string(i), "ScaleSet"))
        std::cerr << "Warning: step " << i << ": detector " << od << " is outside this build, its views are skipped" << std::endl;
    // [Matching<i>] SeparateDetectors (io_mods.cpp:320): the detectors whose lists MatchImgReps searches in this step.  The
    // reference matches nothing when the key is absent; an absent key means "every detector of the step" here.
    const std::string msec_det = "Matching" + std::to_string(i);
    std::vector<std::string> sep_det;
    const bool have_sep_det = it.Has(msec_det, "SeparateDetectors");
    if (have_sep_det)
      for (std::string name : it.GetStringVector(msec_det, "SeparateDetectors")) {
        name.erase(0, name.find_first_not_of(" \t"));
        name.erase(name.find_last_not_of(" \t") + 1);
        sep_det.push_back(name);
      }
  for (int di = 0; di < kDets; di++) {
    const std::string det_name = ss_detectors[di].name;
    const std::string sec = det_name + std::to_string(i);
    mods_ladder_step st;
    memset(&st, 0, sizeof(st));
    st.phi = it.GetDouble(sec, "Phi", 360);
    st.initSigma = it.GetDouble(sec, "initSigma", 0.5);
    st.doBlur = 1;
    // DSPLevels / minSigma / maxSigma (io_mods.cpp:480-482): SetVSPars stores them in ViewSynthParameters
    // (synth-detection.cpp:244-289) and nothing in the reference reads them again
    if (cfg->verbose && (it.Has(sec, "DSPLevels") || it.Has(sec, "minSigma") || it.Has(sec, "maxSigma")))
      std::cerr << "Note: [" << sec << "] DSPLevels / minSigma / maxSigma are stored and never used by the reference; no effect" << std::endl;
    st.fginn_ratio = 0.0;
    if (it.Has(sec, "TiltSet") && it.Has(sec, "ScaleSet")) {
      const std::vector<double> tilts = it.GetDoubleVector(sec, "TiltSet"), scales = it.GetDoubleVector(sec, "ScaleSet");
      if (tilts.size() > 8 || scales.size() > 8) { std::cerr << sec << ": at most 8 tilts and 8 scales per step" << std::endl; return 1; }
      st.n_tilts = (int)tilts.size(); st.n_scales = (int)scales.size();
      for (size_t k = 0; k < tilts.size(); k++) st.tilt_set[k] = tilts[k];
      for (size_t k = 0; k < scales.size(); k++) st.scale_set[k] = scales[k];
      const std::vector<std::string> descs = it.GetStringVector(sec, "Descriptors");
      const std::vector<double> fginn = it.GetDoubleVector(sec, "FGINNThreshold");
      // DistanceThreshold, same order: > 0 runs MatchFLANNDistance on that descriptor's lists (correspondencebank.cpp:320-334)
      const std::vector<double> dthr = it.Has(sec, "DistanceThreshold") ? it.GetDoubleVector(sec, "DistanceThreshold") : std::vector<double>();
      bool has_root = false, has_zmq = false;
      double zmq_ratio = 0, zmq_dist = 0;
      for (size_t k = 0; k < descs.size(); k++) {
        std::string name = descs[k];
        name.erase(0, name.find_first_not_of(" \t"));
        name.erase(name.find_last_not_of(" \t") + 1);
        // a "Half*" name anywhere in the list switches the step's orientation estimate to doHalfSIFT mode for every descriptor
        // (imagerepresentation.cpp:725-731)
        if (name.find("Half") != std::string::npos) st.half_orientation = 1;
        if (name == "RootSIFT") { has_root = true; st.fginn_ratio = k < fginn.size() ? fginn[k] : 0.0; st.dist_threshold = k < dthr.size() ? dthr[k] : 0.0; }
        else if (name == "ZMQ") { has_zmq = true; zmq_ratio = k < fginn.size() ? fginn[k] : 0.0; zmq_dist = k < dthr.size() ? dthr[k] : 0.0; }
        else if (name == "HalfRootSIFT") {
          st.dist_threshold_half = k < dthr.size() ? dthr[k] : 0.0;
          // the reference indexes FGINNThreshold by the descriptor's position without a bounds check (synth-detection.cpp:221):
          // a missing entry is read past the end of the vector; it counts as 0 = "not matched" here
          if (k < fginn.size()) st.fginn_ratio_half = fginn[k];
          else std::cerr << "Warning: " << sec << ": no FGINNThreshold for HalfRootSIFT: its lists are not matched" << std::endl;
        }
        else if (!name.empty()) std::cerr << "Warning: " << sec << ": descriptor " << name << " is outside this build" << std::endl;
      }
      if (st.fginn_ratio_half != 0 && !(st.fginn_ratio_half > 0 && st.fginn_ratio_half < 1)) { std::cerr << sec << ": FGINNThreshold of HalfRootSIFT must lie in (0, 1)" << std::endl; return 1; }
      // [Matching<i>] SeparateDescriptors: only the listed descriptors are matched (correspondencebank.cpp:288-299)
      const std::string msec = "Matching" + std::to_string(i);
      if (it.Has(msec, "SeparateDescriptors")) {
        bool root_listed = false, half_listed = false;
        for (std::string name : it.GetStringVector(msec, "SeparateDescriptors")) {
          name.erase(0, name.find_first_not_of(" \t"));
          name.erase(name.find_last_not_of(" \t") + 1);
          root_listed = root_listed || name == "RootSIFT" || name == "ZMQ";
          half_listed = half_listed || name == "HalfRootSIFT";
        }
        if (!half_listed) { st.fginn_ratio_half = st.fginn_ratio_half > 0 ? -1.0 : 0.0; st.dist_threshold_half = 0.0; }   // its list is left alone
        if (!root_listed && has_root) std::cerr << "Warning: " << msec << ": SeparateDescriptors does not list the step's descriptor; it is matched anyway" << std::endl;
      }
      if (has_zmq && !has_root) {      // the daemon's descriptor takes the place of RootSIFT for the whole run
        cfg->use_zmq = true; has_root = true; st.fginn_ratio = zmq_ratio; st.dist_threshold = zmq_dist;
      } else if (has_zmq) std::cerr << "Warning: " << sec << ": RootSIFT and ZMQ in one step: RootSIFT is used" << std::endl;
      if (!has_root) { std::cerr << "Warning: " << sec << " does not ask for RootSIFT; the step is skipped" << std::endl; st.n_tilts = st.n_scales = -1; }
      else if (!(st.fginn_ratio > 0 && st.fginn_ratio < 1) && !(st.dist_threshold > 0)) { std::cerr << sec << ": FGINNThreshold of RootSIFT must lie in (0, 1)" << std::endl; return 1; }
      if (st.dist_threshold < 0 || st.dist_threshold_half < 0) { std::cerr << sec << ": DistanceThreshold must be >= 0" << std::endl; return 1; }
    } else st.n_tilts = st.n_scales = -1;      // no views of this detector in this step
    if (st.n_tilts >= 0 && have_sep_det && std::find(sep_det.begin(), sep_det.end(), det_name) == sep_det.end() &&
        std::find(sep_det.begin(), sep_det.end(), std::string("All")) == sep_det.end()) {
      if (cfg->verbose) std::cerr << sec << ": not in SeparateDetectors of [" << msec_det << "]: described, not matched" << std::endl;
      st.fginn_ratio = -1.0;
      if (st.fginn_ratio_half > 0) st.half_orientation = 1;   // still described in doHalfSIFT mode
      st.fginn_ratio_half = -1.0;
      st.dist_threshold = st.dist_threshold_half = 0.0;
    }
    all[di].push_back(st);
  }
  }
  for (int di = 0; di < kDets; di++) {
    bool any = false;
    for (const mods_ladder_step &st : all[di]) any = any || st.n_tilts >= 0;
    if (!any) continue;
    cfg->det_names.push_back(ss_detectors[di].name);
    cfg->det_params.push_back(di == 2 ? p.det : ss_detectors[di].type == MODS_DET_MSER ? read_mser() : read_det(ss_detectors[di].name, ss_detectors[di].type));
  }
  const int n_det = (int)cfg->det_names.size();
  for (int i = 0; i < cfg->max_steps; i++)
    for (int di = 0, d = 0; di < kDets; di++) {
      if (d < n_det && cfg->det_names[d] == ss_detectors[di].name) { cfg->steps.push_back(all[di][i]); d++; }
    }
  // grouped matching: thresholds are the [Matching]-wide ones (io_mods.cpp:448-452), the detector order is the order named
  auto trimmed = [](std::string name) { name.erase(0, name.find_first_not_of(" \t")); name.erase(name.find_last_not_of(" \t") + 1); return name; };
  bool any_group = false;
  std::vector<mods_ladder_group> groups((size_t)cfg->max_steps);
  for (int i = 0; i < cfg->max_steps; i++) {
    mods_ladder_group &g = groups[i];
    memset(&g, 0, sizeof(g));
    g.fginn_ratio = g.fginn_ratio_half = -1.0;
    const std::string msec = "Matching" + std::to_string(i);
    if (!it.Has(msec, "GroupDetectors") || !it.Has(msec, "GroupDescriptors")) continue;
    for (const std::string &raw : it.GetStringVector(msec, "GroupDetectors")) {
      const std::string name = trimmed(raw);
      if (name.empty()) continue;
      const auto at = std::find(cfg->det_names.begin(), cfg->det_names.end(), name);
      if (at == cfg->det_names.end()) { std::cerr << "Warning: [" << msec << "] GroupDetectors: " << name << " has no views in this build's steps, left out of the group" << std::endl; continue; }
      if (g.n_dets < 8) g.dets[g.n_dets++] = (int)(at - cfg->det_names.begin());
    }
    bool any_desc = false;
    for (const std::string &raw : it.GetStringVector(msec, "GroupDescriptors")) {
      const std::string name = trimmed(raw);
      if (name == "RootSIFT" || (name == "ZMQ" && cfg->use_zmq)) {
        g.fginn_ratio = ini.GetDouble("Matching", "matchRatio" + name, 0.0); g.dist_threshold = ini.GetDouble("Matching", "matchDistance" + name, 0.0); any_desc = true;
      } else if (name == "HalfRootSIFT") {
        g.fginn_ratio_half = ini.GetDouble("Matching", "matchRatioHalfRootSIFT", 0.0); g.dist_threshold_half = ini.GetDouble("Matching", "matchDistanceHalfRootSIFT", 0.0); any_desc = true;
      } else if (!name.empty()) std::cerr << "Warning: [" << msec << "] GroupDescriptors: " << name << " is outside this build" << std::endl;
    }
    if (!any_desc || g.n_dets == 0) { g.n_dets = 0; continue; }
    if ((g.fginn_ratio > 0 && !(g.fginn_ratio < 1)) || (g.fginn_ratio_half > 0 && !(g.fginn_ratio_half < 1))) { std::cerr << "[Matching] matchRatio* must lie in (0, 1)" << std::endl; return 1; }
    any_group = true;
  }
  if (any_group) {
    cfg->groups = groups;
    cfg->group_pos = 0;
    for (const std::string &nm : cfg->det_names) if (nm < std::string("Group")) cfg->group_pos++;
  }
  return 0;
}

int usage() {
  std::cerr << " ************************************************************************** " << std::endl
            << " **** mods: two-view matching with view synthesis, MI355X build        **** " << std::endl
            << " ************************************************************************** " << std::endl
            << "Usage: mods img1 img2 out1 out2 k1 k2 matchings log [logOnly=1] [ver_type=0] [H/F file] [config.ini] [iters.ini]" << std::endl
            << "  img1, img2   PNG or binary PGM/PPM" << std::endl
            << "  out1, out2   accepted for compatibility (match images are not drawn)" << std::endl
            << "  k1, k2       keypoint + descriptor files (text)" << std::endl
            << "  matchings    verified correspondences, x1 y1 x2 y2 per line" << std::endl
            << "  log          one line: time matches tentatives inlier% regions1 regions2 steps" << std::endl
            << "  ver_type     0 LO-RANSAC homography, 1 ground-truth homography (read from the H/F file), 2 LO-RANSAC (DEGENSAC) epipolar geometry" << std::endl;
  return 1;
}

bool load_grey(const std::string &fn, int load_color, GreyImage *img) {
  std::string err;
  if (!modscli::read_image(fn, load_color != 0, img, &err)) { std::cerr << err << std::endl; return false; }
  return true;
}

// SaveRegions, imagerepresentation.cpp:1219-1255: the detectors in name order (the region map's key order), one descriptor list each
void write_regions(const std::string &fn, const std::vector<mods_imgrep *> &reps, const std::vector<std::string> &det_names, const char *desc_name) {
  std::ofstream kp(fn);
  if (!kp.is_open()) { std::cerr << "Cannot open file " << fn << " to save keypoints" << std::endl; return; }
  kp << reps.size() << std::endl;
  for (size_t d = 0; d < reps.size(); d++) {
    const int n = mods_imgrep_count(reps[d]);
    std::vector<mods_region> regs((size_t)std::max(n, 1));
    if (n > 0 && mods_imgrep_fetch(reps[d], 0, n, regs.data())) { std::cerr << mods_last_error() << std::endl; return; }
    kp << det_names[d] << " " << 1 << std::endl;
    kp << desc_name << " " << n << std::endl;
    if (n > 0) kp << 128 << std::endl;
    for (int i = 0; i < n; i++) {
      const mods_region &r = regs[i];
      kp << r.x << " " << r.y << " " << r.s << " " << r.a11 << " " << r.a12 << " " << r.a21 << " " << r.a22;
      kp << " " << 128 << " ";
      for (int j = 0; j < 128; j++) kp << (float)r.desc[j] << " ";
      kp << std::endl;
    }
  }
}

// SaveRegionsNPZ, imagerepresentation.cpp:1257-1316: xy, scales, responses, A (reproj_kp, doubles) and descs (uchar)
bool write_regions_npz(const std::string &fn, const std::vector<mods_imgrep *> &reps) {
  int n = 0;
  for (mods_imgrep *rep : reps) n += mods_imgrep_count(rep);
  std::vector<mods_region> regs((size_t)std::max(n, 1));
  for (size_t d = 0, at = 0; d < reps.size(); d++) {      // all detectors, in name order, in one set of arrays (:1257-1316)
    const int nd = mods_imgrep_count(reps[d]);
    if (nd > 0 && mods_imgrep_fetch(reps[d], 0, nd, regs.data() + at)) { std::cerr << mods_last_error() << std::endl; return false; }
    at += (size_t)nd;
  }
  modscli::NpyArray xy, sc, rs, A, ds;
  auto f8 = [&](modscli::NpyArray &a, size_t cols) { a.descr = "<f8"; a.shape = {(size_t)n, cols}; a.data.resize(sizeof(double) * n * cols); };
  f8(xy, 2); f8(sc, 1); f8(rs, 1); f8(A, 4);
  ds.descr = "|u1"; ds.shape = {(size_t)n, 128}; ds.data.resize((size_t)n * 128);
  for (int i = 0; i < n; i++) {
    const mods_region &r = regs[i];
    const double v2[2] = {r.x, r.y}, v4[4] = {r.a11, r.a12, r.a21, r.a22};
    memcpy(&xy.data[sizeof(double) * 2 * i], v2, sizeof(v2));
    memcpy(&sc.data[sizeof(double) * i], &r.s, sizeof(double));
    memcpy(&rs.data[sizeof(double) * i], &r.response, sizeof(double));
    memcpy(&A.data[sizeof(double) * 4 * i], v4, sizeof(v4));
    memcpy(&ds.data[(size_t)128 * i], r.desc, 128);
  }
  std::string err;
  if (!modscli::npz_write(fn, {{"xy", xy}, {"scales", sc}, {"responses", rs}, {"A", A}, {"descs", ds}}, &err)) { std::cerr << err << std::endl; return false; }
  return true;
}

// PreLoadRegionsNPZ, imagerepresentation.cpp:1355-1503: xy, scales, responses, descs and either A, angles (degrees) or neither
// (upright circles); det_kp = reproj_kp
bool read_regions_npz(const std::string &fn, std::vector<mods_region> *out) {
  std::map<std::string, modscli::NpyArray> z;
  std::string err;
  if (!modscli::npz_read(fn, &z, &err)) { std::cerr << err << std::endl; return false; }
  for (const char *key : {"xy", "scales", "responses", "descs"})
    if (!z.count(key)) { std::cerr << fn << ": no array " << key << std::endl; return false; }
  const modscli::NpyArray &xy = z["xy"], &sc = z["scales"], &rs = z["responses"], &ds = z["descs"];
  if (xy.descr != "<f8" || sc.descr != "<f8" || rs.descr != "<f8" || ds.descr != "|u1" || xy.shape.empty() || ds.shape.size() != 2) {
    std::cerr << fn << ": xy / scales / responses must be float64 and descs uint8 (n x dim)" << std::endl; return false;
  }
  const size_t n = xy.shape[0];
  if (ds.shape[1] != 128) { std::cerr << fn << ": descriptors of dimension " << ds.shape[1] << " (this build matches 128-byte descriptors)" << std::endl; return false; }
  if (xy.count() != 2 * n || sc.count() != n || rs.count() != n || ds.shape[0] != n) { std::cerr << fn << ": array lengths differ" << std::endl; return false; }
  const double *pxy = (const double *)xy.data.data(), *psc = (const double *)sc.data.data(), *prs = (const double *)rs.data.data();
  const double *pA = nullptr, *pang = nullptr;
  if (z.count("A")) { if (z["A"].descr != "<f8" || z["A"].count() != 4 * n) { std::cerr << fn << ": bad A" << std::endl; return false; } pA = (const double *)z["A"].data.data(); }
  else if (z.count("angles")) { if (z["angles"].descr != "<f8" || z["angles"].count() != n) { std::cerr << fn << ": bad angles" << std::endl; return false; } pang = (const double *)z["angles"].data.data(); }
  out->assign(n, mods_region());
  for (size_t i = 0; i < n; i++) {
    mods_region &r = (*out)[i];
    memset(&r, 0, sizeof(r));
    r.x = pxy[2 * i]; r.y = pxy[2 * i + 1]; r.s = psc[i]; r.response = prs[i];
    if (pA) { r.a11 = pA[4 * i]; r.a12 = pA[4 * i + 1]; r.a21 = pA[4 * i + 2]; r.a22 = pA[4 * i + 3]; }
    else {
      const double angle = pang ? pang[i] * M_PI / 180.0 : 0.0;
      r.a11 = cos(angle); r.a12 = sin(angle); r.a21 = -sin(angle); r.a22 = cos(angle);
    }
    r.id = (int)i; r.parent = -1;
    memcpy(r.desc, &ds.data[(size_t)128 * i], 128);
  }
  return true;
}

// ImageRepresentation::LoadRegions, imagerepresentation.cpp:1317-1354: the text keypoint file.  Header as SaveRegions writes
// it (number of detectors; per detector: name, number of descriptors; per descriptor: name, number of regions, dimension),
// rows as loadAR reads them (:241-253: id img_id img_reproj_id parent_id, det_kp and reproj_kp as "x y a11 a12 a21 a22
// pyramid_scale octave_number s sub_type", dimension, values).  The reference's own SaveRegions writes a different row
// (saveAR, :198-204: "x y s a11 a12 a21 a22 dimension values", the file this program writes too), which its LoadRegions cannot
// read back; both row forms are accepted here, told apart by the number of values on the first row.  The regions of the first
// 128-value descriptor list of the HessianAffine detector (or of the first detector) are returned.
bool read_regions_txt(const std::string &fn, std::vector<mods_region> *out) {
  std::ifstream f(fn);
  if (!f.is_open()) { std::cerr << "Cannot open file " << fn << " to load keypoints" << std::endl; return false; }
  int n_det = 0;
  if (!(f >> n_det) || n_det < 0 || n_det > 64) { std::cerr << fn << ": not a keypoint file" << std::endl; return false; }
  out->clear();
  bool have = false;
  for (int d = 0; d < n_det; d++) {
    std::string det_name;
    int n_desc = 0;
    if (!(f >> det_name >> n_desc) || n_desc < 0 || n_desc > 64) { std::cerr << fn << ": bad detector header" << std::endl; return false; }
    for (int k = 0; k < n_desc; k++) {
      std::string desc_name;
      long n = 0, dim = 0;
      if (!(f >> desc_name >> n) || n < 0 || n > (1 << 24)) { std::cerr << fn << ": bad descriptor header" << std::endl; return false; }
      if (n > 0 && (!(f >> dim) || dim < 0 || dim > 4096)) { std::cerr << fn << ": bad descriptor dimension" << std::endl; return false; }
      std::string rest;
      std::getline(f, rest);
      const bool take = !have && dim == 128 && (det_name == "HessianAffine" || d == 0) && desc_name != "HalfRootSIFT";
      for (long i = 0; i < n; i++) {
        std::string line;
        if (!std::getline(f, line)) { std::cerr << fn << ": truncated (" << desc_name << ", row " << i << ")" << std::endl; return false; }
        if (!take) continue;
        std::istringstream ls(line);
        std::vector<double> v;
        double t;
        while (ls >> t) v.push_back(t);
        mods_region r;
        memset(&r, 0, sizeof(r));
        size_t dpos;
        if (v.size() == (size_t)(8 + dim)) {            // saveAR row
          r.x = v[0]; r.y = v[1]; r.s = v[2]; r.a11 = v[3]; r.a12 = v[4]; r.a21 = v[5]; r.a22 = v[6];
          if ((long)v[7] != dim) { std::cerr << fn << ": row " << i << ": dimension mismatch" << std::endl; return false; }
          dpos = 8; r.id = (int)i; r.parent = -1;
        } else if (v.size() == (size_t)(25 + dim)) {    // loadAR row: reproj_kp is what matching and verification use
          r.id = (int)v[0]; r.parent = (int)v[3];
          const double *kp = &v[14];
          r.x = kp[0]; r.y = kp[1]; r.a11 = kp[2]; r.a12 = kp[3]; r.a21 = kp[4]; r.a22 = kp[5]; r.s = kp[8]; r.sub_type = (int)kp[9];
          if ((long)v[24] != dim) { std::cerr << fn << ": row " << i << ": dimension mismatch" << std::endl; return false; }
          dpos = 25;
        } else { std::cerr << fn << ": row " << i << " of " << desc_name << " has " << v.size() << " values" << std::endl; return false; }
        for (long q = 0; q < 128; q++) {
          const double dv = v[dpos + q];
          r.desc[q] = (uint8_t)(dv <= 0 ? 0 : (dv >= 255 ? 255 : (int)(dv + 0.5)));
        }
        out->push_back(r);
      }
      if (take) have = true;
    }
  }
  if (!have) { std::cerr << fn << ": no 128-value descriptor list" << std::endl; return false; }
  return true;
}

bool read_regions_any(const std::string &fn, std::vector<mods_region> *out) {   // mods.cpp:216-229: .npz or the text format
  return ends_with(fn, ".npz") ? read_regions_npz(fn, out) : read_regions_txt(fn, out);
}

}  // namespace

int main(int argc, char **argv) {
  if (argc == 4 && std::string(argv[1]) == "--npz-echo") {   // read an .npz keypoint file and write it back (format check, no GPU involved)
    std::map<std::string, modscli::NpyArray> z;
    std::string err;
    if (!modscli::npz_read(argv[2], &z, &err)) { std::cerr << err << std::endl; return 1; }
    std::vector<std::pair<std::string, modscli::NpyArray>> members(z.begin(), z.end());
    if (!modscli::npz_write(argv[3], members, &err)) { std::cerr << err << std::endl; return 1; }
    return 0;
  }
  if (argc < Tmin) return usage();
  const double c_start = now_s();
  const std::string img1_fn = argv[1], img2_fn = argv[2], k1_fn = argv[5], k2_fn = argv[6], match_fn = argv[7], log_fn = argv[8];
  int log_only = 1, ver_type = 0;
  std::string config_fn = "config_iter.ini", iters_fn = "iters.ini";
  if (argc >= Tmin + 1) log_only = atoi(argv[Tmin]);
  if (argc >= Tmin + 2) {
    ver_type = atoi(argv[Tmin + 1]);
    if (ver_type != 0 && ver_type != 1 && ver_type != 2) {
      std::cerr << ver_type << " is wrong correspondence verification type." << std::endl
                << "Try 0 for LO-RANSAC(homography), 1 for ground truth matrix or 2 for LO-RANSAC(epipolar) (3, ORSA, is not part of this build)" << std::endl;
      return 1;
    }
    if (ver_type == 1 && argc < Tmin + 3) {   // io_mods.cpp:594-599
      std::cerr << "Ground truth homography file is needed for verification type 1" << std::endl;
      return 1;
    }
  }
  if (argc >= Tmin + 4) config_fn = argv[Tmin + 3];
  if (argc >= Tmin + 5) iters_fn = argv[Tmin + 4];
  const bool pre_extracted = argc >= Tmin + 6 && atoi(argv[Tmin + 5]) > 0;   // conf1.read_pre_extracted, io_mods.cpp:602
  // argv[Tmin + 6] = match_one_to_many: parsed by the reference (io_mods.cpp:603) and used nowhere; ignored here too
  Config cfg;
  memset(&cfg.pair, 0, sizeof(cfg.pair));
  if (read_config(config_fn, iters_fn, ver_type, &cfg)) return 1;
  if (ver_type == 1) {   // mods.cpp:89-104: the file holds the matrix row by row
    std::ifstream gf(argv[Tmin + 2]);
    double *g = cfg.pair.ransac.gtH;
    if (!gf.is_open() || !(gf >> g[0] >> g[1] >> g[2] >> g[3] >> g[4] >> g[5] >> g[6] >> g[7] >> g[8])) {
      std::cerr << "Cannot open ground truth file " << argv[Tmin + 2] << std::endl;
      return 1;
    }
  }

  GreyImage img1, img2;
  if (!load_grey(img1_fn, cfg.load_color, &img1)) { std::cerr << "Cannot read image " << img1_fn << std::endl; return 1; }
  if (!load_grey(img2_fn, cfg.load_color, &img2)) { std::cerr << "Cannot read image " << img2_fn << std::endl; return 1; }
  if (cfg.verbose) std::cerr << "Image1: " << img1.w << "x" << img1.h << ", Image2: " << img2.w << "x" << img2.h << std::endl;

  if (mods_device_count() <= 0) { std::cerr << "mods: no MI355X / HIP device available (" << mods_last_error() << "); this build has no CPU path" << std::endl; return 1; }
  const int device = getenv("MODS_DEVICE") ? atoi(getenv("MODS_DEVICE")) : 0;
  const double diag = std::ceil(std::max(std::hypot((double)img1.w, (double)img1.h), std::hypot((double)img2.w, (double)img2.h))) + 2;
  mods_ctx *ctx = nullptr;
  const int n_det = (int)cfg.det_names.size();
  if (n_det == 0) { std::cerr << "The iterations file has no HessianAffine / DoG / HarrisAffine / MSER step with RootSIFT; nothing to do" << std::endl; return 1; }
  const bool hessian_only = n_det == 1 && cfg.det_names[0] == "HessianAffine" && cfg.groups.empty();
  std::vector<mods_imgrep *> reps1((size_t)n_det, nullptr), reps2((size_t)n_det, nullptr);
  void *d1 = nullptr, *d2 = nullptr;
  auto fail = [&](const char *what) { std::cerr << "mods: " << what << ": " << mods_last_error() << std::endl; return 1; };
  // two image slots: the same view of both images goes through one chain of launches when the images have one size.
  // One run of one pair: further contexts for the views of a step (MODS_LADDER_WORKERS) take longer to set up than they save
  setenv("MODS_LADDER_WORKERS", "1", 0);
  if (mods_ctx_create(device, (int)diag, (int)diag, 2, &ctx)) return fail("context");
  for (int d = 0; d < n_det; d++)
    if (mods_imgrep_create(ctx, 1 << 20, &reps1[d]) || mods_imgrep_create(ctx, 1 << 20, &reps2[d])) return fail("region banks");
  if (mods_dev_alloc(sizeof(float) * img1.px.size(), &d1) || mods_dev_alloc(sizeof(float) * img2.px.size(), &d2)) return fail("device memory");
  if (mods_dev_upload(d1, img1.px.data(), sizeof(float) * img1.px.size()) || mods_dev_upload(d2, img2.px.data(), sizeof(float) * img2.px.size())) return fail("upload");

  // one HessianAffine detector (the MODS_DEVICES path takes this form): steps without views are dropped from the ladder but
  // still count as steps of the loop
  std::vector<mods_ladder_step> steps;
  std::vector<int> step_index;
  if (hessian_only)
    for (size_t i = 0; i < cfg.steps.size(); i++)
      if (cfg.steps[i].n_tilts >= 0) { steps.push_back(cfg.steps[i]); step_index.push_back((int)i); }
  if (cfg.verbose) {
    std::cerr << "Detectors:";
    for (const std::string &nm : cfg.det_names) std::cerr << " " << nm;
    std::cerr << "; " << cfg.steps.size() / n_det << " step(s), minMatches = " << cfg.min_matches << std::endl;
  }
  double first_ratio = 0.8;
  for (const mods_ladder_step &st : cfg.steps) if (st.n_tilts >= 0 && st.fginn_ratio > 0) { first_ratio = st.fginn_ratio; break; }
  if (cfg.use_zmq) {
    while (!cfg.zmq_port.empty() && isspace((unsigned char)cfg.zmq_port.back())) cfg.zmq_port.pop_back();
    if (cfg.verbose) std::cerr << "Descriptors from the daemon at " << cfg.zmq_port << " (" << cfg.zmq_ps << "x" << cfg.zmq_ps << " patches)" << std::endl;
    if (mods_ctx_set_external_descriptor(ctx, &mods_zmq_descriptor_hook, (void *)cfg.zmq_port.c_str(), cfg.zmq_mr, cfg.zmq_ps)) return fail("external descriptor");
  }
  auto rtrim = [](std::string &t) { while (!t.empty() && isspace((unsigned char)t.back())) t.pop_back(); };
  if (cfg.aff_zmq) {
    rtrim(cfg.aff_port);
    if (cfg.verbose) std::cerr << "Affine shapes from the daemon at " << cfg.aff_port << std::endl;
    if (cfg.pair.det.doBaumberg) std::cerr << "Warning: [AffineAdaptation] useZMQ=1 with doBaumberg=1 in [HessianAffine]: the Baumberg frames are replaced" << std::endl;
    if (mods_ctx_set_external_shape(ctx, &mods_zmq_descriptor_hook, (void *)cfg.aff_port.c_str(), cfg.aff_mr, cfg.aff_ps)) return fail("external shape");
  }
  if (cfg.ori_zmq) {
    rtrim(cfg.ori_port);
    if (cfg.verbose) std::cerr << "Orientations from the daemon at " << cfg.ori_port << std::endl;
    if (mods_ctx_set_external_orientation(ctx, &mods_zmq_descriptor_hook, (void *)cfg.ori_port.c_str(), cfg.ori_mr, cfg.ori_ps)) return fail("external orientation");
  }
  // MODS_DEVICES=0,1,...: several GPUs for one pair (not with the ZMQ daemons: their hooks belong to one context)
  mods_multi *multi = nullptr;
  if (const char *md = getenv("MODS_DEVICES")) {
    std::vector<int> devs;
    for (const char *q = md; *q;) {
      char *e = nullptr;
      const long v = strtol(q, &e, 10);
      if (e == q) break;
      devs.push_back((int)v);
      q = *e == ',' ? e + 1 : e;
    }
    if (pre_extracted || cfg.use_zmq || cfg.aff_zmq || cfg.ori_zmq) std::cerr << "Note: MODS_DEVICES is ignored in pre-extracted mode and with ZMQ daemons" << std::endl;
    else if (!devs.empty()) {
      if (mods_multi_create(devs.data(), (int)devs.size(), std::max(img1.w, img2.w), std::max(img1.h, img2.h), 1 << 20, &multi)) return fail("multi-GPU setup");
      if (cfg.verbose) std::cerr << devs.size() << " device(s), exchange over " << (mods_multi_uses_rccl(multi) ? "RCCL" : "device copies") << std::endl;
    }
  }
  mods_ladder_result res;
  std::vector<double> matches((size_t)4 << 20);
  if (pre_extracted) {   // mods.cpp:196-229: one step, the banks come from the keypoint files
    std::vector<mods_region> r1, r2;
    if (!read_regions_any(k1_fn, &r1) || !read_regions_any(k2_fn, &r2)) return 1;
    if (cfg.verbose) std::cerr << "Pre-extracted regions: " << r1.size() << " | " << r2.size() << std::endl;
    if ((!r1.empty() && mods_imgrep_append_host(reps1[0], r1.data(), (int)r1.size())) || (!r2.empty() && mods_imgrep_append_host(reps2[0], r2.data(), (int)r2.size())))
      return fail("region banks");
    if (mods_match_verify_reps(ctx, reps1[0], reps2[0], first_ratio, &cfg.pair, &res, matches.data(), 1 << 20)) return fail("matching");
    res.n_unoriented[0] = (int)r1.size(); res.n_unoriented[1] = (int)r2.size();
  } else if (multi) {   // MODS_DEVICES: the views of every step sharded over several GPUs, one all-gather of the regions per step
    // (every detector, descriptor list and group of the configuration: the step loop of mods_match_ladder_groups_dev)
    if (mods_match_ladder_groups_multi(multi, img1.px.data(), img1.w, img1.h, img2.px.data(), img2.w, img2.h, cfg.steps.data(), cfg.det_params.data(),
                                       cfg.groups.empty() ? nullptr : cfg.groups.data(), cfg.group_pos, (int)cfg.steps.size() / n_det, n_det,
                                       cfg.min_matches, &cfg.pair, &res, matches.data(), 1 << 20))
      return fail("matching");
    for (int d = 0; d < n_det; d++) {      // for the keypoint files; owned by `multi`
      mods_imgrep_destroy(reps1[d]); mods_imgrep_destroy(reps2[d]);
      reps1[d] = mods_multi_bank_det(multi, 0, d); reps2[d] = mods_multi_bank_det(multi, 1, d);
    }
  } else if (hessian_only) {
    if (steps.empty()) { res = mods_ladder_result(); }
    else if (mods_match_ladder_dev(ctx, (const float *)d1, img1.w, img1.h, (const float *)d2, img2.w, img2.h, steps.data(), (int)steps.size(),
                                   cfg.min_matches, &cfg.pair, reps1[0], reps2[0], &res, matches.data(), 1 << 20))
      return fail("matching");
  } else if (mods_match_ladder_groups_dev(ctx, (const float *)d1, img1.w, img1.h, (const float *)d2, img2.w, img2.h, cfg.steps.data(), cfg.det_params.data(),
                                          cfg.groups.empty() ? nullptr : cfg.groups.data(), cfg.group_pos, (int)cfg.steps.size() / n_det, n_det, cfg.min_matches,
                                          &cfg.pair, reps1.data(), reps2.data(), &res, matches.data(), 1 << 20))
    return fail("matching");
  const double final_time = now_s() - c_start;
  const int final_step = res.steps_done <= 0 ? 0 : (hessian_only && !pre_extracted && !multi) ? step_index[res.steps_done - 1] + 1 : res.steps_done;
  if (cfg.verbose) {
    std::cerr << res.n_views << " views synthesised, " << res.n_tentatives << " tentatives found." << std::endl;
    std::cerr << res.n_unique << " unique tentatives left" << std::endl;
    if (cfg.pair.ransac.groundTruth) {
      std::cerr << "Ground truth verification is used..." << std::endl
                << res.gt_true << " true matches got with error threshold = " << cfg.pair.ransac.err_threshold << std::endl;
      if (cfg.pair.ransac.groundTruth >= 2)
        std::cerr << "Now RANSAC" << std::endl << res.gt_ransac_inliers << " RANSAC matches are identified" << std::endl
                  << res.gt_true_of_ransac << " RANSAC true matches are identified" << std::endl;
    } else {
      std::cerr << (cfg.pair.ransac.useF ? "LO-RANSAC(epipolar)" : "LO-RANSAC(homography)") << " verification is used..." << std::endl;
      std::cerr << res.n_inliers << " RANSAC correspondences got" << std::endl;
    }
  }
  std::cerr << "Done in " << final_step << " iterations" << std::endl << "*********************" << std::endl;

  {   // WriteLog, io_mods.cpp:10-66
    std::ofstream lf(log_fn);
    if (lf.is_open()) {
      const double ratio = res.n_unique > 0 ? (double)res.n_inliers / (double)res.n_unique : std::nan("");
      if (cfg.pair.ransac.groundTruth) {   // GR_TRUTH / GR_PLUS_RANSAC rows of WriteLog
        const double r_all = res.n_unique > 0 ? (double)res.gt_true / (double)res.n_unique : std::nan("");
        lf << std::setprecision(3) << final_time << " ";
        if (cfg.pair.ransac.groundTruth >= 2) {
          const double r_rs = res.gt_ransac_inliers > 0 ? (double)res.gt_true_of_ransac / (double)res.gt_ransac_inliers : std::nan("");
          lf << res.gt_true_of_ransac << " " << res.gt_ransac_inliers << " " << r_rs * 100 << " ";
        }
        lf << res.gt_true << " " << res.n_unique << " " << r_all * 100 << " " << res.n_described[0] << " " << res.n_described[1] << " "
           << final_step << " " << std::endl;
      } else {
        lf << std::setprecision(3) << final_time << " " << res.n_inliers << " " << res.n_unique << " " << ratio * 100 << " ";
        if (cfg.pair.ransac.useF) lf << res.n_described[0] << " " << res.n_described[1] << " ";
        else lf << res.n_unoriented[0] << " " << res.n_unoriented[1] << " ";
        lf << final_step << " " << std::endl;
      }
    }
  }
  if (cfg.output_h && argc >= Tmin + 3) {   // WriteH
    std::ofstream hf(argv[Tmin + 2]);
    if (hf.is_open())
      hf << res.H[0] << " " << res.H[1] << " " << res.H[2] << std::endl << res.H[3] << " " << res.H[4] << " " << res.H[5] << std::endl
         << res.H[6] << " " << res.H[7] << " " << res.H[8] << std::endl;
  }
  if (!log_only) {
    if (cfg.write_matches) {
      std::ofstream mf(match_fn);
      if (mf.is_open())
        for (int i = 0; i < res.n_inliers; i++)
          mf << matches[4 * (size_t)i] << " " << matches[4 * (size_t)i + 1] << " " << matches[4 * (size_t)i + 2] << " " << matches[4 * (size_t)i + 3] << std::endl;
    }
    if (cfg.write_keypoints && !pre_extracted) {   // mods.cpp:433-447
      if (ends_with(k1_fn, ".npz")) write_regions_npz(k1_fn, reps1);
      else write_regions(k1_fn, reps1, cfg.det_names, cfg.use_zmq ? "ZMQ" : "RootSIFT");
      if (ends_with(k2_fn, ".npz")) write_regions_npz(k2_fn, reps2);
      else write_regions(k2_fn, reps2, cfg.det_names, cfg.use_zmq ? "ZMQ" : "RootSIFT");
    }
  }
  std::cerr << "Image1: regions descriptors | Image2: regions descriptors " << std::endl;
  std::cerr << res.n_unoriented[0] << " " << res.n_described[0] << " | " << res.n_unoriented[1] << " " << res.n_described[1] << std::endl << std::endl;
  std::cerr << "True matches | unique tentatives" << std::endl;
  {
    const int tm = cfg.pair.ransac.groundTruth ? res.gt_true : res.n_inliers;   // TrueMatch1st
    if (res.n_unique > 0) std::cerr << tm << " | " << res.n_unique << " | " << std::setprecision(3) << 100.0 * tm / res.n_unique << "%  1st geom inc" << std::endl;
    else std::cerr << tm << " | " << res.n_unique << " |  -  1st geom inc" << std::endl;
    if (cfg.pair.ransac.groundTruth >= 2) {   // mods.cpp:515-519
      if (res.gt_ransac_inliers > 0) std::cerr << res.gt_true_of_ransac << " | " << res.gt_ransac_inliers << " | " << std::setprecision(3)
                                               << 100.0 * res.gt_true_of_ransac / res.gt_ransac_inliers << "% RANSACed  1st geom inc" << std::endl;
      else std::cerr << res.gt_true_of_ransac << " | " << res.gt_ransac_inliers << " | -  RANSACed  1st geom inc" << std::endl;
    }
  }
  const double total = now_s() - c_start;
  std::cerr << std::endl << "Main matching | All Time: " << std::endl << final_time << " | " << total << " seconds" << std::endl;
  if (cfg.time_log) {   // WriteTimeLog(TimingLog, file, 0, 1, 0): Synth Detect Orient Desc Match RANSAC MISC Total (seconds)
    const double dd = res.ms_detect_describe / 1000, mt = res.ms_match / 1000, rs = (res.ms_duplicates + res.ms_ransac) / 1000;
    std::ofstream tf("time.log");
    if (tf.is_open()) tf << 0.0 << " " << dd << " " << 0.0 << " " << 0.0 << " " << mt << " " << rs << " " << total - (dd + mt + rs) << " " << total << std::endl;
    std::cerr << "Timings: (sec) " << std::endl << "Synth+Detect+Orient+Desc|Match|RANSAC|MISC|Total " << std::endl
              << dd << " " << mt << " " << rs << " " << total - (dd + mt + rs) << " " << total << std::endl;
  }
  if (!multi) for (int d = 0; d < n_det; d++) { mods_imgrep_destroy(reps1[d]); mods_imgrep_destroy(reps2[d]); }
  mods_dev_free(d1); mods_dev_free(d2);
  mods_ctx_destroy(ctx);
  return 0;
}
