// ORACLE — TEST INFRASTRUCTURE ONLY.  First-geometrically-inconsistent nearest-neighbour
// matching with an exact linear search, and duplicate filtering, restated from
// matching/matching.cpp.
#include "orc.h"
#include <algorithm>
#include <cmath>

namespace orc {

// MatchFlannFGINN, matching.cpp:356-460, with [Matching] vector_matcher = linear
// (io_mods.cpp:389-390): cv::flann::Index(LinearIndexParams).knnSearch(k = nn).
//   * distance = squared L2 accumulated in float (FLANN L2<float>); descriptors are integers
//     0..255, every partial sum < 2^24, so the value is the exact integer distance;
//   * neighbour order = ascending distance, ties by ascending train index (FLANN's
//     KNNSimpleResultSet inserts an equal distance after the existing entries and the linear
//     index feeds points in index order).  FLANN is not in the image: tie rule unpinned.
// When the train list holds fewer than nn points only the existing neighbours are walked (the
// reference would read FLANN's -1 padding).
int match_fginn(const std::vector<Region> &list1, const std::vector<Region> &list2, std::vector<Tentative> &out,
                double currMatchRatio, double contradDist, int nn) {
  out.clear();
  const double sqminratio = currMatchRatio * currMatchRatio;
  const double contrDistSq = contradDist * contradDist;
  if (list1.empty() || list2.empty()) return 0;
  const int M = (int)list2.size();
  const int K = std::min(nn, M);
  std::vector<Tentative> slot(list1.size());
  std::vector<char> have(list1.size(), 0);
#pragma omp parallel num_threads(g_threads)
  {
  std::vector<std::pair<float, int>> all(M);     // per-thread scratch
  std::vector<int> idx(K);
  std::vector<float> dst(K);
#pragma omp for schedule(dynamic, 8)
  for (long i = 0; i < (long)list1.size(); i++) {
    for (int t = 0; t < M; t++) {
      float d = 0;
      for (int q = 0; q < 128; q++) {
        const float diff = (float)list1[i].desc[q] - (float)list2[t].desc[q];
        d += diff * diff;
      }
      all[t] = std::make_pair(d, t);
    }
    std::partial_sort(all.begin(), all.begin() + K, all.end());   // (dist, index) lexicographic
    for (int j = 0; j < K; j++) { dst[j] = all[j].first; idx[j] = all[j].second; }
    for (int j = 1; j < K; j++) {
      const double ratio = dst[0] / dst[j];
      const Region &n0 = list2[idx[0]], &nj = list2[idx[j]];
      const double dx = n0.x - nj.x, dy = n0.y - nj.y;
      const double dist1 = dx * dx + dy * dy;
      bool emit;
      if (sqminratio >= 1.0) emit = (j == nn - 1) || (dist1 > contrDistSq);
      else emit = (ratio <= sqminratio);
      if (emit) {
        Tentative tc;
        tc.q = (int)i; tc.t = idx[0]; tc.t_bad = idx[j]; tc.t_2nd = idx[1];
        tc.d1 = dst[0]; tc.d2 = dst[j]; tc.d2nd = dst[1];
        tc.ratio = std::sqrt(ratio);
        slot[i] = tc; have[i] = 1;
        break;
      }
      if (sqminratio < 1.0 && dist1 > contrDistSq) break;   // first contradictive
    }
  }
  }   // omp parallel
  for (size_t i = 0; i < list1.size(); i++)
    if (have[i]) out.push_back(slot[i]);
  return (int)out.size();
}

// MatchFLANNDistance, matching.cpp:572-633, with [Matching] binary_matcher = linear and binary_dist = Hamming (the default,
// io_mods.cpp:365): the descriptor bytes (floor of the stored values) of list1 against list2, the two nearest by Hamming
// distance - ascending distance, ties by ascending train index, as FLANN's KNNSimpleResultSet orders a linear scan - and a
// tentative for every query whose nearest lies within max_distance = (int)float(matchDistanceThreshold):
// d1, d2 = the two distances, ratio = d1 / d2 in double.  With a single train the reference reads a distance FLANN never
// wrote; it is INT_MAX (index -1) here.  The function clears the list it is given, so it REPLACES what MatchFlannFGINN put
// there when a descriptor has both thresholds (correspondencebank.cpp:328-334).
int match_distance(const std::vector<Region> &list1, const std::vector<Region> &list2, std::vector<Tentative> &out, double matchDistanceThreshold) {
  out.clear();
  const int max_distance = (int)float(matchDistanceThreshold);
  if (list1.empty() || list2.empty()) return 0;
  for (size_t i = 0; i < list1.size(); i++) {
    long long k1 = -1, k2 = -1;   // (distance << 32) | index, smallest two
    for (size_t j = 0; j < list2.size(); j++) {
      int d = 0;
      for (int b = 0; b < 128; b++) d += __builtin_popcount((unsigned)(list1[i].desc[b] ^ list2[j].desc[b]));
      const long long key = ((long long)d << 32) | (long long)j;
      if (k1 < 0 || key < k1) { k2 = k1; k1 = key; }
      else if (k2 < 0 || key < k2) k2 = key;
    }
    const int d1 = (int)(k1 >> 32);
    if (d1 <= max_distance) {
      Tentative tc;
      tc.q = (int)i; tc.t = (int)(k1 & 0xffffffffll);
      const int d2 = k2 < 0 ? 2147483647 : (int)(k2 >> 32);
      tc.t_bad = tc.t_2nd = k2 < 0 ? -1 : (int)(k2 & 0xffffffffll);
      tc.d1 = (float)d1; tc.d2 = (float)d2; tc.d2nd = (float)d2;
      tc.ratio = (double)tc.d1 / (double)tc.d2;
      out.push_back(tc);
    }
  }
  return (int)out.size();
}

// DuplicateFiltering, matching.cpp:2615-2679.  mode 1 = MODE_FGINN (sort by ratio), 2 =
// MODE_DISTANCE (sort by d1), 3 = MODE_BIGGER_REGION (sort by |s| of the first image's region, ascending:
// CompareCorrespondenceByScale, matching.cpp:74), 0 = MODE_RANDOM (keep order).  std::sort is unstable on equal
// keys; ties are fixed here as list order (stable sort).
void duplicate_filter(std::vector<Tentative> &tc, const std::vector<Region> &q, const std::vector<Region> &t,
                      double r, int mode) {
  if (r <= 0) return;
  const double r_sq = r * r;
  if (mode == 1)
    std::stable_sort(tc.begin(), tc.end(), [](const Tentative &a, const Tentative &b) { return std::fabs(a.ratio) < std::fabs(b.ratio); });
  else if (mode == 2)
    std::stable_sort(tc.begin(), tc.end(), [](const Tentative &a, const Tentative &b) { return std::fabs((double)a.d1) < std::fabs((double)b.d1); });
  else if (mode == 3)
    std::stable_sort(tc.begin(), tc.end(), [&](const Tentative &a, const Tentative &b) { return std::fabs((double)q[a.q].s) < std::fabs((double)q[b.q].s); });
  const size_t n = tc.size();
  std::vector<char> uniq(n, 1);
  for (size_t i = 0; i < n; i++) {
    if (!uniq[i]) continue;
    for (size_t j = i + 1; j < n; j++) {
      if (!uniq[j]) continue;
      double dx = q[tc[i].q].x - q[tc[j].q].x;
      double dy = q[tc[i].q].y - q[tc[j].q].y;
      const double d1_sq = dx * dx + dy * dy;
      if (d1_sq > r_sq) continue;
      dx = t[tc[i].t].x - t[tc[j].t].x;
      dy = t[tc[i].t].y - t[tc[j].t].y;
      const double d2_sq = dx * dx + dy * dy;
      if (d2_sq <= r_sq) uniq[j] = 0;
    }
  }
  std::vector<Tentative> keep;
  keep.reserve(n);
  for (size_t i = 0; i < n; i++)
    if (uniq[i]) keep.push_back(tc[i]);
  tc.swap(keep);
}

}  // namespace orc
