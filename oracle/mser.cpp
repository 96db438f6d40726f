// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's MSER detector (the "CMP implementation",
// detectors/mser/extrema/): the grey-level union-find growth, the stability search, the region boundaries by flood fill, the
// run-length form, the ellipse moments and the DetectMSERs / DetectAffineRegions wrappers.  It keeps the reference's own flow
// (labels that point at labels, the linked region list with recycling, boundary -> run pairs) so that the product's different
// organisation (slot arrays + a time-stamped merge tree + GPU membership kernels) is checked by something built another way.
//
// PARITY UNPINNED: every unit of detectors/mser/ includes libExtrema.h -> extremaParams.h -> ../../helpers.h, structures.hpp ->
// <opencv2/core/core.hpp>; OpenCV is not in the image, so the reference's MSER cannot be compiled here (no stand-in headers),
// and the reference tree holds no MSER region list or count to compare with.  What pins this file is the reading of the
// source alone, plus the properties tests/test_cpu_mser.py checks (every region is the 4-connected component of its threshold
// set around its seed, areas equal the growth's counts, stability margins hold).
#include "orc.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

namespace orc {
namespace {

// getExtrema.h:19-45 with A64 (the reference's top CMakeLists.txt:5 defines it): 64-bit labels
constexpr uint64_t LABELPTR_MASK = 3, MINREG_FLAG = 1, REGION_FLAG = 2;
constexpr uint64_t LABEL_MASK = ~(uint64_t)3;
constexpr uint64_t REGION_SIZE_MASK = 0x1fffc;
constexpr int REGION_SIZE_SHIFT = 2, BORDER_SIZE_SHIFT = 17;

struct BPix { unsigned ofs; unsigned char direct; };             // t_borderpixel, extremaTypes.h:84-93

struct ThreshDef {                                               // t_thresh_def, extremaTypes.h:99-105
  int thresh, pos, margin;
  bool has_boundary = false;
  std::vector<BPix> boundary;
};

struct Reg {                                                     // t_region, extremaTypes.h:49-61
  int minimum_int = 0, pixel_total = 0, border_total = 0, seed = 0, maximum_int = 0;
  std::vector<ThreshDef> th;
  int pixels[256], borders[256];
  int prev = -1, next = -1;                                      // the regions list (LL): creation order, unlinked when dropped
};

struct ThreshPar { int min_size, min_size_int, max_size; double min_margin; bool relative; int invert; };   // t_thresh_par

struct Grower {
  int cols = 0, rows = 0;                    // padded sizes (BAry(-1, h, -1, w), preprocess.cpp:19-20)
  std::vector<unsigned char> img;            // padded image
  std::vector<uint64_t> lab;                 // labels_ptr: 0 unlabelled | (slot << 2) pointer | packed min-region | (region << 2) | 2
  std::vector<Reg> regs;
  std::deque<int> free_items;                // the suballocator's item list: returned items at the front, fresh blocks at the back
  int first = -1, last = -1;
  ThreshPar tp;
  int labelled[4], label_num = 0, border_num = 0;

  // suballoc.h:33-50, suballoc.cpp:27-52: a block of 100 zeroed items is appended when the list is empty; an item that
  // comes back keeps its contents (pixels[] / borders[] are NOT cleared: every level above the current one is still zero in
  // a recycled item, because an item is only ever written at levels <= the level at which it was returned)
  int get_item() {
    if (free_items.empty()) {
      const int base = (int)regs.size();
      regs.resize(base + 100);
      for (int i = 0; i < 100; i++) {
        Reg &r = regs[base + i];
        std::fill(r.pixels, r.pixels + 256, 0);
        std::fill(r.borders, r.borders + 256, 0);
        free_items.push_back(base + i);
      }
    }
    const int it = free_items.front();
    free_items.pop_front();
    return it;
  }
  void link_last(int r) {
    regs[r].prev = last; regs[r].next = -1;
    if (last >= 0) regs[last].next = r; else first = r;
    last = r;
  }
  void unlink_return(int r) {                // SuballocatorReturnItem(regionSuballocator, UnlinkLL(region))
    Reg &g = regs[r];
    if (g.prev >= 0) regs[g.prev].next = g.next; else first = g.next;
    if (g.next >= 0) regs[g.next].prev = g.prev; else last = g.prev;
    g.prev = g.next = -1;
    free_items.push_front(r);
  }

  // FindEquivLabel, getExtrema.cpp:178-204 (the flattening changes no result; kept so that the walk lengths are the same)
  int find_equiv(int l) {
    int ptr = (int)(lab[l] >> 2);
    if (lab[ptr] & LABELPTR_MASK) return ptr;
    do ptr = (int)(lab[ptr] >> 2); while ((lab[ptr] & LABELPTR_MASK) == 0);
    const int fin = ptr;
    ptr = l;
    while ((lab[ptr] & LABELPTR_MASK) == 0) {
      ptr = (int)(lab[ptr] >> 2);
      lab[l] = (uint64_t)fin << 2;
      l = ptr;
    }
    return fin;
  }

  // GetLabelled, getExtrema.cpp:206-258: the root slots around the pixel, in the order up, left, right, down
  void get_labelled(int ofs) {
    int l1 = ofs - cols, l2 = ofs - 1, l3 = ofs + 1, l4 = ofs + cols;
    label_num = 0; border_num = 0;
    if (lab[l1] != 0) {
      if ((lab[l1] & LABELPTR_MASK) == 0) l1 = find_equiv(l1);
      labelled[label_num++] = l1;
      border_num++;
    }
    if (lab[l2] != 0) {
      if ((lab[l2] & LABELPTR_MASK) == 0) l2 = find_equiv(l2);
      if (l2 != l1) labelled[label_num++] = l2;
      border_num++;
    }
    if (lab[l3] != 0) {
      if ((lab[l3] & LABELPTR_MASK) == 0) l3 = find_equiv(l3);
      if (l3 != l1 && l3 != l2) labelled[label_num++] = l3;
      border_num++;
    }
    if (lab[l4] != 0) {
      if ((lab[l4] & LABELPTR_MASK) == 0) l4 = find_equiv(l4);
      if (l4 != l1 && l4 != l2 && l4 != l3) labelled[label_num++] = l4;
      border_num++;
    }
    border_num = 2 * border_num;
  }

  // UpgradeRegion, getExtrema.cpp:98-137
  void upgrade_region(int slot, int intensity) {
    const int ri = get_item();
    Reg &r = regs[ri];
    const uint64_t min_reg = lab[slot] & LABEL_MASK;
    r.pixel_total = (int)((min_reg & REGION_SIZE_MASK) >> REGION_SIZE_SHIFT);
    r.border_total = (int)(min_reg >> BORDER_SIZE_SHIFT);
    r.seed = slot;                                   // minimum_pos.x = region_label - labels_ptr
    r.minimum_int = r.maximum_int = intensity;
    r.pixels[intensity] = r.pixel_total;
    r.borders[intensity] = r.border_total;
    r.th.clear();
    link_last(ri);
    lab[slot] = ((uint64_t)ri << 2) | REGION_FLAG;
  }

  // InsMarkPixel, getExtrema.cpp:139-166
  void ins_mark_pixel(int slot, int ofs, int intensity) {
    lab[ofs] = (uint64_t)slot << 2;
    if (lab[slot] & MINREG_FLAG) {
      lab[slot] += (uint64_t)(int64_t)(0x00080004 - (border_num << BORDER_SIZE_SHIFT));   // int arithmetic, then widened
      if ((int)(lab[slot] & REGION_SIZE_MASK) >= tp.min_size_int) upgrade_region(slot, intensity);
    } else {
      Reg &r = regs[(int)(lab[slot] >> 2)];
      r.maximum_int = intensity;
      r.pixel_total++;
      r.border_total += 4 - border_num;
      r.pixels[intensity]++;
      r.borders[intensity] += 4 - border_num;
    }
  }

  // SuppresOverlappingTresholds4StableRegions, optThresh.cpp:15-69, on a vector (DelElmLL(next) = erase(k + 1);
  // DelElmPrLL(current) + break + the loop's step = erase(k) and go on with the element that took its place)
  static void suppress_overlapping(Reg &r, const int *cum) {
    std::vector<ThreshDef> &t = r.th;
    for (size_t k = 0; k < t.size();) {
      bool removed_current = false;
      while (k + 1 < t.size()) {
        const ThreshDef &a = t[k], &b = t[k + 1];
        if ((a.pos + a.margin < b.thresh) && (a.thresh < b.pos)) break;
        if (b.margin <= a.margin) t.erase(t.begin() + k + 1);
        else { t.erase(t.begin() + k); removed_current = true; break; }
      }
      if (!removed_current) k++;
    }
    for (size_t k = 0; k < t.size(); k++) {
      while (k + 1 < t.size()) {
        ThreshDef &a = t[k]; const ThreshDef &b = t[k + 1];
        if (a.pos + a.margin < b.pos) break;
        if (cum[b.thresh] - cum[a.thresh] <= 0.1 * cum[a.thresh]) {
          a.margin = b.pos - a.pos + b.margin;
          a.thresh = a.pos + a.margin / 2;
          t.erase(t.begin() + k + 1);
        } else break;
      }
    }
  }

  // FastSetOptThresholds4StableRegion, optThresh.cpp:73-165
  void set_opt_thresholds(Reg &r) {
    if (r.pixel_total < tp.min_size) return;
    const int invertCons = tp.invert ? 255 : 0, invertMulti = tp.invert ? -1 : 1;
    int *cum = r.pixels, *cumB = r.borders;
    for (int i = r.minimum_int + 1; i <= r.maximum_int; i++) { r.pixels[i] += r.pixels[i - 1]; r.borders[i] += r.borders[i - 1]; }
    int up, localMaxMargin = -1, localMaxPos = -1;
    int i = r.minimum_int;
    auto emit = [&]() {
      ThreshDef t;
      t.thresh = localMaxPos + localMaxMargin / 2;
      if (cum[t.thresh] <= tp.max_size && cum[t.thresh] > tp.min_size) {
        t.pos = localMaxPos; t.margin = localMaxMargin;
        r.th.push_back(t);
      }
    };
    do {
      const int area_i = cum[i], radius_i = cumB[i];
      up = (int)(i + tp.min_margin);
      if (up > r.maximum_int) break;
      while ((cum[up] - area_i < radius_i) && (up < r.maximum_int)) up++;
      const int margin = up - i;
      double quality = (double)margin;
      if (tp.relative) quality /= invertCons + invertMulti * (i + (margin / 2));
      if (quality > tp.min_margin && margin >= localMaxMargin) {
        localMaxMargin = margin;
        localMaxPos = i;
      } else {
        if (localMaxPos >= 0) { emit(); localMaxPos = -1; }
        localMaxMargin = margin;
      }
      i++;
    } while (up < r.maximum_int);
    if (localMaxPos >= 0) { emit(); localMaxMargin = localMaxPos = -1; }
    suppress_overlapping(r, cum);
  }

  // MergeRegions, getExtrema.cpp:260-355
  void merge_regions(int ofs, int intensity) {
    unsigned maxSize = 0;
    int maxLabel = labelled[0];
    int num_large = 0;
    for (int i = 0; i < label_num; i++) {
      const uint64_t r = lab[labelled[i]];
      if (!(r & MINREG_FLAG)) {
        const Reg &g = regs[(int)(r >> 2)];
        const unsigned size = (unsigned)(g.pixel_total - g.pixels[intensity]);
        num_large++;
        if (size > maxSize) { maxSize = size; maxLabel = labelled[i]; }
      }
    }
    if (!num_large) {
      for (int i = 1; i < label_num; i++) {
        lab[maxLabel] += lab[labelled[i]] & LABEL_MASK;
        lab[labelled[i]] = (uint64_t)maxLabel << 2;
      }
    } else {
      const bool max_has_minstats = (lab[maxLabel] & MINREG_FLAG) != 0;
      const int maxRegion = (int)(lab[maxLabel] >> 2);      // meaningful only when !max_has_minstats
      for (int i = 0; i < label_num; i++) {
        const int label = labelled[i];
        if (label == maxLabel) continue;
        const uint64_t min_reg = lab[label];
        const int ri = (int)(min_reg >> 2);
        lab[label] = (uint64_t)maxLabel << 2;
        const bool merging_min_reg = (min_reg & MINREG_FLAG) != 0;
        int pixel_total, border_total;
        if (merging_min_reg) {
          pixel_total = (int)((min_reg & REGION_SIZE_MASK) >> REGION_SIZE_SHIFT);
          border_total = (int)(min_reg >> BORDER_SIZE_SHIFT);
        } else {
          pixel_total = regs[ri].pixel_total;
          border_total = regs[ri].border_total;
        }
        if (max_has_minstats) {
          // (pixel_total << 2) + (border_total << 17) is evaluated in int and widened with its sign (getExtrema.cpp:320)
          const uint32_t s = ((uint32_t)pixel_total << REGION_SIZE_SHIFT) + ((uint32_t)border_total << BORDER_SIZE_SHIFT);
          lab[maxLabel] += (uint64_t)(int64_t)(int32_t)s;
        } else {
          Reg &m = regs[maxRegion];
          m.pixel_total += pixel_total;
          m.border_total += border_total;
          m.pixels[intensity] += pixel_total;
          m.borders[intensity] += border_total;
        }
        if (!merging_min_reg) {
          Reg &g = regs[ri];
          if (!tp.relative && (intensity - g.minimum_int + 1) <= tp.min_margin) unlink_return(ri);
          else {
            g.maximum_int = intensity;
            set_opt_thresholds(g);
            if (g.th.empty()) unlink_return(ri);
          }
        }
      }
    }
    ins_mark_pixel(maxLabel, ofs, intensity);
  }

  // GetExtrema, getExtrema.cpp:385-437 (+ PrepareThresholds :373-382, CalcHistogram / BinSortPixels sortPixels.cpp:75-131)
  void grow(double min_margin, int min_size, double max_area, bool relative, bool invert) {
    tp.min_size = min_size;
    tp.min_size_int = std::min(10000, min_size) * 4;
    tp.max_size = (int)((cols - 2) * (rows - 2) * max_area);
    tp.min_margin = min_margin;
    if (relative) tp.min_margin /= 100.0;
    tp.invert = invert;
    tp.relative = relative;
    lab.assign((size_t)rows * cols, 0);
    regs.clear(); free_items.clear(); first = last = -1;
    std::vector<std::vector<int>> pix(256);
    for (int y = 1; y < rows - 1; y++)
      for (int x = 1; x < cols - 1; x++) pix[img[(size_t)y * cols + x]].push_back(y * cols + x);
    for (int i = 0; i < 256; i++)
      for (int ofs : pix[i]) {
        get_labelled(ofs);
        switch (label_num) {
          case 0: lab[ofs] = 0x00080004 | MINREG_FLAG; break;       // ConsRegion
          case 1: ins_mark_pixel(labelled[0], ofs, i); break;
          default: merge_regions(ofs, i);
        }
      }
    int root = cols + 1;
    if ((lab[root] & LABELPTR_MASK) == 0) root = find_equiv(root);
    if (lab[root] & REGION_FLAG) set_opt_thresholds(regs[(int)(lab[root] >> 2)]);   // (the reference assumes a region here)
  }

  // RegionBoundaries, boundary.cpp:99-208 (+ ConnectedComponent :49-65, add_if_inside :32-47)
  void region_boundaries() {
    std::vector<std::vector<std::pair<int, int>>> buckets(256);       // SortRegionThresholds :73-96
    int num = 0;
    for (int r = first; r >= 0; r = regs[r].next)
      for (size_t k = 0; k < regs[r].th.size(); k++) { buckets[regs[r].th[k].thresh].push_back({r, (int)k}); num++; }
    if (!num) return;
    std::vector<unsigned char> bck((size_t)rows * cols, 0);
    for (int x = 1; x < cols - 1; x++) bck[x] = bck[(size_t)(rows - 1) * cols + x] = 255;
    for (int y = 1; y < rows - 1; y++) bck[(size_t)y * cols] = bck[(size_t)y * cols + cols - 1] = 255;
    int dir_tab[9] = {0};
    dir_tab[1] = cols; dir_tab[2] = -1; dir_tab[4] = -cols; dir_tab[8] = 1;
    std::vector<BPix> inside;
    auto add_if_inside = [&](std::vector<BPix> &boundary, unsigned char max_int, int position, unsigned char direct, unsigned char marker) {
      const unsigned char v = bck[position];
      if (marker == v) return;
      BPix p; p.ofs = (unsigned)position; p.direct = direct;
      if (img[position] <= max_int && v != 255) { inside.push_back(p); bck[position] = marker; }
      else boundary.push_back(p);
    };
    for (int i = 0; i < 255; i++) {
      if (buckets[i].empty()) continue;
      inside.clear();
      for (auto &e : buckets[i]) {
        Reg &r = regs[e.first];
        ThreshDef &t = r.th[e.second];
        std::vector<BPix> boundary;
        const unsigned char thresh = (unsigned char)t.thresh;
        const unsigned char marker = thresh;
        if (t.has_boundary) {
          std::vector<BPix> &tmp = t.boundary;
          while (!tmp.empty()) {
            const BPix curr = tmp.back();
            if (!(img[curr.ofs] <= thresh && bck[curr.ofs] != 255)) boundary.push_back(curr);
            else {
              bck[(int)curr.ofs + dir_tab[curr.direct]] = marker;
              if (bck[curr.ofs] != marker) { inside.push_back(curr); bck[curr.ofs] = marker; }
            }
            tmp.pop_back();
          }
        } else {
          BPix p; p.ofs = (unsigned)r.seed; p.direct = 0;
          inside.push_back(p);
          bck[p.ofs] = marker;
        }
        while (!inside.empty()) {
          const int ofs = (int)inside.back().ofs;
          inside.pop_back();
          add_if_inside(boundary, thresh, ofs + cols, 4, marker);
          add_if_inside(boundary, thresh, ofs - cols, 1, marker);
          add_if_inside(boundary, thresh, ofs + 1, 2, marker);
          add_if_inside(boundary, thresh, ofs - 1, 8, marker);
        }
        std::sort(boundary.begin(), boundary.end(), [](const BPix &a, const BPix &b) { return a.ofs < b.ofs; });
        t.boundary.swap(boundary);
        t.has_boundary = true;
        if ((size_t)e.second + 1 < r.th.size()) {
          ThreshDef &nx = r.th[e.second + 1];
          nx.boundary = t.boundary;
          nx.has_boundary = true;
        }
      }
    }
  }
};

}  // namespace

// RLE2Ellipse, libExtrema.cpp:117-160
void mser_rle_to_ellipse(const std::vector<MserRun> &rle, double &barX, double &barY, double &sumX2, double &sumXY, double &sumY2) {
  double area = 0, sumX = 0, sumY = 0;
  for (size_t j = 0; j < rle.size(); j++) {
    const double line = rle[j].line, m = rle[j].col1, n = 1 + rle[j].col2;
    sumX += (n * n - m * m) / 2;
    sumY += (n - m) * (2 * line + 1) / 2;
    area += n - m;
  }
  barX = sumX / area;
  barY = sumY / area;
  sumX2 = sumY2 = sumXY = 0;
  for (size_t j = 0; j < rle.size(); j++) {
    const double line = rle[j].line - barY, m = rle[j].col1 - barX, n = 1 + rle[j].col2 - barX;
    const double l2 = line * line, m2 = m * m, n2 = n * n;
    sumX2 += (n2 * n - m2 * m) / 3;
    sumY2 += (n - m) * (3 * l2 + 3 * line + 1) / 3;
    sumXY += -.25 * (m2 - n2) * (2 * line + 1);
  }
  sumX2 /= area;
  sumY2 /= area;
  sumXY /= area;
}

// extremaRLERegions for one polarity (libExtrema.cpp:383-398): GetExtrema -> RegionBoundaries -> OutputRLEAndEll (:311-338;
// ReduceBoundary :237-252, ReducedBoundary2RLE :162-187).  img8: w x h, already inverted for MSER-.
void mser_regions(const unsigned char *img8, int w, int h, const MserParams &par, bool inverted, std::vector<MserRegion> &out) {
  Grower g;
  g.cols = w + 2; g.rows = h + 2;
  g.img.assign((size_t)g.cols * g.rows, 0);
  for (int y = 0; y < h; y++) std::copy(img8 + (size_t)y * w, img8 + (size_t)(y + 1) * w, g.img.begin() + (size_t)(y + 1) * g.cols + 1);
  g.grow(par.min_margin, par.min_size, par.max_area, par.relative, inverted);
  g.region_boundaries();
  for (int r = g.first; r >= 0; r = g.regs[r].next) {
    const Reg &reg = g.regs[r];
    for (const ThreshDef &t : reg.th) {
      MserRegion o;
      o.thresh = t.thresh; o.margin = t.margin; o.min_int = reg.minimum_int; o.max_int = reg.maximum_int;
      o.area = reg.pixels[t.thresh]; o.border = reg.borders[t.thresh];
      o.seed_x = reg.seed % g.cols - 1; o.seed_y = reg.seed / g.cols - 1;
      bool start = true;
      MserRun run{0, 0, 0};
      for (const BPix &b : t.boundary) {
        if (b.direct & 0x05) continue;
        const int line = (int)(b.ofs / g.cols) - 1, col = (int)(b.ofs % g.cols) - 1;
        if (start) { run.line = line; run.col1 = col + 1; }
        else { run.col2 = col - 1; o.rle.push_back(run); }
        start = !start;
      }
      mser_rle_to_ellipse(o.rle, o.cx, o.cy, o.sxx, o.sxy, o.syy);
      out.push_back(std::move(o));
    }
  }
}

// utls::Matrix2::schur_sym + sqrt + the product U * sqrt(T) * U^T (utls/matrix.cpp:185-216, 118-122, 63-71), as DetectMSERs uses them
static void ellipse_to_affine(double sxx, double sxy, double syy, double A[4]) {
  double t, r;
  if (sxy != 0) {
    r = (syy - sxx) / (2 * sxy);
    if (r >= 0) t = 1.0 / (r + std::sqrt(1 + r * r));
    else t = -1.0 / (-r + std::sqrt(1 + r * r));
    r = 1.0 / std::sqrt(1 + t * t);
    t = t * r;
  } else { r = 1; t = 0; }
  const double Q[4] = {r, t, -t, r};
  // T = Q^T * C * Q, off-diagonal elements set to zero
  const double Qt[4] = {Q[0], Q[2], Q[1], Q[3]};
  const double C[4] = {sxx, sxy, sxy, syy};
  double M[4] = {Qt[0] * C[0] + Qt[1] * C[2], Qt[0] * C[1] + Qt[1] * C[3], Qt[2] * C[0] + Qt[3] * C[2], Qt[2] * C[1] + Qt[3] * C[3]};
  double T[4] = {M[0] * Q[0] + M[1] * Q[2], M[0] * Q[1] + M[1] * Q[3], M[2] * Q[0] + M[3] * Q[2], M[2] * Q[1] + M[3] * Q[3]};
  T[1] = 0; T[2] = 0;
  const double S[4] = {std::sqrt(T[0]), std::sqrt(T[1]), std::sqrt(T[2]), std::sqrt(T[3])};
  const double US[4] = {Q[0] * S[0] + Q[1] * S[2], Q[0] * S[1] + Q[1] * S[3], Q[2] * S[0] + Q[3] * S[2], Q[2] * S[1] + Q[3] * S[3]};
  A[0] = US[0] * Qt[0] + US[1] * Qt[2];
  A[1] = US[0] * Qt[1] + US[1] * Qt[3];
  A[2] = US[2] * Qt[0] + US[3] * Qt[2];
  A[3] = US[2] * Qt[1] + US[3] * Qt[3];
}

// DetectMSERs, the ScalePyramid overload that DetectAffineRegions calls (extrema.cpp:196-295; the per-region "s" and the
// up-is-up rectification are commented out there for both polarities), prepareKeysForExport (:31-90), then the loop of
// DetectAffineRegions (synth-detection.hpp:96-110).  tilt, zoom: SynthImage fields of the view.
void detect_mser(const Img &image, const HessAffParams &p, double tilt, double zoom, std::vector<AffKey> &out,
                 std::vector<MserRegion> *regions_out) {
  MserParams ep;
  ep.min_size = p.mser_min_size; ep.max_area = p.mser_max_area; ep.relative = false;
  int reg_number = p.reg_number;
  if ((tilt > 2.0) || (zoom < 0.5)) reg_number = (int)std::floor(zoom * 2.0 * reg_number / tilt);
  ep.min_margin = p.mode != 0 ? 1.0 : p.mser_min_margin;
  const int w = image.w, h = image.h;
  std::vector<unsigned char> im((size_t)w * h);
  for (size_t i = 0; i < im.size(); i++) im[i] = (unsigned char)(int)image.d[i];   // *ptr = (unsigned char)*in_ptr, extrema.cpp:230
  std::vector<MserRegion> plus, minus;
  mser_regions(im.data(), w, h, ep, false, plus);
  for (auto &v : im) v = 255 - v;                                    // InvertImageAndHistogram, sortPixels.cpp:134-153
  mser_regions(im.data(), w, h, ep, true, minus);
  std::vector<AffKey> keys;
  keys.reserve(plus.size() + minus.size());
  for (int pol = 0; pol < 2; pol++)
    for (const MserRegion &r : pol ? minus : plus) {
      AffKey k;
      double A[4];
      ellipse_to_affine(r.sxx, r.sxy, r.syy, A);
      k.x = r.cx; k.y = r.cy; k.s = 1.0;
      k.a11 = A[0]; k.a12 = A[1]; k.a21 = A[2]; k.a22 = A[3];
      k.response = r.margin;
      k.sub_type = pol ? 20 : 21;
      k.octave = r.thresh; k.level = pol; k.r0 = r.seed_y; k.c0 = r.seed_x;   // provenance for the tests
      keys.push_back(k);
    }
  if (regions_out) { *regions_out = plus; regions_out->insert(regions_out->end(), minus.begin(), minus.end()); }
  if (p.mode != 0 && !keys.empty()) {
    // std::sort, not stable_sort: the responses are small integers, so the order inside a group of equal margins is
    // whatever libstdc++'s introsort leaves - the product runs the same std::sort on the same sequence
    auto cmp = [](const AffKey &k1, const AffKey &k2) { return std::fabs(k1.response) > std::fabs(k2.response); };
    std::sort(keys.begin(), keys.end(), cmp);
    const double maxResponse = std::fabs(keys[0].response);
    const int regNumber = (int)keys.size();
    auto above = [&](double thr) { int m = 0; while (m < regNumber && std::fabs(keys[m].response) > std::fabs(thr)) m++; return m; };
    int keep = regNumber;
    switch (p.mode) {
      case 1: keep = above(maxResponse * p.rel_threshold); break;                      // double effectiveThreshold here
      case 2: if (reg_number < regNumber && reg_number >= 0) keep = reg_number; break;
      case 3: keep = (int)std::floor(p.rel_reg_number * (double)keys.size()); break;
      case 4: {
        const int fixTh = above(1.0);                                                   // tempKey.response = par.min_margin (= 1.0 here)
        keep = fixTh < reg_number ? std::min(reg_number, regNumber) : std::min(fixTh, regNumber);
        break;
      }
    }
    if (keep < 0) keep = 0;
    if (keep < regNumber) keys.resize(keep);
  }
  out.clear();
  out.reserve(keys.size());
  for (AffKey k : keys) {
    k.s = k.s * std::sqrt(std::fabs(k.a11 * k.a22 - k.a12 * k.a21));
    rectify_transformation(k.a11, k.a12, k.a21, k.a22);
    out.push_back(k);
  }
}

}  // namespace orc
