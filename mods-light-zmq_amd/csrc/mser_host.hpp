// Host half of the MSER detector (reference: detectors/mser/extrema/getExtrema.cpp, optThresh.cpp, sortPixels.cpp).
//
// The grey-level growth is sequential BY DEFINITION of its result: pixels enter in raster order inside a grey level, and
// that order decides which of two equally large regions survives a merge (getExtrema.cpp:266-283: the first label met, in
// the order up / left / right / down), the position of a new region in the region list (= the order of the output) and
// which label slot a run of small components ends up in.  Like DuplicateFiltering and the LO step of RANSAC it is
// control logic around counters, so it runs on a host core - one (image, polarity) per thread - and hands the device
//   * pix_slot[p] : the label slot (padded pixel offset) that was the root of p's component when p entered,
//   * tpar / tlev : for every slot, the slot it was merged into and the grey level of that merge (no path compression), bit 31 of
//                   tpar = "this slot's region has stable thresholds",
//   * the stable (slot, threshold, margin, area) list in the reference's output order,
// from which the kernels of mser.hip recover every region's pixel set (a pixel belongs to (slot, t) iff its level is <= t
// and the walk along tpar with merge levels <= t ends in slot), its runs, moments and affine frame.  The reference does that part
// by one flood fill per region (boundary.cpp) - time proportional to the summed region areas, which is where the GPU is used.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
#ifdef MSER_PROF
#include <chrono>
#endif

namespace mods {
namespace mser {

constexpr uint32_t kNoParent = 0x7fffffffu, kHasStable = 0x80000000u;
constexpr size_t kAhead = 24;   // pixels of look-ahead for the label prefetch

// The growth visits the pixels grey level by grey level, i.e. in an order that jumps all over the 8 - 16 MB of the label and order
// arrays: with 4 KB pages nearly every visit is also a TLB miss.  These two arrays ask for transparent huge pages (2 MB aligned,
// MADV_HUGEPAGE; where the kernel does not grant them nothing changes).
template <class T> class HugeBuf {
 public:
  HugeBuf() {}
  HugeBuf(const HugeBuf &) = delete;
  HugeBuf &operator=(const HugeBuf &) = delete;
  ~HugeBuf() { std::free(p_); }
  void resize(size_t n, bool zero) {
    if (n > cap_) {
      std::free(p_);
      const size_t bytes = (n * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
      p_ = (T *)std::aligned_alloc((size_t)2 << 20, bytes);
      if (!p_) throw std::bad_alloc();
      (void)madvise(p_, bytes, MADV_HUGEPAGE);
      cap_ = bytes / sizeof(T);
    }
    n_ = n;
    if (zero) std::memset(p_, 0, n * sizeof(T));
  }
  size_t size() const { return n_; }
  T *data() { return p_; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }
 private:
  T *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};

struct Stable { int slot, thresh, margin, area; };
struct GrowParams { int min_size; double max_area, min_margin; bool relative, invert; };

class Grower {
 public:
  // img: padded (w + 2) x (h + 2) bytes, the frame is never read.  pix_slot / tpar / tlev: (w + 2) * (h + 2) entries each (only
  // interior entries are written).  out: stable thresholds, regions in creation order, thresholds in list order.
  void run(const uint8_t *img, int w, int h, const GrowParams &gp, int32_t *pix_slot, uint32_t *tpar, uint8_t *tlev, std::vector<Stable> &out) {
#ifdef MSER_PROF
    prof_t0 = std::chrono::steady_clock::now();
#endif
    cols_ = w + 2;
    const int rows = h + 2;
    pix_slot_ = pix_slot; tpar_ = tpar; tlev_ = tlev;
    min_size_ = gp.min_size;
    min_size_int_ = std::min(10000, gp.min_size) * 4;                    // PrepareThresholds, getExtrema.cpp:373-382
    max_size_ = (int)((cols_ - 2) * (rows - 2) * gp.max_area);
    min_margin_ = gp.relative ? gp.min_margin / 100.0 : gp.min_margin;
    relative_ = gp.relative; invert_ = gp.invert;
    lab_.resize((size_t)rows * cols_, true);
    regs_.clear(); free_.clear(); free_head_ = 0; first_ = last_ = -1;
    // CalcHistogram + BinSortPixels (sortPixels.cpp:75-131): offsets of every grey level in raster order
    size_t hist[257] = {0};
    for (int y = 1; y <= h; y++) {
      const uint8_t *row = img + (size_t)y * cols_;
      for (int x = 1; x <= w; x++) hist[row[x] + 1]++;
    }
    for (int i = 0; i < 256; i++) hist[i + 1] += hist[i];
    order_.resize((size_t)w * h, false);
    {
      size_t cur[256];
      std::memcpy(cur, hist, sizeof(cur));
      for (int y = 1; y <= h; y++) {
        const uint8_t *row = img + (size_t)y * cols_;
        for (int x = 1; x <= w; x++) order_[cur[row[x]]++] = (uint32_t)(y * cols_ + x);
      }
    }
#ifdef MSER_PROF
    prof_t1 = std::chrono::steady_clock::now();
#endif
    const size_t n_px = order_.size();
    for (int level = 0; level < 256; level++)
      for (size_t k = hist[level]; k < hist[level + 1]; k++) {
        const int ofs = (int)order_[k];
        if (k + kAhead < n_px) {      // the visiting order is known: the three label lines of a later pixel are requested now
          const uint32_t nx = order_[k + kAhead];     // (105 -> 91 ms per 1080p polarity; requesting the slots those labels point at as
          __builtin_prefetch(&lab_[nx - cols_]);      // well costs more than it hides: 145 ms)
          __builtin_prefetch(&lab_[nx]);
          __builtin_prefetch(&lab_[nx + cols_]);
          __builtin_prefetch(&pix_slot_[nx], 1);
        }
        gather(ofs);
        if (n_lab_ == 0) {                                               // ConsRegion: size 1, border 4, a min-region
          lab_[ofs] = 0x00080004 | 1;
          pix_slot_[ofs] = ofs;
          tpar_[ofs] = kNoParent;
        } else if (n_lab_ == 1) insert(lab_slot_[0], ofs, level);
        else merge(ofs, level);
      }
#ifdef MSER_PROF
    prof_t2 = std::chrono::steady_clock::now();
#endif
    int root = cols_ + 1;
    if ((lab_[root] & 3) == 0) root = find(root);
    if (lab_[root] & 2) set_thresholds(regs_[(size_t)(lab_[root] >> 2)]);
    out.clear();
    for (int r = first_; r >= 0; r = regs_[r].next) {
      const Region &g = regs_[r];
      for (const Thr &t : g.th) out.push_back({g.slot, t.thresh, t.margin, g.hist[t.thresh].px});
      if (!g.th.empty()) tpar_[g.slot] |= kHasStable;
    }
  }

#ifdef MSER_PROF
  std::chrono::steady_clock::time_point prof_t0, prof_t1, prof_t2;
#endif
 private:
  struct Thr { int thresh, pos, margin; };
  struct Region {
    int minimum_int, maximum_int, pixel_total, border_total, slot, prev, next;
    std::vector<Thr> th;
    struct { int px, bd; } hist[256];     // pixels / borders of the reference's t_region, side by side: one line per insertion
  };

  // labels (getExtrema.cpp:19-34 with A64): 0 = not entered yet; low bits 00 = (slot << 2), a pointer to a slot; bit 0 = packed
  // min-region (size << 2 | border << 17); bit 1 = (region index << 2)
  int find(int l) {
    int p = (int)(lab_[l] >> 2);
    if (lab_[p] & 3) return p;
    do p = (int)(lab_[p] >> 2); while ((lab_[p] & 3) == 0);
    const int root = p;
    for (p = l; (lab_[p] & 3) == 0;) { const int nx = (int)(lab_[p] >> 2); lab_[p] = (uint64_t)root << 2; p = nx; }
    return root;
  }
  void gather(int ofs) {                                                 // GetLabelled, getExtrema.cpp:206-258
    const int nb[4] = {ofs - cols_, ofs - 1, ofs + 1, ofs + cols_};
    n_lab_ = 0;
    int touched = 0;
    uint64_t seen_v[4];
    int seen_root[4], n_seen = 0;
    for (int k = 0; k < 4; k++) {
      const uint64_t v = lab_[nb[k]];
      if (!v) continue;
      touched++;
      int root = -1;
      if (v & 3) root = nb[k];
      else {
        // neighbours inside one component mostly carry the same pointer: the slot it leads to is looked up once
        for (int q = 0; q < n_seen; q++) if (seen_v[q] == v) root = seen_root[q];
        if (root < 0) { root = find(nb[k]); seen_v[n_seen] = v; seen_root[n_seen++] = root; }
      }
      bool seen = false;
      for (int q = 0; q < n_lab_; q++) seen |= lab_slot_[q] == root;
      if (!seen) lab_slot_[n_lab_++] = root;
    }
    border_num_ = 2 * touched;
  }
  int new_region() {                                                     // suballoc.h: returned items first (LIFO), else a fresh zeroed one
    if (!free_.empty()) { const int r = free_.back(); free_.pop_back(); return r; }
    regs_.emplace_back();
    Region &g = regs_.back();
    std::memset(g.hist, 0, sizeof(g.hist));
    return (int)regs_.size() - 1;
  }
  void drop(int r) {
    Region &g = regs_[r];
    if (g.prev >= 0) regs_[g.prev].next = g.next; else first_ = g.next;
    if (g.next >= 0) regs_[g.next].prev = g.prev; else last_ = g.prev;
    // an item keeps its history arrays when it is recycled (the reference does not clear them either); every entry above the
    // current level is zero, and entries at or below it are overwritten or never read by the next owner
    free_.push_back(r);
  }
  void upgrade(int slot, int level) {                                    // UpgradeRegion, getExtrema.cpp:98-137
    const int ri = new_region();
    Region &g = regs_[ri];
    const uint64_t packed = lab_[slot] & ~(uint64_t)3;
    g.pixel_total = (int)((packed & 0x1fffc) >> 2);
    g.border_total = (int)(packed >> 17);
    g.slot = slot;
    g.minimum_int = g.maximum_int = level;
    g.hist[level].px = g.pixel_total;
    g.hist[level].bd = g.border_total;
    g.th.clear();
    g.prev = last_; g.next = -1;
    if (last_ >= 0) regs_[last_].next = ri; else first_ = ri;
    last_ = ri;
    lab_[slot] = ((uint64_t)ri << 2) | 2;
  }
  void insert(int slot, int ofs, int level) {                            // InsMarkPixel, getExtrema.cpp:139-166
    lab_[ofs] = (uint64_t)slot << 2;
    pix_slot_[ofs] = slot;
    if (lab_[slot] & 1) {
      lab_[slot] += (uint64_t)(int64_t)(0x00080004 - (border_num_ << 17));
      if ((int)(lab_[slot] & 0x1fffc) >= min_size_int_) upgrade(slot, level);
    } else {
      Region &g = regs_[(size_t)(lab_[slot] >> 2)];
      g.maximum_int = level;
      g.pixel_total++;
      g.border_total += 4 - border_num_;
      g.hist[level].px++;
      g.hist[level].bd += 4 - border_num_;
    }
  }
  void link(int from, int to, int level) {                               // the merge tree the kernels walk
    tpar_[from] = (uint32_t)to;
    tlev_[from] = (uint8_t)level;
  }
  void merge(int ofs, int level) {                                       // MergeRegions, getExtrema.cpp:260-355
    unsigned best = 0;
    int keep = lab_slot_[0], n_large = 0;
    for (int i = 0; i < n_lab_; i++) {
      const uint64_t v = lab_[lab_slot_[i]];
      if (v & 1) continue;
      const Region &g = regs_[(size_t)(v >> 2)];
      const unsigned size = (unsigned)(g.pixel_total - g.hist[level].px);   // its size one level below
      n_large++;
      if (size > best) { best = size; keep = lab_slot_[i]; }
    }
    if (!n_large) {
      for (int i = 1; i < n_lab_; i++) {
        lab_[keep] += lab_[lab_slot_[i]] & ~(uint64_t)3;
        lab_[lab_slot_[i]] = (uint64_t)keep << 2;
        link(lab_slot_[i], keep, level);
      }
    } else {
      const bool keep_is_min = (lab_[keep] & 1) != 0;
      const size_t keep_region = (size_t)(lab_[keep] >> 2);
      for (int i = 0; i < n_lab_; i++) {
        const int s = lab_slot_[i];
        if (s == keep) continue;
        const uint64_t v = lab_[s];
        const bool is_min = (v & 1) != 0;
        const int ri = (int)(v >> 2);
        lab_[s] = (uint64_t)keep << 2;
        link(s, keep, level);
        const int pt = is_min ? (int)((v & 0x1fffc) >> 2) : regs_[ri].pixel_total;
        const int bt = is_min ? (int)(v >> 17) : regs_[ri].border_total;
        if (keep_is_min) {                                               // int arithmetic widened with its sign (getExtrema.cpp:320)
          const uint32_t add = ((uint32_t)pt << 2) + ((uint32_t)bt << 17);
          lab_[keep] += (uint64_t)(int64_t)(int32_t)add;
        } else {
          Region &m = regs_[keep_region];
          m.pixel_total += pt; m.border_total += bt;
          m.hist[level].px += pt; m.hist[level].bd += bt;
        }
        if (!is_min) {
          Region &g = regs_[ri];
          if (!relative_ && (level - g.minimum_int + 1) <= min_margin_) drop(ri);
          else {
            g.maximum_int = level;
            set_thresholds(g);
            if (g.th.empty()) drop(ri);
          }
        }
      }
    }
    insert(keep, ofs, level);
  }
  // FastSetOptThresholds4StableRegion + SuppresOverlappingTresholds4StableRegions, optThresh.cpp:15-165
  void set_thresholds(Region &g) {
    if (g.pixel_total < min_size_) return;
    auto *H = g.hist;
    for (int i = g.minimum_int + 1; i <= g.maximum_int; i++) { H[i].px += H[i - 1].px; H[i].bd += H[i - 1].bd; }
    const int icons = invert_ ? 255 : 0, imul = invert_ ? -1 : 1;
    int up, best_margin = -1, best_pos = -1, i = g.minimum_int;
    auto emit = [&]() {
      const int th = best_pos + best_margin / 2;
      if (H[th].px <= max_size_ && H[th].px > min_size_) g.th.push_back({th, best_pos, best_margin});
    };
    do {
      up = (int)(i + min_margin_);
      if (up > g.maximum_int) break;
      while ((H[up].px - H[i].px < H[i].bd) && (up < g.maximum_int)) up++;
      const int margin = up - i;
      double quality = (double)margin;
      if (relative_) quality /= icons + imul * (i + (margin / 2));
      if (quality > min_margin_ && margin >= best_margin) { best_margin = margin; best_pos = i; }
      else {
        if (best_pos >= 0) { emit(); best_pos = -1; }
        best_margin = margin;
      }
      i++;
    } while (up < g.maximum_int);
    if (best_pos >= 0) emit();
    std::vector<Thr> &t = g.th;
    for (size_t k = 0; k < t.size();) {                                   // overlapping ranges: the larger margin stays
      bool gone = false;
      while (k + 1 < t.size()) {
        if ((t[k].pos + t[k].margin < t[k + 1].thresh) && (t[k].thresh < t[k + 1].pos)) break;
        if (t[k + 1].margin <= t[k].margin) t.erase(t.begin() + k + 1);
        else { t.erase(t.begin() + k); gone = true; break; }
      }
      if (!gone) k++;
    }
    for (size_t k = 0; k < t.size(); k++)                                 // neighbours within 10 % of area are joined
      while (k + 1 < t.size()) {
        if (t[k].pos + t[k].margin < t[k + 1].pos) break;
        if (H[t[k + 1].thresh].px - H[t[k].thresh].px <= 0.1 * H[t[k].thresh].px) {
          t[k].margin = t[k + 1].pos - t[k].pos + t[k + 1].margin;
          t[k].thresh = t[k].pos + t[k].margin / 2;
          t.erase(t.begin() + k + 1);
        } else break;
      }
  }

  int cols_ = 0, min_size_ = 0, min_size_int_ = 0, max_size_ = 0;
  double min_margin_ = 0;
  bool relative_ = false, invert_ = false;
  int32_t *pix_slot_ = nullptr; uint32_t *tpar_ = nullptr; uint8_t *tlev_ = nullptr;
  HugeBuf<uint64_t> lab_;
  HugeBuf<uint32_t> order_;
  std::vector<Region> regs_;
  std::vector<int> free_;
  size_t free_head_ = 0;
  int first_ = -1, last_ = -1;
  int lab_slot_[4], n_lab_ = 0, border_num_ = 0;
};

}  // namespace mser
}  // namespace mods
