/*
 * mods_degensac.h — the degensac C ABI exported by libmodsgpu.so with the reference's exact
 * signatures, so that the reference's own matching/matching.cpp links against it unchanged.
 *
 * Replaces (reference root relative):
 *   degensac/exp_ranH.h:32-36   Score exp_ransacHcustom(...)
 *   degensac/exp_ranF.h:71-73   int exp_ransacFcustom(...)            (DEGENSAC, 7-point F)
 *   degensac/Htools.h           HDs, HDsSym, HDsSymMax, HDsi, HDsiSym, HDsiSymMax, HDsidx, HDsSymidx, HDsSymidxMax
 *   degensac/Ftools.h, matching/matching.cpp:44-70   FDs, FDsSym, FDsfull, exFDs, exFDsSym
 *   degensac/Fcustomdef.h:3-4   FDsPtr, exFDsPtr
 *   degensac/rtools.h:14-21     struct Score
 * Behaviour: same sampling sequence (bit-exact glibc srand/rand/random), same decisions; every
 * hypothesis is scored over all correspondences on the GPU.  th is in squared pixels
 * (matching.cpp:731).  *resids is malloc'ed here and freed by the caller (matching.cpp:732).
 * Unlike the reference (global HASH_TABLE, libc generator state) the entry points are re-entrant.
 */
#ifndef MODS_DEGENSAC_H
#define MODS_DEGENSAC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  unsigned I;   /* number of inliers */
  double J;     /* MSAC score */
} Score;

typedef void (*HDsPtr)(const double *, const double *, const double *, double *, int);
typedef void (*HDsiPtr)(const double *, const double *, const double *, double *, int, int *, int);
typedef void (*HDsidxPtr)(const double *, const double *, const double *, double *, int, int *, int);

Score exp_ransacHcustom(double *u, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                        int iter_type, int *data_out, int oriented_constraint, unsigned inlLimit, double **resids,
                        HDsPtr HDS1, HDsiPtr HDSi1, HDsidxPtr HDSidx1, int doSymCheck);

void HDs(const double *lin, const double *u, const double *H, double *p, int len);
void HDsSym(const double *lin, const double *u, const double *H, double *p, int len);
void HDsSymMax(const double *lin, const double *u, const double *H, double *p, int len);
void HDsi(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsiSym(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsiSymMax(const double *lin, const double *u6, const double *H, double *p, int len, int *pts, int ni);
void HDsidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz);
void HDsSymidx(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz);
void HDsSymidxMax(const double *lin, const double *mu, const double *H, double *p, int len, int *idx, int siz);

/* ---- fundamental matrix (matching.cpp:714-726 calls it with inlLimit 0, th = err_threshold^2) ----
 * F comes back as the reference stores it: x2^T F x1 = 0 with F[3*c + r] = entry (r, c) (error
 * functions index it as _f1.._f9 = F[0..8]).  data_out must be zero-initialised by the caller and
 * hold at least len + 3 ints: [0] samples, [1] LO runs, [2 + k] histogram of per-sample best inlier
 * counts (exp_ranF.c:1028).  *Ih = size of the largest plane (homography) consensus met on the
 * degenerate branch.  H_best is not written (the reference copies zero elements, exp_ranF.c:1199). */
typedef void (*FDsPtr)(const double *, const double *, double *, int);
typedef void (*exFDsPtr)(const double *, const double *, double *, double *, int);

int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl,
                      int *data_out, int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih,
                      exFDsPtr EXFDS1, FDsPtr FDS1, int doSymCheck);

void FDs(const double *u, const double *F, double *p, int len);          /* Sampson */
void FDsSym(const double *u, const double *F, double *p, int len);       /* symmetric epipolar distance */
void FDsfull(const double *u, const double *F, double *p, int len);
void exFDs(const double *u, const double *F, double *p, double *w, int len);
void exFDsSym(const double *u, const double *F, double *p, double *w, int len);

#ifdef __cplusplus
}
#endif
#endif
