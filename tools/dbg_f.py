import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import __graft_entry__ as ge, fsynth, refdeg
pkg = ge.load_package()
n, ratio, plane, noise, seed = 1000, 0.35, 0.0, 0.7, 7
u, _, _ = fsynth.two_view(n, ratio, plane, noise, seed=seed + n)
want = refdeg.ransac_f(u, 16.0, max_sam=20000, err="sampson", sym_check=0, seed_time=seed)
got = pkg.ransac_f(u, 16.0, max_sam=20000, err="sampson", sym_check=0, seed_time=seed)
print('ref', want['samples'], want['lo'], want['I'], want['Ih'], 'got', got['samples'], got['lo'], got['I'], got['Ih'], 'mask eq', np.array_equal(got['inl'], want['inl']))
dif = np.nonzero(got['hist'] != want['hist'])[0]
print('bins', dif, want['hist'][dif], got['hist'][dif])
# bisect on max_sam for the first differing sample
lo, hi = 1, 20000
while lo < hi:
    mid = (lo + hi) // 2
    w = refdeg.ransac_f(u, 16.0, max_sam=mid, err="sampson", sym_check=0, seed_time=seed, do_lo=0)
    g = pkg.ransac_f(u, 16.0, max_sam=mid, err="sampson", sym_check=0, seed_time=seed, do_lo=0)
    if np.array_equal(w['hist'], g['hist']): lo = mid + 1
    else: hi = mid
print('first differing sample', lo)
