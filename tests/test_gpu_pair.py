"""-m gpu: the whole hot path on one pair vs the CPU oracle chain (identical counts, inlier set, H)."""
import numpy as np
import pytest

import refdeg
import synth

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not refdeg.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("w,h,seed", [(800, 600, 3), (1280, 960, 5)])
def test_pair_end_to_end(pkg, w, h, seed):
    import torch
    import pipeline_oracle as po
    a, b, Htrue = synth.pair(w, h, seed=seed)
    want = po.match_pair(a, b, seed_time=777)
    ctx = pkg.Context(0, w, h, 2)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(777)
    res, m = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, max_matches=100000)
    assert list(res.n_detected) == want["n_detected"] and list(res.n_described) == want["n_described"]
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] and res.n_inliers > 50
    wm = want["u6"][want["mask"]][:, [0, 1, 3, 4]]
    assert np.array_equal(m, wm)                                   # identical inlier set, same order
    Hg, Hw = np.array(res.H).reshape(3, 3), want["H"]
    assert np.max(np.abs(Hg / Hg[2, 2] - Hw / Hw[2, 2]) / np.maximum(1e-3, np.abs(Hw / Hw[2, 2]))) < 1e-4
    # and the recovered homography is the generating one
    assert np.max(np.abs(Hg / Hg[2, 2] - Htrue) / np.maximum(1.0, np.abs(Htrue))) < 5e-2
    ctx.close()


def test_pair_no_overlap(pkg):
    """Two unrelated images: verification must fail cleanly (H = -1, no inliers)."""
    import torch
    a, b = synth.texture(640, 480, 1), synth.texture(640, 480, 2)
    ctx = pkg.Context(0, 640, 480, 2)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(5)
    res, _ = pkg.match_pair_dev(ctx, t.data_ptr(), 640, 480)
    assert res.n_inliers == 0
    ctx.close()
