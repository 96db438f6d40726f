"""Synthetic two-view correspondences for the F-matrix tests: a general 3-D scene seen by two cameras,
optionally dominated by a plane (the DEGENSAC case), plus uniform outliers and pixel noise."""
import numpy as np


def _rot(rng, max_deg):
    a = np.radians(rng.uniform(-max_deg, max_deg, 3))
    cx, cy, cz = np.cos(a)
    sx, sy, sz = np.sin(a)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def two_view(n, inlier_ratio=0.6, plane_ratio=0.0, noise=0.5, seed=0, size=(1920, 1080)):
    """Returns u6 (n x 6: x1 y1 1 x2 y2 1), boolean mask of true inliers, boolean mask of on-plane points."""
    rng = np.random.default_rng(seed)
    w, h = size
    K = np.array([[1.2 * w, 0, w / 2], [0, 1.2 * w, h / 2], [0, 0, 1]])
    R = _rot(rng, 12)
    t = np.array([rng.uniform(0.4, 0.9) * rng.choice([-1, 1]), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)])
    n_in = int(round(n * inlier_ratio))
    n_pl = int(round(n_in * plane_ratio))
    x1 = np.column_stack([rng.uniform(0, w, n_in), rng.uniform(0, h, n_in)])
    rays = (np.linalg.inv(K) @ np.column_stack([x1, np.ones(n_in)]).T).T
    depth = rng.uniform(3.0, 9.0, n_in)
    if n_pl:
        nrm = np.array([0.15, -0.1, 1.0])
        depth[:n_pl] = 5.0 / (rays[:n_pl] @ nrm)
    X = rays * depth[:, None]
    p2 = (K @ (R @ X.T + t[:, None])).T
    x2 = p2[:, :2] / p2[:, 2:3]
    ok = (p2[:, 2] > 0.1) & (x2[:, 0] > 0) & (x2[:, 0] < w) & (x2[:, 1] > 0) & (x2[:, 1] < h)
    x1 = x1 + rng.normal(0, noise, x1.shape)
    x2 = x2 + rng.normal(0, noise, x2.shape)
    u = np.ones((n, 6))
    u[:n_in, 0:2] = x1
    u[:n_in, 3:5] = x2
    n_out = n - n_in
    u[n_in:, 0:2] = np.column_stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)])
    u[n_in:, 3:5] = np.column_stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)])
    true_in = np.zeros(n, bool)
    true_in[:n_in] = ok
    on_plane = np.zeros(n, bool)
    on_plane[:n_pl] = ok[:n_pl]
    bad = np.where(~ok)[0]
    u[bad, 3:5] = np.column_stack([rng.uniform(0, w, len(bad)), rng.uniform(0, h, len(bad))])
    perm = rng.permutation(n)
    return np.ascontiguousarray(u[perm]), true_in[perm], on_plane[perm]
