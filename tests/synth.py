"""Seeded synthetic inputs: blob texture + value noise, and pairs related
by a random homography.  Pure numpy; used by tests, smoke() and bench.py."""
import numpy as np


def _value_noise(rng, w, h, octaves=6):
    out = np.zeros((h, w), np.float32)
    amp = 1.0
    for o in range(octaves):
        cell = max(2, 2 ** (octaves - o + 1))
        gw, gh = w // cell + 2, h // cell + 2
        g = rng.standard_normal((gh, gw)).astype(np.float32)
        ys = (np.arange(h, dtype=np.float32) / cell)
        xs = (np.arange(w, dtype=np.float32) / cell)
        y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        amp *= 0.6
    return out


def texture(w, h, seed=0, blobs=None):
    """uint8-valued float32 image, mean ~128: Gaussian blobs + multi-octave value noise + 2% uniform noise."""
    rng = np.random.default_rng(seed)
    if blobs is None:
        blobs = max(40, int(6000 * (w * h) / (1920.0 * 1080.0)))   # ~10k HessianAffine keypoints at 1080p
    img = np.full((h, w), 128.0, np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(blobs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        s = rng.uniform(1.5, 12.0) if rng.uniform() < 0.3 else rng.uniform(1.5, 4.0)
        amp = rng.uniform(20, 90) * rng.choice([-1.0, 1.0])
        r = int(4 * s) + 1
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x1 <= x0 or y1 <= y0:
            continue
        d2 = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2
        img[y0:y1, x0:x1] += (amp * np.exp(-d2 / (2 * s * s))).astype(np.float32)
    img += 12.0 * _value_noise(rng, w, h)
    img += rng.uniform(-2.5, 2.5, (h, w)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.float32)


def random_homography(rng, w, h):
    ang = np.deg2rad(rng.uniform(-25, 25))
    sc = rng.uniform(0.7, 1.4)
    c, s = np.cos(ang) * sc, np.sin(ang) * sc
    cx, cy = w / 2.0, h / 2.0
    T = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    Tb = np.array([[1, 0, cx + rng.uniform(-0.1, 0.1) * w], [0, 1, cy + rng.uniform(-0.1, 0.1) * h], [0, 0, 1.0]])
    P = np.eye(3)
    P[2, 0] = rng.uniform(-2e-4, 2e-4)
    P[2, 1] = rng.uniform(-2e-4, 2e-4)
    H = Tb @ P @ R @ T
    return H / H[2, 2]


def warp(img, H, noise_sigma=2.0, seed=0):
    """img2(x') = img(H^-1 x'), bilinear, out-of-image = 128, plus independent noise."""
    h, w = img.shape
    Hi = np.linalg.inv(H)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    sx = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den
    sy = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den
    x0 = np.floor(sx).astype(np.int64); y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0).astype(np.float32); fy = (sy - y0).astype(np.float32)
    valid = (x0 >= 0) & (y0 >= 0) & (x0 < w - 1) & (y0 < h - 1)
    x0c = np.clip(x0, 0, w - 2); y0c = np.clip(y0, 0, h - 2)
    a = img[y0c, x0c]; b = img[y0c, x0c + 1]; c = img[y0c + 1, x0c]; d = img[y0c + 1, x0c + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    out = np.where(valid, out, 128.0).astype(np.float32)
    rng = np.random.default_rng(seed + 7919)
    out += rng.normal(0, noise_sigma, out.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.float32)


def pair(w, h, seed=0):
    """(img1, img2, H) with img2 = img1 warped by H (img1 -> img2)."""
    img1 = texture(w, h, seed)
    rng = np.random.default_rng(seed + 104729)
    H = random_homography(rng, w, h)
    return img1, warp(img1, H, seed=seed), H


def _warp_src(H, xx, yy):
    Hi = np.linalg.inv(H)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    return (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den, (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den


def pair_two_planes(w, h, seed=0, noise_sigma=2.0, blobs=None):
    """(img1, img2, F, HA, HB): a scene of two planes seen from two viewpoints.  Image 1 is split by a slanted
    line; the two halves move with homographies HA and HB = HA + e a^T (same epipole e), so that all
    true correspondences satisfy x2^T F x1 = 0 with F = [e]x HA, and no single homography explains them."""
    img1 = texture(w, h, seed, blobs=blobs)
    rng = np.random.default_rng(seed + 15485863)
    HA = random_homography(rng, w, h)
    e = np.array([w * rng.uniform(2.5, 4.0) * rng.choice([-1, 1]), h * rng.uniform(-1.0, 2.0), 1.0])
    a = np.array([rng.uniform(-1, 1) * 6e-6, rng.uniform(-1, 1) * 6e-6, rng.uniform(0.012, 0.02) * rng.choice([-1, 1])])
    HB = HA + np.outer(e, a)
    HB = HB / HB[2, 2]
    # region A: left of a slanted line through the image centre
    nx, ny = np.cos(np.deg2rad(rng.uniform(-30, 30))), np.sin(np.deg2rad(rng.uniform(-30, 30)))
    side = lambda x, y: (x - w / 2.0) * nx + (y - h / 2.0) * ny < 0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    sxa, sya = _warp_src(HA, xx, yy)
    sxb, syb = _warp_src(HB, xx, yy)
    use_a = side(sxa, sya) & (sxa >= 0) & (sya >= 0) & (sxa < w - 1) & (sya < h - 1)
    use_b = ~use_a & ~side(sxb, syb) & (sxb >= 0) & (syb >= 0) & (sxb < w - 1) & (syb < h - 1)
    sx = np.where(use_a, sxa, sxb)
    sy = np.where(use_a, sya, syb)
    valid = use_a | use_b
    x0 = np.clip(np.floor(sx).astype(np.int64), 0, w - 2)
    y0 = np.clip(np.floor(sy).astype(np.int64), 0, h - 2)
    fx = (sx - x0).astype(np.float32)
    fy = (sy - y0).astype(np.float32)
    p, q, r, t = img1[y0, x0], img1[y0, x0 + 1], img1[y0 + 1, x0], img1[y0 + 1, x0 + 1]
    out = (p * (1 - fx) + q * fx) * (1 - fy) + (r * (1 - fx) + t * fx) * fy
    out = np.where(valid, out, 128.0).astype(np.float32)
    out += np.random.default_rng(seed + 7919).normal(0, noise_sigma, out.shape).astype(np.float32)
    img2 = np.clip(np.rint(out), 0, 255).astype(np.float32)
    ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
    F = ex @ HA
    return img1, img2, F / np.linalg.norm(F), HA, HB


def pair_partial(w, h, seed=0, frac=0.4):
    """(img1, img2, H): as pair(), but only the left `frac` of image 2 follows H; the rest shows the same scene under a second,
    unrelated homography (a second motion), so that about 1 - frac of the tentative matches are outliers to the dominant model
    - the verification then needs hundreds of samples instead of three (bench.py --inlier-ratio)."""
    img1 = texture(w, h, seed)
    rng = np.random.default_rng(seed + 104729)
    H = random_homography(rng, w, h)
    H2 = random_homography(np.random.default_rng(seed + 350377), w, h)
    a, b = warp(img1, H, seed=seed), warp(img1, H2, seed=seed + 1)
    cut = int(round(w * frac))
    out = a.copy()
    out[:, cut:] = b[:, cut:]
    return img1, out, H
