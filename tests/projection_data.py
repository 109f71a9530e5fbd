"""Raw XYZI returns for the projection tests: plausible sweeps plus the awkward cases (axis-aligned
points, the +-180 degree seam, the origin, column boundaries, non-finite values)."""
import numpy as np


def raw_returns(n, seed, nonfinite=True):
    rng = np.random.default_rng(seed)
    az = rng.random(n) * 2 * np.pi
    el = np.deg2rad(rng.uniform(-32, 12, n))
    r = rng.uniform(0.5, 90, n)
    x = r * np.cos(el) * np.cos(az)
    y = r * np.cos(el) * np.sin(az)
    z = r * np.sin(el)
    pts = np.stack([x, y, z, rng.random(n)], axis=1).astype(np.float32)
    k = n // 20
    # exact column boundaries of a 1024 / 1056-column image, both sides
    for cols, sl in ((1024, slice(0, k)), (1056, slice(k, 2 * k))):
        m = sl.stop - sl.start
        a = (rng.integers(0, cols, m) + 0.5) / cols * 2 * np.pi
        a = np.nextafter(a.astype(np.float32), np.float32(np.where(rng.random(m) < 0.5, -10, 10)))
        pts[sl, 0] = np.cos(a) * 20
        pts[sl, 1] = np.sin(a) * 20
    special = np.array([[1, 0, 0, 1], [-1, 0, 0, 1], [-1, -0.0, 0, 1], [-1, 1e-30, 0, 1], [-1, -1e-30, 0, 1],
                        [0, 1, 0, 1], [0, -1, 0, 1], [0, 0, 0, 1], [0, 0, 5, 1], [0, 0, -5, 1], [-0.0, 0.0, 1, 1],
                        [1e-40, 1e-40, 1e-40, 1], [3e38, 3e38, 3e38, 1], [1, 1, 1e30, 1], [1, -1, -1e30, 1]], np.float32)
    pts[2 * k:2 * k + len(special)] = special
    if nonfinite:
        bad = np.array([[np.nan, 1, 1, 1], [1, np.nan, 1, 1], [1, 1, np.nan, 1], [np.inf, 1, 1, 1], [1, -np.inf, 1, 1],
                        [np.inf, np.inf, 0, 1], [-np.inf, np.inf, np.inf, 1]], np.float32)
        pts[2 * k + len(special):2 * k + len(special) + len(bad)] = bad
    return pts


def kitti_returns(seed, variant="sweep"):
    """Raw KITTI-style (n, 4) returns in file order: ring after ring, the azimuth of a ring running 0+ .. 180, -180 .. 0-,
    so that a new ring shows as an azimuth zero crossing (KittiPointCloudSelect.cpp:214).
    variants: sweep (64 rings, ragged), late_start (first azimuth < 0), noisy_seam (several sign flips at each seam and
    returns exactly on the axes), short_rings (rings too short to be accepted), many_rings (more than 64), random (no
    structure at all: a crossing every few points), nonfinite."""
    rng = np.random.default_rng(seed)
    rings = 72 if variant == "many_rings" else 64
    az_all, el_all = [], []
    for r in range(rings):
        m = int(rng.integers(1700, 2084))
        if variant == "short_rings" and r % 5 == 2:
            m = int(rng.integers(200, 1300))   # around the 1250-point acceptance limit
        a = (np.arange(m) + rng.uniform(0.05, 0.95, m)) / m * 360.0
        if variant == "late_start" and r == 0:
            a = np.concatenate([rng.uniform(300, 359.9, 40), a])   # the file starts before the seam
        if variant == "noisy_seam":
            k = 12
            a[:k] = rng.uniform(-0.05, 0.05, k)                     # flips on both sides of zero
            a[-k:] = 360.0 + rng.uniform(-0.05, 0.05, k)
        az_all.append(a)
        el_all.append(np.full(len(a), 2.0 - 0.42 * r) + rng.normal(0, 0.02, len(a)))
    az = np.deg2rad(np.concatenate(az_all))
    el = np.deg2rad(np.concatenate(el_all))
    n = len(az)
    if variant == "random":
        az = rng.uniform(0, 2 * np.pi, n)
    rr = rng.uniform(3, 80, n)
    pts = np.stack([rr * np.cos(el) * np.cos(az), rr * np.cos(el) * np.sin(az), rr * np.sin(el), rng.random(n)],
                   axis=1).astype(np.float32)
    if variant == "noisy_seam":   # returns exactly on the axes: azimuth 0, -0, 180, -180, 90 and the origin
        idx = rng.choice(n, 60, replace=False)
        axes = np.array([[5, 0, 0], [5, -0.0, 0], [-5, 0, 0], [-5, -0.0, 0], [0, 5, 0], [0, 0, 0]], np.float32)
        pts[idx, :3] = axes[np.arange(60) % len(axes)]
    if variant == "nonfinite":
        idx = rng.choice(n, 40, replace=False)
        bad = np.array([[np.nan, 1, 1], [1, np.nan, 1], [1, 1, np.nan], [np.inf, 1, 1], [1, -np.inf, 1],
                        [np.inf, np.inf, 0], [-np.inf, -np.inf, 1], [np.nan, np.nan, 0]], np.float32)
        pts[idx, :3] = bad[np.arange(40) % len(bad)]
    return pts


KITTI_VARIANTS = ("sweep", "late_start", "noisy_seam", "short_rings", "many_rings", "random", "nonfinite")
