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
