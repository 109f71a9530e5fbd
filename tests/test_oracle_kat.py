"""CPU: hand-derived known-answer cases for the oracle.  The reference ships no
tests or golden vectors, so these are worked out by hand from the
reference source text; each case quotes the lines it exercises."""
import numpy as np

import bev_amd
import oracle_lib as orc
from bev_amd import POINT_DTYPE


def pt(x=0.0, y=0.0, z=0.0, intensity=0.5, row=0, col=0, t=0, label=-2):
    a = np.zeros(1, POINT_DTYPE)
    a["x"], a["y"], a["z"], a["intensity"] = x, y, z, intensity
    a["row"], a["col"], a["t"], a["label"] = row, col, t, label
    return a


def cloud(*pts):
    return np.concatenate(pts) if pts else np.empty(0, POINT_DTYPE)


def test_sensor_table():
    # src/Utility.cpp:96-118
    for kind, (n, h, g, res) in enumerate([(32, 1056, 20, 0.5), (64, 2083, 50, 0.25), (64, 1024, 31, 1.0)]):
        s = orc.sensor_kind(kind)
        assert (s.n_scan, s.horizon_scan, s.ground_upper_scan, s.height_res) == (n, h, g, res)


def test_belonging_grid():
    # BatchMultiBevGen.h:78-96: floor((p + offset) / 2), clamped to 75 x 50
    assert orc.belonging_grid(0.0, 0.0) == (37, 25)
    assert orc.belonging_grid(-75.0, -50.0) == (0, 0)
    assert orc.belonging_grid(-73.0001, -48.0001) == (0, 0)
    assert orc.belonging_grid(-73.0, -48.0) == (1, 1)
    assert orc.belonging_grid(74.99, 49.99) == (74, 49)
    assert orc.belonging_grid(1e6, 1e6) == (74, 49)      # clamped high (:84-86, :91-93)
    assert orc.belonging_grid(-1e6, -1e6) == (0, 0)      # clamped low  (:87-89, :94-96)
    assert orc.belonging_grid(1.9999, -0.0001) == (38, 24)
    # out-of-int-range and NaN: x86 cvttsd2si gives INT_MIN, which then clamps to 0
    assert orc.belonging_grid(3e38, float("nan")) == (0, 0)


def test_order_last_writer_wins_bounds_and_zero_fill():
    sp = orc.sensor_kind(0)  # HDL_32E
    a = pt(x=1, row=3, col=7, t=1)
    b = pt(x=2, row=3, col=7, t=2)            # same slot, later in input order -> wins (:115)
    c = pt(x=3, row=32, col=0)                # row == N_SCAN: dropped (:106-108)
    d = pt(x=4, row=0, col=1056)              # col == Horizon_SCAN: dropped (:109-111)
    e = pt(x=5, row=31, col=1055, label=7)    # last slot
    out = orc.order_cloud(sp, cloud(a, b, c, d, e))
    assert out.shape == (32 * 1056,)
    assert out[3 * 1056 + 7]["x"] == 2 and out[3 * 1056 + 7]["t"] == 2
    assert out[-1]["x"] == 5 and out[-1]["label"] == 7
    touched = np.zeros(len(out), bool)
    touched[[3 * 1056 + 7, len(out) - 1]] = True
    assert out[~touched].tobytes() == bytes(32 * int((~touched).sum()))  # value-initialised (:98)
    # order of a and b swapped: now a wins
    out2 = orc.order_cloud(sp, cloud(b, a))
    assert out2[3 * 1056 + 7]["x"] == 1


def test_bev_bins_layer_and_height():
    sp = orc.sensor_kind(0)  # HEIGHT_RES 0.5
    # x bin = round((x + 112) / 1 + 0.5), half away from zero (:279): x=0 -> round(112.5) = 113
    m = orc.multi_bev(sp, pt(x=0, y=0, z=0))
    # layer = round(0 / 0.5 + 2) = 2 (:281)
    assert np.argwhere(m == 255).tolist() == [[2, 113, 113]]
    s = orc.single_bev(pt(x=0, y=0, z=0))
    assert s[113, 113] == 8 and s.sum() == 8                      # int((0 + 2) * 4) = 8 (:345)
    # x = -112.5 -> v = -0.5 + 0.5 = 0 -> bin 0 (the only way to reach index 0)
    assert orc.single_bev(pt(x=-112.5, y=-112.5, z=1.0))[0, 0] == 12
    # x = -113 -> round(-0.5) = -1 -> skipped (:284)
    assert orc.single_bev(pt(x=-113.0, y=0, z=1.0)).sum() == 0
    # x = 110.4 -> round(222.9) = 223 (last row); x = 111 -> round(223.5) = 224 -> out of range
    assert orc.single_bev(pt(x=110.4, y=0, z=1.0))[223, 113] == 12
    assert orc.single_bev(pt(x=111.0, y=0, z=1.0)).sum() == 0
    # label 0 is skipped by both rasters (:285, :349)
    assert orc.single_bev(pt(z=1.0, label=0)).sum() == 0 and orc.multi_bev(sp, pt(z=1.0, label=0)).sum() == 0
    # height clamps to [0, 255] (:346); layer outside [0, 24) drops the point from the multi raster only
    assert orc.single_bev(pt(z=100.0))[113, 113] == 255
    assert orc.single_bev(pt(z=-5.0))[113, 113] == 0
    assert orc.multi_bev(sp, pt(z=100.0)).sum() == 0
    assert orc.multi_bev(sp, pt(z=-1.73)).sum() == 0              # round(-3.46 + 2) = -1
    assert np.argwhere(orc.multi_bev(sp, pt(z=10.6)) == 255).tolist() == [[23, 113, 113]]  # round(21.2 + 2) = 23
    assert orc.multi_bev(sp, pt(z=10.9)).sum() == 0                # round(21.8 + 2) = 24: past the last layer
    # max, not last-writer, in the single BEV (:353-355)
    s = orc.single_bev(cloud(pt(z=3.0), pt(z=1.0)))
    assert s[113, 113] == 20


def _flat_ring_cloud(sp, z=-1.7, r0=4.0, dr=1.0):
    """All slots valid; ring `row` sits on a horizontal plane z at radius r0 + dr * (N-1-row)."""
    N, H = sp.n_scan, sp.horizon_scan
    c = np.zeros(N * H, POINT_DTYPE)
    rows, cols = np.divmod(np.arange(N * H), H)
    rad = r0 + dr * (N - 1 - rows)
    az = 2 * np.pi * cols / H
    c["x"], c["y"], c["z"] = rad * np.cos(az), rad * np.sin(az), z
    c["intensity"], c["row"], c["col"], c["label"] = 0.5, rows, cols, -2
    return c


def test_mark_ground_flat_plane():
    sp = orc.sensor_kind(0)  # N=32, G=20 -> rows 12..31 are tested, row 11 is marked through row 12
    N, H, lo = 32, 1056, 12
    cl, gm, avg = orc.mark_ground(sp, _flat_ring_cloud(sp))
    # consecutive rings differ only horizontally -> angle 0 -> ground (:173-182) for rows lo..N-1 and lo-1
    assert (gm[lo - 1:] == 1).all() and (gm[:lo - 1] == 0).all()
    lab = cl["label"].reshape(N, H)
    assert (lab[lo - 1:] == 0).all() and (lab[:lo - 1] == -2).all()          # :244-246
    # every non-empty cell averages to sum/(n + 0.01) of identical heights: slightly above -1.7
    ne = avg != 0
    assert ne.any() and (avg[ne] > -1.7).all() and (avg[ne] < -1.5).all()


def test_mark_ground_invalid_marker_fallbacks_and_roof():
    sp = orc.sensor_kind(0)
    N, H, lo = 32, 1056, 12
    base = _flat_ring_cloud(sp)
    g = base.reshape(N, H)
    # (a) lower point has no return -> ground_mat = -1 there (:162-167); the row above (20, 100) is still
    #     tested as a lower point itself and stays ground
    c = base.copy().reshape(N, H)
    c[21, 100]["intensity"] = -1
    _, gm, _ = orc.mark_ground(sp, c.reshape(-1))
    assert gm[21, 100] == -1
    # row 22 at col 100 uses (21, 100) as upper -> falls back to (21, 102) (:146-149): still flat -> ground
    assert gm[22, 100] == 1
    # (b) upper and all three fallbacks invalid -> invalid (:151-167)
    c = base.copy().reshape(N, H)
    for rr, cc in [(21, 100), (21, 102), (21, 98), (20, 100)]:
        c[rr, cc]["intensity"] = -1
    _, gm, _ = orc.mark_ground(sp, c.reshape(-1))
    assert gm[22, 100] == -1
    # (c) col < 2: "(col - 2) % H" is negative in C++, the flat index lands in the previous row's tail
    #     (:151-154).  Make (21,0) and (21,2) invalid; the third candidate for (22,0) is flat index
    #     21*H - 2 = (20, H-2).  Give that point a height that makes the angle steep.
    c = base.copy().reshape(N, H)
    c[21, 0]["intensity"] = -1
    c[21, 2]["intensity"] = -1
    c[20, H - 2]["z"] = 50.0
    _, gm, _ = orc.mark_ground(sp, c.reshape(-1))
    # (22,0): steep vs the substituted upper -> not marked by its own test, but row 23's test
    # (upper = (22,0), flat) writes ground_mat(22,0) = 1 (:180-181)
    assert gm[22, 0] == 1
    # with row 23's lower point invalid, nothing marks (22,0) any more
    c[23, 0]["intensity"] = -1
    _, gm2, _ = orc.mark_ground(sp, c.reshape(-1))
    assert gm2[23, 0] == -1 and gm2[22, 0] == 0
    # (d) a "car roof": one flat patch 1.5 m above the ground is first marked ground by the angle
    #     test, then un-grounded because it is > 0.30 m above a neighbour cell's average (:227-241)
    c = base.copy().reshape(N, H)
    c[25:27, 200:260]["z"] = -0.2
    cl, gm, _ = orc.mark_ground(sp, c.reshape(-1))
    lab = cl["label"].reshape(N, H)
    assert (gm[25:27, 205:255] == 0).all() and (lab[25:27, 205:255] == -2).all()


def test_empty_cells_average_to_zero_and_unground_high_points():
    # App. B: an empty neighbour cell has avg 0/0.01 = 0, so a ground-like point at z > 0.30 next to
    # an empty cell is un-grounded
    sp = orc.sensor_kind(0)
    c = _flat_ring_cloud(sp, z=0.5)
    cl, gm, avg = orc.mark_ground(sp, c)
    assert (gm != 1).all() or (gm == 1).sum() < (32 - 11) * 1056  # most points lose the ground flag
    assert (avg[avg != 0] > 0.3).all()
