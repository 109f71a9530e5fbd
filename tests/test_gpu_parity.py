"""GPU: the HIP path, called through the C ABI, against the CPU oracle.
Bit-exact everywhere: every output is integer / byte / index data, and the one
float channel (per-cell average heights) is compared bit for bit too."""
import hashlib

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu

SENSORS = ["HDL_32E", "HDL_64E", "OS1_64"]


@pytest.fixture(scope="module")
def ctxs():
    made = {}

    def get(sensor, max_batch=4, max_points=200000):
        key = (sensor, max_batch, max_points)
        if key not in made:
            p = bev_amd.params_for_sensor(sensor)
            made[key] = (p, bev_amd.BevContext(p, device=0, max_batch=max_batch, max_points=max_points))
        return made[key]

    yield get
    for _, c in made.values():
        c.close()


def _oracle(p, pts):
    sp = orc.sensor_from_params(p)
    ordered, gm, multi, single = orc.process_frame(sp, pts)
    _, _, avg = orc.mark_ground(sp, orc.order_cloud(sp, pts))
    return ordered, gm, multi, single, avg


def _check_batch(p, ctx, frames):
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    # the debug hook exposes the averages of the LAST chunk only (host-buffer calls run in chunks of max_batch / 2)
    chunk = max(1, ctx.max_batch // 2)
    first = ((len(frames) - 1) // chunk) * chunk if frames else 0
    avg = ctx.cell_avg(0, len(frames) - first) if frames else None
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single, o_avg = _oracle(p, pts)
        if i >= first:
            assert avg[i - first].tobytes() == o_avg.tobytes(), f"frame {i}: per-cell averages differ"
        assert np.array_equal(gm[i], o_gm), f"frame {i}: ground_mat differs at {np.argwhere(gm[i] != o_gm)[:4]}"
        assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud / labels differ"
        assert np.array_equal(multi[i], o_multi), f"frame {i}: multi BEV differs"
        assert np.array_equal(single[i], o_single), f"frame {i}: single BEV differs"


@pytest.mark.parametrize("sensor", SENSORS)
def test_sweep_frames(ctxs, sensor):
    p, ctx = ctxs(sensor)
    _check_batch(p, ctx, [synth.sweep(p, f) for f in range(3)])


@pytest.mark.parametrize("sensor", SENSORS)
def test_firing_order_frames(ctxs, sensor):
    p, ctx = ctxs(sensor)
    _check_batch(p, ctx, [synth.firing_order(p, f) for f in range(2)])


@pytest.mark.parametrize("sensor", SENSORS)
@pytest.mark.parametrize("nonfinite", [False, True])
def test_adversarial_frames(ctxs, sensor, nonfinite):
    p, ctx = ctxs(sensor)
    _check_batch(p, ctx, [synth.adversarial(p, 50000 + 7777 * s, s, nonfinite) for s in range(3)])


def test_ragged_batch_with_empty_and_tiny_frames(ctxs):
    p, ctx = ctxs("HDL_32E")
    full = synth.sweep(p, 0)
    frames = [np.empty(0, bev_amd.POINT_DTYPE), full[:1], full, full[:1000], np.empty(0, bev_amd.POINT_DTYPE),
              synth.sweep(p, 1, keep=0.3, n_dup=20000)]
    _check_batch(p, ctx, frames)  # 6 frames through a max_batch=4 context: host-buffer chunks of max_batch / 2 frames


def test_host_buffer_call_through_the_two_stage_pipeline(ctxs):
    """bev_process_batch works in chunks of max_batch / 2 frames; a chunk of 8 or more frames is cut into two sub-batches
    so that the front of one overlaps the back of the other (workspace hand-over between the stages)."""
    p, ctx = ctxs("HDL_32E", 32)
    frames = [synth.sweep(p, 200 + f, keep=0.7 + 0.01 * f, n_dup=100 * f) for f in range(37)]  # chunks of 16, 16, 5
    _check_batch(p, ctx, frames)


def test_every_turn_of_the_cell_quarters(ctxs):
    """k_cell_sums hands the four cell quarters of a frame to its four workgroups in an order that turns from frame to
    frame of the launch ((f >> 3) mod 4: a quarter that runs long must not always land on the same CUs).  56 frames through a
    max_batch = 128 context are one host-buffer chunk cut into two launches of 28: every turn is taken.  Firing-order frames:
    their quarters differ most."""
    p, ctx = ctxs("OS1_64", 128)
    frames = [synth.firing_order(p, 300 + f) if f % 3 else synth.sweep(p, 300 + f, keep=0.9, n_dup=500) for f in range(56)]
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    for i, pts in enumerate(frames):  # (labels and rasters hang on the averages; the debug hook shows the last launch's alone)
        o_ord, o_gm, o_multi, o_single, _ = _oracle(p, pts)
        assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud / labels differ"
        assert np.array_equal(gm[i], o_gm), i
        assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i


def test_zero_heights_in_the_origins_cell(ctxs):
    """Records without a return (x = y = z = 0) are ground candidates of the cell that holds the origin; k_cell_sums counts
    the cell's zero heights instead of sorting and adding them (s + 0 = s: the running sum is never -0).  Frames where that
    cell also holds -0.0 heights, heights that cancel to zero and ordinary ones, in every order the slots give, must leave
    the same averages — bit for bit, the sign of a zero included — labels and rasters as the oracle's chain."""
    p, ctx = ctxs("OS1_64", 16)
    rng = np.random.default_rng(77)
    frames = []
    for k in range(6):
        f = synth.firing_order(p, 400 + k).copy()
        n = len(f)
        # a share of the records dropped (all-zero), another share moved INTO the origin's cell (x in [-1, 1), y in [0, 2))
        # with heights from a small pool: +-0, values that cancel, an ordinary one
        drop = rng.random(n) < (0.0, 0.02, 0.2, 0.5, 0.2, 0.2)[k]
        f[drop] = np.zeros(1, f.dtype)[0]
        f["row"][drop] = (np.arange(n) % p.n_scan)[drop]                      # (firing order: beam = position mod N)
        f["col"][drop] = np.minimum(np.arange(n) // p.n_scan, p.horizon_scan - 1)[drop]
        near = rng.random(n) < (0.0, 0.01, 0.05, 0.05, 0.3, 0.05)[k]
        f["x"][near] = rng.uniform(-0.9, 0.9, near.sum()).astype(np.float32)
        f["y"][near] = rng.uniform(0.1, 1.9, near.sum()).astype(np.float32)
        pool = np.array([0.0, -0.0, 1.5, -1.5, 0.25, -0.25, 3e-39, -1.7], np.float32) if k != 4 else np.array([-0.0, 0.0], np.float32)
        f["z"][near] = pool[rng.integers(0, len(pool), near.sum())]
        if k == 5:
            f["z"][drop] = np.float32(-0.0)                                      # dropped records with a NEGATIVE zero height
        frames.append(f)
    _check_batch(p, ctx, frames)


def test_degenerate_clouds(ctxs):
    p, ctx = ctxs("HDL_32E")
    base = synth.sweep(p, 5)
    allsame = base[:5000].copy()
    allsame["row"] = 31
    allsame["col"] = 0
    noret = base.copy()
    noret["intensity"] = -1.0
    lab0 = base.copy()
    lab0["label"] = 0
    oob = base[:3000].copy()
    oob["row"] = 40
    _check_batch(p, ctx, [allsame, noret, lab0, oob])


@pytest.mark.parametrize("n_sweeps", [20, 60])
def test_oxford_concat(ctxs, n_sweeps):
    """BASELINE configs[4]: 60 concatenated sweeps = ~2 M points into 33,792 slots (P >> S, getOrderedCloud
    BatchMultiBevGen.cpp:102-116); at that size the winner entries need 21 index bits and keep 11 tag bits."""
    p = bev_amd.params_for_sensor("HDL_32E")
    pts = synth.concat(p, 0, n_sweeps=n_sweeps)
    if n_sweeps == 60:
        assert len(pts) > 1_900_000
    p, ctx = ctxs("HDL_32E", 2, len(pts))
    _check_batch(p, ctx, [pts, synth.concat(p, 1, n_sweeps=n_sweeps)])


# ---- per-function entry points (what the reference-named C++ functions call) ----
@pytest.mark.parametrize("sensor", SENSORS)
def test_per_function_entry_points(ctxs, sensor):
    p, ctx = ctxs(sensor)
    sp = orc.sensor_from_params(p)
    pts = synth.sweep(p, 11)
    ordered = ctx.order_cloud(pts)                      # getOrderedCloud
    o_ordered = orc.order_cloud(sp, pts)
    assert ordered.tobytes() == o_ordered.tobytes()
    marked, gm = ctx.mark_ground(ordered)               # markGroundPoints
    o_marked, o_gm, _ = orc.mark_ground(sp, o_ordered)
    assert np.array_equal(gm, o_gm)
    assert marked.tobytes() == o_marked.tobytes()
    assert np.array_equal(ctx.multi_bev(marked), orc.multi_bev(sp, o_marked))    # computeAndSaveMultiBev raster
    assert np.array_equal(ctx.single_bev(marked), orc.single_bev(o_marked))      # computeAndSaveSingleBev raster
    # rasters of an UNORDERED cloud (any point list is legal input)
    adv = synth.adversarial(p, 30000, 9, True)
    assert np.array_equal(ctx.multi_bev(adv), orc.multi_bev(sp, adv))
    assert np.array_equal(ctx.single_bev(adv), orc.single_bev(adv))
    # markGroundPoints is idempotent on its own output
    again, gm2 = ctx.mark_ground(marked)
    assert again.tobytes() == marked.tobytes() and np.array_equal(gm2, gm)


def test_angle_predicate_device_vs_libm(ctxs):
    """The transcendental-free angle test on the device == the reference's atan2f
    expression evaluated by the host libm, including +-1 ulp around the threshold."""
    p, ctx = ctxs("HDL_32E")
    rng = np.random.default_rng(1234)
    n = 2_000_000
    s = np.ldexp(1.0 + rng.random(n), rng.integers(-20, 20, n)).astype(np.float32)
    t = (np.float64(0.17632698070846498) * s.astype(np.float64)).astype(np.float32)
    dz = (t.view(np.uint32) + rng.integers(-8, 9, n).astype(np.int64)).astype(np.uint32).view(np.float32)
    dz = np.where(rng.random(n) < 0.5, -dz, dz).astype(np.float32)
    ang = rng.random(n) * 2 * np.pi
    dx = (s.astype(np.float64) * np.cos(ang)).astype(np.float32)
    dy = (s.astype(np.float64) * np.sin(ang)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1e-45, 3e38, 1e-38, -1.0], np.float32)
    g = np.array(np.meshgrid(special, special, special)).reshape(3, -1)
    dx = np.concatenate([dx, g[0]]); dy = np.concatenate([dy, g[1]]); dz = np.concatenate([dz, g[2]])
    dev = ctx.angle_predicate(dx, dy, dz)
    lib = orc.lib()
    ref = np.fromiter((lib.oracle_angle_is_ground(float(a), float(b), float(c)) for a, b, c in
                       zip(dx[:200000], dy[:200000], dz[:200000])), np.uint8, 200000)
    assert np.array_equal(dev[:200000], ref)
    import hostcheck_lib as hc
    assert np.array_equal(dev, hc.angle(dx, dy, dz)), "device and host evaluation of bev_exact.h differ"
    refs = np.fromiter((lib.oracle_angle_is_ground(float(a), float(b), float(c)) for a, b, c in
                        zip(g[0], g[1], g[2])), np.uint8, g.shape[1])
    assert np.array_equal(dev[n:], refs)


def test_device_resident_matches_host_entry(ctxs):
    torch = pytest.importorskip("torch")
    p, ctx = ctxs("HDL_64E", 4)
    frames = [synth.sweep(p, 100 + f) for f in range(6)]
    ordered, multi, single, _ = ctx.process_batch(frames)
    offs = np.zeros(len(frames) + 1, np.uint64)
    offs[1:] = np.cumsum([len(f) for f in frames])
    packed = np.concatenate(frames)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(packed.view(np.uint8).reshape(-1)).to(dev)
    S, M, L = p.slots, p.mat_size, p.n_layers
    d_ord = torch.empty(len(frames) * S * 32, dtype=torch.uint8, device=dev)
    d_multi = torch.empty(len(frames) * L * M * M, dtype=torch.uint8, device=dev)
    d_single = torch.empty(len(frames) * M * M, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.process_device(len(frames), d_in.data_ptr(), offs, d_ord.data_ptr(), d_multi.data_ptr(), d_single.data_ptr())
    ctx.synchronize()
    assert d_ord.cpu().numpy().tobytes() == ordered.tobytes()
    assert d_multi.cpu().numpy().tobytes() == multi.tobytes()
    assert d_single.cpu().numpy().tobytes() == single.tobytes()


# ---- float max-height BEV of batch_cloud_manip / cloud_manip ----
@pytest.mark.parametrize("sensor", ["HDL_64E", "HDL_32E"])
def test_float_bev(ctxs, sensor):
    p, ctx = ctxs(sensor)
    sp = orc.sensor_from_params(p)
    marked, _, _ = orc.mark_ground(sp, orc.order_cloud(sp, synth.sweep(p, 21)))
    for cloud in (marked, synth.adversarial(p, 40000, 17, True)):
        for interval, skip in ((1.0, True), (1.0, False), (2.0, True), (0.5, False)):
            got = ctx.float_bev(cloud, interval, skip)
            want = orc.float_bev(cloud, interval, skip)
            assert got.shape == want.shape
            # north_star allows 1e-5 on float height channels; a max has no rounding, so demand equality
            assert got.tobytes() == want.tobytes(), (interval, skip, np.abs(got - want).max())
    assert ctx.float_bev(marked, 1.0, True).shape == (201, 201)


@pytest.mark.parametrize("geom", [(128, 2048, 100, 0.25), (16, 1800, 10, 1.0), (64, 4000, 50, 0.5), (128, 1024, 126, 0.5)])
def test_other_sensor_geometries(geom):
    """Sensors outside the reference's table (128 beams, 16 beams, 4000 columns, every row but two ground-tested): the
    kernels take N_SCAN / Horizon_SCAN / GROUND_UPPER_SCAN as parameters; (G + 1) * strips stays within the 1024
    candidate segments a frame may have."""
    n, h, g, res = geom
    p = bev_amd.params_for_sensor("HDL_64E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan, p.height_res = n, h, g, res
    sp = orc.sensor_from_params(p)
    frames = [synth.sweep(p, 40 + i, keep=0.95, n_dup=3000) for i in range(2)] + [synth.adversarial(p, 50000, 8, True)]
    ctx = bev_amd.BevContext(p, device=0, max_batch=4, max_points=max(len(f) for f in frames))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes(), (geom, i)
            assert np.array_equal(gm[i], o_gm) and np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (geom, i)
    finally:
        ctx.close()
    # one row more of ground testing than the kernels' segment table holds is refused, not mis-computed
    p.n_scan, p.horizon_scan, p.ground_upper_scan = 200, 2048, 150
    with pytest.raises(bev_amd.BevError):
        bev_amd.BevContext(p, device=0, max_batch=1, max_points=1000)


def test_transform_cloud(ctxs):
    """cloud_manip's rigid transform (CloudManip.cpp:119-128) on the device, bit-identical to the oracle."""
    p, ctx = ctxs("HDL_64E")
    cases = [(0, 0, 0, 0), (1.5, -2.25, 0.125, 30), (-3, 4, 1, -45.5), (10, 20, -1, 180), (0.1, 0.2, 0.3, 359.9)]
    for seed, (tx, ty, tz, yaw) in enumerate(cases):
        m = bev_amd.yaw_translate_matrix(tx, ty, tz, yaw)
        assert np.array_equal(m, orc.yaw_translate_matrix(tx, ty, tz, yaw))
        for cloud in (synth.sweep(p, seed), synth.adversarial(p, 150001, seed, nonfinite=True)):
            got, want = ctx.transform_cloud(cloud, m), orc.transform_cloud(cloud, m)
            nan = np.isnan(want["x"]) | np.isnan(want["y"]) | np.isnan(want["z"])
            assert got[~nan].tobytes() == want[~nan].tobytes(), seed
            for f in ("x", "y", "z"):  # NaN results: NaN on both sides (payload bits are not specified)
                assert np.array_equal(np.isnan(got[f]), np.isnan(want[f]))
            assert got[nan][["intensity", "row", "col", "t", "label"]].tobytes() == want[nan][["intensity", "row", "col", "t", "label"]].tobytes()
    assert len(ctx.transform_cloud(np.empty(0, bev_amd.POINT_DTYPE), m)) == 0
    # the transformed cloud feeds the float raster like cloud_manip does (:136-137)
    out = ctx.transform_cloud(synth.sweep(p, 3), m)
    assert ctx.float_bev(out, 1.0, False).tobytes() == orc.float_bev(orc.transform_cloud(synth.sweep(p, 3), m), 1.0, False).tobytes()


# ---- range-image projection of raw XYZI returns (atan2f restated on the device) ----
@pytest.mark.parametrize("kind", [0, 1])
def test_projection_matches_oracle(ctxs, kind):
    from projection_data import raw_returns

    p, ctx = ctxs("OS1_64" if kind == 0 else "HDL_32E", 4, 300000)
    for seed in (0, 1):
        pts = raw_returns(250_000, seed)
        xyzi = pts if kind == 0 else np.ascontiguousarray(pts.T)
        got = ctx.project_xyzi(kind, xyzi)
        want = orc.project(kind, xyzi)
        assert got.tobytes() == want.tobytes(), np.flatnonzero((got["col"] != want["col"]) | (got["row"] != want["row"]))[:8]
    # projected returns feed the hot path like any selector output
    sp = orc.sensor_from_params(p)
    cloud = ctx.project_xyzi(kind, xyzi)[:65536 if kind == 0 else 120000]
    ordered, multi, single, gm = ctx.process_batch([cloud], want_ground_mat=True)
    o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, cloud)
    assert ordered[0].tobytes() == o_ord.tobytes() and np.array_equal(gm[0], o_gm)
    assert np.array_equal(multi[0], o_multi) and np.array_equal(single[0], o_single)


def test_kitti_projection_matches_oracle(ctxs):
    """SURVEY §8(f) N3, KITTI: ring index = counter of azimuth zero crossings (a sequential loop in the reference)."""
    from projection_data import KITTI_VARIANTS, kitti_returns

    p, ctx = ctxs("HDL_64E", 2, 600000)
    sp = orc.sensor_from_params(p)
    for variant in KITTI_VARIANTS:
        for seed in (0, 1):
            xyzi = kitti_returns(seed, variant)
            got, want = ctx.project_xyzi(2, xyzi), orc.project(2, xyzi)
            assert got.shape == (64 * 2083,)
            assert got.tobytes() == want.tobytes(), (variant, seed)
    xyzi = kitti_returns(3, "noisy_seam")
    for n in (0, 1, 2, 255, 256, 257, 1250, 1251, 5000):
        assert ctx.project_xyzi(2, xyzi[:n]).tobytes() == orc.project(2, xyzi[:n]).tobytes(), n
    big = np.concatenate([kitti_returns(s, "sweep") for s in range(4)])   # 480 k returns: rings run out after 64
    assert ctx.project_xyzi(2, big).tobytes() == orc.project(2, big).tobytes()
    # the structured cloud feeds the hot path like any selector output (empty slots are all-zero points at slot 0)
    cloud = ctx.project_xyzi(2, kitti_returns(9, "sweep"))
    ordered, multi, single, gm = ctx.process_batch([cloud], want_ground_mat=True)
    o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, cloud)
    assert ordered[0].tobytes() == o_ord.tobytes() and np.array_equal(gm[0], o_gm)
    assert np.array_equal(multi[0], o_multi) and np.array_equal(single[0], o_single)


@pytest.mark.parametrize("interval", [0.5, 2.0])
def test_other_grid_intervals(interval):
    """computeAndSave{Multi,Single}Bev take `interval` (default 1.0f, BatchMultiBevGen.cpp:261 / :331): 448 x 448 and
    112 x 112 rasters.  The raster kernel splits a frame into more x-bands when a band's LDS planes would not fit."""
    p = bev_amd.params_for_sensor("HDL_64E")
    p.interval = interval
    ctx = bev_amd.BevContext(p, device=0, max_batch=4, max_points=200000)
    try:
        sp = orc.sensor_from_params(p)
        frames = [synth.sweep(p, 3), synth.adversarial(p, 60000, 2, True), synth.firing_order(p, 1)]
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        M = int(224 / interval)
        assert multi.shape[1:] == (24, M, M) and single.shape[1:] == (M, M)
        for i, pts in enumerate(frames):
            o_ord, o_gm, _, _ = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm)
            assert np.array_equal(multi[i], orc.multi_bev(sp, o_ord, interval).reshape(24, M, M)), i
            assert np.array_equal(single[i], orc.single_bev(o_ord, interval).reshape(M, M)), i
        # the per-function entry points on an arbitrary cloud
        assert np.array_equal(ctx.multi_bev(frames[1]).reshape(24, M, M), orc.multi_bev(sp, frames[1], interval).reshape(24, M, M))
        assert np.array_equal(ctx.single_bev(frames[1]).reshape(M, M), orc.single_bev(frames[1], interval).reshape(M, M))
    finally:
        ctx.close()
