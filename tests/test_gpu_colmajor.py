"""GPU: clouds in firing order — what the reference's MulRan selector writes (MulranPointCloudSelect.cpp:112-130): S
returns, position k holding beam k % N of firing k / N, its column computed from the azimuth (the firing's number plus a
small displacement; 1024 = Horizon_SCAN after rounding is dropped by getOrderedCloud's bounds test,
BatchMultiBevGen.cpp:109-111).  k_probe recognises the layout from its samples; the walk fetches a strip's firings band by
band, settles the last writer of every slot (:112-115) in an LDS index row and checks every record; a frame that fails is
redone the general way.  Whatever route a frame takes, its outputs must equal the oracle's."""
import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu
COLMAJOR, COLMAJOR_GEN, REDO, GENERAL = 4, 5, 2, 0   # 4: the plain sweep (round 4's walk), 5: any phase / direction / stagger / no-returns


def _run(p, frames, max_batch=16):
    ctx = bev_amd.BevContext(p, device=0, max_batch=max_batch, max_points=max(8, max(len(f) for f in frames)))
    try:
        assert len(frames) < max_batch // 2   # one chunk, one sub-batch: frame_info covers every frame
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        info = ctx.frame_info(0, len(frames))
    finally:
        ctx.close()
    sp = orc.sensor_from_params(p)
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud / labels differ"
        assert np.array_equal(gm[i], o_gm), i
        assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
    return [int(m) for m in info[:, 1]], info


def _with_invalid(f, seed, share=0.15):
    """a share of the returns marked invalid (intensity -1): phase A's fallbacks — (c + 2) % H, flat c - 2, row - 2,
    BatchMultiBevGen.cpp:146-160 — are taken all over the frame, at the row ends and strip edges too"""
    f = f.copy()
    rng = np.random.default_rng(seed)
    f["intensity"][rng.random(len(f)) < share] = -1.0
    return f


def _displaced(p, fid, max_disp, seed):
    """firing order with columns up to max_disp past the firing's number: several firings fight over a column (the later
    one wins), columns stay empty, the last firings run out of range"""
    f = synth.firing_order(p, fid)
    rng = np.random.default_rng(seed)
    fire = np.arange(len(f)) // p.n_scan
    f["col"] = (fire + rng.integers(0, max_disp + 1, len(f))).astype(np.uint16)
    return f


@pytest.mark.parametrize("sensor", ["OS1_64", "HDL_64E", "HDL_32E"])
def test_firing_order_is_read_in_place(sensor):
    p = bev_amd.params_for_sensor(sensor)
    frames = [synth.firing_order(p, 1), _with_invalid(synth.firing_order(p, 2), 2), _displaced(p, 3, 8, 3),
              _with_invalid(_displaced(p, 4, 5, 4), 4, 0.3), _displaced(p, 5, 0, 5)]
    modes, info = _run(p, frames)
    assert modes == [COLMAJOR] * 5, info
    for i in range(5):
        assert int(info[i, 0]) == p.slots and int(info[i, 2]) == p.slots and (int(info[i, 3]) & 1) == 0   # (bit 1: a wrap-around halo fell back on column 0 — compared by k_verdict)


def test_defects_hidden_from_the_samples_are_caught_and_redone():
    p = bev_amd.params_for_sensor("OS1_64")
    base = _with_invalid(synth.firing_order(p, 10), 10)

    def at(i):   # a position the probe does not look at (samples: multiples of 63 and their successors)
        while i % 63 in (0, 1):
            i += 1
        return i

    wrong_beam = base.copy()
    wrong_beam[at(30000)]["row"] = (int(base[at(30000)]["row"]) + 1) % 64
    far_column = base.copy()              # twenty columns past its firing: more than the walk's window (13 columns from the row's base) allows for
    far_column[at(20000)]["col"] = at(20000) // 64 + 20
    behind = base.copy()                  # a column well BEFORE its firing's number
    k = at(40000)
    behind[k]["col"] = max(1, k // 64 - 9)
    swapped = base.copy()                 # two beams of one firing swapped
    a = at(50000)
    a -= a % 64
    a += 5
    swapped[[a, a + 1]] = swapped[[a + 1, a]]
    out_of_range_row = base.copy()
    out_of_range_row[at(10000)]["row"] = 64
    modes, info = _run(p, [wrong_beam, far_column, behind, swapped, out_of_range_row, base])
    assert modes == [REDO, REDO, REDO, REDO, REDO, COLMAJOR], info


def test_odd_sensors_and_mixed_sub_batches():
    p = bev_amd.params_for_sensor("HDL_32E")
    for n, h, g in [(33, 505, 20), (8, 300, 5), (96, 700, 60), (17, 236, 9), (64, 2083, 50)]:
        p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
        frames = [_with_invalid(_displaced(p, 20 + n, d, n + d), n, 0.2) for d in (1, 8)]
        frames.append(synth.sweep(p, 7, n_dup=50))
        frames.append(synth.structured(p, 8, 0.9))
        modes, info = _run(p, frames)
        assert modes[:2] == [COLMAJOR, COLMAJOR] and modes[3] == 3, (n, h, g, info)


# ---- round 5: what mulran_point_cloud_select writes for REAL sweeps (MulranPointCloudSelect.cpp:112-130): any start
# azimuth, either direction of rotation, the four staggered laser columns, no-return records (x = y = 0 -> atan2(0, 0) = 0
# -> column 0 of the record's row, the last one in input order winning the slot)
@pytest.mark.parametrize("sensor", ["OS1_64", "HDL_32E", "HDL_64E"])
def test_real_mulran_sweeps_are_read_in_place(sensor):
    p = bev_amd.params_for_sensor(sensor)
    frames = [synth.firing_real(p, 11), synth.firing_real(p, 12, noret=0.0), synth.firing_real(p, 13, noret=0.3),
              synth.firing_real(p, 14, phase=0, direction=1, stagger=0.0), synth.firing_real(p, 15, phase=p.horizon_scan - 1, direction=-1),
              synth.firing_real(p, 16, noret=0.002), synth.firing_real(p, 17, phase=5, direction=-1, stagger=0.5)]
    modes, info = _run(p, frames)
    # (frame 14: phase 0, forward, no stagger, but 3 % no-return records: not the plain sweep either)
    assert modes == [COLMAJOR_GEN] * len(frames), info


def test_real_sweeps_with_invalid_returns_match_whatever_the_route():
    """intensity -1 in column H - 2 makes phase A fall back on column 0 of the row above (BatchMultiBevGen.cpp:146-149) —
    where a no-return record of ANY firing may sit, which only strip 0 hears of: the strip with the wrap-around halo says
    what it took, k_verdict compares, the frame is redone when they differ.  Equal to the oracle either way."""
    p = bev_amd.params_for_sensor("OS1_64")
    frames = [_with_invalid(synth.firing_real(p, 21 + i, noret=nr), 21 + i, share) for i, (nr, share) in
              enumerate([(0.03, 0.15), (0.0, 0.3), (0.3, 0.05), (0.01, 0.5), (0.1, 0.02)])]
    modes, info = _run(p, frames)
    assert set(modes) <= {COLMAJOR_GEN, REDO}, info
    assert modes[1] == COLMAJOR_GEN, info      # no no-return records: nothing to disagree about


def test_real_sweep_defects_are_caught():
    """a return far from where its firing's returns lie (and not in column 0), a wrong beam: the frame goes the general
    way after the walk has seen it — the probe's samples do not"""
    p = bev_amd.params_for_sensor("OS1_64")
    good = synth.firing_real(p, 31)
    a = good.copy()
    k = 64 * 400 + 7            # firing 400, beam 7
    a["col"][k] = (int(a["col"][k]) + 200) % p.horizon_scan or 1
    b2 = good.copy()
    b2["row"][64 * 500 + 9] = 10
    modes, info = _run(p, [a, b2, good])
    assert modes[2] == COLMAJOR_GEN and modes[0] in (REDO, GENERAL) and modes[1] in (REDO, GENERAL), info


# ---- round 6 (advisor, round 5): deterministic cases for the branches the soaks reached only by chance
def _unsampled(k):
    """a position k_probe does not look at (frames of exactly S records: every 63rd and its successor)"""
    while k % 63 in (0, 1):
        k += 1
    return k


def test_a_plain_sweep_with_one_return_nine_columns_off_is_redone():
    """the plain sweep's walk (mode 4) takes column = firing + 0 .. 8 (kPlainDisp); a single return at + 9 hidden from the
    probe's samples: the probe says mode 4, the walk's check fails, the frame is redone — same outputs"""
    p = bev_amd.params_for_sensor("OS1_64")
    f = _displaced(p, 40, 8, 40)
    k = _unsampled(64 * 300 + 11)
    f["col"][k] = k // 64 + 9
    modes, info = _run(p, [f, _displaced(p, 41, 8, 41)])
    assert modes == [REDO, COLMAJOR], info


def test_a_stray_no_return_record_that_would_win_column_zero_is_caught():
    """A real sweep WITHOUT no-return records among the probe's samples is walked without the strips' talk (kCfQuiet).  One
    no-return record (x = y = z = 0 -> column 0 of its row, MulranPointCloudSelect.cpp:123-125) at a position the probe does
    not sample, in a firing the LAST strip owns, later in the input than the return strip 0 put into column 0 of that row:
    the last writer wins (BatchMultiBevGen.cpp:112-115), so it must end up in column 0 — the strip that owns it leaves it in
    cm_sync (kInfoCmStray), k_verdict sees that it beats strip 0's, the frame is redone.  The same record EARLIER than strip
    0's winner changes nothing: the frame stays on route 5."""
    p = bev_amd.params_for_sensor("OS1_64")
    base = synth.firing_real(p, 51, noret=0.0, phase=0, direction=1, stagger=1.0)

    def with_noret(firing, beam):
        f = base.copy()
        k = _unsampled(64 * firing + beam)
        assert k // 64 == firing
        for name in ("x", "y", "z"):
            f[name][k] = 0.0
        f["col"][k] = 0
        return f

    # beam 23 sits 9 columns BEHIND its firing's azimuth (beam mod 4 == 3): row 23's column 0 holds the return of firing 9, and a
    # no-return record of firing 1000 — the last strip's: columns 944 ... — is later in the input; firing 0 is strip 0's own
    late, early = with_noret(1000, 23), with_noret(0, 21)
    modes, info = _run(p, [late, early, base])
    assert modes[0] == REDO and modes[2] == COLMAJOR_GEN and modes[1] in (COLMAJOR_GEN, REDO), info


@pytest.mark.parametrize("n,h,g", [(31, 300, 20), (16, 245, 8), (5, 473, 2), (33, 300, 20), (7, 473, 3)])
def test_real_sweeps_on_small_and_odd_sensors(n, h, g):
    """two strips and an odd number of rows (a last band of one row; backward rotation puts its second row's pieces past the
    frame's end), 237 < H < 252 (own_at != 16: the strips' counted ranges tile a short circle), a last strip of one column.
    (33 rows: at every 63rd record and its successor k_probe saw beams 0 and 1 mod 3 of a 33-beam sensor, never beam 2 mod 3 —
    rows without a sample took another row's base and staggered beams failed the walk's checks: frames of exactly S records
    from a sensor whose beam count shares a factor with 63 are sampled at every 61st instead.)"""
    p = bev_amd.params_for_sensor("HDL_32E")
    p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
    frames = [synth.firing_real(p, 60, noret=0.05, direction=-1), synth.firing_real(p, 61, noret=0.0, direction=1),
              _with_invalid(synth.firing_real(p, 62, noret=0.2, direction=-1, stagger=0.5), 62, 0.3),
              synth.firing_real(p, 63, noret=0.01, phase=h - 1, direction=1)]
    modes, info = _run(p, frames)
    assert set(modes) <= {COLMAJOR_GEN, REDO}, (n, h, g, info)
    assert COLMAJOR_GEN in modes, (n, h, g, info)
