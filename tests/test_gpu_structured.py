"""GPU: structured clouds — what the reference's KITTI selector writes (KittiPointCloudSelect.cpp:206-207,240): exactly
S records, record i being the point of slot i or an all-zero record (row = col = 0).  getOrderedCloud
(BatchMultiBevGen.cpp:98,102-116) then is the identity except for slot 0, where every all-zero record lands: slot 0 ends
up all-zero iff a record after the first is all-zero.  k_probe recognises the layout from its samples (and guesses
whether an empty record exists), the walk reads the records in place, once, and checks every one; a frame that fails is
redone the general way.  Whatever route a frame takes, its outputs must equal the oracle's."""
import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu
STRUCTURED, REDO, GENERAL, STREAM, COLMAJOR = 3, 2, 0, 1, 4


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


@pytest.mark.parametrize("sensor", ["HDL_64E", "HDL_32E", "OS1_64"])
def test_structured_clouds_are_read_in_place(sensor):
    p = bev_amd.params_for_sensor(sensor)
    frames = [synth.structured(p, 10, 0.98), synth.structured(p, 11, 1.0), synth.structured(p, 12, 0.5),
              synth.structured(p, 13, 0.9, kitti_intensity=True), synth.structured(p, 14, 0.02)]
    modes, info = _run(p, frames)
    assert modes == [STRUCTURED] * 5, info
    for i in range(5):
        assert int(info[i, 0]) == p.slots and int(info[i, 2]) == p.slots and (int(info[i, 3]) & 1) == 0


def test_slot_zero_takes_the_last_all_zero_record():
    """record 0 holds the real point of slot (0, 0); any later all-zero record overwrites slot 0 (last writer wins,
    BatchMultiBevGen.cpp:112-115); with no empty record slot 0 keeps its point — and rasterises"""
    p = bev_amd.params_for_sensor("HDL_32E")
    full = synth.structured(p, 20, 1.0)
    assert full[0]["label"] == -2
    full[0]["z"] = 1.0                      # a non-ground point inside the image: its BEV cell must appear / disappear
    one_hole = full.copy()
    one_hole[63 * 100 + 1] = np.zeros(1, bev_amd.POINT_DTYPE)[0]   # an empty record the probe samples (successor of a sample)
    zero_first = full.copy()
    zero_first[0] = np.zeros(1, bev_amd.POINT_DTYPE)[0]            # only record 0 empty: not "a record after the first"
    holes = synth.structured(p, 21, 0.9)
    holes[0] = full[0]
    modes, info = _run(p, [full, one_hole, zero_first, holes])
    assert modes == [STRUCTURED] * 4, info
    assert [(int(x) >> 1) & 1 for x in info[:, 3]] == [0, 1, 0, 1]   # an all-zero record after the first was seen


def test_defects_hidden_from_the_samples_are_caught_and_redone():
    p = bev_amd.params_for_sensor("HDL_64E")
    H = p.horizon_scan
    full = synth.structured(p, 30, 1.0)
    holes = synth.structured(p, 31, 0.97)
    zero = np.zeros(1, bev_amd.POINT_DTYPE)[0]

    def at(i):   # a position the probe does not look at (samples: multiples of 63 and their successors)
        while i % 63 in (0, 1):
            i += 1
        return i

    hidden_hole = full.copy()               # the guess "no empty record" is wrong: slot 0 must become all-zero
    hidden_hole[at(70000)] = zero
    wrong_slot = holes.copy()               # a point that claims another slot: it must move there
    i = at(50000)
    wrong_slot[i] = full[i]
    wrong_slot[i]["col"] = (int(full[i]["col"]) + 7) % H
    claims_zero = holes.copy()              # row = col = 0 but not empty: lands in slot 0 with its contents
    j = at(90000)
    claims_zero[j] = full[j]
    claims_zero[j]["row"], claims_zero[j]["col"] = 0, 0
    out_of_range = holes.copy()
    k = at(20000)
    out_of_range[k] = full[k]
    out_of_range[k]["row"] = 64
    pad_only = holes.copy()                 # an "empty" record with a non-zero padding word is not all-zero
    m = at(110000)
    pad_only[m] = zero
    pad_only[m]["_pad0"] = 1.0
    modes, info = _run(p, [hidden_hole, wrong_slot, claims_zero, out_of_range, pad_only, holes])
    assert modes == [REDO, REDO, REDO, REDO, REDO, STRUCTURED], info


def test_structured_sorted_and_unordered_frames_in_one_sub_batch():
    p = bev_amd.params_for_sensor("OS1_64")
    frames = [synth.structured(p, 40, 0.95), synth.sweep(p, 41, n_dup=2000), synth.firing_order(p, 42),
              synth.structured(p, 43, 1.0), np.empty(0, bev_amd.POINT_DTYPE), synth.sweep(p, 44, keep=1.0, n_dup=0)]
    modes, info = _run(p, frames)
    # (a full sorted sweep without appended points IS a structured cloud; firing order has S points too but is not one)
    assert modes == [STRUCTURED, STREAM, COLMAJOR, STRUCTURED, GENERAL, STRUCTURED], info


def test_the_kitti_projection_feeds_the_structured_route():
    """bev_project_xyzi(KITTI) writes what the selector writes; every real point carries intensity -1 (:238), so phase A
    marks nothing and every point rasterises"""
    from projection_data import kitti_returns

    p = bev_amd.params_for_sensor("HDL_64E")
    ctx = bev_amd.BevContext(p, device=0, max_batch=2, max_points=600000)
    try:
        clouds = [ctx.project_xyzi(2, kitti_returns(s, v)) for s, v in [(9, "sweep"), (2, "noisy_seam"), (4, "short_rings")]]
    finally:
        ctx.close()
    modes, info = _run(p, clouds)
    assert modes == [STRUCTURED] * 3, info


def test_knob_off_and_sensors_the_in_place_source_cannot_take():
    import os
    p = bev_amd.params_for_sensor("HDL_32E")
    old = os.environ.get("BEV_STREAM")
    os.environ["BEV_STREAM"] = "0"
    try:
        modes, _ = _run(p, [synth.structured(p, 50, 0.9)])
        assert modes == [GENERAL]
    finally:
        if old is None:
            del os.environ["BEV_STREAM"]
        else:
            os.environ["BEV_STREAM"] = old
    # 96 rows: more than the in-place source keeps estimates for (no tail lists are allocated) — structured clouds do
    # not need them
    p.n_scan, p.horizon_scan, p.ground_upper_scan = 96, 700, 60
    modes, _ = _run(p, [synth.structured(p, 51, 0.9), synth.sweep(p, 52, n_dup=100)])
    assert modes == [STRUCTURED, GENERAL]


def test_a_walk_that_was_not_launched_costs_time_not_results(monkeypatch):
    """The host launches the walk of a mode only while the workspace set's last BEV_MODE_TTL looks at k_verdict's word
    (mapped host memory: the modes k_probe gave the set's last finished sub-batch) showed the mode.  With a TTL of 1
    (round 4's rule) a structured cloud that arrives after sweeps finds its walk not launched: its count fails, it is
    redone the general way (mode 2) — same outputs — and the next call sees the mode again."""
    monkeypatch.setenv("BEV_MODE_TTL", "1")
    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    calls = [[synth.sweep(p, 60 + i, n_dup=300) for i in range(3)], [synth.structured(p, 63 + i, 0.9) for i in range(3)],
             [synth.structured(p, 66 + i, 0.9) for i in range(3)], [synth.firing_order(p, 70), synth.structured(p, 71, 0.9)],
             [synth.firing_order(p, 72), synth.sweep(p, 73, n_dup=300)]]
    want_modes = [[STREAM] * 3, [REDO] * 3, [STRUCTURED] * 3, [REDO, STRUCTURED], [COLMAJOR, REDO]]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for c in calls for f in c))
    try:
        for frames, want in zip(calls, want_modes):
            ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
            modes = [int(m) for m in ctx.frame_info(0, len(frames))[:, 1]]
            for i, pts in enumerate(frames):
                o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
                assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), (want, i)
                assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (want, i)
            assert modes == want, (modes, want)
    finally:
        ctx.close()


def test_alternating_layouts_are_read_in_place_after_warm_up():
    """Round 5 (advisor, round 4): the mode hint is sticky (default TTL 8 looks per workspace set).  Calls that alternate
    layouts — sweeps, structured clouds, firing order, one after the other — are all read in place: no frame is redone."""
    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    makers = [lambda i: synth.sweep(p, 200 + i, n_dup=300), lambda i: synth.structured(p, 200 + i, 0.9),
              lambda i: synth.firing_order(p, 200 + i)]
    want = [STREAM, STRUCTURED, COLMAJOR]
    frames0 = [m(0) for m in makers]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames0))
    try:
        for rnd in range(12):
            k = rnd % 3
            frames = [makers[k](3 * rnd + j) for j in range(2)]
            ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
            modes = [int(m) for m in ctx.frame_info(0, len(frames))[:, 1]]
            assert modes == [want[k]] * 2, (rnd, modes)
            for i, pts in enumerate(frames):
                o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
                assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), (rnd, i)
                assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (rnd, i)
    finally:
        ctx.close()


def test_layout_hint_right_and_wrong():
    """bev_set_layout_hint (round 6): frames of exactly S records are TAKEN for what the caller says they are — k_probe does
    not look — and verified by the walk like any guess.  Right hint: read in place (mode 3 / 4).  Wrong hint (structured
    clouds announced as firing order, firing order announced as structured, a complete structured cloud whose slot 0 the
    hint's guess gets wrong): every frame is redone the general way (mode 2) — the same bytes as the oracle's.  Frames of
    another size are probed as ever.  bev_project_xyzi(KITTI) sets the hint by itself."""
    p = bev_amd.params_for_sensor("HDL_32E")
    structured = [synth.structured(p, 300 + i, 0.9) for i in range(3)]
    firing = [synth.firing_order(p, 310 + i) for i in range(3)]
    sweeps = [synth.sweep(p, 320 + i, n_dup=200) for i in range(2)]
    sp = orc.sensor_from_params(p)
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in structured + firing + sweeps))

    def run(frames, want):
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        modes = [int(m) for m in ctx.frame_info(0, len(frames))[:, 1]]
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), (want, i)
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (want, i)
        assert modes == want, (modes, want)

    try:
        ctx.set_layout_hint(bev_amd.LAYOUT_STRUCTURED)
        run(structured + sweeps, [STRUCTURED] * 3 + [STREAM] * 2)      # right; the sweeps (another size) are probed
        run(firing, [REDO] * 3)                                          # wrong: redone
        run([synth.structured(p, 330, 1.0)], [REDO])                    # a cloud without a single dropped return: slot 0 guessed wrong
        ctx.set_layout_hint(bev_amd.LAYOUT_FIRING_ORDER)
        run(firing, [COLMAJOR] * 3)
        run(structured, [REDO] * 3)
        ctx.set_layout_hint(bev_amd.LAYOUT_UNKNOWN)
        run(structured + firing, [STRUCTURED] * 3 + [COLMAJOR] * 3)
    finally:
        ctx.close()
