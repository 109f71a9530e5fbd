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
COLMAJOR, REDO, GENERAL = 4, 2, 0


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
        assert int(info[i, 0]) == p.slots and int(info[i, 2]) == p.slots and int(info[i, 3]) == 0


def test_defects_hidden_from_the_samples_are_caught_and_redone():
    p = bev_amd.params_for_sensor("OS1_64")
    base = _with_invalid(synth.firing_order(p, 10), 10)

    def at(i):   # a position the probe does not look at (samples: multiples of 63 and their successors)
        while i % 63 in (0, 1):
            i += 1
        return i

    wrong_beam = base.copy()
    wrong_beam[at(30000)]["row"] = (int(base[at(30000)]["row"]) + 1) % 64
    far_column = base.copy()              # nine columns past its firing: more than the walk's window allows for
    far_column[at(20000)]["col"] = at(20000) // 64 + 9
    behind = base.copy()                  # a column BEFORE its firing's number
    k = at(40000)
    behind[k]["col"] = max(0, k // 64 - 3)
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
