"""GPU: k_tile — the walk over structured clouds (and bev_mark_ground's identity walk) as row-block TILES instead of a row
loop: a wave owns 59 columns x 8 rows, all its records requested at once, no barrier per row.  Opt-in
(BEV_TILE=1 in the environment of bev_create): measured at parity with the row loop, not ahead of it.  Same outputs."""
import os

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_on(monkeypatch):
    monkeypatch.setenv("BEV_TILE", "1")


@pytest.mark.parametrize("sensor", ["HDL_64E", "OS1_64", "HDL_32E"])
def test_tiles_give_the_oracles_outputs(sensor):
    p = bev_amd.params_for_sensor(sensor)
    sp = orc.sensor_from_params(p)
    rng = np.random.default_rng(3)
    noisy = synth.structured(p, 15, 0.9)
    real = noisy["label"] == -2
    noisy["intensity"][real & (rng.random(len(noisy)) < 0.2)] = -1.0   # phase A's fallbacks all over the frame
    frames = [synth.structured(p, 10, 0.98), synth.structured(p, 11, 1.0), synth.structured(p, 12, 0.5),
              synth.structured(p, 13, 0.9, kitti_intensity=True), synth.sweep(p, 14), noisy, synth.firing_order(p, 16)]
    ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=max(len(f) for f in frames))
    try:
        ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
        assert [int(m) for m in ctx.frame_info(0, len(frames))[:, 1]] == [3, 3, 3, 3, 1, 3, 4]
        for i, pts in enumerate(frames):
            o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
            assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), i
            assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), i
        # bev_mark_ground: the identity source through the tiles
        o_ord, o_gm, _, _ = orc.process_frame(sp, frames[0])
        got, got_gm = ctx.mark_ground(orc.order_cloud(sp, frames[0]))
        assert got.tobytes() == o_ord.tobytes() and np.array_equal(got_gm, o_gm)
    finally:
        ctx.close()


def test_odd_geometries_and_hidden_defects():
    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    for n, h, g in [(33, 505, 20), (8, 300, 5), (96, 700, 60), (17, 236, 9), (13, 59, 7)]:
        p.n_scan, p.horizon_scan, p.ground_upper_scan = n, h, g
        sp = orc.sensor_from_params(p)
        bad = synth.structured(p, 30 + n, 1.0)
        k = (n * h) // 2 + 5
        while k % 63 in (0, 1):
            k += 1
        bad[k]["col"] = (int(bad[k]["col"]) + 3) % h          # hidden from the samples: caught by the tile that owns it
        frames = [synth.structured(p, 20 + n, 0.9), synth.structured(p, 21 + n, 1.0), bad]
        ctx = bev_amd.BevContext(p, device=0, max_batch=16, max_points=n * h)
        try:
            ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
            assert [int(m) for m in ctx.frame_info(0, 3)[:, 1]] == [3, 3, 2], (n, h, g)
            for i, pts in enumerate(frames):
                o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
                assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(gm[i], o_gm), (n, h, g, i)
                assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single), (n, h, g, i)
        finally:
            ctx.close()


def test_a_thousand_structured_frames_through_tiles_and_tiny_code_lists():
    """the tiles of a strip add their counts to the strip's code lists with global atomics: at BASELINE scale, pipelined,
    and once more with lists of 64 entries (every band falls back to the ordered cloud)"""
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    import torch

    p = bev_amd.params_for_sensor("HDL_64E")
    n = 1000
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda f: synth.structured(p, f, keep=0.98), range(n)))
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)
    dev = torch.device("cuda:0")
    offs = np.arange(n + 1, dtype=np.uint64) * S
    d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
    digests = []
    for cap in (None, "64"):
        if cap:
            os.environ["BEV_CODE_CAP"] = cap
        try:
            ctx = bev_amd.BevContext(p, device=0, max_batch=500, max_points=S)
        finally:
            os.environ.pop("BEV_CODE_CAP", None)
        outs = [torch.zeros(n * k, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
        for _ in range(2):
            ctx.process_device(n, d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
        ctx.synchronize()
        ords, multis, singles = (o.cpu().numpy() for o in outs)
        ctx.close()
        h = hashlib.sha256()
        for a in (ords, multis, singles):
            h.update(a.tobytes())
        digests.append(h.hexdigest())
        if cap is None:
            def check(i):
                o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
                return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes()
                        and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                        and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())
            with ThreadPoolExecutor(16) as ex:
                bad = [i for i, ok in enumerate(ex.map(check, range(n))) if not ok]
            assert not bad, f"{len(bad)} of {n} frames differ from the oracle, first: {bad[:8]}"
    assert digests[0] == digests[1]
