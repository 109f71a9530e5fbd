"""CPU: the synthetic-frame generator is bit-reproducible (integer hashing + IEEE
basic ops only), so the GPU box regenerates exactly the frames the fixtures were
made from."""
import hashlib

import numpy as np

import bev_amd
from bev_amd import synth



def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_sweep_is_deterministic_and_plausible():
    p = bev_amd.params_for_sensor("HDL_64E")
    a, b = synth.sweep(p, 5), synth.sweep(p, 5)
    assert a.tobytes() == b.tobytes()
    assert synth.sweep(p, 6).tobytes() != a.tobytes()
    n = len(a)
    assert abs(n - (0.98 * p.slots + 5000)) < 600            # ~135.6k points
    assert 0.04 < (a["intensity"] == -1).mean() < 0.06        # 5 % no-return markers
    assert (a["label"] == -2).all() and (a["t"] == 5).all()
    body = a[:-5000]
    slot = body["row"].astype(np.int64) * p.horizon_scan + body["col"]
    assert (np.diff(slot) > 0).all()                          # row-major, unique
    assert np.isfinite(a["x"]).all() and np.abs(a["x"]).max() <= 80.0
    down = a[a["row"] > 20]
    assert np.median(down["z"]) < -1.5                        # ground at -1.73 m


def test_pinned_hashes():
    from golden_data import SYNTH_SHA256

    for sensor, want in SYNTH_SHA256.items():
        p = bev_amd.params_for_sensor(sensor)
        assert _sha(synth.sweep(p, 0)) == want["sweep0"], sensor
        assert _sha(synth.firing_order(p, 1)) == want["firing1"], sensor
        assert _sha(synth.adversarial(p, 20000, 3, True)) == want["adv3"], sensor


def test_firing_order_has_overflow_column():
    p = bev_amd.params_for_sensor("OS1_64")
    a = synth.firing_order(p, 0)
    assert len(a) == p.slots
    assert (a["row"] == np.arange(len(a)) % 64).all()
    assert (a["col"] == 1024).any() and (a["col"] <= 1024).all()   # MulranPointCloudSelect.cpp:125 can give 1024
    assert (a["intensity"] != -1).all()


def test_sweep_unique_config1():
    p = bev_amd.params_for_sensor("HDL_32E")
    a = synth.sweep_unique(p, 0, 16384)
    assert len(a) == 16384
    slot = a["row"].astype(np.int64) * p.horizon_scan + a["col"]
    assert len(np.unique(slot)) == 16384
