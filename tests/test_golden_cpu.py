"""CPU: the oracle against the committed fixtures (tests/golden_data.py,
tests/golden/tiny_hdl32.npz).  The fixtures are this repo's own oracle outputs
made in the build container — regression vectors, NOT outputs of the reference
binary (which has no tests and cannot be built here): parity stays unpinned."""
import numpy as np
import pytest

import bev_amd
import golden_util as gu
import oracle_lib as orc
from golden_data import INPUT_SHA256, ORACLE_OUTPUTS

CASES = [(s, n) for s, d in ORACLE_OUTPUTS.items() for n in d]


@pytest.mark.parametrize("sensor,name", CASES)
def test_oracle_reproduces_fixture(sensor, name):
    p = bev_amd.params_for_sensor(sensor)
    pts = gu.make_input(p, sensor, name)
    assert gu.sha(pts) == INPUT_SHA256[sensor][name], "synthetic input is not reproducible on this machine"
    want = ORACLE_OUTPUTS[sensor][name]
    assert len(pts) == want["n_points"]
    sp = orc.sensor_from_params(p)
    ordered, gm, multi, single = orc.process_frame(sp, pts)
    _, _, avg = orc.mark_ground(sp, orc.order_cloud(sp, pts))
    got = gu.summarize(ordered, gm, avg, multi, single)
    for k, v in got.items():
        assert v == want[k], f"{sensor}/{name}: {k} differs from the fixture"


def test_tiny_full_data_fixture():
    p, pts, ordered, gm, multi, single = gu.load_tiny()
    sp = orc.sensor_from_params(p)
    o, g, m, s = orc.process_frame(sp, pts)
    assert o.tobytes() == ordered.tobytes()
    assert np.array_equal(g, gm) and np.array_equal(m, multi) and np.array_equal(s, single)
    # the fixture exercises what it claims to
    slot = pts["row"].astype(np.int64) * 1056 + pts["col"]
    ok = (pts["row"] < 32) & (pts["col"] < 1056)
    assert (~ok).any() and len(np.unique(slot[ok])) < ok.sum() and (pts["intensity"] == -1).any()


def test_config1_plumbing_sizes():
    # BASELINE.json configs[0]: single 16k-pt HDL_32E cloud through the CPU path: output sizes
    p = bev_amd.params_for_sensor("HDL_32E")
    pts = gu.make_input(p, "HDL_32E", "config1_16k")
    o, g, m, s = orc.process_frame(orc.sensor_from_params(p), pts)
    assert m.nbytes == 1204224 and s.nbytes == 50176 and o.nbytes == 33792 * 32



def test_reference_dumps_if_present():
    """tests/golden/REF_PIN.md: outputs of the REFERENCE binary on the committed inputs, made by tests/golden/ref_dump.cpp
    on a machine with PCL / OpenCV.  Present: the oracle (and the host writers) must equal them byte for byte — that is
    what pins parity.  Absent (this image cannot build the reference): skipped, and parity stays unpinned."""
    import ctypes as C
    from pathlib import Path
    d = Path(__file__).resolve().parent / "golden" / "ref_pin"
    names = sorted(f.stem for f in d.glob("*.points") if (d / f"{f.stem}.ordered").exists())
    if not names:
        pytest.skip("no reference dumps under tests/golden/ref_pin (see tests/golden/REF_PIN.md): parity unpinned")
    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    for name in names:
        pts = np.frombuffer((d / f"{name}.points").read_bytes(), bev_amd.POINT_DTYPE)
        o, g, m, s = orc.process_frame(sp, pts)
        assert (d / f"{name}.ordered").read_bytes() == o.tobytes(), f"{name}: labelled ordered cloud"
        if (d / f"{name}.gm").exists():
            assert (d / f"{name}.gm").read_bytes() == g.tobytes(), f"{name}: ground_mat"
        assert (d / f"{name}.bin").read_bytes() == m.tobytes(), f"{name}: multi-BEV .bin"
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            rc = orc.lib().oracle_save_bin_csv(m.ctypes.data, s.ctypes.data, p.mat_size, p.n_layers,
                                               str(Path(td) / "o.bin").encode(), str(Path(td) / "o.csv").encode())
            assert rc == 0
            assert (d / f"{name}.csv").read_bytes() == (Path(td) / "o.csv").read_bytes(), f"{name}: single-BEV .csv framing"
        if (d / f"{name}.pcd").exists():
            import pcd_util
            head, cloud = pcd_util.read_pcd_binary(d / f"{name}.pcd")
            assert cloud.tobytes() == o.tobytes(), f"{name}: PCD payload"
            assert f"POINTS {p.slots}" in head and "FIELDS x y z intensity row col t label" in head
