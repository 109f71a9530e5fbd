"""GPU: the HIP path against the committed fixtures (no oracle in the loop)."""
import numpy as np
import pytest

import bev_amd
import golden_util as gu
from golden_data import INPUT_SHA256, ORACLE_OUTPUTS

pytestmark = pytest.mark.gpu
CASES = [(s, n) for s, d in ORACLE_OUTPUTS.items() for n in d]


@pytest.fixture(scope="module")
def ctxs():
    made = {}
    for sensor in ORACLE_OUTPUTS:
        p = bev_amd.params_for_sensor(sensor)
        made[sensor] = (p, bev_amd.BevContext(p, device=0, max_batch=2, max_points=250000))
    yield made
    for _, c in made.values():
        c.close()


@pytest.mark.parametrize("sensor,name", CASES)
def test_hip_path_reproduces_fixture(ctxs, sensor, name):
    p, ctx = ctxs[sensor]
    pts = gu.make_input(p, sensor, name)
    assert gu.sha(pts) == INPUT_SHA256[sensor][name]
    ordered, multi, single, gm = ctx.process_batch([pts], want_ground_mat=True)
    avg = ctx.cell_avg(0, 1)[0]
    got = gu.summarize(ordered[0], gm[0], avg, multi[0], single[0])
    want = ORACLE_OUTPUTS[sensor][name]
    for k, v in got.items():
        assert v == want[k], f"{sensor}/{name}: {k} differs from the fixture"


def test_tiny_full_data_fixture(ctxs):
    p, pts, ordered, gm, multi, single = gu.load_tiny()
    _, ctx = ctxs["HDL_32E"]
    o, m, s, g = ctx.process_batch([pts], want_ground_mat=True)
    assert o[0].tobytes() == ordered.tobytes()
    assert np.array_equal(g[0], gm) and np.array_equal(m[0], multi) and np.array_equal(s[0], single)
