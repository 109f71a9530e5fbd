"""GPU: the experimental sorted-prefix fast path (BEV_FAST=1; k_prefix_len / k_prefix_bounds /
k_tail / STRIP_FAST in csrc/bev_kernels.hip).  It guesses where the slot-sorted prefix of a frame
ends and where each (row, strip) tile starts, then VERIFIES while consuming; frames that fail are
redone by the general path.  Whatever the guesses, results must equal the oracle's."""
import os

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fast_ctx():
    old = os.environ.get("BEV_FAST")
    os.environ["BEV_FAST"] = "1"   # read by bev_create
    made = {}

    def get(sensor):
        if sensor not in made:
            p = bev_amd.params_for_sensor(sensor)
            made[sensor] = (p, bev_amd.BevContext(p, device=0, max_batch=16, max_points=800000))  # host-buffer calls run in chunks of max_batch / 2
        return made[sensor]

    yield get
    for _, c in made.values():
        c.close()
    if old is None:
        os.environ.pop("BEV_FAST", None)
    else:
        os.environ["BEV_FAST"] = old


def _check(p, ctx, frames, expect_failed=None, expect_prefix=None):
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    ln, fl = ctx.fast_path_stats(len(frames))
    sp = orc.sensor_from_params(p)
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = orc.process_frame(sp, pts)
        assert np.array_equal(gm[i], o_gm), f"frame {i}: ground_mat differs (prefix {ln[i]}, failed {fl[i]})"
        assert ordered[i].tobytes() == o_ord.tobytes(), f"frame {i}: ordered cloud differs (prefix {ln[i]}, failed {fl[i]})"
        assert np.array_equal(multi[i], o_multi) and np.array_equal(single[i], o_single)
    if expect_failed is not None:
        assert [int(v != 0) for v in fl] == expect_failed, (ln, fl)
    if expect_prefix is not None:
        assert list(ln) == expect_prefix
    return ln, fl


@pytest.mark.parametrize("sensor", ["HDL_32E", "HDL_64E", "OS1_64"])
def test_sorted_prefix_with_duplicate_tail(fast_ctx, sensor):
    p, ctx = fast_ctx(sensor)
    frames = [synth.sweep(p, f) for f in range(4)]
    # row-major kept slots, then 5000 appended duplicates: the prefix is everything but the tail
    _check(p, ctx, frames, expect_failed=[0, 0, 0, 0], expect_prefix=[len(f) - 5000 for f in frames])


def test_inputs_that_defeat_the_guess_fall_back(fast_ctx):
    p, ctx = fast_ctx("HDL_32E")
    base = synth.sweep(p, 7, n_dup=0)
    swapped = base.copy()
    swapped[[1000, 1001]] = swapped[[1001, 1000]]           # one descent deep inside: prefix stops there
    dup_inside = np.concatenate([base[:5000], base[4999:5000], base[5000:]])  # a repeated slot
    oob_inside = base.copy()
    oob_inside["row"][20000] = 60                             # an out-of-range row inside the prefix
    rot = np.concatenate([base[len(base) // 2:], base[:len(base) // 2]])      # two sorted halves
    sparse_bad = base.copy()
    sparse_bad[[123, 15000]] = sparse_bad[[15000, 123]]       # far swap: sampled predicate may not see it
    frames = [base, swapped, dup_inside, oob_inside, rot, sparse_bad, synth.firing_order(p, 1),
              synth.adversarial(p, 40000, 5, True)]
    ln, fl = _check(p, ctx, frames)
    assert fl[0] == 0 and ln[0] == len(base)                 # clean frame: whole input is the prefix
    # a violation is either cut off by the prefix guess or caught by verification (-> general path)
    assert fl[1] != 0 or ln[1] <= 1001
    assert fl[2] != 0 or ln[2] <= 5000
    assert fl[3] != 0 or ln[3] <= 20000


def test_empty_tiny_and_all_tail(fast_ctx):
    p, ctx = fast_ctx("HDL_32E")
    full = synth.sweep(p, 3)
    rev = full[::-1].copy()                                   # strictly descending: prefix of length 1
    frames = [np.empty(0, bev_amd.POINT_DTYPE), full[:1], full[:300], rev[:20000], synth.concat(p, 0, n_sweeps=3)]
    _check(p, ctx, frames)
