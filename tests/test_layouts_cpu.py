"""CPU: what the reference's scatter (getOrderedCloud, BatchMultiBevGen.cpp:98,102-116) does to the two producer layouts
that the device reads in place without sorting — stated on the ORACLE, so that the premises of the device routes
(tests/test_gpu_structured.py, tests/test_gpu_colmajor.py) are checked where no GPU is needed:
  structured clouds (KittiPointCloudSelect.cpp:206-207,240): the identity on every slot but slot 0, which ends up all-zero
      iff a record after the first is all-zero;
  firing order (MulranPointCloudSelect.cpp:112-130): per slot the LAST firing that claims it, columns >= H dropped."""
import numpy as np

import bev_amd
import oracle_lib as orc
from bev_amd import synth


def test_structured_cloud_scatter_is_the_identity_but_for_slot_zero():
    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    zero = np.zeros(1, bev_amd.POINT_DTYPE)[0]
    full = synth.structured(p, 1, 1.0)
    assert len(full) == p.slots and (full["label"] == -2).all()
    assert orc.order_cloud(sp, full).tobytes() == full.tobytes()            # no empty record: slot 0 keeps its point
    holes = synth.structured(p, 2, 0.9)
    holes[0] = full[0]                                                      # record 0 real, empty records later
    want = holes.copy()
    want[0] = zero
    assert orc.order_cloud(sp, holes).tobytes() == want.tobytes()
    only_first = full.copy()
    only_first[0] = zero                                                    # only record 0 empty: nothing lands on top of it
    assert orc.order_cloud(sp, only_first).tobytes() == only_first.tobytes()
    one_late = full.copy()
    one_late[len(full) - 1] = zero                                          # the very last record empty: slot 0 is overwritten last
    want = one_late.copy()
    want[0] = zero
    assert orc.order_cloud(sp, one_late).tobytes() == want.tobytes()
    claims = holes.copy()                                                   # row = col = 0 WITH contents: the last such record is slot 0
    k = 5000
    claims[k] = full[k]
    claims[k]["row"], claims[k]["col"] = 0, 0
    got = orc.order_cloud(sp, claims)
    last_zero = np.flatnonzero((claims["row"] == 0) & (claims["col"] == 0))[-1]
    assert got[0].tobytes() == claims[last_zero].tobytes()
    assert got[k].tobytes() == zero.tobytes()                               # its own slot stays value-initialised


def test_firing_order_scatter_keeps_the_last_firing_per_slot():
    p = bev_amd.params_for_sensor("OS1_64")
    sp = orc.sensor_from_params(p)
    n, h = p.n_scan, p.horizon_scan
    f = synth.firing_order(p, 3)
    assert len(f) == p.slots and (f["row"] == np.arange(len(f)) % n).all()
    disp = f["col"].astype(np.int64) - np.arange(len(f)) // n
    assert disp.min() == 0 and disp.max() == 1 and (f["col"] == h).any()    # the synthetic config: + 0 .. 1, the seam's overflow
    got = orc.order_cloud(sp, f)
    slot = f["row"].astype(np.int64) * h + f["col"]
    ok = f["col"] < h
    last = np.full(p.slots, -1, np.int64)
    np.maximum.at(last, slot[ok], np.flatnonzero(ok))                       # input order = firing order within a row
    want = np.zeros(p.slots, bev_amd.POINT_DTYPE)
    want[last >= 0] = f[last[last >= 0]]
    assert got.tobytes() == want.tobytes()
    assert (last < 0).sum() > 0 and len(np.unique(slot[ok])) < ok.sum()     # empty slots and contested slots both occur
