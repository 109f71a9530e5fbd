"""GPU: the batch machinery at BASELINE scale — many frames, device-resident, through the fused launches
(k_stage: a sub-batch's walk beside the later stages of its neighbours, two streams, eight workspace sets; BEV_LANES=1 /
lanes=1 below: every kernel a launch of its own).  Size-independent properties: results do not depend on how frames are cut into sub-batches or
dealt to lanes; a checksum over all frames is stable across repeated asynchronous calls; sampled frames
equal the oracle."""
import hashlib
import os

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
from bev_amd import synth

pytestmark = pytest.mark.gpu
N_FRAMES = 300


def _run(p, frames, sub_batch, lanes, repeats=1, want_info=False):
    import torch

    old = os.environ.get("BEV_LANES")
    os.environ["BEV_LANES"] = str(lanes)
    try:
        ctx = bev_amd.BevContext(p, device=0, max_batch=sub_batch, max_points=max(len(f) for f in frames))
    finally:
        if old is None:
            os.environ.pop("BEV_LANES", None)
        else:
            os.environ["BEV_LANES"] = old
    dev = torch.device("cuda:0")
    S, M, L = p.slots, p.mat_size, p.n_layers
    offs = np.zeros(len(frames) + 1, np.uint64)
    offs[1:] = np.cumsum([len(f) for f in frames])
    d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
    d_ord = torch.zeros(len(frames) * S * 32, dtype=torch.uint8, device=dev)
    d_multi = torch.zeros(len(frames) * L * M * M, dtype=torch.uint8, device=dev)
    d_single = torch.zeros(len(frames) * M * M, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(repeats):  # back-to-back asynchronous calls over the same buffers
        ctx.process_device(len(frames), d_in.data_ptr(), offs, d_ord.data_ptr(), d_multi.data_ptr(), d_single.data_ptr())
    ctx.synchronize()
    out = (d_ord.cpu().numpy(), d_multi.cpu().numpy(), d_single.cpu().numpy())
    if want_info:  # how the frames of the LAST sub-batch reached their slots (bev_debug_get_frame_info)
        n_last = len(frames) - ((len(frames) - 1) // sub_batch) * sub_batch
        out = out + (ctx.frame_info(0, n_last),)
    ctx.close()
    return out


def _digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(a.tobytes())
    return h.hexdigest()


def test_sub_batching_and_lanes_do_not_change_results():
    p = bev_amd.params_for_sensor("HDL_64E")
    frames = [synth.sweep(p, 5000 + f) for f in range(N_FRAMES)]
    ref = _run(p, frames, sub_batch=64, lanes=2, repeats=3)
    assert _digest(ref) == _digest(_run(p, frames, sub_batch=7, lanes=1))     # 43 ragged sub-batches, one lane
    assert _digest(ref) == _digest(_run(p, frames, sub_batch=300, lanes=3))   # one sub-batch
    assert _digest(ref) == _digest(_run(p, frames, sub_batch=37, lanes=4, repeats=2))
    # launches past bench.py's 500 frames (4,500 walk workgroups): the same 300 frames three times over
    want3 = _digest([np.tile(a, 3) for a in ref])
    assert want3 == _digest(_run(p, frames * 3, sub_batch=600, lanes=2, repeats=2))   # launches of 600 and 300 frames
    assert want3 == _digest(_run(p, frames * 3, sub_batch=900, lanes=2))              # one launch of 900 (8,100 workgroups)
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)
    for i in (0, 63, 64, 150, N_FRAMES - 1):  # sub-batch edges and the last frame
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        assert ref[0][i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes()
        assert ref[1][i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
        assert ref[2][i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes()
    # every frame produced something plausible: a ground ring and an occupied BEV
    labels = ref[0].view(bev_amd.POINT_DTYPE).reshape(N_FRAMES, S)["label"]
    assert ((labels == 0).sum(axis=1) > 20000).all()
    assert (ref[1].reshape(N_FRAMES, -1).astype(np.int64).sum(axis=1) > 0).all()


def test_winner_generation_wraps():
    """The winner table is generation-tagged instead of cleared per sub-batch; with max_points = 2^27 only 4 tag bits
    remain, so the tag wraps (and the table is cleared) every 15 sub-batches.  Different frames follow each other in
    the same workspace slot, so a stale entry surviving a generation would show."""
    import oracle_lib as orc
    p = bev_amd.params_for_sensor("HDL_32E")
    ctx = bev_amd.BevContext(p, device=0, max_batch=2, max_points=(1 << 27) - 1)
    try:
        sp = orc.sensor_from_params(p)
        frames = [synth.sweep(p, 1), synth.adversarial(p, 30000, 4, False), synth.firing_order(p, 2),
                  synth.sweep(p, 5, keep=0.5)]
        want = [orc.process_frame(sp, f) for f in frames]
        for it in range(40):
            k = (it * 7 + it // 3) % len(frames)
            ordered, multi, single, gm = ctx.process_batch([frames[k]], want_ground_mat=True)
            o_ord, o_gm, o_multi, o_single = want[k]
            assert ordered[0].tobytes() == o_ord.tobytes(), it
            assert np.array_equal(gm[0], o_gm) and np.array_equal(multi[0], o_multi) and np.array_equal(single[0], o_single), it
    finally:
        ctx.close()


@pytest.mark.parametrize("layout,sub_batch", [("sweep", 500), ("sweep", 256), ("structured", 500)])
def test_baseline_config_every_frame_matches_oracle(layout, sub_batch):
    """BASELINE configs[1] as bench.py runs it (1000 HDL_64E frames, sub-batches of 500 — bench.py's default launch
    size — and of 256, fused launches over two streams, three back-to-back asynchronous steps over the same buffers): EVERY frame
    of the last step against the oracle.  "structured": the same sweeps in the layout the KITTI selector writes
    (bench.py --workload hdl64_structured)."""
    from concurrent.futures import ThreadPoolExecutor

    p = bev_amd.params_for_sensor("HDL_64E")
    n = 1000
    make = (lambda f: synth.sweep(p, f, keep=0.98, n_dup=5000)) if layout == "sweep" else (lambda f: synth.structured(p, f, keep=0.98))
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(make, range(n)))
    ords, multis, singles = _run(p, frames, sub_batch=sub_batch, lanes=2, repeats=3)
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)

    def check(i):  # the oracle is plain C behind ctypes: the threads run it in parallel
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes()
                and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())

    with ThreadPoolExecutor(16) as ex:
        bad = [i for i, ok in enumerate(ex.map(check, range(n))) if not ok]
    assert not bad, f"{len(bad)} of {n} frames differ from the oracle, first: {bad[:8]}"


def test_unstaged_call_followed_by_staged_call_without_sync():
    """A call that fits one sub-batch runs wholly on lane 0 with workspace set 0; a call with several sub-batches that
    follows WITHOUT a synchronisation runs its front on another stream and reuses set 0: the second call's front must
    wait for the first call's back (the hand-over event is recorded by every sub-batch, staged or not)."""
    import torch

    p = bev_amd.params_for_sensor("HDL_64E")
    sb = 24
    small = [synth.sweep(p, 9000 + f) for f in range(sb)]
    big = [synth.sweep(p, 9100 + f, keep=0.9, n_dup=3000) for f in range(4 * sb + 5)]
    ctx = bev_amd.BevContext(p, device=0, max_batch=sb, max_points=max(len(f) for f in small + big))
    dev = torch.device("cuda:0")
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)

    def stage(frames):
        offs = np.zeros(len(frames) + 1, np.uint64)
        offs[1:] = np.cumsum([len(f) for f in frames])
        d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
        outs = [torch.zeros(len(frames) * k, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
        return offs, d_in, outs

    a, b = stage(small), stage(big)
    torch.cuda.synchronize()
    try:
        for _ in range(3):  # small, big, small, big, ... never synchronised in between
            for offs, d_in, outs in (a, b):
                ctx.process_device(len(offs) - 1, d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
        ctx.synchronize()
        for frames, (_, _, outs) in ((small, a), (big, b)):
            got = [o.cpu().numpy() for o in outs]
            for i, pts in enumerate(frames):
                o_ord, _, o_multi, o_single = orc.process_frame(sp, pts, want_gm=False)
                assert got[0][i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes(), (len(frames), i)
                assert got[1][i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes(), (len(frames), i)
                assert got[2][i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes(), (len(frames), i)
    finally:
        ctx.close()


def test_os1_firing_order_config_every_frame_matches_oracle():
    """BASELINE configs[2] at its stated size: 1000 MulRan-style OS1_64 clouds in firing order (65,536 points each,
    column-major, incl. the col == 1024 overflow; MulranPointCloudSelect.cpp:120-125) through the pipelined
    device-resident path, EVERY frame against the oracle.  This is where k_order_scan regroups scattering blocks by row
    in LDS (BatchMultiBevGen.cpp:102-116 sees an unordered cloud)."""
    from concurrent.futures import ThreadPoolExecutor

    p = bev_amd.params_for_sensor("OS1_64")
    n = 1000
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda f: synth.firing_order(p, f), range(n)))
    ords, multis, singles = _run(p, frames, sub_batch=500, lanes=2, repeats=2)   # bench.py's launch size
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)

    def check(i):
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes()
                and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())

    with ThreadPoolExecutor(16) as ex:
        bad = [i for i, ok in enumerate(ex.map(check, range(n))) if not ok]
    assert not bad, f"{len(bad)} of {n} frames differ from the oracle, first: {bad[:8]}"


def _every_frame(p, frames, outs):
    from concurrent.futures import ThreadPoolExecutor

    ords, multis, singles = outs[:3]
    S, M, L = p.slots, p.mat_size, p.n_layers
    sp = orc.sensor_from_params(p)

    def check(i):  # the oracle is plain C behind ctypes: the threads run it in parallel
        o_ord, _, o_multi, o_single = orc.process_frame(sp, frames[i], want_gm=False)
        return (ords[i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes()
                and multis[i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes()
                and singles[i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes())

    with ThreadPoolExecutor(16) as ex:
        bad = [i for i, ok in enumerate(ex.map(check, range(len(frames)))) if not ok]
    assert not bad, f"{len(bad)} of {len(frames)} frames differ from the oracle, first: {bad[:8]}"


def test_real_mulran_sweeps_at_launch_size_every_frame_matches_oracle():
    """What mulran_point_cloud_select writes for REAL Ouster sweeps (MulranPointCloudSelect.cpp:112-130: any start azimuth,
    either direction, staggered laser columns, 3 % no-return records in column 0) at bench.py's launch size — 1000 OS1_64
    frames, sub-batches of 500, two asynchronous steps: EVERY frame against the oracle, and the frames of the last
    sub-batch on route 5 (`bench.py --workload os1_firing_real`; round 5 tested the route on a dozen small batches)."""
    from concurrent.futures import ThreadPoolExecutor

    p = bev_amd.params_for_sensor("OS1_64")
    n = 1000
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda f: synth.firing_real(p, f, noret=0.03), range(n)))
    outs = _run(p, frames, sub_batch=500, lanes=2, repeats=2, want_info=True)
    modes = outs[3][:, 1]
    assert (modes == 5).sum() >= 0.99 * len(modes), np.unique(modes, return_counts=True)
    _every_frame(p, frames, outs)


def test_mixed_layouts_at_launch_size_every_frame_matches_oracle():
    """The layouts of the reference's producers alternating in groups of 32 HDL_64E frames (`bench.py --workload mixed`:
    sorted sweeps with appended duplicates, structured clouds, real firing order), 1000 frames, sub-batches of 500: EVERY
    frame against the oracle; no frame of the last sub-batch — frames 500 ... 999, far behind the first three groups —
    is redone (the mode hint is sticky, BEV_MODE_TTL)."""
    from concurrent.futures import ThreadPoolExecutor

    p = bev_amd.params_for_sensor("HDL_64E")
    n = 1000

    def make(f):
        kind = (f // 32) % 3
        if kind == 0:
            return synth.sweep(p, f, keep=0.98, n_dup=5000)
        return synth.structured(p, f, keep=0.98) if kind == 1 else synth.firing_real(p, f, noret=0.03)

    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(make, range(n)))
    outs = _run(p, frames, sub_batch=500, lanes=2, repeats=2, want_info=True)
    modes = outs[3][:, 1]
    assert not (modes == 2).any() and set(np.unique(modes)) == {1, 3, 5}, np.unique(modes, return_counts=True)
    _every_frame(p, frames, outs)


def test_many_small_calls_without_a_synchronisation():
    """The later stages of a call's last sub-batches ride in the NEXT call's launches (or bev_synchronize launches them):
    thirty calls of one to three small sub-batches each, never synchronised in between, go three times round the eight
    workspace sets and alternate between the two streams; every call has its own buffers; one bev_synchronize at the end;
    every frame of every call against the oracle.  (A bare device synchronisation is NOT enough — include/bev_mi355x.h.)"""
    import torch

    p = bev_amd.params_for_sensor("HDL_32E")
    sb = 5
    sp = orc.sensor_from_params(p)
    dev = torch.device("cuda:0")
    S, M, L = p.slots, p.mat_size, p.n_layers
    sizes = [1 + (7 * i) % (3 * sb) for i in range(30)]
    calls = []
    seed = 12000
    for n in sizes:
        frames = []
        for _ in range(n):
            k = seed % 4
            frames.append(synth.sweep(p, seed, keep=0.9, n_dup=200) if k == 0 else synth.structured(p, seed, 0.9) if k == 1
                          else synth.firing_order(p, seed) if k == 2 else synth.adversarial(p, 20000, 3, False))
            seed += 1
        calls.append(frames)
    ctx = bev_amd.BevContext(p, device=0, max_batch=sb, max_points=max(len(f) for c in calls for f in c))
    staged = []
    for frames in calls:
        offs = np.zeros(len(frames) + 1, np.uint64)
        offs[1:] = np.cumsum([len(f) for f in frames])
        d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
        outs = [torch.zeros(len(frames) * k, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
        staged.append((offs, d_in, outs))
    torch.cuda.synchronize()
    try:
        for offs, d_in, outs in staged:
            ctx.process_device(len(offs) - 1, d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
        ctx.synchronize()
        for c, (frames, (_, _, outs)) in enumerate(zip(calls, staged)):
            got = [o.cpu().numpy() for o in outs]
            for i, pts in enumerate(frames):
                o_ord, _, o_multi, o_single = orc.process_frame(sp, pts, want_gm=False)
                assert got[0][i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes(), (c, i)
                assert got[1][i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes(), (c, i)
                assert got[2][i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes(), (c, i)
    finally:
        ctx.close()


def test_shuffled_sweeps_at_launch_size_every_frame_matches_oracle():
    """getOrderedCloud's actual contract is ANY input order (BatchMultiBevGen.cpp:102-116, last writer wins): BASELINE
    configs[1]'s 1000 HDL_64E frames with their points in a random order (`bench.py --workload hdl64_shuffled`) go the
    general way — order scan + gather walk, the route of every layout the probe does not recognise and of every frame that
    fails its checks — at sub-batch 500: EVERY frame against the oracle, every frame of the last sub-batch on route 0."""
    from concurrent.futures import ThreadPoolExecutor

    p = bev_amd.params_for_sensor("HDL_64E")
    n = 1000

    def make(f):
        pts = synth.sweep(p, f, keep=0.98, n_dup=5000)
        return pts[np.random.default_rng(0x5EED0000 + f).permutation(len(pts))]

    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(make, range(n)))
    outs = _run(p, frames, sub_batch=500, lanes=2, repeats=2, want_info=True)
    assert (outs[3][:, 1] == 0).all(), np.unique(outs[3][:, 1], return_counts=True)
    _every_frame(p, frames, outs)


def test_buffers_filled_on_the_default_stream_right_before_the_call():
    """The library's streams are non-blocking; what the caller has queued on the DEFAULT stream before a device-resident call
    — here torch's fill of the very output buffers, 5.6 GB of it — is waited for on the device (an event on the default stream
    at the head of every call).  Without that the fill's tail lands on top of the second sub-batch's outputs (round 6's soak
    found it: the fused launches of BOTH streams start writing at once)."""
    import torch

    p = bev_amd.params_for_sensor("HDL_64E")
    n = 1000
    base = [synth.sweep(p, 7000 + f, keep=0.97, n_dup=1000) for f in range(40)]
    frames = [base[f % len(base)] for f in range(n)]
    sp = orc.sensor_from_params(p)
    want = [orc.process_frame(sp, f, want_gm=False) for f in base]
    dev = torch.device("cuda:0")
    S, M, L = p.slots, p.mat_size, p.n_layers
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum([len(f) for f in frames])
    ctx = bev_amd.BevContext(p, device=0, max_batch=500, max_points=max(len(f) for f in frames))
    try:
        d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
        for rep in range(2):
            torch.cuda.synchronize()
            outs = [torch.full((n * k,), 0x5A, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
            ctx.process_device(n, d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())   # no synchronisation in between
            ctx.synchronize()
            for i in (0, 1, 250, 499, 500, 501, 640, 777, 900, 998, 999):
                o_ord, _, o_multi, o_single = want[i % len(base)]
                assert outs[0][i * S * 32:(i + 1) * S * 32].cpu().numpy().tobytes() == o_ord.tobytes(), (rep, i)
                assert outs[1][i * L * M * M:(i + 1) * L * M * M].cpu().numpy().tobytes() == o_multi.tobytes(), (rep, i)
                assert outs[2][i * M * M:(i + 1) * M * M].cpu().numpy().tobytes() == o_single.tobytes(), (rep, i)
            del outs
    finally:
        ctx.close()


def test_other_entry_points_and_mode_switches_between_asynchronous_calls():
    """What a device-resident call leaves pending must survive whatever the caller does next with the context: a
    single-cloud entry point (they use workspace set 0 and the context's own stream: they launch what is pending first), a
    switch to serial launches and back (bev_set_lanes), a host-buffer call.  Every device-resident call has its own
    buffers and is checked after ONE final bev_synchronize."""
    import torch

    p = bev_amd.params_for_sensor("HDL_32E")
    sp = orc.sensor_from_params(p)
    dev = torch.device("cuda:0")
    S, M, L = p.slots, p.mat_size, p.n_layers
    sb = 6
    ctx = bev_amd.BevContext(p, device=0, max_batch=sb, max_points=40000)

    def stage(frames):
        offs = np.zeros(len(frames) + 1, np.uint64)
        offs[1:] = np.cumsum([len(f) for f in frames])
        d_in = torch.from_numpy(np.concatenate(frames).view(np.uint8).reshape(-1)).to(dev)
        outs = [torch.zeros(len(frames) * k, dtype=torch.uint8, device=dev) for k in (S * 32, L * M * M, M * M)]
        return frames, offs, d_in, outs

    calls = [stage([synth.sweep(p, 15000 + 20 * c + i, keep=0.9, n_dup=100 * c) for i in range(n)]) for c, n in enumerate((13, 5, 9, 17, 6, 11))]
    single = synth.sweep(p, 15999, keep=0.8)
    torch.cuda.synchronize()
    try:
        def go(k):
            frames, offs, d_in, outs = calls[k]
            ctx.process_device(len(frames), d_in.data_ptr(), offs, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())

        go(0)
        o = ctx.order_cloud(single)                                  # a single-cloud entry point in between
        assert o.tobytes() == orc.order_cloud(sp, single).tobytes()
        go(1)
        assert ctx.set_lanes(1) == 1                                 # serial launches ...
        go(2)
        assert ctx.set_lanes(2) > 1                                  # ... and fused again
        go(3)
        ordered, multi, single_bev, _ = ctx.process_batch([single, single[:1000]])   # a host-buffer call in between
        for i, pts in enumerate([single, single[:1000]]):
            o_ord, _, o_multi, o_single = orc.process_frame(sp, pts, want_gm=False)
            assert ordered[i].tobytes() == o_ord.tobytes() and np.array_equal(multi[i], o_multi) and np.array_equal(single_bev[i], o_single)
        go(4)
        mb = ctx.multi_bev(single)                                   # (raster of an arbitrary cloud)
        assert np.array_equal(mb.reshape(L, M, M), orc.multi_bev(sp, single))
        go(5)
        ctx.synchronize()
        for c, (frames, _, _, outs) in enumerate(calls):
            got = [t.cpu().numpy() for t in outs]
            for i, pts in enumerate(frames):
                o_ord, _, o_multi, o_single = orc.process_frame(sp, pts, want_gm=False)
                assert got[0][i * S * 32:(i + 1) * S * 32].tobytes() == o_ord.tobytes(), (c, i)
                assert got[1][i * L * M * M:(i + 1) * L * M * M].tobytes() == o_multi.tobytes(), (c, i)
                assert got[2][i * M * M:(i + 1) * M * M].tobytes() == o_single.tobytes(), (c, i)
    finally:
        ctx.close()
