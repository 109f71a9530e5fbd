"""usage (on the GPU box): python3 scripts/stream_soak.py [BEV_STREAM value, default 1] [repetitions, default 40]
The same three sorted HDL_64E sweeps (2000 appended points, labels from {-2, -1, 0, 1, 7}) through a fresh context per
repetition, every output of every frame against the oracle; prints which rows / strips / kinds of slots differed.  Found
the stream walk's in-flight register copy: wrong tail points in one run of six, 0 since."""
import os, sys
sys.path.insert(0,"/root/repo/tests"); sys.path.insert(0,"/root/repo/point-cloud-preprocessing-tools_amd")
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
os.environ["BEV_STREAM"] = mode
import bev_amd, oracle_lib as orc
from bev_amd import synth
p = bev_amd.params_for_sensor("HDL_64E")
rng = np.random.default_rng(5)
frames = []
for i in range(3):
    f = synth.sweep(p, 200 + i, keep=0.95, n_dup=2000)
    f["label"] = rng.choice(np.array([-2, -1, 0, 1, 7], dtype=np.int16), size=len(f), p=[0.5, 0.2, 0.1, 0.1, 0.1])
    frames.append(f)
sp = orc.sensor_from_params(p)
H = p.horizon_scan
ref = [orc.process_frame(sp, pts) for pts in frames]
shown = 0
import collections
rows = collections.Counter(); strips = collections.Counter(); kinds = collections.Counter(); incidents = collections.Counter()
for rep in range(reps):
    ctx = bev_amd.BevContext(p, device=0, max_batch=8, max_points=max(len(f) for f in frames))
    ordered, multi, single, gm = ctx.process_batch(frames, want_ground_mat=True)
    info = ctx.frame_info(0, 3)
    for i, pts in enumerate(frames):
        o_ord, o_gm, o_multi, o_single = ref[i]
        a = np.frombuffer(ordered[i].tobytes(), dtype=bev_amd.POINT_DTYPE); b = np.frombuffer(o_ord.tobytes(), dtype=bev_amd.POINT_DTYPE)
        bad = np.unique(np.nonzero(a.view(np.uint8).reshape(-1, 32) != b.view(np.uint8).reshape(-1, 32))[0])
        ngm = int((gm[i] != o_gm).sum()); nm = int((multi[i] != o_multi).sum()); ns = int((single[i] != o_single).sum())
        for s_ in bad:
            r_, c_ = divmod(int(s_), H)
            rows[r_] += 1; strips[c_ // 252] += 1; incidents[(rep, i, r_, c_ // 252)] += 1
            src_ = np.nonzero((pts["row"] == r_) & (pts["col"] == c_))[0]
            T_ = int(info[i][0])
            want_tail = len(src_) and src_.max() >= T_
            ours_empty = a[s_]["x"] == 0 and a[s_]["intensity"] == 0
            kinds[("oracle=tail" if want_tail else "oracle=prefix", "ours=empty" if ours_empty else ("ours=labelonly" if a[s_]["x"] == b[s_]["x"] else "ours=otherpoint"))] += 1
        if len(bad) or ngm or nm or ns:
            print("rep", rep, "frame", i, "info", info[i], "slots", len(bad), "gm", ngm, "multi", nm, "single", ns)
            for s in bad[:4]:
                if shown > 4: break
                shown += 1
                r, c = divmod(int(s), H)
                src = np.nonzero((pts["row"] == r) & (pts["col"] == c))[0]
                print("   slot", s, (r, c), "ours", a[s]["label"], a[s]["x"], "oracle", b[s]["label"], b[s]["x"], "gm", gm[i].reshape(-1)[s], o_gm.reshape(-1)[s], "input", [(int(j), int(pts[j]["label"])) for j in src])
    ctx.close()
print("rows", sorted(rows.items())); print("strips", sorted(strips.items())); print("kinds", kinds); print("incidents (rep, frame, row, strip) -> slots", sorted(incidents.items())[:40]); print("done", reps)
