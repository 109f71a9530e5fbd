"""End-to-end rate of the batch_multi_bev_gen CLI (PCD in, .bin / .csv / .png / .pcd out) on synthetic HDL_64E frames.
usage (GPU box): python scripts/cli_rate.py [n_frames]"""
import os, subprocess, sys, tempfile, time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd"))
sys.path.insert(0, str(REPO / "tests"))
import bev_amd, pcd_util
from bev_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bev_amd.params_for_sensor("HDL_64E")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    root = Path(d) / "kf"
    (root / "keyframe_point_cloud").mkdir(parents=True)
    for i in range(n):
        pcd_util.write_pcd_binary(root / "keyframe_point_cloud" / f"{i:06d}.pcd", synth.sweep(p, i))
    (root / "keyframe_pose.csv").write_text("\n".join(
        ",".join([str(i), repr(3.0 * i), "0.0", "0.0", "0", "0", "0"] + ["1.0", "0.0", "0.0", "0.0", "1.0", "0.0", "0.0", "0.0", "1.0"]) for i in range(n)) + "\n")
    cli = str(bev_amd.PKG_DIR / "host" / "batch_multi_bev_gen")
    for env_extra, tag in (({"BEV_NO_PNG": "1"}, "no png"), ({}, "with 25 png per frame")):
        for threads in (os.environ.get("BEV_IO_THREADS_LIST", "1,16").split(",")):
            env = dict(os.environ, BEV_IO_THREADS=threads, **env_extra)
            t0 = time.time()
            r = subprocess.run([cli, str(root), "HDL_64E"], env=env, capture_output=True, text=True)
            dt = time.time() - t0
            line = [l for l in r.stdout.splitlines() if l.startswith("[TIME]")]
            print(f"{tag:24s} io threads {threads:>3s}: {n / dt:8.1f} frames/s wall ({dt:.2f} s), rc {r.returncode}, {line[-1] if line else r.stderr[-200:]}")
