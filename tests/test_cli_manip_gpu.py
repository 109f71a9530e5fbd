"""GPU: the two older tools of the same path end to end: batch_cloud_manip <root>
(BatchCloudManip.cpp:269-335) and cloud_manip <pcd> tx ty tz yaw (CloudManip.cpp:111-161, without its viewer):
directory tree, file names, CSV / PNG / PCD payloads against the oracle."""
import subprocess

import numpy as np
import pytest

import bev_amd
import oracle_lib as orc
import pcd_util
from bev_amd import synth

pytestmark = pytest.mark.gpu
BATCH_CLI = bev_amd.PKG_DIR / "host" / "batch_cloud_manip"
ONE_CLI = bev_amd.PKG_DIR / "host" / "cloud_manip"


def _csv_text(grid):
    """cv::Formatter FMT_CSV with set32fPrecision(4) as the host writes it: '%.4g', ', ' between, one line per row."""
    return "".join(", ".join("%.4g" % float(v) for v in row) + "\n" for row in grid)


def _png_of(grid):
    """cv::imwrite of a CV_32F Mat: saturate_cast<uchar> = round half to even, clamp"""
    return np.clip(np.rint(grid), 0, 255).astype(np.uint8)


def test_batch_cloud_manip_end_to_end(tmp_path):
    assert BATCH_CLI.exists(), "host CLI not built"
    p = bev_amd.params_for_sensor("HDL_64E")   # the tool's hard-coded constants (BatchCloudManip.cpp:13-14, :85)
    sp = orc.sensor_from_params(p)
    root = tmp_path / "kf"
    (root / "keyframe_point_cloud").mkdir(parents=True)
    frames = {"000000": synth.sweep(p, 0), "000001": synth.adversarial(p, 20000, 9), "000002": np.empty(0, bev_amd.POINT_DTYPE)}
    for name, pts in frames.items():
        pcd_util.write_pcd_binary(root / "keyframe_point_cloud" / f"{name}.pcd", pts)
    (root / "output_bvm").mkdir()
    (root / "output_bvm" / "stale.csv").write_text("must be removed")  # rm -rf semantics (:294)

    r = subprocess.run([str(BATCH_CLI), str(root)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert [l for l in r.stdout.splitlines() if l.startswith("Converting file: ")] == [f"Converting file: {n}" for n in frames]
    assert r.stdout.count("[TIME] Preprocessing and BEV generation: ") == 3
    assert "[TIME] Average preprocessing and BEV generation: " in r.stdout and "Done." in r.stdout
    assert not (root / "output_bvm" / "stale.csv").exists()
    assert sorted(f.name for f in (root / "output_bvm").iterdir()) == sorted(f"{n}.{e}" for n in frames for e in ("csv", "png"))

    for name, pts in frames.items():
        ordered, _, _ = orc.mark_ground(sp, orc.order_cloud(sp, pts))
        want = orc.float_bev(ordered, 1.0, True)               # label == 0 skipped (:218)
        assert want.shape == (201, 201)
        assert (root / "output_bvm" / f"{name}.csv").read_text() == _csv_text(want)
        assert np.array_equal(pcd_util.read_png_gray8(root / "output_bvm" / f"{name}.png"), _png_of(want))
        head, cloud = pcd_util.read_pcd_binary(root / "non_ground_point_cloud" / f"{name}.pcd")
        assert f"POINTS {p.slots}" in head
        assert cloud.tobytes() == ordered.tobytes()


def test_batch_cloud_manip_usage():
    r = subprocess.run([str(BATCH_CLI)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Usage:" in r.stdout


def test_cloud_manip_end_to_end(tmp_path):
    assert ONE_CLI.exists(), "host CLI not built"
    p = bev_amd.params_for_sensor("HDL_64E")
    pts = synth.sweep(p, 5)
    (tmp_path / "in").mkdir()
    src = tmp_path / "in" / "000123.pcd"
    pcd_util.write_pcd_binary(src, pts)
    args = ("1.5", "-2.25", "0.125", "30")
    r = subprocess.run([str(ONE_CLI), str(src), *args], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    theta = np.float32(np.float64(np.float32(30) / np.float32(180)) * np.pi)
    assert r.stdout.startswith("rotating yaw radiance: %g" % theta)

    m = orc.yaw_translate_matrix(*[float(a) for a in args])
    moved = orc.transform_cloud(pts, m)
    for tag, cloud in (("input", pts), ("output", moved)):
        want = orc.float_bev(cloud, 1.0, False)                # no label test in this tool (:88)
        assert (tmp_path / f"000123.pcd_{tag}.csv").read_text() == _csv_text(want)
        assert np.array_equal(pcd_util.read_png_gray8(tmp_path / f"000123.pcd_{tag}.csv.png"), _png_of(want))
        head, got = pcd_util.read_pcd_binary(tmp_path / f"000123.pcd_{tag}.pcd")
        assert f"POINTS {len(pts)}" in head
        assert pcd_util.to_packed(got).tobytes() == pcd_util.to_packed(cloud).tobytes()


def test_cloud_manip_usage(tmp_path):
    r = subprocess.run([str(ONE_CLI)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Usage:" in r.stdout
    r = subprocess.run([str(ONE_CLI), str(tmp_path / "missing.pcd"), "0", "0", "0", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Can not read" in r.stderr
