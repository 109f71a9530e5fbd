"""CPU: without a GPU the CLI must fail loudly (no CPU fallback)."""
import subprocess

import pytest

import bev_amd
import pcd_util
from bev_amd import synth

CLI = bev_amd.PKG_DIR / "host" / "batch_multi_bev_gen"


def test_cli_fails_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = bev_amd.params_for_sensor("HDL_32E")
    (tmp_path / "keyframe_point_cloud").mkdir()
    pcd_util.write_pcd_binary(tmp_path / "keyframe_point_cloud" / "000000.pcd", synth.sweep(p, 0)[:100])
    r = subprocess.run([str(CLI), str(tmp_path), "HDL_32E"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no usable HIP device" in r.stderr
    assert not list((tmp_path / "output_multi_bev" / "binary").glob("*.bin"))  # nothing was produced by some other path
