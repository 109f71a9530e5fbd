"""bench.py's command-line contract: `--gpus N` starts its own N ranks (or refuses clearly when the machine has fewer
GPUs), the JSON line carries `roofline` and `cpu_baseline`, and the single-rank RCCL path works under
torch.distributed.run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
BENCH = str(REPO / "bench.py")


def _device_count():
    import torch
    return torch.cuda.device_count()  # does not initialise the GPU


def test_gpus_beyond_the_machine_is_refused_before_any_gpu_call():
    n = _device_count()
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(max(2, n + 1)), "--no-build"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stdout + r.stderr
    assert f"needs {max(2, n + 1)} visible GPUs, this machine shows {n}" in r.stderr
    assert r.stdout.strip() == ""


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


# two sub-batches of the default size: the front of one overlaps the back of the other, as in the full run
SMALL = ["--steps", "2", "--warmup", "1", "--frames", "512", "--cpu-sample", "8"]


@pytest.mark.gpu
def test_json_line_has_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, BENCH, *SMALL], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["unit"] == "frames/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["frames_per_step"] == 512
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["kernel"].startswith("k_")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["pipeline"]["frac"] < 1
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["gpu_output_matches_oracle_on_sampled_frame"] is True
    assert cb["timed_region"]["value"] > 0 and cb["timed_region"]["value"] <= cb["value"] * 1.05
    # the timed region runs fused launches (k_stage: a sub-batch's walk beside the later stages of its neighbours); the
    # one-lane pass behind it runs the same device code as one launch per kernel, which is what the roofline is read from
    assert "k_stage" in {k["name"] for k in d["kernels_pipelined"]}
    assert {"k_walk", "k_cell_sums", "k_ground_resolve", "k_bev_raster", "k_probe"} <= {k["name"] for k in d["kernels"]}
    assert rf["kernel"] == "k_walk"


@pytest.mark.gpu
def test_single_rank_under_torch_distributed_run_goes_through_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", BENCH, "--gpus", "1", *SMALL, "--no-cpu", "--no-profile"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0
    # the line shows by itself what RCCL saw: the world size after init_process_group, the broadcast table, every rank's own figures
    assert d["ranks_seen"] == 1 and d["frame_range_table"] == [[0, 512]]
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["frames"] == 512 and d["per_rank"][0]["frames_per_s"] > 0


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    n = _device_count()
    if n < 2:
        pytest.skip(f"needs 2 GPUs, this box shows {n}; the rank-spawning path is covered up to the launch by "
                    "test_gpus_beyond_the_machine_is_refused_before_any_gpu_call, the collectives by tests/test_distributed_cpu.py (gloo)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", *SMALL, "--no-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 2 * 512 and d["value"] > 0
    assert d["ranks_seen"] == 2 and d["frame_range_table"] == [[0, 512], [512, 512]]
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["frames_per_s"] > 0 for r in d["per_rank"])
