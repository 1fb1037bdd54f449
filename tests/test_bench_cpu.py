"""bench.py's launch logic: `python bench.py --gpus N` starts N ranks itself and never prints an
N = 1 number for an N > 1 request.  Replaces the reference's single-process multi-GPU entry (main.py:99)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    assert bench.launch_plan(1, ["--steps", "3"], {}, 1) is None


def test_rank_of_a_torchrun_launch_runs_in_process():
    assert bench.launch_plan(8, ["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "3"}, 8) is None


def test_world_size_mismatch_is_refused():
    with pytest.raises(SystemExit):
        bench.launch_plan(4, [], {"WORLD_SIZE": "8"}, 8)
    with pytest.raises(SystemExit):
        bench.launch_plan(1, [], {"WORLD_SIZE": "2"}, 8)


def test_too_few_gpus_is_refused_loudly():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(2, ["--gpus", "2"], {}, 1)
    assert "1 GPU" in str(e.value)


def test_plain_command_launches_one_rank_per_gpu():
    cmd = bench.launch_plan(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], {}, 8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_shared_device_test_mode_waives_the_device_count():
    cmd = bench.launch_plan(2, ["--gpus", "2"], {"DRN_FORCE_DEVICE": "0", "DRN_DIST_BACKEND": "gloo"}, 1)
    assert cmd is not None and cmd[cmd.index("--nproc-per-node") + 1] == "2"


def test_cli_refuses_two_gpus_on_a_box_without_them():
    """End to end on this GPU-less container: exit code != 0 and no JSON line on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DRN_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "{" not in r.stdout
    assert "refusing" in r.stderr
