"""bench.py --gpus N without a launcher: the script re-executes itself under torch.distributed.run ("make the scaling run
possible").  CPU: the launch line and the refusal when the box has fewer GPUs; GPU (needs >= 2 devices): a real 2-rank run over RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_spawn_command_is_the_drivers_launch_line():
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_gpus_n_without_launcher_spawns_and_refuses_missing_devices():
    """No WORLD_SIZE in the environment and fewer devices than asked for: one JSON error line, exit code 2 -- not a SystemExit message
    asking for an external launcher."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices; the gpu test below runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["device_count"] == torch.cuda.device_count() and "error" in line


@pytest.mark.gpu
def test_bench_two_gpus_self_spawned():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on the box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["per_rank_images_per_s"]) == 2 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["rccl_version"]
