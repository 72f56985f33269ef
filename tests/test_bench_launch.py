"""bench.py --gpus N: the N > 1 line is measured on N ranks or not at all.

The reference is single-device (scripts/training_M2.py:31-33); BASELINE.json config 5 (M2 data-parallel on 8 GPUs) is this build's
own claim, and the driver's command line is `python3 bench.py --gpus N ...` -- so bench.py must start the ranks itself when no launcher
did, and must refuse (exit code != 0) whatever would print a line for fewer ranks or fewer devices than asked for.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "DVAE_DIST_BACKEND"), timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_world_size_that_is_not_gpus_is_an_error():
    for ws, n in (("2", "8"), ("1", "8"), ("2", "1")):
        r = _run(["--gpus", n], {"WORLD_SIZE": ws, "RANK": "0", "LOCAL_RANK": "0"})
        assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and '"metric"' not in r.stdout, (ws, n, r.stderr)


def test_fewer_devices_than_ranks_under_rccl_is_an_error():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and "needs 8 visible GPUs" in r.stderr and '"metric"' not in r.stdout, r.stderr
    # a rank started by a launcher refuses as well (the launcher's own environment, fewer devices than ranks)
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29499"})
    assert r.returncode != 0 and '"metric"' not in r.stdout, r.stderr


@pytest.mark.gpu
def test_two_ranks_without_a_launcher_print_one_two_rank_line():
    """One-GPU box: the gloo rehearsal backend lets the two ranks share the device (not a scaling number -- the flow)."""
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--prewarm-ms", "20"], {"DVAE_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["multi_gpu"]["nranks"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["global_frames_per_step"] == 2 * rec["config"]["frames_per_step_per_gpu"]
    assert rec["steps"] == 5 and rec["warmup"] == 2


@pytest.mark.gpu
def test_two_ranks_over_rccl_on_one_device_exit_nonzero():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the RCCL run is legitimate here")
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline"])
    assert r.returncode != 0 and '"metric"' not in r.stdout
