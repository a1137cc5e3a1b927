"""The plain-C host program (host/drone_host.c): builds against the C-ABI with
gcc alone, fails loudly without a GPU, runs on one."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "drone_host")


@pytest.fixture(scope="module")
def exe(hip):
    subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "-B"], check=True, capture_output=True)
    return EXE


def test_c_host_builds_and_refuses_without_gpu(exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--envs", "16", "--steps", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "drone_vec_init failed" in r.stderr


@pytest.mark.gpu
def test_c_host_runs(exe):
    r = subprocess.run([exe, "--envs", "8192", "--steps", "256", "--rollout", "64", "--task", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert lines[0]["env_steps_per_s"] > 0 and lines[1]["env_steps_per_s"] > 0
    assert lines[2]["log"]["n"] > 0
