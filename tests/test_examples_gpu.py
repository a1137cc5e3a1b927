"""The examples run (small sizes, as subprocesses): they are documentation that must not rot."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_torch_policy_loop_example(hip):
    out = run("torch_policy_loop.py", "--envs", "4096", "--steps", "40")
    assert "captured as one hipGraph" in out and "frame skip 4" in out


@pytest.mark.gpu
def test_dlpack_loop_example(hip):
    subprocess.run(["make", "-C", os.path.join(ROOT, "bindings")], check=True, capture_output=True)
    out = run("dlpack_loop.py", "--envs", "4096", "--steps", "40", "--task", "2")
    assert "library-owned HBM buffers through DLPack" in out


@pytest.mark.gpu
def test_host_buffer_loop_example(hip):
    subprocess.run(["make", "-C", os.path.join(ROOT, "bindings")], check=True, capture_output=True)
    out = run("host_buffer_loop.py", "--envs", "2048", "--steps", "60", "--task", "1")
    assert "host transport 2" in out and "vec_send, policy, vec_recv" in out  # heap arrays of a small shard: pinned stand-ins
