"""The host copy pool of transport 3 (drone_amd/csrc/drone_host_copy.hpp) on the CPU, under ThreadSanitizer: it touches no
HIP, so its hand-offs — a job started by drone_vec_step_send on one thread and finished by drone_vec_step_recv on another, a
second handle that finds it busy, workers that fell asleep between jobs — can be checked where sanitizers run."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "copy_pool_host", "pool_driver.cpp")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_pool_hand_offs_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "pool_driver")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-pthread",
                         "-I", os.path.join(ROOT, "drone_amd", "csrc"), SRC, "-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, DRONE_HOST_COPY_THREADS="4", TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("OK parts=4"), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
