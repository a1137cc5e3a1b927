"""AddressSanitizer + UBSan on the CPU builds (GPU sanitizers are not available on this pool): the scalar oracle
(oracle/asan_driver.c) and the PRODUCT's per-lane math host-compiled (tests/lane_host/san_driver.cpp: all four tasks,
wrapping env ids and step counters, NaN / infinite / denormal / huge states and actions)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean(r):
    assert r.returncode == 0, r.stdout + r.stderr
    for needle in ("runtime error", "AddressSanitizer", "LeakSanitizer"):
        assert needle not in r.stderr, r.stderr[-2000:]


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "oracle_asan"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ROOT, "oracle", "oracle_asan")], capture_output=True, text=True, timeout=300)
    _clean(r)
    assert r.stdout.count("task ") == 4


def test_lane_math_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "lane_san")
    d = os.path.join(ROOT, "tests", "lane_host")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", os.path.join(d, "lane_host.cpp"), os.path.join(d, "san_driver.cpp"),
                    "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    _clean(r)
    assert r.stdout.count("episode ends") == 4
