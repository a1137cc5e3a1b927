"""bench.py's one-line JSON contract (driver-facing), on a real GPU at a small size."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields(hip):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5", "--envs-per-gpu", "65536",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 50 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert 0.0 < rf["frac"] < 1.0
    # value and the roofline come from the same launches: bytes/launch / launch time ~ value * bytes per env-step
    assert rf["achieved"] * 1e9 == pytest.approx(d["value"] * rf["algorithmic_bytes_per_env_step"], rel=0.2)
    # round 2: the cache-free HBM fraction beside the headline one, and the config-5 roofline of the fused rollout
    assert 0.0 < rf["frac_hbm_only"] < 1.0 and rf["hbm_only"]["envs"] >= 1 << 22
    fr = d["fused_rollout"]["roofline"]
    assert fr["bound"] == "valu-f32" and fr["unit"] == "TFLOP/s" and fr["peak"] == 157.3
    assert fr["frac"] == pytest.approx(fr["achieved"] / fr["peak"]) and 0.0 < fr["frac"] < 1.0
    assert 0.0 < fr["frac_of_measured_issue_rate"] < 1.05 and fr["valu_per_wave_step"] < 500
    assert d["rccl_ranks"] == 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > cb["value"]


@pytest.mark.gpu
def test_two_rank_launch_reports_the_metrics_configuration(hip):
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on a 1-GPU box: the oversubscribed gloo smoke path.
    2^20-style strong split (here 2^17 total to keep it short), the four named records, `value` = configs[2]."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--total-envs", "131073"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 6 and d["warmup"] == 2
    assert set(d["records"]) == {"no_gather", "gather_step", "gather_overlap", "rollout_no_gather", "rollout_gather"}  # + rollout_gather_overlap on RCCL
    assert d["value_from"] == "gather_step" and d["value"] == d["records"]["gather_step"]["env_steps_per_s"]
    assert d["ms_per_step"] == d["records"]["gather_step"]["ms_per_step"]
    assert d["config"]["envs_per_gpu"] == 65537  # ragged split of 131073: rank 0 takes the extra env (shard_range)
    for rec in d["records"].values():
        assert rec["env_steps_per_s"] > 0 and rec["ms_per_step"] > 0
    assert d["records"]["rollout_gather"]["horizon"] == 128
    assert "rccl_ranks" in d and "warning" in d  # gloo smoke path on one GPU: flagged as not a measurement
