"""bench.py's one-line JSON contract (driver-facing), on a real GPU at a small size."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields(hip):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "20", "--envs-per-gpu", "65536",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 400 and d["warmup"] == 20
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert 0.0 < rf["frac"] < 1.0
    # round 4: `roofline` is the kernel BEYOND the Infinity Cache (2^22 envs, timed live): its own bytes / its own launch time
    assert rf["envs"] == 1 << 22 and rf["infinity_cache_assisted"] is False
    # the PMC traffic of that kernel, measured by this very run (two rocprofv3 child passes) — or the committed passes if the
    # profiler was not usable; either way within a few per cent of the algorithmic bytes
    assert rf["traffic_measured_in_this_run"] in (True, False) and rf["traffic"] > 0 and rf["traffic"] == rf["traffic_detail"]["hbm_bytes_per_launch"]  # the contract's number: bytes per launch
    assert 0.97 < rf["traffic_over_algorithmic"] < 1.05, rf["traffic_detail"]
    if rf["traffic_measured_in_this_run"]:
        # rocprofv3's kernel durations against HIP events over the SAME launches (the profiled child's own line): the tight check
        # (events see the ~1 us between launches too). Against THIS process's events only loosely: two processes on one box place
        # their 1.1 GB differently and time this HBM-bound kernel up to 10 % apart (0.746 against 0.825 in one run of round 5;
        # profiles/README.md) — and each now picks its own sweep order at its first reset.
        td = rf["traffic_detail"]
        assert td["rocprof_kernel_avg_us"] == pytest.approx(td["same_process_hip_event_launch_us"], rel=0.03), td
        assert rf["frac_from_rocprof_kernel_avg"] == pytest.approx(rf["frac"], rel=0.15), (rf["frac_from_rocprof_kernel_avg"], rf["frac"])
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_env_step"] * rf["envs"] / (rf["launch_us"] * 1e-6) / 1e9)
    assert rf["frac_2pow22"] == rf["frac"] and rf["frac_of_measured_copy_peak"] == pytest.approx(rf["achieved"] / 6290.0)
    pts = rf["beyond_infinity_cache"]
    assert pts[str(1 << 22)]["envs"] == 1 << 22 and pts[str(1 << 22)]["frac"] == rf["frac"]
    assert str(1 << 23) in pts and ("frac" in pts[str(1 << 23)] or "skipped" in pts[str(1 << 23)])
    # the size `value` is measured at, named for what it is; value and that figure come from the same launches:
    # bytes/launch / launch time ~ value * bytes per env-step (value is wall-clock between the fences, achieved is HIP-event
    # time: 400 x 5 us of launches keep the fences' share small)
    am = rf["at_metric_size"]
    assert am["envs"] == 65536 and am["frac_incl_infinity_cache"] == rf["frac_incl_infinity_cache"] == pytest.approx(am["achieved"] / 8000.0)
    assert am["achieved"] * 1e9 == pytest.approx(d["value"] * am["algorithmic_bytes_per_env_step"], rel=0.3)
    # round 5: the timed window is steady state and says so — a fixed untimed pre-roll ahead of --warmup, the episodes that ended
    # inside the timed launches (vec_log deltas), the working set from the handle's own byte count
    assert d["pre_roll_steps"] == 512 and d["episodes_in_timed_window"] > 0
    assert d["episode_count_window_launches"] == 420  # counted over the warm-up launches too (the drain sits ahead of them: nothing but launches between it and the clock), pro-rated
    assert d["episode_ends_per_env_step"] == pytest.approx(d["episodes_in_timed_window"] / (65536 * 400), rel=0.01) and 0.002 < d["episode_ends_per_env_step"] < 0.02
    assert am["working_set_bytes_per_step"] == am["algorithmic_bytes_per_env_step"] * 65536 and "285 MB" not in am["note"]
    sm131 = d["configs"]["configs[2]/shard"]["step_many"]["K8"]
    assert sm131["timed_stream_ms"] >= 15.0 and sm131["warmup_launches"] >= 30 and sm131["timed_launches"] >= 64
    # which kernel instantiation every timed handle ran
    assert set(d["variants"]) >= {"hover:65536", "hover:4194304", "hover:1024", "hover:131072", "waypoint:262144"}
    big = d["variants"]["hover:4194304"]  # the footprint table's entry (round 6: the online measurement is opt-in, DRONE_AUTOTUNE=1)
    assert "dt=1" in big and " order=8 " in big and "mem=2" in big and "dt=0" in d["variants"]["hover:65536"]
    assert not any("autotuned" in text for text in d["variants"].values())
    # round 3: every other single-GPU BASELINE workload timed in the same run, each with its own bytes
    cf = d["configs"]
    assert set(cf) == {"configs[0]", "configs[1]", "configs[2]/shard", "configs[3]"}
    c0 = cf["configs[0]"]  # BASELINE's CPU plumbing case: 1024 envs on one host thread, and the same through the HIP path
    assert c0["envs"] == 1024 and c0["cpu"]["threads"] == 1 and c0["cpu"]["env_steps_per_s"] > 1e6
    assert c0["per_step"]["launch_us"] > 0 and c0["host_buffers"]["transport"] == "zero-copy" and c0["host_buffers"]["wall_us_per_step"] > 0
    for name, envs, task, nbytes in (("configs[1]", 65536, "hover", 278), ("configs[2]/shard", 131072, "hover", 278), ("configs[3]", 262144, "waypoint", 310)):
        c = cf[name]
        assert "skipped" not in c, c
        ps = c["per_step"]
        assert c["envs"] == envs and c["task"] == task and ps["algorithmic_bytes_per_env_step"] == nbytes
        assert ps["launch_us"] > 0 and 0.0 < ps["frac"] < 1.3  # 262 144 envs run from the Infinity Cache: above the HBM line by design
        assert ps["achieved_GBps"] == pytest.approx(nbytes * envs / (ps["launch_us"] * 1e-6) / 1e9)
    assert cf["configs[1]"]["fused_rollout"]["roofline"]["bound"] == "valu-f32"
    sm = cf["configs[1]"]["step_many"]
    assert sm["K32"]["algorithmic_bytes_per_env_step"] == pytest.approx(102 + 176 / 32) and sm["K32"]["us_per_env_step_launch_avg"] < ps_limit(cf)
    fr = d["fused_rollout"]["roofline"]
    assert fr["bound"] == "valu-f32" and fr["unit"] == "TFLOP/s" and fr["peak"] == 157.3
    assert fr["frac"] == pytest.approx(fr["achieved"] / fr["peak"]) and 0.0 < fr["frac"] < 1.0
    assert 0.0 < fr["frac_of_measured_issue_rate"] < 1.05 and fr["valu_per_wave_step"] < 500
    assert d["rccl_ranks"] == 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"].startswith("1048576 envs")  # at the metric's N
    assert cb["cache_resident_sample"]["value"] > 0
    assert d["value"] > cb["value"]


def ps_limit(cf):
    """K steps per launch must beat one launch per step at 65 536 envs (that is what it is for)."""
    return cf["configs[1]"]["per_step"]["launch_us"]


CORE = {"n1_same_box", "no_gather", "gather_step", "rollout_no_gather", "rollout_gather"}
OPTIONAL = {"gather_root_step", "gather_peer_store", "rollout_gather_peer_store", "gather_overlap", "rollout_gather_root"}  # + gather_step_cabi, rollout_gather_overlap on RCCL


def check_two_rank_line(r, optional_ok=True):
    assert r.returncode == 0, r.stderr[-3000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    assert last.startswith("{"), "the JSON line must be the LAST line on stdout"
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 6 and d["warmup"] == 2
    want = CORE | {"c_host_mp"} | (OPTIONAL if optional_ok else set())
    assert set(d["records"]) == want, set(d["records"]) ^ want
    # `value` is BASELINE configs[2]: the per-step kernel WITH the host-boundary all-gather
    assert d["value_from"] == "gather_step" and d["value"] == d["records"]["gather_step"]["env_steps_per_s"]
    assert d["ms_per_step"] == d["records"]["gather_step"]["ms_per_step"]
    n1 = d["records"]["n1_same_box"]
    assert n1["envs"] == 131073 and n1["gpus"] == 1 and n1["per_step"]["env_steps_per_s"] > 0 and n1["rollout"]["env_steps_per_s"] > 0
    assert d["scaling_vs_n1"] == pytest.approx(d["value"] / n1["per_step"]["env_steps_per_s"])
    sec = d["secondary_values"]
    assert sec["no_gather"]["scaling_vs_n1"] == pytest.approx(sec["no_gather"]["value"] / n1["per_step"]["env_steps_per_s"])
    assert sec["rollout_gather"]["scaling_vs_n1"] == pytest.approx(sec["rollout_gather"]["value"] / n1["rollout"]["env_steps_per_s"])
    assert "gather_step" not in sec  # it is `value`
    ch = d["records"]["c_host_mp"]  # the plain-C multi-process host, run as a child with a timeout after the line was complete
    assert ch["per_step"]["gpus"] == 2 and ch["per_step"]["envs"] == 131073 and ch["per_step"]["env_steps_per_s"] > 0
    assert ch["rollout"]["horizon"] == 128 and ch["rollout"]["env_steps_per_s"] > 0
    check_preflight(d, 2)
    return d


PREFLIGHT_ITEMS = {"devices", "peer_access", "flag_page_host", "ipc_store_roundtrip", "nccl_allgather_1mib", "cabi_rccl_gather"}


def check_preflight(d, world, failed=()):
    """the ~10 s preflight child job between the core records and the optional ones, in rank 0's line. On the
    1-GPU box the ranks share the device over gloo: the items that need one device per rank say so, the rest must pass — the
    flag page seen by every rank, and rows stored through an IPC mapping of the root's batch arriving bit for bit."""
    pf = d["preflight"]
    assert pf["status"].startswith("ok"), pf
    assert PREFLIGHT_ITEMS <= set(pf), PREFLIGHT_ITEMS - set(pf)
    assert pf["devices"]["ranks"] == world and pf["devices"]["per_rank"] == [1] * world and pf["devices"]["backend"] == "gloo"
    for item in ("peer_access", "nccl_allgather_1mib", "cabi_rccl_gather"):
        assert pf[item].startswith("n/a:") and "1 device" in pf[item], (item, pf[item])
    for item in ("flag_page_host", "ipc_store_roundtrip"):
        assert pf[item].startswith("failed" if item in failed else "ok"), (item, pf[item])
    assert float(pf["status"].split("(")[1].split()[0]) < 30.0, pf["status"]  # "ok (9 s)": what it adds to the job


@pytest.mark.gpu
def test_plain_python_launch_with_gpus_2_starts_its_own_ranks(hip):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts the ranks as a child process (before touching
    HIP), relays rank 0's line last and returns the children's exit code. The optional exchanges run
    as a second, fresh child job and are merged in."""
    env = {k: val for k, val in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--total-envs", "131073"],
                       capture_output=True, text=True, timeout=900, env=env)
    d = check_two_rank_line(r)
    assert "rccl_ranks" in d and "warning" in d and d["optional"].startswith("ok")
    # the peer-store exchange ran between two processes sharing the GPU (IPC-mapped buffers, flag page in /dev/shm)
    assert d["records"]["gather_peer_store"]["env_steps_per_s"] > 0 and d["records"]["rollout_gather_peer_store"]["env_steps_per_s"] > 0
    assert d["host_boundary_gather"]["gather_peer_store"]["ms_per_step"] > 0


def test_workload_argv():
    """What the optional child job inherits from its parent's command line: the workload, nothing else."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.workload_argv(["--gpus", "8", "--steps", "20", "--warmup", "5"]) == ["--steps", "20", "--warmup", "5"]
    assert bench.workload_argv(["--gpus=8", "--steps=20", "--value-from", "no_gather", "--task", "race", "--force-dist", "--total-envs=4096"]) == [
        "--steps=20", "--task", "race", "--total-envs=4096"]
    assert bench.workload_argv(["--phase", "core", "--optional-timeout", "45", "--seed", "7", "--no-extras", "--ring", "2"]) == ["--seed", "7", "--ring", "2"]
    assert bench.workload_argv([]) == []
    # every record a preflight item can gate is an optional record of the line (a renamed record must not silently lose its gate)
    gated = {name for names in bench.PREFLIGHT_GATES.values() for name in names}
    assert gated <= OPTIONAL | {"gather_step_cabi", "rollout_gather_overlap"}, gated - OPTIONAL
    assert set(bench.PREFLIGHT_GATES) <= PREFLIGHT_ITEMS
    # every handle bench.py says it times is listed once
    assert len({(t, n) for t, n, _ in bench.TIMED_HANDLES}) == len(bench.TIMED_HANDLES)


def test_self_launch_returns_the_childrens_failure_without_a_gpu():
    """No GPU here: the ranks die on their first assert; the launcher must come back non-zero (and must not hang or print a line)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    env = {k: val for k, val in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-envs", "1000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_two_rank_launch_reports_the_metrics_configuration(hip):
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on a 1-GPU box: the oversubscribed gloo smoke path.
    2^20-style strong split (here 2^17 total to keep it short), the named records, `value` = configs[2] (per-step kernel +
    host-boundary gather), everything else in secondary_values with its scaling against the same box's one-GPU run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--total-envs", "131073"],
                       capture_output=True, text=True, timeout=900, env=env)
    d = check_two_rank_line(r)
    assert d["optional"].startswith("ok")
    assert d["host_boundary_gather"]["gather_step"]["env_steps_per_s"] == d["records"]["gather_step"]["env_steps_per_s"]
    assert "rollout_gather" in d["host_boundary_gather"] and "gather_root_step" in d["host_boundary_gather"]
    assert d["config"]["envs_per_gpu"] == 65537  # ragged split of 131073: rank 0 takes the extra env (shard_range)
    for name, rec in d["records"].items():
        if name not in ("c_host_mp", "n1_same_box"):
            assert rec["env_steps_per_s"] > 0 and rec["ms_per_step"] > 0, (name, rec)
    assert d["records"]["rollout_gather"]["horizon"] == 128
    assert "rccl_ranks" in d and "warning" in d  # gloo smoke path on one GPU: flagged as not a measurement


@pytest.mark.gpu
def test_a_failed_preflight_item_skips_the_records_that_rest_on_it(hip):
    """an item the preflight finds broken (here injected on the last rank) turns the optional records that rest
    on it into "skipped: preflight <item>" — by name, on every rank alike, at once — while the other optional records run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRONE_BENCH_FAULT="preflight:ipc_store_roundtrip")
    env = {k: val for k, val in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--total-envs", "131073"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    check_preflight(d, 2, failed=("ipc_store_roundtrip",))
    assert "injected" in d["preflight"]["ipc_store_roundtrip"] and "rank 1" in d["preflight"]["ipc_store_roundtrip"]
    for name in ("gather_peer_store", "rollout_gather_peer_store"):
        assert d["records"][name]["skipped"].startswith("preflight ipc_store_roundtrip"), d["records"][name]
    assert d["optional"].startswith("ok") and d["records"]["gather_root_step"]["env_steps_per_s"] > 0 and d["value"] > 0
    assert "gather_peer_store" not in d["secondary_values"]


@pytest.mark.gpu
@pytest.mark.parametrize("fault,how,pg_timeout", [("hang:gather_root_step", "timed out", "600"), ("die:gather_peer_store", "failed", "25")])
def test_a_rank_lost_in_an_optional_record_cannot_cost_the_line(hip, fault, how, pg_timeout):
    """one rank of the OPTIONAL job hangs (never enters the record's collectives) or dies mid-record.
    The core job's rank 0 kills the child job after --optional-timeout (or sees it fail) and still prints a valid line
    with rc 0: `value` from configs[2]'s record, the core records complete, `optional` saying what happened."""
    # hang: the healthy rank sits in the record's collective (its own timeout far away) until the PARENT's timeout kills the job;
    # die: the launcher sees a rank exit and tears the child job down by itself
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRONE_BENCH_FAULT=fault, DRONE_BENCH_PG_TIMEOUT=pg_timeout)
    env = {k: val for k, val in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--total-envs", "131073",
                        "--optional-timeout", "45"], capture_output=True, text=True, timeout=900, env=env)
    d = check_two_rank_line(r, optional_ok=False)
    assert how in d["optional"], d["optional"]
    assert d["value"] > 0 and d["secondary_values"]["no_gather"]["value"] > 0


@pytest.mark.gpu
def test_eight_ranks_oversubscribed_at_the_metrics_shape(hip):
    """the driver's own multi-GPU command — `python bench.py --gpus 8 --steps 20 --warmup 5`, 2^20 envs
    as eight shards of 131 072 — rehearsed on the ONE GPU there is (eight ranks over gloo): every rank-count-dependent path
    of the line (shard offsets, the n1_same_box handle beside rank 0's shard, eight-way gathers, the optional child job's
    budget, the C host with eight processes) runs before an 8-GPU node ever sees it. Not a measurement (`warning` says so);
    what is checked is that the line arrives, complete, well inside the driver's patience."""
    import time

    env = {k: val for k, val in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for attempt in (1, 2):
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, timeout=900, env=env)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        last = [l for l in r.stdout.splitlines() if l.strip()][-1]
        assert last.startswith("{"), "the JSON line must be the LAST line on stdout"
        d = json.loads(last)
        print(f"bench.py --gpus 8 oversubscribed on one GPU (attempt {attempt}): {wall:.0f} s wall, optional: {d.get('optional')}")
        # Eight processes time-slicing ONE GPU: an optional exchange record can lose its budget to the scheduler (seen once in eight
        # full runs of round 5; the line itself arrived). The core of the line is never retried; a second miss of the optional job fails.
        if attempt == 2 or str(d.get("optional", "")).startswith("ok"):
            break
    assert wall < 600, f"{wall:.0f} s"
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 20 and d["warmup"] == 5
    assert d["value_from"] == "gather_step" and d["value"] == d["records"]["gather_step"]["env_steps_per_s"] > 0
    assert d["config"]["envs_per_gpu"] == 131072 and "1048576 envs in total" in d["config"]["workload"]
    n1 = d["records"]["n1_same_box"]
    assert n1["envs"] == 1 << 20 and n1["per_step"]["env_steps_per_s"] > 0 and n1["rollout"]["env_steps_per_s"] > 0
    assert CORE <= set(d["records"])
    assert "timed out" not in d["optional"] and d["optional"].startswith("ok"), d["optional"]
    pf = d["preflight"]  # eight ranks on the one GPU: the IPC store + flag round trip with seven peers, bit for bit
    assert pf["status"].startswith("ok") and pf["ipc_store_roundtrip"].startswith("ok") and pf["flag_page_host"].startswith("ok") and pf["devices"]["ranks"] == 8, pf
    assert OPTIONAL <= set(d["records"]), OPTIONAL - set(d["records"])
    for name in CORE | OPTIONAL:
        if name != "n1_same_box":
            assert d["records"][name].get("env_steps_per_s", 0) > 0, (name, d["records"][name])
    assert d["records"]["gather_peer_store"]["env_steps_per_launch_per_gpu"] == 131072
    assert d["records"]["rollout_gather"]["horizon"] == 128 and d["records"]["rollout_gather_peer_store"]["env_steps_per_launch_per_gpu"] == 131072 * 128
    ch = d["records"]["c_host_mp"]
    assert ch["per_step"]["gpus"] == 8 and ch["per_step"]["envs"] == 1 << 20 and ch["rollout"]["env_steps_per_s"] > 0
    assert "warning" in d and d["roofline"]["envs"] == 131072
    assert 0 < d["job_wall_s"] <= wall  # the line's own account of how long rank 0's job took (DESIGN.md quotes it)
