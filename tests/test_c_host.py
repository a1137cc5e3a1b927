"""The plain-C host programs (host/drone_host.c, host/drone_host_mp.c): build
against the C-ABI with gcc alone, fail loudly without a GPU, and on a GPU produce
the oracle's outputs — compared through a CRC-32 chained over every step's
observations / rewards / terminals / truncations, for all four tasks."""
import json
import os
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "drone_host")
EXE_MP = os.path.join(ROOT, "host", "drone_host_mp")


@pytest.fixture(scope="module")
def exe(hip):
    subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "-B"], check=True, capture_output=True)
    return EXE


def oracle_crc(oracle, task, envs, steps, seed, rollout=0):
    """What the C hosts print with --crc 1, from the CPU oracle."""
    o = oracle.OracleVec(envs, seed=seed, cfg=oracle.default_config(task), threads=4)
    o.reset(seed)
    crc = zlib.crc32(o.observations.tobytes())
    launches = steps if not rollout else (steps + rollout - 1) // rollout
    for _ in range(launches):
        if rollout:
            o.rollout(rollout)
        else:
            o.fill_random_actions()
            o.step()
        for buf in (o.observations, o.rewards, o.terminals, o.truncations):
            crc = zlib.crc32(buf.tobytes(), crc)
    n = o.log()["n"]
    o.close()
    return crc, n


def test_c_host_builds_and_refuses_without_gpu(exe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--envs", "16", "--steps", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "drone_vec_init failed" in r.stderr
    r = subprocess.run([EXE_MP, "--envs", "16", "--steps", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "drone_vec_init failed" in r.stderr


def test_c_host_rejects_unknown_task(exe):
    r = subprocess.run([exe, "--task", "4"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown task" in r.stderr
    r = subprocess.run([EXE_MP, "--task", "9"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown task" in r.stderr


@pytest.mark.gpu
def test_c_host_runs(exe):
    r = subprocess.run([exe, "--envs", "8192", "--steps", "256", "--rollout", "64", "--task", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert lines[0]["env_steps_per_s"] > 0 and lines[1]["env_steps_per_s"] > 0
    assert lines[2]["log"]["n"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("task", [0, 1, 2, 3])
def test_c_host_crc_matches_oracle(exe, oracle, task):
    """24-float rows for tasks 2 and 3 (the round-1 host under-allocated them)."""
    envs, steps, seed = 4096, 300, 5
    r = subprocess.run([exe, "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed), "--crc", "1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    want, n = oracle_crc(oracle, task, envs, steps, seed)
    assert got["crc32"] == want, f"task {task}: C host crc {got['crc32']:#x} != oracle {want:#x}"
    assert got["episodes"] == n and n > 0


@pytest.mark.gpu
@pytest.mark.parametrize("task,many", [(0, 7), (1, 32), (2, 5), (3, 300)])
def test_c_host_step_many_crc_matches_oracle(exe, oracle, task, many):
    """The plain-C host stepping through drone_vec_step_many (host blocks, K env steps per call, a ragged last call):
    the CRC chained over every step's observations / rewards / flags equals the per-step one — and the oracle's."""
    envs, steps, seed = 4000, 300, 5
    r = subprocess.run([exe, "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed), "--crc", "1", "--many", str(many)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    want, n = oracle_crc(oracle, task, envs, steps, seed)
    assert got["steps_per_call"] == many and got["crc32"] == want, f"task {task}: C host crc {got['crc32']:#x} != oracle {want:#x}"
    assert got["episodes"] == n and n > 0


@pytest.mark.gpu
@pytest.mark.parametrize("task,rollout", [(0, 0), (1, 0), (3, 0), (0, 32)])
def test_c_host_mp_gather_matches_oracle(exe, oracle, task, rollout):
    """One rank on the 1-GPU box: fork-before-HIP, RCCL communicator bootstrapped
    from C, the gathered batch of every launch CRC'd against the oracle."""
    envs, steps, seed = 8192, 96, 11
    cmd = [EXE_MP, "--gpus", "1", "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed), "--crc", "1", "--gather", "1"]
    if rollout:
        cmd += ["--rollout", str(rollout)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    got = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    want, _ = oracle_crc(oracle, task, envs, steps, seed, rollout=rollout)
    assert got["crc32"] == want, f"task {task}: gathered crc {got['crc32']:#x} != oracle {want:#x}"
    assert got["env_steps_per_s"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,envs", [(2, 6000), (3, 10001)])
def test_c_host_mp_forks_shards_on_one_gpu(exe, oracle, ranks, envs):
    """Several ranks forked before HIP, sharing the one GPU (no gather: RCCL refuses two ranks on a device): every
    rank's own slice, CRC'd over all launches, equals the oracle run of that shard — ragged split, global env ids."""
    steps, seed, task = 64, 23, 1
    cmd = [EXE_MP, "--gpus", str(ranks), "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed),
           "--crc", "1", "--gather", "0", "--share-devices", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert got["gpus"] == ranks and len(got["rank_crc32"]) == ranks
    off = 0
    for rank in range(ranks):
        cnt = envs // ranks + (1 if rank < envs % ranks else 0)
        o = oracle.OracleVec(cnt, seed=seed, cfg=oracle.default_config(task, env_offset=off), threads=4)
        o.reset(seed)
        crc = zlib.crc32(o.observations.tobytes())
        for _ in range(steps):
            o.fill_random_actions()
            o.step()
            for buf in (o.observations, o.rewards, o.terminals, o.truncations):
                crc = zlib.crc32(buf.tobytes(), crc)
        o.close()
        assert got["rank_crc32"][rank] == crc, f"rank {rank} (envs {off}..{off + cnt})"
        off += cnt
