"""The rank > 0 side of the C-ABI gather on a ONE-GPU box. RCCL refuses two ranks on one device, so these tests load
tests/rccl_stub (a shared-memory test double with RCCL's documented semantics, selected with DRONE_RCCL_LIB) and run
2 and 3 ranks that share the GPU: slice offsets, ragged counts (the all-gather-v branch), in-place device buffers,
host staging. What stays untested without a multi-GPU box is RCCL itself, not this library's use of it."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from helpers import usable_cores

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE_MP = os.path.join(ROOT, "host", "drone_host_mp")


@pytest.fixture(scope="module")
def stub(tmp_path_factory, hip):
    out = str(tmp_path_factory.mktemp("stub") / "librccl_stub.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "rccl_stub", "rccl_stub.c"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib", "-o", out], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "-B"], check=True, capture_output=True)
    return out


def oracle_crcs(oracle, envs, task, seed, steps, rollout, ranks, threads=4):
    """What host/drone_host_mp --crc 1 must print: the CRC-32 chained over the whole batch after the reset (observations) and
    after every launch (all four buffers), from ONE oracle run over all envs — and the same chain over each rank's own rows."""
    o = oracle.OracleVec(envs, seed=seed, cfg=oracle.default_config(task), threads=threads)
    o.reset(seed)
    cuts = [(r * (envs // ranks) + min(r, envs % ranks), envs // ranks + (1 if r < envs % ranks else 0)) for r in range(ranks)]
    crc = zlib.crc32(o.observations.tobytes())
    own = [zlib.crc32(o.observations[a:a + c].tobytes()) for a, c in cuts]
    launches = steps if not rollout else (steps + rollout - 1) // rollout
    for _ in range(launches):
        if rollout:
            o.rollout(rollout)
        else:
            o.fill_random_actions()
            o.step()
        for buf in (o.observations, o.rewards, o.terminals, o.truncations):
            crc = zlib.crc32(buf.tobytes(), crc)
            own = [zlib.crc32(buf[a:a + c].tobytes(), x) for x, (a, c) in zip(own, cuts)]
    o.close()
    return crc, own


# BASELINE configs[2] and configs[4] at their REAL shape — eight ranks x 131 072 hover envs — on the one GPU there is
# (VERDICT r4 item 1): eight processes, eight communicator slots, shard offsets at 2^17 granularity, 10.5 MB of
# observations per rank and exchange, the ragged all-gather-v with eight broadcasts per buffer. What an 8-GPU node adds is
# RCCL itself and xGMI; everything of this library that depends on the rank count runs here.
@pytest.mark.parametrize("envs,rollout,root", [(1 << 20, 0, -1), (1 << 20, 0, 0), ((1 << 20) + 5, 0, -1), (1 << 20, 128, -1), (1 << 20, 128, 5)])
def test_world8_at_the_real_shape_c_host_mp(stub, oracle, envs, rollout, root):
    """host/drone_host_mp --gpus 8 --envs 1048576: per-step (40 launches, configs[2]) and fused 128-step rollouts (two launches,
    configs[4]); all-gather, gather to one rank, ragged shards. The gathered batch after EVERY launch is CRC'd by the receiving
    rank and must equal ONE oracle run over all 2^20 envs; each rank's own rows likewise (so a failure names the rank)."""
    ranks, seed = 8, 41
    steps = 256 if rollout else 40
    cmd = [EXE_MP, "--gpus", str(ranks), "--envs", str(envs), "--steps", str(steps), "--task", "0", "--seed", str(seed),
           "--crc", "1", "--gather", "1", "--share-devices", "1", "--root", str(root), "--timeout", "280"]
    if rollout:
        cmd += ["--rollout", str(rollout)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, DRONE_RCCL_LIB=stub))
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    got = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert got["gpus"] == 8 and got["envs"] == envs and got["root"] == root
    crc, own = oracle_crcs(oracle, envs, 0, seed, steps, rollout, ranks, threads=usable_cores())
    assert got["rank_crc32"] == own, [r for r in range(ranks) if got["rank_crc32"][r] != own[r]]
    assert got["crc32"] == crc, f"8 ranks x {envs // 8} envs: gathered crc {got['crc32']:#x} != oracle {crc:#x}"


@pytest.mark.parametrize("ranks,envs,task,rollout,root", [(2, 6000, 0, 0, -1), (3, 10001, 1, 0, -1), (2, 4096, 3, 16, -1), (3, 6144, 2, 0, -1),
                                                          (2, 6000, 0, 0, 0), (3, 10001, 1, 0, 2), (3, 6144, 2, 16, 1), (2, 4097, 3, 0, 1)])
def test_c_host_mp_gathers_across_ranks(stub, oracle, ranks, envs, task, rollout, root):
    """host/drone_host_mp with several ranks: fork before HIP, id through the shared page, gather_init with the ranks'
    counts (equal -> all-gather, ragged -> one broadcast per rank), the gathered batch of every launch CRC'd on rank 0
    against ONE oracle run over all envs — so every rank's slice landed at its global offset. root >= 0: the gather to
    one rank (ncclSend / ncclRecv in one group), first and last rank, equal and ragged counts; the CRC is the root's."""
    steps, seed = 48, 31
    cmd = [EXE_MP, "--gpus", str(ranks), "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed),
           "--crc", "1", "--gather", "1", "--share-devices", "1", "--root", str(root)]
    if rollout:
        cmd += ["--rollout", str(rollout)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, DRONE_RCCL_LIB=stub))
    assert r.returncode == 0, r.stderr + r.stdout
    got = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert got["root"] == root
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
    assert got["crc32"] == crc, f"{ranks} ranks: gathered crc {got['crc32']:#x} != oracle {crc:#x}"


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np, torch
from drone_amd import abi, binding
from drone_amd.dist import shard_range
rank, world, total, task, steps, seed = (int(x) for x in sys.argv[1:7])
idfile = sys.argv[7]
root, inplace = int(sys.argv[9]), int(sys.argv[10])
receives = root < 0 or root == rank
dev = torch.device("cuda:0")
off, cnt = shard_range(total, rank, world)
od = abi.obs_dim(task)
rows = total if receives else cnt  # a rank that receives nothing needs no global buffers
g_obs = torch.zeros((rows, od), dtype=torch.float32, device=dev); g_rew = torch.zeros(rows, dtype=torch.float32, device=dev)
g_term = torch.zeros(rows, dtype=torch.uint8, device=dev); g_trunc = torch.zeros(rows, dtype=torch.uint8, device=dev)
act = torch.zeros((cnt, 4), dtype=torch.float32, device=dev)
sl = slice(off, off + cnt) if receives else slice(0, cnt)
if inplace or not receives:
    local = (g_obs[sl], act, g_rew[sl], g_term[sl], g_trunc[sl])
else:  # the root writes into buffers of its own: its rows reach the global buffers by a device copy inside drone_vec_gather
    local = (torch.zeros((cnt, od), dtype=torch.float32, device=dev), act, torch.zeros(cnt, dtype=torch.float32, device=dev),
             torch.zeros(cnt, dtype=torch.uint8, device=dev), torch.zeros(cnt, dtype=torch.uint8, device=dev))
v = binding.DroneVec(cnt, seed=seed, cfg=binding.default_config(task, env_offset=off, horizon=20), buffers=local)
if rank == 0:
    uid = binding.gather_unique_id()
    with open(idfile + ".tmp", "wb") as fh: fh.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    import time
    while not os.path.exists(idfile): time.sleep(0.01)
    uid = open(idfile, "rb").read()
glob = (g_obs, g_rew, g_term, g_trunc) if receives else (None, None, None, None)
v.gather_init(uid, rank, world, *glob, counts=[shard_range(total, r, world)[1] for r in range(world)], root=root)
v.reset(seed); v.gather()
for t in range(steps):
    v.fill_random_actions(); v.step(); v.gather()
torch.cuda.synchronize()
import zlib
arrs = dict(obs=g_obs.cpu().numpy(), rew=g_rew.cpu().numpy(), term=g_term.cpu().numpy(), trunc=g_trunc.cpu().numpy())
if total > 200000:  # the full-size rehearsal: eight ranks x 92 MB of files would be noise; the checksums say the same
    np.savez(sys.argv[8], **dict((k + "_crc", np.uint32(zlib.crc32(a.tobytes()))) for k, a in arrs.items()))
else:
    np.savez(sys.argv[8], **arrs)
v.gather_close(); v.close()
"""


@pytest.mark.parametrize("world,total,task,root,inplace", [(2, 8192, 0, -1, 1), (3, 7001, 1, -1, 1), (2, 8192, 0, 0, 1), (3, 7001, 1, 2, 1), (3, 7001, 3, 1, 0),
                                                           (8, 1 << 20, 0, -1, 1), (8, (1 << 20) + 5, 0, 6, 1)])  # configs[2] at its real shape, all-gather and gather to one rank (ragged)
def test_in_place_device_gather_across_ranks(stub, oracle, tmp_path, world, total, task, root, inplace):
    """Device buffers: every rank's output buffers ARE its slice of its global buffers (the in-place form bench.py's
    C-ABI record uses); after each gather every receiving rank holds the whole batch, identical to one oracle run.
    root >= 0: only that rank receives (the others pass no global buffers); inplace = 0: the root's own rows are
    copied into place on the device."""
    steps, seed = 30, 17
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    idfile = str(tmp_path / "uid")
    outs = [str(tmp_path / f"out{r}.npz") for r in range(world)]
    env = dict(os.environ, DRONE_RCCL_LIB=stub)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(total), str(task), str(steps), str(seed), idfile, outs[r], str(root), str(inplace)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
    o = oracle.OracleVec(total, seed=seed, cfg=oracle.default_config(task, horizon=20), threads=4 if total < 200000 else usable_cores())
    o.reset(seed)
    for _ in range(steps):
        o.fill_random_actions()
        o.step()
    for r in range(world):
        if root >= 0 and r != root:
            continue
        g = np.load(outs[r])
        if "obs_crc" in g.files:
            for k, buf in (("obs", o.observations), ("rew", o.rewards), ("term", o.terminals), ("trunc", o.truncations)):
                assert int(g[k + "_crc"]) == zlib.crc32(buf.tobytes()), f"rank {r}: {k} of the gathered 2^20-env batch"
            continue
        assert g["obs"].tobytes() == o.observations.tobytes(), f"rank {r}: observations"
        assert g["rew"].tobytes() == o.rewards.tobytes(), f"rank {r}: rewards"
        assert g["term"].tobytes() == o.terminals.tobytes() and g["trunc"].tobytes() == o.truncations.tobytes(), f"rank {r}: flags"
