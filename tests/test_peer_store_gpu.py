"""The host-boundary exchange WITHOUT a collective (round 4): peer stores. The root exports its global
observation / reward / flag buffers as IPC handles; every other rank — another PROCESS — maps them and its step kernel's
output stores land in the root's HBM directly; `drone_vec_gather` is only a handshake through a shared page of flags
(two one-wave kernels on the stream, or the host with DRONE_PEER_HOST_WAIT=1). On the 1-GPU box the ranks share the device (IPC between processes works
the same; what an 8-GPU node adds is that the stores cross xGMI): 2 and 3 processes, ragged shards, root first and last,
20- and 24-float rows, per-step and fused launches — the root's batch after EVERY launch must be what one oracle run
over all envs produces, bit for bit. Same pattern as tests/test_gather_multirank_gpu.py, without the RCCL test double."""
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

_WORKER = r"""
import os, sys, time, zlib
sys.path.insert(0, {root!r})
import numpy as np, torch
from drone_amd import abi, binding
from drone_amd.dist import shard_range
rank, world, total, task, steps, seed, root, rollout = (int(x) for x in sys.argv[1:9])
tokfile, flagfile, out = sys.argv[9:12]
dev = torch.device("cuda:0")
off, cnt = shard_range(total, rank, world) if task != 2 else (rank * (total // world), total // world)
od = abi.obs_dim(task)
over = dict(horizon=20, env_offset=off)
if task == 2: over.update(agents_per_env=8, collision_radius=0.5)
v = binding.DroneVec(cnt, seed=seed, cfg=binding.default_config(task, **over), device=dev)
flags = np.memmap(flagfile, dtype=np.uint32, mode="r+", shape=(1024,))   # one shared 4 KiB page (mmap: page-aligned)
counts = [(shard_range(total, r, world)[1] if task != 2 else total // world) for r in range(world)]
if rank == root:
    g = (torch.zeros((total, od), dtype=torch.float32, device=dev), torch.zeros(total, dtype=torch.float32, device=dev),
         torch.zeros(total, dtype=torch.uint8, device=dev), torch.zeros(total, dtype=torch.uint8, device=dev))
    tok = v.gather_peer_export(*g)
    with open(tokfile + ".tmp", "wb") as fh: fh.write(tok)
    os.rename(tokfile + ".tmp", tokfile)
else:
    t0 = time.time()
    while not os.path.exists(tokfile):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    tok = open(tokfile, "rb").read()
v.gather_init_peer(tok, flags, rank, world, root=root, counts=counts)
crc = 0
rounds = 0
def consume():
    global crc, rounds
    if rank != root: return
    rounds += 1
    if os.environ.get("SLOW_ROOT") and rounds <= 12:
        time.sleep(0.03)  # a slow consumer: the other ranks have long finished this round and want to write the next one
    for t in g:  # stream-ordered copies behind the handshake's waits: the consumer of this round
        crc = zlib.crc32(t.cpu().numpy().tobytes(), crc)
v.reset(seed); v.gather(); consume()
done = 0
while done < steps:
    if rollout:
        v.rollout(rollout); done += rollout
    else:
        v.fill_random_actions(); v.step(); done += 1
    v.gather(); consume()
torch.cuda.synchronize()
v.gather_close()
# the handle has its own output buffers back: one more step must land THERE and leave the global batch alone
before = g[0].clone() if rank == root else None
v.fill_random_actions(); v.step(); torch.cuda.synchronize()
if rank == root:
    assert torch.equal(before, g[0]), "a step after gather_close still wrote the exported buffers"
    np.savez(out, crc=np.uint32(crc), final_obs_own=v.observations.cpu().numpy())
v.close()
"""


def run_ranks(tmp_path, world, total, task, steps, seed, root, rollout, extra_env=None, skip_ranks=(), timeout=120):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    tokfile, out = str(tmp_path / "token"), str(tmp_path / "out.npz")
    flagfile = f"/dev/shm/drone_peer_flags_{os.getpid()}_{abs(hash(str(tmp_path))) % 10**8}"  # tmpfs: plain shared memory behind a name
    with open(flagfile, "wb") as fh:
        fh.write(b"\0" * 4096)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = {r: subprocess.Popen([sys.executable, str(script)] + [str(x) for x in (r, world, total, task, steps, seed, root, rollout)] + [tokfile, flagfile, out],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world) if r not in skip_ranks}
    res = {}
    try:
        for r, p in procs.items():
            so, se = p.communicate(timeout=timeout)
            res[r] = (p.returncode, se)
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        os.unlink(flagfile)
    return res, out


@pytest.mark.parametrize("world,total,task,root,rollout,host_wait", [(2, 8192, 0, 0, 0, 0), (3, 7001, 1, 2, 0, 0), (2, 4096, 2, 1, 0, 0), (3, 6001, 3, 0, 16, 0),
                                                                    (2, 5000, 0, 0, 0, 1), (3, 9000, 0, 1, 0, 2), (2, 5000, 1, 0, 0, 3),
                                                                    (3, 7001, 0, 1, 0, 4), (3, 6001, 3, 2, 16, 4), (3, 9000, 0, 0, 0, 5), (2, 5000, 0, 1, 0, 6), (2, 5000, 0, 0, 0, 7),
                                                                    # configs[2] / configs[4] at their real shape: eight processes x 131 072 envs
                                                                    (8, 1 << 20, 0, 0, 0, 0), (8, (1 << 20) + 5, 0, 3, 0, 0), (8, 1 << 20, 0, 7, 128, 0), (8, 1 << 20, 0, 0, 0, 2)])
def test_peer_stores_land_every_ranks_rows_in_the_roots_batch(oracle, hip, tmp_path, world, total, task, root, rollout, host_wait):
    """host_wait: 0 stream-side handshake (round 5: the publications ride on the output-writing launches), 1 host-side; 2 / 3:
    the same two with a SLOW consumer on the root (it sleeps before reading the batch): the other ranks finish their round at
    once and must NOT overwrite their rows with the next one until the root has begun its own next launch — the back-pressure
    half of the handshake. 4: round 4's form, the publications as one-wave launches of their own (DRONE_PEER_INKERNEL=0);
    5: that form with a slow root. 6 / 7: DRONE_PEER_TIMEOUT_MS = 0 / -5 (stream-side / host-side): malformed budgets are the
    default, not "give up at once" or "never" (ADVICE r4)."""
    big = total > 200000
    steps, seed = (40 if big else 48), 23
    extra = {"DRONE_PEER_HOST_WAIT": "1"} if host_wait in (1, 3, 7) else {}
    if host_wait in (2, 3, 5):
        extra["SLOW_ROOT"] = "1"
    if host_wait in (4, 5):
        extra["DRONE_PEER_INKERNEL"] = "0"
    if host_wait in (6, 7):
        extra["DRONE_PEER_TIMEOUT_MS"] = "0" if host_wait == 6 else "-5"
    res, out = run_ranks(tmp_path, world, total, task, steps, seed, root, rollout, extra_env=extra or None, timeout=400 if big else 120)
    assert all(rc == 0 for rc, _ in res.values()), "\n".join(f"--- rank {r}: rc {rc}\n{se[-1500:]}" for r, (rc, se) in res.items())
    over = dict(horizon=20)
    if task == 2:
        over.update(agents_per_env=8, collision_radius=0.5)
    o = oracle.OracleVec(total, seed=seed, cfg=oracle.default_config(task, **over), threads=usable_cores() if big else 4)
    o.reset(seed)
    crc = 0
    for buf in (o.observations, o.rewards, o.terminals, o.truncations):
        crc = zlib.crc32(buf.tobytes(), crc)
    done = 0
    while done < steps:
        if rollout:
            o.rollout(rollout)
            done += rollout
        else:
            o.fill_random_actions()
            o.step()
            done += 1
        for buf in (o.observations, o.rewards, o.terminals, o.truncations):
            crc = zlib.crc32(buf.tobytes(), crc)
    got = np.load(out)
    assert int(got["crc"]) == crc, f"{world} ranks, root {root}: the root's batches differ from one oracle run over all {total} envs"


@pytest.mark.parametrize("host_wait", [0, 1])
def test_a_dead_peer_is_an_error_not_a_hang(hip, tmp_path, host_wait):
    """Rank 1 never starts. The root's wait gives up after DRONE_PEER_TIMEOUT_MS: on the host (DRONE_PEER_HOST_WAIT=1) the
    gather call itself fails; on the stream the polling lane gives up, sets the error word, and the next call on the
    handle fails. Either way the failure sticks to the handle as an error and the process ends."""
    res, _ = run_ranks(tmp_path, 2, 4096, 0, 4, 1, 0, 0, extra_env={"DRONE_PEER_HOST_WAIT": str(host_wait), "DRONE_PEER_TIMEOUT_MS": "400"}, skip_ranks=(1,))
    rc, se = res[0]
    assert rc != 0 and ("did not reach" in se or "gave up" in se), se[-2000:]


EXE_MP = os.path.join(ROOT, "host", "drone_host_mp")


@pytest.mark.parametrize("ranks,envs,task,rollout,root", [(2, 6000, 0, 0, 0), (3, 10001, 1, 0, 2), (3, 6144, 2, 16, 1), (2, 4097, 3, 0, 1),
                                                          (8, 1 << 20, 0, 0, 0), (8, (1 << 20) + 5, 0, 128, 4)])  # the real shape of configs[2] / configs[4]
def test_plain_c_host_peer_store_exchange(oracle, hip, ranks, envs, task, rollout, root):
    """host/drone_host_mp --exchange peer: the same exchange from plain C (north-star: "host side stays C calling HIP
    through a thin C-ABI") — fork before HIP, the root's export and the flag page through shared mappings made before the
    fork, device memory through drone_device_malloc, no HIP or RCCL header in the host. The root copies every batch to
    the host and chains a CRC-32 over it, which must equal ONE oracle run over all envs."""
    big = envs > 200000
    steps, seed = (256 if rollout else 40) if big else 48, 31
    cmd = [EXE_MP, "--gpus", str(ranks), "--envs", str(envs), "--steps", str(steps), "--task", str(task), "--seed", str(seed),
           "--crc", "1", "--share-devices", "1", "--exchange", "peer", "--root", str(root), "--timeout", "280" if big else "120"]
    if rollout:
        cmd += ["--rollout", str(rollout)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr + r.stdout
    import json

    got = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert got["root"] == root and "peer-store" in got["mode"] and got["gpus"] == ranks
    o = oracle.OracleVec(envs, seed=seed, cfg=oracle.default_config(task), threads=usable_cores() if big else 4)
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
    assert got["crc32"] == crc, f"{ranks} ranks, root {root}: crc {got['crc32']:#x} != oracle {crc:#x}"


def test_plain_c_host_peer_store_with_a_dead_rank_ends(hip):
    """A rank SIGKILLed before the first launch: the parent reaps it, kills the rest and exits non-zero well inside the
    time limit (the ranks blocked in the handshake are killed by the parent; their stream-side waits die with them)."""
    r = subprocess.run([EXE_MP, "--gpus", "3", "--envs", "6000", "--steps", "10", "--share-devices", "1", "--exchange", "peer", "--die-rank", "1", "--timeout", "60"],
                       capture_output=True, text=True, timeout=200, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode != 0 and "ended abnormally" in r.stderr, r.stderr[-1500:]


_WORKER_DIST = r"""
import os, sys, zlib
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from drone_amd import binding
from drone_amd.dist import PeerStoreGather, shard_range
rank, world, total, task, steps, seed, root = (int(x) for x in sys.argv[1:8])
store, out = sys.argv[8:10]
dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
off, cnt = shard_range(total, rank, world)
v = binding.DroneVec(cnt, seed=seed, cfg=binding.default_config(task, horizon=20, env_offset=off), device="cuda:0")
ps = PeerStoreGather(v, total, root=root)   # export + token / flag page over torch.distributed + output rebinding, all in here
crc = 0
def consume(batch):
    global crc
    if batch is not None:
        for t in batch:
            crc = zlib.crc32(t.cpu().numpy().tobytes(), crc)
v.reset(seed); consume(ps())
for _ in range(steps):
    v.fill_random_actions(); v.step(); consume(ps())
torch.cuda.synchronize()
ps.close()
if rank == root:
    np.savez(out, crc=np.uint32(crc))
v.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,total,task,root", [(2, 6000, 0, 0), (3, 7003, 1, 1), (8, 1 << 20, 0, 0)])
def test_peer_store_gather_helper_for_torch_consumers(oracle, hip, tmp_path, world, total, task, root):
    """drone_amd.dist.PeerStoreGather: the same exchange behind the helper torch consumers (and bench.py) use — the token and
    the flag page's name cross torch.distributed (gloo here) once, in its constructor."""
    steps, seed = 40, 29
    shm_before = set(os.listdir("/dev/shm"))
    script = tmp_path / "worker_dist.py"
    script.write_text(_WORKER_DIST.format(root=ROOT))
    store, out = str(tmp_path / "store"), str(tmp_path / "out.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)] + [str(x) for x in (r, world, total, task, steps, seed, root)] + [store, out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=400)
            errs.append((p.returncode, se))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(rc == 0 for rc, _ in errs), "\n".join(f"--- rc {rc}\n{se[-1500:]}" for rc, se in errs)
    o = oracle.OracleVec(total, seed=seed, cfg=oracle.default_config(task, horizon=20), threads=usable_cores())
    o.reset(seed)
    crc = 0
    for buf in (o.observations, o.rewards, o.terminals, o.truncations):
        crc = zlib.crc32(buf.tobytes(), crc)
    for _ in range(steps):
        o.fill_random_actions()
        o.step()
        for buf in (o.observations, o.rewards, o.terminals, o.truncations):
            crc = zlib.crc32(buf.tobytes(), crc)
    assert int(np.load(out)["crc"]) == crc
    left = [f for f in set(os.listdir("/dev/shm")) - shm_before if f.startswith("drone_peer_flags_")]
    assert not left, f"the flag page's name must be gone once every rank has mapped it: {left}"


def test_paths_the_handshake_cannot_cover_are_refused(hip):
    """graph-safe stepping and the K-steps-per-launch calls run outside the handshake (its round numbers are host
    state baked into each launch; step_many writes the caller's blocks, not the root's rows). One process, world 1: the
    refusals are local decisions. Each must fail with a message and leave the exchange usable."""
    import torch

    n = 4096
    flags = np.zeros(2048, dtype=np.uint32)
    page = flags[(-flags.ctypes.data % 4096) // 4:][:1024]  # a page-aligned 4 KiB window
    assert page.ctypes.data % 4096 == 0

    def glob(od):
        return (torch.zeros((n, od), dtype=torch.float32, device="cuda:0"), torch.zeros(n, dtype=torch.float32, device="cuda:0"),
                torch.zeros(n, dtype=torch.uint8, device="cuda:0"), torch.zeros(n, dtype=torch.uint8, device="cuda:0"))

    v = hip.DroneVec(n, seed=3, task=0, device="cuda:0")
    g = glob(20)
    tok = v.gather_peer_export(*g)
    v.gather_init_peer(tok, page, 0, 1, root=0)
    v.reset(3); v.gather()
    with pytest.raises(RuntimeError, match="peer-store"):
        v.enable_graph_capture(True)
    v.clear_status()
    bufs = v.alloc_step_many(4)
    with pytest.raises(RuntimeError, match="peer-store"):
        v.step_many(bufs)
    v.clear_status()
    v.fill_random_actions(); v.step(); v.gather()   # still works, and lands in the exported batch
    torch.cuda.synchronize()
    assert float(g[0].abs().sum()) > 0 and v.status()[0] == 0
    v.gather_close()
    with pytest.raises(RuntimeError, match="export"):   # the export was consumed by the closed exchange
        v.gather_init_peer(tok, page, 0, 1, root=0)
    v.clear_status()
    v.close()
    # ... and the other way round: a handle already in graph-safe mode cannot start the exchange
    v = hip.DroneVec(n, seed=3, task=0, device="cuda:0")
    v.enable_graph_capture(True)
    g = glob(20)
    tok = v.gather_peer_export(*g)
    with pytest.raises(RuntimeError, match="graph-safe"):
        v.gather_init_peer(tok, page, 0, 1, root=0)
    v.close()


_WORKER_QUEUED = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch
from drone_amd import binding
flagfile = sys.argv[1]
v = binding.DroneVec(8192, seed=1, cfg=binding.default_config(0), device="cuda:0")
flags = np.memmap(flagfile, dtype=np.uint32, mode="r+", shape=(1024,))
g = (torch.zeros((16384, 20), dtype=torch.float32, device="cuda:0"), torch.zeros(16384, dtype=torch.float32, device="cuda:0"),
     torch.zeros(16384, dtype=torch.uint8, device="cuda:0"), torch.zeros(16384, dtype=torch.uint8, device="cuda:0"))
v.gather_init_peer(v.gather_peer_export(*g), flags, 0, 2, root=0)   # rank 1 never comes
t0 = time.time()
failed_at = None
for k in range(40):   # a caller that does not sync every launch: every round's wait is queued behind the first one
    try:
        if k == 0: v.reset(1)
        else: v.step()
        v.gather()
    except RuntimeError as exc:
        failed_at = (k, str(exc)); break
torch.cuda.synchronize()
print("QUEUED", failed_at, round(time.time() - t0, 2), flush=True)
try:
    v.step()
    print("NOERROR", flush=True)
except RuntimeError as exc:
    print("ERROR", str(exc)[:200], flush=True)
print("ELAPSED", round(time.time() - t0, 2), flush=True)
"""


def test_waits_queued_behind_a_dead_peers_timeout_return_at_once(hip, tmp_path):
    """with the handshake on the stream a caller may have enqueued many rounds before the first wait gives up
    (DRONE_PEER_TIMEOUT_MS); each later wait must see the raised error word and return at once instead of spinning its own
    budget: 40 rounds x 1.5 s would be a minute, the whole run must end within a few seconds of ONE budget."""
    script = tmp_path / "worker_q.py"
    script.write_text(_WORKER_QUEUED.format(root=ROOT))
    flagfile = f"/dev/shm/drone_peer_flags_q_{os.getpid()}"
    with open(flagfile, "wb") as fh:
        fh.write(b"\0" * 4096)
    try:
        r = subprocess.run([sys.executable, str(script), flagfile], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRONE_PEER_TIMEOUT_MS="1500"))
    finally:
        os.unlink(flagfile)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.split(" ", 1)[0] in ("QUEUED", "ERROR", "NOERROR", "ELAPSED"))
    assert "ERROR" in lines and "gave up" in lines["ERROR"], r.stdout
    assert float(lines["ELAPSED"]) < 12.0, f"queued waits spun their budgets one after the other: {r.stdout}"


_WORKER_QUIET_ROOT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch
from drone_amd import binding
rank, tokfile, flagfile, rollout = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4])
n = 4096
dev = torch.device("cuda:0")
v = binding.DroneVec(n, seed=3, cfg=binding.default_config(0, env_offset=rank * n), device=dev)
flags = np.memmap(flagfile, dtype=np.uint32, mode="r+", shape=(1024,))
if rank == 0:
    g = (torch.zeros((2 * n, 20), dtype=torch.float32, device=dev), torch.zeros(2 * n, dtype=torch.float32, device=dev),
         torch.zeros(2 * n, dtype=torch.uint8, device=dev), torch.zeros(2 * n, dtype=torch.uint8, device=dev))
    tok = v.gather_peer_export(*g)
    with open(tokfile + ".tmp", "wb") as fh: fh.write(tok)
    os.rename(tokfile + ".tmp", tokfile)
else:
    t0 = time.time()
    while not os.path.exists(tokfile):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    tok = open(tokfile, "rb").read()
v.gather_init_peer(tok, flags, rank, 2, root=0)
v.reset(3); v.gather()
torch.cuda.synchronize()
if rank == 0:
    # round 1 is in: the consumer takes its time over it and launches nothing more, so round 1 is never acknowledged
    first = g[0].clone()
    assert first[n:].abs().sum().item() > 0, "rank 1's reset rows never arrived"
    t0 = time.time()
    while not os.path.exists(tokfile + ".rank1_done"):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    torch.cuda.synchronize()
    print("ROWS_UNTOUCHED", bool(torch.equal(first, g[0])), flush=True)
    time.sleep(0.2)
else:
    err = None
    try:
        for k in range(6):   # queued behind the acknowledgement that never comes: the first wait gives up after the budget ...
            if rollout: v.rollout(rollout)
            else: v.fill_random_actions(); v.step()
            v.gather()
        torch.cuda.synchronize()
        v.step()
    except RuntimeError as exc:
        err = str(exc)
    torch.cuda.synchronize()
    print("RANK1_ERROR", err, flush=True)
    open(tokfile + ".rank1_done", "w").close()
    time.sleep(1.0)   # keep the mapping alive while the root compares
"""


@pytest.mark.parametrize("rollout", [0, 16])
def test_launches_queued_behind_a_failed_wait_store_nothing(hip, tmp_path, rollout):
    """ADVICE r4 / VERDICT r5 item 2: the launches a rank queued behind a timed-out acknowledgement wait used to run anyway and
    overwrite the root's batch — which the root, merely slow, may still be reading. The wait now raises a stop word in HBM as
    well (drone_kernels.h LaunchSig) and the PEER instantiations of the step / rollout / reset kernels — which only handles in
    an exchange launch; every other handle's kernels are round 5's, instruction for instruction (tests/test_build_variants.py)
    — store and publish nothing once it is raised: the root's whole batch is bit for bit what it was while rank 1's six queued
    launches drain."""
    script = tmp_path / "worker_quiet.py"
    script.write_text(_WORKER_QUIET_ROOT.format(root=ROOT))
    tokfile = str(tmp_path / "token")
    flagfile = f"/dev/shm/drone_peer_flags_{os.getpid()}_quiet{rollout}"
    with open(flagfile, "wb") as fh:
        fh.write(b"\0" * 4096)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRONE_PEER_TIMEOUT_MS="1500")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), tokfile, flagfile, str(rollout)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        os.unlink(flagfile)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert "RANK1_ERROR" in outs[1][0] and "gave up" in outs[1][0], outs[1][0]
    assert "ROWS_UNTOUCHED True" in outs[0][0], outs[0][0]


def test_two_launches_without_a_gather_are_refused(hip, tmp_path):
    """with the publications riding on the output-writing launches, reset(); step(); gather() would let the second
    launch overwrite this rank's rows of the root's batch while the root may be consuming the round the first one announced
    (its wait is already satisfied). The library now insists on one drone_vec_gather per launch while the exchange is active
    — on every rank, the root included — and says so; after the gather the handle steps again."""
    import torch

    n = 2048
    dev = torch.device("cuda:0")
    v = hip.DroneVec(n, seed=5, task=0, device=dev)
    g = (torch.zeros((n, 20), dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev),
         torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))
    flagfile = f"/dev/shm/drone_peer_flags_{os.getpid()}_twice"
    with open(flagfile, "wb") as fh:
        fh.write(b"\0" * 4096)
    try:
        flags = np.memmap(flagfile, dtype=np.uint32, mode="r+", shape=(1024,))
        tok = v.gather_peer_export(*g)
        v.gather_init_peer(tok, flags, 0, 1, root=0)
        v.reset(5)
        v.fill_random_actions()
        with pytest.raises(RuntimeError, match="drone_vec_gather must follow every"):
            v.step()
        v.clear_status()
        v.gather()
        v.step(); v.gather()
        v.rollout(4)
        with pytest.raises(RuntimeError, match="drone_vec_gather must follow every"):
            v.rollout(4)
        v.clear_status()
        v.gather()
        torch.cuda.synchronize()
        assert "peer" not in v.variant[0] or v.variant[1]["mem"] == 0
        v.gather_close()
        v.step(); v.step()  # no exchange, no rule
        torch.cuda.synchronize()
    finally:
        v.close()
        os.unlink(flagfile)



def test_export_of_a_virtual_memory_allocation_names_the_way_out(hip, tmp_path):
    """hipIpcGetMemHandle has no handle for a virtual-memory mapping (hipMemCreate / hipMemAddressReserve /
    hipMemMap — what an expandable-segments allocator hands out). drone_vec_gather_peer_export must then say what to do
    (drone_device_malloc) instead of relaying "invalid argument". tests/c/vmm_export_probe.c builds such a mapping with the HIP
    virtual-memory API, places the four global buffers in it and exports them; the same export from drone_device_malloc
    buffers is the positive control. (This torch build's allocator hands out plain allocations even under
    expandable_segments:True, so the probe is plain C.)"""
    exe = str(tmp_path / "vmm_export_probe")
    subprocess.run(["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "vmm_export_probe.c"),
                    "-L" + os.path.join(ROOT, "drone_amd"), "-l:libdrone_hip.so", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "drone_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.split(" ", 1)[0] in ("PLAIN", "VMM", "VMM_UNAVAILABLE"))
    assert lines["PLAIN"].startswith("rc=0"), r.stdout
    if "VMM_UNAVAILABLE" in lines:
        pytest.skip("this runtime has no virtual-memory API: " + lines["VMM_UNAVAILABLE"])
    if lines["VMM"].startswith("rc=0"):
        pytest.skip("this runtime gives a virtual-memory mapping an IPC handle: the export works")
    assert "drone_device_malloc" in lines["VMM"] and "hipIpcGetMemHandle" in lines["VMM"] and "virtual-memory" in lines["VMM"], lines["VMM"]


@pytest.mark.parametrize("world,rollout", [(2, 0), (3, 16)])
def test_many_rounds_per_process_start_every_batch_checked(hip, world, rollout):
    """tests/peer_stress.py (round 6): 1 500 handshake rounds between `world` processes, the root dawdling at random, every
    round's batch compared with what each rank computes in a twin handle — the dense form of the cases above, which spend
    their seconds importing torch. tools/r06_flake.sh runs it by the thousand rounds per library variant."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "peer_stress.py"), "--world", str(world), "--rounds", "1500", "--rollout", str(rollout), "--seed", "7"],
                       capture_output=True, text=True, timeout=400)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert r.returncode == 0 and line["mismatches"] == 0 and line["ranks_rc"] == [0] * world, (line, r.stderr[-2000:])
