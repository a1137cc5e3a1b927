#!/usr/bin/env python3
"""A/B the per-step kernel across builds / init-time settings of the library in
ONE process with interleaved rounds (cdna_hip_programming.md §5.4 rule 24).

Memory placement alone moves this HBM-bound kernel by ±6 % (identical code,
different allocations: gpurun_out/ab3.txt), so every variant is created, timed
and CLOSED in turn: the allocators hand the next variant the same blocks and
the comparison is at equal placement. `--fresh` keeps all variants alive
instead (different placements; shows the spread).

  python tools/ab_step.py [--envs N] [--rounds 8] [--steps 200] "name=-Dflags;ENV=val,ENV2=val" "old=@tools/ab_libs/lib.so" ...
"""
import argparse
import gc
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--task", default="hover")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mode", default="step", choices=["step", "rollout", "many"])
    ap.add_argument("--k", type=int, default=32, help="--mode many: env steps per launch (reported per env step)")
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--fresh", action="store_true")
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--warm", type=int, default=150, help="untimed steps first (150: past the synchronised first episode ends)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    task = {"hover": abi.TASK_HOVER, "waypoint": abi.TASK_WAYPOINT, "swarm": abi.TASK_SWARM, "race": abi.TASK_RACE}[a.task]
    specs = {}
    for spec in a.variants:
        name, _, rest = spec.partition("=")
        flags, _, envs = rest.partition(";")
        lib = f"/tmp/ab_{abs(hash(flags)) % 10**8}.so"
        if flags.startswith("@"):  # a library built elsewhere (e.g. last round's sources: tools/ab_libs/), relative to the repo root
            lib = os.path.join(ROOT, flags[1:])
            assert os.path.exists(lib), lib
        elif not os.path.exists(lib):
            r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "drone_amd", "csrc"), "-B", f"OUT={lib}", f"EXTRA={flags}"],
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(name, "BUILD FAILED", r.stderr[-400:])
                continue
        specs[name] = (binding.load_variant(lib), dict(kv.split("=", 1) for kv in filter(None, envs.split(","))), rest)

    import ctypes as C

    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hiprt.hipFree.argtypes = [C.c_void_p]
    FLAGS = {"finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}

    class RawTensor:
        """torch view of memory from hipExtMallocWithFlags (experiment: OUT_ALLOC / ACT_ALLOC = uncached | finegrained)."""

        def __init__(self, like, kind):
            self.ptr = C.c_void_p()
            nbytes = like.numel() * like.element_size()
            rc = hiprt.hipExtMallocWithFlags(C.byref(self.ptr), nbytes, FLAGS[kind])
            assert rc == 0, f"hipExtMallocWithFlags({kind}) -> {rc}"
            typestr = {torch.float32: "<f4", torch.uint8: "|u1"}[like.dtype]
            self.__cuda_array_interface__ = {"shape": tuple(like.shape), "typestr": typestr, "data": (self.ptr.value, False), "version": 2}
            self.tensor = torch.as_tensor(self, device=like.device)

        def free(self):
            hiprt.hipFree(self.ptr)

    def make(name):
        fns, envs, _ = specs[name]
        os.environ.update(envs)
        over = {k[4:]: (float(val) if "." in val or "e" in val else int(val)) for k, val in envs.items() if k.startswith("CFG_")}  # CFG_bound=1e6,CFG_horizon=1000000000
        v = binding.DroneVec(a.envs, seed=0, task=task, device="cuda:0", fns=fns, **over)
        for k in envs:
            os.environ.pop(k, None)
        v._raw = []
        if envs.get("OUT_ALLOC"):
            outs = [RawTensor(t, envs["OUT_ALLOC"]) for t in (v.observations, v.rewards, v.terminals, v.truncations)]
            v._raw += outs
            v.bind_outputs(*[r.tensor for r in outs])
        v.reset(0)
        if envs.get("ACT_ALLOC"):
            raws = [RawTensor(v.actions, envs["ACT_ALLOC"]) for _ in range(int(envs.get('RING', a.ring)))]
            v._raw += raws
            ring = [r.tensor for r in raws]
        else:
            ring = [torch.empty_like(v.actions) for _ in range(int(envs.get('RING', a.ring)))]
        for k, r_ in enumerate(ring):
            v.fill_random_actions(gstep=k, out=r_)
        return v, ring

    def timeit(v, ring):
        if a.mode == "step":
            for k in range(a.warm):
                v.bind_actions(ring[k % len(ring)])
                v.step()
            torch.cuda.synchronize()
            v.timer_start()
            for k in range(a.steps):
                v.bind_actions(ring[k % len(ring)])
                v.step()
            return v.timer_stop() * 1e3 / a.steps
        if a.mode == "many":
            bufs = v.alloc_step_many(a.k)
            for k in range(a.k):
                bufs.actions[k].copy_(ring[k % len(ring)])
            reps = max(4, a.steps // a.k)
            for _ in range(3):
                v.step_many(bufs)
            torch.cuda.synchronize()
            v.timer_start()
            for _ in range(reps):
                v.step_many(bufs)
            return v.timer_stop() * 1e3 / (reps * a.k)
        for _ in range(30):  # a pure-VALU kernel follows the clock transient of a GPU leaving idle
            v.rollout(a.horizon)
        torch.cuda.synchronize()
        v.timer_start()
        for _ in range(10):
            v.rollout(a.horizon)
        return v.timer_stop() * 1e3 / 10

    times = {k: [] for k in specs}
    said = {k: [] for k in specs}  # drone_vec_variant() of every handle timed (an HBM-bound handle says which sweep order it measured and picked)
    live = {k: make(k) for k in specs} if a.fresh else None
    for rnd in range(a.rounds):
        for name in specs:
            if a.fresh:
                times[name].append(timeit(*live[name]))
            else:
                v, ring = make(name)
                times[name].append(timeit(v, ring))
                said[name].append(v.variant[0].split("> ", 1)[-1])  # behind the timed steps: a handle that measures its sweep order on its first ~260 steps has said so by now
                torch.cuda.synchronize()
                v.close()
                for r in v._raw:
                    r.free()
                del v, ring
                gc.collect()
    base = None
    for name, ts in times.items():
        med = statistics.median(ts)
        base = base or med
        print(json.dumps({"variant": name, "spec": specs[name][2], "median_us": round(med, 2), "min_us": round(min(ts), 2),
                          "max_us": round(max(ts), 2), "vs_first": round(med / base, 4), "all": [round(t, 1) for t in ts],
                          **({"handles": said[name]} if any("autotuned" in x for x in said[name]) else {})}))


if __name__ == "__main__":
    main()
