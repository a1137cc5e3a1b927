"""PufferLib-shaped env class over the C-ABI.

Mirrors the shape of a PufferLib ocean env's Python side — flat shared buffers
``observations / actions / rewards / terminals / truncations``, ``num_agents``,
``single_observation_space`` / ``single_action_space``, ``reset(seed)``,
``step(actions)`` returning ``(obs, rewards, terminals, truncations, infos)``
with the aggregated log appended to ``infos`` every ``log_interval`` steps,
``close()``. The upstream class cannot be cited or import-tested: the
``pufferlib`` submodule is empty in the reference snapshot
(``/root/reference/.gitmodules:1-3``); the drop-in claim is to that extent
unverified.
"""
from dataclasses import dataclass

import numpy as np

from . import abi, binding


@dataclass(frozen=True)
class Box:
    """Minimal stand-in for gymnasium.spaces.Box (gymnasium is not a dependency)."""
    low: float
    high: float
    shape: tuple
    dtype: type = np.float32


class Drone:
    def __init__(self, num_envs=1024, task="hover", device=None, seed=0, log_interval=128, buf=None, **config):
        """``buf``: PufferLib-style buffer holder — any object with ``observations``,
        ``actions``, ``rewards``, ``terminals``, ``truncations`` arrays (numpy for the host
        path, torch CUDA tensors for the device path), e.g. a worker's slice of the
        vec-env's shared memory. Without it the env allocates its own."""
        task_id = {"hover": abi.TASK_HOVER, "waypoint": abi.TASK_WAYPOINT, "swarm": abi.TASK_SWARM, "race": abi.TASK_RACE}[task] if isinstance(task, str) else int(task)
        buffers = None if buf is None else (buf.observations, buf.actions, buf.rewards, buf.terminals, buf.truncations)
        self.vec = binding.DroneVec(num_envs, seed=seed, task=task_id, device=device, buffers=buffers, **config)
        self.num_agents = self.vec.num_envs
        self.single_observation_space = Box(-np.inf, np.inf, (abi.obs_dim(task_id),))
        self.single_action_space = Box(-1.0, 1.0, (abi.ACT_DIM,))
        self.log_interval = int(log_interval)
        self.seed = seed
        self.tick = 0

    # the shared buffers, exactly the arrays / tensors the C side reads and writes
    observations = property(lambda self: self.vec.observations)
    actions = property(lambda self: self.vec.actions)
    rewards = property(lambda self: self.vec.rewards)
    terminals = property(lambda self: self.vec.terminals)
    truncations = property(lambda self: self.vec.truncations)

    def reset(self, seed=None):
        if seed is not None:
            self.seed = seed
        self.vec.reset(self.seed)
        self.tick = 0
        return self.observations, []

    def step(self, actions):
        if actions is not self.vec.actions:
            if self.vec.torch_device is None:
                self.vec.actions[:] = actions
            else:
                self.vec.actions.copy_(actions, non_blocking=True)
        if self.vec.torch_device is not None:
            self.vec.use_torch_stream()  # follow torch.cuda.stream(...) contexts; a no-op while the stream is unchanged
        self.vec.step()
        self.tick += 1
        infos = []
        if self.log_interval and self.tick % self.log_interval == 0:
            log = self.vec.log()
            if log["n"] > 0:
                infos.append(log)
        return self.observations, self.rewards, self.terminals, self.truncations, infos

    def send(self, actions):
        """Async half of ``step`` (PufferLib vec-envs' ``send`` / ``recv`` shape): hand over the actions and return while
        the env steps; ``recv()`` then returns what ``step`` would have."""
        if actions is not self.vec.actions:
            if self.vec.torch_device is None:
                self.vec.actions[:] = actions
            else:
                self.vec.actions.copy_(actions, non_blocking=True)
        if self.vec.torch_device is not None:
            self.vec.use_torch_stream()
        self.vec.step_send()

    def recv(self):
        self.vec.step_recv()
        self.tick += 1
        infos = []
        if self.log_interval and self.tick % self.log_interval == 0:
            log = self.vec.log()
            if log["n"] > 0:
                infos.append(log)
        return self.observations, self.rewards, self.terminals, self.truncations, infos

    def step_many(self, actions=None, k_steps=None):
        """K env steps in ONE launch with every step's outputs (``drone_vec_step_many``): open-loop action segments,
        action repeat, or — ``actions=None`` with ``k_steps`` — the device-side random policy. ``actions`` is a
        ``[K][num_envs][4]`` block of the env's buffer kind. Returns K-major ``(observations [K][N][O], rewards,
        terminals, truncations [K][N], infos)``; the env's per-step buffers are not touched. The blocks are reused
        from call to call for the same K."""
        K = int(k_steps if actions is None else actions.shape[0])
        bufs = getattr(self, "_many", None)
        if bufs is None or bufs.k_steps != K:
            if bufs is not None:
                self.vec.free_step_many(bufs)  # a new K: the old pinned K x N host blocks must not stay registered until close
            bufs = self._many = self.vec.alloc_step_many(K)
        if actions is not None and actions is not bufs.actions:
            if self.vec.torch_device is None:
                bufs.actions[:] = actions
            else:
                bufs.actions.copy_(actions, non_blocking=True)
        if self.vec.torch_device is not None:
            self.vec.use_torch_stream()
        self.vec.step_many(bufs, policy=actions is None)
        before = self.tick
        self.tick += K
        infos = []
        if self.log_interval and before // self.log_interval != self.tick // self.log_interval:
            log = self.vec.log()
            if log["n"] > 0:
                infos.append(log)
        return bufs.observations, bufs.rewards, bufs.terminals, bufs.truncations, infos

    def save(self, path):
        """Checkpoint the env (state, step counter, seed, config, buffer contents): ``DroneVec.save_checkpoint``."""
        self.vec.save_checkpoint(path)

    def load(self, path):
        """Resume from ``save``: the env continues bit for bit like the one that wrote the file. Returns the observations."""
        self.vec.load_checkpoint(path)
        self.seed = self.vec._seed
        self.tick = self.vec.gstep
        return self.observations

    def close(self):
        self.vec.close()
