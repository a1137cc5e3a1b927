import sys, torch
sys.path.insert(0, "/root/repo")
from drone_amd import binding, abi
for A in (1, 2, 8, 16, 64):
    v = binding.DroneVec(1 << 20, seed=0, task=abi.TASK_SWARM, device="cuda:0", agents_per_env=A)
    v.reset(0)
    ring = [torch.empty_like(v.actions) for _ in range(4)]
    for k, r in enumerate(ring): v.fill_random_actions(gstep=k, out=r)
    for k in range(50): v.bind_actions(ring[k % 4]); v.step()
    torch.cuda.synchronize(); v.timer_start()
    for k in range(300): v.bind_actions(ring[k % 4]); v.step()
    us = v.timer_stop() * 1e3 / 300
    v.rollout(128); torch.cuda.synchronize(); v.timer_start(); v.rollout(128); ms = v.timer_stop()
    print(f"A={A:2d}: step {us:6.2f} us   fused 128-step rollout {ms:7.3f} ms")
    v.close()
