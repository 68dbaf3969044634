"""Per-launch fixed cost vs per-step cost of the pick kernel (GPU probe): time of mir_rollout launches of K steps, K = 1..16."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B)
env.reset(seed=0)
t = env._env
gen = torch.Generator(device=dev).manual_seed(1)
acts = torch.empty((64, B, 9), device=dev).uniform_(-1, 1, generator=gen)
res = []
for K in (1, 2, 4, 8, 16, 32):
    rows = torch.zeros((K, B, 22), device=dev)
    for _ in range(3): t._mir.rollout(acts[:K], rows)
    torch.cuda.synchronize()
    n = max(4, 256 // K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): t._mir.rollout(acts[(i % 2) * K:(i % 2) * K + K], rows)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    res.append((K, us))
    print(f"K={K:2d}: {us:8.1f} us per launch, {us / K:6.1f} us per step")
(k0, t0), (k1, t1) = res[0], res[-1]
b = (t1 - t0) / (k1 - k0)
print(f"fit: {t0 - b * k0:.1f} us per launch + {b:.1f} us per step")
