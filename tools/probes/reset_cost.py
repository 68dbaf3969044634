"""What env.reset() costs inside the README loop (bench.py's headline resets every 200 steps): wall time of 200 x (19 steps + reset + 1 step)
against 200 x 20 steps, and of the pieces of one reset."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(root, "gym-genesis_amd")]
import torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(1)
acts = list(torch.empty((64, B, 9), device="cuda").uniform_(-1, 1, generator=g).unbind(0))
def loop(n, with_reset):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for t in range(20):
            env.step(acts[t])
        if with_reset:
            env.reset()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for _ in range(2):
    a, b = loop(200, False), loop(200, True)
print(f"20 steps: {a * 1e6:.1f} us; 20 steps + reset: {b * 1e6:.1f} us; one reset in the loop costs {(b - a) * 1e6:.1f} us = {(b - a) / 200 * 1e6 / 1:.2f} us per step at one reset per 200 steps")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    env.reset()
torch.cuda.synchronize(); print(f"reset back to back: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us each")
