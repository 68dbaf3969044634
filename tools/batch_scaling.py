"""Throughput vs batch size on one MI355X (GPU probe): pick (16-lane kernel) and stack (wave kernel), one launch per step and
16-step rollout launches, fresh random PD targets per step."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
for task, sizes in (("cube_pick", (1024, 4096, 16384, 65536, 262144)), ("cube_stack", (1024, 4096, 16384, 65536))):
    for B in sizes:
        env = GenesisEnv(task=task, robot="franka", num_envs=B)
        env.reset(seed=0)
        t = env._env
        gen = torch.Generator(device=dev).manual_seed(1)
        acts = t._home[0] + torch.empty((32, B, 9), device=dev).uniform_(-1, 1, generator=gen)
        for k in range(10): t.step_raw(acts[k])
        torch.cuda.synchronize()
        n = 64
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n): t.step_raw(acts[k % 32])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        rows = torch.zeros((16, B, t._mir.agent_dim + t._mir.env_dim + 2), device=dev)
        t._mir.rollout(acts[:16], rows); torch.cuda.synchronize()
        e0.record()
        for k in range(4): t._mir.rollout(acts[16 * (k % 2):16 * (k % 2) + 16], rows)
        e1.record(); torch.cuda.synchronize()
        us_ro = e0.elapsed_time(e1) * 1e3 / 64
        print(f"{task} B={B:7d}: {us:9.1f} us/step {B/us:8.1f} M env-steps/s | rollout16 {us_ro:9.1f} us/step {B/us_ro:8.1f} M env-steps/s")
        del env, t, acts, rows
        torch.cuda.empty_cache()
