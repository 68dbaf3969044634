"""Developer tool: the reference's expert (examples/franka/pick_cube_state.py = /root/reference/examples/franka/pick_cube_state.py:16-54,86-88
restated) at 4096 envs through GenesisEnv.step with the manifolds thinned at 16 points and with exact contacts: microseconds per
env.step() call (the time inside the call, the policy's IK outside it), microseconds per loop iteration, and for the exact run the
per-call times binned by the number of envs the step deferred.  MIR_EXACT_WAVE=1: the deferred envs on the wave-per-env kernel (round 5).

    python3 tools/expert_time.py [episodes]
"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
import numpy as np
import torch

from gym_genesis.env import GenesisEnv

spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
ex = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ex)
B = 4096
EPISODES = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(exact):
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=exact)
    mir = env._env._mir
    in_step, loop, per, ndef = [], [], [], []
    for ep in range(EPISODES + 1):  # (the first episode warms up)
        obs, _ = env.reset(seed=ep)
        mir.exact_stats(reset=True)
        torch.cuda.synchronize()
        t_in = 0.0
        prev = 0
        t0 = time.perf_counter()
        for stage in ex.STAGES:
            for _ in range(40):
                a = ex.expert_policy(env.get_robot(), obs, stage)
                ta = time.perf_counter()
                obs, reward, term, trunc, info = env.step(a)
                dt = time.perf_counter() - ta
                t_in += dt
                if ep and exact:
                    n = mir.exact_stats()["overflow_env_steps"]
                    per.append(dt * 1e6); ndef.append(n - prev); prev = n
        torch.cuda.synchronize()
        if ep:
            in_step.append(t_in / 200 * 1e6); loop.append((time.perf_counter() - t0) / 200 * 1e6)
    st = mir.exact_stats()
    route = mir.exact_route() if exact else {}
    print(f"exact={exact}: env.step {np.round(in_step, 1)} us per call, loop {np.round(loop, 1)} us per iteration; lifted {float(term.mean()):.3f}; last episode {st} {route}")
    if exact:
        per, ndef = np.array(per), np.array(ndef)
        for lo, hi in ((0, 0), (1, 16), (17, 256), (257, 1024), (1025, 2048), (2049, 4096)):
            sel = (ndef >= lo) & (ndef <= hi)
            if sel.any():
                print(f"  steps with {lo}..{hi} deferred envs: {int(sel.sum())}, env.step median {np.median(per[sel]):.1f} us, mean {per[sel].mean():.1f} us")
    del env


for exact in (False, True, False, True):
    run(exact)
