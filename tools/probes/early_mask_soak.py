"""Soak of the early terminated bytes: 5000 random-action env.steps (with the README loop's resets) and the reference expert's pick
episodes at 4096 envs; the kernel's own re-check of every early mask against the integrated state must count zero mismatches, and the
host masks must equal the device masks."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
mir = env._env._mir
mir.set_diag(True)
mir.early_mask_stats(reset=True)
dev = env._env.device
g = torch.Generator(device=dev).manual_seed(3)
env.reset(seed=0)
nt = 0
for t in range(5000):
    a = torch.empty((B, 9), device=dev).uniform_(-1, 1, generator=g)
    obs, rew, term, trunc, info = env.step(a)
    if t % 97 == 0:
        assert np.array_equal(term, info["is_success"].cpu().numpy())
    nt += int(term.sum())
    if term.any() or t % 200 == 199:
        env.reset()
early, bad = mir.early_mask_stats(reset=True)
print(f"random actions: 5000 steps, terminated envs seen {nt}, early workgroup-launches {early} of {5000 * 1024}, mismatches {bad}")
spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
tot = 0
for ep in range(3):
    obs, _ = env.reset(seed=10 + ep)
    states, envs, acts, rews = ex.run_episode(env, obs)
    tot += int((np.stack([np.asarray(r.cpu() if hasattr(r, "cpu") else r) for r in rews]).max(0) > 0).sum())
early, bad = mir.early_mask_stats(reset=True)
print(f"expert episodes: 3 x 200 steps, lifted {tot} of {3 * B}, early workgroup-launches {early} of {3 * 200 * 1024}, mismatches {bad}")
assert bad == 0
