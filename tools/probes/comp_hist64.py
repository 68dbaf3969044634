"""(-DMIR_PROFILE_SINGLE -DMIR_PROFILE_COMP build) which blocks the contacts couple, per env: histogram of the coupling structures."""
import os, sys, collections
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = 4096
for robot in ("franka", "so101"):
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B)
    env.reset(seed=0)
    task = env._env; sc = task._mir; sc.set_diag(True)
    home = task._home
    g = torch.Generator(device=sc.device); g.manual_seed(1)
    for t in range(40):
        task.step_raw(home + torch.rand((B, home.shape[-1]), device=sc.device, generator=g) * 2 - 1)
    comp = sc.get_diag()[1].cpu().numpy()
    def comps(c):
        seen, out = set(), []
        for p in range(4):
            if p in seen: continue
            mem = tuple(q for q in range(4) if (c >> (4 * p + q)) & 1)
            seen.update(mem)
            if len(mem) > 1: out.append(mem)
        return tuple(out)
    h = collections.Counter(comps(int(c)) for c in comp)
    print(robot, [(k, v) for k, v in h.most_common()])
