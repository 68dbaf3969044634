"""(-DMIR_PROFILE_SINGLE build) how long each env of a stack-task launch keeps its workgroup busy up to the end of the Newton solve,
by iteration count / block coupling / contact count: what the launch's tail is made of."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B)
env.reset(seed=0)
task = env._env; sc = task._mir; sc.set_diag(True)
home = task._home
g = torch.Generator(device=sc.device); g.manual_seed(1)
for t in range(40):
    task.step_raw(home + torch.rand((B, 9), device=sc.device, generator=g) * 2 - 1)
    if t % 10 == 9:
        dg = [x.cpu().numpy() for x in sc.get_diag()]
        ncon, niter, raw = dg[0], dg[2], dg[1]
        cyc = raw & ((1 << 30) - 1); cpl = (raw >> 30) & 1
        print(f"step {t}: cycles mean {cyc.mean():.0f} median {np.median(cyc):.0f} p90 {np.percentile(cyc, 90):.0f} p99 {np.percentile(cyc, 99):.0f} max {cyc.max()}  coupled {cpl.mean():.3f}")
        for c in (0, 1):
            for it in range(0, 6):
                m = (cpl == c) & (niter == it)
                if m.sum():
                    print(f"   coupled={c} niter={it}: n={m.sum():5d} cycles mean {cyc[m].mean():8.0f} max {cyc[m].max():8d} ncon mean {ncon[m].mean():.1f}")
