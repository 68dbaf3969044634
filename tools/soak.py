"""Long free-running soak of every task on the GPU: random PD targets, thousands of steps, all envs; reports non-finite
states, contact-cap saturation, Newton iteration tails and how far things fly."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
import ctypes as C
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for task, robot in (("cube_pick", "franka"), ("cube_pick", "so101"), ("cube_stack", "franka"), ("cube_stack", "so101")):
    env = GenesisEnv(task=task, robot=robot, num_envs=B)
    env.reset(seed=0)
    t = env._env; sc = t._mir; sc.set_diag(True); dev = sc.device
    nu = sc.nu
    home = getattr(t, "_home", torch.zeros((B, nu), device=dev))[:, :nu]
    gen = torch.Generator(device=dev).manual_seed(7)
    worst_iter = 0; cap_hits = 0; nonfinite = 0; maxcon = 0; rewards = 0.0
    sc.lib.mir_debug_poison_lds.argtypes = [C.c_int, C.c_void_p]
    for k in range(T):
        if k % 25 == 0:  # LDS full of NaN patterns every 25 steps: a read-before-write shows up as a non-finite state
            sc.lib.mir_debug_poison_lds(dev.index or 0, sc._stream())
        a = home + torch.empty((B, nu), device=dev).uniform_(-1.5, 1.5, generator=gen)
        t.step_raw(a)
        if k % 50 == 49:
            ncon, nefc, niter = sc.get_diag()
            q, v, _, _ = sc.get_state()
            nonfinite += int((~torch.isfinite(q)).any(1).sum().item()) + int((~torch.isfinite(v)).any(1).sum().item())
            worst_iter = max(worst_iter, int(niter.max().item())); maxcon = max(maxcon, int(ncon.max().item()))
            cap_hits += int((ncon >= sc.spec.opt.max_contacts).sum().item())
            rewards += float(t._reward.sum().item())
        if k % 200 == 199:
            t.reset()
    q, v, _, _ = sc.get_state()
    print(f"{task}/{robot} B={B} T={T}: nonfinite {nonfinite}, max niter {worst_iter}, max ncon {maxcon} (cap {sc.spec.opt.max_contacts}, hits {cap_hits}), "
          f"max|qvel| {v.abs().max().item():.1f}, max|qpos| {q.abs().max().item():.2f}, reward hits {rewards:.0f}")
    del env
