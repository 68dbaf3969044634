"""(-DMIR_PROFILE_SINGLE build) kernel durations and launch-to-launch gaps of a running GenesisEnv.step loop, measured by the kernels
themselves (s_memrealtime, 100 MHz, of workgroup 0's entry and of the last exit of a collision wave among the workgroups of its XCD): no profiler in
the way of the host.  argv[1] = lean: the bare go / end loop instead of env.step."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv

B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env
mir = task._mir
lib = mir.lib
lib.mir_debug_profile_next_step.argtypes = [C.c_void_p, C.c_void_p]
lib.mir_debug_profile_next_step.restype = C.c_int
g = torch.Generator(device=task.device).manual_seed(0)
acts = [torch.empty((B, 9), device=task.device).uniform_(-1, 1, generator=g) for _ in range(25)]
NP = 6
profs = [torch.zeros(160, dtype=torch.int64, device=task.device) for _ in range(NP)]
ptrs = [C.c_void_p(p.data_ptr()) for p in profs]
res = []
for rep in range(60):
    for t in range(40):
        env.step(acts[(rep + t) % 25])
    for p in profs:
        p.zero_(); p[29] = 2 ** 62
    torch.cuda.synchronize()
    for t in range(10):
        env.step(acts[t])
    for k in range(NP):
        lib.mir_debug_profile_next_step(mir.h, ptrs[k])
        env.step(acts[(rep + k) % 25])
    for t in range(3):
        env.step(acts[t])
    torch.cuda.synchronize()
    P = np.stack([p.cpu().numpy() for p in profs]).astype(np.float64)
    start, end = P[:, 26], P[:, 31]
    if (start == 0).any():
        continue
    res.append(np.concatenate([(end - start) / 100.0, (start[1:] - end[:-1]) / 100.0, (start[1:] - start[:-1]) / 100.0, (P[:, 130] - start) / 100.0]))
R = np.array(res)
print(f"split_step {mir.split_step}; {len(R)} samples of {NP} consecutive launches (us)")
print("  duration       ", np.round(R[:, :NP].mean(0), 2), " mean", round(R[:, 1:NP - 1].mean(), 2))
print("  gap to the next", np.round(R[:, NP:2 * NP - 1].mean(0), 2), " mean", round(R[:, NP + 1:2 * NP - 1].mean(), 2))
print("  start to start ", np.round(R[:, 2 * NP - 1:3 * NP - 2].mean(0), 2))
print("  last terminated store of the XCD after the start", np.round(R[:, 3 * NP - 2:].mean(0), 2))
