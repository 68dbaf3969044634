"""(-DMIR_PROFILE_SINGLE build) Which narrowphase path the candidate pairs of the stack scene take, env by env: slab fast path,
plane-box, plane-round, box-box (15-axis routine), convex (lane-private GJK / MPR) and how many of the convex ones end in a contact."""
import ctypes as C, os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = 4096
robot = sys.argv[1] if len(sys.argv) > 1 else "franka"
env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B)
env.reset(seed=0)
task = env._env; sc = task._mir; sc.set_diag(True)
sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
sc.lib.mir_debug_profile_step.restype = C.c_int
g = torch.Generator(device="cuda").manual_seed(3)
home = task._home
for t in range(40):
    task.step_raw(home + torch.empty_like(home).uniform_(-1, 1, generator=g) * (t > 5))
prof = torch.zeros(48, dtype=torch.int64, device=sc.device)
prof[33] = 77
sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
torch.cuda.synchronize()
w = sc.get_diag()[1].cpu().numpy()
names = ["slab", "plane-box", "plane-round", "box-box", "convex", "convex with contact"]
cnt = np.stack([(w >> s) & 31 for s in (0, 5, 10, 15, 20, 25)], 1)
print(robot, "candidates per env (mean / max):", {n: (round(float(cnt[:, i].mean()), 2), int(cnt[:, i].max())) for i, n in enumerate(names)})
print("envs with at least one box-box routine call %.3f, with at least one convex call %.3f" % ((cnt[:, 3] > 0).mean(), (cnt[:, 4] > 0).mean()))
p = prof.cpu().numpy().astype(np.float64)
print("workgroup 0, collision wave: broadphase end -> plane-box end %.0f | slab path %.0f | box-box loop %.0f (%d candidates) | convex block + bookkeeping %.0f | kinds of env 0: %s" % (
    p[7] - p[6], p[36] - p[7], p[37] - p[36], int(p[38]), p[8] - p[37], cnt[0].tolist()))
