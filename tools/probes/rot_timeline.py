"""(-DMIR_PROFILE_SINGLE build) time line of workgroup 0 of ONE rotated launch taken out of a running GenesisEnv.step loop: every
stamp both waves left, sorted, in shader cycles and us after the kernel's entry.  Shows how much of the launch lies before the
terminated bytes leave (the host's earliest return) and how much after (the time the host has to come back with the next action)."""
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
NAMES = {24: "w0 entry", 0: "w0 state loaded", 1: "w0 table ready", 48: "w0 prologue done", 6: "w0 rows done (newton init next)", 7: "w0 newton init done",
         16: "w0 it0 start", 51: "w0 at (4)", 53: "w0 past (4)", 8: "w0 newton done", 30: "w0 TERMINATED BYTES STORED", 9: "w0 integrated",
         25: "w0 exit", 11: "w0 [next] fk/poses in", 12: "w0 [next] crb", 13: "w0 [next] rne+M", 5: "w0 [next] dyn done", 10: "w0 [next] top of pass",
         54: "w1 rows loaded = at (3)", 55: "w1 hessian done = at (4)", 56: "w1 past (5)", 57: "w1 closing fk done = at (6)", 58: "w1 [next] geoms placed", 59: "w1 [next] broadphase done", 60: "w1 [next] plane-box done", 61: "w1 [next] box-box done", 40: "w1 [next] past (1)",
         41: "w1 [next] inertias out", 42: "w1 [next] detection done", 45: "w1 [next] contacts done", 46: "w1 [next] past (3)", 47: "w1 [next] rows stored",
         43: "w1 [next] 43", 44: "w1 [next] 44", 49: "w0 49", 50: "w0 50", 52: "w0 52", 3: "w0 3", 4: "w0 4", 2: "w0 2", 32: "w0 32", 33: "w0 33", 34: "w0 34", 35: "w0 35",
         22: "w0 22", 23: "w0 23", 14: "w0 it0 14", 15: "w0 it0 15", 17: "w0 it0 17", 18: "w0 it0 18", 19: "w0 it0 19", 20: "w0 it0 20", 21: "w0 it0 end"}
acc = {}
n = 0
for rep in range(40):
    for t in range(30):
        env.step(acts[(rep + t) % 25])
    prof = torch.zeros(160, dtype=torch.int64, device=task.device)
    assert lib.mir_debug_profile_next_step(mir.h, C.c_void_p(prof.data_ptr())) == 0
    env.step(acts[rep % 25])
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if p[24] == 0:
        continue
    n += 1
    for k in list(range(0, 26)) + [30] + list(range(32, 36)) + list(range(40, 62)):
        if p[k] > 0:
            acc.setdefault(k, []).append(p[k] - p[24])
    for it in range(8):
        if p[64 + 8 * it] > 0:
            acc.setdefault(64 + 8 * it, []).append(p[64 + 8 * it] - p[24])
print(f"split_step {mir.split_step}; {n} profiled launches; cycles after entry (shader clock) of workgroup 0")
rows = sorted((np.mean(v), k, len(v)) for k, v in acc.items())
end = max(r[0] for r in rows)
for c, k, cnt in rows:
    nm = NAMES.get(k, f"w0 iteration {(k - 64) // 8} start" if k >= 64 else str(k))
    print(f"  {c:9.0f}  {100 * c / end:5.1f} %   {nm}   (n={cnt})")
