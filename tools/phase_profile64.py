"""Phase timing of the wave-per-env kernel (shader clock of env 0): GPU probe for the stack tasks."""
import ctypes as C, os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B)
env.reset(seed=0)
task = env._env; sc = task._mir; sc.set_diag(True)
sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
sc.lib.mir_debug_profile_step.restype = C.c_int
home = task._home
if len(sys.argv) > 2:  # canonical layout for env 0: five cubes resting apart on the slab, arm at home (no coupling)
    from gym_genesis.backend import models
    pos = torch.from_numpy(task.sample_spawn()).to(sc.device).float()
    pos[0] = torch.tensor([[0.1 + 0.12 * i - 0.3, (-1) ** i * 0.15, models.STACK_CUBE_Z] for i in range(5)], device=sc.device)
    if sys.argv[2] == "coupled":  # cube 2 (block 1) and cube 3 (block 2) touching side by side: one coupled pair of blocks
        pos[0, 2, 0] = pos[0, 1, 0] + 0.0399; pos[0, 2, 1] = pos[0, 1, 1]
    sc.reset(pos, task._quat, home)
for t in range(30): task.step_raw(home)
if len(sys.argv) > 2 and sys.argv[2] == "real":  # (-DMIR_PROFILE_SINGLE build) env 0 := a copy of an env whose blocks couple (bit 30 of the nefc slot)
    raw = sc.get_diag()[1].cpu().numpy()
    e = int(np.nonzero((raw >> 30) & 1)[0][int(sys.argv[3]) if len(sys.argv) > 3 else 0])
    st = [x.clone() for x in sc.get_state()]
    for x in st: x[0] = x[e]
    sc.set_state(*st)
    print("env 0 := env", e)
    for t in range(3): task.step_raw(home)
dg = sc.get_diag()
print("ncon hist", np.bincount(dg[0].cpu().numpy())[15:40], "niter hist", np.bincount(dg[2].cpu().numpy()))
names = ["load", "fk/cache", "cdof+cinert", "vel+crb", "rne+M", "smooth solve", "geom+broad", "plane-box", "box-box", "compact+finish", "J + limit rows",
         "warm start", "grad(it0)", "hessian(it0)", "GJ64(it0)", "linesearch(it0)", "rest of newton", "integrate", "fk", "store+obs"]
acc = np.zeros(19)
n = 20
dual = np.zeros(12)
fine = np.zeros(6)
for r in range(n):
    prof = torch.zeros(48, dtype=torch.int64, device=sc.device)
    sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if p[24] > 0:  # a -DMIR_PROFILE_SINGLE build: the two-wave single-step instantiation; stamps 6..8 and 24..28 are wave 1's
        t0 = p[0]
        dual += np.array([p[1] - t0, p[5] - p[1], p[9] - p[5], p[10] - p[9], p[25] - t0, p[6] - p[25], p[7] - p[6], p[8] - p[7], p[26] - p[8],
                          p[27] - p[26], p[28] - p[27], p[19] - t0])
        p[6] = p[7] = p[8] = p[5]
        fine += np.array([p[20] - p[11], p[21] - p[20], p[22] - p[21], p[12] - p[22], p[23] - p[12], p[29] - p[28]])
    acc += np.diff(p[:20])
    if p[36] > 0 and r == 0: print(  # (stamps 36 - 38: around the loop of the 15-axis routine, its candidate count)
        f'   box-box loop: {p[37]-p[36]:.0f} cycles for {int(p[38])} candidates of the 15-axis routine ({(int(p[38]) + 3) // 4} trips)')
    if p[30] > 0: print(f'   it0: H walk {p[30]-p[13]:.0f}, coupled_solve {p[31]-p[30]:.0f} cycles') if r == 0 else None
acc /= n
if dual.any():
    dual /= n
    for nm, c in zip(["w0 load+fk -> (1)", "w0 dynamics", "w0 wait (2)", "w0 J + rows (3)", "w1 tables -> (1)", "w1 geom+broad", "w1 plane-box", "w1 box-box",
                      "w1 compact+finish", "w1 wait (2)", "w1 J half", "whole step"], dual):
        print(f"{nm:18s} {c:9.0f} cycles")
    for nm, c in zip(["w0 coupling graph", "w0 row forces", "w0 gradient loop", "w0 |g| + test", "w0 wait (4)", "w1 hessian"], fine / n):
        print(f"{nm:18s} {c:9.0f} cycles")
for nm, c in zip(names[1:], acc):
    print(f"{nm:18s} {c:9.0f} cycles")
print(f"{'total':18s} {acc.sum():9.0f} cycles; diag: ncon {sc.get_diag()[0][0].item()} niter {sc.get_diag()[2][0].item()}")
