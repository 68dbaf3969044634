"""(-DMIR_PROFILE_SINGLE build) time line of one workgroup of the SECOND-HALF launch of an overflow run (mir_step_kernel<9, ., 3>: rows from
the scratch rows, solver, outputs) on the reference's expert episode at 4096 envs, envs in their natural order (MIR_EXACT_HEAVY_SORT=0):
stamps of both waves in shader cycles after the kernel's entry, averaged over the profiled launches whose watched workgroup held at
least `min_contacts` contacts in one of its envs.  Usage: python3 tools/probes/big_timeline.py [min_contacts]"""
import ctypes as C, importlib.util, os, sys
os.environ["MIR_EXACT_HEAVY_SORT"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv

spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
B = 4096
MINC = int(sys.argv[1]) if len(sys.argv) > 1 else 24
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
mir = env._env._mir
mir.set_diag(True)
lib = mir.lib
lib.mir_debug_profile_next_step.argtypes = [C.c_void_p, C.c_void_p]
lib.mir_debug_profile_next_step.restype = C.c_int
NAMES = {24: "w0 entry", 0: "w0 state loaded", 48: "w0 at (1)", 3: "w0 M + bias done", 49: "w0 at (2)", 4: "w0 smooth solve done", 50: "w0 limit rows done = at (3)", 52: "w0 past (3)",
         6: "w0 contact rows done", 7: "w0 warm start done", 16: "w0 it0 forces", 17: "w0 it0 gradient", 51: "w0 at (4)", 53: "w0 past (4)", 18: "w0 it0 hessian", 15: "w0 it0 direction",
         19: "w0 it0 J s", 20: "w0 it0 search", 21: "w0 it0 end", 8: "w0 newton done", 30: "w0 TERMINATED BYTES STORED", 9: "w0 integrated", 10: "w0 outputs start", 25: "w0 outputs stored",
         40: "w1 opening fk done = at (1)", 41: "w1 past (1)", 58: "w1 geoms placed", 59: "w1 broadphase done", 60: "w1 plane-box done", 61: "w1 box-box done", 42: "w1 detection done",
         22: "w1 compaction done", 43: "w1 contacts computed = at (2)", 44: "w1 past (2)", 5: "w1 contact stores done", 45: "w1 jacobians done = at (3)", 46: "w1 past (3)",
         47: "w1 all-active hessian done = at (4)"}
acc, ncs, its, n_all = {}, [], [], 0
prev = None
obs, _ = env.reset(seed=0)
t = 0
for stage in ex.STAGES:
    for _ in range(40):
        a = ex.expert_policy(env.get_robot(), obs, stage)
        prof = None
        if prev is not None and t >= 85 and mir.exact_route()["big_steps"] > 0:
            # watch a workgroup that held many contacts in the previous step
            wg = int(torch.argmax(prev.view(-1, 4).max(1).values).item())
            prof = torch.zeros(512, dtype=torch.int64, device=mir.device)
            prof[63] = wg
            assert lib.mir_debug_profile_next_step(mir.h, C.c_void_p(prof.data_ptr())) == 0
        obs, r, term, _, _ = env.step(a)
        dg = mir.get_diag(points=True)
        prev = dg[0].clone()
        if prof is not None:
            torch.cuda.synchronize()
            p = prof.cpu().numpy().astype(np.float64)
            wg = int(p[63])
            nc = int(dg[0][4 * wg:4 * wg + 4].max()); it = int(dg[2][4 * wg:4 * wg + 4].max())
            n_all += 1
            if p[24] > 0 and p[30] > 0 and nc >= MINC:
                ncs.append(nc); its.append(it)
                for k in range(160):
                    if k != 63 and p[k] > 0 and k not in (26, 27, 28, 29, 31, 130):
                        acc.setdefault(k, []).append(p[k] - p[24])
        t += 1
print(f"route {mir.exact_route()}; {len(ncs)} of {n_all} profiled launches with >= {MINC} contacts in the watched workgroup: contacts mean {np.mean(ncs):.1f} max {max(ncs)}, Newton iterations (max of the 4 envs) mean {np.mean(its):.2f} max {max(its)}")
rows = sorted((np.mean(v), k, len(v)) for k, v in acc.items())
end = max(r[0] for r in rows)
for c, k, cnt in rows:
    nm = NAMES.get(k, f"w0 iteration {(k - 64) // 8} stamp {(k - 64) % 8}" if 64 <= k < 128 else str(k))
    print(f"  {c:9.0f}  {100 * c / end:5.1f} %   {nm}   (n={cnt})")
