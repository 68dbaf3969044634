"""(-DMIR_PROFILE_SINGLE build) time line of one workgroup of the LIST INSTANTIATION of exact contacts (mir_step_kernel<6, ., 3>): the
scripted grasp of tests/golden/grasp_targets.json on a scene whose every env takes the deferred envs' route (set_exact_contacts("all")),
profiled in the closed-grasp stage (pads on the cube, fingertips on the floor: 20+ candidate points).  Every stamp both waves left in
the FIRST pass (the whole step), sorted, in shader cycles after the kernel's entry; the end of the second pass (next step's
action-independent half).  Usage: python3 tools/probes/list_timeline.py [num_envs] [first_step] [last_step]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t0, t1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (125, 160)
G_ = json.load(open(os.path.join(ROOT, "tests", "golden", "grasp_targets.json")))
T = np.array(G_["targets"], np.float32)
pos4 = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
acts4 = np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0)
rep = n // 4
pos = np.tile(pos4, (rep, 1)); acts = np.tile(acts4, (1, rep, 1))
sc = MirScene(models.franka_cube_pick_scene().build(), n)
sc.set_exact_contacts("all")
sc.set_diag(True)
sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (n, 1)))
lib = sc.lib
lib.mir_debug_profile_next_list_step.argtypes = [C.c_void_p, C.c_void_p]
lib.mir_debug_profile_next_list_step.restype = C.c_int
bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
d = torch.as_tensor(acts, device=sc.device)
NAMES = {24: "w0 entry", 0: "w0 state loaded", 1: "w0 table ready", 48: "w0 at (1)", 32: "w0 subspaces", 33: "w0 velocities", 2: "w0 crb", 34: "w0 cddq", 35: "w0 body forces",
         3: "w0 M + bias done", 49: "w0 at (2)", 4: "w0 smooth solve done", 50: "w0 limit rows done = at (3)", 52: "w0 past (3)", 6: "w0 contact rows done", 7: "w0 warm start done",
         16: "w0 it0 forces", 17: "w0 it0 gradient", 14: "w0 it0 converged?", 51: "w0 at (4)", 53: "w0 past (4)", 18: "w0 it0 hessian", 15: "w0 it0 direction", 19: "w0 it0 J s", 20: "w0 it0 search",
         21: "w0 it0 end", 8: "w0 newton done", 30: "w0 TERMINATED BYTES STORED", 9: "w0 integrated", 10: "w0 outputs start", 25: "w0 outputs stored (pass 0 done)",
         40: "w1 opening fk done = at (1)", 41: "w1 past (1)", 58: "w1 geoms placed", 59: "w1 broadphase done", 60: "w1 plane-box done", 61: "w1 box-box done", 42: "w1 detection done",
         13: "w1 13", 22: "w1 compaction done", 23: "w1 cmap published", 43: "w1 contacts computed = at (2)", 44: "w1 past (2)", 5: "w1 contact stores done", 45: "w1 jacobians done = at (3)", 46: "w1 past (3)",
         47: "w1 all-active hessian done = at (4)", 142: "w1 closing fk done, second pass starts", 140: "w0 SECOND PASS DONE", 141: "w1 SECOND PASS DONE (rows stored)"}
acc, its, ncs = {}, [], []
for t in range(acts.shape[0]):
    prof = None
    if t0 <= t < t1:
        prof = torch.zeros(512, dtype=torch.int64, device=sc.device)
        prof[63] = (7 * t) % (n // 4)  # the workgroup to watch
        assert lib.mir_debug_profile_next_list_step(sc.h, C.c_void_p(prof.data_ptr())) == 0
    sc.step_fused(d[t], *bufs)
    if prof is not None:
        torch.cuda.synchronize()
        p = prof.cpu().numpy().astype(np.float64)
        if p[24] == 0:
            continue
        dg = sc.get_diag(points=True)
        wg = int(p[63])
        ncs.append(int(dg[0][4 * wg:4 * wg + 4].max())); its.append(int(dg[2][4 * wg:4 * wg + 4].max()))
        for k in range(160):
            if k != 63 and p[k] > 0:
                acc.setdefault(k, []).append(p[k] - p[24])
print(f"{n} envs, steps {t0}..{t1}: {len(ncs)} profiled launches; contacts in the watched workgroup (max of its 4 envs) mean {np.mean(ncs):.1f} max {max(ncs)}, Newton iterations mean {np.mean(its):.2f} max {max(its)}")
rows = sorted((np.mean(v), k, len(v)) for k, v in acc.items())
end = max(r[0] for r in rows)
for c, k, cnt in rows:
    nm = NAMES.get(k, f"w0 iteration {(k - 64) // 8} stamp {(k - 64) % 8}" if 64 <= k < 128 else str(k))
    print(f"  {c:9.0f}  {100 * c / end:5.1f} %   {nm}   (n={cnt})")
