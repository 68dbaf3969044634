"""Developer tool (needs `make -C gym-genesis_amd/csrc EXTRA=-DMIR_DEBUG_TRACE`): the 16-lane kernel's Newton solver, iteration by
iteration, for ONE env of a saved state -- per lane: qacc, the four pyramid rows' residuals of the lane's contact, its active flags,
gradient, Newton direction, step lengths, improvement.  Usage: python tools/solver_trace.py state.npz env
(state.npz: q (B,nq), v (B,nv), ws (B,nv), act (B,nu) float32, e.g. written by a parity test that found an outlier)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "gym-genesis_amd"))
from gym_genesis.backend import models  # noqa: E402
from gym_genesis.backend.lib import MirScene  # noqa: E402

d = np.load(sys.argv[1])
env = int(sys.argv[2])
q, v, ws, act = d["q"], d["v"], d["ws"], d["act"]
B = q.shape[0]
sc = MirScene(models.franka_cube_pick_scene().build(), B)
sc.set_state(qpos=q, qvel=v, target=act, warmstart=ws)
prof = torch.zeros(256 + 8 * 16 * 16 // 2 + 64, dtype=torch.int64, device=sc.device)
prof[255] = env
sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
torch.cuda.synchronize()
tr = prof[256:].view(torch.float32).cpu().numpy()[: 8 * 16 * 16].reshape(8, 16, 16)
np.set_printoptions(linewidth=220, precision=5, suppress=True)
names = ["qacc", "jar0", "jar1", "jar2", "jar3", "bits", "g", "sv", "alpha", "ald", "alc", "improv", "ncross", "done", "flipmask", "partial"]
for it in range(8):
    if not tr[it].any():
        break
    print(f"---- iteration {it}: alpha {tr[it, 0, 8]:.6f} improvement {tr[it, 0, 11]:.6e} ncross {tr[it, 0, 12]} done {tr[it, 0, 13]} flipmask {int(tr[it, 0, 14])} partial {tr[it, 0, 15]}")
    for k in (0, 6, 7, 9):
        print(f"  {names[k]:6s} (lane = dof)    ", tr[it, :, k])
    for k in (1, 2, 3, 4, 5, 10):
        print(f"  {names[k]:6s} (lane = contact)", tr[it, :6, k])
