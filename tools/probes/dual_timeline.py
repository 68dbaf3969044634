"""(-DMIR_PROFILE_SINGLE build) time line of the two waves of workgroup 0 of the 16-lane kernel: when each wave reaches and leaves the
four barriers.  argv[1] = grasp for the contact-rich state."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "tools")]
from phase_profile import setup, setup_grasp
B = 4096
sc, acts, bufs = setup_grasp(B) if len(sys.argv) > 1 and sys.argv[1] == "grasp" else setup(B)
sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
sc.lib.mir_debug_profile_step.restype = C.c_int
acc = np.zeros(16); n = 20
for k in range(n):
    sc.set_pd_targets(acts[k % 64])
    prof = torch.zeros(160, dtype=torch.int64, device=sc.device); prof[29] = 2**62
    sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
    p = prof.cpu().numpy().astype(np.float64); t0 = p[24]
    acc += np.array([p[48], p[1], p[49], p[50], p[52], p[51], p[53], p[25], p[40], p[41], p[42], p[43], p[44], p[45], p[46], p[47]]) - t0
acc /= n
names = ["w0 at (1)", "w0 past (1)", "w0 at (2)", "w0 at (3)", "w0 past (3)", "w0 at (4)", "w0 past (4)", "w0 exit",
         "w1 at (1)", "w1 past (1)", "w1 detection done", "w1 at (2)", "w1 past (2)", "w1 contacts+J done = at (3)", "w1 past (3)", "w1 hessian done = at (4)"]
for nm, c in zip(names, acc):
    print(f"  {nm:28s} {c:8.0f}")
