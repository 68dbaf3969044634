"""Developer tool (needs a -DMIR_PROFILE_SINGLE build): phase timing of the SLOWEST workgroups of a launch.  A launch ends with
its last workgroup, and that one is a workgroup whose env needs 3-4 Newton iterations: this tool finds such a workgroup (step,
read the per-env iteration counts, restore the state), repeats the very same step with the stamps on that workgroup, and prints
the cycles of every Newton iteration split into its phases, next to the workgroup's wall-clock exit time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "tools")]
from phase_profile import setup  # noqa: E402

NAMES = ["row forces", "gradient", "|g| + test", "H update", "GJ solve", "M s, J s", "line search", "update + tests"]


def main():
    B = 4096
    sc, acts, bufs = setup(B)
    sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    sc.lib.mir_debug_profile_step.restype = C.c_int
    per_it = {}
    exits = []
    k = 0
    samples = 0
    while samples < 24 and k < 400:
        k += 1
        st = [x.clone() for x in sc.get_state()]
        sc.step_fused(acts[k % 64], *bufs)
        ni = sc.get_diag()[2].cpu().numpy()
        nc = sc.get_diag()[0].cpu().numpy()
        e = int(np.argmax(ni))
        if ni[e] < 3:
            continue
        st_after = [x.clone() for x in sc.get_state()]
        sc.set_state(qpos=st[0], qvel=st[1], target=st[2], warmstart=st[3])
        sc.set_pd_targets(acts[k % 64])
        prof = torch.zeros(160, dtype=torch.int64, device=sc.device)
        prof[29] = 2**62
        prof[63] = e // 4
        sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
        p = prof.cpu().numpy().astype(np.float64)
        ni2 = sc.get_diag()[2].cpu().numpy()
        if ni2[e] != ni[e]:
            continue
        samples += 1
        start = p[7]
        for it in range(int(ni[e])):
            s = p[64 + 8 * it: 72 + 8 * it]
            if s[7] == 0:
                break
            d = np.diff(np.concatenate([[start], s]))
            per_it.setdefault(it, []).append(d)
            start = s[7]
        exits.append(((p[27] - p[26]) / 100.0, (p[28] - p[26]) / 100.0, p[25] - p[24], ni[e], nc[e], p[8] - p[7]))
        sc.set_state(qpos=st_after[0], qvel=st_after[1], target=st_after[2], warmstart=st_after[3])
    print(f"{samples} straggler workgroups profiled (an env with >= 3 Newton iterations)")
    for it in sorted(per_it):
        d = np.mean(per_it[it], axis=0)
        print(f"  iteration {it} (n={len(per_it[it])}): total {d.sum():6.0f} cycles | " + " | ".join(f"{n} {c:.0f}" for n, c in zip(NAMES, d)))
    ex = np.array(exits)
    print(f"  this workgroup exits {ex[:,0].mean():.1f} us after its entry; last workgroup of its XCD {ex[:,1].mean():.1f} us; whole kernel {ex[:,2].mean():.0f} cycles; "
          f"newton loop {ex[:,5].mean():.0f} cycles; niter {ex[:,3].mean():.2f}, contacts {ex[:,4].mean():.1f}")


if __name__ == "__main__":
    main()
