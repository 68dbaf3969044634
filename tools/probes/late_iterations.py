"""What Newton iterations 3 and 4 buy.  Same states, same actions, one step with the iteration cap at 50 / 3 / 2: the constrained
acceleration of the envs that ran more iterations than the cap, compared with the uncapped one, relative to the env's largest
|qacc| -- to be read against the float32 floor of the solve itself (kernel vs the float64 QP minimiser: median 1e-5, 99 % 7e-4 of
the same scale, tests/test_gpu_parity.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B = 4096
scs = {}
for cap in (50, 3, 2):
    sb = models.franka_cube_pick_scene(); sb.opt["iterations"] = cap
    scs[cap] = MirScene(sb.build(), B); scs[cap].set_diag(True)
full = scs[50]
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
full.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
g = torch.Generator(device=full.device).manual_seed(1)
bufs = {c: (s.empty(9), s.empty(11), s.empty(), s.empty(dtype=torch.uint8)) for c, s in scs.items()}
rel = {3: [], 2: []}
for t in range(150):
    a = torch.empty((B, 9), device=full.device).uniform_(-1, 1, generator=g)
    q, v, tg, ws = full.get_state()
    for c in (3, 2):
        scs[c].set_state(qpos=q, qvel=v, target=tg, warmstart=ws)
    for c, s in scs.items():
        s.step_fused(a, *bufs[c])
    if t < 30:
        continue
    it = full.get_diag()[2].cpu().numpy()
    qa = full.get_state()[3].cpu().numpy().astype(np.float64)
    scale = np.abs(qa).max(1) + 1e-9
    for c in (3, 2):
        qc = scs[c].get_state()[3].cpu().numpy().astype(np.float64)
        d = np.abs(qc - qa).max(1) / scale
        sel = it > c
        if sel.any():
            rel[c].append(d[sel])
for c in (3, 2):
    d = np.concatenate(rel[c])
    print(f"cap {c}: {len(d)} env-steps ran more than {c} iterations ({len(d) / (120 * B):.4%}); max |dqacc| / max |qacc|: median {np.median(d):.2e}  90 % {np.percentile(d, 90):.2e}  99 % {np.percentile(d, 99):.2e}  max {d.max():.2e}")
