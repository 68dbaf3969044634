"""Prints a digest of the pick scene's state after 300 fused steps and 300 env.step-style (rotated) steps from a fixed seed: two builds
of the library that claim bit-identical results must print the same line (run once per build on the same GPU box)."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B = 4096
sc = MirScene(models.franka_cube_pick_scene().build(), B)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
pos[::2, 2] = 0.14
sc.reset(pos, np.tile(np.array([1, 0, 0, 0], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
acts = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (64, B, 9)).astype(np.float32), device=sc.device)
bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
h = hashlib.sha256()
for t in range(300):
    sc.step_fused(acts[t % 64], *bufs)
for t in range(300):
    sc.step_begin(acts[t % 64], *bufs); sc.step_end()
    if t % 50 == 0:
        for x in bufs: h.update(x.cpu().numpy().tobytes())
for x in sc.get_state(): h.update(x.cpu().numpy().tobytes())
print("state digest", h.hexdigest()[:24], "spec", sc.spec_active)
