"""How many workgroups send their terminated bytes early, step by step, on the headline workload (CubePick-v0, 4096 envs, U(-1,1)
actions through step_begin / step_end) and on the tiled grasp fixture; with the iteration counts of the same steps."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(root, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B = 4096
spec = models.franka_cube_pick_scene().build()
sc = MirScene(spec, B)
sc.set_diag(True)
HOME = np.array(models.FRANKA_HOME, np.float32)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)); arm = np.tile(HOME, (B, 1))
sc.reset(pos, quat, arm)
gen = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.empty((200, B, 9), dtype=torch.float32, device="cuda").uniform_(-1.0, 1.0, generator=gen)
b = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
sc.early_mask_stats(reset=True)
last = 0
fr, it1 = [], []
for t in range(200):
    sc.step_begin(acts[t], *b); sc.step_end()
    s = sc.early_mask_stats()[0]
    ni = sc.get_diag()[2].cpu().numpy()
    fr.append((s - last) / (B // 4)); last = s
    it1.append((ni.reshape(-1, 4).max(1) <= 1).mean())
fr, it1 = np.array(fr), np.array(it1)
print("random actions: fraction of workgroups that sent early per step: mean %.3f min %.3f; first 12 steps %s" % (fr.mean(), fr.min(), np.round(fr[:12], 2).tolist()))
print("  fraction of workgroups whose four envs needed <= 1 iteration: mean %.3f; correlation with not-early %.3f" % (it1.mean(), np.corrcoef(1 - fr, it1)[0, 1]))
