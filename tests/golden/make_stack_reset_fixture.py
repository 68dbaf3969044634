"""Generates tests/golden/stack_reset_rng.json: the cube spawn streams of the two stack tasks' reset().

Restates the RNG draws (and only those) of
  /root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:71-91  (vector draws x1,y1,x2,y2, then xd,yd per distractor)
  /root/reference/gym_genesis/tasks/so101/cube_stack_batch.py:72-103          (per-env scalar rejection loop, then vector distractors)
with self._random = np.random.RandomState(seed).  NumPy's legacy MT19937 stream is frozen, so these are the values the
reference draws.  z = island_top_z + 0.02 + 0.001 with island_top_z = 0.7000312834978104
(/root/reference/examples/franka/stack_cube_one_image.py:38).
"""
import json
import os

import numpy as np

TOP = 0.7000312834978104
Z = TOP + 0.02 + 0.001


def franka(seed, B):
    r = np.random.RandomState(seed)
    x1 = r.uniform(-0.3, -0.1, size=(B,)); y1 = r.uniform(-0.15, 0.15, size=(B,))
    x2 = r.uniform(-0.3, -0.1, size=(B,)); y2 = r.uniform(-0.15, 0.15, size=(B,))
    cols = [np.stack([x1, y1, np.full(B, Z)], 1), np.stack([x2, y2, np.full(B, Z)], 1)]
    for _ in range(3):
        xd = r.uniform(-0.35, 0.0, size=(B,)); yd = r.uniform(-0.2, 0.2, size=(B,))
        cols.append(np.stack([xd, yd, np.full(B, Z)], 1))
    return np.stack(cols, 1).astype(np.float32), float(r.uniform())


def so101(seed, B):
    r = np.random.RandomState(seed)
    p1, p2 = [], []
    for _ in range(B):
        while True:
            x1 = r.uniform(-0.3, -0.1); y1 = r.uniform(-0.1, 0.1)
            x2 = r.uniform(-0.3, -0.1); y2 = r.uniform(-0.1, 0.1)
            if ((x2 - x1) ** 2 + (y2 - y1) ** 2) ** 0.5 >= 0.06:
                p1.append((x1, y1, Z)); p2.append((x2, y2, Z))
                break
    cols = [np.array(p1), np.array(p2)]
    for _ in range(3):
        xd = r.uniform(-0.35, 0.0, size=(B,)); yd = r.uniform(-0.2, 0.2, size=(B,))
        cols.append(np.stack([xd, yd, np.full(B, Z)], 1))
    return np.stack(cols, 1).astype(np.float32), float(r.uniform())


out = {}
for name, fn in (("franka", franka), ("so101", so101)):
    for seed, B in ((0, 4), (3, 33), (11, 1)):
        pos, nxt = fn(seed, B)
        out[f"{name}_seed{seed}_B{B}"] = {"robot": name, "seed": seed, "B": B, "first_env_f32": pos[0].tolist(), "last_env_f32": pos[-1].tolist(),
                                          "sum_f64": float(pos.astype(np.float64).sum()), "next_uniform_f64": nxt}
with open(os.path.join(os.path.dirname(__file__), "stack_reset_rng.json"), "w") as f:
    json.dump(out, f, indent=1)
