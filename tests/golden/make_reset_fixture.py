"""Generates tests/golden/reset_rng.json: the cube spawn stream of the reference's reset().

Restates two lines of /root/reference/gym_genesis/tasks/franka/cube_pick.py (90-91):
    x = self._random.uniform(0.45, 0.80, size=(B,)); y = self._random.uniform(-0.25, 0.25, size=(B,))
with self._random = np.random.RandomState(seed) (cube_pick.py:117), z = 0.02 (92), cast to float32 (93).
NumPy's legacy MT19937 RandomState stream is frozen, so these values are what the reference draws.
SURVEY.md 8c-1 quotes seed 0, B=4 -> x=[0.64208473 0.70031628 0.66096718 0.64070911].
"""
import json
import os

import numpy as np

out = {}
for seed, B in ((0, 4), (0, 4096), (42, 10), (7, 1)):
    rs = np.random.RandomState(seed)
    x = rs.uniform(0.45, 0.80, size=(B,))
    y = rs.uniform(-0.25, 0.25, size=(B,))
    x2 = rs.uniform(0.45, 0.80, size=(B,))  # second reset() continues the same stream
    pos = np.stack([x, y, np.full(B, 0.02)], 1).astype(np.float32)
    out[f"seed{seed}_B{B}"] = {
        "seed": seed, "B": B,
        "first_rows_f32": pos[:4].tolist(), "last_row_f32": pos[-1].tolist(),
        "sum_x_f64": float(x.sum()), "sum_y_f64": float(y.sum()), "second_reset_x0_f64": float(x2[0]),
    }
with open(os.path.join(os.path.dirname(__file__), "reset_rng.json"), "w") as f:
    json.dump(out, f, indent=1)
