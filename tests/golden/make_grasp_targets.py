"""Generates tests/golden/grasp_targets.json: joint-space PD targets of a scripted pick
(hover -> descend -> close -> lift), shaped like the reference's only in-repo usage trace of the hot
path (/root/reference/examples/franka/pick_cube_state.py:86-93: stages x 40 steps, hand pointing
down with quat (0,1,0,0), fingers 0.04 open / closing onto the cube).

The reference obtains its targets from Genesis IK, which is not available; here they come from a
damped least-squares IK on the ORACLE's forward kinematics (build-owned, test infrastructure).
Heights are chosen for this repo's box-pad finger geometry (pad centre 0.103 m below the hand frame).
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle")]
import orc  # noqa: E402
from gym_genesis.backend import models  # noqa: E402

spec = models.franka_cube_pick_scene().build()
o = orc.Oracle(spec, 1)
HAND = spec.task.eef_body
HOME = np.array(models.FRANKA_HOME)
DOWN = np.array([0.0, 1.0, 0.0, 0.0])  # wxyz: 180 deg about x -> hand z axis points at the floor


def fk(q7):
    q = o.read(orc.F_QPOS)
    q[:7] = q7
    o.write(orc.F_QPOS, q)
    o.fk()
    return o.read(orc.F_XPOS).reshape(-1, 3)[HAND], o.read(orc.F_XQUAT).reshape(-1, 4)[HAND]


def qerr(q, qd):
    """Small-angle orientation error vector between quaternions q and qd."""
    w = qd[0] * q[0] + qd[1:] @ q[1:]
    v = qd[0] * q[1:] - q[0] * qd[1:] - np.cross(qd[1:], q[1:])
    return 2 * v * np.sign(w)


def ik(target_pos, seed):
    def res(q7):
        p, q = fk(q7)
        return np.r_[p - target_pos, 0.3 * qerr(q, DOWN), 1e-3 * (q7 - seed)]

    lo = np.array([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]) + 0.02
    hi = np.array([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]) - 0.02
    sol = least_squares(res, np.clip(seed, lo, hi), bounds=(lo, hi), xtol=1e-12, ftol=1e-12)
    p, q = fk(sol.x)
    assert np.abs(p - target_pos).max() < 1e-5 and np.abs(qerr(q, DOWN)).max() < 1e-4, (p, target_pos)
    return sol.x


cubes = [(0.55, 0.0), (0.50, 0.15), (0.62, -0.12), (0.47, -0.05)]
out = {"steps_per_stage": 40, "stages": ["hover", "stabilize", "descend", "close", "lift"], "cube_xy": cubes, "targets": []}
for (x, y) in cubes:
    seed = HOME[:7].copy()
    per_env = []
    for stage, (dz, grip) in (("hover", (0.25, 0.04)), ("stabilize", (0.25, 0.04)), ("descend", (0.104, 0.04)), ("close", (0.104, 0.0)),
                               ("lift", (0.40, 0.0))):
        q7 = ik(np.array([x, y, 0.02 + dz]), seed)
        seed = q7
        per_env.append(list(map(float, q7)) + [grip, grip])
    out["targets"].append(per_env)
with open(os.path.join(os.path.dirname(__file__), "grasp_targets.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(cubes), "envs x", len(out["stages"]), "stages")
