"""Scene descriptions (host data only) for the tasks on the hot path.

``franka_cube_pick_scene`` restates what the reference builds with
``gs.morphs.Plane()``, ``gs.morphs.MJCF("xml/franka_emika_panda/panda.xml")`` and
``gs.morphs.Box(size=0.04^3, pos=(0.65, 0, 0.02))``
(/root/reference/gym_genesis/tasks/franka/cube_pick.py:37-54).

Provenance of the numbers: the Panda MJCF ships inside
the external Genesis package and is NOT available here, so the kinematic tree,
joint ranges and inertial parameters below are re-stated from the public
MuJoCo-Menagerie Franka Emika Panda description (UPSTREAM-RECALL, unverified);
the collision meshes are replaced by boxes.  PD gains / force ranges are the ones
the reference itself spells out for the same robot
(/root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:101-106).
Every value is a field of the scene spec, so a captured panda.xml can be dropped
in without touching the kernels.
"""
from __future__ import annotations

import math

from .spec import (CTRL_POSITION, GEOM_BOX, GEOM_PLANE, JNT_FREE, JNT_PRISMATIC, JNT_REVOLUTE, SceneBuilder,
                   box_inertia)

# reference: cube_pick.py:7-17
FRANKA_JOINTS = ("joint1", "joint2", "joint3", "joint4", "joint5", "joint6", "joint7", "finger_joint1",
                 "finger_joint2")
# reference: cube_pick.py:100
FRANKA_HOME = (0.0, -0.4, 0.0, -2.2, 0.0, 2.0, 0.8, 0.04, 0.04)
# reference: cube_stack_kitchen_batch.py:101-106
FRANKA_KP = (4500.0, 4500.0, 3500.0, 3500.0, 2000.0, 2000.0, 2000.0, 100.0, 100.0)
FRANKA_KV = (450.0, 450.0, 350.0, 350.0, 200.0, 200.0, 200.0, 10.0, 10.0)
FRANKA_FRC = (87.0, 87.0, 87.0, 87.0, 87.0, 87.0, 87.0, 100.0, 100.0)

_S = math.sqrt(0.5)

# name, parent, pos, quat(wxyz), range, mass, com, fullinertia(xx yy zz xy xz yz)
_PANDA_LINKS = (
    ("link1", "link0", (0, 0, 0.333), (1, 0, 0, 0), (-2.8973, 2.8973), 4.970684, (0.003875, 0.002081, -0.04762),
     (0.70337, 0.70661, 0.0091170, -0.00013900, 0.0067720, 0.019169)),
    ("link2", "link1", (0, 0, 0), (_S, -_S, 0, 0), (-1.7628, 1.7628), 0.646926, (-0.003141, -0.02872, 0.003495),
     (0.0079620, 2.8110e-2, 2.5995e-2, -3.925e-3, 1.0254e-2, 7.04e-4)),
    ("link3", "link2", (0, -0.316, 0), (_S, _S, 0, 0), (-2.8973, 2.8973), 3.228604, (2.7518e-2, 3.9252e-2, -6.6502e-2),
     (3.7242e-2, 3.6155e-2, 1.083e-2, -4.761e-3, -1.1396e-2, -1.2805e-2)),
    ("link4", "link3", (0.0825, 0, 0), (_S, _S, 0, 0), (-3.0718, -0.0698), 3.587895, (-5.317e-2, 1.04419e-1, 2.7454e-2),
     (2.5853e-2, 1.9552e-2, 2.8323e-2, 7.796e-3, -1.332e-3, 8.641e-3)),
    ("link5", "link4", (-0.0825, 0.384, 0), (_S, -_S, 0, 0), (-2.8973, 2.8973), 1.225946, (-1.1953e-2, 4.1065e-2, -3.8437e-2),
     (3.5549e-2, 2.9474e-2, 8.627e-3, -2.117e-3, -4.037e-3, 2.29e-4)),
    ("link6", "link5", (0, 0, 0), (_S, _S, 0, 0), (-0.0175, 3.7525), 1.666555, (6.0149e-2, -1.4117e-2, -1.0517e-2),
     (1.964e-3, 4.354e-3, 5.433e-3, 1.09e-4, -1.158e-3, 3.41e-4)),
    ("link7", "link6", (0.088, 0, 0), (_S, _S, 0, 0), (-2.8973, 2.8973), 0.735522, (1.0517e-2, -4.252e-3, 6.1597e-2),
     (1.2516e-2, 1.0027e-2, 4.815e-3, -4.28e-4, -1.196e-3, -7.41e-4)),
)

# box stand-ins for the link collision meshes: (body, half extents, centre in body frame)
_PANDA_BOXES = (
    ("link1", (0.055, 0.055, 0.10), (0.0, -0.02, -0.09)),
    ("link2", (0.055, 0.10, 0.055), (0.0, -0.07, 0.02)),
    ("link3", (0.055, 0.055, 0.09), (0.03, 0.02, -0.07)),
    ("link4", (0.055, 0.09, 0.055), (-0.04, 0.06, 0.02)),
    ("link5", (0.05, 0.06, 0.13), (0.0, 0.04, -0.12)),
    ("link6", (0.06, 0.05, 0.05), (0.05, 0.01, 0.0)),
    ("link7", (0.045, 0.045, 0.04), (0.0, 0.0, 0.07)),
    ("hand", (0.0316, 0.102, 0.033), (0.0, 0.0, 0.033)),
)
# finger body + the large fingertip pad (Menagerie pad box 1), in finger frame
_FINGER_BODY_BOX = ((0.0105, 0.0085, 0.0268), (0.0, 0.0145, 0.0268))
_FINGER_PAD_BOX = ((0.0085, 0.004, 0.0085), (0.0, 0.0055, 0.0445))


def franka_cube_pick_scene(cube_size=0.04, cube_pos=(0.65, 0.0, 0.02), cube_rho=200.0) -> SceneBuilder:
    sb = SceneBuilder()
    # ground plane (gs.morphs.Plane, cube_pick.py:50)
    sb.add_geom(0, GEOM_PLANE)
    # Panda (cube_pick.py:51)
    sb.add_body("link0", 0, mass=0.629769, ipos=(-0.041018, -0.00014, 0.049974),
                inertia=(0.00315, 0.00388, 0.004285, 8.2904e-7, 0.00015, 8.2299e-6))
    for i, (name, parent, pos, quat, rng, mass, com, inertia) in enumerate(_PANDA_LINKS):
        sb.add_body(name, parent, pos=pos, quat=quat, jtype=JNT_REVOLUTE, axis=(0, 0, 1), mass=mass, ipos=com,
                    inertia=inertia, joint_name=FRANKA_JOINTS[i], limited=1, range=rng, armature=0.1, damping=1.0,
                    ctrl_mode=CTRL_POSITION, kp=FRANKA_KP[i], kv=FRANKA_KV[i],
                    frc_range=(-FRANKA_FRC[i], FRANKA_FRC[i]))
    sb.add_body("hand", "link7", pos=(0, 0, 0.107), quat=(0.9238795, 0, 0, -0.3826834), mass=0.73,
                ipos=(-0.01, 0, 0.03), inertia=(0.001, 0.0025, 0.0017, 0, 0, 0))
    for k, (name, quat) in enumerate((("left_finger", (1, 0, 0, 0)), ("right_finger", (0, 0, 0, 1)))):
        sb.add_body(name, "hand", pos=(0, 0, 0.0584), quat=quat, jtype=JNT_PRISMATIC, axis=(0, 1, 0), mass=0.015,
                    inertia=(2.375e-6, 2.375e-6, 7.5e-7, 0, 0, 0), joint_name=FRANKA_JOINTS[7 + k], limited=1,
                    range=(0.0, 0.04), armature=0.1, damping=1.0, ctrl_mode=CTRL_POSITION, kp=FRANKA_KP[7 + k],
                    kv=FRANKA_KV[7 + k], frc_range=(-FRANKA_FRC[7 + k], FRANKA_FRC[7 + k]))
    for body, half, centre in _PANDA_BOXES:
        sb.add_geom(body, GEOM_BOX, size=half, pos=centre)
    for finger in ("left_finger", "right_finger"):
        sb.add_geom(finger, GEOM_BOX, size=_FINGER_BODY_BOX[0], pos=_FINGER_BODY_BOX[1])
        sb.add_geom(finger, GEOM_BOX, size=_FINGER_PAD_BOX[0], pos=_FINGER_PAD_BOX[1])
    # cube (gs.morphs.Box, cube_pick.py:52-54); Genesis rigid material default density 200 kg/m^3
    h = cube_size / 2
    mass = cube_rho * cube_size ** 3
    sb.add_body("cube", 0, pos=cube_pos, quat=(1, 0, 0, 0), jtype=JNT_FREE, mass=mass,
                inertia=box_inertia(mass, (h, h, h)))
    sb.add_geom("cube", GEOM_BOX, size=(h, h, h))
    # task extraction (cube_pick.py:66-68,134,142)
    sb.task = dict(eef_body=sb.body_index("hand"), obj_body=sb.body_index("cube"),
                   grip_dof=(sb.dof_index("finger_joint1"), sb.dof_index("finger_joint2")), reward_z=0.1)
    return sb
