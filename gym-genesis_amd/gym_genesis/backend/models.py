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

from .spec import (AGENT_QPOS, CTRL_POSITION, GEOM_BOX, GEOM_CAPSULE, GEOM_PLANE, JNT_FREE, JNT_PRISMATIC, JNT_REVOLUTE, MIR_MAX_CONTACT,
                   REWARD_STACK, SceneBuilder, box_inertia)

# reference: cube_pick.py:7-17
FRANKA_JOINTS = ("joint1", "joint2", "joint3", "joint4", "joint5", "joint6", "joint7", "finger_joint1",
                 "finger_joint2")
# reference: cube_pick.py:100
FRANKA_HOME = (0.0, -0.4, 0.0, -2.2, 0.0, 2.0, 0.8, 0.04, 0.04)
# reference: cube_stack_kitchen_batch.py:101-106
FRANKA_KP = (4500.0, 4500.0, 3500.0, 3500.0, 2000.0, 2000.0, 2000.0, 100.0, 100.0)
FRANKA_KV = (450.0, 450.0, 350.0, 350.0, 200.0, 200.0, 200.0, 10.0, 10.0)
# the stack task sets +-87 on all seven arm joints explicitly (cube_stack_kitchen_batch.py:103-106)
FRANKA_FRC = (87.0, 87.0, 87.0, 87.0, 87.0, 87.0, 87.0, 100.0, 100.0)
# FrankaCubePickBatch sets NO force range (cube_pick.py:99-105), so it inherits the <actuator> block of panda.xml:
# +-87 N m on joints 1-4, +-12 N m on the wrist joints 5-7, +-100 N on the fingers (UPSTREAM-RECALL)
FRANKA_FRC_MJCF = (87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0, 100.0, 100.0)

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


def _capsule_of_box(half):
    """The capsule inscribed along the longest edge of a link box: (radius, half length of the axis segment), quaternion that
    turns the capsule's z axis onto that edge."""
    k = max(range(3), key=lambda i: half[i])
    r = min(half[(k + 1) % 3], half[(k + 2) % 3])
    quat = ((_S, 0.0, _S, 0.0), (_S, -_S, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))[k]  # z -> x, z -> y, z -> z
    return (r, max(half[k] - r, 0.0)), quat


def _add_franka(sb: SceneBuilder, pos=(0.0, 0.0, 0.0), scale=1.0, frc=FRANKA_FRC_MJCF, link_shape="capsule") -> None:
    """The Panda (bodies, joints, collision boxes) mounted at `pos`, uniformly scaled like gs.morphs.MJCF(scale=...):
    lengths x s, masses x s^3, inertias x s^5, prismatic ranges x s; joint-level constants (armature, damping,
    PD gains, force ranges) and revolute ranges unchanged.  `frc`: per-joint force limits (the MJCF defaults unless the task
    overrides them with set_dofs_force_range).  `link_shape`: "box" (the stand-ins of round 1) or "capsule" -- links 1-7 as
    capsules along the longest edge of those boxes (closer to the rounded link meshes; convex narrowphase: GJK / MPR).  Hand and
    fingers stay boxes (the finger pads ARE boxes in the MJCF)."""
    s = float(scale)
    s3, s5 = s ** 3, s ** 5
    sc = lambda v: tuple(x * s for x in v)  # noqa: E731
    white, dark = (0.92, 0.92, 0.9), (0.18, 0.18, 0.2)
    sb.add_body("link0", 0, pos=pos, mass=0.629769 * s3, ipos=sc((-0.041018, -0.00014, 0.049974)),
                inertia=tuple(v * s5 for v in (0.00315, 0.00388, 0.004285, 8.2904e-7, 0.00015, 8.2299e-6)))
    for i, (name, parent, lpos, quat, rng, mass, com, inertia) in enumerate(_PANDA_LINKS):
        sb.add_body(name, parent, pos=sc(lpos), quat=quat, jtype=JNT_REVOLUTE, axis=(0, 0, 1), mass=mass * s3, ipos=sc(com),
                    inertia=tuple(v * s5 for v in inertia), joint_name=FRANKA_JOINTS[i], limited=1, range=rng, armature=0.1,
                    damping=1.0, ctrl_mode=CTRL_POSITION, kp=FRANKA_KP[i], kv=FRANKA_KV[i],
                    frc_range=(-frc[i], frc[i]))
    sb.add_body("hand", "link7", pos=sc((0, 0, 0.107)), quat=(0.9238795, 0, 0, -0.3826834), mass=0.73 * s3,
                ipos=sc((-0.01, 0, 0.03)), inertia=tuple(v * s5 for v in (0.001, 0.0025, 0.0017, 0, 0, 0)))
    for k, (name, quat) in enumerate((("left_finger", (1, 0, 0, 0)), ("right_finger", (0, 0, 0, 1)))):
        sb.add_body(name, "hand", pos=sc((0, 0, 0.0584)), quat=quat, jtype=JNT_PRISMATIC, axis=(0, 1, 0), mass=0.015 * s3,
                    inertia=tuple(v * s5 for v in (2.375e-6, 2.375e-6, 7.5e-7, 0, 0, 0)), joint_name=FRANKA_JOINTS[7 + k], limited=1,
                    range=(0.0, 0.04 * s), armature=0.1, damping=1.0, ctrl_mode=CTRL_POSITION, kp=FRANKA_KP[7 + k],
                    kv=FRANKA_KV[7 + k], frc_range=(-frc[7 + k], frc[7 + k]))
    for body, half, centre in _PANDA_BOXES:
        if link_shape == "capsule" and body != "hand":
            (r, hl), quat = _capsule_of_box(half)
            sb.add_geom(body, GEOM_CAPSULE, size=(r * s, hl * s, 0.0), pos=sc(centre), quat=quat, rgb=white)
        else:
            sb.add_geom(body, GEOM_BOX, size=sc(half), pos=sc(centre), rgb=dark if body == "hand" else white)
    for finger in ("left_finger", "right_finger"):
        sb.add_geom(finger, GEOM_BOX, size=sc(_FINGER_BODY_BOX[0]), pos=sc(_FINGER_BODY_BOX[1]), rgb=dark)
        sb.add_geom(finger, GEOM_BOX, size=sc(_FINGER_PAD_BOX[0]), pos=sc(_FINGER_PAD_BOX[1]), rgb=dark)


def _add_cube(sb: SceneBuilder, name: str, pos, size=0.04, rho=200.0, friction=1.0, rgb=(0.85, 0.2, 0.15)) -> None:
    """gs.morphs.Box(size=size^3) as a free body; Genesis rigid material default density 200 kg/m^3."""
    h = size / 2
    mass = rho * size ** 3
    sb.add_body(name, 0, pos=pos, quat=(1, 0, 0, 0), jtype=JNT_FREE, mass=mass, inertia=box_inertia(mass, (h, h, h)))
    sb.add_geom(name, GEOM_BOX, size=(h, h, h), friction=friction, rgb=rgb)


def franka_cube_pick_scene(cube_size=0.04, cube_pos=(0.65, 0.0, 0.02), cube_rho=200.0, link_shape="capsule", frc=FRANKA_FRC_MJCF) -> SceneBuilder:
    sb = SceneBuilder()
    # ground plane (gs.morphs.Plane, cube_pick.py:50)
    sb.add_geom(0, GEOM_PLANE)
    # Panda (cube_pick.py:51)
    _add_franka(sb, link_shape=link_shape, frc=frc)
    # cube (gs.morphs.Box, cube_pick.py:52-54)
    _add_cube(sb, "cube", cube_pos, size=cube_size, rho=cube_rho)
    # visual-only pedestal of the fixed base link (never collides: contype = conaffinity = 0); appended last so
    # the indices of the colliding geoms and the candidate-pair order are unchanged
    sb.add_geom("link0", GEOM_BOX, size=(0.09, 0.08, 0.07), pos=(-0.04, 0.0, 0.07), contype=0, conaffinity=0, rgb=(0.92, 0.92, 0.9))
    # task extraction (cube_pick.py:66-68,134,142)
    sb.task = dict(eef_body=sb.body_index("hand"), obj_body=sb.body_index("cube"),
                   grip_dof=(sb.dof_index("finger_joint1"), sb.dof_index("finger_joint2")), reward_z=0.1)
    return sb


# ------------------------------------------------------------------------------------------------
# SO-101 pick scene (BASELINE.json configs[3]; reference: tasks/so101/cube_pick.py + tasks/utils.py:428-590)
#
# The SO-101 MJCF lives in an un-vendored git submodule (/root/reference/.gitmodules:1-3, directory
# empty), so the arm is RE-STATED from the public SO-ARM100/SO-101 description (UPSTREAM-RECALL,
# unverified; parity unpinned either way): six revolute joints -- shoulder_pan (z), shoulder_lift,
# elbow_flex, wrist_flex (y), wrist_roll (x), gripper jaw (z) -- link offsets of roughly
# 0.062 / 0.054 / 0.113 / 0.135 / 0.061 / 0.098 m, link masses 0.08-0.15 kg, STS3215 servo joint
# defaults (armature 0.028, damping 0.6).  Collision meshes are replaced by boxes.
# What IS taken from the reference: the x4 scale and mount pose (utils.py:559-568), the island slab
# top height 0.7000313 (utils.py:571-578) with its +-0.915 x +-0.401 m extent,
# the cube size / spawn height top+0.02+0.001 (utils.py:581-586), friction 5 on robot and cube
# (so101/cube_pick.py:38-39), kp 1000 / kv 200 on the five arm dofs (:41-42), dt 0.01.
SO101_JOINTS = ("joint1", "joint2", "joint3", "joint4", "joint5", "joint6")  # so101/cube_pick.py:7-14
ISLAND_TOP_Z = 0.7000312834978104  # literal at examples/franka/stack_cube_one_image.py:38
SO101_SCALE = 4.0

# name, parent, pos in parent (unscaled), axis, range, mass (unscaled), box half extents, box centre
_SO101_LINKS = (
    ("shoulder", "so101_base", (0.0388, 0.0, 0.0624), (0, 0, 1), (-1.92, 1.92), 0.100, (0.025, 0.025, 0.027), (0.0, 0.0, 0.027)),
    ("upper_arm", "shoulder", (0.0, 0.0, 0.0542), (0, 1, 0), (-1.745, 1.745), 0.103, (0.018, 0.018, 0.0563), (0.0, 0.0, 0.0563)),
    ("lower_arm", "upper_arm", (0.0, 0.0, 0.1126), (0, 1, 0), (-1.69, 1.69), 0.104, (0.0675, 0.016, 0.016), (0.0675, 0.0, 0.0)),
    ("wrist", "lower_arm", (0.1349, 0.0, 0.0), (0, 1, 0), (-1.658, 1.658), 0.079, (0.0305, 0.015, 0.015), (0.0305, 0.0, 0.0)),
    ("gripper", "wrist", (0.0611, 0.0, 0.0), (1, 0, 0), (-2.74, 2.84), 0.087, (0.025, 0.02, 0.012), (0.025, 0.0, 0.0)),
    ("jaw", "gripper", (0.03, 0.012, 0.0), (0, 0, 1), (-0.17, 1.745), 0.012, (0.034, 0.004, 0.01), (0.036, 0.006, 0.0)),
)


def so101_cube_pick_scene(cube_size=0.04, cube_pos=(-0.3, 0.0, ISLAND_TOP_Z + 0.021), cube_rho=200.0) -> SceneBuilder:
    sb = SceneBuilder()
    s = SO101_SCALE
    fr = 5.0
    sb.add_geom(0, GEOM_PLANE)                                                        # kitchen floor, z = 0
    sb.add_geom(0, GEOM_BOX, size=(0.915, 0.401, 0.05), pos=(0.0, 0.0, ISLAND_TOP_Z - 0.05), rgb=(0.75, 0.72, 0.68))  # island slab (static)
    sb.add_body("so101_base", 0, pos=(-0.5, 0.0, 0.7), mass=0.147 * s ** 3, ipos=(0.0, 0.0, 0.03 * s),
                inertia=box_inertia(0.147 * s ** 3, (0.04 * s, 0.04 * s, 0.03 * s)))
    sb.add_geom("so101_base", GEOM_BOX, size=(0.04 * s, 0.04 * s, 0.03 * s), pos=(0.0, 0.0, 0.03 * s + 0.002), friction=fr)
    for i, (name, parent, pos, axis, rng, mass, half, centre) in enumerate(_SO101_LINKS):
        m = mass * s ** 3
        hs = tuple(h * s for h in half)
        cs = tuple(c * s for c in centre)
        arm = i < 5
        sb.add_body(name, parent, pos=tuple(p * s for p in pos), jtype=JNT_REVOLUTE, axis=axis, mass=m, ipos=cs,
                    inertia=box_inertia(m, hs), joint_name=SO101_JOINTS[i], limited=1, range=rng, armature=0.028, damping=0.6,
                    ctrl_mode=CTRL_POSITION, kp=1000.0 if arm else 100.0, kv=200.0 if arm else 10.0, frc_range=(-1e30, 1e30))
        sb.add_geom(name, GEOM_BOX, size=hs, pos=cs, friction=fr)
    # fixed finger of the gripper, opposite the moving jaw
    sb.add_geom("gripper", GEOM_BOX, size=(0.034 * s, 0.004 * s, 0.01 * s), pos=(0.066 * s, -0.018 * s, 0.0), friction=fr)
    h = cube_size / 2
    mass = cube_rho * cube_size ** 3
    sb.add_body("cube", 0, pos=cube_pos, quat=(1, 0, 0, 0), jtype=JNT_FREE, mass=mass, inertia=box_inertia(mass, (h, h, h)))
    sb.add_geom("cube", GEOM_BOX, size=(h, h, h), friction=fr, rgb=(0.85, 0.2, 0.15))
    # eef link "gripper" (so101/cube_pick.py:37), gripper dof = joint6 (:36,:120), reward z > 0.1 (:112)
    sb.task = dict(eef_body=sb.body_index("gripper"), obj_body=sb.body_index("cube"), grip_dof=(sb.dof_index("joint6"),), reward_z=0.1)
    return sb



# ------------------------------------------------------------------------------------------------
# Stack scenes (gym_genesis/CubeStack-v0): arm + FIVE free cubes on the kitchen-island slab
# (/root/reference/gym_genesis/tasks/utils.py:239-426 build_house, :593-794 build_house_task_cube_stack).
# Only what carries physics on the hot path is restated: floor plane, island slab (top z = ISLAND_TOP_Z, the
# decomposed island mesh replaced by its top slab), the robot, cube_1 (red, picked), cube_2 (green, target) and three
# distractor cubes (the reference places them with the unseeded global np.random at build time, utils.py:411-423 /
# :777-789; every reset() re-draws them, so fixed build positions are used here).  Walls, fridge, ceiling lamp and
# the stove are visual-only in the reference (collision=False).
STACK_CUBES = ("cube_1", "cube_2", "distractor_1", "distractor_2", "distractor_3")
_STACK_CUBE_XY = ((0.1, 0.0), (-0.1, 0.05), (0.2, -0.15), (-0.2, -0.2), (0.05, 0.2))   # utils.py:395,404 for the first two
_STACK_CUBE_RGB = ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.2, 0.4, 0.9), (0.9, 0.8, 0.2), (0.7, 0.3, 0.8))
STACK_CUBE_Z = ISLAND_TOP_Z + 0.02 + 0.001  # utils.py:388-395: island_top_z + 0.02 + z_offset


def _stack_common(sb: SceneBuilder, friction=1.0) -> None:
    for name, (x, y), rgb in zip(STACK_CUBES, _STACK_CUBE_XY, _STACK_CUBE_RGB):
        _add_cube(sb, name, (x, y, STACK_CUBE_Z), friction=friction, rgb=rgb)
    sb.opt["max_contacts"] = MIR_MAX_CONTACT  # five resting cubes alone are 20 contact points


def franka_cube_stack_scene(link_shape="capsule") -> SceneBuilder:
    """build_house (utils.py:239-426): Panda MJCF at (-0.5, 0, 0.7), scale 0.6 (:370-377); PD gains / force ranges as
    reset() sets them (cube_stack_kitchen_batch.py:101-106) -- the same arrays the pick scene uses.  The SAME Panda as in the pick
    scene: links 1-7 are capsules (`link_shape="box"` restores the box links the stack scenes had before the wave kernel learnt
    the round geoms)."""
    sb = SceneBuilder()
    sb.add_geom(0, GEOM_PLANE)                                                        # kitchen floor, z = 0
    sb.add_geom(0, GEOM_BOX, size=(0.915, 0.401, 0.05), pos=(0.0, 0.0, ISLAND_TOP_Z - 0.05), rgb=(0.75, 0.72, 0.68))
    _add_franka(sb, pos=(-0.5, 0.0, 0.7), scale=0.6, frc=FRANKA_FRC, link_shape=link_shape)  # set explicitly by the task
    _stack_common(sb)
    sb.task = dict(eef_body=sb.body_index("hand"), obj_body=sb.body_index("cube_1"), obj2_body=sb.body_index("cube_2"),
                   grip_dof=(sb.dof_index("finger_joint1"), sb.dof_index("finger_joint2")), reward_z=0.1,
                   reward_mode=REWARD_STACK, reward_xy=0.05, reward_dz=0.03)          # cube_stack_kitchen_batch.py:138-146
    return sb


# SO-101 stack variant: so101_old_calib.xml scaled 1.3, yaw 90 deg, at (-0.5, 0, 0.7) (utils.py:730-744).  The MJCF is in
# an empty submodule, so the chain of the pick scene is reused; what the "old calibration" changes is where the joint
# zeros are, and the reference only tells us that deg (0, -177, 165, 72, -83, 0) is its home pose
# (cube_stack_batch.py:106).  The zero offsets are therefore CHOSEN so that this home pose is the folded rest pose
# SO101_STACK_REST of the re-stated chain (arm tucked above its base, clear of the slab); joint ranges move with them.
SO101_STACK_HOME_DEG = (0.0, -177.0, 165.0, 72.0, -83.0, 0.0)
SO101_STACK_REST = (0.0, -0.9, 1.3, 0.3, 0.0, 0.0)   # same pose in the pick chain's own joint angles (clear of the slab: tools scan)


def so101_cube_stack_scene() -> SceneBuilder:
    sb = SceneBuilder()
    s = 1.3
    sb.add_geom(0, GEOM_PLANE)
    sb.add_geom(0, GEOM_BOX, size=(0.915, 0.401, 0.05), pos=(0.0, 0.0, ISLAND_TOP_Z - 0.05), rgb=(0.75, 0.72, 0.68))
    yaw = math.radians(90.0)
    sb.add_body("so101_base", 0, pos=(-0.5, 0.0, 0.7), quat=(math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)), mass=0.147 * s ** 3,
                ipos=(0.0, 0.0, 0.03 * s), inertia=box_inertia(0.147 * s ** 3, (0.04 * s, 0.04 * s, 0.03 * s)))
    sb.add_geom("so101_base", GEOM_BOX, size=(0.04 * s, 0.04 * s, 0.03 * s), pos=(0.0, 0.0, 0.03 * s + 0.002))
    for i, (name, parent, pos, axis, rng, mass, half, centre) in enumerate(_SO101_LINKS):
        m = mass * s ** 3
        hs = tuple(h * s for h in half)
        cs = tuple(c * s for c in centre)
        arm = i < 5
        # joint zero offset: model angle = q + q0, baked into the body frame as a rotation about the joint axis
        q0 = SO101_STACK_REST[i] - math.radians(SO101_STACK_HOME_DEG[i])
        quat = (math.cos(q0 / 2),) + tuple(a * math.sin(q0 / 2) for a in axis)
        sb.add_body(name, parent, pos=tuple(p * s for p in pos), quat=quat, jtype=JNT_REVOLUTE, axis=axis, mass=m, ipos=cs,
                    inertia=box_inertia(m, hs), joint_name=SO101_JOINTS[i], limited=1, range=(rng[0] - q0, rng[1] - q0), armature=0.028,
                    damping=0.6, ctrl_mode=CTRL_POSITION, kp=1000.0 if arm else 100.0, kv=200.0 if arm else 10.0, frc_range=(-1e30, 1e30))
        sb.add_geom(name, GEOM_BOX, size=hs, pos=cs)
    sb.add_geom("gripper", GEOM_BOX, size=(0.034 * s, 0.004 * s, 0.01 * s), pos=(0.066 * s, -0.018 * s, 0.0))  # fixed finger
    _stack_common(sb)
    # eef = link "gripper", agent_pos = so_101.get_qpos() (cube_stack_batch.py:37,169), stack reward (:143-153)
    sb.task = dict(eef_body=sb.body_index("gripper"), obj_body=sb.body_index("cube_1"), obj2_body=sb.body_index("cube_2"), grip_dof=(),
                   reward_z=0.1, reward_mode=REWARD_STACK, reward_xy=0.05, reward_dz=0.03, agent_mode=AGENT_QPOS)
    return sb
