"""ctypes mirror of include/mirigid.h (MirSceneSpec and friends) plus a small
builder API used to describe a scene the way the reference's tasks do with
``scene.add_entity`` (/root/reference/gym_genesis/tasks/franka/cube_pick.py:50-54).

Pure host-side data: no physics here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

MIR_VERSION = 3
MIR_MAX_BODY = 32
MIR_MAX_DOF = 48
MIR_MAX_Q = 56
MIR_MAX_GEOM = 40
MIR_MAX_PAIR = 256
MIR_MAX_CONTACT = 48
MIR_MAX_GRIP = 4
MIR_MAX_FREE = 8
# limits of the 16-lanes-per-env kernel (the pick tasks); larger scenes run on the wave-per-env kernel
K16_MAX_CONTACT = 16
REWARD_LIFT, REWARD_STACK = 0, 1
AGENT_EEF, AGENT_QPOS = 0, 1

JNT_FIXED, JNT_REVOLUTE, JNT_PRISMATIC, JNT_FREE = 0, 1, 2, 3
GEOM_PLANE, GEOM_BOX, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3  # sphere: size = (radius,); capsule: (radius, half length), axis z
GEOM_HULL = 4  # convex hull of the vertices given to add_geom(vertices=...), geom frame, origin inside the hull
MIR_MAX_HULL_VERT, MIR_MAX_VERT = 32, 96
CTRL_NONE, CTRL_POSITION = 0, 1

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


class MirBodySpec(C.Structure):
    _fields_ = [
        ("parent", C.c_int32),
        ("jtype", C.c_int32),
        ("pos", C.c_double * 3),
        ("quat", C.c_double * 4),
        ("axis", C.c_double * 3),
        ("mass", C.c_double),
        ("ipos", C.c_double * 3),
        ("inertia", C.c_double * 6),
    ]


class MirDofSpec(C.Structure):
    _fields_ = [
        ("limited", C.c_int32),
        ("ctrl_mode", C.c_int32),
        ("range", C.c_double * 2),
        ("armature", C.c_double),
        ("damping", C.c_double),
        ("kp", C.c_double),
        ("kv", C.c_double),
        ("frc_range", C.c_double * 2),
        ("solref", C.c_double * 2),
        ("solimp", C.c_double * 5),
    ]


class MirGeomSpec(C.Structure):
    _fields_ = [
        ("body", C.c_int32),
        ("type", C.c_int32),
        ("contype", C.c_int32),
        ("conaffinity", C.c_int32),
        ("size", C.c_double * 3),
        ("pos", C.c_double * 3),
        ("quat", C.c_double * 4),
        ("friction", C.c_double),
        ("solref", C.c_double * 2),
        ("solimp", C.c_double * 5),
    ]


class MirOptions(C.Structure):
    _fields_ = [
        ("dt", C.c_double),
        ("gravity", C.c_double * 3),
        ("tolerance", C.c_double),
        ("ls_tolerance", C.c_double),
        ("iterations", C.c_int32),
        ("ls_iterations", C.c_int32),
        ("enable_collision", C.c_int32),
        ("enable_joint_limit", C.c_int32),
        ("enable_self_collision", C.c_int32),
        ("enable_adjacent_collision", C.c_int32),
        ("max_contacts", C.c_int32),
        ("implicit_damping", C.c_int32),
    ]


class MirTaskSpec(C.Structure):
    _fields_ = [
        ("eef_body", C.c_int32),
        ("obj_body", C.c_int32),
        ("n_grip", C.c_int32),
        ("grip_dof", C.c_int32 * MIR_MAX_GRIP),
        ("reward_z", C.c_double),
        ("obj2_body", C.c_int32),
        ("reward_mode", C.c_int32),
        ("agent_mode", C.c_int32),
        ("_pad", C.c_int32),
        ("reward_xy", C.c_double),
        ("reward_dz", C.c_double),
    ]


class MirSceneSpec(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("version", C.c_int32),
        ("nbody", C.c_int32),
        ("ndof", C.c_int32),
        ("ngeom", C.c_int32),
        ("_pad", C.c_int32),
        ("opt", MirOptions),
        ("task", MirTaskSpec),
        ("body", MirBodySpec * MIR_MAX_BODY),
        ("dof", MirDofSpec * MIR_MAX_DOF),
        ("geom", MirGeomSpec * MIR_MAX_GEOM),
        ("nvert", C.c_int32),
        ("_pad2", C.c_int32),
        ("vert", (C.c_double * 3) * MIR_MAX_VERT),
    ]


class MirDims(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32),
        ("nbody", C.c_int32),
        ("nq", C.c_int32),
        ("nv", C.c_int32),
        ("ngeom", C.c_int32),
        ("npair", C.c_int32),
        ("agent_dim", C.c_int32),
        ("env_dim", C.c_int32),
        ("nfree", C.c_int32),
        ("kernel", C.c_int32),
    ]


class MirCameraSpec(C.Structure):
    """scene.add_camera(res=(W,H), pos, lookat, fov) (reference cube_pick.py:56-63)."""
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("pos", C.c_double * 3),
        ("lookat", C.c_double * 3),
        ("up", C.c_double * 3),
        ("fov_deg", C.c_double),
    ]


class MirVisualSpec(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("_pad", C.c_int32),
        ("geom_rgb", (C.c_double * 3) * MIR_MAX_GEOM),
        ("light_dir", C.c_double * 3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("sky_rgb", C.c_double * 3),
        ("checker_rgb", (C.c_double * 3) * 2),
        ("checker_size", C.c_double),
    ]


class MirIkOptions(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32),
        ("respect_joint_limit", C.c_int32),
        ("damping", C.c_double),
        ("pos_tol", C.c_double),
        ("rot_tol", C.c_double),
        ("max_step", C.c_double),
    ]


class MirIkRows(C.Structure):
    """include/mirigid.h: MirIkRows (mir_inverse_kinematics_rows)"""
    _fields_ = [
        ("env_idx", C.c_void_p),
        ("n_rows", C.c_int32),
        ("flags", C.c_uint32),
        ("init_col0", C.c_int32),
        ("init_ncols", C.c_int32),
    ]


IK_POS_BY_ENV, IK_QUAT_BY_ENV, IK_QUAT_ONE, IK_INIT_BY_ENV = 1, 2, 4, 8

IK_DEFAULTS = dict(max_iters=20, respect_joint_limit=1, damping=0.05, pos_tol=5e-4, rot_tol=5e-3, max_step=0.5)

RENDER_PER_ENV, RENDER_GLOBAL = 0, 1


def make_camera(width, height, pos, lookat, fov, up=(0.0, 0.0, 1.0)) -> MirCameraSpec:
    c = MirCameraSpec()
    c.width, c.height, c.fov_deg = int(width), int(height), float(fov)
    c.pos[:], c.lookat[:], c.up[:] = [float(v) for v in pos], [float(v) for v in lookat], [float(v) for v in up]
    return c


def quat_normalize(q: Sequence[float]) -> tuple:
    n = math.sqrt(sum(x * x for x in q))
    return tuple(x / n for x in q)


def box_hull_vertices(half: Sequence[float]) -> list:
    """The 8 corners of a box as hull vertices, in the corner order of the plane - box narrowphase (bit 0 = +x, bit 1 = +y,
    bit 2 = +z): a GEOM_HULL of these reproduces the GEOM_BOX contact points."""
    return [((1 if c & 1 else -1) * half[0], (1 if c & 2 else -1) * half[1], (1 if c & 4 else -1) * half[2]) for c in range(8)]


def icosphere_vertices(radius: float, level: int = 0) -> list:
    """Vertices on the sphere of `radius`: the icosahedron's 12 (level 0), or those plus its 20 face centres pushed out to the
    sphere (level 1: 32 vertices = MIR_MAX_HULL_VERT, the pentakis dodecahedron)."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    norm = lambda p: tuple(radius * c / (p[0] ** 2 + p[1] ** 2 + p[2] ** 2) ** 0.5 for c in p)  # noqa: E731
    out = [norm(p) for p in v]
    if level >= 1:
        out += [norm(tuple(v[a][i] + v[b][i] + v[c][i] for i in range(3))) for a, b, c in f]
    return out


def box_inertia(mass: float, half: Sequence[float]) -> tuple:
    """Solid-box inertia about its centre, (xx, yy, zz, xy, xz, yz)."""
    x, y, z = (2 * h for h in half)
    return (mass * (y * y + z * z) / 12, mass * (x * x + z * z) / 12, mass * (x * x + y * y) / 12, 0.0, 0.0, 0.0)


def sphere_inertia(mass: float, radius: float) -> tuple:
    i = 0.4 * mass * radius * radius
    return (i, i, i, 0.0, 0.0, 0.0)


def capsule_inertia(mass: float, radius: float, half: float) -> tuple:
    """Solid capsule (cylinder of length 2 half + two hemispherical caps) of total mass `mass`, axis z, about its centre."""
    r, h = radius, 2.0 * half
    vc, vs = math.pi * r * r * h, 4.0 / 3.0 * math.pi * r ** 3
    mc, ms = mass * vc / (vc + vs), mass * vs / (vc + vs)
    izz = 0.5 * mc * r * r + 0.4 * ms * r * r
    ixx = mc * (h * h / 12 + r * r / 4) + ms * (0.4 * r * r + h * h / 4 + 3.0 / 8.0 * h * r)
    return (ixx, ixx, izz, 0.0, 0.0, 0.0)


class SceneBuilder:
    """Accumulates bodies / dofs / geoms, then emits a MirSceneSpec.

    Body 0 is the world.  Names are kept host-side only (``get_link`` lookups).
    """

    def __init__(self):
        self.bodies: List[dict] = [dict(name="world", parent=0, jtype=JNT_FIXED, pos=(0, 0, 0), quat=(1, 0, 0, 0),
                                        axis=(0, 0, 1), mass=0.0, ipos=(0, 0, 0), inertia=(0,) * 6)]
        self.dofs: List[dict] = []
        self.geoms: List[dict] = []
        self.verts: List[tuple] = []   # vertex pool of the GEOM_HULL geoms
        self.dof_names: List[str] = []
        self.opt = dict(dt=0.01, gravity=(0.0, 0.0, -9.81), tolerance=1e-8, ls_tolerance=0.01, iterations=50,
                        ls_iterations=50, enable_collision=1, enable_joint_limit=1, enable_self_collision=0,
                        enable_adjacent_collision=0, max_contacts=K16_MAX_CONTACT, implicit_damping=1)
        self.task = dict(eef_body=0, obj_body=0, grip_dof=(), reward_z=0.1)

    # -- bodies -----------------------------------------------------------------------------
    def add_body(self, name: str, parent: str | int, pos=(0, 0, 0), quat=(1, 0, 0, 0), jtype=JNT_FIXED,
                 axis=(0, 0, 1), mass=0.0, ipos=(0, 0, 0), inertia=(0,) * 6, joint_name: Optional[str] = None,
                 **dof_kw) -> int:
        pidx = parent if isinstance(parent, int) else self.body_index(parent)
        idx = len(self.bodies)
        self.bodies.append(dict(name=name, parent=pidx, jtype=jtype, pos=tuple(pos), quat=quat_normalize(quat),
                                axis=tuple(axis), mass=float(mass), ipos=tuple(ipos), inertia=tuple(inertia)))
        ndof = {JNT_FIXED: 0, JNT_REVOLUTE: 1, JNT_PRISMATIC: 1, JNT_FREE: 6}[jtype]
        for k in range(ndof):
            d = dict(limited=0, ctrl_mode=CTRL_NONE, range=(0.0, 0.0), armature=0.0, damping=0.0, kp=0.0, kv=0.0,
                     frc_range=(-1e30, 1e30), solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP)
            if jtype != JNT_FREE:
                d.update(dof_kw)
            self.dofs.append(d)
            self.dof_names.append(joint_name or name if ndof == 1 else f"{name}/{k}")
        return idx

    def body_index(self, name: str) -> int:
        for i, b in enumerate(self.bodies):
            if b["name"] == name:
                return i
        raise KeyError(name)

    def dof_index(self, joint_name: str) -> int:
        return self.dof_names.index(joint_name)

    # -- geoms ------------------------------------------------------------------------------
    def add_geom(self, body: str | int, gtype: int, size=(0, 0, 0), pos=(0, 0, 0), quat=(1, 0, 0, 0),
                 friction=1.0, contype=1, conaffinity=1, solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP,
                 rgb=(0.8, 0.8, 0.8), vertices=None) -> int:
        bidx = body if isinstance(body, int) else self.body_index(body)
        if gtype == GEOM_HULL:
            verts = [tuple(float(c) for c in v) for v in vertices]
            if not 4 <= len(verts) <= MIR_MAX_HULL_VERT:
                raise ValueError("a hull geom takes 4 .. MIR_MAX_HULL_VERT vertices")
            if len(self.verts) + len(verts) > MIR_MAX_VERT:
                raise ValueError("scene exceeds MIR_MAX_VERT hull vertices")
            size = (float(len(self.verts)), float(len(verts)), 0.0)
            self.verts.extend(verts)
        self.geoms.append(dict(body=bidx, type=gtype, size=tuple(size), pos=tuple(pos), quat=quat_normalize(quat),
                               friction=float(friction), contype=contype, conaffinity=conaffinity,
                               solref=tuple(solref), solimp=tuple(solimp), rgb=tuple(rgb)))
        return len(self.geoms) - 1

    def visual(self, light_dir=(0.3, -0.4, 0.85), ambient=0.35, diffuse=0.65, sky_rgb=(0.55, 0.7, 0.9),
               checker_rgb=((0.82, 0.82, 0.82), (0.42, 0.42, 0.45)), checker_size=0.5) -> "MirVisualSpec":
        """Appearance used by mir_render: per-geom albedo from add_geom(rgb=...), one directional light."""
        v = MirVisualSpec()
        v.struct_size = C.sizeof(MirVisualSpec)
        for i, g in enumerate(self.geoms):
            v.geom_rgb[i][:] = g["rgb"]
        v.light_dir[:] = light_dir
        v.ambient, v.diffuse, v.checker_size = ambient, diffuse, checker_size
        v.sky_rgb[:] = sky_rgb
        v.checker_rgb[0][:], v.checker_rgb[1][:] = checker_rgb[0], checker_rgb[1]
        return v

    # -- emit -------------------------------------------------------------------------------
    def build(self) -> MirSceneSpec:
        if len(self.bodies) > MIR_MAX_BODY or len(self.dofs) > MIR_MAX_DOF or len(self.geoms) > MIR_MAX_GEOM:
            raise ValueError("scene exceeds MIR_MAX_* capacity")
        s = MirSceneSpec()
        s.struct_size = C.sizeof(MirSceneSpec)
        s.version = MIR_VERSION
        s.nbody, s.ndof, s.ngeom = len(self.bodies), len(self.dofs), len(self.geoms)
        o = self.opt
        s.opt.dt = o["dt"]
        s.opt.gravity[:] = o["gravity"]
        for k in ("tolerance", "ls_tolerance", "iterations", "ls_iterations", "enable_collision", "enable_joint_limit",
                  "enable_self_collision", "enable_adjacent_collision", "max_contacts", "implicit_damping"):
            setattr(s.opt, k, o[k])
        t = self.task
        s.task.eef_body, s.task.obj_body = t["eef_body"], t["obj_body"]
        s.task.n_grip = len(t["grip_dof"])
        for i, g in enumerate(t["grip_dof"]):
            s.task.grip_dof[i] = g
        s.task.reward_z = t["reward_z"]
        s.task.obj2_body = t.get("obj2_body", -1)
        s.task.reward_mode = t.get("reward_mode", REWARD_LIFT)
        s.task.agent_mode = t.get("agent_mode", AGENT_EEF)
        s.task.reward_xy, s.task.reward_dz = t.get("reward_xy", 0.05), t.get("reward_dz", 0.03)
        for i, b in enumerate(self.bodies):
            sb = s.body[i]
            sb.parent, sb.jtype, sb.mass = b["parent"], b["jtype"], b["mass"]
            sb.pos[:], sb.quat[:], sb.axis[:] = b["pos"], b["quat"], b["axis"]
            sb.ipos[:], sb.inertia[:] = b["ipos"], b["inertia"]
        for i, d in enumerate(self.dofs):
            sd = s.dof[i]
            sd.limited, sd.ctrl_mode = d["limited"], d["ctrl_mode"]
            sd.range[:] = d["range"]
            sd.armature, sd.damping, sd.kp, sd.kv = d["armature"], d["damping"], d["kp"], d["kv"]
            sd.frc_range[:] = d["frc_range"]
            sd.solref[:], sd.solimp[:] = d["solref"], d["solimp"]
        for i, g in enumerate(self.geoms):
            sg = s.geom[i]
            sg.body, sg.type, sg.contype, sg.conaffinity = g["body"], g["type"], g["contype"], g["conaffinity"]
            sg.size[:], sg.pos[:], sg.quat[:] = g["size"], g["pos"], g["quat"]
            sg.friction = g["friction"]
            sg.solref[:], sg.solimp[:] = g["solref"], g["solimp"]
        s.nvert = len(self.verts)
        for i, v in enumerate(self.verts):
            s.vert[i][:] = v
        return s
