"""GPU parity of the convex narrowphase (sphere / capsule geoms: GJK on the cores, MPR when the cores overlap, closed-form plane
cases) against the float64 oracle, through the C ABI.

A batch IS a set of poses: two free bodies per env, every env in its own random relative pose, one forward-dynamics evaluation
(mir_forward): contact counts must agree and the constrained accelerations -- which see every contact's position, normal and
depth -- must match.  Then free-running rollouts of bodies dropped on the plane and on a static box.
"""
import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models
from gym_genesis.backend import spec as S

pytestmark = pytest.mark.gpu

SHAPES = {
    "sphere": (S.GEOM_SPHERE, (0.04,)),
    "capsule": (S.GEOM_CAPSULE, (0.03, 0.07)),
    "box": (S.GEOM_BOX, (0.04, 0.03, 0.05)),
}


def _inertia(kind, mass):
    t, size = SHAPES[kind]
    if t == S.GEOM_SPHERE:
        return S.sphere_inertia(mass, size[0])
    if t == S.GEOM_CAPSULE:
        return S.capsule_inertia(mass, size[0], size[1])
    return S.box_inertia(mass, size)


def _scene(kind_a, kind_b):
    """plane + a static box + two free bodies."""
    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    sb.add_geom(0, S.GEOM_BOX, size=(0.15, 0.15, 0.05), pos=(0.0, 0.0, 0.05))
    for name, kind, x in (("a", kind_a, -0.3), ("b", kind_b, 0.3)):
        t, size = SHAPES[kind]
        sb.add_body(name, 0, pos=(x, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=_inertia(kind, 0.3))
        sb.add_geom(name, t, size=tuple(size) + (0.0,) * (3 - len(size)))
    sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    return sb.build()


def _mir(spec, B):
    from gym_genesis.backend.lib import MirScene

    return MirScene(spec, B)


def _rand_quat(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


@pytest.mark.parametrize("pair", [("sphere", "sphere"), ("sphere", "capsule"), ("capsule", "capsule"), ("capsule", "box"), ("sphere", "box")])
def test_pose_batch_contacts_match_oracle(pair):
    """512 random relative poses of the two bodies in mid-air (only their mutual contact is possible), from grazing to
    overlapping cores: contact count identical; for contacts that go through GJK on the cores (shallower than the radii) the
    constrained acceleration is within 1e-3 relative of the oracle's."""
    spec = _scene(*pair)
    B = 512
    rng = np.random.default_rng(11)
    q = np.zeros((B, 14), np.float32)
    q[:, 0:3] = rng.uniform(-0.02, 0.02, (B, 3)) + [0, 0, 1.0]
    d = rng.normal(size=(B, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q[:, 7:10] = q[:, 0:3] + d * rng.uniform(0.01, 0.16, (B, 1))
    q[:, 3:7], q[:, 10:14] = _rand_quat(rng, B), _rand_quat(rng, B)
    v = rng.uniform(-0.2, 0.2, (B, 12)).astype(np.float32)
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    assert sc.kernel == 16
    sc.set_state(qpos=q, qvel=v, warmstart=np.zeros((B, 12), np.float32))
    _, _, _, qacc = (t.cpu().numpy() for t in sc.forward())
    ncon = sc.get_diag()[0].cpu().numpy()
    nco = np.zeros(B, int)
    worst = 0.0
    deep = shallow = 0
    for e in range(B):
        o.write(orc.F_QPOS, q[e], e)
        o.write(orc.F_QVEL, v[e], e)
        o.forward(e)
        nco[e] = o.counts(e)[0]
        if nco[e] != ncon[e]:
            continue
        qo = o.read(orc.F_QACC, e)
        dist = o.read(orc.F_CDIST, e)
        if nco[e] and dist[0] > -1e-4:
            continue  # grazing: the force is discontinuous at zero depth
        radii = sum(SHAPES[k][1][0] for k in pair if k != "box")
        err = np.abs(qacc[e] - qo).max() / max(1.0, np.abs(qo).max())
        if nco[e] and -dist[0] > 0.9 * radii:
            # cores overlapping (or about to): MPR, whose depth and normal depend on where the portal refinement stops -- the two
            # precisions may stop one refinement apart, so only the contact itself is required to agree (count, above)
            deep += 1
        else:
            shallow += int(nco[e] > 0)
            worst = max(worst, err)
    mism = int((nco != ncon).sum())
    print(f"{pair}: {int((nco > 0).sum())} contacts in {B} poses ({shallow} through GJK, {deep} through MPR), {mism} count mismatches, qacc rel err {worst:.2e}")
    assert mism <= 2                      # a pose exactly at touching distance may flip between the precisions
    assert (nco > 0).sum() > 100 and shallow > 50
    assert worst < 1e-3


@pytest.mark.parametrize("pair", [("sphere", "capsule"), ("capsule", "box")])
def test_dropped_bodies_settle_like_the_oracle(pair):
    """The two bodies fall on the plane / on the static box from random poses: plane-sphere, plane-capsule (two end points),
    capsule-box and sphere-box through GJK, free-running for 120 steps; the final resting heights agree and every env's position
    error stays below 2e-3 (tumbling contacts amplify float32 rounding; medians are ~1e-5)."""
    spec = _scene(*pair)
    B = 64
    rng = np.random.default_rng(5)
    pos = np.zeros((B, 2, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.05, 0.05, (B, 3)) + [0.0, 0.0, 0.22]    # above the static box
    pos[:, 1] = rng.uniform(-0.05, 0.05, (B, 3)) + [0.5, 0.0, 0.12]    # above the bare plane
    quat = np.stack([_rand_quat(rng, B), _rand_quat(rng, B)], 1).astype(np.float32)
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    arm = np.zeros((B, 0), np.float32)
    sc.reset(pos, quat, arm)
    o.reset(pos, quat, arm)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(120):
        sc.step_fused(None, *bufs)
        o.step_batch(None)
        if t == 20:
            qh = sc.get_state()[0].cpu().numpy()
            early = np.abs(qh - o.state()[0]).max()
    qh = sc.get_state()[0].cpu().numpy()
    qo = o.state()[0]
    err = np.abs(qh - qo)[:, [0, 1, 2, 7, 8, 9]].max(1)
    ncon = sc.get_diag()[0].cpu().numpy()
    print(f"{pair}: after 21 steps L-inf {early:.2e}; after 120 steps position err median {np.median(err):.2e}, max {err.max():.2e}; contacts {ncon.min()}..{ncon.max()}")
    assert early < 1e-5
    assert np.median(err) < 1e-4 and err.max() < 5e-3
    assert (ncon >= 1).all()
    assert np.abs(qh[:, 2] - qo[:, 2]).max() < 2e-3 and np.abs(qh[:, 9] - qo[:, 9]).max() < 2e-3   # resting heights


def test_round_geoms_on_the_wave_kernel_settle_like_the_oracle():
    """The wave-per-env kernel (scenes beyond 15 dofs: the five-cube stack tasks) carries the same convex narrowphase since round 3:
    four free bodies (24 dofs) -- sphere, capsule, box, capsule -- dropped from random poses on the plane and on a static box, free
    running for 120 steps against the oracle.  Plane - sphere / capsule in closed form, capsule - box and sphere - box through GJK
    on the cores, MPR when a core dips into the box."""
    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    sb.add_geom(0, S.GEOM_BOX, size=(0.4, 0.15, 0.05), pos=(0.0, 0.0, 0.05))
    kinds = ("sphere", "capsule", "box", "capsule")
    for i, kind in enumerate(kinds):
        t, size = SHAPES[kind]
        sb.add_body(f"b{i}", 0, pos=(0.0, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=_inertia(kind, 0.3))
        sb.add_geom(f"b{i}", t, size=tuple(size) + (0.0,) * (3 - len(size)))
    sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    spec = sb.build()
    B = 64
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    assert sc.kernel == 64
    rng = np.random.default_rng(11)
    pos = np.zeros((B, 4, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.04, 0.04, (B, 3)) + [-0.25, 0.0, 0.22]   # sphere above the static box
    pos[:, 1] = rng.uniform(-0.04, 0.04, (B, 3)) + [0.0, 0.0, 0.24]     # capsule above the static box
    pos[:, 2] = rng.uniform(-0.04, 0.04, (B, 3)) + [0.25, 0.0, 0.22]    # box above the static box
    pos[:, 3] = rng.uniform(-0.04, 0.04, (B, 3)) + [0.0, 0.6, 0.12]     # capsule above the bare plane
    quat = np.stack([_rand_quat(rng, B) for _ in range(4)], 1).astype(np.float32)
    arm = np.zeros((B, 0), np.float32)
    sc.reset(pos, quat, arm)
    o.reset(pos, quat, arm)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(120):
        sc.step_fused(None, *bufs)
        o.step_batch(None)
        if t == 20:
            early = np.abs(sc.get_state()[0].cpu().numpy() - o.state()[0]).max()
    qh, qo = sc.get_state()[0].cpu().numpy(), o.state()[0]
    cols = [7 * k + c for k in range(4) for c in range(3)]
    err = np.abs(qh - qo)[:, cols].max(1)
    ncon = sc.get_diag()[0].cpu().numpy()
    print(f"wave kernel, round geoms: after 21 steps L-inf {early:.2e}; after 120 steps position err median {np.median(err):.2e}, max {err.max():.2e}; contacts {ncon.min()}..{ncon.max()}")
    assert early < 1e-5
    assert np.median(err) < 1e-4 and err.max() < 5e-3
    assert (ncon >= 3).all()
    assert np.abs(qh[:, [2, 9, 16, 23]] - qo[:, [2, 9, 16, 23]]).max() < 2e-3   # resting heights


@pytest.mark.parametrize("kernel", [16, 64])
def test_hull_contacts_match_the_oracle_pose_by_pose(kernel):
    """Vertex-hull geoms (MIR_GEOM_HULL: an 8-vertex cube and a 32-vertex ball) in the 16-lane kernel (pool of 40 vertices beside the
    model table) and in the wave kernel (pool of 96, brought into the contact arrays' space for the narrowphase; the same scenes are
    sent there by a contact capacity of 48), one forward evaluation at 512 random relative poses of a hull and a round body in mid-air (hull - sphere through GJK on the
    polytope and the point, MPR when the centre dips into the hull): contact counts as the oracle's, constrained accelerations of the
    shallow contacts within 1e-3.  (Against a ROUND partner the closest feature pair is unique; two polytopes resting face to face
    have a whole polygon of closest points, and which of them a single-point narrowphase reports is decided by rounding.)"""
    for kind_a, verts in (("cube", S.box_hull_vertices((0.04, 0.03, 0.05))), ("ball", S.icosphere_vertices(0.05, 1))):
        sb = S.SceneBuilder()
        sb.add_geom(0, S.GEOM_PLANE)
        sb.add_body("a", 0, pos=(-0.3, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=S.box_inertia(0.3, (0.04, 0.03, 0.05)))
        sb.add_geom("a", S.GEOM_HULL, vertices=verts)
        sb.add_body("b", 0, pos=(0.3, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=S.sphere_inertia(0.3, 0.04))
        sb.add_geom("b", S.GEOM_SPHERE, size=(0.04, 0.0, 0.0))
        sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
        if kernel == 64:
            sb.opt["max_contacts"] = 48
        spec = sb.build()
        B = 512
        rng = np.random.default_rng(11)
        q = np.zeros((B, 14), np.float32)
        q[:, 0:3] = rng.uniform(-0.02, 0.02, (B, 3)) + [0, 0, 1.0]
        d = rng.normal(size=(B, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        q[:, 7:10] = q[:, 0:3] + d * rng.uniform(0.02, 0.13, (B, 1))
        q[:, 3:7], q[:, 10:14] = _rand_quat(rng, B), _rand_quat(rng, B)
        v = rng.uniform(-0.2, 0.2, (B, 12)).astype(np.float32)
        sc, o = _mir(spec, B), orc.Oracle(spec, B)
        assert sc.kernel == kernel
        sc.set_state(qpos=q, qvel=v, warmstart=np.zeros((B, 12), np.float32))
        _, _, _, qacc = (t.cpu().numpy() for t in sc.forward())
        ncon = sc.get_diag()[0].cpu().numpy()
        nco = np.zeros(B, int)
        worst, shallow, deep = 0.0, 0, 0
        for e in range(B):
            o.write(orc.F_QPOS, q[e], e)
            o.write(orc.F_QVEL, v[e], e)
            o.forward(e)
            nco[e] = o.counts(e)[0]
            if nco[e] != ncon[e] or not nco[e]:
                continue
            dist = o.read(orc.F_CDIST, e)
            if dist[0] > -1e-4:
                continue
            if -dist[0] > 0.9 * 0.04:
                deep += 1
                continue
            qo = o.read(orc.F_QACC, e)
            worst = max(worst, np.abs(qacc[e] - qo).max() / max(1.0, np.abs(qo).max()))
            shallow += 1
        mism = int((nco != ncon).sum())
        print(f"[kernel {kernel}] hull {kind_a} vs sphere: {int((nco > 0).sum())} contacts in {B} poses ({shallow} through GJK, {deep} through MPR), {mism} count mismatches, qacc rel err {worst:.2e}")
        assert mism <= 2 and (nco > 0).sum() > 100 and shallow > 50
        assert worst < 1e-3


def _hull_scene(kind_a, kind_b, kernel=16):
    """plane + two free bodies: an 8-vertex cube (or the same cube as GEOM_BOX) and a 32-vertex ball."""
    sb = S.SceneBuilder()
    if kernel == 64:
        sb.opt["max_contacts"] = 48
    sb.add_geom(0, S.GEOM_PLANE)
    h = (0.03, 0.03, 0.03)
    sb.add_body("a", 0, pos=(-0.3, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=S.box_inertia(0.3, h))
    if kind_a == "hull":
        sb.add_geom("a", S.GEOM_HULL, vertices=S.box_hull_vertices(h))
    else:
        sb.add_geom("a", S.GEOM_BOX, size=h)
    sb.add_body("b", 0, pos=(0.3, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.3, inertia=S.sphere_inertia(0.3, 0.04))
    sb.add_geom("b", S.GEOM_HULL, vertices=S.icosphere_vertices(0.04, 1))
    sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    return sb.build()


@pytest.mark.parametrize("kernel", [16, 64])
def test_hull_bodies_settle_on_the_plane_like_the_oracle(kernel):
    """The cube given as its 8 corners and a 32-vertex ball dropped from random poses on the plane (plane - hull: the penetrating
    vertices, reduced to four like plane - box), 150 free-running steps against the oracle; and the cube given as its corners follows
    the GEOM_BOX cube of the same kernel bit for bit (the lane-private hull path and the 8-lane plane - box path do the same
    arithmetic)."""
    B = 64
    rng = np.random.default_rng(9)
    pos = np.zeros((B, 2, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.05, 0.05, (B, 3)) + [0.5, 0.0, 0.12]
    pos[:, 1] = rng.uniform(-0.05, 0.05, (B, 3)) + [0.0, 0.0, 0.12]
    quat = np.stack([_rand_quat(rng, B), _rand_quat(rng, B)], 1).astype(np.float32)
    arm = np.zeros((B, 0), np.float32)
    spec = _hull_scene("hull", "hull", kernel)
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    assert sc.kernel == kernel
    ref = _mir(_hull_scene("box", "hull", kernel), B)
    for x in (sc, o, ref):
        x.reset(pos, quat, arm)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(150):
        sc.step_fused(None, *bufs)
        ref.step_fused(None, *bufs)
        o.step_batch(None)
        if t == 20:
            early = np.abs(sc.get_state()[0].cpu().numpy() - o.state()[0]).max()
    qh, qo, qr = sc.get_state()[0].cpu().numpy(), o.state()[0], ref.get_state()[0].cpu().numpy()
    err = np.abs(qh - qo)[:, [0, 1, 2, 7, 8, 9]].max(1)
    ncon = sc.get_diag()[0].cpu().numpy()
    print(f"[kernel {kernel}] hull bodies on the plane: after 21 steps L-inf {early:.2e}; after 150 steps position err median {np.median(err):.2e}, max {err.max():.2e}; contacts {ncon.min()}..{ncon.max()}; "
          f"hull cube vs box cube {np.abs(qh[:, :7] - qr[:, :7]).max():.2e}")
    assert early < 1e-5
    assert np.median(err) < 1e-4 and err.max() < 5e-3
    assert (ncon >= 4).all()
    assert np.abs(qh[:, 2] - qo[:, 2]).max() < 2e-3 and np.abs(qh[:, 9] - qo[:, 9]).max() < 2e-3   # resting heights
    assert np.array_equal(qh[:, :7], qr[:, :7])                                                      # cube as corners == cube as box


def test_hull_vertex_pool_beyond_the_16_lane_capacity_runs_on_the_wave_kernel():
    """More hull vertices than the 16-lane kernel keeps in LDS (K16_MAX_VERT = 40): the scene goes to the wave kernel (96), which
    round 3 refused hulls on.  Two 32-vertex balls dropped on the plane and on each other, against the oracle; beyond 96 vertices
    the builder itself refuses."""
    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    for i in range(2):
        sb.add_body(f"b{i}", 0, pos=(0.2 * i, 0, 0.1), jtype=S.JNT_FREE, mass=0.1, inertia=S.sphere_inertia(0.1, 0.04))
        sb.add_geom(f"b{i}", S.GEOM_HULL, vertices=S.icosphere_vertices(0.04, 1))
    sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    spec = sb.build()
    B = 32
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    assert sc.kernel == 64
    rng = np.random.default_rng(4)
    pos = np.zeros((B, 2, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.02, 0.02, (B, 3)) + [0.0, 0.0, 0.08]
    pos[:, 1] = pos[:, 0] + rng.uniform(-0.015, 0.015, (B, 3)) + [0.0, 0.0, 0.10]   # lands on the first ball, rolls off
    quat = np.stack([_rand_quat(rng, B), _rand_quat(rng, B)], 1).astype(np.float32)
    arm = np.zeros((B, 0), np.float32)
    sc.reset(pos, quat, arm); o.reset(pos, quat, arm)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(25):
        sc.step_fused(None, *bufs)
        o.step_batch(None)
        if t == 7:   # (free fall and the first ball's touch-down on the plane; polytope on polytope later: a polygon of closest points)
            early = np.abs(sc.get_state()[0].cpu().numpy() - o.state()[0]).max()
    late = np.abs(sc.get_state()[0].cpu().numpy() - o.state()[0]).max()
    print(f"two 32-vertex hulls on the wave kernel: L-inf after 8 steps {early:.2e}, after 25 steps {late:.2e}")
    assert early < 5e-6 and late < 1e-3 and np.isfinite(sc.get_state()[0].cpu().numpy()).all()
    with pytest.raises(ValueError):
        for i in range(2):
            sb.add_body(f"c{i}", 0, pos=(0.5, 0.2 * i, 0.1), jtype=S.JNT_FREE, mass=0.1, inertia=S.sphere_inertia(0.1, 0.04))
            sb.add_geom(f"c{i}", S.GEOM_HULL, vertices=S.icosphere_vertices(0.04, 1))


def test_hull_cubes_follow_box_cubes_bit_for_bit_in_the_franka_five_cube_scene():
    """The stack tasks' robot and five free cubes (39 dofs: the wave kernel) with the cubes given as their 8 corners
    (MIR_GEOM_HULL: the stand-in for the reference's mesh cubes, tasks/utils.py:372,561,732) and as GEOM_BOX, on the PLANE and
    apart from each other: plane - hull keeps the penetrating vertices exactly as plane - box keeps the penetrating corners, so
    the two scenes stay bit-identical over 150 steps of random arm targets.  (On the kitchen SLAB -- a box -- a hull cube would
    rest on the one closest point GJK reports where the box cube gets a clipped face patch: not a parity case)"""
    def scene(hull):
        sb = models.franka_cube_stack_scene()
        spec0 = sb.build()
        # the five cubes' geoms: re-add them as hulls (same size, same body), everything else as it is
        cube_geoms = [g for g in sb.geoms if sb.bodies[g["body"]]["name"] in models.STACK_CUBES]
        assert len(cube_geoms) == 5 and all(g["type"] == S.GEOM_BOX for g in cube_geoms)
        if hull:
            for g in cube_geoms:
                v0 = len(sb.verts)
                verts = S.box_hull_vertices(tuple(g["size"][:3]))
                sb.verts.extend(tuple(v) for v in verts)
                g["type"] = S.GEOM_HULL
                g["size"] = (float(v0), float(len(verts)), 0.0)
        # drop the slab: the cubes rest on the floor plane
        sb.geoms = [g for g in sb.geoms if not (g["type"] == S.GEOM_BOX and g["body"] == 0)]
        return sb.build(), spec0

    (spec_h, _), (spec_b, _) = scene(True), scene(False)
    B = 64
    sc, ref = _mir(spec_h, B), _mir(spec_b, B)
    assert sc.kernel == 64 and ref.kernel == 64
    rng = np.random.default_rng(2)
    pos = np.zeros((B, 5, 3), np.float32)
    for k in range(5):
        pos[:, k] = rng.uniform(-0.02, 0.02, (B, 3)) + [0.45 + 0.09 * k, -0.3 + 0.15 * k, 0.06]
    quat = np.stack([_rand_quat(rng, B) for _ in range(5)], 1).astype(np.float32)
    arm = np.tile(np.asarray(models.FRANKA_HOME, np.float32), (B, 1))
    for x in (sc, ref):
        x.reset(pos, quat, arm)
    b1 = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(ref.agent_dim), ref.empty(ref.env_dim), ref.empty(), ref.empty(dtype=torch.uint8))
    acts = torch.as_tensor(arm[None] + rng.uniform(-0.3, 0.3, (150, B, 9)).astype(np.float32), device=sc.device)
    for t in range(150):
        sc.step_fused(acts[t], *b1)
        ref.step_fused(acts[t], *b2)
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)
    for x, y in zip(b1, b2):
        assert torch.equal(x, y)
    ncon = sc.get_diag()[0].cpu().numpy()
    assert (ncon >= 15).all()    # five cubes at rest on the plane: 20 points, a tilted one in transit fewer


def _device_pairs(rows):
    """Run the kernel-side convex_pair on (n, 22) float32 rows; returns (n, 8)."""
    import ctypes as C

    from gym_genesis.backend import lib

    L = lib.load_library()
    L.mir_debug_convex_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_void_p]
    L.mir_debug_convex_pairs.restype = C.c_int
    t_in = torch.as_tensor(np.ascontiguousarray(rows, np.float32), device="cuda")
    t_out = torch.zeros((rows.shape[0], 8), dtype=torch.float32, device="cuda")
    assert L.mir_debug_convex_pairs(t_in.data_ptr(), t_out.data_ptr(), rows.shape[0], torch.cuda.current_device(),
                                    torch.cuda.current_stream().cuda_stream) == 0
    return t_out.cpu().numpy()


def test_kernel_narrowphase_equals_oracle_pair_by_pair():
    """4000 random pairs of {box, sphere, capsule} (not box-box): the kernel's convex_pair against the oracle's, pair by pair, and
    against the host build of the same source (tests/test_convex_host.py), with which it must agree to float32 rounding."""
    import ctypes as C

    import test_convex_host as H

    rows = H.random_pairs(4000, 21)
    got = _device_pairs(rows)
    shallow, deep = H.compare(rows, got, H.oracle_pairs(rows))
    print(f"pair by pair vs oracle: {shallow} shallow (GJK), {deep} deep (MPR) contacts agree")
    assert shallow > 300 and deep > 100
    host = np.zeros_like(got)
    H.host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), host.ctypes.data_as(C.c_void_p), rows.shape[0])
    same = (got[:, 0] == host[:, 0])
    assert same.mean() > 0.995
    both = same & (got[:, 0] == 1)
    assert np.median(np.abs(got[both, 4] - host[both, 4])) < 1e-7


def test_kernel_mpr_on_analytic_cases():
    """The deep pairs with a closed-form answer (tests/test_convex_host.py: analytic_deep_pairs -- a sphere / capsule whose core lies
    inside a box on one of its symmetry planes): the device build gives the analytic depth within 5 % and the normal within 2e-2 rad
    (before round 3 these came back as a depth-r contact with an unnormalised normal: GJK left through its repeated-vertex exit
    with the origin ON the simplex and reported the cores as apart)."""
    import test_convex_host as H

    rows, want = H.analytic_deep_pairs()
    got = _device_pairs(rows)
    for i, (depth, n) in enumerate(want):
        assert got[i, 0] == 1, (i, rows[i], got[i])
        assert abs(-got[i, 4] - depth) < 0.05 * depth, (i, -got[i, 4], depth)
        assert np.arccos(np.clip(np.dot(got[i, 5:8], n), -1, 1)) < 2e-2, (i, got[i, 5:8], n)


def test_franka_pick_with_capsule_links_matches_oracle():
    """CubePick-v0 with links 1-7 as capsules (the default; GenesisEnv(..., link_shape="box") gives round 1's boxes): the
    benchmark's random-action workload at
    256 envs, 60 free-running steps against the oracle (link-cube and link-plane pairs go through GJK / the closed-form plane
    cases when they come close), then the scripted grasp still lifts the cube."""
    import json
    import os

    from gym_genesis.backend import models
    from gym_genesis.env import GenesisEnv

    B = 256
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, link_shape="capsule")
    task = env._env
    assert task._mir.kernel == 16
    spec = task._mir.spec
    assert sum(1 for g in range(spec.ngeom) if spec.geom[g].type == S.GEOM_CAPSULE) == 7
    env.reset(seed=0)
    o = orc.Oracle(spec, B)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    home = np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1))
    o.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), home)
    o.step_batch(None)
    acts = np.random.default_rng(8).uniform(-1, 1, (60, B, 9)).astype(np.float32)
    for t in range(60):
        obs, reward, terminated, truncated, info = env.step(acts[t])
        o.step_batch(acts[t])
    err = np.abs(task._mir.get_state()[0].cpu().numpy() - o.state()[0]).max(1)
    print(f"capsule-link Franka, 60 random steps: qpos err median {np.median(err):.2e}, 99 % {np.quantile(err, 0.99):.2e}, max {err.max():.2e}")
    assert np.quantile(err, 0.99) < 1e-4
    # scripted grasp (the fixture's joint targets were solved for the hand, which is unchanged)
    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    T = np.array(G_["targets"], np.float32)
    n = T.shape[0]
    env2 = GenesisEnv(task="cube_pick", robot="franka", num_envs=n, enable_pixels=False, link_shape="capsule")
    t2 = env2._env
    cube = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    t2._mir.reset(cube, np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)), home[:n])
    lifted = np.zeros(n, bool)
    for s in range(T.shape[1]):
        for _ in range(G_["steps_per_stage"]):
            _, reward, term, _, _ = env2.step(T[:, s])
            lifted |= term
    assert lifted.all()
