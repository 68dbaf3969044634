"""First-principles known-answer tests that pin the CPU oracle.

The reference has no tests or golden vectors for its physics (it delegates to an
external engine), so these closed-form / independent-method checks are what the
oracle is anchored on.  CPU only.
"""
import numpy as np
import pytest

import orc
from gym_genesis.backend import models, spec as S

HOME = np.array(models.FRANKA_HOME)
G = 9.81


def make(cube_pos=(0.65, 0.0, 0.02), **opt):
    sb = models.franka_cube_pick_scene(cube_pos=cube_pos)
    sb.opt.update(opt)
    return sb


def passive_arm(sb):
    """Strip actuation/damping so the arm is a conservative mechanism."""
    for d in sb.dofs:
        d.update(ctrl_mode=S.CTRL_NONE, kp=0.0, kv=0.0, damping=0.0, limited=0)
    sb.opt.update(implicit_damping=0, enable_collision=0, enable_joint_limit=0)
    return sb


def test_free_fall_matches_discrete_formula():
    sb = make(cube_pos=(0.65, 0.0, 5.0))
    o = orc.Oracle(sb.build())
    o.reset([0.65, 0, 5.0], [1, 0, 0, 0], HOME)
    # the oracle simulates the same float32-rounded dt / gravity the product stores
    dt, g = float(np.float32(0.01)), float(np.float32(G))
    for k in range(1, 51):
        o.step()
        q, v = o.state()
        assert abs(v[0, 11] - (-g * k * dt)) < 1e-12
        assert abs(q[0, 11] - (5.0 - g * dt * dt * k * (k + 1) / 2)) < 1e-12
        assert np.allclose(q[0, [9, 10]], [0.65, 0.0], atol=1e-14)
        assert np.allclose(q[0, 12:16], [1, 0, 0, 0], atol=1e-14)


def test_cube_rest_force_balance_and_penetration():
    sp = make().build()
    o = orc.Oracle(sp)
    o.reset([0.65, 0, 0.02], [0, 0, 0, 1], HOME)
    for _ in range(400):
        o.step()
    q, v = o.state()
    assert np.abs(v[0, 9:]).max() < 1e-9
    assert np.allclose(q[0, 9:11], [0.65, 0.0], atol=1e-9)  # no lateral drift
    ncon, nefc, _ = o.counts()
    assert ncon == 4
    o.forward()
    J = o.read(orc.F_J).reshape(-1, o.nv)
    f = o.read(orc.F_EFCFORCE)
    m = sp.body[sp.task.obj_body].mass
    qf = J.T @ f
    assert abs(qf[11] - m * G) < 1e-6 * m * G  # contact force carries the weight
    # closed-form penetration: 16 pyramid rows, each D * k * imp * depth
    depth = 0.02 - q[0, 11]
    x = depth / 0.001
    imp = 0.9 + 0.05 * (x / 0.5) ** 2 * 0.5
    w = 1.0 / m
    R = 2 * (1 - imp) / imp * w * 2
    k = 1 / (0.95 ** 2 * 0.02 ** 2)
    assert abs(16 * (1 / R) * k * imp * depth - m * G) < 2e-5 * m * G  # constants are f32-rounded in the oracle
    # the cube reads (0,0,0,1) as the reference sets it (cube_pick.py:94)
    assert np.allclose(q[0, 12:16], [0, 0, 0, 1], atol=1e-9)


def test_arm_holds_home_pose_pd_balances_gravity():
    o = orc.Oracle(make().build())
    o.reset([0.65, 0, 0.02], [0, 0, 0, 1], HOME)
    for _ in range(600):
        o.step()
    q, v = o.state()
    assert np.abs(v[0, :9]).max() < 1e-8
    o.forward()
    act = o.read(orc.F_QFRC_ACT)[:7]
    bias = o.read(orc.F_QFRC_BIAS)[:7]
    assert np.allclose(act, bias, atol=1e-6)
    # steady-state offset kp (q* - q) = g(q)
    assert np.allclose(np.array(models.FRANKA_KP[:7]) * (HOME[:7] - q[0, :7]), bias, atol=1e-4)
    assert np.abs(q[0, :7] - HOME[:7]).max() < 0.02


def _kinetic_energy_fd(o, q, v, eps=1e-6):
    """T = 1/2 sum m |v_com|^2 + 1/2 w^T I w from finite differences of FK only."""
    sp = o.spec
    nb = sp.nbody

    def fk(qq):
        o.write(orc.F_QPOS, qq)
        o.fk()
        return o.read(orc.F_XIPOS).reshape(nb, 3), o.read(orc.F_XQUAT).reshape(nb, 4)

    def integrate(qq, vv, h):
        out = qq.copy()
        out[:9] += h * vv[:9]
        out[9:12] += h * vv[9:12]
        w = vv[12:15]
        ang = np.linalg.norm(w) * h
        if abs(ang) > 0:
            ax = w / np.linalg.norm(w) * np.sign(h)
            dq = np.r_[np.cos(abs(ang) / 2), ax * np.sin(abs(ang) / 2)]
            out[12:16] = qmul(dq, qq[12:16])
        return out

    p1, r1 = fk(integrate(q, v, eps))
    p0, r0 = fk(integrate(q, v, -eps))
    T = 0.0
    pc, rc = fk(q)
    for b in range(1, nb):
        vcom = (p1[b] - p0[b]) / (2 * eps)
        dq = qmul(r1[b], qconj(r0[b]))
        w = 2 * dq[1:] / (2 * eps) * np.sign(dq[0])
        R = q2mat(rc[b])
        i = sp.body[b].inertia
        Ib = np.array([[i[0], i[3], i[4]], [i[3], i[1], i[5]], [i[4], i[5], i[2]]])
        Iw = R @ Ib @ R.T
        T += 0.5 * sp.body[b].mass * vcom @ vcom + 0.5 * w @ Iw @ w
    return T


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def qconj(a):
    return np.array([a[0], -a[1], -a[2], -a[3]])


def q2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _random_state(rng):
    q = np.zeros(16)
    q[:7] = rng.uniform(-1.5, 1.5, 7)
    q[3] = rng.uniform(-2.8, -0.3)
    q[7:9] = rng.uniform(0, 0.04, 2)
    q[9:12] = rng.uniform(-0.5, 0.5, 3) + [0, 0, 1.0]
    quat = rng.normal(size=4)
    q[12:16] = quat / np.linalg.norm(quat)
    v = rng.uniform(-1, 1, 15)
    return q, v


def test_mass_matrix_symmetric_pd_and_matches_kinetic_energy():
    sb = passive_arm(make())
    o = orc.Oracle(sb.build())
    rng = np.random.default_rng(0)
    arm = np.array([d["armature"] for d in sb.dofs])
    for _ in range(5):
        q, v = _random_state(rng)
        o.write(orc.F_QPOS, q)
        o.write(orc.F_QVEL, v)
        o.forward()
        M = o.read(orc.F_M).reshape(15, 15)
        assert np.allclose(M, M.T, atol=1e-14)
        assert np.linalg.eigvalsh(M).min() > 0
        T = _kinetic_energy_fd(o, q, v)
        assert abs(0.5 * v @ (M - np.diag(arm)) @ v - T) < 1e-6 * max(1.0, T)


def test_crb_cholesky_equals_aba():
    """Two independent forward-dynamics algorithms agree (App. D-4)."""
    sb = make()
    sb.opt.update(implicit_damping=0, enable_collision=0, enable_joint_limit=0)
    o = orc.Oracle(sb.build())
    rng = np.random.default_rng(1)
    for _ in range(5):
        q, v = _random_state(rng)
        o.write(orc.F_QPOS, q)
        o.write(orc.F_QVEL, v)
        o.write(orc.F_TARGET, np.r_[rng.uniform(-1, 1, 9), np.zeros(6)])
        o.forward()
        a_crb = o.read(orc.F_QACC_SMOOTH)
        a_aba = o.aba()
        assert np.allclose(a_crb, a_aba, rtol=1e-9, atol=1e-9)


def test_bias_force_matches_lagrangian_finite_differences():
    """c(q, qd) = d/dt(dT/dqd) - dT/dq + dV/dq evaluated numerically for the arm joints."""
    sb = passive_arm(make())
    o = orc.Oracle(sb.build())
    sp = o.spec
    rng = np.random.default_rng(2)
    q, v = _random_state(rng)
    v[9:] = 0  # keep the free body out of it: its coordinates are not Lagrangian (quaternion)
    o.write(orc.F_QPOS, q)
    o.write(orc.F_QVEL, v)
    o.forward()
    bias = o.read(orc.F_QFRC_BIAS)[:9]

    def Mof(qq):
        o.write(orc.F_QPOS, qq)
        o.forward()
        return o.read(orc.F_M).reshape(15, 15)[:9, :9]

    def Vof(qq):
        o.write(orc.F_QPOS, qq)
        o.fk()
        com = o.read(orc.F_XIPOS).reshape(-1, 3)
        return sum(sp.body[b].mass * G * com[b, 2] for b in range(1, sp.nbody - 1))

    eps = 1e-6
    va = v[:9]
    c = np.zeros(9)
    M0 = Mof(q)
    dM = []
    for k in range(9):
        e = np.zeros(16)
        e[k] = eps
        dM.append((Mof(q + e) - Mof(q - e)) / (2 * eps))
        c[k] += (Vof(q + e) - Vof(q - e)) / (2 * eps)
    Mdot = sum(dM[k] * va[k] for k in range(9))
    c += Mdot @ va
    for k in range(9):
        c[k] -= 0.5 * va @ dM[k] @ va
    assert np.allclose(bias, c, atol=2e-5)


def test_energy_conservation_passive_mechanism():
    sb = passive_arm(make(cube_pos=(0.65, 0, 3.0)))
    for d in sb.dofs:
        d["armature"] = 0.0
    drift = []
    for dt in (1e-3, 5e-4):
        sb.opt["dt"] = dt
        o = orc.Oracle(sb.build())
        sp = o.spec
        q = np.r_[HOME, [0.65, 0, 3.0], [1, 0, 0, 0]]
        v = np.r_[0.3, -0.2, 0.4, 0.1, -0.5, 0.2, 0.3, 0, 0, 0.1, 0.2, 0.0, 1.0, -2.0, 0.5]
        o.write(orc.F_QPOS, q)
        o.write(orc.F_QVEL, v)

        def energy():
            o.forward()
            M = o.read(orc.F_M).reshape(15, 15)
            vv = o.read(orc.F_QVEL)
            com = o.read(orc.F_XIPOS).reshape(-1, 3)
            V = sum(sp.body[b].mass * G * com[b, 2] for b in range(1, sp.nbody))
            return 0.5 * vv @ M @ vv + V

        E0 = energy()
        for _ in range(int(round(0.2 / dt))):
            o.step()
        drift.append(abs(energy() - E0) / abs(E0))
    assert drift[0] < 2e-3
    assert drift[1] < 0.65 * drift[0]  # first-order integrator: halves with dt
    # angular momentum of the free cube about its COM is conserved (sphere-like inertia -> w const)
    assert np.allclose(o.read(orc.F_QVEL)[12:15], [1.0, -2.0, 0.5], atol=1e-9)


def test_joint_limit_holds_against_pd():
    o = orc.Oracle(make().build())
    o.reset([0.65, 0, 0.02], [0, 0, 0, 1], HOME)
    tgt = HOME.copy()
    tgt[3] = 1.0  # joint4 upper limit is -0.0698
    o.set_targets(tgt)
    for _ in range(400):
        o.step()
    q, v = o.state()
    assert q[0, 3] > -0.0698 - 1e-6  # pushed into the limit ...
    assert q[0, 3] < -0.0698 + 0.02  # ... but held there by the soft constraint
    assert abs(v[0, 3]) < 1e-3


def test_box_box_stack_rests():
    """Cube resting on a second cube (face-face, 4 points) which rests on the plane."""
    sb = make()
    m = sb.bodies[sb.body_index("cube")]["mass"]
    sb.add_body("cube2", 0, pos=(0.65, 0.0, 0.06), jtype=S.JNT_FREE, mass=m, inertia=S.box_inertia(m, (0.02,) * 3))
    sb.add_geom("cube2", S.GEOM_BOX, size=(0.02, 0.02, 0.02))
    # trim the arm so the spec fits MIR_MAX_DOF: drop gripper + wrist (not needed here)
    keep = [b for b in sb.bodies if b["name"] in ("world", "link0", "link1", "cube", "cube2")]
    names = [b["name"] for b in keep]
    sb2 = S.SceneBuilder()
    sb2.add_geom(0, S.GEOM_PLANE)
    sb2.add_body("cube", 0, pos=(0.65, 0, 0.02), jtype=S.JNT_FREE, mass=m, inertia=S.box_inertia(m, (0.02,) * 3))
    sb2.add_geom("cube", S.GEOM_BOX, size=(0.02,) * 3)
    sb2.add_body("cube2", 0, pos=(0.655, 0.003, 0.06), quat=(np.cos(0.2), 0, 0, np.sin(0.2)), jtype=S.JNT_FREE,
                 mass=m, inertia=S.box_inertia(m, (0.02,) * 3))
    sb2.add_geom("cube2", S.GEOM_BOX, size=(0.02,) * 3)
    sb2.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    assert names  # silence lints
    o = orc.Oracle(sb2.build())
    for _ in range(500):
        o.step()
    q, v = o.state()
    assert np.abs(v).max() < 1e-3  # soft pyramidal friction leaves a slow creep, as in MuJoCo
    ncon, nefc, _ = o.counts()
    assert ncon == 12 and nefc == 48  # 4 plane-box + 8 box-box (rotated square on square -> octagon)
    assert 0.0195 < q[0, 2] < 0.02 and 0.059 < q[0, 9] < 0.06
    assert np.allclose(q[0, 7:9], [0.655, 0.003], atol=2e-4)  # friction holds it in place
    o.forward()
    f = o.read(orc.F_EFCFORCE)
    J = o.read(orc.F_J).reshape(-1, o.nv)
    qf = J.T @ f
    assert abs(qf[2] - m * G) < 1e-5 and abs(qf[8] - m * G) < 1e-5


def test_reward_threshold_is_strict_float32():
    sb = make()
    o = orc.Oracle(sb.build())
    for z, want in ((0.1, 0), (np.float32(0.1), 0), (0.10001, 1), (0.2, 1), (0.0999, 0)):
        o.reset([0.5, 0, z], [0, 0, 0, 1], HOME)
        _, env, r, t = o.get_obs()
        assert r[0] == want and t[0] == want
        assert env.shape == (1, 11)


def test_scripted_grasp_lifts_cube_and_reward_flips_at_threshold():
    """App. D-7: symmetric finger closure keeps the cube in place, the lift raises it, and the
    reward / terminated flag flips exactly when the cube's z exceeds 0.1 (cube_pick.py:134, env.py:63)."""
    import json
    import os

    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    B = len(G_["cube_xy"])
    o = orc.Oracle(make().build(), B)
    pos = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    o.reset(pos, np.tile([0, 0, 0, 1.0], (B, 1)), np.tile(HOME, (B, 1)))
    T = np.array(G_["targets"], np.float32)
    flipped = np.zeros(B, bool)
    for s, name in enumerate(G_["stages"]):
        for k in range(G_["steps_per_stage"]):
            o.step_batch(T[:, s])
            a, e, r, term = o.get_obs()
            assert np.array_equal(r == 1, np.float32(e[:, 2]) > np.float32(0.1)) and np.array_equal(term.astype(bool), r == 1)
            flipped |= r == 1
            # line search: phi' is resolved to the evaluation's own rounding floor, not bisected beyond it (NOTEBOOK.md section 2, item 5):
            # a handful of evaluations per Newton iteration even with a dozen stiff contact rows
            assert max(o.read(orc.F_DBG_LS, i).max(initial=0) for i in range(B)) <= 16
        if name == "close":
            # squeezed between the pads: the cube stays within 2 cm (the wrist joints, limited to +-12 N m by the MJCF
            # actuators, are still settling while the fingers close, so it is nudged by a few millimetres)
            assert np.abs(e[:, :2] - pos[:, :2]).max() < 2e-2
            assert all(o.counts(i)[0] >= 8 for i in range(B))  # plane + two finger pads
    assert flipped.all() and (e[:, 2] > 0.2).all()             # every env picked its cube up
    assert np.abs(a[:, 7] - a[:, 8]).max() < 1e-3               # fingers closed symmetrically on the 4 cm cube
    assert np.abs(a[:, 7] - 0.0185).max() < 2e-3


def test_constraint_solve_reaches_the_optimum_of_the_convex_problem():
    """Independent check of the oracle's constraint solve: the constrained acceleration must minimise
    1/2 (a - a0)^T Mt (a - a0) + sum_r 1/2 D_r min(0, J_r a - aref_r)^2 -- verified with SciPy on the exported M~, J, aref, D
    in a contact-rich state of the stack scene (cubes resting, a cube pressed into the finger, joints pushed into limits)."""
    from scipy.optimize import minimize

    b = models.franka_cube_stack_scene()
    spec = b.build()
    o = orc.Oracle(spec, 1)
    z = models.STACK_CUBE_Z
    pos = [[(-0.3, -0.2, z), (-0.1, 0.2, z), (0.1, -0.2, z), (0.2, 0.2, z), (0.3, 0.0, z)]]
    home = np.array([models.FRANKA_HOME])
    o.reset(np.array(pos), np.tile([0, 0, 0, 1.0], (1, 5, 1)), home)
    for _ in range(30):
        o.step()
    names = [x["name"] for x in b.bodies]
    hand = names.index("hand")
    q = o.read(orc.F_QPOS)
    q[3] = -0.05                                                            # joint4 past its upper limit (-0.0698)
    o.write(orc.F_QPOS, q)
    o.fk()
    q[9:12] = o.read(orc.F_XPOS).reshape(-1, 3)[hand] + [0.0, 0.0, 0.012]  # cube_1 overlapping the hand box
    o.write(orc.F_QPOS, q)
    v = o.read(orc.F_QVEL)
    v[:9] = np.random.default_rng(0).uniform(-1, 1, 9)
    o.write(orc.F_QVEL, v)
    o.forward()
    ncon, nefc, niter = o.counts()
    assert ncon > 16 and nefc > 4 * ncon  # 16 resting contacts + arm-cube contacts + at least one limit row
    nv = o.nv
    Mt = o.read(orc.F_MT).reshape(-1, nv)[:nv]
    J = o.read(orc.F_J).reshape(-1, nv)[:nefc]
    aref, D = o.read(orc.F_AREF)[:nefc], o.read(orc.F_EFCD)[:nefc]
    a0, a_orc = o.read(orc.F_QACC_SMOOTH), o.read(orc.F_QACC)

    def cost(a):
        r = np.minimum(0.0, J @ a - aref)
        d = a - a0
        return 0.5 * d @ Mt @ d + 0.5 * np.sum(D * r * r)

    def grad(a):
        r = np.minimum(0.0, J @ a - aref)
        return Mt @ (a - a0) + J.T @ (D * r)

    sol = minimize(cost, a0, jac=grad, method="BFGS", options=dict(gtol=1e-9, maxiter=5000))
    scale = max(1.0, np.abs(a_orc).max())
    assert cost(a_orc) <= cost(sol.x) * (1 + 1e-9) + 1e-12          # the oracle's point is at least as good as SciPy's
    assert np.linalg.norm(grad(a_orc)) < 1e-6 * max(1.0, np.linalg.norm(Mt @ a0))  # stationarity (the problem is strictly convex)
    assert np.abs(sol.x - a_orc).max() < 1e-4 * scale


def test_contact_jacobian_matches_finite_differences_of_the_kinematics():
    """Rows of the constraint Jacobian (pyramidal contact rows n +- mu t) against finite differences of the world position of
    the contact point carried by each of the two bodies, dof by dof (scalar joints, free translations, free rotations in
    world axes): pins frames, signs and the chain rule of the contact rows independently of the Jacobian code."""
    b = models.franka_cube_stack_scene()
    spec = b.build()
    o = orc.Oracle(spec, 1)
    z = models.STACK_CUBE_Z
    pos = [[(-0.3, -0.2, z), (-0.1, 0.2, z), (0.1, -0.2, z), (0.2, 0.2, z), (0.3, 0.0, z)]]
    o.reset(np.array(pos), np.tile([0, 0, 0, 1.0], (1, 5, 1)), np.array([models.FRANKA_HOME]))
    for _ in range(20):
        o.step()
    names = [x["name"] for x in b.bodies]
    q0 = o.read(orc.F_QPOS)
    q0[9:12] = o.read(orc.F_XPOS).reshape(-1, 3)[names.index("left_finger")] + [0.0, 0.0, -0.012]
    tilt = np.array([np.cos(0.15), np.sin(0.15) * 0.6, np.sin(0.15) * 0.8, 0.0])
    q0[12:16] = tilt / np.linalg.norm(tilt)
    o.write(orc.F_QPOS, q0)
    o.forward()
    ncon, nefc, _ = o.counts()
    nv = o.nv
    J = o.read(orc.F_J).reshape(-1, nv)[:nefc]
    cpos = o.read(orc.F_CPOS).reshape(-1, 3)[:ncon]
    cfrm = o.read(orc.F_CFRAME).reshape(-1, 3, 3)[:ncon]
    xpos0, xquat0 = o.read(orc.F_XPOS).reshape(-1, 3), o.read(orc.F_XQUAT).reshape(-1, 4)

    def rot(qw):
        w, x, y, zz = np.asarray(qw) / np.linalg.norm(qw)  # (model quaternions are float32-rounded: unit only to 1e-8)
        return np.array([[1 - 2 * (y * y + zz * zz), 2 * (x * y - w * zz), 2 * (x * zz + w * y)],
                         [2 * (x * y + w * zz), 1 - 2 * (x * x + zz * zz), 2 * (y * zz - w * x)],
                         [2 * (x * zz - w * y), 2 * (y * zz + w * x), 1 - 2 * (x * x + y * y)]])

    def qmul(a, c):
        return np.array([a[0] * c[0] - a[1:] @ c[1:], *(a[0] * c[1:] + c[0] * a[1:] + np.cross(a[1:], c[1:]))])

    # which two bodies each contact joins: recover them as the bodies whose surface the point lies on is not needed -- the
    # finite-difference relative motion is computed for EVERY ordered body pair and the pair that reproduces the row is found
    eps = 1e-6
    nb = spec.nbody
    dP = np.zeros((nv, nb, ncon, 3))  # d(world position of contact c as carried by body b) / d(dof i)
    qadr_free = {}
    nq_scalar = 9
    for k in range(5):
        qadr_free[k] = (9 + 7 * k, 9 + 6 * k)
    for i in range(nv):
        q = q0.copy()
        if i < nq_scalar:
            q[i] += eps
        else:
            k, r = divmod(i - 9, 6)
            qa = 9 + 7 * k
            if r < 3:
                q[qa + r] += eps
            else:
                ax = np.zeros(3)
                ax[r - 3] = 1.0
                dq = np.array([np.cos(eps / 2), *(np.sin(eps / 2) * ax)])
                q[qa + 3:qa + 7] = qmul(dq, q[qa + 3:qa + 7])  # world-frame angular velocity
        o.write(orc.F_QPOS, q)
        o.fk()
        xp, xq = o.read(orc.F_XPOS).reshape(-1, 3), o.read(orc.F_XQUAT).reshape(-1, 4)
        for bb in range(nb):
            R0, R1 = rot(xquat0[bb]), rot(xq[bb])
            local = (cpos - xpos0[bb]) @ R0          # (ncon,3): R0^T (p - x0)
            dP[i, bb] = (xp[bb] + local @ R1.T - cpos) / eps
    o.write(orc.F_QPOS, q0)
    o.fk()
    mu = 1.0
    checked = 0
    # locate the 4-row block of every contact by matching: each contact must have four rows r with
    # J[r] = d/dq [(p_b2 - p_b1) . (n +- mu t)] for some body pair (b1, b2)
    dirs = lambda c: [cfrm[c][0] + mu * cfrm[c][1], cfrm[c][0] - mu * cfrm[c][1], cfrm[c][0] + mu * cfrm[c][2], cfrm[c][0] - mu * cfrm[c][2]]  # noqa: E731
    for c in range(ncon):
        found = False
        for b1 in range(nb):
            for b2 in range(nb):
                if b1 == b2:
                    continue
                rel = dP[:, b2, c, :] - dP[:, b1, c, :]            # (nv,3)
                want = np.stack([rel @ d for d in dirs(c)])        # (4,nv)
                if np.abs(want).max() < 1e-9:
                    continue
                for r0 in range(0, nefc - 3):
                    if np.abs(J[r0:r0 + 4] - want).max() < 2e-4 * max(1.0, np.abs(want).max()):
                        found = True
                        break
                if found:
                    break
            if found:
                break
        assert found, f"no four Jacobian rows reproduce the finite-difference relative motion at contact {c}"
        checked += 1
    assert checked == ncon and ncon > 16


def test_box_box_narrowphase_against_a_numpy_separating_axis_test():
    """Random pairs of overlapping / separated cubes (the stack scene's cube_1 and cube_2 floating in the air): contacts are
    reported iff a NumPy 15-axis separating-axis test finds no separating axis; the reported normal is the axis of least
    penetration (up to the routine's edge-axis bias), pointing from the first box to the second; the deepest point's
    distance equals minus that overlap; every contact point lies inside both boxes inflated by the penetration."""
    b = models.franka_cube_stack_scene()
    spec = b.build()
    o = orc.Oracle(spec, 1)
    rng = np.random.default_rng(5)
    h = 0.02

    def rot(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def sat(pa, Ra, pb, Rb):
        t = pb - pa
        axes = [Ra[:, i] for i in range(3)] + [Rb[:, j] for j in range(3)]
        for i in range(3):
            for j in range(3):
                c = np.cross(Ra[:, i], Rb[:, j])
                if np.linalg.norm(c) > 1e-3:
                    axes.append(c / np.linalg.norm(c))
        best = -np.inf
        for L in axes:
            ra = h * np.abs(Ra.T @ L).sum()
            rb = h * np.abs(Rb.T @ L).sum()
            best = max(best, abs(t @ L) - (ra + rb))
        return best  # > 0: separated by that much; < 0: minus the least overlap

    hits = misses = 0
    far = [(0.5, 0.3, 2.0), (0.5, -0.3, 2.0), (-0.6, 0.3, 2.0)]
    for trial in range(60):
        qa, qb = rng.normal(size=4), rng.normal(size=4)
        qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
        pa = np.array([0.3, 0.0, 1.5])
        pb = pa + rng.normal(size=3) * 0.025
        q = o.read(orc.F_QPOS)
        q[9:12], q[12:16], q[16:19], q[19:23] = pa, qa, pb, qb
        for k, f in enumerate(far):
            q[23 + 7 * k:26 + 7 * k] = f
        o.write(orc.F_QPOS, q)
        o.forward()
        ncon = o.counts()[0]
        s = sat(pa, rot(qa), pb, rot(qb))
        if s > 1e-9:
            assert ncon == 0, (trial, s)
            misses += 1
            continue
        if s > -1e-6:
            continue  # grazing: either answer is fine
        hits += 1
        assert ncon >= 1, (trial, s)
        cpos = o.read(orc.F_CPOS).reshape(-1, 3)[:ncon]
        dist = o.read(orc.F_CDIST)[:ncon]
        n = o.read(orc.F_CFRAME).reshape(-1, 3, 3)[0, 0]
        assert n @ (pb - pa) > -1e-9                                  # from cube_1 towards cube_2
        assert dist.min() >= s * 1.06 - 1e-6 and dist.min() <= s * 0.94 + 1e-6 or abs(dist.min() - s) < 2e-3 * h, (trial, dist.min(), s)
        for p in cpos:                                                # inside both boxes (inflated by the overlap)
            for pc, R in ((pa, rot(qa)), (pb, rot(qb))):
                assert (np.abs(R.T @ (p - pc)) <= h - s + 1e-6).all(), trial
    assert hits >= 15 and misses >= 10
