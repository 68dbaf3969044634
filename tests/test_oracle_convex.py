"""Known-answer tests of the oracle's convex narrowphase (Minkowski Portal Refinement on support mappings, plus the closed-form
plane-sphere / plane-capsule cases) against analytic sphere / capsule / box configurations.  The reference delegates this to
Genesis (MPR is its default convex-convex path); there are no reference vectors, so the anchors are
geometry.  Spheres and capsules are handled as a core (point / segment) plus a radius: GJK on the cores gives depth, normal (from
geom 1 to geom 2) and the mid-surface position exactly; only when the cores themselves overlap does MPR on the full shapes take
over, with its known character: the penetration is measured where the ray from the interior point leaves A - B, so the
depth is >= the true minimum and the position is a common point of the two shapes inside the overlap."""
import ctypes as C

import numpy as np
import pytest

import orc

PLANE, BOX, SPHERE, CAPSULE = 0, 1, 2, 3
I = (1.0, 0, 0, 0)


def narrow(t1, s1, p1, q1, t2, s2, p2, q2, f32=False):
    lib = orc.load(f32)
    lib.orc_narrowphase.restype = C.c_int
    arr = lambda v, n: (C.c_double * n)(*(list(v) + [0.0] * (n - len(v))))  # noqa: E731
    pts, nrm = (C.c_double * 32)(), (C.c_double * 3)()
    cnt = lib.orc_narrowphase(int(t1), arr(s1, 3), arr(p1, 3), arr(q1, 4), int(t2), arr(s2, 3), arr(p2, 3), arr(q2, 4), pts, nrm)
    P = np.array(pts[:4 * cnt]).reshape(cnt, 4)
    return cnt, P[:, :3], P[:, 3], np.array(nrm[:])


def quat_axis(axis, ang):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    return (np.cos(ang / 2), *(np.sin(ang / 2) * a))


def test_sphere_sphere_depth_normal_position():
    rng = np.random.default_rng(0)
    for _ in range(50):
        r1, r2 = rng.uniform(0.02, 0.3, 2)
        c1 = rng.uniform(-1, 1, 3)
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        gap = rng.uniform(-0.9, 0.3) * min(r1, r2)          # negative = penetration
        c2 = c1 + (r1 + r2 + gap) * u
        cnt, pos, dist, n = narrow(SPHERE, [r1], c1, I, SPHERE, [r2], c2, I)
        if gap >= 0:
            assert cnt == 0
            continue
        assert cnt == 1
        assert abs(dist[0] - gap) < 5e-6
        assert np.abs(n - u).max() < 2e-3 * max(1.0, 0.05 / -gap)   # the normal is ill-conditioned for grazing contacts
        mid = 0.5 * ((c1 + r1 * u) + (c2 - r2 * u))
        assert np.abs(pos[0] - mid).max() < 0.5 * -gap + 2e-3 * max(r1, r2) + 2e-5


def test_sphere_box_face_edge_corner():
    h = np.array([0.1, 0.2, 0.15])
    r = 0.05
    # face: sphere above the +z face, 1 cm deep
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, SPHERE, [r], [0.03, -0.05, h[2] + r - 0.01], I)
    assert cnt == 1 and abs(dist[0] + 0.01) < 1e-5 and np.abs(n - [0, 0, 1]).max() < 1e-3
    assert np.abs(pos[0][:2] - [0.03, -0.05]).max() < 1e-3 and h[2] - 0.01 - 1e-6 <= pos[0][2] <= h[2] + 1e-6  # inside the overlap
    # corner: along the diagonal of the (+,+,+) corner
    u = np.ones(3) / np.sqrt(3)
    c = h + (r - 0.004) * u
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, SPHERE, [r], c, I)
    assert cnt == 1 and abs(dist[0] + 0.004) < 1e-9 and np.abs(n - u).max() < 1e-9
    assert np.abs(pos[0] - (h + 0.5 * -0.004 * u)).max() < 1e-9
    # edge: (+x, +z) edge, sphere centre in the plane y = 0.07
    u = np.array([1, 0, 1]) / np.sqrt(2)
    c = np.array([h[0], 0.07, h[2]]) + (r - 0.006) * u
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, SPHERE, [r], c, I)
    assert cnt == 1 and abs(dist[0] + 0.006) < 1e-9 and np.abs(n - u).max() < 1e-9
    # rotated and translated frame gives the same answer expressed in that frame
    q = quat_axis([1, 2, 3], 0.7)
    lib = orc.load(False)
    Rm = np.zeros(9)
    qq = (C.c_double * 4)(*q)
    from scipy.spatial.transform import Rotation as Rsc
    Rw = Rsc.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
    t = np.array([0.3, -0.2, 0.5])
    cnt2, pos2, dist2, n2 = narrow(BOX, h, t, q, SPHERE, [r], t + Rw @ c, I)
    assert cnt2 == 1 and abs(dist2[0] - dist[0]) < 1e-6 and np.abs(n2 - Rw @ n).max() < 1e-5
    assert np.abs(pos2[0] - (t + Rw @ pos[0])).max() < 1e-5
    # separated
    assert narrow(BOX, h, [0, 0, 0], I, SPHERE, [r], [0, 0, h[2] + r + 1e-3], I)[0] == 0
    del lib, Rm, qq


def test_deep_penetration_goes_through_portal_refinement():
    """Cores overlapping (sphere centre inside the box, crossing capsule axes): MPR on the full shapes.  Along the centre ray
    its answer is exact; off it, depth >= the true minimum."""
    h = np.array([0.1, 0.2, 0.15])
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, SPHERE, [0.05], [0, 0, 0.14], I)      # centre 1 cm below the top face
    assert cnt == 1 and abs(dist[0] + 0.06) < 1e-5 and np.abs(n - [0, 0, 1]).max() < 1e-3
    assert -0.06 + 0.15 - 0.05 - 1e-6 <= pos[0][2] <= 0.15 + 1e-6                          # a common point of the two shapes
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, SPHERE, [0.05], [0.02, 0.05, 0.14], I)  # off the centre ray
    assert cnt == 1 and 0.06 - 1e-6 <= -dist[0] <= 0.075 and n[2] > 0.9
    # two capsules whose axes cross at right angles through each other: separation along the common normal = 2 r
    qx = quat_axis([0, 1, 0], np.pi / 2)
    cnt, pos, dist, n = narrow(CAPSULE, [0.04, 0.2], [0, 0, 0], I, CAPSULE, [0.04, 0.2], [0, 0.001, 0], qx)
    assert cnt == 1 and abs(dist[0] + 0.079) < 1e-4 and abs(n[1]) > 0.999
    # concentric spheres: any direction is as good as another, the depth is r1 + r2
    cnt, pos, dist, n = narrow(SPHERE, [0.05], [0.1, 0.1, 0.1], I, SPHERE, [0.03], [0.1, 0.1, 0.1], I)
    assert cnt == 1 and abs(dist[0] + 0.08) < 1e-5 and abs(np.linalg.norm(n) - 1) < 1e-6


def test_capsule_capsule_crossed_and_parallel():
    r, hl = 0.04, 0.2
    # crossed at right angles, axes z and x, centres 0.07 apart along y: closest points are the centres
    qx = quat_axis([0, 1, 0], np.pi / 2)   # z axis -> x axis
    cnt, pos, dist, n = narrow(CAPSULE, [r, hl], [0, 0, 0], I, CAPSULE, [r, hl], [0, 0.07, 0], qx)
    assert cnt == 1 and abs(dist[0] - (0.07 - 2 * r)) < 5e-6 and np.abs(n - [0, 1, 0]).max() < 1e-3
    assert np.abs(pos[0] - [0, 0.035, 0]).max() < 0.005 + 1e-3
    # end-to-end along z: spherical caps, 5 mm deep
    d = 2 * hl + 2 * r - 0.005
    cnt, pos, dist, n = narrow(CAPSULE, [r, hl], [0, 0, 0], I, CAPSULE, [r, hl], [0, 0, d], I)
    assert cnt == 1 and abs(dist[0] + 0.005) < 5e-6 and np.abs(n - [0, 0, 1]).max() < 1e-3
    # parallel side by side: depth and normal are determined, the position anywhere on the common segment
    cnt, pos, dist, n = narrow(CAPSULE, [r, hl], [0, 0, 0], I, CAPSULE, [r, hl], [0.06, 0, 0.05], I)
    assert cnt == 1 and abs(dist[0] + 0.02) < 5e-6 and np.abs(n - [1, 0, 0]).max() < 1e-3
    assert abs(pos[0][0] - 0.03) < 0.01 + 1e-3 and abs(pos[0][1]) < 1e-3 and -hl + 0.05 - 1e-3 <= pos[0][2] <= hl + 1e-3
    # apart
    assert narrow(CAPSULE, [r, hl], [0, 0, 0], I, CAPSULE, [r, hl], [0.081, 0, 0], I)[0] == 0


def test_sphere_capsule_and_capsule_box():
    r, hl, rs = 0.03, 0.1, 0.05
    # sphere next to the cylindrical part
    cnt, pos, dist, n = narrow(SPHERE, [rs], [0.07, 0, 0.04], I, CAPSULE, [r, hl], [0, 0, 0], I)
    assert cnt == 1 and abs(dist[0] + 0.01) < 5e-6 and np.abs(n - [-1, 0, 0]).max() < 1e-3
    # sphere beyond the cap
    c = np.array([0.0, 0.0, hl]) + (r + rs - 0.008) * np.array([0.6, 0.0, 0.8])
    cnt, pos, dist, n = narrow(CAPSULE, [r, hl], [0, 0, 0], I, SPHERE, [rs], c, I)
    assert cnt == 1 and abs(dist[0] + 0.008) < 1e-9 and np.abs(n - [0.6, 0, 0.8]).max() < 1e-9
    # capsule lying on a box face (axis along x), 3 mm deep: normal = face normal, depth exact
    h = np.array([0.2, 0.2, 0.05])
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, CAPSULE, [r, hl], [0.02, 0.01, h[2] + r - 0.003], quat_axis([0, 1, 0], np.pi / 2))
    assert cnt == 1 and abs(dist[0] + 0.003) < 1e-5 and np.abs(n - [0, 0, 1]).max() < 2e-3
    # capsule standing on a corner region of the box top
    cnt, pos, dist, n = narrow(BOX, h, [0, 0, 0], I, CAPSULE, [r, hl], [0.19, 0.19, h[2] + r + hl - 0.002], I)
    assert cnt == 1 and abs(dist[0] + 0.002) < 1e-5 and np.abs(n - [0, 0, 1]).max() < 2e-3


def test_plane_sphere_and_plane_capsule_closed_form():
    cnt, pos, dist, n = narrow(PLANE, [0, 0, 0], [0, 0, 0], I, SPHERE, [0.05], [0.3, 0.2, 0.04], I)
    assert cnt == 1 and abs(dist[0] + 0.01) < 1e-12 and np.allclose(n, [0, 0, 1]) and np.allclose(pos[0], [0.3, 0.2, -0.005])
    assert narrow(PLANE, [0, 0, 0], [0, 0, 0], I, SPHERE, [0.05], [0, 0, 0.0501], I)[0] == 0
    # capsule lying flat: both end spheres touch; tilted: only the lower one
    q = quat_axis([0, 1, 0], np.pi / 2)
    cnt, pos, dist, n = narrow(PLANE, [0, 0, 0], [0, 0, 0], I, CAPSULE, [0.03, 0.1], [0, 0, 0.025], q)
    assert cnt == 2 and np.allclose(dist, -0.005) and np.allclose(sorted(pos[:, 0]), [-0.1, 0.1], atol=1e-12)
    q = quat_axis([0, 1, 0], np.pi / 2 - 0.3)
    cnt, pos, dist, n = narrow(PLANE, [0, 0, 0], [0, 0, 0], I, CAPSULE, [0.03, 0.1], [0, 0, 0.05], q)
    assert cnt == 1 and abs(dist[0] - (0.05 - 0.1 * np.sin(0.3) - 0.03)) < 1e-12
    # tilted plane
    qp = quat_axis([1, 0, 0], 0.4)
    nz = np.array([0, -np.sin(0.4), np.cos(0.4)])
    cnt, pos, dist, n = narrow(PLANE, [0, 0, 0], [0.1, 0.2, 0.3], qp, SPHERE, [0.05], np.array([0.1, 0.2, 0.3]) + 0.03 * nz, I)
    assert cnt == 1 and abs(dist[0] + 0.02) < 1e-12 and np.allclose(n, nz)


def test_float32_port_agrees_with_float64_on_random_convex_pairs():
    """The float32 build of the same source (the yardstick for the kernel).  Contacts shallower than the radii go through GJK
    on the cores, an exact feature pair: depth within 2e-6, normal within 1e-4, position within 1e-5.  Deep ones (cores
    overlapping) go through MPR, whose result depends on where the portal refinement stops: both precisions must report the
    contact, depths within 20 %."""
    rng = np.random.default_rng(3)
    shallow = deep = 0
    for _ in range(1500):
        t1, t2 = rng.choice([BOX, SPHERE, CAPSULE], 2)
        if t1 == BOX and t2 == BOX:
            continue
        s1, s2 = rng.uniform(0.03, 0.15, 3), rng.uniform(0.03, 0.15, 3)
        p1, p2 = rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.2, 0.2, 3)
        q1, q2 = rng.normal(size=4), rng.normal(size=4)
        a = narrow(t1, s1, p1, q1, t2, s2, p2, q2)
        b = narrow(t1, s1, p1, q1, t2, s2, p2, q2, f32=True)
        if a[0] == 0 or a[2][0] > -1e-4:
            continue
        assert b[0] == 1
        radii = (0 if t1 == BOX else s1[0]) + (0 if t2 == BOX else s2[0])
        if -a[2][0] < 0.9 * radii:      # cores apart by a margin: the GJK path in both precisions
            assert abs(a[2][0] - b[2][0]) < 2e-6
            assert np.abs(a[3] - b[3]).max() < 1e-4 * max(1.0, 0.01 / (radii + a[2][0]))  # (normal = difference of close points / distance)
            assert np.abs(a[1][0] - b[1][0]).max() < 1e-5
            shallow += 1
        elif -a[2][0] > 1.1 * radii:
            assert abs(a[2][0] - b[2][0]) < 0.2 * abs(a[2][0])
            deep += 1
    assert shallow > 100 and deep > 20, (shallow, deep)


def test_sphere_and_capsule_rest_on_the_plane():
    """App. D-2 for the round geoms: a sphere and a lying capsule dropped on the plane come to rest with their centre one radius
    (minus the soft-contact penetration, < 1 mm) above it; the capsule keeps two contact points (its end spheres)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-genesis_amd"))
    from gym_genesis.backend import spec as S

    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    sb.add_body("ball", 0, pos=(0, 0, 0.2), jtype=S.JNT_FREE, mass=0.2, inertia=S.sphere_inertia(0.2, 0.05))
    sb.add_geom("ball", S.GEOM_SPHERE, size=(0.05, 0, 0))
    sb.add_body("rod", 0, pos=(0.5, 0, 0.2), quat=quat_axis([0, 1, 0], np.pi / 2), jtype=S.JNT_FREE, mass=0.2,
                inertia=S.capsule_inertia(0.2, 0.03, 0.1))
    sb.add_geom("rod", S.GEOM_CAPSULE, size=(0.03, 0.1, 0))
    sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
    o = orc.Oracle(sb.build())
    for _ in range(400):
        o.step()
    q, v = o.state()
    assert np.abs(v).max() < 1e-3
    assert 0.049 < q[0, 2] <= 0.05 and 0.029 < q[0, 9] <= 0.03
    assert o.counts()[0] == 3  # one point under the ball, two under the rod
