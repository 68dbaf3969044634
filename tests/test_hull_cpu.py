"""CPU tier for the vertex-hull geoms (MIR_GEOM_HULL: the stand-in for the reference's mesh collision geometry, whose meshes are
Genesis assets outside /root/reference).  A hull enters the narrowphase only through its support mapping (the vertex of largest
projection), so two properties pin it without any reference data:
  * a box given as its 8 corners reproduces the GEOM_BOX results of the same paths -- plane contacts point for point, GJK / MPR
    contacts against spheres and capsules pair by pair;
  * a polytope inscribed in a sphere converges to the sphere: the 12-vertex icosahedron and the 32-vertex pentakis dodecahedron
    give depths between the sphere's depth and the sphere's depth minus (r - inradius), and the 32-vertex error is the smaller.
Both for the float64 oracle and for the HOST build of the device source (tests/convex_host.cpp = mir_convex.h in float32)."""
import ctypes as C

import numpy as np

import orc
import test_convex_host as H
from gym_genesis.backend import spec as S

PLANE, BOX, SPHERE, CAPSULE, HULL = 0, 1, 2, 3, 4
I = (1.0, 0.0, 0.0, 0.0)


def _set_pools(verts):
    flat = (C.c_double * (3 * len(verts)))(*[c for v in verts for c in v])
    orc.load(False).orc_set_hull_pool(flat, len(verts))
    H.host_lib().convex_host_set_pool(flat, len(verts))


def _oracle(t1, s1, p1, q1, t2, s2, p2, q2):
    lib = orc.load(False)
    lib.orc_narrowphase.restype = C.c_int
    arr = lambda v, n: (C.c_double * n)(*(list(v) + [0.0] * (n - len(v))))  # noqa: E731
    pts, nrm = (C.c_double * 32)(), (C.c_double * 3)()
    cnt = lib.orc_narrowphase(int(t1), arr(s1, 3), arr(p1, 3), arr(q1, 4), int(t2), arr(s2, 3), arr(p2, 3), arr(q2, 4), pts, nrm)
    P = np.array(pts[:4 * cnt]).reshape(cnt, 4)
    return cnt, P[:, :3], P[:, 3], np.array(nrm[:])


def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_plane_contacts_of_a_box_given_as_its_corners():
    h = tuple(float(np.float32(x)) for x in (0.05, 0.03, 0.08))   # (the vertex pool is float32, as the device model holds it)
    _set_pools(S.box_hull_vertices(h))
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(300):
        q, p = _rand_quat(rng), [0.0, 0.0, rng.uniform(0.0, 0.09)]
        if rng.random() < 0.3:
            q = np.array(I)          # lying flat: four equal depths, the extremes rule decides
        a = _oracle(PLANE, [0, 0, 0], [0, 0, 0], I, BOX, h, p, q)
        b = _oracle(PLANE, [0, 0, 0], [0, 0, 0], I, HULL, [0, 8, 0], p, q)
        assert a[0] == b[0]
        seen.add(a[0])
        if a[0]:
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert {0, 1, 2, 4} <= seen


def _pairs_box_vs_round(n, seed, hull):
    rows = H.random_pairs(n, seed)
    rows = rows[((rows[:, 0] == 1) ^ (rows[:, 11] == 1))]          # exactly one box
    h = tuple(float(np.float32(x)) for x in (0.06, 0.09, 0.04))
    for r in rows:
        k = 0 if r[0] == 1 else 11
        r[k + 1:k + 4] = (0, 8, 0) if hull else h
        r[k] = HULL if hull else BOX
    return rows, h


def test_gjk_and_mpr_contacts_of_a_box_given_as_its_corners():
    """Box against sphere / capsule, shallow (GJK on the cores) and deep (MPR): the 8-vertex hull gives the box's contact -- in the
    oracle to float64 rounding, in the host build of the device source to float32 rounding."""
    rows_b, h = _pairs_box_vs_round(3000, 33, hull=False)
    rows_h, _ = _pairs_box_vs_round(3000, 33, hull=True)
    _set_pools(S.box_hull_vertices(h))
    ob, oh = H.oracle_pairs(rows_b), H.oracle_pairs(rows_h)
    assert np.array_equal(ob[:, 0], oh[:, 0]) and ob[:, 0].sum() > 300
    hit = ob[:, 0] == 1
    assert np.abs(ob[hit, 1:] - oh[hit, 1:]).max() < 1e-9
    got_b, got_h = np.zeros((rows_b.shape[0], 8), np.float32), np.zeros((rows_b.shape[0], 8), np.float32)
    lib = H.host_lib()
    lib.convex_host_pairs(rows_b.ctypes.data_as(C.c_void_p), got_b.ctypes.data_as(C.c_void_p), rows_b.shape[0])
    lib.convex_host_pairs(rows_h.ctypes.data_as(C.c_void_p), got_h.ctypes.data_as(C.c_void_p), rows_h.shape[0])
    same = got_b[:, 0] == got_h[:, 0]
    assert same.mean() > 0.998
    both = same & (got_b[:, 0] == 1)
    assert np.median(np.abs(got_b[both, 4] - got_h[both, 4])) < 1e-7 and np.quantile(np.abs(got_b[both, 4] - got_h[both, 4]), 0.99) < 2e-5
    # and the device source with hulls against the oracle with hulls, by the bars of the primitive shapes
    rows_cmp = rows_b.copy()     # (compare() reads the radii from the rows: the box rows carry the same radii)
    shallow, deep = H.compare(rows_cmp, got_h, oh)
    assert shallow > 100 and deep > 30, (shallow, deep)


def test_inscribed_polytopes_converge_to_the_sphere():
    r = 0.05
    errs = {}
    for level, inradius in ((0, 0.79465), (1, 0.9)):      # inradius / circumradius of the icosahedron; a lower bound for the 32-vertex solid
        verts = S.icosphere_vertices(r, level)
        assert len(verts) == (12, 32)[level]
        _set_pools(verts)
        rng = np.random.default_rng(7)
        e = []
        for _ in range(200):
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            depth = rng.uniform(0.1, 0.6) * r
            c = (0.1 + r - depth) * u                      # against a sphere of radius 0.1 at the origin
            q = _rand_quat(rng)
            a = _oracle(SPHERE, [0.1], [0, 0, 0], I, SPHERE, [r], c, I)
            b = _oracle(SPHERE, [0.1], [0, 0, 0], I, HULL, [0, len(verts), 0], c, q)
            assert a[0] == 1 and abs(a[2][0] + depth) < 1e-9
            d_hull = -b[2][0] if b[0] else 0.0
            assert d_hull <= depth + 1e-9                   # the polytope lies inside the sphere
            assert d_hull >= depth - (1.0 - inradius) * r - 1e-9
            e.append(depth - d_hull)
            if b[0]:
                assert np.arccos(np.clip(np.dot(b[3], u), -1, 1)) < (0.75, 0.45)[level]
        errs[level] = float(np.mean(e))
    assert errs[1] < 0.6 * errs[0], errs
    print(f"mean depth deficit vs the sphere: 12 vertices {errs[0] / r:.3f} r, 32 vertices {errs[1] / r:.3f} r")


def test_a_cube_given_as_its_corners_falls_and_rests_like_the_box_cube():
    """Scene level, oracle: the pick scene's cube as GEOM_BOX and as an 8-vertex GEOM_HULL, dropped tilted on the plane: the same
    contacts step by step, hence the same trajectory."""
    def scene(hull):
        sb = S.SceneBuilder()
        sb.add_geom(0, S.GEOM_PLANE)
        sb.add_body("cube", 0, pos=(0, 0, 0.2), jtype=S.JNT_FREE, mass=0.0128, inertia=S.box_inertia(0.0128, (0.02, 0.02, 0.02)))
        if hull:
            sb.add_geom("cube", S.GEOM_HULL, vertices=S.box_hull_vertices((0.02, 0.02, 0.02)))
        else:
            sb.add_geom("cube", S.GEOM_BOX, size=(0.02, 0.02, 0.02))
        sb.task = dict(eef_body=1, obj_body=1, grip_dof=(), reward_z=0.1)
        return sb.build()

    B = 8
    rng = np.random.default_rng(3)
    pos = (rng.uniform(-0.05, 0.05, (B, 3)) + [0, 0, 0.08]).astype(np.float32)
    quat = np.stack([_rand_quat(rng) for _ in range(B)]).astype(np.float32)
    arm = np.zeros((B, 0), np.float32)
    oa, ob = orc.Oracle(scene(False), B), orc.Oracle(scene(True), B)
    oa.reset(pos, quat, arm); ob.reset(pos, quat, arm)
    for t in range(150):
        oa.step_batch(None); ob.step_batch(None)
    qa, qb = oa.state()[0], ob.state()[0]
    assert np.abs(qa - qb).max() < 1e-9
    assert (qa[:, 2] < 0.03).all()        # at rest on the plane
