"""Child process of tests/test_sanitizers_cpu.py: runs with libasan preloaded and ORC_SANITIZE=1, so that every C / C++ line below
executes under AddressSanitizer + UndefinedBehaviorSanitizer (-fno-sanitize-recover: the first report aborts the process).

  * the float64 oracle (oracle/liborc64_asan.so): pick rollout with random actions, the scripted grasp (arm-cube contacts, manifold
    thinning at the 16-point capacity), SO-101 pick, both five-cube stack scenes, a vertex-hull scene, the ray-cast renderer, the IK;
  * the product's host-compiled code: both scene compilers and the spec emitter on every scene above (libmirhost_asan.so), the
    _mirfast module against stand-in entry points, the convex narrowphase header compiled for the host.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle"), HERE]
assert os.environ.get("ORC_SANITIZE") == "1"
import orc  # noqa: E402
from gym_genesis.backend import models  # noqa: E402
from gym_genesis.backend.spec import make_camera  # noqa: E402

HOME = np.array(models.FRANKA_HOME, dtype=np.float32)
done = []


def pick_rollout():
    spec = models.franka_cube_pick_scene().build()
    B = 8
    o = orc.Oracle(spec, B)
    assert "asan" in o.lib._name
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
    o.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(HOME, (B, 1)))
    acts = np.random.default_rng(1).uniform(-1, 1, (60, B, 9)).astype(np.float32)
    for t in range(60):
        o.step_batch(acts[t], 2)
    q, v = o.state()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    o.get_obs_all(); o.counts_all()
    return spec


def grasp():
    G_ = json.load(open(os.path.join(HERE, "golden", "grasp_targets.json")))
    T = np.array(G_["targets"], np.float32)
    pos = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    acts = np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0)
    spec = models.franka_cube_pick_scene().build()
    o = orc.Oracle(spec, 4)
    o.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (4, 1)), np.tile(HOME, (4, 1)))
    lifted = np.zeros(4, bool)
    for t in range(acts.shape[0]):
        o.step_batch(acts[t], 2)
        lifted |= o.get_obs()[3].astype(bool)
    assert lifted.all()


def so101_pick():
    spec = models.so101_cube_pick_scene().build()
    B = 4
    o = orc.Oracle(spec, B)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(-0.32, -0.28, B), rng.uniform(-0.05, 0.05, B), np.full(B, models.ISLAND_TOP_Z + 0.021)], 1).astype(np.float32)
    o.reset(pos, np.tile(np.array([1, 0, 0, 0], np.float32), (B, 1)), np.zeros((B, 6), np.float32))
    acts = np.random.default_rng(2).uniform(-1, 1, (60, B, 6)).astype(np.float32)
    for t in range(60):
        o.step_batch(acts[t], 2)
    assert np.isfinite(o.state()[0]).all()
    return spec


def stacks():
    specs = []
    z = models.STACK_CUBE_Z
    cubes = [(0.1, 0.0, z + 0.0405), (0.1, 0.0, z), (0.2, -0.15, z), (-0.2, -0.2, z), (0.05, 0.2, z)]
    for builder, home in ((models.franka_cube_stack_scene, np.asarray(models.FRANKA_HOME, np.float64)), (models.so101_cube_stack_scene, None)):
        spec = builder().build()
        o = orc.Oracle(spec, 1)
        n_arm = sum(1 for b in range(1, spec.nbody) if spec.body[b].jtype in (1, 2))
        arm = home if home is not None else np.zeros(n_arm)
        o.reset(np.asarray(cubes, np.float64)[None], np.tile([0, 0, 0, 1.0], (1, 5, 1)), arm[None])
        acts = np.random.default_rng(3).uniform(-0.5, 0.5, (80, 1, o.nu)).astype(np.float32)
        for t in range(80):
            o.step_batch(acts[t] + (arm[None, :o.nu].astype(np.float32) if home is not None else 0), 1)
        assert np.isfinite(o.state()[0]).all()
        o.get_obs()
        specs.append(spec)
    return specs


def hull_scene():
    """a free cube given as its 8 corners (MIR_GEOM_HULL) dropped tilted on the plane: support mappings, GJK / MPR, plane-hull contacts"""
    from gym_genesis.backend import spec as S
    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    sb.add_body("cube", 0, pos=(0, 0, 0.2), jtype=S.JNT_FREE, mass=0.0128, inertia=S.box_inertia(0.0128, (0.02, 0.02, 0.02)))
    sb.add_geom("cube", S.GEOM_HULL, vertices=S.box_hull_vertices((0.02, 0.02, 0.02)))
    sb.add_body("ball", 0, pos=(0.1, 0, 0.2), jtype=S.JNT_FREE, mass=0.05, inertia=(2e-5, 2e-5, 2e-5, 0, 0, 0))
    sb.add_geom("ball", S.GEOM_SPHERE, size=(0.03, 0, 0))
    sb.task = dict(eef_body=1, obj_body=1, grip_dof=(), reward_z=0.1)
    spec = sb.build()
    B = 4
    rng = np.random.default_rng(3)
    pos = np.zeros((B, 2, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.02, 0.02, (B, 3)) + [0, 0, 0.08]
    pos[:, 1] = pos[:, 0] + [0.0, 0.0, 0.06]       # the ball lands on the hull cube
    quat = rng.normal(size=(B, 2, 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=2, keepdims=True)
    o = orc.Oracle(spec, B)
    o.reset(pos, quat, np.zeros((B, 0), np.float32))
    for _ in range(120):
        o.step_batch(None, 2)
    assert np.isfinite(o.state()[0]).all()
    return spec


def render_and_ik(spec):
    o = orc.Oracle(spec, 2)
    o.fk(0); o.fk(1)
    xpos = np.stack([o.read(orc.F_XPOS, e).reshape(-1, 3) for e in range(2)])
    xquat = np.stack([o.read(orc.F_XQUAT, e).reshape(-1, 4) for e in range(2)])
    cam = make_camera(64, 48, (3.5, 0.0, 2.5), (0.0, 0.0, 0.5), 30.0)
    vis = models.franka_cube_pick_scene().visual()
    img = orc.render_image(spec, cam, vis, xpos[:1], xquat[:1])
    assert img.shape == (48, 64, 3)
    tp = np.array([[0.55, 0.0, 0.3], [0.5, 0.1, 0.25]])
    tq = np.tile([0.0, 1.0, 0.0, 0.0], (2, 1))
    q, err = o.ik(spec.task.eef_body, tp, tq, np.tile(np.asarray(models.FRANKA_HOME, np.float64), (2, 1)))
    assert np.isfinite(q).all()


def host_compilers(specs):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gym-genesis_amd", "csrc"), "asan-host"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, "_build", "libmirhost_asan.so"))
    L.asan_compile_spec.restype = C.c_int
    rcs = []
    for spec in specs:
        e16, e64 = C.create_string_buffer(256), C.create_string_buffer(256)
        r = L.asan_compile_spec(C.byref(spec), e16, e64)
        rcs.append((r & 255, r >> 8 & 255, r >> 16 & 1))
    assert rcs[0] == (0, 0, 1) or rcs[0][0] == 0, rcs       # the pick scene compiles for the 16-lane kernel and emits its literals
    assert all(a == 0 or b == 0 for a, b, _ in rcs), rcs    # every scene compiles for one of the two kernels
    assert rcs[-1][1] == 0, rcs                               # the hull scene compiles for the wave kernel too (round 4)
    # a spec of garbage sizes must be refused, not walked over
    bad = type(specs[0]).from_buffer_copy(bytes(specs[0]))
    bad.nbody = 10 ** 6
    e16, e64 = C.create_string_buffer(256), C.create_string_buffer(256)
    r = L.asan_compile_spec(C.byref(bad), e16, e64)
    assert (r & 255) != 0 and (r >> 8 & 255) != 0
    # _mirfast against stand-in entry points
    sys.path.insert(0, os.path.join(HERE, "_build", "_mirfast_asan"))
    import _mirfast
    assert "_mirfast_asan" in _mirfast.__file__
    addr = lambda f: C.cast(f, C.c_void_p).value  # noqa: E731
    _mirfast.bind(addr(L.asan_stub_prepare), addr(L.asan_stub_go), addr(L.asan_stub_end))
    buf = np.zeros(16, np.float32); host = np.zeros(8, np.uint8)
    p = buf.ctypes.data
    assert _mirfast.prepare(1, p, p, p, host.ctypes.data) == 0 and _mirfast.go(1, p, None) == 0 and _mirfast.end(1, host.ctypes.data) == 0
    assert _mirfast.end(1, None) == 0 and host.all()
    for badargs in ((), (1, 2), ("x", p, p, p, p)):
        try:
            _mirfast.prepare(*badargs)
            raise SystemExit("prepare accepted bad arguments")
        except (TypeError, ValueError):
            pass
    return rcs


def convex_host():
    so = os.path.join(HERE, "_build", "libconvex_host_asan.so")
    srcs = [os.path.join(HERE, "convex_host.cpp"), os.path.join(ROOT, "gym-genesis_amd", "csrc", "mir_convex.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", srcs[0], "-o", so])
    import test_convex_host as T
    L = C.CDLL(so)
    for rows in (T.random_pairs(1500, 7), T.analytic_deep_pairs()[0] if isinstance(T.analytic_deep_pairs(), tuple) else T.random_pairs(200, 9)):
        rows = np.ascontiguousarray(rows, np.float32)
        got = np.zeros((rows.shape[0], 8), np.float32)
        L.convex_host_pairs(rows.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), rows.shape[0])
        assert np.isfinite(got).all()
    return 1700


pick = pick_rollout(); done.append("pick")
grasp(); done.append("grasp")
so = so101_pick(); done.append("so101")
st = stacks(); done.append("stack")
hs = hull_scene(); done.append("hull" if hs is not None else "hull(skipped)")
render_and_ik(pick); done.append("render+ik")
rcs = host_compilers([pick, so] + st + ([hs] if hs is not None else [])); done.append("compilers+mirfast")
n = convex_host(); done.append(f"convex_host({n})")
print("SANITIZER_WORKER_OK", " ".join(done), rcs)
