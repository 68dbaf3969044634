"""GPU parity of the tiled rasteriser (mir_render, through the C ABI) against the float64 brute-force ray
caster of the oracle (oracle/orc_render.c) on the same link poses.

Bar: uint8 RGB, every pixel within 1 LSB of the oracle, except pixels where the two disagree on WHICH surface is
nearest (silhouette / checker-edge pixels whose ray passes within float32 rounding of an edge); those are counted
and must stay below 0.05 % of the image.
"""
import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models
from gym_genesis.backend.spec import make_camera

pytestmark = pytest.mark.gpu

HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _stepped_scene(builder, B, steps=30, seed=0):
    from gym_genesis.backend.lib import MirScene

    sc = MirScene(builder.build(), B)
    rng = np.random.RandomState(seed)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(HOME, (B, 1)))
    acts = torch.as_tensor(rng.uniform(-1, 1, (steps, B, 9)).astype(np.float32), device=sc.device)
    for t in range(steps):
        sc.set_pd_targets(acts[t])
        sc.step(1)
    return sc


def _compare(img, ref, max_bad_frac=5e-4):
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16)).max(axis=-1)
    bad = d > 1
    assert bad.mean() <= max_bad_frac, f"{bad.sum()} pixels differ by more than 1 LSB ({bad.mean():.2e} of the image)"
    return int(bad.sum()), int(d.max())


@pytest.mark.parametrize("res", [(640, 480), (128, 96)])
def test_per_env_images_match_oracle(res):
    B = 6
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B)
    cam = make_camera(res[0], res[1], (3.5, 0.0, 2.5), (0, 0, 0.5), 30)
    vis = builder.visual()
    img = sc.render(cam, vis, mode=0).cpu().numpy()
    assert img.shape == (B, res[1], res[0], 3) and img.dtype == np.uint8
    xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
    total_bad = 0
    for e in range(B):
        ref = orc.render_image(builder.build(), cam, vis, xpos[e:e + 1], xquat[e:e + 1])
        nbad, _ = _compare(img[e], ref)
        total_bad += nbad
        assert len(np.unique(ref.reshape(-1, 3), axis=0)) > 6  # robot, cube, both checker colours are in view
    # different envs show different images (cube spawn + random actions)
    assert not np.array_equal(img[0], img[1])


def test_close_camera_and_odd_resolution():
    """Primitives that straddle the camera plane / fill the screen, W not a multiple of 4 (scalar store path),
    H not a multiple of the tile height."""
    B = 3
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=5)
    vis = builder.visual()
    for cam in (make_camera(202, 77, (0.9, 0.3, 0.6), (0.3, 0.0, 0.4), 70), make_camera(64, 64, (0.3, 0.0, 0.05), (0.65, 0.0, 0.0), 100),
                make_camera(96, 40, (0.0, 0.0, 3.0), (0.4, 0.0, 0.0), 45, up=(1.0, 0.0, 0.0))):
        img = sc.render(cam, vis, mode=0).cpu().numpy()
        xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
        for e in range(B):
            ref = orc.render_image(builder.build(), cam, vis, xpos[e:e + 1], xquat[e:e + 1])
            _compare(img[e], ref, max_bad_frac=2e-3)


def test_global_image_matches_oracle():
    """One image of all envs at their grid offsets (camera_capture_mode='global', cube_pick.py:174-176); B large
    enough that a tile's primitive list needs several LDS rounds."""
    B = 400
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=3)
    cam = make_camera(320, 240, (14.0, -3.0, 9.0), (0, 0, 0.5), 40)
    vis = builder.visual()
    side = int(np.ceil(np.sqrt(B)))
    idx = np.arange(B)
    off = np.stack([(idx % side - (side - 1) / 2) * 1.0, (idx // side - (side - 1) / 2) * 1.0, np.zeros(B)], 1).astype(np.float32)
    img = sc.render(cam, vis, mode=1, env_offset=torch.as_tensor(off, device=sc.device)).cpu().numpy()
    assert img.shape == (240, 320, 3)
    xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
    ref = orc.render_image(builder.build(), cam, vis, xpos, xquat, offsets=off)
    _compare(img, ref, max_bad_frac=3e-3)  # hundreds of small silhouettes: more edge pixels per image


def _single_env_image(builder, qpos_row, cam, vis):
    """The image of ONE env holding `qpos_row`, rendered by a scene of its own."""
    from gym_genesis.backend.lib import MirScene

    one = MirScene(builder.build(), 1)
    one.set_state(qpos=qpos_row[None])
    return one.render(cam, vis, mode=0)[0]


def test_cfg5_real_size_1024_envs_480x640():
    """BASELINE configs[4] at its real size -- 1024 per-env images of 480 x 640 x 3 (943.7 MB per render): the first, a middle and
    the last image equal the render of a single-env scene holding that env's state bit for bit (no image sees another env's
    primitives, tiles or store offsets), and the middle one is within 1 LSB of the float64 ray caster."""
    B = 1024
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=12)
    cam = make_camera(640, 480, (3.5, 0.0, 2.5), (0, 0, 0.5), 30)   # cube_pick.py:56-63
    vis = builder.visual()
    img = sc.render(cam, vis, mode=0)
    assert img.shape == (B, 480, 640, 3) and img.dtype == torch.uint8 and img.numel() == 943718400
    q = sc.get_state()[0]
    for e in (0, 511, 1023):
        assert torch.equal(img[e], _single_env_image(builder, q[e], cam, vis)), f"env {e}"
    xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
    ref = orc.render_image(builder.build(), cam, vis, xpos[511:512], xquat[511:512])
    _compare(img[511].cpu().numpy(), ref)
    assert not torch.equal(img[0], img[1023])


def test_more_than_2_31_bytes_of_pixels_and_the_addressing_limits():
    """2400 x 480 x 640 x 3 = 3.3 GB in one call: the images on either side of the 2^31- and 2^32-byte marks (envs 2330 and -- past
    the end here -- none for 2^32, so also 2399, the last) equal their single-env renders; an image of 2^32 bytes or more is
    refused with MIR_E_CAPACITY before anything is launched."""
    import ctypes as C

    B = 2400
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=3)
    cam = make_camera(640, 480, (3.5, 0.0, 2.5), (0, 0, 0.5), 30)
    vis = builder.visual()
    img = sc.render(cam, vis, mode=0)
    assert img.numel() > 2 ** 31
    q = sc.get_state()[0]
    for e in (2329, 2330, 2331, 2399):
        assert torch.equal(img[e], _single_env_image(builder, q[e], cam, vis)), f"env {e}"
    del img
    torch.cuda.empty_cache()
    big = make_camera(40000, 36000, (3.5, 0.0, 2.5), (0, 0, 0.5), 30)   # 4.3e9 bytes per image
    dummy = torch.empty(16, dtype=torch.uint8, device=sc.device)
    rc = sc.lib.mir_render(sc.h, C.byref(big), C.byref(vis), 0, None, C.c_void_p(dummy.data_ptr()), sc._stream())
    assert rc == -2 and b"2^32" in sc.lib.mir_last_error()   # MIR_E_CAPACITY


def test_render_is_deterministic_and_does_not_disturb_physics():
    B = 8
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=4)
    q0, v0, _, _ = (t.clone() for t in sc.get_state())
    cam = make_camera(160, 120, (3.5, 0.0, 2.5), (0, 0, 0.5), 30)
    vis = builder.visual()
    a = sc.render(cam, vis).clone()
    b = sc.render(cam, vis)
    assert torch.equal(a, b)
    q1, v1, _, _ = sc.get_state()
    assert torch.equal(q0, q1) and torch.equal(v0, v1)


def test_pixels_observation_through_the_env():
    """GenesisEnv(enable_pixels=True): obs keys / shapes / dtypes of the reference (cube_pick.py:73-84,159-180)."""
    from gym_genesis.env import GenesisEnv

    B, H, W = 5, 96, 128
    for mode, shape in (("per_env", (B, H, W, 3)), ("global", (H, W, 3))):
        env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H,
                         observation_width=W, camera_capture_mode=mode)
        obs, _ = env.reset(seed=0)
        assert set(obs) == {"agent_pos", "pixels"}  # strip_environment_state=True (env.py:28)
        assert tuple(obs["pixels"].shape) == shape and obs["pixels"].dtype == torch.uint8
        obs, reward, terminated, truncated, info = env.step(np.zeros((B, 9), np.float32))
        assert tuple(obs["pixels"].shape) == shape
        frame = env.render()
        assert isinstance(frame, np.ndarray) and frame.shape == (H, W, 3) and frame.dtype == np.uint8
        assert env.observation_space["pixels"].shape == (H, W, 3)
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=2, enable_pixels=False)
    assert env.render() is None
    with pytest.raises(ValueError):
        env.get_cams()


def test_per_env_cameras_match_oracle():
    """mir_render_cams: every env seen from its own camera (the link-mounted wrist cameras of the stack tasks), including
    a top-down camera whose view is parallel to the +z up hint (falls back to +y)."""
    B = 4
    builder = models.franka_cube_stack_scene()
    from gym_genesis.backend.lib import MirScene

    sc = MirScene(builder.build(), B)
    rng = np.random.RandomState(0)
    pos = np.zeros((B, 5, 3), np.float32)
    pos[:, :, :2] = rng.uniform(-0.3, 0.3, (B, 5, 2))
    pos[:, :, 2] = models.STACK_CUBE_Z
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 5, 1)), np.tile(np.asarray(models.FRANKA_HOME, np.float32), (B, 1)))
    sc.step(3)
    vis = builder.visual()
    cam = make_camera(160, 120, (0, 0, 0), (1, 0, 0), 70)
    cp = np.array([[0.6, 0.3 * e - 0.4, 1.3] for e in range(B)], np.float32)
    cl = np.array([[-0.1, 0.0, 0.75]] * B, np.float32)
    cl[0] = cp[0] - [0, 0, 1.0]  # straight down: parallel to up = +z
    cu = None
    img = sc.render_cams(cam, vis, cp, cl, cu).cpu().numpy()
    xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
    for e in range(B):
        ce = make_camera(160, 120, cp[e], cl[e], 70)
        ref = orc.render_image(builder.build(), ce, vis, xpos[e:e + 1], xquat[e:e + 1])
        _compare(img[e], ref, max_bad_frac=2e-3)
    # explicit per-env up vectors
    cu = np.tile(np.array([[0.0, 1.0, 0.2]], np.float32), (B, 1))
    img = sc.render_cams(cam, vis, cp, cl, cu).cpu().numpy()
    for e in range(B):
        ce = make_camera(160, 120, cp[e], cl[e], 70, up=cu[e])
        ref = orc.render_image(builder.build(), ce, vis, xpos[e:e + 1], xquat[e:e + 1])
        _compare(img[e], ref, max_bad_frac=2e-3)


@pytest.mark.parametrize("robot", ["franka", "so101"])
def test_stack_task_pixels(robot):
    from gym_genesis.env import GenesisEnv

    B, H, W = 3, 60, 80
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B, enable_pixels=True, observation_height=H, observation_width=W,
                     camera_capture_mode="per_env")
    obs, _ = env.reset(seed=0)
    px = obs["pixels"]
    assert set(px) == {"top", "side", "wrist"}
    assert tuple(px["top"].shape) == (B, H, W, 3) and tuple(px["wrist"].shape) == (B, 480, 640, 3) and px["side"].is_cuda
    # the top view shows the slab and the five cubes: at least five distinct cube colours present
    cols = {tuple(c) for c in px["top"][0].reshape(-1, 3).cpu().numpy()[::7]}
    assert len(cols) >= 5
    obs, *_ = env.step(np.tile(np.asarray(env._env._home_qpos(), np.float32), (B, 1)))
    assert tuple(obs["pixels"]["side"].shape) == (B, H, W, 3)



def test_binned_kernel_equals_generic_kernel_bit_for_bit():
    """Per-env images take the binned pixel kernel (per-strip lists built once per render, floor pass without a depth buffer); the
    generic kernel (per-workgroup culling) draws the same images bit for bit -- pick scene, stack scene (plane + slab + cubes),
    rolled and close cameras (the floor pass does not apply: general plane path), every strip height."""
    from gym_genesis.backend.lib import MirScene

    B = 5
    pick = models.franka_cube_pick_scene()
    sc = _stepped_scene(pick, B, steps=12)
    cams = [make_camera(640, 480, (3.5, 0.0, 2.5), (0, 0, 0.5), 30), make_camera(128, 96, (3.5, 0.0, 2.5), (0, 0, 0.5), 30),
            make_camera(200, 77, (0.9, 0.3, 0.6), (0.3, 0.0, 0.4), 70), make_camera(96, 40, (0.0, 0.0, 3.0), (0.4, 0.0, 0.0), 45, up=(1.0, 0.0, 0.0)),
            make_camera(64, 64, (0.3, 0.0, 0.05), (0.65, 0.0, 0.0), 100), make_camera(320, 200, (1.5, 1.0, 1.0), (0.4, 0.0, 0.2), 50, up=(0.3, 0.0, 1.0))]
    scenes = [(sc, pick.visual())]
    b = models.franka_cube_stack_scene()
    st = MirScene(b.build(), B)
    rng = np.random.RandomState(3)
    pos = np.zeros((B, 5, 3), np.float32)
    pos[:, :, 0] = np.array([-0.3, -0.15, 0.0, 0.15, 0.3]) + rng.uniform(-0.03, 0.03, (B, 5))
    pos[:, :, 1] = rng.uniform(-0.2, 0.2, (B, 5))
    pos[:, :, 2] = models.STACK_CUBE_Z
    st.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 5, 1)), np.tile(HOME, (B, 1)))
    scenes.append((st, b.visual()))
    stack_cam = make_camera(256, 160, (1.2, 0.0, 1.6), (-0.2, 0.0, 0.75), 50)
    for scene, vis in scenes:
        for cam in cams + [stack_cam]:
            scene.debug_render_path(generic=True)
            ref = scene.render(cam, vis, mode=0).cpu().numpy()
            for rows in (0, 32, 160, 480):
                scene.debug_render_path(generic=False, strip_rows=rows)
                img = scene.render(cam, vis, mode=0).cpu().numpy()
                assert np.array_equal(img, ref), f"binned kernel differs from the generic one ({cam.width}x{cam.height}, strip rows {rows})"
            scene.debug_render_path()
            assert len(np.unique(ref.reshape(-1, 3), axis=0)) >= 2


@pytest.mark.parametrize("kind", ["pick", "stack"])
def test_render_behind_a_step_needs_no_pose_refresh_and_shows_the_same_image(kind):
    """Once a render has been asked for, every integrating launch leaves the link poses of its final state for the rasteriser
    (launch() in mir_api.hip) and mir_render skips its pose-refresh launch.  The images must be those of a twin that is given the
    same state through mir_set_state (which invalidates the poses: its render refreshes them) -- bit for bit, after fused steps,
    split / rotated steps, a K-step rollout, an in-kernel auto-reset, a masked reset and a state write."""
    from gym_genesis.backend.lib import MirScene

    B = 64
    rng = np.random.RandomState(5)
    if kind == "pick":
        builder = models.franka_cube_pick_scene()
        pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
        quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
        home = HOME
        cam = make_camera(256, 192, (3.5, 0.0, 2.5), (0, 0, 0.5), 30)
    else:
        builder = models.franka_cube_stack_scene()
        pos = np.zeros((B, 5, 3), np.float32)
        pos[:, :, 0] = np.array([-0.3, -0.15, 0.0, 0.15, 0.3]) + rng.uniform(-0.03, 0.03, (B, 5))
        pos[:, :, 1] = rng.uniform(-0.2, 0.2, (B, 5))
        pos[:, :, 2] = models.STACK_CUBE_Z
        quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 5, 1))
        home = HOME
        cam = make_camera(256, 160, (1.2, 0.0, 1.6), (-0.2, 0.0, 0.75), 50)
    vis = builder.visual()
    a, b = MirScene(builder.build(), B), MirScene(builder.build(), B)
    for sc in (a, b):
        sc.reset(pos, quat, np.tile(home, (B, 1)))
    acts = torch.as_tensor(rng.uniform(-1, 1, (40, B, a.nu)).astype(np.float32), device=a.device)
    bufs = (a.empty(a.agent_dim), a.empty(a.env_dim), a.empty(), a.empty(dtype=torch.uint8))

    def same(tag):
        b.set_state(*[x.clone() for x in a.get_state()])
        ia, ib = a.render(cam, vis, mode=0).cpu().numpy(), b.render(cam, vis, mode=0).cpu().numpy()
        assert np.array_equal(ia, ib), f"{tag}: the image rendered from the step's own poses differs from a refreshed render"
        return ia

    first = same("first render")                     # (turns the hand-over on for `a`)
    for t in range(5):
        a.step_fused(acts[t], *bufs)
    img = same("fused steps")
    assert not np.array_equal(img, first)
    for t in range(5, 12):                           # split / rotated launches (16-lane kernel), plain ones on the wave kernel
        a.step_begin(acts[t], *bufs)
        a.step_end()
    same("step_begin / step_end")
    a.set_pd_targets(acts[12])
    a.step(3)
    same("mir_step(3)")
    mask = np.zeros(B, np.uint8); mask[::3] = 1
    a.reset(pos, quat, np.tile(home, (B, 1)), env_mask=mask)
    same("masked reset (no step behind it)")
    for t in range(21, 24):
        a.step_fused(acts[t], *bufs)
    same("steps after the masked reset")


def test_stack_scene_views_match_oracle():
    """The five-cube scene on the kitchen slab, seen from cameras that stand INSIDE one or two of the slab's (and the cubes') slabs --
    straight above the footprint, level with the slab's side, close over a cube -- against the oracle's ray caster.  These are the box
    records with one and two upper bounds (camera outside one / two slabs only: mir_render.hip, k_render_setup)."""
    from gym_genesis.backend.lib import MirScene

    B = 4
    b = models.franka_cube_stack_scene()
    spec = b.build()
    sc = MirScene(spec, B)
    rng = np.random.RandomState(11)
    pos = np.zeros((B, 5, 3), np.float32)
    pos[:, :, 0] = np.array([-0.3, -0.15, 0.0, 0.15, 0.3]) + rng.uniform(-0.03, 0.03, (B, 5))
    pos[:, :, 1] = rng.uniform(-0.2, 0.2, (B, 5))
    pos[:, :, 2] = models.STACK_CUBE_Z
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 5, 1)), np.tile(HOME, (B, 1)))
    acts = torch.as_tensor(HOME + rng.uniform(-0.5, 0.5, (10, B, 9)).astype(np.float32), device=sc.device)
    for t in range(10):
        sc.set_pd_targets(acts[t])
        sc.step(1)
    vis = b.visual()
    xpos, xquat = (t.cpu().numpy() for t in sc.get_links())
    z = float(models.STACK_CUBE_Z)
    cams = [make_camera(256, 160, (0.0, 0.0, z + 1.8), (0.0, 0.0, z), 50, up=(1.0, 0.0, 0.0)),   # straight above: inside the x and y slabs of the slab
            make_camera(200, 120, (1.6, 0.05, z - 0.03), (0.0, 0.0, z), 40),                        # level with the slab's top: inside its z slab (and y)
            make_camera(160, 120, (0.02, 0.01, z + 0.25), (0.0, 0.0, z), 80, up=(1.0, 0.0, 0.0)),   # close over the middle cube
            make_camera(320, 200, (1.2, 0.0, 1.6), (-0.2, 0.0, 0.75), 50)]                          # the task's own side view
    for cam in cams:
        img = sc.render(cam, vis, mode=0).cpu().numpy()
        for e in range(B):
            ref = orc.render_image(spec, cam, vis, xpos[e:e + 1], xquat[e:e + 1])
            _compare(img[e], ref)
        assert len(np.unique(img.reshape(-1, 3), axis=0)) > 4


def test_global_view_of_many_envs_equals_the_tiled_kernel():
    """The global view of more than 512 primitives is drawn box by box into a depth / colour buffer and resolved in a second pass
    (k_global_splat / k_global_resolve: 17.8 ms -> a fraction of a millisecond at 4096 envs); the tiled kernel stays for small scenes
    and as the reference here: same expressions per pixel, so the images agree except where two surfaces tie in depth to the last bit."""
    B = 400
    builder = models.franka_cube_pick_scene()
    sc = _stepped_scene(builder, B, steps=3)
    vis = builder.visual()
    side = int(np.ceil(np.sqrt(B)))
    idx = np.arange(B)
    off = torch.as_tensor(np.stack([(idx % side - (side - 1) / 2) * 1.0, (idx // side - (side - 1) / 2) * 1.0, np.zeros(B)], 1).astype(np.float32), device=sc.device)
    # far, farther, INSIDE the grid (boxes behind and across the camera plane: most are dropped by the setup kernel's view-pyramid test,
    # the rest listed as work items), a close-up (one link fills much of the screen: a box in 32 shares), and the first one again -- every
    # render finds the depth buffer and the work list as the one before left them (k_global_resolve hands them back clean, whatever the
    # resolution), and drawing twice gives the same image
    cams = (make_camera(320, 240, (14.0, -3.0, 9.0), (0, 0, 0.5), 40), make_camera(640, 480, (0.0, -18.0, 12.0), (0, 0, 0.0), 50),
            make_camera(202, 99, (3.0, 2.0, 1.5), (0, 0, 0.3), 70), make_camera(640, 480, (0.45, 0.35, 0.5), (0.0, 0.0, 0.35), 60),
            make_camera(320, 240, (14.0, -3.0, 9.0), (0, 0, 0.5), 40))
    for cam in cams:
        sc.debug_render_path(generic=True)
        ref = sc.render(cam, vis, mode=1, env_offset=off).cpu().numpy()
        sc.debug_render_path()
        img = sc.render(cam, vis, mode=1, env_offset=off).cpu().numpy()
        again = sc.render(cam, vis, mode=1, env_offset=off).cpu().numpy()
        assert np.array_equal(img, again)
        diff = (img != ref).any(axis=-1)
        assert diff.mean() <= 1e-4, f"{diff.sum()} pixels differ between the two global paths ({cam.width}x{cam.height})"
        assert len(np.unique(img.reshape(-1, 3), axis=0)) > 6


def test_camera_rolled_by_180_degrees_draws_the_rotated_image():
    """The SO-101 stack task's wrist camera image is rotated by 180 degrees in the reference (np.rot90(img, k=2)); here the camera is
    rolled by 180 degrees about its view axis instead (up -> -up): the same image without a second pass over 236 MB."""
    from gym_genesis.backend.lib import MirScene

    B = 6
    b = models.so101_cube_stack_scene()
    sc = MirScene(b.build(), B)
    rng = np.random.RandomState(2)
    st = [x.clone() for x in sc.get_state()]
    sc.set_state(*st)
    for _ in range(3):
        sc.step(1)
    cam = make_camera(640, 480, (0.4, 0.0, 0.7), (0.0, 0.0, 1.0), 70)
    vis = b.visual()
    pos = torch.as_tensor(rng.uniform(-0.3, 0.3, (B, 3)).astype(np.float32) + np.array([0.0, 0.0, 1.1], np.float32), device=sc.device)
    look = torch.as_tensor(rng.uniform(-0.2, 0.2, (B, 3)).astype(np.float32) + np.array([-0.2, 0.0, 0.75], np.float32), device=sc.device)
    up = torch.as_tensor(np.tile(np.array([0.0, 0.3, 1.0], np.float32), (B, 1)), device=sc.device)
    a = sc.render_cams(cam, vis, pos, look, up)
    r = sc.render_cams(cam, vis, pos, look, -up)
    diff = (r != torch.flip(a, dims=(1, 2))).any(dim=-1).float().mean().item()
    assert diff <= 1e-4, f"{diff:.2e} of the pixels differ between the rolled camera and the flipped image"
    assert len(torch.unique(a.reshape(-1, 3), dim=0)) > 4


def test_render_reuses_the_observation_image_only_while_nothing_has_moved():
    """GenesisEnv.render() behind an env.step() with global pixels copies out the image the observation holds (mir_get_state_version
    says nothing moved); after a step, a reset, a state write or a camera move it draws again."""
    from gym_genesis.env import GenesisEnv

    B = 64
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=120, observation_width=160,
                     camera_capture_mode="global")
    obs, _ = env.reset(seed=0)
    mir, cam = env._env._mir, env._env.cam
    v0 = mir.state_version
    a = env.render()
    assert np.array_equal(a, obs["pixels"].cpu().numpy()) and mir.state_version == v0
    act = np.random.default_rng(0).uniform(-1, 1, (B, 9)).astype(np.float32)
    for _ in range(30):
        obs, *_ = env.step(act)
    assert mir.state_version > v0
    b = env.render()
    assert np.array_equal(b, obs["pixels"].cpu().numpy()) and not np.array_equal(a, b)
    b0 = b.copy()
    b[:] = 0                                           # the caller's array is its own
    assert np.array_equal(env.render(), obs["pixels"].cpu().numpy())
    obs["pixels"].div_(2, rounding_mode="floor")       # an in-place edit of the observation: render() draws the scene again
    assert np.array_equal(env.render(), b0) and not np.array_equal(b0, obs["pixels"].cpu().numpy())
    st = [x.clone() for x in mir.get_state()]
    st[0][:, :7] += 0.3                                # a state write: the image must change
    mir.set_state(*st)
    c = env.render()
    assert not np.array_equal(c, obs["pixels"].cpu().numpy())
    cam.set_pose(pos=(5.0, 1.0, 3.0))                  # a camera move: drawn again from the new pose
    d = env.render()
    assert not np.array_equal(c, d)
    env.reset()
    assert not np.array_equal(env.render(), d)


def test_recording_from_reset_to_save_video(tmp_path):
    """cam.start_recording() at reset, a frame per global render (the `global` pixels observation of every step, env.render()),
    env.save_video -> an mp4 of Motion-JPEG frames (tasks/video.py) that decodes to the observations (env.py:71-79, cube_pick.py:109-110);
    nothing is copied while recording: the frames are the observation tensors."""
    import warnings
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks.video import read_mjpeg_mp4

    B = 64
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=120, observation_width=160,
                     camera_capture_mode="global", record_video=True)
    obs, _ = env.reset(seed=0)
    want = [obs["pixels"]]
    act = np.random.default_rng(0).uniform(-1, 1, (B, 9)).astype(np.float32)
    for _ in range(5):
        obs, *_ = env.step(act)
        want.append(obs["pixels"])
    want.append(torch.from_numpy(env.render()))          # (the reused image: a frame of its own, as in Genesis)
    cam = env._env.cam
    assert len(cam._frames) == 7 and all(f.data_ptr() == w.data_ptr() for (f, _), w in zip(cam._frames[:6], want))
    path = str(tmp_path / "episode.mp4")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        env.save_video(save_video=True, file_name=path, fps=60)
    assert len(rec) == 1 and "stops the camera recording" in str(rec[0].message)
    frames, fps, wh = read_mjpeg_mp4(path)
    assert len(frames) == 7 and fps == 60.0 and wh == (160, 120)
    for got, w in zip(frames, want):
        assert np.abs(got.astype(int) - w.cpu().numpy().astype(int)).mean() < 4.0
    # the cap: only the newest frames are kept
    env.reset()
    cam.max_recorded_frames = 3
    for _ in range(5):
        env.step(act)
    assert len(cam._frames) == 3 and cam._frames_dropped == 3
    with pytest.warns(UserWarning, match="older ones were dropped"):
        cam.stop_recording(save_to_filename=str(tmp_path / "tail.mp4"), fps=30)
    assert len(read_mjpeg_mp4(str(tmp_path / "tail.mp4"))[0]) == 3
    # the three-camera stack tasks: save_video writes the top view (the reference has no `cam` there and raises AttributeError)
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=4, enable_pixels=True, observation_height=96, observation_width=128,
                     camera_capture_mode="global", record_video=True)
    env.reset(seed=0)
    env.step(np.zeros((4, 9), np.float32))
    with pytest.warns(UserWarning):
        env.save_video(save_video=True, file_name=str(tmp_path / "stack.mp4"))
    frames, _, wh = read_mjpeg_mp4(str(tmp_path / "stack.mp4"))
    assert len(frames) == 2 and wh == (128, 96)


def test_render_arrays_are_lent_pinned_buffers_that_are_never_reused_while_held():
    """env.render() returns the pinned buffer the device copied into (no host memcpy); the camera takes a buffer back only when its
    array has been garbage-collected, and a caller that holds more than the camera lends gets ordinary copies: whatever is held stays
    as it was."""
    import gc
    from gym_genesis.env import GenesisEnv

    B = 64
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=120, observation_width=160,
                     camera_capture_mode="global")
    env.reset(seed=0)
    cam = env._env.cam
    act = np.random.default_rng(0).uniform(-1, 1, (B, 9)).astype(np.float32)
    held, want = [], []
    for k in range(12):                                   # more than _PIN_MAX = 8 frames held at once
        obs, *_ = env.step(act)
        held.append(env.render())
        want.append(obs["pixels"].cpu().numpy().copy())
    assert cam._pin_out == cam._PIN_MAX and len({h.ctypes.data for h in held}) == 12
    for _ in range(5):                                    # later renders touch none of them
        env.step(act); env.render()
    assert all(np.array_equal(h, w) for h, w in zip(held, want))
    view = held[0][10:20]                                 # a view keeps its buffer on loan
    first = held[0].ctypes.data
    del held[0]
    gc.collect()
    assert cam._pin_out == cam._PIN_MAX and np.array_equal(view, want[0][10:20])
    env.step(act)
    x = env.render()
    assert x.ctypes.data != first and np.array_equal(view, want[0][10:20])
    del view, held, x
    gc.collect()
    assert cam._pin_out == 0 and len(cam._pin_pool) == cam._PIN_MAX
    env.step(act)
    y = env.render()                                      # a returned buffer is lent again
    assert cam._pin_out == 1 and len(cam._pin_pool) == cam._PIN_MAX - 1
