#!/usr/bin/env python3
"""bench.py — env-steps/sec of the CubePick-v0 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] / BASELINE.md §3): GenesisEnv(task="cube_pick", robot="franka",
num_envs=4096 per GPU, enable_pixels=False); reset(seed=0); a fresh a_t ~ U(-1,1)^(B x 9) float32
per step (generated on the device BEFORE the timed region: inputs are resident in HBM); one
fused hot-path launch per step (control + scene.step + reward + observations); reset-all every
200 steps (TimeLimit parity).  Env axis sharded across ranks (weak scaling: 4096 envs per GPU);
with N > 1 the packed observation/reward rows are all-gathered over RCCL in rollout chunks of
--gather-every steps (default 8; 1 = every step), double-buffered so a chunk's gather overlaps the
next chunk's physics.

One JSON line on rank 0: value = total env-steps / wall time (max over ranks) of exactly K steps.
Extra objects: "roofline" (HBM; algorithmic 489 B per env-step), "cpu_baseline" (the float32 CPU port
of the oracle, OpenMP over all host cores; rank 0, N=1 only) and "pixels" (secondary: BASELINE configs[4], the tiled
rasteriser at 1024 x 480 x 640, HBM-write roofline; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENVS_PER_GPU = 4096
EPISODE_STEPS = 200
ALGO_BYTES_PER_ENV_STEP = 489.0  # 55 f32 read + 67 f32 + 1 B written
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ROW_STRIDE = 24


def cpu_baseline(budget_s: float = 12.0):
    """Time the float32 CPU port of the oracle (oracle/liborc32.so) on all host cores, same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from gym_genesis.backend import models

    try:
        cores = len(os.sched_getaffinity(0))  # cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    B = ENVS_PER_GPU
    spec = models.franka_cube_pick_scene().build()
    o = orc.Oracle(spec, B, f32=True)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    o.reset(pos, np.tile(np.array([0, 0, 0, 1.0], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
    acts = np.random.default_rng(1234).uniform(-1, 1, (64, B, 9)).astype(np.float32)
    # the host may expose more cores than its CPU quota sustains: calibrate the OpenMP team size
    best, used = None, cores
    for nt in sorted({n for n in (4, 8, 16, 32, 64, 128, cores) if n <= cores}):
        o.step_batch(acts[0], nt)  # thread-pool / page-fault warm-up
        t0 = time.perf_counter()
        o.step_batch(acts[1], nt)
        o.step_batch(acts[2], nt)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, used = dt, nt
    cores = used
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and steps < 2000:
        o.step_batch(acts[steps % 64], cores)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps * B / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x {B} envs, same random-action workload, float32 C port of the oracle, OpenMP over envs"}


def pixels_bench(dev, renders: int = 30):
    """BASELINE.json configs[4]: CubePick-v0, 1024 envs, enable_pixels=True, per-env 480x640 RGB8 images rendered by
    the tiled HIP rasteriser (mir_render) from the state resident in HBM; one step + one render per iteration is NOT
    what is timed here -- only the render launches (setup + pixel kernels), with HIP events on the launching stream.
    Algorithmic bytes = B*H*W*3 written once (cfg 5: 943 MB/step, HBM-write-bound)."""
    from gym_genesis.env import GenesisEnv

    B, H, W = 1024, 480, 640
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W,
                     camera_capture_mode="per_env")
    task = env._env
    env.reset(seed=0)
    gen = torch.Generator(device=dev).manual_seed(99)
    for _ in range(20):  # move the arms apart so the images differ
        task.step_raw(torch.empty((B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen))
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    for _ in range(3):
        task.cam.render_envs(out=out)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(renders):
        task.cam.render_envs(out=out)
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / renders
    nbytes = float(B * H * W * 3)
    achieved = nbytes / (us * 1e-6) / 1e9
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1", "render_pmc.json")) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    del out, env
    torch.cuda.empty_cache()
    # reduced-resolution variant reported alongside: 96x128 images for 4096 envs
    Bs, Hs, Ws = ENVS_PER_GPU, 96, 128
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=Bs, enable_pixels=True, observation_height=Hs, observation_width=Ws,
                     camera_capture_mode="per_env")
    env.reset(seed=0)
    small = torch.empty((Bs, Hs, Ws, 3), dtype=torch.uint8, device=dev)
    for _ in range(3):
        env._env.cam.render_envs(out=small)
    torch.cuda.synchronize(dev)
    ev0.record()
    for _ in range(renders):
        env._env.cam.render_envs(out=small)
    ev1.record()
    torch.cuda.synchronize(dev)
    us_small = ev0.elapsed_time(ev1) * 1e3 / renders
    del small, env
    torch.cuda.empty_cache()
    return {"workload": "CubePick-v0 robot=franka enable_pixels=True per_env 480x640 RGB8, num_envs=1024 (BASELINE configs[4])",
            "env_frames_per_s": B / (us * 1e-6), "us_per_render": us, "dtype": "u8 out / f32 rays",
            "reduced_96x128_num_envs_4096": {"env_frames_per_s": Bs / (us_small * 1e-6), "us_per_render": us_small,
                                             "GBps": Bs * Hs * Ws * 3 / (us_small * 1e-6) / 1e9},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "mir_render_kernel",
                         "note": "algorithmic bytes = 1024x480x640x3 written once per render; time = whole mir_render call "
                                 "(FK refresh + primitive setup + pixel kernel), HIP events"}}


def grasp_bench(dev):
    """Secondary (SURVEY.md 8d, config 2's scripted-grasp scenario): the reference's expert pick -- five stages of 40 steps
    (examples/franka/pick_cube_state.py:86-88), joint-space targets precomputed by this repo's batched IK from the reset
    state -- on 4096 envs, so finger-pad/cube box-box contacts, friction, the arm-cube coupling in the Newton system and
    terminated=True are all exercised.  One fused launch per step, HIP events on the launching stream."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    task = env._env
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    eef = robot.get_link("hand")
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    stages = [(0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.104, 0.0), (0.40, 0.0)]
    targets, q_prev = [], None
    for dz, grip in stages:
        q = robot.inverse_kinematics(link=eef, pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        targets.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ncon_max = 0
    ev0.record()
    for tg in targets:
        for _ in range(40):
            task.step_raw(tg)
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / 200
    success = float((task._reward == 1).float().mean().item())
    ncon_max = int(task._mir.get_diag()[0].max().item())
    del env
    return {"workload": "CubePick-v0 robot=franka scripted pick (hover, stabilize, grasp, grasp, lift; 5 x 40 steps; IK-precomputed joint "
                        "targets), num_envs=4096", "env_steps_per_s": B / (us * 1e-6), "us_per_step": us, "lifted_frac": success,
            "max_contacts_last_step": ncon_max}


def so101_bench(dev, steps: int = 400):
    """Secondary (BASELINE configs[3] / SURVEY.md 8d config 4): SO-101 cube-pick (6 arm dofs, cube on the kitchen slab: box-box
    contact every step) at 4096 envs, U(-1,1) joint targets around the rest pose."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="so101", num_envs=B, enable_pixels=False)
    env.reset(seed=0)
    task = env._env
    gen = torch.Generator(device=dev).manual_seed(77)
    acts = torch.empty((256, B, task._zero.shape[1]), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
    for t in range(20):
        task.step_raw(acts[t])
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for t in range(steps):
        task.step_raw(acts[t % 256])
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    del env
    return {"workload": "CubePick-v0 robot=so101 (12 dofs, cube on the slab) state-only obs, U(-1,1) joint targets, num_envs=4096",
            "env_steps_per_s": B / (us * 1e-6), "us_per_step": us}


def stack_bench(dev, steps: int = 300):
    """Secondary: gym_genesis/CubeStack-v0 (robot=franka: Panda x0.6 + five free cubes on the island slab, 39 dofs) at 4096
    envs on the wave-per-env kernel (mir_step64).  Fresh PD targets = home + U(-1,1) per step, resident in HBM; one fused
    launch per step; HIP events on the launching stream.  Algorithmic bytes per env-step, same accounting as SURVEY.md 8d:
    read qpos 44 + qvel 39 + action 9 + warm start 39 = 131 f32, write qpos 44 + qvel 39 + warm start 39 + obs 23 + reward 1
    = 146 f32 + 1 B mask => 1109 B."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B, enable_pixels=False)
    task = env._env
    env.reset(seed=0)
    gen = torch.Generator(device=dev).manual_seed(4321)
    acts = task._home[0] + torch.empty((256, B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
    for t in range(30):
        task.step_raw(acts[t])
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for t in range(steps):
        task.step_raw(acts[t % 256])
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    ncon, _, niter = (x.float().mean().item() for x in task._mir.get_diag())
    # the same workload in 16-step rollout launches (mir_rollout)
    rows = torch.zeros((16, B, 9 + 14 + 2), dtype=torch.float32, device=dev)
    task._mir.rollout(acts[:16], rows)
    torch.cuda.synchronize(dev)
    ev0.record()
    for i in range(8):
        task._mir.rollout(acts[16 * i:16 * i + 16], rows)
    ev1.record()
    torch.cuda.synchronize(dev)
    us_ro = ev0.elapsed_time(ev1) * 1e3 / (8 * 16)
    algo = 1109.0
    achieved = algo * B / (us * 1e-6) / 1e9
    del env
    return {"workload": "CubeStack-v0 robot=franka (39 dofs, 5 cubes) state-only obs, home + U(-1,1) joint targets, num_envs=4096",
            "env_steps_per_s": B / (us * 1e-6), "us_per_step": us, "rollout16_env_steps_per_s": B / (us_ro * 1e-6),
            "mean_contacts": ncon, "mean_newton_iterations": niter,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "mir_step64_kernel",
                         "note": "1109 algorithmic B/env-step; one wave per env, 4 envs per CU: latency/occupancy-bound like the pick kernel"}}


def ik_bench(dev, calls: int = 200):
    """Secondary: the batched IK the reference's expert policies call once per env.step()
    (examples/franka/pick_cube_state.py:46-51): hand pose targets above the cube, 4096 envs, seed = current state."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    task = env._env
    target = (obs["environment_state"][:, :3] + torch.tensor([0.0, 0.0, 0.25], device=dev)).contiguous()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    hand = task.eef.idx
    for _ in range(5):
        q, err = task._mir.inverse_kinematics(hand, target, quat, return_error=True)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(calls):
        task._mir.inverse_kinematics(hand, target, quat)
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / calls
    return {"workload": "robot.inverse_kinematics(hand, pos, quat) from the home pose, num_envs=4096, <= 32 damped-least-squares iterations",
            "env_solves_per_s": B / (us * 1e-6), "us_per_call": us, "converged_frac": float(((err[:, 0] < 5e-4) & (err[:, 1] < 5e-3)).float().mean().item())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL observation gather")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N>1: all-gather the packed rows of this many consecutive steps in one collective (1 = every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pixels", action="store_true", help="skip the secondary pixels (configs[4]) measurement")
    ap.add_argument("--no-stack", action="store_true", help="skip the secondary CubeStack-v0 measurement")
    ap.add_argument("--core-only", action="store_true",
                    help="only the timed headline loop (no API / autoreset / rollout / pixels / stack / ik / CPU legs): the command "
                         "profiled by tools/collect_profiles.sh, so that every mir_step_kernel launch in the trace is a headline step")
    ap.add_argument("--force-gather", action="store_true", help="exercise the RCCL gather path even with one rank (plumbing check)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    use_pg = world > 1 or args.force_gather
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from gym_genesis.env import GenesisEnv
    from gym_genesis.sharding import gather_rows  # noqa: F401  (collective lives there)

    B = args.envs_per_gpu
    K, W = args.steps, args.warmup
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B * world, enable_pixels=False, shard=(rank, world))
    task = env._env
    env.reset(seed=0)

    # inputs resident in HBM before the timed region: one fresh action batch per step
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = min(W + K, 4096)
    actions = torch.empty((n_act, B, 9), dtype=torch.float32, device=dev)
    for i in range(0, n_act, 256):
        actions[i:i + 256].uniform_(-1.0, 1.0, generator=gen)

    gather = use_pg and not args.no_gather
    # One collective per step costs ~60 us of host time in torch.distributed (measured at world_size 1), more than the
    # 41 us step itself, so rows are gathered in chunks of S steps: the host then stays ahead of the GPU.
    S = max(1, args.gather_every)
    rows = [torch.zeros((S, B, ROW_STRIDE), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * S, B, ROW_STRIDE), dtype=torch.float32, device=dev) for _ in range(2)] if gather else None
    pending = [None, None]
    launches = 0
    filled = 0  # steps written into the current chunk

    chunk = 0

    def flush():
        """All-gather the rows written so far in the current chunk (async: overlaps the next chunk's physics)."""
        nonlocal filled, chunk
        if gather and filled:
            s = chunk & 1
            if filled == S:
                pending[s] = dist.all_gather_into_tensor(gathered[s], rows[s], async_op=True)
            else:  # partial chunk at the end of a run
                pending[s] = dist.all_gather_into_tensor(gathered[s][:world * filled], rows[s][:filled], async_op=True)
            chunk += 1
            filled = 0

    def one_step(t: int):
        nonlocal launches, filled, chunk
        a = actions[t % n_act]
        if gather:
            s = chunk & 1
            if filled == 0 and pending[s] is not None:
                pending[s].wait()      # stream-level: the buffer's previous gather has drained
                pending[s] = None
            task._mir.step_packed(a, rows[s][filled])
            filled += 1
            if filled == S:
                flush()
        else:
            task.step_raw(a)
        launches += 1
        if (t + 1) % EPISODE_STEPS == 0:  # reset-all, as gymnasium's TimeLimit(200) makes the README loop do
            task.reset()
            launches += 1

    def sync_all():
        flush()
        for p in pending:
            if p is not None:
                p.wait()
        torch.cuda.synchronize(dev)
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for t in range(W):
        one_step(t)
    sync_all()
    launches = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for t in range(W, W + K):
        one_step(t)
    ev1.record()
    sync_all()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if use_pg:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())

    if args.core_only:
        args.no_pixels = args.no_stack = args.no_cpu_baseline = True
    # end-to-end env.step() (with the API's per-step D->H `terminated` copy), reported beside the hot path
    api_steps = 0 if args.core_only else 200
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for t in range(api_steps):
        env.step(actions[t % n_act])
    torch.cuda.synchronize(dev)
    api_rate = api_steps * B * world / (time.perf_counter() - t1) if api_steps else None
    # the README loop's variant: actions sampled on the HOST (README.md:34): + one 147 KB H->D copy per step
    np_rate = None
    if api_steps:
        acts_np = np.random.default_rng(5).uniform(-1, 1, (8, B, 9)).astype(np.float32)
        torch.cuda.synchronize(dev)
        t1b = time.perf_counter()
        for t in range(api_steps):
            env.step(acts_np[t % 8])
        torch.cuda.synchronize(dev)
        np_rate = api_steps * B * world / (time.perf_counter() - t1b)

    # physics only: mir_step without the observation / reward outputs, PD targets unchanged
    phys_rate = None
    if api_steps:
        torch.cuda.synchronize(dev)
        t1c = time.perf_counter()
        for t in range(api_steps):
            task._mir.step(1)
        torch.cuda.synchronize(dev)
        phys_rate = api_steps * B * world / (time.perf_counter() - t1c)

    # device-resident episode loop: fused step + on-device truncation/termination/re-spawn, no host sync
    if api_steps:
        task.enable_autoreset(max_episode_steps=EPISODE_STEPS)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    for t in range(api_steps):
        task.step_autoreset(actions[t % n_act])
    torch.cuda.synchronize(dev)
    loop_rate = api_steps * B * world / (time.perf_counter() - t2) if api_steps else None

    # K-step rollout launches (mir_rollout): the same fresh-action workload with the state kept on chip between steps
    RK = 16
    rows_ro = torch.zeros((RK, B, ROW_STRIDE), dtype=torch.float32, device=dev)
    nro = 0 if args.core_only else max(1, min(200, n_act) // RK)
    if nro:
        task.reset()
    for i in range(0 if args.core_only else 2):
        task._mir.rollout(actions[i * RK:(i + 1) * RK], rows_ro)
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    for i in range(nro):
        task._mir.rollout(actions[i * RK:(i + 1) * RK], rows_ro)
    torch.cuda.synchronize(dev)
    rollout_rate = nro * RK * B * world / (time.perf_counter() - t3) if nro else None
    # the same with the device-side episode loop (truncation at 200 steps, re-spawn from the pre-drawn pool) inside the launch
    loop_rollout_rate = None
    if nro:
        rows_ar = torch.zeros((RK, B, ROW_STRIDE), dtype=torch.float32, device=dev)
        task.reset()
        task._episode_len.zero_()
        task.rollout_autoreset(actions[:RK], rows_ar)
        torch.cuda.synchronize(dev)
        t4 = time.perf_counter()
        for i in range(nro):
            task.rollout_autoreset(actions[i * RK:(i + 1) * RK], rows_ar)
        torch.cuda.synchronize(dev)
        loop_rollout_rate = nro * RK * B * world / (time.perf_counter() - t4)

    if rank == 0:
        value = K * B * world / wall_max
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes (profiles/, measured offline on this kernel)
        try:
            with open(os.path.join(ROOT, "profiles", "r1", "pmc_hbm_traffic.json")) as f:
                traffic = json.load(f)["hbm_bytes_per_launch_uncorrected"] if B == ENVS_PER_GPU else None
        except (OSError, KeyError, ValueError):
            pass
        launch_us = gpu_ms * 1e3 / max(launches, 1)  # HIP events on the launching stream, per step-kernel launch
        achieved = ALGO_BYTES_PER_ENV_STEP * B / (launch_us * 1e-6) / 1e9
        out = {
            "metric": "env-steps/sec (num_envs x sim-steps/sec), CubePick-v0 @ num_envs=4096",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": wall_max * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "CubePick-v0 robot=franka state-only obs, U(-1,1) joint-target actions, reset-all every 200 steps",
                       "num_envs_per_gpu": B, "global_num_envs": B * world, "parallelism": f"env-axis shard x{world}",
                       "obs_gather": f"rccl all_gather of {S}-step row chunks, overlapped with the next chunk" if gather else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "mir_step_kernel", "kernel_us": launch_us,
                         "note": "489 algorithmic B/env-step x 4096 envs per launch; the path is latency/occupancy-bound, not HBM-bound (SURVEY.md 8d)"},
            "env_step_api_rate": api_rate,
            "env_step_api_numpy_actions_rate": np_rate,
            "physics_only_rate": phys_rate,
            "device_autoreset_loop_rate": loop_rate,
            "device_rollout16_rate": rollout_rate,
            "device_autoreset_rollout16_rate": loop_rollout_rate,
        }
        if world == 1 and not args.no_pixels:
            out["pixels"] = pixels_bench(dev)
        if world == 1 and not args.no_stack:
            out["scripted_grasp"] = grasp_bench(dev)
            out["so101_pick"] = so101_bench(dev)
            out["stack"] = stack_bench(dev)
            out["ik"] = ik_bench(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
