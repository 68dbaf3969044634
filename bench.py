#!/usr/bin/env python3
"""bench.py — env-steps/sec of the CubePick-v0 `env.step()` hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: spawns N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): GenesisEnv(task="cube_pick", robot="franka", num_envs=4096 per
GPU, enable_pixels=False); reset(seed=0); a fresh a_t ~ U(-1,1)^(B x 9) float32 per step, generated on the device BEFORE the
timed region (inputs resident in HBM).  The timed loop is the reference's own loop shape (README.md:32-43, test.py:11-22):

    obs, reward, terminated, truncated, info = env.step(a_t)      # GenesisEnv.step, terminated = NumPy bool (B,)
    if terminated.any() or truncated.any() or t % 200 == 199:      # TimeLimit(200) parity (gym_genesis/__init__.py:6)
        env.reset()

`value` = num_envs x env.step() calls per wall-second THROUGH GenesisEnv.step, max over ranks.  Exactly K steps
are timed per repetition, bracketed by barrier + synchronize; when K steps are too short to time (K = 20 is ~1 ms) the bracketed
K-step region is REPEATED until >= --min-time seconds have been measured, and `value` is total steps / total bracketed time
(`repeats` in the line).  `hot_path_rate` is the bare fused launch (task.step_raw: no per-step host hand-over of `terminated`)
over the same number of steps; its HIP-event time per launch is the kernel duration used by `roofline`.

Env axis sharded across ranks (weak scaling: 4096 envs per GPU); with N > 1 the observation/reward outputs of --gather-every
consecutive steps are all-gathered over RCCL in one collective, overlapped with the following steps.

The headline line is assembled first; every secondary leg (raw launch, rollout, autoreset, pixels, grasp, SO-101, stack, IK,
CPU baseline) runs in its own try/except and can only add an {"error": ...} object, and the line is printed in a `finally`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]

ENVS_PER_GPU = 4096
EPISODE_STEPS = 200
ALGO_BYTES_PER_ENV_STEP = 489.0  # 55 f32 read + 67 f32 + 1 B written
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: peak FP32 (vector), spec
ROW_STRIDE = 24
N_ACT = 256                      # pre-drawn action batches (device resident), cycled
RK = 16                          # steps per rollout launch in the secondary rollout legs
assert N_ACT % RK == 0
METRIC = "env-steps/sec (num_envs x sim-steps/sec), CubePick-v0 @ num_envs=4096"


# ---- pure helpers (covered by tests/test_bench_cpu.py) ---------------------------------------------------------------
def action_index(t: int, n_act: int = N_ACT) -> int:
    """Index of the action batch of global step t."""
    return t % n_act


def rollout_slice(i: int, rk: int = RK, n_act: int = N_ACT):
    """[lo, hi) of the i-th rk-step rollout in the action pool: always rk rows, always inside the pool."""
    lo = (i * rk) % n_act
    if lo + rk > n_act:
        lo = 0
    return lo, lo + rk


def repeats_for(seconds_per_rep: float, min_time: float, max_reps: int = 2000) -> int:
    """How many K-step repetitions reach min_time seconds of measured time (at least 1)."""
    if seconds_per_rep <= 0.0:
        return max_reps
    return int(max(1, min(max_reps, -(-min_time // seconds_per_rep))))


def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def rank_cpu_plan(local_rank: int, world: int, allowed, gpu_numa=None, numa_cpus=None):
    """Which host cores rank `local_rank` of `world` takes: (sorted cores of its slice, the core of its stepping thread).

    env.step is a ~22 us host / device ping-pong in which the host SPINS on the launch's terminated bytes (mir_step_end): every
    rank needs a core of its own for that thread, close to its GPU, and its helper threads (HIP runtime, RCCL proxy, allocator)
    must not share it.  The cores the process may use that belong to the NUMA node of the rank's GPU are split evenly among the
    ranks whose GPUs sit on that node; without topology information (or with fewer than two such cores per rank) the allowed
    cores are split evenly among all ranks.  Returns ([], None) when there are fewer allowed cores than ranks."""
    allowed = sorted(allowed)
    if len(allowed) < world or not (0 <= local_rank < world):
        return [], None
    group, pool = list(range(world)), allowed
    if gpu_numa and numa_cpus and local_rank < len(gpu_numa) and gpu_numa[local_rank] is not None:
        node = gpu_numa[local_rank]
        peers = [r for r in range(world) if r < len(gpu_numa) and gpu_numa[r] == node]
        near = sorted(set(numa_cpus.get(node, ())) & set(allowed))
        if len(near) >= 2 * len(peers):
            group, pool = peers, near
    i, n = group.index(local_rank), len(group)
    mine = pool[len(pool) * i // n:len(pool) * (i + 1) // n]
    return mine, (mine[0] if mine else None)


def gpu_numa_nodes():
    """NUMA node of every GPU in HIP device order, from sysfs (no GPU call): the KFD topology lists the GPU nodes (simd_count > 0)
    in the order the runtime enumerates them; *_VISIBLE_DEVICES given as integer lists re-map it.  None where unknown."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    nodes = []
    try:
        for n in sorted(os.listdir(base), key=int):
            props = {}
            with open(os.path.join(base, n, "properties")) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) <= 0:
                continue
            numa = None
            try:
                with open(f"/sys/class/drm/renderD{int(props['drm_render_minor'])}/device/numa_node") as f:
                    numa = int(f.read())
                if numa < 0:
                    numa = None
            except (OSError, KeyError, ValueError):
                numa = None
            nodes.append(numa)
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                nodes = [nodes[int(x)] for x in v.split(",") if x.strip() != ""]
            except (ValueError, IndexError):
                return None
    return nodes or None


def numa_cpu_lists():
    """{NUMA node: [cores]} from sysfs."""
    out, base = {}, "/sys/devices/system/node"
    try:
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                with open(os.path.join(base, d, "cpulist")) as f:
                    out[int(d[4:])] = parse_cpulist(f.read())
    except OSError:
        return {}
    return out


def move_other_threads(cpus, keep_tid: int) -> int:
    """Affinity of every thread of this process except `keep_tid` <- cpus; returns how many were moved."""
    moved = 0
    try:
        for t in os.listdir("/proc/self/task"):
            tid = int(t)
            if tid != keep_tid:
                try:
                    os.sched_setaffinity(tid, cpus)
                    moved += 1
                except OSError:
                    pass
    except OSError:
        pass
    return moved


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--min-time", type=float, default=0.5, help="repeat the timed K-step region until this many seconds are measured")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL observation gather")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N>1: all-gather the outputs of this many consecutive steps in one collective (1 = every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pixels", action="store_true", help="skip the secondary pixels (configs[4]) measurement")
    ap.add_argument("--no-stack", action="store_true", help="skip the secondary grasp / SO-101 / CubeStack-v0 / IK measurements")
    ap.add_argument("--core-only", action="store_true",
                    help="only the headline loop and the raw-launch loop (the command profiled by tools/collect_profiles.sh)")
    ap.add_argument("--raw-only", action="store_true",
                    help="with --core-only: skip the API loop too, so that every mir_step_kernel launch in a trace is a raw launch")
    ap.add_argument("--force-gather", action="store_true", help="exercise the gather path even with one rank (plumbing check)")
    ap.add_argument("--gather-path", choices=("copy", "rccl"), default="copy",
                    help="N>1: how the observation gather travels -- copy: peer-to-peer device copies on the SDMA engines + sequence words "
                         "(sharding.CopyPathGather; no kernel takes CUs / LDS from the step kernel), verified at start-up against the RCCL "
                         "collective and replaced by it where peer access does not work; rccl: all_gather_into_tensor")
    ap.add_argument("--output-ring", type=int, default=0,
                    help="experiment: step outputs as rows of a ring of this many steps even without a gather (not the headline's semantics: "
                         "GenesisEnv.step hands out fresh tensors by default)")
    ap.add_argument("--no-gather-ab", action="store_true",
                    help="skip the second headline measurement WITHOUT the gather that gives `gather_overhead_us` (runs whenever a gather is on)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for plumbing checks)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="plumbing check: let several ranks share a GPU (device = LOCAL_RANK %% device_count; use with --dist-backend gloo)")
    ap.add_argument("--cpu-baseline-child", type=float, default=None, metavar="SECONDS",
                    help="internal: run the CPU baseline in this (GPU-free) process for about SECONDS and print its JSON object")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="only spawn the ranks, rendezvous over gloo on CPU and print a line (no GPU, no physics)")
    return ap.parse_args(argv)


# ---- self-launch -----------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(gpus: int, argv, port: int):
    """The driver's own launch line (one rank per GPU over torch.distributed.run)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


PREFLIGHT_CHILD = r'''
import json, torch
n = torch.cuda.device_count()
print(json.dumps([[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]))
'''


def preflight(gpus: int, oversubscribe: bool = False, device_count=None, peer_matrix=None, out=sys.stderr):
    """Before any rank is started, in the launching parent: one line per rank.  The parent starts children only and exec's nothing; the
    peer-access matrix comes from a short-lived child, the device count from torch.cuda.device_count() -- which reads it without
    initialising HIP where amdsmi is present and falls back to hipGetDeviceCount where it is not (harmless here, nothing is
    exec'ed from this process, but "never touches the GPU" was too strong): one line per rank -- device, NUMA node of the GPU, the host
    cores the rank will take (rank_cpu_plan), which peers its GPU can address (what the copy-path gather needs; without it the
    ranks agree on the RCCL collective) -- and a verdict.  Returns 0 when every rank can be placed on a GPU of its own, 2 otherwise.
    `device_count` / `peer_matrix` are injectable for the CPU tests."""
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()
    numa, cpus = gpu_numa_nodes(), numa_cpu_lists()
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    if peer_matrix is None and device_count > 1:
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            r = subprocess.run([sys.executable, "-c", PREFLIGHT_CHILD], capture_output=True, text=True, timeout=180, env=env)
            peer_matrix = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
        except Exception:  # noqa: BLE001
            peer_matrix = None
    print(f"[bench preflight] {gpus} rank(s) requested, {device_count} GPU(s) visible, {len(allowed)} host core(s) allowed", file=out)
    ok = True
    for r in range(gpus):
        dev = r if r < device_count else (r % device_count if (oversubscribe and device_count) else None)
        mine, spin = rank_cpu_plan(r, gpus, allowed, numa, cpus)
        node = numa[dev] if (numa and dev is not None and dev < len(numa)) else None
        peers = "?" if (peer_matrix is None or dev is None or dev >= len(peer_matrix)) else "".join("1" if x else "0" for x in peer_matrix[dev])
        where = "NONE (more ranks than GPUs)" if dev is None else f"cuda:{dev}" + (" (shared: --oversubscribe)" if r >= device_count else "")
        print(f"[bench preflight]   rank {r}: device {where}, GPU NUMA node {'?' if node is None else node}, host cores "
              f"{('%d-%d' % (mine[0], mine[-1])) if mine else 'left to the scheduler'} (stepping thread on {spin if spin is not None else '-'}), "
              f"can address peers {peers}", file=out)
        ok = ok and dev is not None
    if not ok:
        print(f"[bench preflight] FAILED: {gpus} ranks do not fit {device_count} GPU(s) (one rank per GPU; --oversubscribe shares GPUs for plumbing checks)", file=out)
    elif peer_matrix is not None and gpus > 1 and not all(all(row[:gpus]) for row in peer_matrix[:gpus]) and not oversubscribe:
        print("[bench preflight] note: not every GPU can address every peer -- the observation gather will use the RCCL collective (the ranks agree on it)", file=out)
    return 0 if ok else 2


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (this parent never touches the GPU,
    and nothing is exec'ed), pass their output through, exit with their code."""
    if not args.selftest_launch:
        rc = preflight(args.gpus, args.oversubscribe)
        if rc != 0:
            return rc
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(args.gpus, argv, _free_port())
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ---- secondary legs --------------------------------------------------------------------------------------------------
def cpu_baseline_child(budget_s: float) -> dict:
    """Runs in a CHILD process of the bench (never touches the GPU; OMP_PROC_BIND=spread and OMP_PLACES=cores are in its environment
    before the first OpenMP call): the float32 CPU port of the oracle (oracle/liborc32.so, kind "port") on the same workload.  The
    OpenMP team size is calibrated first (the host may expose more cores than its CPU quota sustains): the best of three timed steps
    per candidate; then five timed runs of the same steps at the chosen size: value = the best of them (a shared host only slows runs
    down), with the median and the slowest beside it."""
    import numpy as np
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
    best, used, k = None, cores, 0
    for nt in sorted({n for n in (4, 8, 16, 32, 64, 128, cores) if n <= cores}):
        o.step_batch(acts[0], nt)  # thread-pool / page-fault warm-up
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            o.step_batch(acts[1 + k % 63], nt)
            dts.append(time.perf_counter() - t0)
            k += 1
        if best is None or min(dts) < best:
            best, used = min(dts), nt
    runs, per_run = [], max(4, min(400, int(budget_s / 5.0 / max(best, 1e-6))))
    # every run times the SAME steps: the scene is put back to one snapshot first (consecutive stretches of one trajectory differ in
    # their contacts and Newton iterations by more than the host's timing noise: 1.65 between the fastest and the slowest stretch)
    snap = [(f, o.read_all(f, 64)) for f in (orc.F_QPOS, orc.F_QVEL, orc.F_QACC_WS)]
    for _ in range(5):
        for f, v in snap:
            o.write_all(f, v)
        t0 = time.perf_counter()
        for i in range(per_run):
            o.step_batch(acts[i % 64], used)
        runs.append(per_run * B / (time.perf_counter() - t0))
    order = [round(r) for r in runs]
    runs.sort()
    # value = the BEST run: the GPU boxes' hosts are shared, and a neighbour's load only ever slows a run down (three calls in a row on
    # one box: medians 1.63 / 1.49 / 1.09 M, bests 1.65 / 1.65 / 1.48 M); the median and the slowest run stand beside it
    return {"value": runs[-1], "median": runs[len(runs) // 2], "min": runs[0], "max": runs[-1], "runs_in_order": order, "unit": "env-steps/s", "cores": used, "threads": used, "kind": "port",
            "sample": f"best of 5 runs of the same {per_run} steps x {B} envs, headline workload, f32 C port of the oracle, OpenMP"}


def cpu_baseline(budget_s: float = 12.0):
    """The CPU baseline of the contract line, timed in a child process (cpu_baseline_child): OMP_PROC_BIND / OMP_PLACES must be set before
    the first OpenMP call of a process, and this one has long made its own (torch) -- and has pinned its stepping thread."""
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_PROC_BIND"], env["OMP_PLACES"] = "spread", "cores"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(budget_s)], env=env, capture_output=True, text=True,
                       timeout=60.0 + 6.0 * budget_s)
    if r.returncode != 0:
        raise RuntimeError(f"cpu baseline child failed: {r.stderr[-400:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def stack_flops_per_env_step(envs: int = 32, steps: int = 40):
    """F_step of the CubeStack-v0 workload of stack_bench (Franka, five cubes, home + U(-1,1) joint targets), counted like
    flops_per_env_step by the instrumented oracle at the full capacities (oracle/liborc_flops_big.so)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from gym_genesis.backend import models

    spec = models.franka_cube_stack_scene().build()
    o = orc.Oracle(spec, envs, f32=True, flops="big")
    rng = np.random.RandomState(0)
    pos = np.zeros((envs, 5, 3), np.float32)
    pos[:, :, 0] = np.array([-0.3, -0.15, 0.0, 0.15, 0.3]) + rng.uniform(-0.03, 0.03, (envs, 5))
    pos[:, :, 1] = rng.uniform(-0.2, 0.2, (envs, 5))
    pos[:, :, 2] = models.STACK_CUBE_Z
    home = np.tile(np.array(models.FRANKA_HOME, np.float32), (envs, 1))
    o.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (envs, 5, 1)), home)
    acts = (home + np.random.default_rng(4321).uniform(-1, 1, (steps, envs, 9))).astype(np.float32)
    for t in range(10):
        o.step_batch(acts[t], 1)
    o.flops_reset()
    for t in range(steps):
        o.step_batch(acts[t], 1)
    f = o.flops()
    n = float(steps * envs)
    kinds = {k: v / n for k, v in f.items()}
    return {"F_step": sum(v for k, v in kinds.items() if k != "cmp"), "per_kind": kinds,
            "sample": f"{steps} steps x {envs} envs of the stack workload, float32 port of the oracle with counted arithmetic (one thread)"}


def flops_per_env_step(envs: int = 256, steps: int = 200):
    """F_step: floating-point operations per env-step of the headline workload, counted by the oracle's instrumented build
    (oracle/liborc_flops.so: the float32 port compiled with an arithmetic type that counts its own +, -, *, /, sqrt, sin / cos / atan2 /
    pow; SURVEY.md 8d "Figures for roofline.achieved").  Same scene, same reset stream, same U(-1,1) joint targets, one episode of
    `steps` steps on a sample of `envs` envs.  An ALGORITHMIC count of a dense scalar restatement: fma counts as two (mul + add)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from gym_genesis.backend import models

    spec = models.franka_cube_pick_scene().build()
    o = orc.Oracle(spec, envs, f32=True, flops=True)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, envs), rng.uniform(-0.25, 0.25, envs), np.full(envs, 0.02)], 1).astype(np.float32)
    o.reset(pos, np.tile(np.array([0, 0, 0, 1.0], np.float32), (envs, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (envs, 1)))
    o.step_batch(None, 1)
    acts = np.random.default_rng(1234).uniform(-1, 1, (steps, envs, 9)).astype(np.float32)
    o.flops_reset()
    for t in range(steps):
        o.step_batch(acts[t], 1)  # (one thread: the counters are per thread)
    f = o.flops()
    n = float(steps * envs)
    kinds = {k: v / n for k, v in f.items()}
    return {"F_step": sum(v for k, v in kinds.items() if k != "cmp"), "per_kind": kinds,
            "sample": f"{steps} steps x {envs} envs of the headline workload, float32 port of the oracle with counted arithmetic (one thread)"}


def valu_roofline(fl: dict, kernel_us: float, kernel: str, B: int, sq_profile: str):
    """The roofline that is nearer to binding than HBM: fp32 vector throughput.  achieved = F_step x envs per launch /
    the kernel's duration; beside it the kernel's own instruction-level count from the committed SQ counter pass
    (SQ_INSTS_VALU_FLOPS_FP32 wave-instructions x 64 lanes per launch: every lane slot counted, fma = 2) for the same kernel."""
    ach = fl["F_step"] * B / (kernel_us * 1e-6) / 1e12
    out = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / VALU_PEAK_TFLOPS, "kernel": kernel,
           "kernel_us": kernel_us, "F_step": fl["F_step"], "F_step_per_kind": fl["per_kind"], "F_step_source": fl["sample"],
           "note": "algorithmic flops of the float32 CPU restatement per env-step x 4096 envs per launch / measured kernel time, against the fp32 "
                   "vector peak; the path is a dependent chain on one or two waves per SIMD (latency/occupancy-bound), so neither this nor the "
                   "HBM figure is near 1"}
    try:
        sq = None
        for rnd in ("r5", "r4", "r3", "r2"):  # (the latest committed counter pass of this kernel)
            path = os.path.join(ROOT, "profiles", rnd, sq_profile)
            if os.path.exists(path):
                with open(path) as f:
                    sq = json.load(f)
                sq_profile = f"{rnd}/{sq_profile}"
                break
        wi = float(sq["SQ_INSTS_VALU_FLOPS_FP32"]["median_per_launch"])
        lane_flops = wi * 64.0
        out["instruction_level"] = {"source": f"profiles/{sq_profile}: SQ_INSTS_VALU_FLOPS_FP32 (median per launch, 4096 envs) x 64 lanes",
                                    "flops_per_launch": lane_flops, "flops_per_env_step": lane_flops / ENVS_PER_GPU,
                                    "achieved": lane_flops / (kernel_us * 1e-6) / 1e12, "frac": lane_flops / (kernel_us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS,
                                    "over_algorithmic": lane_flops / ENVS_PER_GPU / fl["F_step"],
                                    "valu_wave_instructions_per_launch": float(sq.get("SQ_INSTS_VALU", {}).get("median_per_launch", 0.0))}
    except (OSError, KeyError, ValueError, TypeError):
        out["instruction_level"] = None
    return out


def _events(torch):
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def pixels_bench(torch, dev, renders: int = 100):
    """BASELINE.json configs[4]: CubePick-v0, 1024 envs, enable_pixels=True, per-env 480x640 RGB8 images rendered by
    the tiled HIP rasteriser (mir_render) from the state resident in HBM; only the render launches are timed (setup +
    pixel kernels), HIP events on the launching stream.  Algorithmic bytes = B*H*W*3 written once (cfg 5)."""
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
    # (a render writes 0.94 GB in ~0.2 ms: the first few dozen launches after the step legs run 15-25 % slower than the
    #  steady state -- tools/probes/render_sweep.py shows the same on its first row -- so warm up for ~20 ms and take the
    #  median of three regions)
    for _ in range(100):
        task.cam.render_envs(out=out)
    torch.cuda.synchronize(dev)
    ev0, ev1 = _events(torch)
    regions, host = [], []
    for _ in range(3):
        ev0.record()
        t0 = time.perf_counter()
        for _ in range(renders):
            task.cam.render_envs(out=out)
        host.append((time.perf_counter() - t0) * 1e6 / renders)  # enqueue only: well below the GPU time, or the region measures the host
        ev1.record()
        torch.cuda.synchronize(dev)
        regions.append(ev0.elapsed_time(ev1) * 1e3 / renders)
    us = sorted(regions)[1]
    # calibration: a plain device fill of the same 0.94 GB (what this box's write path sustains; boxes of the pool differ)
    flat = out.view(-1).view(torch.int32)
    for _ in range(20):
        flat.fill_(7)
    torch.cuda.synchronize(dev)
    ev0.record()
    for _ in range(renders):
        flat.fill_(7)
    ev1.record()
    torch.cuda.synchronize(dev)
    fill_us = ev0.elapsed_time(ev1) * 1e3 / renders
    del flat
    nbytes = float(B * H * W * 3)
    achieved = nbytes / (us * 1e-6) / 1e9
    traffic = _profile_number("render_pmc.json", "hbm_bytes_per_launch")
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
    # the registry's DEFAULT pixel mode (camera_capture_mode="global", reference __init__.py:16; also GenesisEnv.render()): one 480x640
    # image of all 4096 envs at their grid offsets
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=ENVS_PER_GPU, enable_pixels=True, observation_height=H, observation_width=W,
                     camera_capture_mode="global")
    env.reset(seed=0)
    for _ in range(10):
        env._env.cam.render_global()
    torch.cuda.synchronize(dev)
    ev0.record()
    for _ in range(renders):
        env._env.cam.render_global()
    ev1.record()
    torch.cuda.synchronize(dev)
    us_global = ev0.elapsed_time(ev1) * 1e3 / renders
    del env
    torch.cuda.empty_cache()
    # the reference README's own loop (README.md:28-44) through the registry's defaults (robot so101, global pixels): NumPy actions,
    # env.step, env.render() every iteration; wall clock per iteration
    import gym_genesis
    import numpy as np
    env = gym_genesis.make("gym_genesis/CubePick-v0", num_envs=ENVS_PER_GPU, enable_pixels=True)
    env.reset(seed=0)
    acts = np.random.default_rng(3).uniform(-1, 1, (ENVS_PER_GPU, 6)).astype(np.float32)
    for _ in range(20):
        env.step(acts)
        env.render()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(renders):
        env.step(acts)
        env.render()
    torch.cuda.synchronize(dev)
    us_readme = (time.perf_counter() - t0) * 1e6 / renders
    del env
    torch.cuda.empty_cache()
    return {"workload": "CubePick-v0 robot=franka enable_pixels=True per_env 480x640 RGB8, num_envs=1024 (BASELINE configs[4])",
            "global_view_480x640_num_envs_4096_us_per_render": us_global,
            "readme_loop_registry_defaults_num_envs_4096_us_per_iteration": us_readme,
            "env_frames_per_s": B / (us * 1e-6), "us_per_render": us, "us_per_render_regions": regions, "fill_us_same_buffer": fill_us, "host_enqueue_us_per_render": sorted(host)[1], "dtype": "u8 out / f32 rays",
            "reduced_96x128_num_envs_4096": {"env_frames_per_s": Bs / (us_small * 1e-6), "us_per_render": us_small,
                                             "GBps": Bs * Hs * Ws * 3 / (us_small * 1e-6) / 1e9},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "mir_render_binned",
                         "note": "algorithmic bytes = 1024x480x640x3 written once per render; time = whole mir_render call "
                                 "(FK refresh + primitive setup + pixel kernel), HIP events"}}


def grasp_bench(torch, dev):
    """Secondary (SURVEY.md 8d, config 2's scripted-grasp scenario): five stages of 40 steps like the reference's expert
    (examples/franka/pick_cube_state.py:86-88: hover, stabilize, grasp, grasp, lift), joint-space targets precomputed by this
    repo's batched IK from the reset state, on 4096 envs: finger-pad/cube contacts, friction, the arm-cube coupling in the
    Newton system and terminated=True are all exercised.  One fused launch per step, HIP events on the launching stream."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    task = env._env
    _bare_launches(task)
    task._mir.set_diag(True)
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    eef = robot.get_link("hand")
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    stages = [(0.25, 0.04), (0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.40, 0.0)]
    targets, q_prev = [], None
    for dz, grip in stages:
        q = robot.inverse_kinematics(link=eef, pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        targets.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    torch.cuda.synchronize(dev)
    ev0, ev1 = _events(torch)
    ev0.record()
    for tg in targets:
        for _ in range(40):
            task.step_raw(tg)
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / 200
    success = float((task._reward == 1).float().mean().item())
    ncon_max = int(task._mir.get_diag()[0].max().item())
    # how often the 16-point contact capacity of the pick kernel is reached: the same 200 steps once more, untimed, reading the
    # candidate-point count of every env after every step (mir_get_diag4; more than 16 = the manifolds of that env-step were thinned)
    env.reset(seed=0)
    hits = pts_max = 0
    for tg in targets:
        for _ in range(40):
            task.step_raw(tg)
            pts = task._mir.get_diag(points=True)[3]
            hits += int((pts > 16).sum().item())
            pts_max = max(pts_max, int(pts.max().item()))
    del env

    # the same 200 steps through GenesisEnv.step (host-visible `terminated` every step), with the manifolds thinned at 16 points (the
    # default) and with exact contacts (GenesisEnv(..., exact_contacts=True): the envs with more than 16 candidate points are stepped
    # by the wave kernel, 48 points, nothing thinned) -- median of five episodes each
    def api_leg(exact: bool):
        e = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=exact)
        mir = e._env._mir
        walls, lifted = [], 0.0
        for rep in range(6):
            e.reset(seed=0)
            mir.exact_stats(reset=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            any_term = None
            for tg in targets:
                for _ in range(40):
                    _, _, term, _, _ = e.step(tg)
                    any_term = term if any_term is None else (any_term | term)
            torch.cuda.synchronize(dev)
            if rep:  # (the first episode warms up)
                walls.append((time.perf_counter() - t0) / 200)
            lifted = float(any_term.mean())
        st = mir.exact_stats()
        med = sorted(walls)[len(walls) // 2]
        res = {"env_steps_per_s": B / med, "us_per_step": med * 1e6, "lifted_frac": lifted}
        if exact:
            res.update({"overflow_env_frac": st["overflow_env_steps"] / (200.0 * B), "overflow_step_frac": st["overflow_steps"] / 200.0,
                        "overflow_envs_max": st["overflow_envs_max"]})
        del e
        return res

    thin, exact = api_leg(False), api_leg(True)
    return {"workload": "CubePick-v0 franka scripted pick, 5 x 40 steps, IK-precomputed joint targets, num_envs=4096", "env_steps_per_s": B / (us * 1e-6),
            "us_per_step": us, "lifted_frac": success, "max_contacts_last_step": ncon_max, "cap_hit_frac": hits / (200.0 * B),
            "max_candidate_points": pts_max, "contact_capacity": 16, "env_step_thinned": thin, "env_step_exact_contacts": exact}


def _bare_launches(task):
    """The legs that time BARE launches (task.step_raw back to back, device-side rollouts) have no host in them to close a step on: they
    run with exact contacts switched off (the pick tasks' default is on since round 6; mir_step_fused would otherwise wait for every
    step's terminated bytes, and mir_rollout* are refused).  Returns whether the switch was on."""
    was = bool(getattr(task._mir, "exact_contacts", False))
    if was:
        task._mir.set_exact_contacts(False)
    return was


def ref_expert_bench(torch, dev, episodes: int = 2):
    """Secondary: the reference's own expert (/root/reference/examples/franka/pick_cube_state.py:16-54,86-88 as restated
    in examples/franka/pick_cube_state.py: IK every step, the hand pressed to 3 cm above the cube -- 27 % of its env-steps above 16
    contact points) at 4096 envs: microseconds INSIDE env.step() per call and per loop iteration, with every contact kept (the
    task's default) and with the manifolds thinned at 16 points (exact_contacts=False)."""
    import importlib.util

    from gym_genesis.env import GenesisEnv

    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    B = ENVS_PER_GPU
    res = {"workload": "reference expert (pick_cube_state.py), 5 x 40 steps, IK every step, num_envs=4096"}
    # (exact_contacts: the task's default, whose overflow steps are TWO launches in a loop that leaves room between its steps -- this one
    #  does: the policy and its IK; exact_contacts_one_launch: MIR_EXACT_BIG=0, the list launches / heavy phase a tight loop gets)
    for key, kw in (("exact_contacts", {}), ("exact_contacts_one_launch", {}), ("thinned", {"exact_contacts": False})):
        old_big = os.environ.get("MIR_EXACT_BIG")
        if key == "exact_contacts_one_launch":
            os.environ["MIR_EXACT_BIG"] = "0"
        try:
            env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, **kw)
        finally:
            if key == "exact_contacts_one_launch":
                if old_big is None:
                    os.environ.pop("MIR_EXACT_BIG", None)
                else:
                    os.environ["MIR_EXACT_BIG"] = old_big
        mir = env._env._mir
        in_step, loop = [], []
        for ep in range(episodes + 1):  # (the first episode warms up)
            obs, _ = env.reset(seed=ep)
            mir.exact_stats(reset=True)
            torch.cuda.synchronize(dev)
            t_in, ok = 0.0, None
            t0 = time.perf_counter()
            for stage in ex.STAGES:
                for _ in range(40):
                    a = ex.expert_policy(env.get_robot(), obs, stage)
                    ta = time.perf_counter()
                    obs, reward, term, trunc, info = env.step(a)
                    t_in += time.perf_counter() - ta
                    ok = term if ok is None else (ok | term)
            torch.cuda.synchronize(dev)
            if ep:
                in_step.append(t_in / 200 * 1e6); loop.append((time.perf_counter() - t0) / 200 * 1e6)
        st = mir.exact_stats()
        res[key] = {"env_step_us": sorted(in_step)[len(in_step) // 2], "loop_us": sorted(loop)[len(loop) // 2], "lifted_frac": float(ok.mean()),
                    "overflow_env_frac": st["overflow_env_steps"] / (200.0 * B), "overflow_step_frac": st["overflow_steps"] / 200.0}
        if key != "thinned":
            res[key].update(mir.exact_route())
        del env
    return res


def box_links_bench(torch, dev, steps: int = 400):
    """Secondary: the headline workload with links 1-7 of the Panda as round 1's boxes (GenesisEnv(..., link_shape="box")): the
    instantiation of the step kernel WITHOUT the convex narrowphase (the default scene has capsule links and runs the
    instantiation with GJK on the cores / MPR, mir_convex.h)."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, link_shape="box")
    env.reset(seed=0)
    task = env._env
    _bare_launches(task)
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.empty((256, B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
    for t in range(20):
        task.step_raw(acts[t])
    torch.cuda.synchronize(dev)
    ev0, ev1 = _events(torch)
    ev0.record()
    for t in range(steps):
        task.step_raw(acts[t % 256])
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    del env
    return {"workload": "CubePick-v0 robot=franka, links 1-7 as boxes (planes-and-boxes instantiation of the kernel), U(-1,1) joint targets, num_envs=4096",
            "env_steps_per_s": B / (us * 1e-6), "us_per_step": us}


def so101_bench(torch, dev, steps: int = 400):
    """Secondary (BASELINE configs[3] / SURVEY.md 8d config 4): SO-101 cube-pick (6 arm dofs, cube on the kitchen slab: box-box
    contact every step) at 4096 envs, U(-1,1) joint targets around the rest pose."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_pick", robot="so101", num_envs=B, enable_pixels=False)
    env.reset(seed=0)
    task = env._env
    was_exact = _bare_launches(task)
    gen = torch.Generator(device=dev).manual_seed(77)
    acts = torch.empty((256, B, task._zero.shape[1]), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
    for t in range(20):
        task.step_raw(acts[t])
    torch.cuda.synchronize(dev)
    ev0, ev1 = _events(torch)
    ev0.record()
    for t in range(steps):
        task.step_raw(acts[t % 256])
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    # the same through GenesisEnv.step (the task's default: exact contacts on)
    if was_exact:
        task._mir.set_exact_contacts(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(200):
        env.step(acts[t % 256])
    torch.cuda.synchronize(dev)
    api = 200 * B / (time.perf_counter() - t0)
    del env
    return {"workload": "CubePick-v0 robot=so101 (12 dofs, cube on the slab) state-only obs, U(-1,1) joint targets, num_envs=4096",
            "env_steps_per_s": B / (us * 1e-6), "us_per_step": us, "env_step_api_rate": api}


def stack_bench(torch, dev, steps: int = 300):
    """Secondary: gym_genesis/CubeStack-v0 (robot=franka: Panda x0.6 + five free cubes on the island slab, 39 dofs) at 4096
    envs on the wave-per-env kernel (mir_step64).  Fresh PD targets = home + U(-1,1) per step, resident in HBM; one fused
    launch per step; HIP events on the launching stream.  Algorithmic bytes per env-step, same accounting as SURVEY.md 8d:
    read qpos 44 + qvel 39 + action 9 + warm start 39 = 131 f32, write qpos 44 + qvel 39 + warm start 39 + obs 23 + reward 1
    = 146 f32 + 1 B mask => 1109 B."""
    from gym_genesis.env import GenesisEnv

    B = ENVS_PER_GPU
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B, enable_pixels=False)
    task = env._env
    task._mir.set_diag(True)
    env.reset(seed=0)
    gen = torch.Generator(device=dev).manual_seed(4321)
    acts = task._home[0] + torch.empty((256, B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
    for t in range(30):
        task.step_raw(acts[t])
    torch.cuda.synchronize(dev)
    ev0, ev1 = _events(torch)
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
        lo, hi = rollout_slice(i, 16, 256)
        task._mir.rollout(acts[lo:hi], rows)
    ev1.record()
    torch.cuda.synchronize(dev)
    us_ro = ev0.elapsed_time(ev1) * 1e3 / (8 * 16)
    algo = 1109.0
    achieved = algo * B / (us * 1e-6) / 1e9
    del env
    try:
        valu = valu_roofline(stack_flops_per_env_step(), us, "mir_step64_kernel", B, "sq_counters_step64.json")
    except Exception as e:  # noqa: BLE001
        valu = {"error": f"{type(e).__name__}: {e}"}
    return {"workload": "CubeStack-v0 robot=franka (39 dofs, 5 cubes) state-only obs, home + U(-1,1) joint targets, num_envs=4096",
            "env_steps_per_s": B / (us * 1e-6), "us_per_step": us, "rollout16_env_steps_per_s": B / (us_ro * 1e-6),
            "mean_contacts": ncon, "mean_newton_iterations": niter,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _profile_number("pmc_hbm_traffic_step64.json", "hbm_bytes_per_launch"), "kernel": "mir_step64_kernel",
                         "note": "1109 algorithmic B/env-step; one wave per env, 4 envs per CU: latency/occupancy-bound like the pick kernel"},
            "roofline_valu": valu}


def ik_bench(torch, dev, calls: int = 200):
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
    import ctypes as C
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    lib = task._mir.lib
    lib.mir_debug_ik_iters.argtypes = [C.c_void_p, C.c_void_p]
    lib.mir_debug_ik_iters(task._mir.h, C.c_void_p(iters.data_ptr()))
    for _ in range(5):
        q, err = task._mir.inverse_kinematics(hand, target, quat, return_error=True)
    torch.cuda.synchronize(dev)
    it = iters.cpu().numpy()
    lib.mir_debug_ik_iters(task._mir.h, None)
    ev0, ev1 = _events(torch)
    ev0.record()
    for _ in range(calls):
        task._mir.inverse_kinematics(hand, target, quat)
    ev1.record()
    torch.cuda.synchronize(dev)
    us = ev0.elapsed_time(ev1) * 1e3 / calls
    # fp32 vector roofline: ~1.7 kflop per env and iteration, counted once (10-link chain: forward kinematics 600, error 60, nine Jacobian
    # columns 380, J J^T 360, the 6 x 6 solve 150, J^T y 100, clamps 30) -- the kernel forms J J^T and solves it in EVERY lane of the env's
    # row, which is not counted; the launch lasts as long as its slowest wave (`iters_wave_max`), the chip idles behind the mean
    f_iter = 1.7e3
    flops = f_iter * float(it.sum())
    # ... and the call as the reference's expert makes it (examples/franka/pick_cube_state.py: one robot.inverse_kinematics per env.step,
    # envs_idx = arange(B), one quaternion expanded to the batch): HIP events around the wrapper, median per stage of its five stages
    expert_call_us = None
    try:
        import importlib.util

        spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
        ex = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ex)
        obs, _ = env.reset(seed=0)
        robot = env.get_robot()
        eef = robot.get_link("hand")
        q1 = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=dev).expand(B, -1)
        idx = torch.arange(B, device=dev)
        e0, e1 = _events(torch)
        expert_call_us = []
        for stage in ex.STAGES:
            dz = 0.115 if stage in ("hover", "stabilize") else (0.03 if stage == "grasp" else 0.25)
            us_stage = []
            for _ in range(40):
                tgt = obs["environment_state"][:, :3] + torch.tensor([0.0, 0.0, dz], device=dev)
                torch.cuda.synchronize(dev)
                e0.record()
                robot.inverse_kinematics(link=eef, pos=tgt, quat=q1, envs_idx=idx)
                e1.record()
                torch.cuda.synchronize(dev)
                us_stage.append(e0.elapsed_time(e1) * 1e3)
                obs, *_ = env.step(ex.expert_policy(robot, obs, stage))
            expert_call_us.append(round(sorted(us_stage)[len(us_stage) // 2], 1))
    except Exception as e:  # (a secondary figure: never the reason the leg fails)
        expert_call_us = f"failed: {type(e).__name__}: {e}"[:100]
    return {"workload": "robot.inverse_kinematics(hand, pos, quat) from the home pose, num_envs=4096, <= 20 Levenberg-Marquardt iterations",
            "expert_call_us_by_stage": expert_call_us,
            "env_solves_per_s": B / (us * 1e-6), "us_per_call": us, "converged_frac": float(((err[:, 0] < 5e-4) & (err[:, 1] < 5e-3)).float().mean().item()),
            "iters_mean": float(it.mean()), "iters_wave_max": int(it.max()),
            "roofline_valu": {"bound": "valu_fp32", "achieved": flops / (us * 1e-6) / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops / (us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS, "flop_per_env_iteration": f_iter}}


def batch_sweep(torch, dev, f_step=None, sizes=(1024, 4096, 16384, 32768, 65536), n: int = 300):
    """Secondary: what the 16-lane kernel does with other batches on ONE GPU -- the numbers under "latency / occupancy-bound".  Per batch:
    the bare fused launch (HIP events: kernel us, env-steps/s), the loop through GenesisEnv.step (wall clock), and the fp32 vector
    fraction F_step x B / kernel time / 157.3 TF.  At 4096 envs every env is resident at once (4 workgroups of 4 envs per CU); beyond
    that the launch runs in rounds, and the rate says what the SIMDs give when they are never short of waves."""
    from gym_genesis.env import GenesisEnv

    rows = []
    for Bs in sizes:
        env = GenesisEnv(task="cube_pick", robot="franka", num_envs=Bs, enable_pixels=False)
        task = env._env
        env.reset(seed=0)
        was_exact = _bare_launches(task)
        acts = torch.empty((8, Bs, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=torch.Generator(device=dev).manual_seed(7))
        al = list(acts.unbind(0))
        for t in range(30):
            task.step_raw(al[t % 8])
        torch.cuda.synchronize(dev)
        e0, e1 = _events(torch)
        e0.record()
        for t in range(n):
            task.step_raw(al[t % 8])
        e1.record()
        torch.cuda.synchronize(dev)
        k_us = e0.elapsed_time(e1) * 1e3 / n
        # the SINGLE-WAVE instantiation on the same batch: 16-step rollout launches run the step-loop instantiation --
        # one wave per workgroup does dynamics AND collision, 4 waves per CU instead of 8 -- with the state staying on the chip
        ro_rate = None
        try:
            rows16 = torch.zeros((8, Bs, 9 + 11 + 2), dtype=torch.float32, device=dev)
            task._mir.rollout(acts[:8], rows16)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(max(2, n // 40)):
                task._mir.rollout(acts[:8], rows16)
            e1.record()
            torch.cuda.synchronize(dev)
            ro_rate = max(2, n // 40) * 8 * Bs / (e0.elapsed_time(e1) * 1e-3)
            del rows16
        except Exception:  # noqa: BLE001
            pass
        if was_exact:
            task._mir.set_exact_contacts(True)
        env.reset(seed=0)
        for t in range(30):
            env.step(al[t % 8])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in range(n):
            env.step(al[t % 8])
        torch.cuda.synchronize(dev)
        api_us = (time.perf_counter() - t0) * 1e6 / n
        rows.append([Bs, k_us, Bs / (k_us * 1e-6), api_us, Bs / (api_us * 1e-6), ALGO_BYTES_PER_ENV_STEP * Bs / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     (f_step * Bs / (k_us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS) if f_step else None, ro_rate])
        del env, task
    return {"cols": ["num_envs", "kernel_us", "bare_launch_rate", "env_step_us", "env_step_rate", "hbm_frac", "valu_frac", "single_wave_rollout8_rate"], "rows": rows,
            "note": "fused launch by HIP events (gap included), GenesisEnv.step by wall clock; one GPU, franka pick, U(-1,1) targets"}


def _profile_number(name: str, key: str):
    """A number measured offline with rocprofv3 PMC passes and committed under profiles/ (latest round first)."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, name)) as f:
                v = json.load(f).get(key)
            if v is not None:
                return v
        except (OSError, ValueError):
            continue
    return None


def _guard(out: dict, key: str, fn, *a, **kw):
    """Run one secondary leg; a failure becomes {"error": ...} under its key and never touches the headline."""
    try:
        out[key] = fn(*a, **kw)
    except BaseException as e:  # noqa: BLE001 (KeyboardInterrupt included: the line must still be printed)
        out[key] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc(limit=3)}
        if isinstance(e, KeyboardInterrupt):
            raise


# ---- the printed line ------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000   # bytes: the driver keeps 8 KB tails of stdout; a longer line loses its HEAD (value, config) in the driver's record
NOTE_LIMIT = 120    # characters per string (the prose lives in DESIGN.md)
HEAD_KEYS = ("metric", "value", "unit", "value_median_region", "hot_path_rate", "api_over_hot_path", "early_mask_sent", "early_mask_mismatches",
             "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
# what goes first when the line is still too long (least important first; dotted = nested key)
DROP_ORDER = ("roofline_valu.instruction_level", "roofline_valu.fused_launch", "stack.roofline_valu.instruction_level", "sync_step_floor", "repeat_us_per_step",
              "roofline.note", "roofline_fused_launch", "stack.roofline_valu", "pixels.reduced_96x128_num_envs_4096", "pixels.us_per_render_regions",
              "config.obs_gather", "config.output_ring", "config.value_is", "no_gather", "best_repeat_value", "roofline_valu", "pixels.roofline", "stack.roofline",
              "box_links", "ik", "so101_pick", "pixels", "stack", "secondary", "scripted_grasp", "ref_expert")


def compact_line(out: dict, limit: int = LINE_LIMIT) -> dict:
    """The dict that is printed: the contract keys and the figures the driver's record must show first (value, its median region, the
    bare-launch rate, the early-mask counters -- flat, right behind `value`, and once more inside `config`), every string cut to
    NOTE_LIMIT characters, traces and per-kind tables left to the full record (bench_line_full.json), and -- should the line still
    exceed `limit` bytes -- secondary detail dropped in DROP_ORDER.  Never touches the contract keys."""
    import copy
    o = copy.deepcopy(out)
    em = o.get("early_mask") if isinstance(o.get("early_mask"), dict) else {}
    o["early_mask_sent"], o["early_mask_mismatches"] = em.get("sent"), em.get("mismatches")
    if isinstance(o.get("config"), dict):
        o["config"]["early_mask"] = {"sent": em.get("sent"), "mismatches": em.get("mismatches"), "workgroup_launches": em.get("workgroup_launches")}
        for k in ("value_median_region", "hot_path_rate", "api_over_hot_path"):
            o["config"][k] = o.get(k)
    o.pop("early_mask", None)

    def shrink(x, key="", leg=False):
        # (inside the secondary legs the prose goes altogether -- workloads and notes are in the docstrings and in DESIGN.md -- and
        #  numbers keep five digits)
        if isinstance(x, dict):
            gone = ("trace", "per_kind", "F_step_per_kind", "F_step_source", "runs_in_order") + (("workload", "note", "source", "dtype", "sample") if leg else ())
            return {k: shrink(v, k, leg) for k, v in x.items() if k not in gone}
        if isinstance(x, list):
            return [shrink(v, key, leg) for v in x]
        if isinstance(x, float):
            return float(f"{x:.5g}" if leg else f"{x:.7g}")
        if isinstance(x, str) and len(x) > NOTE_LIMIT and key != "metric":
            return x[:NOTE_LIMIT - 3] + "..."
        return x

    LEGS = ("secondary", "pixels", "scripted_grasp", "ref_expert", "box_links", "so101_pick", "stack", "ik", "roofline_valu", "roofline_fused_launch", "sync_step_floor",
            "no_gather", "gather_every_1", "repeat_us_per_step")
    o = {k: shrink(v, k, k in LEGS) for k, v in o.items()}
    o = {**{k: o[k] for k in HEAD_KEYS if k in o}, **{k: v for k, v in o.items() if k not in HEAD_KEYS}}
    dropped = []
    for path in DROP_ORDER:
        if len(json.dumps(o)) <= limit - 40 * (len(dropped) + 1):   # (room for the list of what was dropped)
            break
        node, parts = o, path.split(".")
        for part in parts[:-1]:
            node = node.get(part) if isinstance(node, dict) else None
        if isinstance(node, dict) and parts[-1] in node:
            del node[parts[-1]]
            dropped.append(path)
    if dropped:
        o["dropped_for_length"] = dropped
    return o


def print_line(out: dict) -> None:
    """Rank 0: the full record to gpurun_out/bench_line_full.json (best effort), the compact line to stdout as the LAST line."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_line_full.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    try:
        line = json.dumps(compact_line(out))
    except Exception:  # noqa: BLE001 (the line must be printed whatever the compaction runs into)
        line = json.dumps(out)
    print(line, flush=True)


# ---- worker ----------------------------------------------------------------------------------------------------------
def selftest_worker(args) -> int:
    """--selftest-launch: the ranks rendezvous over gloo on CPU, agree on a sum, rank 0 prints a line.  No GPU, no physics."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "requested_gpus": args.gpus, "rank_sum": float(t.item())}), flush=True)
    return 0 if world == args.gpus else 1


def worker(args) -> int:
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ---- host placement, part 1 (N > 1; BEFORE the first GPU call, so that every thread the HIP runtime, RCCL and torch create
    # starts inside the rank's own cores): the cores this process may use that sit on the NUMA node of the rank's GPU, split among
    # the ranks on that node (rank_cpu_plan).  MIR_BENCH_PIN=0 leaves everything to the scheduler.
    pin_on = os.environ.get("MIR_BENCH_PIN", "1") == "1"
    try:
        free_cpus = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        free_cpus, pin_on = None, False
    my_cpus, spin_cpu, gpu_numa = [], None, None
    if world > 1 and pin_on:
        try:
            gpu_numa = gpu_numa_nodes()
            my_cpus, spin_cpu = rank_cpu_plan(local_rank, world, free_cpus, gpu_numa, numa_cpu_lists())
            if my_cpus:
                os.sched_setaffinity(0, my_cpus)
        except Exception:  # noqa: BLE001
            my_cpus, spin_cpu = [], None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not args.oversubscribe:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {ndev} GPU(s) visible (one rank per GPU)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    placement = {"pinned_cpu": None, "helpers_moved": 0}

    def pin_stepping_thread():
        """Host placement, part 2 (called once everything that creates threads has run: process group, scene, first steps): the
        stepping thread gets a core to itself.  env.step is a host / device ping-pong of ~25 us, and every migration of the host
        thread costs it a cold cache in the middle of one (single rank, same box, five runs each, driver command: 159.5 M free,
        162.1 M pinned; tools/probes/ab_pin.sh).  One rank: the core it is on.  N ranks: the first core of the rank's slice, and
        every other thread of the process (HIP runtime, RCCL proxy) is moved to the rest of the slice -- the stepping thread
        busy-waits in mir_step_end, and a helper thread scheduled behind it would wait for a time slice."""
        if not pin_on:
            return
        try:
            if world == 1:
                import ctypes
                cpu = ctypes.CDLL(None).sched_getcpu()
                if cpu >= 0:
                    os.sched_setaffinity(0, {cpu})
                    placement["pinned_cpu"] = cpu
            elif spin_cpu is not None:
                import threading
                rest = set(my_cpus) - {spin_cpu}
                if rest:
                    placement["helpers_moved"] = move_other_threads(rest, threading.get_native_id())
                os.sched_setaffinity(0, {spin_cpu})
                placement["pinned_cpu"] = spin_cpu
        except Exception:  # noqa: BLE001
            placement["pinned_cpu"] = None

    def unpin():
        """Back to the affinity the process started with (the CPU baseline spreads over every core it may use)."""
        if free_cpus:
            # EVERY thread of the process (an OpenMP pool or a runtime helper that already exists keeps the mask it was created
            # with or moved to -- the rank's slice -- while `cores` below reports the whole machine)
            try:
                tids = [int(t) for t in os.listdir("/proc/self/task")]
            except OSError:
                tids = [0]
            for tid in tids + [0]:
                try:
                    os.sched_setaffinity(tid, free_cpus)
                except Exception:  # noqa: BLE001
                    pass

    def host_thread_note() -> str:
        if placement["pinned_cpu"] is None:
            return "not pinned"
        if world == 1:
            return "pinned to cpu %d for the headline and raw loops" % placement["pinned_cpu"]
        numa = gpu_numa[local_rank] if gpu_numa and local_rank < len(gpu_numa) else None
        return ("rank 0 of %d: stepping thread on cpu %d, %d helper threads on the other %d cores of its slice %s (NUMA node of its GPU: %s)"
                % (world, placement["pinned_cpu"], placement["helpers_moved"], max(0, len(my_cpus) - 1),
                   "%d-%d" % (my_cpus[0], my_cpus[-1]) if my_cpus else "-", "unknown" if numa is None else numa))

    use_pg = world > 1 or args.force_gather
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    pg_world = dist.get_world_size() if use_pg else 1

    from gym_genesis.env import GenesisEnv

    B = args.envs_per_gpu
    K, W = max(1, args.steps), max(0, args.warmup)
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B * world, enable_pixels=False, shard=(rank, world))
    task = env._env
    env.reset(seed=0)

    # inputs resident in HBM before the timed region: N_ACT pre-drawn action batches, one per step, cycled
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    actions = torch.empty((N_ACT, B, 9), dtype=torch.float32, device=dev)
    actions.uniform_(-1.0, 1.0, generator=gen)
    act_list = list(actions.unbind(0))  # one (B, 9) device tensor per pre-drawn step: no view is built inside the timed loop

    for t in range(4):  # (first launches: lazily created runtime threads exist before the stepping thread is pinned)
        env.step(act_list[t])
    env.reset(seed=0)
    torch.cuda.synchronize(dev)
    pin_stepping_thread()

    gather = use_pg and not args.no_gather
    gather_on = [gather]  # (switched off for the A/B region that gives gather_overhead_us)
    S = max(1, args.gather_every)
    flat = B * 21  # agent_pos 9 + environment_state 11 + reward 1 per env (terminated == (reward == 1), env.py:63)
    gathered = [torch.empty((pg_world * S * flat,), dtype=torch.float32, device=dev) for _ in range(2)] if gather else None
    pending = [None, None]
    copy_gather, gather_note = None, ""
    if gather and args.gather_path == "copy":
        from gym_genesis.sharding import make_copy_gather
        copy_gather, gather_note = make_copy_gather(S * flat, dev)
    last_seq = [0]
    check_gather = [False]
    # the step kernels write straight into the gather's send buffer: the outputs of consecutive steps are consecutive rows of one
    # ring (4 chunks of S steps: two in flight, one being filled, one spare), a chunk = S contiguous rows, no concatenation
    ring = task._mir.use_output_ring(4 * S, 9, 11) if gather else None   # (RING_CHUNKS x S rows)
    if not gather and args.output_ring > 0:
        task._mir.use_output_ring(args.output_ring, 9, 11)
    RING_CHUNKS = 4
    RING_ROWS = RING_CHUNKS * S
    # (how many consecutive steps travel in one gather: S, the --gather-every of the headline; 1 in the `gather_every_1` regions --
    #  SURVEY.md cfg 3's "all-gather obs each step" -- where a ring chunk is a single row; and which path carries it: [0] = the copy
    #  path's object or None for the RCCL collective)
    Sx, cg = [S], [copy_gather]
    # `sent`: first row of the current ring chunk that has not been gathered yet, `row`: the row the latest step wrote (-1: none)
    state = {"chunk": 0, "t": 0, "resets": 0, "sent": 0, "row": -1}
    last_block = [None]
    readers = [[] for _ in range(RING_ROWS)]   # per ring chunk (of Sx[0] rows): the pushes that read it since it was last written
    gather_stats = {"pushes": 0, "partial_pushes": 0, "lag_max": 0, "checks": 0, "checks_failed": 0}

    def flush():
        """Gather the rows written since the last flush (a whole chunk of S steps in the loop; whatever there is at the end of a
        timed region) -- asynchronously: the transfer overlaps the following steps."""
        row, lo = state["row"], state["sent"]
        if not gather_on[0] or row < lo:
            return
        send = ring[lo:row + 1].reshape(-1)   # rows of one ring chunk: one contiguous block, no concatenation
        Sc = Sx[0]
        if cg[0] is not None:
            last_seq[0] = cg[0].push(send)   # device-to-device copies on the side streams: nothing to wait for here
            ch = lo // Sc
            readers[ch].append(last_seq[0])
            if (row + 1) % Sc == 0:
                # the chunk is complete and the step kernels are about to write the NEXT ring chunk: they wait (on the device, the
                # host goes on) until the copies of the pushes that read it on the previous lap are through with it
                nxt = (ch + 1) % (RING_ROWS // Sc)
                for sq in readers[nxt]:
                    cg[0].wait_source(sq)
                readers[nxt].clear()
            gather_stats["lag_max"] = max(gather_stats["lag_max"], cg[0].lag())
        else:
            s = state["chunk"] & 1
            if pending[s] is not None:
                pending[s].wait()
            pending[s] = dist.all_gather_into_tensor(gathered[s][:pg_world * send.numel()], send, async_op=True)
        last_block[0] = send
        gather_stats["pushes"] += 1
        gather_stats["partial_pushes"] += 1 if (row + 1 - lo) < Sc else 0
        state["chunk"] += 1
        # (never wrapped here: RING_ROWS = "everything up to the end of the ring has gone, nothing is pending"; the step that writes row 0
        #  starts a new chunk and sets it -- a wrapped 0 made the closing flush of a region that ended on the ring's last row send the ring)
        state["sent"] = row + 1

    def api_loop(k: int):
        """k iterations of the README loop through GenesisEnv.step (README.md:32-43)."""
        t, step, n, Sc = state["t"], env.step, N_ACT, Sx[0]
        for _ in range(k):
            if kill_at is not None and t >= kill_at:
                os._exit(17)   # (tests/test_gpu_bench.py: a rank that dies in the middle of a timed region -- MIR_BENCH_KILL_RANK="rank:step")
            obs, reward, terminated, truncated, info = step(act_list[t % n])
            if gather_on[0]:
                row = obs["agent_pos"].storage_offset() // flat   # the ring row this step's kernel wrote
                if row % Sc == 0 or row != (state["row"] + 1) % RING_ROWS:
                    state["sent"] = row                           # a new chunk (or the rows in between were not part of a gather)
                state["row"] = row
                if row % Sc == Sc - 1:                            # a chunk is complete: its rows are one contiguous block
                    flush()
            t += 1
            if terminated.any() or truncated.any() or t % EPISODE_STEPS == 0:
                env.reset()
                state["resets"] += 1
        state["t"] = t

    kill_at = None
    if os.environ.get("MIR_BENCH_KILL_RANK"):
        kr, _, ks = os.environ["MIR_BENCH_KILL_RANK"].partition(":")
        if int(kr) == rank:
            kill_at = int(ks or 0)

    def raw_loop(k: int):
        """k bare fused launches into persistent buffers (no host hand-over, no reset: every launch in the bracket is one
        mir_step_kernel<0>, so HIP-event time / launches is that kernel's average duration including the launch gap)."""
        t, step, n = state["t"], task.step_raw, N_ACT
        for _ in range(k):
            step(act_list[t % n])
            t += 1
        state["t"] = t

    def sync_all():
        flush()
        for i, p in enumerate(pending):
            if p is not None:
                p.wait()
                pending[i] = None
        if cg[0] is not None and last_seq[0]:
            cg[0].wait(last_seq[0])   # every rank's last block has landed HERE (the words of the earlier ones came first)
        torch.cuda.synchronize(dev)
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)
        if cg[0] is not None and last_seq[0] and last_block[0] is not None and check_gather[0]:
            # (outside the timed brackets' clock?  No: sync_all is the bracket.  The check is therefore switched on only for the
            #  verification pass after the measurements, see below)
            gather_stats["checks"] += 1
            if not cg[0].check_against_collective(last_seq[0], last_block[0]):
                gather_stats["checks_failed"] += 1

    def max_over_ranks(x: float) -> float:
        if not use_pg:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(loop_fn, k: int, events: bool):
        """Exactly k steps bracketed by barrier + synchronize on both sides; (wall seconds max over ranks, HIP-event ms or 0).
        HIP events are recorded only around the raw launch loop (they give the kernel's duration there); the API loop is
        timed by the wall clock alone, so nothing of this harness sits inside its bracket."""
        sync_all()
        if events:
            ev0, ev1 = _events(torch)
            t0 = time.perf_counter()
            ev0.record()
            loop_fn(k)
            ev1.record()
        else:
            t0 = time.perf_counter()
            loop_fn(k)
        sync_all()
        wall = time.perf_counter() - t0
        return max_over_ranks(wall), (ev0.elapsed_time(ev1) if events else 0.0)

    def measure(loop_fn):
        """W warm-up steps, then the K-step timed region, repeated until --min-time seconds are measured."""
        events = loop_fn is raw_loop
        loop_fn(W)
        first, ev_ms = timed(loop_fn, K, events)
        walls, evs = [first], [ev_ms]
        # (the regions are repeated until --min-time seconds are MEASURED, not a number of times derived from the first region: one
        #  slow first region -- a stall of the shared host -- would otherwise cut the whole measurement short.  Every wall value is
        #  already the maximum over the ranks, so all ranks take the same decisions.)
        max_reps = repeats_for(0.0, args.min_time)
        total = first
        while total < args.min_time and len(walls) < max_reps:
            w, e = timed(loop_fn, K, events)
            walls.append(w)
            evs.append(e)
            total += w
        return walls, evs

    out = None
    rc = 0
    try:
        # ---- headline: the loop through GenesisEnv.step ---------------------------------------------------------------
        if args.core_only and args.raw_only:
            _bare_launches(task)   # (bare launches only: exact contacts off, see below)
            walls, evs = measure(raw_loop)
            api_walls = None
        else:
            task._mir.early_mask_stats(reset=True)
            api_walls, _ = measure(api_loop)
            walls = api_walls
            if copy_gather is not None:
                # one more region, NOT part of the measurement: its last gather is compared with all_gather_into_tensor of the same
                # block on every rank (a torn or lagging copy path shows here, not only in the start-up verification)
                check_gather[0] = True
                timed(api_loop, K, False)
                check_gather[0] = False
        # the early `terminated` bytes of exactly these launches (warm-up included): workgroup-launches that sent their bytes from inside the
        # solver loop, and how many of those the kernel's own check against the integrated state found wrong (a non-zero count would
        # also have failed the loop with MIR_E_MASK, include/mirigid.h: mir_step_begin)
        em_sent, em_bad = task._mir.early_mask_stats()
        total_wall = sum(walls)
        value = len(walls) * K * B * world / total_wall
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": total_wall * 1e3 / (len(walls) * K),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "CubePick-v0 franka, 4096 envs/GPU, state obs, U(-1,1) targets in HBM, README loop via GenesisEnv.step, reset / 200 steps",
                       "num_envs_per_gpu": B, "global_num_envs": B * world, "parallelism": f"env-axis shard x{world}",
                       "world_size_observed": pg_world, "dist_backend": args.dist_backend if use_pg else None,
                       "obs_gather": ("none" if not gather else
                                      (f"copy path: [agent_pos|environment_state|reward] of {S} steps pushed to every rank by peer-to-peer device "
                                       f"copies + sequence words (no collective kernel), process group {args.dist_backend}") if copy_gather is not None else
                                      (f"{args.dist_backend} all_gather of [agent_pos|environment_state|reward] of {S} steps per collective, "
                                       "overlapped with the following steps" + (f" (copy path not used: {gather_note})" if gather_note else ""))),
                       "gather_path": None if not gather else ("copy" if copy_gather is not None else "rccl"),
                       "gather_note": (gather_note[:110] or None) if gather else None,   # (why the copy path was not taken, when it was asked for)
                       "output_ring": (f"step outputs are rows of a reusable ring of {RING_CHUNKS * S} steps (the gather's send buffer): a step's "
                                       "tensors are overwritten that many steps later; fresh tensors per step without a gather") if gather else
                                      (f"ring of {args.output_ring} steps (--output-ring)" if args.output_ring > 0 else "none: fresh tensors every step"),
                       "gather_stats": dict(gather_stats) if gather else None,
                       "terminated_sync_mode": task._mir.sync_mode, "split_step": int(getattr(task._mir, "split_step", 0)),
                       "early_terminated_bytes": bool(task._mir.early_mask),
                       "exact_contacts": bool(getattr(task._mir, "exact_contacts", False)),
                       "value_is": "mean over the repeated K-step regions (value_median_region: their median)",
                       "host_thread": host_thread_note()},
            "early_mask": {"sent": em_sent, "mismatches": em_bad, "workgroup_launches": (len(walls) * K + W) * ((B + 3) // 4) if api_walls is not None else 0,
                           "note": "rank 0; counted by the kernel over the headline loop's launches (warm-up included; the launches of its resets are "
                                   "plain steps and send nothing to the host)"},
            "repeats": len(walls),
            "timed_steps_total": len(walls) * K,
            "timed_seconds_total": total_wall,
            "best_repeat_value": K * B * world / min(walls),
            # `value` is the MEAN over the repeated K-step regions (resets and whatever else the shared host does included); the
            # median region is the figure that reproduces from run to run
            "value_median_region": K * B * world / sorted(walls)[len(walls) // 2],
            # spread of the repeated K-step regions (us per step): the headline is their mean, resets included
            "repeat_us_per_step": {"min": min(walls) * 1e6 / K, "median": sorted(walls)[len(walls) // 2] * 1e6 / K,
                                   "p95": sorted(walls)[min(len(walls) - 1, (95 * len(walls)) // 100)] * 1e6 / K, "max": max(walls) * 1e6 / K},
            "resets_in_loop": state["resets"],
            "path": "GenesisEnv.step" if api_walls is not None else "task.step_raw (--raw-only)",
        }
        # ---- what the observation gather costs: the same loop once more with the gather switched off (all ranks take this branch
        # together: `gather` and the flag are the same everywhere) ---------------------------------------------------------
        if gather and api_walls is not None and not args.no_gather_ab:
            try:
                sync_all()
                gather_on[0] = False
                ng_walls, _ = measure(api_loop)
                gather_on[0] = True
                us_g = sum(walls) * 1e6 / (len(walls) * K)          # (means over all regions: every region ends with a push)
                us_ng = sum(ng_walls) * 1e6 / (len(ng_walls) * K)
                out["gather_overhead_us"] = us_g - us_ng
                out["no_gather"] = {"value": len(ng_walls) * K * B * world / sum(ng_walls), "value_median_region": K * B * world / sorted(ng_walls)[len(ng_walls) // 2],
                                    "mean_us_per_step": us_ng, "mean_us_per_step_with_gather": us_g, "repeats": len(ng_walls),
                                    "note": "the headline loop with the observation gather switched off, measured right after the headline; "
                                            "gather_overhead_us = difference of the mean us per step"}
            except Exception as e:  # noqa: BLE001
                gather_on[0] = True
                out["gather_overhead_us"] = None
                out["no_gather"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- SURVEY.md cfg 3 to the letter: "all-gather obs each step".  The headline loop once more with ONE step per gather, over
        # the copy path (when it is up) and over the collective (all ranks take these branches together) ------------------------------
        if gather and api_walls is not None and not args.no_gather_ab and S > 1:
            g1 = {}
            for name, obj in (("copy", copy_gather), ("rccl", None)):
                if name == "copy" and copy_gather is None:
                    continue
                try:
                    sync_all()
                    for lst in readers:
                        lst.clear()
                    state["sent"], state["row"] = 0, -1
                    Sx[0], cg[0] = 1, obj
                    w1, _ = measure(api_loop)
                    us1 = sum(w1) * 1e6 / (len(w1) * K)
                    g1[name] = {"value": len(w1) * K * B * world / sum(w1), "mean_us_per_step": us1,
                                "overhead_us_vs_no_gather": (us1 - out["no_gather"]["mean_us_per_step"]) if "mean_us_per_step" in out.get("no_gather", {}) else None}
                except Exception as e:  # noqa: BLE001
                    g1[name] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    sync_all()
                    for lst in readers:
                        lst.clear()
                    state["sent"], state["row"] = 0, -1
                    Sx[0], cg[0] = S, copy_gather
            out["gather_every_1"] = g1
        # ---- the bare fused launch over the same number of steps: hot_path_rate + kernel duration for the roofline ----
        try:
            state["t"] = 0
            task.reset()
            # (what follows -- bare fused launches, device-side rollouts, the kernel timing of the roofline -- has no host in it to close a
            #  step on: those legs run with exact contacts switched OFF, at the 16-point capacity; mir_rollout* / mir_step_packed are
            #  refused with the switch on.  The headline loop above ran with the task's default: on.)
            if getattr(task._mir, "exact_contacts", False):
                out["config"]["exact_contacts_headline_loop"] = dict(task._mir.exact_stats(), **task._mir.exact_route())
                task._mir.set_exact_contacts(False)
            raw_walls, raw_evs = (walls, evs) if api_walls is None else measure(raw_loop)
            n_launch = len(raw_walls) * K
            kernel_us = sum(raw_evs) * 1e3 / n_launch  # HIP events on the launching stream, per launch
            achieved = ALGO_BYTES_PER_ENV_STEP * B / (kernel_us * 1e-6) / 1e9
            out["hot_path_rate"] = n_launch * B * world / sum(raw_walls)
            out["api_over_hot_path"] = out["value"] / out["hot_path_rate"]
            # what a synchronous step cannot go below on this machine: the kernel + one empty-kernel launch/completion round trip
            try:
                null_us = task._mir.null_roundtrip_us(2000)
                out["sync_step_floor"] = {"null_launch_roundtrip_us": null_us, "kernel_us": kernel_us,
                                          "floor_us_per_step": kernel_us + null_us, "measured_us_per_step": out["ms_per_step"] * 1e3,
                                          "note": "env.step must hand a NumPy `terminated` to the host every step (env.py:64), so the next launch "
                                                  "cannot be queued before that mask exists.  The kernel sends the mask bytes from inside its solver loop, as soon as a convexity bound says the object's height cannot reach the threshold any more (same mask, checked against the integrated state; csrc/mir_model.h: term_bound_ok), each workgroup into its own 64-byte line of pinned host memory: the host is back with the next launch while this one is still running, and the measured step is the kernel plus the gap between two dependent launches -- under this floor, which prices a launch + completion round trip per step"}
            except Exception as e:  # noqa: BLE001
                out["sync_step_floor"] = {"error": f"{type(e).__name__}: {e}"}
            # (the instantiation rocprofv3 lists: FEAT = 1 round geoms | 4 the headline scene's sizes as literals, csrc/mir_spec_pick.h)
            feat = 5 if getattr(task._mir, "spec_active", False) else 1
            fused = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _profile_number("pmc_hbm_traffic.json", "hbm_bytes_per_launch") if B == ENVS_PER_GPU else None,
                     "kernel": f"mir_step_kernel<0, {feat}>", "kernel_us": kernel_us, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * B,
                     "note": "489 algorithmic B/env-step x 4096 envs per launch (SURVEY.md 8d); kernel_us = HIP events over the "
                             "back-to-back raw launches of the same K-step region; the path is latency/occupancy-bound, not HBM-bound"}
            out["roofline"] = fused
            # The launches of the HEADLINE loop are a different instantiation when the split step is on (mir_get_split_step): the
            # rotated kernel, second half of this step + first half of the next.  Its duration: HIP events on the launching stream
            # around every launch of a begin / end loop of its own (the timed loop itself carries no events).
            split = int(getattr(task._mir, "split_step", 0))
            if split == 1 and api_walls is not None:
                try:
                    mir = task._mir
                    task.reset()
                    # (the launches write the four outputs of a GenesisEnv.step launch; not its host-visible terminated bytes, which
                    # back to back and with nobody reading them time the PCIe write path instead of the kernel)
                    outs_api = (mir.empty(9), mir.empty(11), mir.empty(), mir.empty(dtype=torch.uint8))
                    mir.rotated_launches(actions, 50, outputs=outs_api)
                    torch.cuda.synchronize(dev)
                    n_ev = 1000
                    e0, e1 = _events(torch)

                    def timed(outputs):
                        e0.record()
                        mir.rotated_launches(actions, n_ev, outputs=outputs)
                        e1.record()
                        torch.cuda.synchronize(dev)
                        return e0.elapsed_time(e1) * 1e3 / n_ev

                    # (three regions each, the median: the first region of the first process on a fresh machine has come out at twice
                    # the time of every later one)
                    rot_all = sorted(timed(outs_api) for _ in range(3))
                    rot_us = rot_all[1]
                    rot_us_bare = sorted(timed(None) for _ in range(3))[1]
                    ach = ALGO_BYTES_PER_ENV_STEP * B / (rot_us * 1e-6) / 1e9
                    out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                       "traffic": _profile_number("pmc_hbm_traffic_api.json", "hbm_bytes_per_launch") if B == ENVS_PER_GPU else None,
                                       "kernel": f"mir_step_kernel<5, {feat}>", "kernel_us": rot_us, "kernel_us_regions": rot_all, "kernel_us_without_outputs": rot_us_bare,
                                       "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * B,
                                       "note": "the kernel of the headline loop: the rotated launch of GenesisEnv.step (this step's action-dependent "
                                               "half, then the next step's action-independent half through a 4.7 KB/env scratch row (~2.4 KB used), which is why its "
                                               "traffic is several times the 489 algorithmic B/env-step: bytes spent to take ~7 us of work out of the "
                                               "host-visible latency); kernel_us = HIP events around 1000 back-to-back launches (mir_debug_rotated_launches) that write the "
                                               "same device outputs as GenesisEnv.step's, launch gap included; median of three such regions"}
                    out["roofline_fused_launch"] = fused
                except Exception as e:  # noqa: BLE001
                    out["roofline_api_kernel_error"] = f"{type(e).__name__}: {e}"
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            rc = 1

        unpin()
        if rank == 0 and not args.core_only and isinstance(out.get("roofline"), dict) and "kernel_us" in out["roofline"]:
            def _valu():
                fl = flops_per_env_step()
                rf = out["roofline"]
                rot = rf["kernel"].startswith("mir_step_kernel<5")
                res = valu_roofline(fl, rf["kernel_us"], rf["kernel"], B, "sq_counters_rotated.json" if rot else "sq_counters.json")
                if "roofline_fused_launch" in out:
                    ff = out["roofline_fused_launch"]
                    res["fused_launch"] = {k: v for k, v in valu_roofline(fl, ff["kernel_us"], ff["kernel"], B, "sq_counters.json").items()
                                           if k in ("achieved", "frac", "kernel", "kernel_us", "instruction_level")}
                return res
            _guard(out, "roofline_valu", _valu)
        if rank == 0 and world == 1 and not args.core_only:
            _guard(out, "secondary", lambda: secondary_rates(torch, dev, env, task, actions, B))
            if isinstance(out.get("secondary"), dict) and "error" not in out["secondary"]:
                fs = out.get("roofline_valu", {}).get("F_step") if isinstance(out.get("roofline_valu"), dict) else None
                _guard(out["secondary"], "batch_sweep", batch_sweep, torch, dev, fs)
            if not args.no_pixels:
                _guard(out, "pixels", pixels_bench, torch, dev)
            if not args.no_stack:
                _guard(out, "scripted_grasp", grasp_bench, torch, dev)
                _guard(out, "box_links", box_links_bench, torch, dev)
                _guard(out, "so101_pick", so101_bench, torch, dev)
                _guard(out, "stack", stack_bench, torch, dev)
                _guard(out, "ik", ik_bench, torch, dev)
                _guard(out, "ref_expert", ref_expert_bench, torch, dev)
        # (rank 0 of any world size: the other ranks wait at the final barrier, their host cores idle)
        if rank == 0 and not (args.no_cpu_baseline or args.core_only):
            _guard(out, "cpu_baseline", cpu_baseline)
    except BaseException as e:  # noqa: BLE001
        rc = 1
        if out is None:
            out = {"metric": METRIC, "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
                   "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc(limit=6)}
        else:
            out["error_after_headline"] = f"{type(e).__name__}: {e}"
    finally:
        try:
            if use_pg:
                dist.barrier()
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        if rank == 0:
            # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            sys.stdout.flush()
            print_line(out)
    return rc


def secondary_rates(torch, dev, env, task, actions, B):
    """Other ways through the same kernel, N = 1 only: host-NumPy actions (README.md:34), physics only (mir_step), the
    device-resident episode loop and K-step rollout launches (mir_rollout)."""
    import numpy as np
    res = {}
    n = 200
    env.reset(seed=0)
    acts_np = np.random.default_rng(5).uniform(-1, 1, (8, B, 9)).astype(np.float32)
    for t in range(50):  # (the first call pins the two staging buffers: milliseconds)
        env.step(acts_np[t % 8])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(5 * n):
        env.step(acts_np[t % 8])
    torch.cuda.synchronize(dev)
    res["env_step_api_numpy_actions_rate"] = 5 * n * B / (time.perf_counter() - t0)

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(n):
        task._mir.step(1)
    torch.cuda.synchronize(dev)
    res["physics_only_rate"] = n * B / (time.perf_counter() - t0)

    task.enable_autoreset(max_episode_steps=EPISODE_STEPS)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(n):
        task.step_autoreset(actions[action_index(t)])
    torch.cuda.synchronize(dev)
    res["device_autoreset_loop_rate"] = n * B / (time.perf_counter() - t0)

    rows = torch.zeros((RK, B, ROW_STRIDE), dtype=torch.float32, device=dev)
    nro = 12
    task.reset()
    for i in range(2):
        lo, hi = rollout_slice(i)
        task._mir.rollout(actions[lo:hi], rows)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(nro):
        lo, hi = rollout_slice(i)
        task._mir.rollout(actions[lo:hi], rows)
    torch.cuda.synchronize(dev)
    res["device_rollout16_rate"] = nro * RK * B / (time.perf_counter() - t0)

    task.reset()
    task._episode_len.zero_()
    lo, hi = rollout_slice(0)
    task.rollout_autoreset(actions[lo:hi], rows)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(nro):
        lo, hi = rollout_slice(i)
        task.rollout_autoreset(actions[lo:hi], rows)
    torch.cuda.synchronize(dev)
    res["device_autoreset_rollout16_rate"] = nro * RK * B / (time.perf_counter() - t0)
    return res


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.cpu_baseline_child is not None:
        print(json.dumps(cpu_baseline_child(args.cpu_baseline_child)), flush=True)
        return 0
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, argv)
    if args.selftest_launch:
        return selftest_worker(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
