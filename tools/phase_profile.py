"""Developer tool: phase-level shader-clock deltas of the fused step kernel (block 0) and the
kernel's launch-to-launch time as a function of the batch size.  Needs a GPU."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
from gym_genesis.backend import models  # noqa: E402
from gym_genesis.backend.lib import MirScene  # noqa: E402

PHASES = ["fk", "cdof+cinert+vel+crb", "rne+M+bias", "gj smooth solve", "collide", "rows (Jb, limits, aref)", "newton init",
          "newton iterations", "integrate", "final fk"]


def setup(B, warm=30):
    spec = models.franka_cube_pick_scene().build()
    sc = MirScene(spec, B)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
    g = torch.Generator(device=sc.device).manual_seed(1)
    acts = torch.empty((64, B, 9), device=sc.device).uniform_(-1, 1, generator=g)
    bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(warm):
        sc.step_fused(acts[t % 64], *bufs)
    torch.cuda.synchronize()
    return sc, acts, bufs


def setup_grasp(B):
    """State in the middle of the scripted pick (fingers closed on the cube: ~12 contacts per env, arm-cube coupling)."""
    from gym_genesis.env import GenesisEnv
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    task = env._env
    dev = task.device
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    q_prev = None
    for dz, grip, n in ((0.25, 0.04, 40), (0.104, 0.04, 40), (0.104, 0.0, 60)):
        q = robot.inverse_kinematics(link=robot.get_link("hand"), pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        tg = torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous()
        for _ in range(n):
            task.step_raw(tg)
    torch.cuda.synchronize()
    sc = task._mir; sc.set_diag(True)
    setup_grasp.keep = env
    return sc, tg[None].repeat(64, 1, 1), None


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    sc, acts, bufs = setup_grasp(B) if len(sys.argv) > 2 and sys.argv[2] == "grasp" else setup(B)
    sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    sc.lib.mir_debug_profile_step.restype = C.c_int
    acc = np.zeros(10)
    sub = np.zeros(7)
    dyn = np.zeros(6)
    nw = np.zeros(8)
    ex = np.zeros(6)
    n = 20
    wall = []
    for k in range(n):
        sc.set_pd_targets(acts[k % 64])
        prof = torch.zeros(64, dtype=torch.int64, device=sc.device)
        prof[29] = 2**62
        sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
        p = prof.cpu().numpy().astype(np.float64)
        if p[48] > 0:  # two-wave single-step instantiation (-DMIR_PROFILE_SINGLE): stamp 5 (end of the collision phase) belongs to
            p[5] = p[4]   # the other wave; tools/probes/dual_timeline.py prints both waves' barrier times instead
        acc += np.diff(p[:11])
        wall.append(((p[27]-p[26])/100.0, (p[28]-p[26])/100.0, (p[29]-p[26])/100.0, (p[25]-p[24])))
        nw += np.array([p[16]-p[7], p[17]-p[16], p[14]-p[17], p[18]-p[14], p[15]-p[18], p[19]-p[15], p[20]-p[19], p[21]-p[20]])
        ex += np.array([p[22]-p[13], p[23]-p[22], p[5]-p[23], p[0]-p[24], p[25]-p[10], p[25]-p[24]])
        dyn += np.array([p[32]-p[1], p[33]-p[32], p[2]-p[33], p[34]-p[2], p[35]-p[34], p[3]-p[35]])
        sub += np.array([p[11]-p[4], p[12]-p[11], p[13]-p[12], p[5]-p[13], p[14]-p[7], p[15]-p[14], p[8]-p[15]])
    acc /= n
    nw /= n
    ex /= n
    sub /= n
    dyn /= n
    print(f"B={B}: phase cycles (block 0, shader clock; avg of {n} steps)")
    for name, c in zip(PHASES, acc):
        print(f"  {name:14s} {c:9.0f} cyc  {100 * c / acc.sum():5.1f}%")
    print(f"  total          {acc.sum():9.0f} cyc")
    w = np.array(wall)
    print("  wall clock (us from block-0 entry): block-0 exit %.1f | first block exit %.1f | LAST block exit %.1f (per-step max %.1f) ; block-0 cycles/us = %.0f MHz" % (w[:,0].mean(), w[:,2].mean(), w[:,1].mean(), w[:,1].max(), (w[:,3]/w[:,0]).mean()))
    print("  dynamics split: cdof+cinert %.0f | velocity scan %.0f | cddq+cvel+crb suffix %.0f | acc scan %.0f | RNE+force suffix %.0f | M fill+bias %.0f" % tuple(dyn))
    if sub[0] >= 0:
        print("  collide split: geoms %.0f | broadphase %.0f | narrowphase %.0f | compaction+contacts %.0f" % (sub[0], sub[1], sub[2], sub[3]))
    print("  newton it0 split: forces+H build %.0f | gsum+GJ %.0f | rest of iteration(s) %.0f" % (sub[4], sub[5], sub[6]))
    print("  compaction split: scan %.0f | per-candidate consts %.0f | contact writes %.0f ; prologue (entry->stamp0) %.0f | epilogue (stamp10->exit) %.0f | whole kernel %.0f" % (ex[0], ex[1], ex[2], ex[3], ex[4], ex[5]))
    print("  newton it0 fine: forces/cfb %.0f | gradient loop %.0f | gsum+check %.0f | H build %.0f | GJ %.0f | mv+jv %.0f | line search %.0f | improvement+update %.0f" % tuple(nw))
    nc, ne, ni = (x.cpu().numpy() for x in sc.get_diag())
    print(f"  ncon mean {nc.mean():.2f} nefc mean {ne.mean():.2f} niter mean {ni.mean():.2f} max {ni.max()}")
    for Bx in (() if len(sys.argv) > 2 else (256, 4096, 65536)):
        s2, a2, b2 = setup(Bx, warm=20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 300
        for t in range(K):
            s2.step_fused(a2[t % 64], *b2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"  B={Bx:6d}: {dt * 1e6:8.1f} us/step  {Bx / dt / 1e6:8.2f} M env-steps/s")


if __name__ == "__main__":
    main()
