"""Developer tool: the scripted grasp (5 x 40 steps, IK-precomputed joint targets, 4096 envs) through GenesisEnv.step with the manifolds
thinned at 16 points and with exact contacts (deferred envs on the wave kernel): us per step over five episodes each, the per-step
times of the overflow steps, the counters of mir_get_exact_stats.  Also the command profiled by tools/profile_exact.sh."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym-genesis_amd"))
import numpy as np, torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B = 4096
def grasp_targets(env):
    obs, _ = env.reset(seed=0)
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    eef = robot.get_link("hand")
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    stages = [(0.25, 0.04), (0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.40, 0.0)]
    targets, q_prev = [], None
    for dz, grip in stages:
        q = robot.inverse_kinematics(link=eef, pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        targets.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    return targets
for exact in (False, True, False, True):
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=exact)
    tg = grasp_targets(env)
    mir = env._env._mir
    ts = []
    for rep in range(5):
        env.reset(seed=0)
        mir.exact_stats(reset=True)
        torch.cuda.synchronize()
        per = []
        t0 = time.perf_counter()
        for t in tg:
            for _ in range(40):
                ta = time.perf_counter()
                o, r, term, tr, info = env.step(t)
                per.append(time.perf_counter() - ta)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
    st = mir.exact_stats()
    per = np.array(per) * 1e6
    print(f"exact={exact}: us/step over 5 episodes {np.round(ts, 2)}  -> {B / np.median(ts):.1f} M env-steps/s; lifted {float(term.mean()):.3f}; stats {st}")
    if exact:
        pts = per.copy()
        print("  per-step us on overflow steps (sorted top):", np.round(np.sort(pts)[-45:], 1))
        print("  median per-step us:", np.median(pts))
    del env
