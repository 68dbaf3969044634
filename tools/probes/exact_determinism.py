import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gym-genesis_amd"))
import numpy as np, torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B = 4096
def run(sync):
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
    obs, _ = env.reset(seed=0)
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    tg, q_prev = [], None
    for dz, grip in [(0.25, 0.04), (0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.40, 0.0)]:
        q = robot.inverse_kinematics(link=robot.get_link("hand"), pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        tg.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    env.reset(seed=0)
    mir = env._env._mir
    mir.exact_stats(reset=True)
    states = []
    for t in tg:
        for _ in range(40):
            o, r, term, *_ = env.step(t)
            if sync: torch.cuda.synchronize()
            states.append(o["environment_state"].clone())
    torch.cuda.synchronize()
    return torch.stack(states), mir.exact_stats(), mir.exact_route(), float(term.mean())
ref = None
for mode, sync, extra in (("0", False, {}), ("2", False, {}), ("2", False, {"MIR_EXACT_BIG_SIDE": "0"}), ("2", False, {"MIR_EXACT_HEAVY_SORT": "0"}),
                          ("2", False, {"MIR_EXACT_BIG_SIDE": "0", "MIR_EXACT_HEAVY_SORT": "0"})):
    for k in ("MIR_EXACT_BIG_SIDE", "MIR_EXACT_HEAVY_SORT"):
        os.environ.pop(k, None)
    os.environ.update(extra)
    os.environ["MIR_EXACT_BIG"] = mode
    st, stats, route, lifted = run(sync)
    if ref is None: ref = st
    d = (st != ref).flatten(1).any(1)
    first = int(torch.nonzero(d)[0]) if d.any() else -1
    nenv = int((st[first] != ref[first]).any(1).sum()) if first >= 0 else 0
    print(f"BIG={mode} sync={sync} {extra}: lifted {lifted:.3f} {stats} {route}; first differing step {first} ({nenv} envs)")
