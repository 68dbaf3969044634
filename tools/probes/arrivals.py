"""Developer probe: on the reference's expert (MODE=expert) or the scripted grasp, per step: envs above 16 candidate points, how many of
them were at most 16 in the step before (ARRIVALS on the deferred list) and how many of last step's fell back (LEAVERS)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
import numpy as np, torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
mir = env._env._mir
mir.set_diag(True)
rows = []
prev = np.zeros(B, bool)
def note():
    global prev
    pts = mir.get_diag(points=True)[3].cpu().numpy()
    over = pts > 16
    rows.append((int(over.sum()), int((over & ~prev).sum()), int((~over & prev).sum())))
    prev = over
if os.environ.get("MODE") == "expert":
    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
    obs, _ = env.reset(seed=1)
    for stage in ex.STAGES:
        for _ in range(40):
            obs, *_ = env.step(ex.expert_policy(env.get_robot(), obs, stage)); note()
else:
    obs, _ = env.reset(seed=0)
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    tg, q_prev = [], None
    for dz, grip in [(0.25, 0.04), (0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.40, 0.0)]:
        q = robot.inverse_kinematics(link=robot.get_link("hand"), pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        tg.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    env.reset(seed=0)
    for t in tg:
        for _ in range(40):
            env.step(t); note()
r = np.array(rows)
ov = r[:, 0] > 0
print(f"steps with envs above 16 points: {int(ov.sum())} of {len(r)}; of those, steps WITHOUT arrivals: {int((ov & (r[:, 1] == 0)).sum())}; env-steps above 16: {int(r[:, 0].sum())}, arrivals {int(r[:, 1].sum())}, leavers {int(r[:, 2].sum())}")
print("per step (above16, arrivals, leavers):", [tuple(x) for x in r[ov]][:140])
