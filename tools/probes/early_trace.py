"""Developer probe (run under rocprofv3 --kernel-trace): one warm-up + one measured episode of the scripted grasp through GenesisEnv.step
with every contact kept, so that the kernel trace shows how the list launches lie beside the main launches (tools/probes/early_trace.sh
prints the overlap).  MODE=expert: the reference's expert instead (IK between the steps)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
import torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
if os.environ.get("MODE") == "expert":
    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
    for ep in range(2):
        obs, _ = env.reset(seed=ep)
        for stage in ex.STAGES:
            for _ in range(40):
                obs, *_ = env.step(ex.expert_policy(env.get_robot(), obs, stage))
else:
    obs, _ = env.reset(seed=0)
    robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
    tg, q_prev = [], None
    for dz, grip in [(0.25, 0.04), (0.25, 0.04), (0.104, 0.04), (0.104, 0.0), (0.40, 0.0)]:
        q = robot.inverse_kinematics(link=robot.get_link("hand"), pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
        q_prev = q
        tg.append(torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous())
    for ep in range(2):
        env.reset(seed=0)
        for t in tg:
            for _ in range(40):
                env.step(t)
torch.cuda.synchronize()
print(env._env._mir.exact_stats(), env._env._mir.exact_route())
