"""Iterations of the batched IK (mir_ik_kernel) per env on the reference's expert episode (examples/franka/pick_cube_state.py: one
robot.inverse_kinematics call per env.step, 5 stages x 40 steps, 4096 envs): histogram by stage, converged fraction, and the kernel's
time per call by HIP events.  Usage: python3 tools/probes/ik_iters.py"""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv

spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
mir = env._env._mir
lib = mir.lib
lib.mir_debug_ik_iters.argtypes = [C.c_void_p, C.c_void_p]
iters = torch.zeros(B, dtype=torch.int32, device=mir.device)
assert lib.mir_debug_ik_iters(mir.h, C.c_void_p(iters.data_ptr())) == 0
obs, _ = env.reset(seed=0)
robot = env.get_robot()
eef = robot.get_link("hand")
quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=mir.device).expand(B, -1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for si, stage in enumerate(ex.STAGES):
    its, errs, us = [], [], []
    for t in range(40):
        cube = obs["environment_state"][:, :3]
        dz = 0.115 if stage in ("hover", "stabilize") else (0.03 if stage == "grasp" else 0.25)
        tgt = cube + torch.tensor([0.0, 0.0, dz], device=mir.device)
        torch.cuda.synchronize()
        e0.record()
        q, err = robot.inverse_kinematics(link=eef, pos=tgt, quat=quat, return_error=True)
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
        its.append(iters.cpu().numpy().copy()); errs.append(err.cpu().numpy().copy())
        a = ex.expert_policy(robot, obs, stage)
        obs, r, term, _, _ = env.step(a)
    its, errs = np.stack(its), np.stack(errs)
    conv = (errs[:, :, 0] < 5e-4) & (errs[:, :, 1] < 5e-3)
    wave_max = its.reshape(40, B // 4, 4).max(2)
    print(f"stage {si} {stage}: IK call {np.median(us):.1f} us (median of 40, events around the wrapper); iterations per env mean {its.mean():.2f}, quantiles 0.5/0.9/0.99/max "
          f"{np.quantile(its, 0.5):.0f}/{np.quantile(its, 0.9):.0f}/{np.quantile(its, 0.99):.0f}/{its.max()}; per WAVE (max of 4 envs) mean {wave_max.mean():.2f}, max {wave_max.max()}; "
          f"first step of the stage: mean {its[0].mean():.2f} max {its[0].max()}; converged {conv.mean():.4f}; histogram {np.bincount(its.ravel(), minlength=33)[:33].tolist()}")
