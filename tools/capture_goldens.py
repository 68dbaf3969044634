#!/usr/bin/env python3
"""Record golden trajectories from the REAL reference stack, for the day a machine with `genesis-world` installed exists.

Not runnable in the build environment (Genesis and gymnasium are absent): this script only uses the
reference's PUBLIC API -- `gym.make("gym_genesis/CubePick-v0", robot=..., num_envs=B)`, `env.reset(seed)`, `env.step(a)`,
`env.get_robot().get_dofs_position()/get_dofs_velocity()` -- with a fixed seed and a fixed action file, and writes
`tests/golden/genesis_<task>_<robot>.npz` holding (seed, actions, per-step agent_pos, environment_state, reward, terminated,
joint positions / velocities).  `tests/test_gpu_parity.py::test_against_captured_genesis_goldens` picks the files up when
they exist and compares the HIP path with them at the north star's bar (joint state L-inf < 1e-4, masks bit-exact);
without them the oracle stays "parity unpinned".  No reference source is needed on the GPU box: only the .npz travels.

    python tools/capture_goldens.py --task cube_pick --robot franka --num-envs 16 --steps 200 --backend cpu

`--backend self` records the same file from THIS repository's GenesisEnv (needs a GPU): it pins nothing -- a backend cannot vouch for
itself -- but it runs the whole pipeline (this recorder, the .npz schema, the replaying test) end to end, so that the day a Genesis
machine exists the capture is one command and not a debugging session (tests/test_gpu_parity.py::test_golden_pipeline_end_to_end).
Such a file never belongs under tests/golden/: give it an --out elsewhere.
"""
import argparse
import os

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="cube_pick", choices=["cube_pick", "cube_stack"])
    ap.add_argument("--robot", default="franka", choices=["franka", "so101"])
    ap.add_argument("--num-envs", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="cpu", choices=["cpu", "gpu", "self"])
    ap.add_argument("--out", default=None, help="output .npz (default: tests/golden/genesis_<task>_<robot>_<scenario>.npz; required with --backend self)")
    ap.add_argument("--scenario", default="home", choices=["home", "smooth", "random"],
                    help="home: hold the home pose; smooth: slow sinusoidal joint targets; random: U(-1,1) targets (chaotic)")
    args = ap.parse_args()

    import torch

    if args.backend == "self":
        import sys
        if not args.out:
            raise SystemExit("--backend self needs --out (its file pins nothing and must not land in tests/golden/)")
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-genesis_amd"))
        from gym_genesis.env import GenesisEnv  # (THIS repository's package)
        version = "self (this repository's MI355X backend: pins nothing)"
        env = GenesisEnv(task=args.task, robot=args.robot, num_envs=args.num_envs, enable_pixels=False)
    else:
        import genesis as gs  # noqa: F401  (the reference's engine; ImportError here means this is not the machine to run on)
        import gymnasium as gym
        import gym_genesis  # noqa: F401  (the REFERENCE package, registers the ids)

        if not gs._initialized:  # the tasks only init when nobody did (cube_pick.py:35-36): pick the backend here
            gs.init(backend=gs.cpu if args.backend == "cpu" else gs.gpu, precision="32")
        version = getattr(gs, "__version__", "unknown")
        env_id = "gym_genesis/CubePick-v0" if args.task == "cube_pick" else "gym_genesis/CubeStack-v0"
        env = gym.make(env_id, robot=args.robot, num_envs=args.num_envs, enable_pixels=False).unwrapped
    obs, _ = env.reset(seed=args.seed)
    robot = env.get_robot()
    B = args.num_envs
    n = env.action_space.shape[0]
    home = robot.get_dofs_position().detach().cpu().numpy()[:, :n].copy()
    rng = np.random.default_rng(args.seed + 1)
    phase = rng.uniform(0, 2 * np.pi, (B, n))
    rec = {k: [] for k in ("actions", "agent_pos", "environment_state", "reward", "terminated", "qpos", "qvel")}
    rec0 = {"agent_pos0": obs["agent_pos"].detach().cpu().numpy(), "environment_state0": obs["environment_state"].detach().cpu().numpy()}
    for t in range(args.steps):
        if args.scenario == "home":
            a = home
        elif args.scenario == "smooth":
            a = home + 0.25 * np.sin(2 * np.pi * t / 150.0 + phase)
        else:
            a = rng.uniform(-1, 1, (B, n))
        a = a.astype(np.float32)
        obs, reward, terminated, truncated, info = env.step(torch.as_tensor(a))
        rec["actions"].append(a)
        rec["agent_pos"].append(obs["agent_pos"].detach().cpu().numpy())
        rec["environment_state"].append(obs["environment_state"].detach().cpu().numpy())
        rec["reward"].append(np.asarray(reward.detach().cpu() if hasattr(reward, "detach") else reward, dtype=np.float32))
        rec["terminated"].append(np.asarray(terminated, dtype=bool))
        rec["qpos"].append(robot.get_dofs_position().detach().cpu().numpy())
        rec["qvel"].append(robot.get_dofs_velocity().detach().cpu().numpy())
    out = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                   f"genesis_{args.task}_{args.robot}_{args.scenario}.npz")
    np.savez_compressed(out, seed=args.seed, num_envs=B, scenario=args.scenario, genesis_version=version,
                        **rec0, **{k: np.stack(v) for k, v in rec.items()})
    print("wrote", out)


if __name__ == "__main__":
    main()
