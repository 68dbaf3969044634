"""Does the 16-point contact capacity of the pick kernel change outcomes of the reference's expert, or is it chaos?  

The verbatim expert (examples/franka/pick_cube_state.py) on the float64 ORACLE at capacity 16 (thinned manifolds) and 48 (never
reached), same spawns, same policy code.  Per env: the first step at which thinning fires (candidate points > 16), the first step
at which the two runs' cube positions differ by more than 1e-6 m, and the verdicts.  CPU only (test infrastructure: the oracle).
Usage: python3 tools/cap_study.py [num_envs]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import fake_scene  # noqa: E402
from gym_genesis.backend import models  # noqa: E402
from gym_genesis.tasks.franka import cube_pick  # noqa: E402


def example():
    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(B, max_contacts, ex):
    cube_pick.MirScene = fake_scene.OracleScene
    real = models.franka_cube_pick_scene

    def scene(**kw):
        sb = real(**kw)
        sb.opt["max_contacts"] = max_contacts
        return sb

    models.franka_cube_pick_scene = scene
    try:
        from gym_genesis.env import GenesisEnv
        env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
        obs, _ = env.reset(seed=0)
        o = env._env._mir.o
        cube, ncand, rew = [], [], []
        for stage in ex.STAGES:
            for _ in range(40):
                a = ex.expert_policy(env.get_robot(), obs, stage)
                obs, reward, done, _, info = env.step(a)
                cube.append(obs["environment_state"][:, :3].numpy().copy())
                ncand.append(o.ncand_all().copy())
                rew.append(np.asarray(reward).copy())
    finally:
        models.franka_cube_pick_scene = real
    return np.stack(cube), np.stack(ncand), np.stack(rew)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ex = example()
    c16, n16, r16 = run(B, 16, ex)
    c48, n48, r48 = run(B, 48, ex)
    ok16, ok48 = (r16 > 0).any(0), (r48 > 0).any(0)
    T = c16.shape[0]
    hit = n16 > 16
    first_hit = np.where(hit.any(0), hit.argmax(0), T)
    diff = np.abs(c16 - c48).max(2) > 1e-6
    first_div = np.where(diff.any(0), diff.argmax(0), T)
    print(f"{B} envs x {T} steps, reference expert on the oracle")
    print(f"lifted: capacity 16 {ok16.mean():.3f}, capacity 48 {ok48.mean():.3f}; same verdict in {np.mean(ok16 == ok48):.3f} of the envs")
    print(f"cap_hit_frac (env-steps with more than 16 candidate points): {hit.mean():.4f}; envs that ever hit the cap: {hit.any(0).mean():.3f}; "
          f"max candidates {n16.max()} (capacity-48 run: {n48.max()})")
    never = ~hit.any(0)
    print(f"envs that never hit the cap: {never.sum()}, of which diverged from the capacity-48 run: {int((first_div[never] < T).sum())}")
    both = hit.any(0)
    print(f"envs that hit the cap: {both.sum()}; divergence starts AT the first cap-hit step (+-1) in {int((np.abs(first_div[both] - first_hit[both]) <= 1).sum())}, "
          f"later in {int((first_div[both] > first_hit[both] + 1).sum())}, never in {int((first_div[both] >= T).sum())}")
    flip = ok16 != ok48
    for e in np.where(flip)[0]:
        print(f"  env {e}: verdict 16 -> {ok16[e]}, 48 -> {ok48[e]}; first cap hit at step {first_hit[e]}, first divergence at step {first_div[e]}, "
              f"cap-hit steps {int(hit[:, e].sum())}, spawn r = {np.hypot(*c16[0, e, :2]):.3f}")
    stage_hit = hit.reshape(5, 40, B).mean((1, 2))
    print("cap_hit_frac by stage (hover, stabilize, grasp, grasp, lift):", np.round(stage_hit, 4).tolist())


if __name__ == "__main__":
    main()
