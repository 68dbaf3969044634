"""gym_genesis — drop-in task registry backed by the MI355X rigid-body backend.

Same ids, entry point and default kwargs as the reference registry
(/root/reference/gym_genesis/__init__.py:3-37).
"""
from ._gym import make, register  # noqa: F401

_DEFAULTS = dict(
    robot="so101",
    enable_pixels=False,
    num_envs=10,
    observation_height=480,
    observation_width=640,
    env_spacing=(1.0, 1.0),
    camera_capture_mode="global",
    strip_environment_state=True,
)

for _id, _task in (("gym_genesis/CubePick-v0", "cube_pick"), ("gym_genesis/CubeStack-v0", "cube_stack")):
    register(
        id=_id,
        entry_point="gym_genesis.env:GenesisEnv",
        max_episode_steps=200,
        nondeterministic=False,
        kwargs=dict(task=_task, **_DEFAULTS),
    )
