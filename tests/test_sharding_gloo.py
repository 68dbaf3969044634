"""Multi-process (gloo, world_size 2, CPU) test of the env-axis sharded path: two ranks each own
half of the global batch, draw the SAME global reset stream, step their shard, all-gather the
packed observation rows; the result must equal the unsharded run bit for bit."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_GLOBAL, STEPS = 6, 5


def _paths():
    for p in (os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run(shard, actions, b_global=B_GLOBAL, pass_num_envs=True):
    """Rollout of this shard on the oracle-backed test double; returns packed rows per step."""
    _paths()
    import fake_scene
    from gym_genesis.tasks.franka import cube_pick

    cube_pick.MirScene = fake_scene.OracleScene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.sharding import gather_rows, pack_rows, shard_bounds

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=b_global, enable_pixels=False, shard=shard)
    rank, world = shard if shard else (0, 1)
    lo, hi = shard_bounds(b_global, rank, world)
    assert env.num_envs == hi - lo
    env.reset(seed=3)
    out = []
    for t in range(STEPS):
        obs, reward, terminated, truncated, info = env.step(actions[t, lo:hi])
        rows = pack_rows(obs, reward, torch.as_tensor(terminated))
        out.append(gather_rows(rows, num_envs=b_global if pass_num_envs else None))
    return torch.stack(out)


def _worker(rank, world, port, actions, q, b_global=B_GLOBAL, pass_num_envs=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = _run((rank, world), actions, b_global, pass_num_envs)
        q.put((rank, rows.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("b_global,pass_num_envs", [(6, True), (7, True), (7, False)])
def test_two_rank_shard_equals_single_process(b_global, pass_num_envs):
    """Equal (3 + 3) and unequal (3 + 4) shards: the gathered rows equal the unsharded run bit for bit."""
    _paths()
    actions = np.random.default_rng(0).uniform(-1, 1, (STEPS, b_global, 9)).astype(np.float32)
    ref = _run(None, actions, b_global).numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, actions, q, b_global, pass_num_envs)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ref.shape == (STEPS, b_global, 22)
    for r in range(2):
        assert np.array_equal(got[r], ref), f"rank {r}: gathered rows differ from the unsharded run"


def test_eight_rank_unequal_shards_equal_single_process():
    """BASELINE configs[2]'s world size on CPU: eight gloo ranks, 37 envs (shards of 4 and 5), the global reset stream sliced per rank;
    the rows every rank gathers equal the unsharded run bit for bit."""
    _paths()
    from gym_genesis.sharding import shard_bounds

    world, b_global = 8, 37
    sizes = [shard_bounds(b_global, r, world)[1] - shard_bounds(b_global, r, world)[0] for r in range(world)]
    assert sorted(set(sizes)) == [4, 5] and sum(sizes) == b_global
    actions = np.random.default_rng(1).uniform(-1, 1, (STEPS, b_global, 9)).astype(np.float32)
    ref = _run(None, actions, b_global).numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, actions, q, b_global, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ref.shape == (STEPS, b_global, 22)
    for r in range(world):
        assert np.array_equal(got[r], ref), f"rank {r}: gathered rows differ from the unsharded run"


def test_pack_unpack_roundtrip():
    _paths()
    from gym_genesis.sharding import pack_rows, shard_bounds, unpack_rows

    obs = {"agent_pos": torch.arange(18.0).reshape(2, 9), "environment_state": torch.arange(22.0).reshape(2, 11)}
    rows = pack_rows(obs, torch.tensor([0.0, 1.0]), torch.tensor([False, True]))
    o2, r2, t2 = unpack_rows(rows)
    assert torch.equal(o2["agent_pos"], obs["agent_pos"]) and torch.equal(o2["environment_state"], obs["environment_state"])
    assert r2.tolist() == [0.0, 1.0] and t2.tolist() == [False, True]
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
