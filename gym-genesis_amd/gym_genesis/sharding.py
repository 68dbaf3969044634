"""Env-axis sharding over the GPUs of one node.

The physics never exchanges data between envs, so a sharded run is N independent processes
(one per MI355X) that each own a contiguous block of the global batch.  The only collective
is the optional observation gather: one all-gather of [agent_pos | environment_state | reward]
rows (+ the terminated mask) over RCCL (``torch.distributed`` backend "nccl" on ROCm; "gloo"
in the CPU tests).  xGMI is point-to-point and the payload is ~85 B/env, so the gather is
latency-bound; everything is packed into ONE float32 buffer so it costs one collective.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global env axis owned by `rank`."""
    return num_envs * rank // world, num_envs * (rank + 1) // world


def pack_rows(obs: Dict[str, torch.Tensor], reward: torch.Tensor, terminated: torch.Tensor) -> torch.Tensor:
    """(B_local, 9 + 11 + 1 + 1) float32: agent_pos | environment_state | reward | terminated."""
    return torch.cat([obs["agent_pos"], obs["environment_state"], reward.reshape(-1, 1).float(),
                      terminated.reshape(-1, 1).float()], dim=1).contiguous()


def unpack_rows(rows: torch.Tensor, agent_dim: int = 9, env_dim: int = 11):
    obs = {"agent_pos": rows[:, :agent_dim], "environment_state": rows[:, agent_dim:agent_dim + env_dim]}
    reward = rows[:, agent_dim + env_dim]
    terminated = rows[:, agent_dim + env_dim + 1] != 0
    return obs, reward, terminated


_COUNTS: dict = {}  # (group, world, local rows) -> rows per rank, for gather_rows(num_envs=None)


def gather_rows(rows: torch.Tensor, group=None, num_envs: int | None = None) -> torch.Tensor:
    """All-gather the row blocks of all ranks into the global (B, D) tensor (rank order = env order).

    shard_bounds gives ranks blocks that differ by one row when num_envs % world != 0, and all_gather_into_tensor needs
    equal blocks: every rank pads its block to the largest one (ceil(num_envs / world)) and the padding is dropped after
    the collective.  `num_envs` = the global batch (what the callers on the data path pass: the block sizes then follow from
    shard_bounds, no communication).  Without it the ranks exchange their block sizes ONCE per (group, local size) -- one small
    collective and one host read -- and the answer is cached (so the block sizes of a group must not change between calls;
    pass `num_envs` where they can)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    n = rows.shape[0]
    if num_envs is None:
        ck = (id(group) if group is not None else None, world, n)
        counts = _COUNTS.get(ck)
        if counts is None:
            sizes = torch.tensor([n], dtype=torch.int64, device=rows.device)
            all_sizes = torch.empty((world,), dtype=torch.int64, device=rows.device)
            dist.all_gather_into_tensor(all_sizes, sizes, group=group)
            counts = _COUNTS[ck] = [int(c) for c in all_sizes.tolist()]
    else:
        counts = [shard_bounds(num_envs, r, world)[1] - shard_bounds(num_envs, r, world)[0] for r in range(world)]
        if counts[dist.get_rank(group)] != n:
            raise ValueError(f"gather_rows: this rank holds {n} rows, shard_bounds({num_envs}) gives it {counts[dist.get_rank(group)]}")
    width = max(counts)
    if min(counts) == width:  # equal shards: one collective, no padding
        out = torch.empty((world * n, *rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
        return out
    send = rows.new_zeros((width, *rows.shape[1:]))
    send[:n] = rows
    out = torch.empty((world * width, *rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, send, group=group)
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)], dim=0)
