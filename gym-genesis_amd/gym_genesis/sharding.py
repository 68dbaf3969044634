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


def gather_rows(rows: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather equally sized row blocks into the global (B, D) tensor (rank order = env order)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    out = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, rows, group=group)
    return out
