"""Thin read/write views over a MirScene that mirror the slice of the Genesis object API the
reference tasks and examples call on the hot path: ``scene.step()``,
``entity.get_pos/get_quat/get_dofs_position/get_qpos/get_link``, ``link.get_pos/get_quat``,
``entity.control_dofs_position`` (/root/reference/gym_genesis/tasks/franka/cube_pick.py:96-146,
/root/reference/gym_genesis/env.py:59-60).  All tensors are float32 on the scene's device,
batch first, quaternions wxyz.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch


class LinkView:
    def __init__(self, mir, body_index: int, name: str):
        self._mir, self.idx, self.name = mir, body_index, name

    def get_pos(self) -> torch.Tensor:
        return self._mir.get_links()[0][:, self.idx, :].contiguous()

    def get_quat(self) -> torch.Tensor:
        return self._mir.get_links()[1][:, self.idx, :].contiguous()


class EntityView:
    """One articulated entity (robot) or free body (cube): the bodies of one kinematic tree."""

    def __init__(self, mir, builder, root: str, dof_names: Sequence[str]):
        self._mir, self._b = mir, builder
        self.root = builder.body_index(root)
        self.dof_names = tuple(dof_names)
        self.dof_idx = [builder.dof_index(n) for n in dof_names]
        # qpos column of each scalar dof (scalar joints come first in every scene built here)
        self._qcols = list(self.dof_idx)
        ctrl = [i for i, d in enumerate(builder.dofs) if d["ctrl_mode"] == 1]
        self._ucols = [ctrl.index(i) for i in self.dof_idx if i in ctrl]

    @property
    def n_dofs(self) -> int:
        return len(self.dof_idx)

    def get_link(self, name: str) -> LinkView:
        return LinkView(self._mir, self._b.body_index(name), name)

    def get_pos(self) -> torch.Tensor:
        return self._mir.get_links()[0][:, self.root, :].contiguous()

    def get_quat(self) -> torch.Tensor:
        return self._mir.get_links()[1][:, self.root, :].contiguous()

    def get_dofs_position(self) -> torch.Tensor:
        return self._mir.get_state()[0][:, self._qcols].contiguous()

    def get_dofs_velocity(self) -> torch.Tensor:
        return self._mir.get_state()[1][:, self.dof_idx].contiguous()

    def get_qpos(self) -> torch.Tensor:
        return self.get_dofs_position()

    def control_dofs_position(self, position, dofs_idx_local=None) -> None:
        """PD targets for a subset of this entity's dofs (others keep their current target)."""
        tgt = self._mir.get_state()[2]
        pos = torch.as_tensor(position, dtype=torch.float32, device=tgt.device)
        idx = list(range(self.n_dofs)) if dofs_idx_local is None else [int(i) for i in np.asarray(dofs_idx_local).ravel()]
        cols = [self._ucols[i] for i in idx]
        tgt[:, cols] = pos.reshape(tgt.shape[0], len(cols))
        self._mir.set_pd_targets(tgt)


class SceneView:
    def __init__(self, mir, env_spacing=(1.0, 1.0), global_num_envs=None, offset=0):
        self._mir = mir
        n = mir.num_envs if global_num_envs is None else int(global_num_envs)
        per_row = max(1, int(math.ceil(math.sqrt(n))))
        idx = np.arange(n)
        rows, cols = idx // per_row, idx % per_row
        off = np.stack([(cols - (per_row - 1) / 2.0) * env_spacing[0], (rows - (math.ceil(n / per_row) - 1) / 2.0) * env_spacing[1],
                        np.zeros(n)], axis=1)
        # visual offsets only (the envs never interact); this shard's slice of the global grid
        self.envs_offset = off[offset:offset + mir.num_envs]
        self.n_envs = mir.num_envs

    def step(self) -> None:
        self._mir.step(1)
