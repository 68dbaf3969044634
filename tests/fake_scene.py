"""Test double for gym_genesis.backend.lib.MirScene that drives the CPU ORACLE.

Lives under tests/ and is only ever monkeypatched in by tests: it lets the host-side logic
(GenesisEnv, task classes, registry, sharding) run in the CPU-only test tier.  The product has
no such path: without a GPU, the real MirScene raises.
"""
from __future__ import annotations

import numpy as np
import torch

import orc
from gym_genesis.backend.lib import StepHelpers


class OracleScene(StepHelpers):
    def __init__(self, spec, num_envs, device=None):
        self.spec = spec
        self.o = orc.Oracle(spec, int(num_envs))
        self.device = torch.device("cpu")
        self.num_envs = int(num_envs)
        self.nq, self.nv, self.nbody = self.o.nq, self.o.nv, spec.nbody
        self.nu = self.o.nu
        self.agent_dim, self.env_dim = self.o.agent_dim, self.o.env_dim
        self.nfree, self.kernel = self.o.nfree, 0
        self.n_arm = sum(1 for b in range(1, spec.nbody) if spec.body[b].jtype in (1, 2))

    def set_diag(self, on):
        pass

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty((self.num_envs, *shape), dtype=dtype)

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def staged(self, rows):
        return torch.from_numpy(np.ascontiguousarray(rows))

    def reset(self, obj_pos, obj_quat, arm_qpos, env_mask=None):
        assert env_mask is None
        self.o.reset(self._np(obj_pos), self._np(obj_quat), self._np(arm_qpos))

    def set_pd_targets(self, tgt):
        self.o.set_targets(self._np(tgt))

    def step(self, n_steps=1):
        for _ in range(n_steps):
            self.o.step_batch(None)

    def _fill(self, agent_pos, env_state, reward, terminated):
        a, e, r, t = self.o.get_obs()
        agent_pos.copy_(torch.from_numpy(a.astype(np.float32)))
        env_state.copy_(torch.from_numpy(e.astype(np.float32)))
        reward.copy_(torch.from_numpy(r.astype(np.float32)))
        terminated.copy_(torch.from_numpy(t))

    def step_fused(self, action, agent_pos, env_state, reward, terminated):
        self.o.step_batch(None if action is None else self._np(action).astype(np.float32))
        self._fill(agent_pos, env_state, reward, terminated)

    # pointer-level launches of StepHelpers.step_fresh: the double looks its tensors up by address
    def _alloc_outputs(self, agent_dim, env_dim):
        outs, ptrs = super()._alloc_outputs(agent_dim, env_dim)
        self.__dict__.setdefault("_by_ptr", {})[ptrs] = outs
        return outs, ptrs

    def as_action(self, action, dim):
        a = super().as_action(action, dim)
        self._last_action = a
        return a

    def step_prepare_ptrs(self, ptrs):
        self._prepared = ptrs

    def step_go_ptr(self, action_ptr):
        import ctypes

        # (the product passes addresses only; on the CPU double the action is read back from its address)
        a = np.ctypeslib.as_array((ctypes.c_float * (self.num_envs * self.nu)).from_address(action_ptr)).reshape(self.num_envs, self.nu).copy()
        ptrs, self._prepared = self._prepared, None
        self.step_begin(torch.from_numpy(a), *self._by_ptr.pop(ptrs))

    def step_fused_ptrs(self, action_ptr, ptrs):
        assert action_ptr == self._last_action.data_ptr()
        self.step_fused(self._last_action, *self._by_ptr.pop(ptrs))

    def step_begin(self, action, agent_pos, env_state, reward, terminated):
        assert getattr(self, "_pending", None) is None, "step_begin twice without step_end"
        self.step_fused(action, agent_pos, env_state, reward, terminated)
        self._pending = terminated.numpy().astype(np.bool_)

    def step_end(self):
        out, self._pending = self._pending.copy(), None
        self._host_pending = None
        return out

    def step_end_ptr(self, host_ptr):
        import ctypes

        out = self.step_end()
        ctypes.memmove(host_ptr, out.ctypes.data, out.nbytes)

    def step_packed(self, action, rows):
        bufs = (self.empty(self.agent_dim), self.empty(self.env_dim), self.empty(), self.empty(dtype=torch.uint8))
        self.step_fused(action, *bufs)
        rows[:, :self.agent_dim] = bufs[0]
        rows[:, self.agent_dim:self.agent_dim + self.env_dim] = bufs[1]
        rows[:, self.agent_dim + self.env_dim] = bufs[2]
        rows[:, self.agent_dim + self.env_dim + 1] = bufs[3].float()

    def get_obs(self):
        bufs = (self.empty(self.agent_dim), self.empty(self.env_dim), self.empty(), self.empty(dtype=torch.uint8))
        self._fill(*bufs)
        return bufs

    def get_state(self):
        q, v = self.o.state()
        tgt = np.stack([self.o.read(orc.F_TARGET, e)[self.o.u_dofs] for e in range(self.num_envs)])
        ws = np.stack([self.o.read(orc.F_QACC_WS, e) for e in range(self.num_envs)])
        return tuple(torch.from_numpy(x.astype(np.float32)) for x in (q, v, tgt, ws))

    def get_links(self):
        pos = np.stack([self.o.read(orc.F_XPOS, e).reshape(-1, 3) for e in range(self.num_envs)])
        quat = np.stack([self.o.read(orc.F_XQUAT, e).reshape(-1, 4) for e in range(self.num_envs)])
        return torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(quat.astype(np.float32))

    def render(self, cam, vis, mode=0, env_offset=None, out=None):
        xpos, xquat = (t.numpy().astype(np.float64) for t in self.get_links())
        if mode == 1:
            off = None if env_offset is None else self._np(env_offset).astype(np.float64)
            return torch.from_numpy(orc.render_image(self.spec, cam, vis, xpos, xquat, offsets=off))
        return torch.from_numpy(np.stack([orc.render_image(self.spec, cam, vis, xpos[e:e + 1], xquat[e:e + 1])
                                          for e in range(self.num_envs)]))

    def render_cams(self, cam, vis, cam_pos, cam_lookat, cam_up=None, out=None):
        from gym_genesis.backend.spec import make_camera

        xpos, xquat = (t.numpy().astype(np.float64) for t in self.get_links())
        P, L = self._np(cam_pos), self._np(cam_lookat)
        U = None if cam_up is None else self._np(cam_up)
        imgs = []
        for e in range(self.num_envs):
            c = make_camera(cam.width, cam.height, P[e], L[e], cam.fov_deg, up=tuple(cam.up) if U is None else U[e])
            imgs.append(orc.render_image(self.spec, c, vis, xpos[e:e + 1], xquat[e:e + 1]))
        return torch.from_numpy(np.stack(imgs))

    def inverse_kinematics(self, link_body, pos, quat=None, init_qpos=None, return_error=False, **opts):
        init = self.get_state()[0][:, :self.n_arm].numpy() if init_qpos is None else self._np(init_qpos)
        kw = dict(max_iters=opts.get("max_iters", 20), damping=opts.get("damping", 0.05), pos_tol=opts.get("pos_tol", 5e-4),
                  rot_tol=opts.get("rot_tol", 5e-3), max_step=opts.get("max_step", 0.5), respect_limits=bool(opts.get("respect_joint_limit", 1)))
        q, err = self.o.ik(int(link_body), self._np(pos), None if quat is None else self._np(quat), init, **kw)
        q, err = torch.from_numpy(q.astype(np.float32)), torch.from_numpy(err.astype(np.float32))
        return (q, err) if return_error else q

    def inverse_kinematics_rows(self, link_body, pos, quat, init_qpos, env_idx, flags, init_col0=0, init_ncols=0, return_error=False, **opts):
        """include/mirigid.h: mir_inverse_kinematics_rows, restated with NumPy indexing around the oracle's solver: row k belongs to env
        env_idx[k]; pos / quat / init_qpos are addressed by row unless their *_BY_ENV flag is set (1 / 2 / 8), quat may be ONE quaternion
        (4); init_qpos holds init_ncols columns from init_col0 on, the other joints are seeded from the scene state."""
        B = self.num_envs
        idx = np.arange(B) if env_idx is None else np.clip(self._np(env_idx).astype(np.int64).reshape(-1), 0, B - 1)
        n = idx.size
        P = self._np(pos).reshape(-1, 3)
        P = P[idx] if flags & 1 else P[:n]
        Q = None
        if quat is not None:
            Qa = self._np(quat).reshape(-1, 4)
            Q = np.tile(Qa[:1], (n, 1)) if flags & 4 else (Qa[idx] if flags & 2 else Qa[:n])
        seed = self.get_state()[0][:, :self.n_arm].numpy()[idx].copy()
        if init_qpos is not None:
            nc = init_ncols or self.n_arm
            I = self._np(init_qpos).reshape(-1, nc)
            seed[:, init_col0:init_col0 + nc] = I[idx] if flags & 8 else I[:n]
        kw = dict(max_iters=opts.get("max_iters", 20), damping=opts.get("damping", 0.05), pos_tol=opts.get("pos_tol", 5e-4),
                  rot_tol=opts.get("rot_tol", 5e-3), max_step=opts.get("max_step", 0.5), respect_limits=bool(opts.get("respect_joint_limit", 1)))
        q, err = self.o.ik(int(link_body), P, Q, seed, **kw)
        q, err = torch.from_numpy(q.astype(np.float32)), torch.from_numpy(err.astype(np.float32))
        return (q, err) if return_error else q

    def close(self):
        pass
