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


def _rows(t: torch.Tensor, envs_idx) -> torch.Tensor:
    """The rows `envs_idx` of a batch-first tensor, as Genesis's getters return them (`envs_idx=None`: all envs).  The reference's
    SO-101 expert asks for `robot.get_qpos(envs_idx=np.arange(B))` and `get_link("gripper").get_pos(envs_idx=torch.arange(B))`
    (/root/reference/examples/so_101/collect_task_stack_cube_batch.py:73,86,90): NumPy, torch (any device) and lists are accepted."""
    if envs_idx is None:
        return t
    if not (isinstance(envs_idx, torch.Tensor) and envs_idx.is_cuda):
        host = np.asarray(envs_idx.cpu() if isinstance(envs_idx, torch.Tensor) else envs_idx).reshape(-1)
        if host.size == t.shape[0] and np.array_equal(host, np.arange(t.shape[0])):
            return t  # (the experts' arange(B): no gather, no upload)
        envs_idx = host
    return t.index_select(0, torch.as_tensor(envs_idx, device=t.device).long().reshape(-1))


# robot.inverse_kinematics as ONE launch (mir_inverse_kinematics_rows); False: the full-batch launch with torch scatter / gather around it
IK_ROWS = True
# (an index outside the batch is clamped by the kernel; checking it here would cost a device -> host synchronisation per call)
CHECK_IK_IDX = False
from ..backend.spec import IK_INIT_BY_ENV, IK_POS_BY_ENV, IK_QUAT_BY_ENV, IK_QUAT_ONE  # noqa: E402


class LinkView:
    def __init__(self, mir, body_index: int, name: str):
        self._mir, self.idx, self.name = mir, body_index, name

    def get_pos(self, envs_idx=None) -> torch.Tensor:
        return _rows(self._mir.get_links()[0][:, self.idx, :].contiguous(), envs_idx)

    def get_quat(self, envs_idx=None) -> torch.Tensor:
        return _rows(self._mir.get_links()[1][:, self.idx, :].contiguous(), envs_idx)


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

    def get_pos(self, envs_idx=None) -> torch.Tensor:
        return _rows(self._mir.get_links()[0][:, self.root, :].contiguous(), envs_idx)

    def get_quat(self, envs_idx=None) -> torch.Tensor:
        return _rows(self._mir.get_links()[1][:, self.root, :].contiguous(), envs_idx)

    def get_dofs_position(self, dofs_idx_local=None, envs_idx=None) -> torch.Tensor:
        q = self._mir.get_state()[0][:, self._qcols].contiguous()
        if dofs_idx_local is not None:
            q = q[:, [int(i) for i in np.asarray(dofs_idx_local).ravel()]].contiguous()
        return _rows(q, envs_idx)

    def get_dofs_velocity(self, dofs_idx_local=None, envs_idx=None) -> torch.Tensor:
        v = self._mir.get_state()[1][:, self.dof_idx].contiguous()
        if dofs_idx_local is not None:
            v = v[:, [int(i) for i in np.asarray(dofs_idx_local).ravel()]].contiguous()
        return _rows(v, envs_idx)

    def get_qpos(self, envs_idx=None) -> torch.Tensor:
        """(`robot.get_qpos(envs_idx=np.arange(B))`: /root/reference/examples/so_101/collect_task_stack_cube_batch.py:73,90)"""
        return self.get_dofs_position(envs_idx=envs_idx)

    def _cols(self, t: torch.Tensor) -> torch.Tensor:
        """Columns `_qcols` of a (B, n) tensor: a slice when they are a run (every scene built here), else through an index tensor kept on
        the device (a Python list as index is uploaded by every call: a synchronous copy in front of the launch)."""
        qc = self._qcols
        if qc == list(range(qc[0], qc[0] + len(qc))):
            return t[:, qc[0]:qc[0] + len(qc)]
        ix = self.__dict__.get("_qcols_t")
        if ix is None or ix.device != t.device:
            ix = self._qcols_t = torch.as_tensor(qc, dtype=torch.long, device=t.device)
        return t.index_select(1, ix)

    def inverse_kinematics(self, link, pos=None, quat=None, init_qpos=None, envs_idx=None, return_error=False, **opts):
        """``robot.inverse_kinematics(link=eef, pos=(B,3), quat=(B,4), init_qpos=..., envs_idx=...)`` -> (B, n_dofs)
        (/root/reference/examples/franka/pick_cube_state.py:46-51).  One batched launch for all envs; `envs_idx` selects
        the rows returned (and addressed by pos / quat / init_qpos) as in Genesis.  `pos` is required (a quat-only target
        is not on the reference's path).  (The reference's experts pass full-batch targets with `envs_idx=arange(B)` on every step of
        their loops: that case costs the launch and one row gather -- no scatter, no clone, no index upload.)"""
        if pos is None:
            raise ValueError("inverse_kinematics needs a target position")
        mir = self._mir
        B = mir.num_envs
        idx = None if envs_idx is None else torch.as_tensor(envs_idx, device=mir.device).long().reshape(-1)
        qc = self._qcols
        if IK_ROWS and hasattr(mir, "inverse_kinematics_rows") and qc == list(range(qc[0], qc[0] + len(qc))):
            # ONE launch (mir_inverse_kinematics_rows): the kernel addresses targets and seeds by row or by env itself and writes row k for
            # env envs_idx[k] -- what the scatter / clone / gather around the full-batch launch below did with five torch kernels
            n = B if idx is None else int(idx.numel())

            def arg(t, k, by_env_flag):
                t = torch.as_tensor(t, dtype=torch.float32, device=mir.device)
                if k == 4 and (t.numel() == 4 or (t.dim() == 2 and t.stride(0) == 0)):  # (one quaternion, e.g. `.expand(B, -1)`: read in place)
                    return (t.reshape(-1, 4)[0] if t.numel() == 4 else t[0]).contiguous(), IK_QUAT_ONE
                t = t.reshape(-1, k).contiguous()
                if t.shape[0] == B:
                    return t, by_env_flag
                if t.shape[0] != n:
                    raise ValueError(f"inverse_kinematics: expected {n} or {B} rows of {k}, got {tuple(t.shape)}")
                return t, 0

            p, fp = arg(pos, 3, IK_POS_BY_ENV)
            q, fq = (None, 0) if quat is None else arg(quat, 4, IK_QUAT_BY_ENV)
            iq, fi = (None, 0) if init_qpos is None else arg(init_qpos, len(qc), IK_INIT_BY_ENV)
            if CHECK_IK_IDX and idx is not None and bool(((idx < 0) | (idx >= B)).any()):
                raise IndexError("inverse_kinematics: envs_idx outside the batch")
            res = mir.inverse_kinematics_rows(link.idx, p, q, iq, None if idx is None else idx.contiguous(), fp | fq | fi, qc[0], len(qc) if iq is not None else 0,
                                              return_error=return_error, **opts)
            qout, err = (res if return_error else (res, None))
            if len(qc) != qout.shape[1]:
                qout = qout[:, qc[0]:qc[0] + len(qc)].contiguous()
            return (qout, err) if return_error else qout

        def full(t, k):
            """-> (tensor of B rows, owned): owned = a tensor made here (rows scattered to `idx`), which may be edited in place; a
            full-batch argument comes back as it is -- possibly the caller's own tensor, never written"""
            if t is None:
                return None, False
            t = torch.as_tensor(t, dtype=torch.float32, device=mir.device).reshape(-1, k)
            if idx is None or t.shape[0] == B:
                return t, False
            out = torch.zeros((B, k), dtype=torch.float32, device=mir.device)
            out[idx] = t
            return out, True

        (p, _), (q, q_owned) = full(pos, 3), full(quat, 4)
        if q is not None and q_owned:
            q[:, 0] += (q.abs().sum(1) == 0).float()  # unaddressed rows: identity, so normalisation stays finite
        init = None
        if init_qpos is not None:
            qc = self._qcols
            iq = torch.as_tensor(init_qpos, dtype=torch.float32, device=mir.device).reshape(-1, len(qc))
            cur = mir.get_state()[0][:, :mir.n_arm].clone()
            if idx is not None and iq.shape[0] != B:
                cur[idx[:, None], torch.as_tensor(qc, device=mir.device)[None, :]] = iq
            elif qc == list(range(qc[0], qc[0] + len(qc))):
                cur[:, qc[0]:qc[0] + len(qc)] = iq
            else:
                cur[:, qc] = iq
            init = cur
        res = mir.inverse_kinematics(link.idx, p, q, init, return_error=return_error, **opts)
        qout, err = (res if return_error else (res, None))
        qout = self._cols(qout).contiguous()
        if idx is not None:
            qout = qout[idx]
            err = None if err is None else err[idx]
        return (qout, err) if return_error else qout

    def control_dofs_position(self, position, dofs_idx_local=None) -> None:
        """PD targets for a subset of this entity's dofs (others keep their current target)."""
        tgt = self._mir.get_state()[2]
        pos = torch.as_tensor(position, dtype=torch.float32, device=tgt.device)
        idx = list(range(self.n_dofs)) if dofs_idx_local is None else [int(i) for i in np.asarray(dofs_idx_local).ravel()]
        cols = [self._ucols[i] for i in idx]
        tgt[:, cols] = pos.reshape(tgt.shape[0], len(cols))
        self._mir.set_pd_targets(tgt)


class CameraView:
    """``scene.add_camera(res, pos, lookat, fov)`` + ``cam.set_pose`` + ``cam.render()``
    (/root/reference/gym_genesis/tasks/franka/cube_pick.py:56-63,166-176; /root/reference/gym_genesis/env.py:97-98).
    ``render()`` draws ALL envs of this process at their grid offsets from the camera's current pose, like the
    reference's single shared scene, and returns ``(rgb, None, None, None)`` with ``rgb`` a NumPy uint8 (H, W, 3)
    array (Genesis returns rgb, depth, segmentation, normal).  ``render_envs()`` is the batched per-env image
    tensor used for the ``pixels`` observation: one launch instead of the reference's B renders."""

    def __init__(self, mir, builder, scene, res, pos, lookat, fov, up=(0.0, 0.0, 1.0)):
        from ..backend.spec import make_camera

        self._mir, self._scene = mir, scene
        self.res = (int(res[0]), int(res[1]))
        self.fov = float(fov)
        self._make = make_camera
        self._vis = builder.visual()
        self._up = tuple(up)
        self._home = (tuple(float(v) for v in pos), tuple(float(v) for v in lookat))
        self.set_pose(pos=pos, lookat=lookat)
        self._offsets = torch.as_tensor(np.ascontiguousarray(scene.envs_offset, dtype=np.float32), device=mir.device)

    def set_pose(self, pos=None, lookat=None) -> None:
        if pos is not None:
            self.pos = tuple(float(v) for v in np.asarray(pos).ravel())
        if lookat is not None:
            self.lookat = tuple(float(v) for v in np.asarray(lookat).ravel())

    def _spec(self, pos, lookat):
        # (the ctypes struct of a pose is built once: the observation renders ask for the same few poses every step)
        key = (tuple(pos), tuple(lookat), self.res, self.fov, self._up)
        cache = self.__dict__.setdefault("_specs", {})
        spec = cache.get(key)
        if spec is None:
            if len(cache) > 64:
                cache.clear()
            spec = cache[key] = self._make(self.res[0], self.res[1], pos, lookat, self.fov, self._up)
        return spec

    def render_global(self) -> torch.Tensor:
        """(H, W, 3) uint8 device tensor: every env at its grid offset, camera at its current pose."""
        img = self._mir.render(self._spec(self.pos, self.lookat), self._vis, mode=1, env_offset=self._offsets)
        # (what render() may reuse: the image is the tensor handed out as observation['pixels'], so its in-place version counter is kept
        #  too -- an edit by the caller, a normalisation or an overlay, makes render() draw again)
        self._last_global = (getattr(self._mir, "state_version", None), self.pos, self.lookat, self.res, self.fov, self._up, id(self._vis),
                             img, img._version)
        self._record(img)
        return img

    def render_envs(self, pos=None, lookat=None, out=None) -> torch.Tensor:
        """(B, H, W, 3) uint8 device tensor: env i alone, seen from `pos` -> `lookat` relative to the env's origin
        (defaults: the pose the camera was created with)."""
        return self._mir.render(self._spec(self._home[0] if pos is None else pos, self._home[1] if lookat is None else lookat),
                                self._vis, mode=0, out=out)

    def render_cams(self, pos, lookat, up=None) -> torch.Tensor:
        """(B, H, W, 3) uint8 device tensor: env i seen from its OWN camera pos[i] -> lookat[i] (B,3 tensors in the env's
        frame; the link-mounted wrist cameras)."""
        return self._mir.render_cams(self._spec(self._home[0], self._home[1]), self._vis, pos, lookat, up)

    def render(self, rgb=True, depth=False, segmentation=False, normal=False):
        if depth or segmentation or normal:
            raise NotImplementedError("only the rgb output of cam.render() is on the reference's path (env.py:98)")
        # (the reference's README loop calls env.render() right behind an env.step() whose pixels observation IS this image: nothing has
        #  moved since -- mir_get_state_version -- so it is copied out, not drawn again; the caller gets a fresh array either way)
        last = self.__dict__.get("_last_global")
        ver = getattr(self._mir, "state_version", None)
        same = (last is not None and ver is not None and last[:7] == (ver, self.pos, self.lookat, self.res, self.fov, self._up, id(self._vis))
                and last[7]._version == last[8])
        if same:
            img = last[7]
            self._record(img)  # (a frame per call, as in Genesis)
        else:
            img = self.render_global()
        return self._to_host(img), None, None, None

    _PIN_MAX = 8  # pinned image buffers a camera lends out at a time

    def _to_host(self, img) -> np.ndarray:
        """A fresh NumPy array of a device image.  The array IS a pinned buffer the device copies into (27 us for 480 x 640): the camera
        lends out up to `_PIN_MAX` of them and takes one back when its array is garbage-collected (`weakref.finalize`; a view keeps its
        base alive, so nothing the caller still holds is ever reused) -- the README loop drops each frame before it asks for the next and
        never pays a host memcpy.  A caller that keeps more frames than that gets copies from a staging buffer (60 - 74 us; `img.cpu()`,
        which this replaces, copies into pageable memory: 67 us per image when the frames are dropped, 224 us when they are kept and
        every array is new pages the driver has to fault in and pin)."""
        if not img.is_cuda:
            return img.cpu().numpy()
        import weakref

        d = self.__dict__
        key = (tuple(img.shape), img.dtype)
        if d.get("_pin_key") != key:
            d["_pin_key"], d["_pin_pool"], d["_pin_stage"] = key, [], None   # (buffers of another size still out die with their arrays)
        pool = d["_pin_pool"]
        buf = pool.pop() if pool else (torch.empty(img.shape, dtype=img.dtype, pin_memory=True) if d.get("_pin_out", 0) < self._PIN_MAX else None)
        if buf is None:
            stage = d.get("_pin_stage")
            if stage is None:
                stage = d["_pin_stage"] = torch.empty(img.shape, dtype=img.dtype, pin_memory=True)
            stage.copy_(img, non_blocking=True)
            torch.cuda.current_stream(img.device).synchronize()
            return stage.numpy().copy()
        buf.copy_(img, non_blocking=True)
        torch.cuda.current_stream(img.device).synchronize()
        arr = buf.numpy()
        d["_pin_out"] = d.get("_pin_out", 0) + 1
        weakref.finalize(arr, self._pin_back, buf, key)
        return arr

    def _pin_back(self, buf, key) -> None:
        d = self.__dict__
        d["_pin_out"] = d.get("_pin_out", 1) - 1
        if d.get("_pin_key") == key and len(d["_pin_pool"]) < self._PIN_MAX:
            d["_pin_pool"].append(buf)

    # ---- recording (cam.start_recording() at reset -- the reference: always with pixels, cube_pick.py:109-110; here: when the env was
    # made with record_video=True, see env.py; env.save_video -> cam.stop_recording: env.py:71-79).  Genesis appends the image of every cam.render() while a recording runs.  Here the GLOBAL renders are recorded --
    # render() and the `global` pixels observation -- as references to the device tensors they returned (no copy, nothing on the
    # env.step() path); the per-env observation images (one launch for all envs, into a reused buffer) are not: the reference's B
    # separate renders of a moving camera make no film either.  At most `max_recorded_frames` are kept (the newest).
    max_recorded_frames = 1024

    @property
    def recording(self) -> bool:
        return bool(self.__dict__.get("_recording"))

    def start_recording(self) -> None:
        self._recording = True
        self._frames = []
        self._frames_dropped = 0

    def _record(self, img) -> None:
        if self.__dict__.get("_recording"):
            fr = self._frames
            fr.append((img, getattr(img, "_version", None)))
            if len(fr) > self.max_recorded_frames:
                del fr[0]
                self._frames_dropped += 1

    def stop_recording(self, save_to_filename=None, fps=60) -> None:
        """Ends the recording and writes it: `.mp4` as Motion-JPEG in an ISO media file, `.gif` (tasks/video.py: this image has no H.264
        encoder).  No file name: the frames are dropped, like Genesis."""
        import warnings

        frames, dropped = self.__dict__.get("_frames") or [], self.__dict__.get("_frames_dropped", 0)
        was_on = bool(self.__dict__.get("_recording"))
        self._recording, self._frames, self._frames_dropped = False, [], 0
        if not was_on:
            # (the reference's collectors call cam.stop_recording("top.mp4") after every episode -- so_101/collect_task_stack_cube_batch.py:
            #  199-201 -- and its tasks start a recording at every reset; here that takes record_video=True: the loop goes on)
            warnings.warn(f"no recording is running (GenesisEnv(..., record_video=True) starts one at every reset, or call "
                          f"cam.start_recording()): nothing was written to {save_to_filename!r}", stacklevel=2)
            return
        if save_to_filename is None:
            return
        if dropped:
            warnings.warn(f"the recording keeps the last {self.max_recorded_frames} frames: {dropped} older ones were dropped", stacklevel=2)
        if not frames:
            warnings.warn(f"no global render since the recording started: nothing was written to {save_to_filename!r}", stacklevel=2)
            return
        edited = sum(1 for f, v in frames if v is not None and f._version != v)
        if edited:  # (frames are references to the tensors handed out as observations, not copies)
            warnings.warn(f"{edited} recorded frames were modified in place after they were handed out: the file shows them as they are now",
                          stacklevel=2)
        from .video import write_video

        write_video(save_to_filename, [self._to_host(f) if hasattr(f, "cpu") else np.asarray(f) for f, _ in frames], fps)


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
