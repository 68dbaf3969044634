"""ctypes binding of libmirigid.so (include/mirigid.h) and a thin torch-facing wrapper.

There is deliberately NO CPU fallback: if the HIP library is missing or no GPU is
visible, creating a scene raises.  PyTorch is used only to own device buffers and
to supply the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from .spec import IK_DEFAULTS, MIR_VERSION, MirCameraSpec, MirDims, MirIkOptions, MirIkRows, MirSceneSpec, MirVisualSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libmirigid.so"))

_lib = None


class MirError(RuntimeError):
    pass


class MirMaskError(MirError):
    """MIR_E_MASK: terminated bytes that a step kernel handed over before its solve had finished turned out wrong (include/mirigid.h,
    mir_step_begin).  A mask returned since the last successful call cannot be trusted; the scene continues with late bytes."""


def load_library() -> C.CDLL:
    """Load libmirigid.so (built in-tree by ``make -C gym-genesis_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirError(f"{LIB_PATH} not found: build it with `make -C gym-genesis_amd/csrc` "
                       "(python __graft_entry__.py build); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    lib.mir_version.restype = C.c_int
    lib.mir_spec_sizeof.restype = C.c_int
    lib.mir_last_error.restype = C.c_char_p
    lib.mir_create.argtypes = [C.POINTER(MirSceneSpec), i32, i32, C.POINTER(vp)]
    lib.mir_destroy.argtypes = [vp]
    lib.mir_get_dims.argtypes = [vp, C.POINTER(MirDims)]
    lib.mir_get_model_consts.argtypes = [vp, vp, vp, vp]
    lib.mir_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_autoreset.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.mir_set_pd_targets.argtypes = [vp, vp, vp]
    lib.mir_step.argtypes = [vp, i32, vp]
    lib.mir_step_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.mir_step_begin.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.mir_step_begin.restype = C.c_int
    lib.mir_step_end.argtypes = [vp, vp]
    lib.mir_step_end.restype = C.c_int
    lib.mir_step_prepare.argtypes = [vp, vp, vp, vp, vp]
    lib.mir_step_prepare.restype = C.c_int
    lib.mir_step_go.argtypes = [vp, vp, vp]
    lib.mir_step_go.restype = C.c_int
    lib.mir_get_sync_mode.argtypes = [vp]
    lib.mir_get_sync_mode.restype = C.c_int
    lib.mir_debug_rotated_launches.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.mir_debug_rotated_launches.restype = C.c_int
    lib.mir_get_split_step.argtypes = [vp]
    lib.mir_get_split_step.restype = C.c_int
    lib.mir_debug_null_roundtrip.argtypes = [vp, i32, vp, C.POINTER(C.c_double)]
    lib.mir_debug_null_roundtrip.restype = C.c_int
    lib.mir_step_packed.argtypes = [vp, vp, vp, i32, vp]
    lib.mir_step_packed.restype = C.c_int
    lib.mir_rollout.argtypes = [vp, vp, i32, vp, i32, vp]
    lib.mir_rollout.restype = C.c_int
    lib.mir_rollout_autoreset.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.mir_rollout_autoreset.restype = C.c_int
    lib.mir_get_obs.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_set_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_get_links.argtypes = [vp, vp, vp, vp]
    lib.mir_get_diag.argtypes = [vp, vp, vp, vp, vp]
    lib.mir_set_diag.argtypes = [vp, i32]
    lib.mir_set_diag.restype = C.c_int
    lib.mir_forward.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_debug_render_path.argtypes = [vp, i32, i32]
    lib.mir_debug_render_path.restype = C.c_int
    lib.mir_debug_spec_active.argtypes = [vp]
    lib.mir_debug_spec_active.restype = C.c_int
    lib.mir_debug_early_mask_stats.argtypes = [vp, C.POINTER(C.c_uint32), i32, vp]
    lib.mir_debug_early_mask_stats.restype = C.c_int
    lib.mir_get_diag4.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mir_get_diag4.restype = C.c_int
    lib.mir_get_bad.argtypes = [vp, vp, C.POINTER(C.c_uint32), i32, vp]
    lib.mir_get_bad.restype = C.c_int
    lib.mir_debug_raise_mask_flag.argtypes = [vp]
    lib.mir_debug_raise_mask_flag.restype = C.c_int
    lib.mir_get_early_mask.argtypes = [vp]
    lib.mir_get_early_mask.restype = C.c_int
    lib.mir_get_state_version.argtypes = [vp]
    lib.mir_get_state_version.restype = C.c_int
    lib.mir_set_exact_contacts.argtypes = [vp, C.POINTER(MirSceneSpec), i32]
    lib.mir_set_exact_contacts.restype = C.c_int
    lib.mir_get_exact_contacts.argtypes = [vp]
    lib.mir_get_exact_contacts.restype = C.c_int
    lib.mir_get_exact_stats.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    lib.mir_get_exact_stats.restype = C.c_int
    lib.mir_get_exact_route.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.mir_get_exact_route.restype = C.c_int
    lib.mir_render.argtypes = [vp, C.POINTER(MirCameraSpec), C.POINTER(MirVisualSpec), i32, vp, vp, vp]
    lib.mir_render.restype = C.c_int
    lib.mir_visual_sizeof.restype = C.c_int
    lib.mir_render_cams.argtypes = [vp, C.POINTER(MirCameraSpec), C.POINTER(MirVisualSpec), vp, vp, vp, vp, vp]
    lib.mir_render_cams.restype = C.c_int
    lib.mir_inverse_kinematics.argtypes = [vp, i32, vp, vp, vp, C.POINTER(MirIkOptions), vp, vp, vp]
    lib.mir_inverse_kinematics.restype = C.c_int
    lib.mir_inverse_kinematics_rows.argtypes = [vp, i32, C.POINTER(MirIkRows), vp, vp, vp, C.POINTER(MirIkOptions), vp, vp, vp]
    lib.mir_inverse_kinematics_rows.restype = C.c_int
    for name in ("mir_create", "mir_destroy", "mir_get_dims", "mir_get_model_consts", "mir_reset", "mir_autoreset", "mir_set_pd_targets",
                 "mir_step", "mir_step_fused", "mir_get_obs", "mir_get_state", "mir_set_state", "mir_get_links",
                 "mir_get_diag", "mir_forward"):
        getattr(lib, name).restype = C.c_int
    if lib.mir_spec_sizeof() != C.sizeof(MirSceneSpec) or lib.mir_version() != MIR_VERSION or lib.mir_visual_sizeof() != C.sizeof(MirVisualSpec):
        raise MirError("libmirigid.so ABI mismatch with gym_genesis.backend.spec (rebuild the library)")
    _lib = lib
    _bind_fast(lib)
    return lib


_fast = None  # the _mirfast built-ins (csrc/mir_pyfast.c), bound to the loaded library; None = ctypes is used for those calls too


def _bind_fast(lib) -> None:
    """The three calls between two env.step launches as CPython built-ins (~0.06 us per call instead of ctypes' ~0.39 us: that
    difference is GPU idle time).  They are the SAME library functions, bound by address; without the module the ctypes route is
    taken, and a warning says so."""
    global _fast
    try:
        from . import _mirfast
    except ImportError as e:
        import warnings

        warnings.warn(f"gym_genesis.backend._mirfast is not built ({e}); env.step uses ctypes calls (run `make -C gym-genesis_amd/csrc`)", RuntimeWarning)
        return
    addr = lambda f: C.cast(f, C.c_void_p).value  # noqa: E731
    _mirfast.bind(addr(lib.mir_step_prepare), addr(lib.mir_step_go), addr(lib.mir_step_end))
    _fast = _mirfast


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# torch.cuda.current_stream(device).cuda_stream costs ~1.5 us per call (two Python objects); the raw getter is ~0.1 us.  It sits
# in front of every launch of the API path, where the GPU is idle while the host prepares the call.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class StepHelpers:
    """Host-side helpers of the API path, shared by MirScene and the CPU test double (tests/fake_scene.py); they need
    `num_envs`, `device` and `step_fused` of the scene."""

    # ---- API-path helpers (GenesisEnv.step): the GPU idles while Python prepares a launch, so nothing that can wait is done
    # before it -- the output tensors of the NEXT call are allocated while this call's kernel runs
    def as_action(self, action, dim: int) -> torch.Tensor:
        """(B, dim) float32 contiguous device tensor from a torch tensor (device or CPU) or anything NumPy understands."""
        if type(action) is torch.Tensor and action.dtype is torch.float32 and action.device == self.device and action.is_contiguous():
            if action.shape[0] != self.num_envs or action.dim() != 2 or action.shape[1] != dim:
                raise ValueError(f"action must have shape {(self.num_envs, dim)}, got {tuple(action.shape)}")
            return action
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action))
        action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.shape != (self.num_envs, dim):
            raise ValueError(f"action must have shape {(self.num_envs, dim)}, got {tuple(action.shape)}")
        return action

    def stage_action(self, action, dim: int) -> int:
        """Address of a (B, dim) float32 copy of a HOST action (NumPy array, list, CPU tensor) in pinned memory that the next step
        launch reads IN PLACE over PCIe (147 KB for 4096 x 9: one more read in the kernel's first batch of loads) -- no copy
        command, no device allocation.  `GenesisEnv.step(env.action_space.sample())` is how the reference is driven
        (reference env.py:61); a pageable array sent through `tensor.to(device)` costs ~25 us per step.  Two buffers take turns:
        a buffer is rewritten two steps later, and a step is closed (mir_step_end: the launch has read its action long before
        its terminated bytes leave) before the next one begins.  Only for the begin / end step path (fast_step)."""
        # (a host tensor may carry requires_grad or be a subclass: detach; device tensors never come here -- see the callers)
        a = np.asarray(action.detach().cpu().numpy() if isinstance(action, torch.Tensor) else action, dtype=np.float32)
        if a.shape != (self.num_envs, dim):
            raise ValueError(f"action must have shape {(self.num_envs, dim)}, got {tuple(a.shape)}")
        ring = self.__dict__.get("_act_stage")
        if ring is None or ring["dim"] != dim:
            bufs = [torch.empty((self.num_envs, dim), dtype=torch.float32) for _ in range(2)]
            if torch.device(self.device).type == "cuda":  # (the CPU test double reads the same address from plain memory)
                bufs = [b.pin_memory() for b in bufs]
            ring = self._act_stage = {"dim": dim, "np": [b.numpy() for b in bufs], "ptr": [b.data_ptr() for b in bufs], "bufs": bufs, "k": 0}
        k = ring["k"]
        ring["k"] = k ^ 1
        np.copyto(ring["np"][k], a)
        return ring["ptr"][k]

    def use_output_ring(self, steps: int, agent_dim: int, env_dim: int) -> torch.Tensor:
        """From now on the outputs of consecutive steps are consecutive ROWS of one preallocated (steps, B * (agent + env + 1)) float32
        buffer (row = [agent_pos (B, a) | environment_state (B, e) | reward (B)] of one step) instead of fresh allocations: the
        send buffer of a sharded run's observation gather IS the place the kernels write to (bench.py, sharding.CopyPathGather), no
        concatenation, no per-step allocation.  A row is overwritten `steps` steps later: what the caller keeps from a step is
        valid for that long (the default -- fresh tensors every step, like the reference -- is unchanged).  Returns the buffer."""
        B, w = self.num_envs, agent_dim + env_dim + 1
        buf = torch.empty((steps, B * w), dtype=torch.float32, device=self.device)
        term = torch.empty((steps, B), dtype=torch.uint8, device=self.device)
        slots = []
        for i in range(steps):
            row, base = buf[i], buf[i].data_ptr()
            outs = (row[:agent_dim * B].view(B, agent_dim), row[agent_dim * B:(agent_dim + env_dim) * B].view(B, env_dim),
                    row[(agent_dim + env_dim) * B:], term[i])
            slots.append((outs, (base, base + 4 * agent_dim * B, base + 4 * (agent_dim + env_dim) * B, term[i].data_ptr())))
        self.__dict__.get("_fresh", {}).clear()   # (outputs registered ahead came from the allocator)
        self._ring = {"dims": (agent_dim, env_dim), "slots": slots, "pos": 0, "n": steps, "buf": buf, "term": term}
        return buf

    def _alloc_outputs(self, agent_dim: int, env_dim: int):
        """Fresh output tensors of one step, with their device addresses (so the launch itself does no attribute lookups):
        ((agent_pos, environment_state, reward, terminated u8), (ptr, ptr, ptr, ptr))."""
        ring = self.__dict__.get("_ring")
        if ring is not None and ring["dims"] == (agent_dim, env_dim):
            i = ring["pos"]
            ring["pos"] = i + 1 if i + 1 < ring["n"] else 0
            return ring["slots"][i]
        B = self.num_envs
        buf = torch.empty(B * (agent_dim + env_dim + 1), dtype=torch.float32, device=self.device)
        term = torch.empty(B, dtype=torch.uint8, device=self.device)
        base = buf.data_ptr()
        outs = (buf[:agent_dim * B].view(B, agent_dim), buf[agent_dim * B:(agent_dim + env_dim) * B].view(B, env_dim),
                buf[(agent_dim + env_dim) * B:], term)
        return outs, (base, base + 4 * agent_dim * B, base + 4 * (agent_dim + env_dim) * B, term.data_ptr())

    def step_fresh(self, action, agent_dim: int, env_dim: int, host_terminated: bool = False):
        """(`action`: anything as_action / stage_action accept.)  One fused step into FRESH output tensors (callers may keep old observations, as with the reference):
        returns (agent_pos, environment_state, reward, terminated u8).  With host_terminated the launch also delivers the
        terminated bytes to the host (step_begin); the caller must then close the step with step_end().
        The GPU idles while Python prepares a launch, so nothing that can wait is done before it: the output tensors of the
        NEXT call (and, on the host path, their registration with the library) and the host array of THIS call are made
        while the kernel runs."""
        key = (agent_dim, env_dim, host_terminated)
        fresh = self.__dict__.get("_fresh")
        if fresh is None:
            fresh = self._fresh = {}
        slot = fresh.pop(key, None)
        if slot is None:
            slot = self._alloc_outputs(agent_dim, env_dim)
            if host_terminated:
                self.step_prepare_ptrs(slot[1])
        outs, ptrs = slot
        if host_terminated:
            # (a HOST action -- NumPy, list, CPU tensor -- is staged in pinned memory and read in place: stage_action; the step is
            #  closed by step_end() before the buffer comes round again)
            # (any tensor on a GPU -- this device or another, a subclass, a Parameter -- goes through as_action's .to(device); the
            #  converted tensor is held until step_end() has returned: with exact contacts the launches for the deferred envs read the
            #  action AGAIN from there, on the library's side stream -- ADVICE r5; include/mirigid.h: mir_set_exact_contacts)
            if isinstance(action, torch.Tensor) and action.is_cuda:
                action = self.as_action(action, self.nu)
                self._pend_action_ref = action
                self.step_go_ptr(action.data_ptr())
            else:
                self.step_go_ptr(self.stage_action(action, self.nu))
            host = np.empty(self.num_envs, dtype=np.bool_)
            self._host_pending = (host, host.ctypes.data)
        else:
            self.step_fused_ptrs(self.as_action(action, self.nu).data_ptr(), ptrs)
        nxt = self._alloc_outputs(agent_dim, env_dim)  # (while the kernel runs)
        if host_terminated:
            self.step_prepare_ptrs(nxt[1])
        fresh[key] = nxt
        return outs

    # pointer-level launches; the CPU test double (tests/fake_scene.py) overrides these
    def step_prepare_ptrs(self, ptrs):
        raise NotImplementedError

    def step_go_ptr(self, action_ptr):
        raise NotImplementedError

    def step_fused_ptrs(self, action_ptr, ptrs):
        raise NotImplementedError


class MirScene(StepHelpers):
    """A compiled, batched scene living on one GPU (``scene.build(n_envs=B)``)."""

    def __init__(self, spec: MirSceneSpec, num_envs: int, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise MirError("no HIP device visible: gym_genesis (MI355X backend) has no CPU path")
        self.lib = load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.spec = spec
        h = C.c_void_p()
        if self.device.index is None:  # "cuda" without an index means the CURRENT device, not device 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        rc = self.lib.mir_create(C.byref(spec), int(num_envs), self.device.index, C.byref(h))
        self._check(rc)
        self.h = h
        self._hbox = [h.value]
        d = MirDims()
        self._check(self.lib.mir_get_dims(self.h, C.byref(d)))
        self.num_envs, self.nbody, self.nq, self.nv = d.num_envs, d.nbody, d.nq, d.nv
        self.ngeom, self.npair, self.agent_dim, self.env_dim = d.ngeom, d.npair, d.agent_dim, d.env_dim
        self.nfree, self.kernel = d.nfree, d.kernel
        self._go, self._devidx = self.lib.mir_step_go, self.device.index  # (looked up once: they sit in front of every API launch)
        self.nu = sum(1 for i in range(spec.ndof) if spec.dof[i].ctrl_mode == 1)
        self.n_arm = sum(1 for b in range(1, spec.nbody) if spec.body[b].jtype in (1, 2))

    # -- helpers -----------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise (MirMaskError if rc == -5 else MirError)(f"libmirigid error {rc}: {self.lib.mir_last_error().decode()}")

    def _stream(self):
        if _raw_stream is not None:
            return _raw_stream(self.device.index)  # int; the c_void_p argtype converts it
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, t, *cols: int) -> torch.Tensor:
        """Accept torch (any device) or NumPy, return a contiguous f32 device tensor (B, *cols)."""
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        # (a buffer handed out by staged() is read in place by the kernel -- its reuse is guarded by an event; any other host
        #  tensor, pinned or not, is copied: the caller may overwrite it as soon as this call returns)
        if not (t.device.type == "cpu" and t.data_ptr() in self.__dict__.get("_staged_ptrs", ())):
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if t.shape != (self.num_envs, *cols):
            raise ValueError(f"expected shape {(self.num_envs, *cols)}, got {tuple(t.shape)}")
        return t

    def _free(self, t, k: int) -> torch.Tensor:
        """Free-body poses: (B, nfree, k); (B, k) is accepted when the scene has one free body."""
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        if t.dim() == 2 and self.nfree == 1:
            t = t.unsqueeze(1)
        return self._f32(t, self.nfree, k)

    def staged(self, rows: np.ndarray) -> torch.Tensor:
        """NumPy float32 rows -> a pinned host tensor that the next reset() reads IN PLACE (its kernel loads the rows over PCIe).
        A pageable array handed to reset() goes through a synchronous copy on the DMA engine instead: 13 us when the engine is
        awake and a few hundred when it is not, on a path (reset-all every 200 steps) where nothing else waits for the GPU.  Two
        buffers per shape take turns; a buffer is reused only after the launch that read it has finished (event)."""
        ring = self.__dict__.setdefault("_stage", {})
        slot = ring.get(rows.shape)
        if slot is None:
            slot = ring[rows.shape] = {"buf": [torch.empty(rows.shape, dtype=torch.float32).pin_memory() for _ in range(2)],
                                       "ev": [torch.cuda.Event(), torch.cuda.Event()], "used": [False, False], "k": 0}
            self.__dict__.setdefault("_staged_ptrs", set()).update(b.data_ptr() for b in slot["buf"])
        k = slot["k"]
        slot["k"] = k ^ 1
        if slot["used"][k]:
            slot["ev"][k].synchronize()
        np.copyto(slot["buf"][k].numpy(), rows)
        self._stage_pending = (slot, k)
        return slot["buf"][k]

    def empty(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty((self.num_envs, *shape), dtype=dtype, device=self.device)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.mir_destroy(self.h)
            self.h = None
            self._hbox[0] = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- C ABI, one method per entry point -------------------------------------------------------
    def model_consts(self):
        dw = np.zeros(self.nv)
        bw = np.zeros(self.nbody)
        mi = C.c_double()
        self._check(self.lib.mir_get_model_consts(self.h, dw.ctypes.data_as(C.c_void_p), bw.ctypes.data_as(C.c_void_p),
                                                  C.cast(C.byref(mi), C.c_void_p)))
        return dw, bw, mi.value

    def reset(self, obj_pos, obj_quat, arm_qpos, env_mask: Optional[torch.Tensor] = None) -> None:
        p = self._free(obj_pos, 3)   # all free bodies in body order (one for the pick scenes)
        q = self._free(obj_quat, 4)
        a = self._f32(arm_qpos, self.n_arm)
        mk = None
        if env_mask is not None:
            mk = torch.as_tensor(env_mask).to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(self.lib.mir_reset(self.h, _ptr(p), _ptr(q), _ptr(a), _ptr(mk), self._stream()))
        pend = self.__dict__.pop("_stage_pending", None)
        if pend is not None:  # (the launch above reads a staged() buffer: it may be refilled once this event has passed)
            slot, k = pend
            slot["ev"][k].record(torch.cuda.current_stream(self.device))
            slot["used"][k] = True

    def autoreset(self, terminated, episode_len, max_len: int, spawn_pool, cursor, obj_quat, arm_qpos,
                  truncated_out=None, done_out=None) -> None:
        """Device-side episode bookkeeping + re-spawn of finished envs (mir_autoreset); every argument is a
        preallocated device tensor of the dtype the header names."""
        self._check(self.lib.mir_autoreset(self.h, _ptr(terminated), _ptr(episode_len), int(max_len), _ptr(spawn_pool),
                                           int(spawn_pool.shape[0]), _ptr(cursor), _ptr(obj_quat), _ptr(arm_qpos),
                                           _ptr(truncated_out), _ptr(done_out), self._stream()))

    def set_pd_targets(self, tgt) -> None:
        t = self._f32(tgt, self.nu)
        self._check(self.lib.mir_set_pd_targets(self.h, _ptr(t), self._stream()))

    def step(self, n_steps: int = 1) -> None:
        self._check(self.lib.mir_step(self.h, int(n_steps), self._stream()))

    def step_fused(self, action, agent_pos, env_state, reward, terminated) -> None:
        """Raw launch: all arguments are preallocated device tensors (action may be None)."""
        self._check(self.lib.mir_step_fused(self.h, _ptr(action), _ptr(agent_pos), _ptr(env_state), _ptr(reward),
                                            _ptr(terminated), self._stream()))

    def step_begin(self, action, agent_pos, env_state, reward, terminated) -> None:
        """mir_step_begin: step_fused whose terminated bytes also go to the host; close with step_end()."""
        self._check(self.lib.mir_step_begin(self.h, _ptr(action), _ptr(agent_pos), _ptr(env_state), _ptr(reward),
                                            _ptr(terminated), self._stream()))
        self._host_pending = None
        self._pend_action_ref = action  # (read again by the launches of step_end() when exact contacts are on)

    def step_prepare_ptrs(self, ptrs) -> None:
        rc = self.lib.mir_step_prepare(self.h, ptrs[0], ptrs[1], ptrs[2], ptrs[3])
        if rc:
            self._check(rc)

    def step_go_ptr(self, action_ptr) -> None:
        rc = self._go(self.h, action_ptr, _raw_stream(self._devidx) if _raw_stream is not None else self._stream())
        if rc:
            self._check(rc)

    def fast_calls(self):
        """-> (prepare, go, end, handle_box, raw_stream, device_index) for the flat env.step closure (tasks/fast_step.py): the
        _mirfast built-ins with everything they need as plain ints (handle_box[0] = the handle's address, 0 once the scene is
        closed: the library then answers "null MirHandle"), or None where ctypes has to do (no module, no raw stream getter)."""
        if _fast is None or _raw_stream is None:
            return None
        return _fast.prepare, _fast.go, _fast.end, self._hbox, _raw_stream, self._devidx

    def step_fused_ptrs(self, action_ptr, ptrs) -> None:
        rc = self.lib.mir_step_fused(self.h, action_ptr, ptrs[0], ptrs[1], ptrs[2], ptrs[3], self._stream())
        if rc:
            self._check(rc)

    def step_end(self) -> np.ndarray:
        """mir_step_end: wait for the launch of step_begin; a FRESH NumPy bool (B,) = terminated (env.py:64)."""
        pend = self.__dict__.get("_host_pending")
        if pend is None:  # (step_begin called directly: nothing was prepared during the kernel)
            host = np.empty(self.num_envs, dtype=np.bool_)
            pend = (host, host.ctypes.data)
        self._host_pending = None
        rc = self.lib.mir_step_end(self.h, pend[1])
        self._pend_action_ref = None
        if rc:
            self._check(rc)
        return pend[0]

    def step_end_ptr(self, host_ptr: int) -> None:
        """mir_step_end into a caller-provided host array (address): the flat fast path of GenesisEnv.step."""
        rc = self.lib.mir_step_end(self.h, host_ptr)
        self._pend_action_ref = None
        if rc:
            self._check(rc)

    def null_roundtrip_us(self, iters: int = 2000) -> float:
        """mir_debug_null_roundtrip: microseconds per empty-kernel launch + host-visible completion (the floor under env.step)."""
        out = C.c_double()
        self._check(self.lib.mir_debug_null_roundtrip(self.h, int(iters), self._stream(), C.byref(out)))
        return out.value

    @property
    def sync_mode(self) -> int:
        return int(self.lib.mir_get_sync_mode(self.h))

    def rotated_launches(self, actions: torch.Tensor, n: int, outputs=None) -> None:
        """mir_debug_rotated_launches: n back-to-back rotated launches cycling through actions (K,B,nu) (bench.py times them);
        `outputs` = (agent_pos, env_state, reward, terminated) tensors: every launch also writes them, like the launches of
        GenesisEnv.step (whose host-visible terminated bytes are left out: see mir_api.hip)."""
        outs = None if outputs is None else (C.c_void_p * 4)(*[t.data_ptr() for t in outputs])
        self._check(self.lib.mir_debug_rotated_launches(self.h, _ptr(actions), int(actions.shape[0]), int(n), outs, self._stream()))

    @property
    def split_step(self) -> int:
        """How step_begin launches (mir_get_split_step): 1 rotated, 2 two launches, 0 fused."""
        return int(self.lib.mir_get_split_step(self.h))

    def step_packed(self, action, rows: torch.Tensor) -> None:
        """One step; all outputs in one (B, row_stride) float32 row tensor (see mir_step_packed)."""
        self._check(self.lib.mir_step_packed(self.h, _ptr(action), _ptr(rows), int(rows.stride(0)), self._stream()))

    def rollout(self, actions: torch.Tensor, rows: torch.Tensor) -> None:
        """K env steps in one launch (mir_rollout): actions (K,B,nu), rows (K,B,row_stride) float32 device tensors."""
        K = actions.shape[0]
        if tuple(actions.shape) != (K, self.num_envs, self.nu) or rows.shape[0] != K or rows.shape[1] != self.num_envs or not (
                actions.is_contiguous() and rows.is_contiguous()):
            raise ValueError("rollout: actions (K,B,nu) and rows (K,B,row_stride) must be contiguous device tensors")
        self._check(self.lib.mir_rollout(self.h, _ptr(actions), int(K), _ptr(rows), int(rows.stride(1)), self._stream()))

    def rollout_autoreset(self, actions: torch.Tensor, rows: torch.Tensor, episode_len, max_len: int, spawn_pool, cursor, obj_quat,
                          arm_qpos) -> None:
        """K env steps + the device-side episode loop in one launch (mir_rollout_autoreset); rows need one spare column
        (truncated) after [agent | env_state | reward | terminated]."""
        K = actions.shape[0]
        if tuple(actions.shape) != (K, self.num_envs, self.nu) or rows.shape[0] != K or rows.shape[1] != self.num_envs or not (
                actions.is_contiguous() and rows.is_contiguous()):
            raise ValueError("rollout_autoreset: actions (K,B,nu) and rows (K,B,row_stride) must be contiguous device tensors")
        self._check(self.lib.mir_rollout_autoreset(self.h, _ptr(actions), int(K), _ptr(rows), int(rows.stride(1)), _ptr(episode_len), int(max_len),
                                                   _ptr(spawn_pool), int(spawn_pool.shape[0]), _ptr(cursor), _ptr(obj_quat), _ptr(arm_qpos),
                                                   self._stream()))

    def get_obs(self):
        agent, env = self.empty(self.agent_dim), self.empty(self.env_dim)
        rew, term = self.empty(), self.empty(dtype=torch.uint8)
        self._check(self.lib.mir_get_obs(self.h, _ptr(agent), _ptr(env), _ptr(rew), _ptr(term), self._stream()))
        return agent, env, rew, term

    def get_state(self):
        q, v = self.empty(self.nq), self.empty(self.nv)
        t, w = self.empty(self.nu), self.empty(self.nv)
        self._check(self.lib.mir_get_state(self.h, _ptr(q), _ptr(v), _ptr(t), _ptr(w), self._stream()))
        return q, v, t, w

    def set_state(self, qpos=None, qvel=None, target=None, warmstart=None) -> None:
        q = None if qpos is None else self._f32(qpos, self.nq)
        v = None if qvel is None else self._f32(qvel, self.nv)
        t = None if target is None else self._f32(target, self.nu)
        w = None if warmstart is None else self._f32(warmstart, self.nv)
        self._check(self.lib.mir_set_state(self.h, _ptr(q), _ptr(v), _ptr(t), _ptr(w), self._stream()))

    def get_links(self):
        pos, quat = self.empty(self.nbody, 3), self.empty(self.nbody, 4)
        self._check(self.lib.mir_get_links(self.h, _ptr(pos), _ptr(quat), self._stream()))
        return pos, quat

    def set_diag(self, on: bool) -> None:
        """Switch the per-env solver diagnostics (ncon / nefc / niter, 16 B per env-step) on or off (mir_set_diag)."""
        self._check(self.lib.mir_set_diag(self.h, 1 if on else 0))

    def get_diag(self, points: bool = False):
        """(ncon, nefc, niter) of the last step, int32 (B,) each; points=True adds the candidate contact points found before the
        contact capacity was applied (mir_get_diag4: more than `max_contacts` = the manifolds were thinned)."""
        a, b, c = (self.empty(dtype=torch.int32) for _ in range(3))
        if points:
            d = self.empty(dtype=torch.int32)
            self._check(self.lib.mir_get_diag4(self.h, _ptr(a), _ptr(b), _ptr(c), _ptr(d), self._stream()))
            return a, b, c, d
        self._check(self.lib.mir_get_diag(self.h, _ptr(a), _ptr(b), _ptr(c), self._stream()))
        return a, b, c

    def get_bad(self, reset: bool = False):
        """Divergence guard (mir_get_bad; diagnostics must be on): (uint8 (B,) device tensor, 1 = the env's state was non-finite after
        the last step launch; env-steps flagged since the counter was last reset)."""
        bad = self.empty(dtype=torch.uint8)
        n = C.c_uint32()
        self._check(self.lib.mir_get_bad(self.h, _ptr(bad), C.byref(n), 1 if reset else 0, self._stream()))
        return bad, int(n.value)

    def forward(self):
        M, bias = self.empty(self.nv, self.nv), self.empty(self.nv)
        qas, qacc = self.empty(self.nv), self.empty(self.nv)
        self._check(self.lib.mir_forward(self.h, _ptr(M), _ptr(bias), _ptr(qas), _ptr(qacc), self._stream()))
        return M, bias, qas, qacc

    def render(self, cam: MirCameraSpec, vis: MirVisualSpec, mode: int = 0, env_offset: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """RGB8 images of the current state (mir_render): (B,H,W,3) per-env, or (H,W,3) in global mode."""
        shape = (cam.height, cam.width, 3) if mode == 1 else (self.num_envs, cam.height, cam.width, 3)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous uint8 tensor of shape {shape}")
        off = None
        if env_offset is not None:
            off = self._f32(env_offset, 3)
        self._check(self.lib.mir_render(self.h, C.byref(cam), C.byref(vis), int(mode), _ptr(off), _ptr(out), self._stream()))
        return out

    @property
    def spec_active(self) -> bool:
        """True when this scene runs the scene-specialised instantiation of the 16-lane kernel (mir_debug_spec_active)."""
        return bool(self.lib.mir_debug_spec_active(self.h))

    def early_mask_stats(self, reset: bool = False):
        """(workgroup-launches that sent their terminated bytes from inside the solver loop, workgroup-launches whose early bytes
        differed from the integrated state: must be 0 -- the library also fails the next call with MIR_E_MASK) -- mir_debug_early_mask_stats."""
        out = (C.c_uint32 * 2)()
        self._check(self.lib.mir_debug_early_mask_stats(self.h, out, 1 if reset else 0, self._stream()))
        return int(out[0]), int(out[1])

    def set_exact_contacts(self, on=True) -> None:
        """mir_set_exact_contacts: from now on the begin / end step path defers every env whose narrowphase finds more contact
        points than the 16-lane kernel keeps (16) and steps it with 48 points, never thinned below that -- on the list instantiation of
        the 16-lane kernel (three contacts per lane), on the wave-per-env kernel what exceeds that too; the other envs are computed as
        before.  step() / step_fused() then wait for their step; step_packed() / rollout*() are refused.
        on="all" (tests): EVERY env of every step takes the deferred envs' route -- the twin a deferred env is compared with."""
        self._check(self.lib.mir_set_exact_contacts(self.h, C.byref(self.spec), 2 if on == "all" else (1 if on else 0)))

    def exact_route(self) -> dict:
        """mir_get_exact_route: deferred env-steps handed to the list instantiation / env-steps stepped by the wave-per-env kernel."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.mir_get_exact_route(self.h, out))
        return {"list_env_steps": int(out[0]), "wave_env_steps": int(out[1]), "heavy_steps": int(out[2]), "big_steps": int(out[3])}

    @property
    def exact_contacts(self) -> bool:
        return bool(self.lib.mir_get_exact_contacts(self.h))

    def exact_stats(self, reset: bool = False) -> dict:
        """mir_get_exact_stats: steps closed by step_end, steps that had deferred envs, deferred env-steps, most deferred envs in a step."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.mir_get_exact_stats(self.h, out, 1 if reset else 0))
        return {"steps": int(out[0]), "overflow_steps": int(out[1]), "overflow_env_steps": int(out[2]), "overflow_envs_max": int(out[3])}

    @property
    def early_mask(self) -> bool:
        """True while mir_step_begin launches may send their terminated bytes early (mir_get_early_mask)."""
        return bool(self.lib.mir_get_early_mask(self.h))

    @property
    def state_version(self) -> int:
        """mir_get_state_version: advances with every call that moves the bodies (steps, resets, state writes)."""
        return int(self.lib.mir_get_state_version(self.h))

    def debug_render_path(self, generic: bool = False, strip_rows: int = 0) -> None:
        """mir_debug_render_path: force the generic pixel kernel / override the strip height for the following renders."""
        self._check(self.lib.mir_debug_render_path(self.h, 1 if generic else 0, int(strip_rows)))

    def render_cams(self, cam: MirCameraSpec, vis: MirVisualSpec, cam_pos, cam_lookat, cam_up=None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B,H,W,3) RGB8 images, env i seen from its own camera cam_pos[i] -> cam_lookat[i] (mir_render_cams)."""
        shape = (self.num_envs, cam.height, cam.width, 3)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        p, l = self._f32(cam_pos, 3), self._f32(cam_lookat, 3)
        u = None if cam_up is None else self._f32(cam_up, 3)
        self._check(self.lib.mir_render_cams(self.h, C.byref(cam), C.byref(vis), _ptr(p), _ptr(l), _ptr(u), _ptr(out), self._stream()))
        return out

    def inverse_kinematics(self, link_body: int, pos, quat=None, init_qpos=None, return_error: bool = False, **opts):
        """Batched damped-least-squares IK (mir_inverse_kinematics): (B, n_arm) joint positions that bring body
        `link_body` to pos (B,3) / quat (B,4 wxyz, optional).  Seed = init_qpos or the scene's current joint positions."""
        p = self._f32(pos, 3)
        q = None if quat is None else self._f32(quat, 4)
        init = None if init_qpos is None else self._f32(init_qpos, self.n_arm)
        o = MirIkOptions(**{**IK_DEFAULTS, **opts})
        out, err = self.empty(self.n_arm), self.empty(2)
        self._check(self.lib.mir_inverse_kinematics(self.h, int(link_body), _ptr(p), _ptr(q), _ptr(init), C.byref(o), _ptr(out), _ptr(err),
                                                    self._stream()))
        return (out, err) if return_error else out

    def inverse_kinematics_rows(self, link_body: int, pos, quat, init_qpos, env_idx, flags: int, init_col0: int = 0, init_ncols: int = 0,
                                return_error: bool = False, **opts):
        """mir_inverse_kinematics_rows: the solver for the rows `env_idx` (int64 device tensor, or None: every env) in ONE launch.
        pos / quat / init_qpos: contiguous float32 device tensors addressed as `flags` say (spec.IK_*); -> (n_rows, n_arm)[, (n_rows, 2)]."""
        n = self.num_envs if env_idx is None else int(env_idx.numel())
        o = self.__dict__.get("_ik_default")
        if opts or o is None:
            o = MirIkOptions(**{**IK_DEFAULTS, **opts})
            if not opts:
                self._ik_default = o
        rows = MirIkRows(None if env_idx is None else env_idx.data_ptr(), n, int(flags), int(init_col0), int(init_ncols))
        out = torch.empty((n, self.n_arm), dtype=torch.float32, device=self.device)
        err = torch.empty((n, 2), dtype=torch.float32, device=self.device) if return_error else None
        self._check(self.lib.mir_inverse_kinematics_rows(self.h, int(link_body), C.byref(rows), _ptr(pos), _ptr(quat), _ptr(init_qpos), C.byref(o),
                                                         _ptr(out), _ptr(err), self._stream()))
        return (out, err) if return_error else out
