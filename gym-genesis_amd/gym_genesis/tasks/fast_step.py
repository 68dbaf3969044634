"""GenesisEnv.step as ONE flat function for the state-only batched tasks (gym_genesis/env.py:61-69 of the reference).

Every Python frame and attribute lookup between two launches is time the GPU idles, so the whole step is a closure over
pre-bound callables.  Order: validate the action, launch (mir_step_go: the outputs were registered while the previous kernel
ran), then -- while this kernel runs -- make everything the API returns besides `terminated` and prepare the next call, then
wait for the launch's terminated bytes (mir_step_end) and return.

The outputs registered for the NEXT launch live in the scene's one slot cache (`mir._fresh`, shared with StepHelpers.step_fresh:
whoever registered last is what the library holds), and an exception between launch and wait closes the step before it propagates.
"""
import numpy as np
import torch


def make_fast_step(task, mir, action_dim: int, agent_obs: int, env_obs: int, coerce=None):
    """-> step(action) returning (observation, reward, terminated, truncated, info) like GenesisEnv.step.

    `task` keeps `_agent / _envst / _reward / _term` pointing at the tensors of the latest step (get_obs(), views);
    `coerce(action)` (optional) is the task's own pre-processing of an action that is not already a (B, action_dim) float32
    device tensor (e.g. the SO-101 tasks' reshape)."""
    go, prepare, alloc, end, as_action = mir.step_go_ptr, mir.step_prepare_ptrs, mir._alloc_outputs, mir.step_end_ptr, mir.as_action
    stage = mir.stage_action  # host actions (NumPy, lists, CPU tensors): read by the launch in place from pinned memory
    B, dev, tensor, f32, tbool = task.num_envs, task.device, torch.Tensor, torch.float32, torch.bool
    shape = torch.Size((B, action_dim))
    np_empty, np_zeros, np_bool = np.empty, np.zeros, np.bool_
    fresh = mir.__dict__.setdefault("_fresh", {})   # the scene's slot cache: the outputs registered with the library for the next launch
    key = (agent_obs, env_obs, True)
    # the next reset's spawn draws, a chunk per step while the kernel runs (tasks/spawn_ahead.py); complete within ~128 steps
    ahead = getattr(task, "_ahead", None)
    chunk = max(1024, -(-ahead.total // 128)) if ahead is not None else 0

    def fast_step(action):
        if not (type(action) is tensor and action.dtype is f32 and action.shape == shape and action.is_contiguous()
                and action.device == dev):
            if coerce is not None:
                action = coerce(action)
            if isinstance(action, tensor) and action.is_cuda:
                action = as_action(action, action_dim)  # (kept alive in `action` until the launch is queued: alloc() below must not get its block)
                aptr = action.data_ptr()
            else:
                aptr = stage(action, action_dim)
        else:
            aptr = action.data_ptr()
        slot = fresh.pop(key, None)
        if slot is None:
            slot = alloc(agent_obs, env_obs)
            prepare(slot[1])
        go(aptr)
        # ---- the kernel is running: nothing below is on the critical path until end()
        try:
            outs = slot[0]
            host = np_empty(B, np_bool)
            n = alloc(agent_obs, env_obs)
            prepare(n[1])
            fresh[key] = n
            if ahead is not None and ahead.left:
                ahead.advance(chunk)
            task._agent, task._envst, task._reward, task._term = outs
            observation = {"agent_pos": outs[0], "environment_state": outs[1]}
            info = {"is_success": outs[3].view(tbool)}
            truncated = np_zeros(B, np_bool)
        except BaseException:
            fresh.pop(key, None)
            end(0)  # (close the step: the env stays usable)
            raise
        end(host.ctypes.data)
        return observation, outs[2], host, truncated, info

    fc = getattr(mir, "fast_calls", None)
    fc = fc() if fc is not None else None
    if fc is None:
        return fast_step

    # the same function over the _mirfast built-ins (csrc/mir_pyfast.c): no ctypes conversion and no Python frame between the
    # arrival of the terminated bytes and the next launch
    fprepare, fgo, fend, hbox, raw_stream, devidx = fc
    check = mir._check

    def fast_step_builtin(action):
        if not (type(action) is tensor and action.dtype is f32 and action.shape == shape and action.is_contiguous()
                and action.device == dev):
            if coerce is not None:
                action = coerce(action)
            if isinstance(action, tensor) and action.is_cuda:
                action = as_action(action, action_dim)
                aptr = action.data_ptr()
            else:
                aptr = stage(action, action_dim)
        else:
            aptr = action.data_ptr()
        h = hbox[0]
        slot = fresh.pop(key, None)
        if slot is None:
            slot = alloc(agent_obs, env_obs)
            p = slot[1]
            rc = fprepare(h, p[0], p[1], p[2], p[3])
            if rc:
                check(rc)
        rc = fgo(h, aptr, raw_stream(devidx))
        if rc:
            check(rc)
        # ---- the kernel is running
        try:
            outs = slot[0]
            host = np_empty(B, np_bool)
            host_ptr = host.ctypes.data
            n = alloc(agent_obs, env_obs)
            p = n[1]
            rc = fprepare(h, p[0], p[1], p[2], p[3])
            if rc:
                check(rc)
            fresh[key] = n
            task._agent, task._envst, task._reward, task._term = outs
            if chunk and ahead.left:
                ahead.advance(chunk)
            result = ({"agent_pos": outs[0], "environment_state": outs[1]}, outs[2], host, np_zeros(B, np_bool), {"is_success": outs[3].view(tbool)})
        except BaseException:
            fresh.pop(key, None)
            fend(h, 0)  # (close the step: the env stays usable)
            raise
        rc = fend(h, host_ptr)
        if rc:
            check(rc)
        return result

    return fast_step_builtin
