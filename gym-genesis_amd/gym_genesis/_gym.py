"""gymnasium, or a minimal stand-in when it is not installed.

The reference subclasses ``gymnasium.Env`` and registers ids with
``gymnasium.envs.registration.register`` (/root/reference/gym_genesis/env.py:13,
/root/reference/gym_genesis/__init__.py:1-37).  gymnasium is not part of this image, so the
package falls back to the few pieces of its surface the hot path touches: ``Env``,
``spaces.Box``/``spaces.Dict``, ``register``/``make`` with a ``TimeLimit`` wrapper.
"""
from __future__ import annotations

import importlib
from typing import Any, Callable, Dict, Optional

import numpy as np

try:  # pragma: no cover - exercised only where gymnasium exists
    import gymnasium as gym
    from gymnasium import spaces
    from gymnasium.envs.registration import register

    make = gym.make
    Env = gym.Env
    HAVE_GYMNASIUM = True
except ModuleNotFoundError:
    HAVE_GYMNASIUM = False

    class Env:  # noqa: D401 - minimal gymnasium.Env
        metadata: Dict[str, Any] = {}
        observation_space = None
        action_space = None
        render_mode = None
        spec = None

        def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
            if seed is not None:
                self._np_random = np.random.default_rng(seed)
            return None

        def step(self, action):
            raise NotImplementedError

        def render(self):
            return None

        def close(self):
            pass

        @property
        def unwrapped(self):
            return self

    class _Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            super().__init__(shape, dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, self.dtype)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class Dict_(_Space):
        def __init__(self, spaces_dict):
            super().__init__()
            self.spaces = dict(spaces_dict)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def seed(self, seed=None):
            for i, s in enumerate(self.spaces.values()):
                s.seed(None if seed is None else seed + i)
            return [seed]

    class _Spaces:
        Box = Box
        Dict = Dict_

    spaces = _Spaces()

    class TimeLimit:
        """gymnasium.wrappers.TimeLimit: ``truncated = True`` once ``max_episode_steps`` elapsed."""

        def __init__(self, env, max_episode_steps: int):
            self.env = env
            self._max = max_episode_steps
            self._t = 0

        def __getattr__(self, name):
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env

        def reset(self, **kw):
            self._t = 0
            return self.env.reset(**kw)

        def step(self, action):
            obs, rew, term, trunc, info = self.env.step(action)
            self._t += 1
            if self._t >= self._max:
                trunc = True
            return obs, rew, term, trunc, info

    _REGISTRY: Dict[str, dict] = {}

    def register(id: str, entry_point: str, max_episode_steps: Optional[int] = None, nondeterministic: bool = False,
                 kwargs: Optional[dict] = None, **_):
        _REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=dict(kwargs or {}))

    def make(id: str, **kwargs):
        if id not in _REGISTRY:
            raise KeyError(f"unknown environment id {id!r}")
        e = _REGISTRY[id]
        mod, cls = e["entry_point"].split(":")
        ctor: Callable = getattr(importlib.import_module(mod), cls)
        kw = dict(e["kwargs"])
        kw.update(kwargs)
        env = ctor(**kw)
        if e["max_episode_steps"]:
            env = TimeLimit(env, e["max_episode_steps"])
        return env
