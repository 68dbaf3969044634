"""The NEXT reset's spawn draws, made ahead of time in small chunks while a step kernel runs.

The reference draws the spawn of every env from the task's `np.random.RandomState` inside `reset()` (x block, then y block:
gym_genesis/tasks/franka/cube_pick.py:90-91 of the reference).  A sharded run must draw the GLOBAL block on every rank so that its
results do not depend on the number of ranks -- 2 x 32 768 doubles at 8 x 4096 envs, ~0.25 ms of host time per
reset on every rank, in front of a launch.  The legacy MT19937 stream is consumed value by value, so `uniform(lo, hi, n)` drawn as
several shorter `uniform(lo, hi, k)` calls yields the same numbers bit for bit (tests/test_host_cpu.py checks it against the golden
fixture): the block is therefore drawn a chunk per `env.step()`, in the window where the host waits for the kernel anyway
(tasks/fast_step.py), and `reset()` finds it complete.  Whatever is still missing when a reset arrives is drawn there.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


class UniformBlocksAhead:
    """blocks = [(lo, hi, n), ...]: the uniform blocks one reset draws, in the order it draws them."""

    def __init__(self, blocks: Sequence[Tuple[float, float, int]]):
        self.blocks = [(float(lo), float(hi), int(n)) for lo, hi, n in blocks]
        self.total = sum(n for _, _, n in self.blocks)
        self._buf = np.empty(self.total, dtype=np.float64)
        self._rs = None     # the RandomState the block in progress is drawn from
        self._pos = 0       # values of it drawn so far
        self.left = 0       # values still to draw ahead (0: nothing to do) -- the step closure tests this attribute

    def invalidate(self) -> None:
        self._rs, self._pos, self.left = None, 0, 0

    def start(self, rs: np.random.RandomState) -> None:
        """Begin the next block from `rs` (called right after a reset has taken the previous one)."""
        self._rs, self._pos, self.left = rs, 0, self.total

    def advance(self, n: int) -> None:
        """Draw up to n more values of the block in progress (one uniform() call per block touched: the bounds differ)."""
        rs, pos, base = self._rs, self._pos, 0
        if rs is None:
            return
        for lo, hi, cnt in self.blocks:
            if n > 0 and pos < base + cnt:
                k = min(n, base + cnt - pos)
                self._buf[pos:pos + k] = rs.uniform(lo, hi, size=(k,))
                pos += k
                n -= k
            base += cnt
        self._pos = pos
        self.left = self.total - pos

    def take(self, rs: np.random.RandomState) -> List[np.ndarray]:
        """The complete block as float64 arrays, one per entry of `blocks` (views of an internal buffer: use before the next
        advance()).  Values drawn ahead are used only if they came from this very RandomState object; anything missing is drawn now."""
        if self._rs is not rs:
            self.start(rs)
        while self.left:
            self.advance(self.left)
        out, base = [], 0
        for _, _, cnt in self.blocks:
            out.append(self._buf[base:base + cnt])
            base += cnt
        self._rs, self._pos, self.left = None, 0, 0
        return out
