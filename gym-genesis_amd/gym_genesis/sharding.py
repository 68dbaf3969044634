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


class CopyPathGather:
    """The observation gather on the COPY path: every rank pushes its block straight into the receive buffers of all ranks with
    peer-to-peer device copies (SDMA engines over xGMI), and a 4-byte sequence word behind each block tells the receiver it has
    landed.  No kernel runs for the gather on any GPU.

    Why not `all_gather_into_tensor`: the step kernel of the pick tasks keeps all 4096 envs of a rank co-resident -- 1024 two-wave
    workgroups x 40 KB of LDS = every CU's 160 KB and 464 of its 512 VGPRs per SIMD lane.  RCCL's collective kernel
    (`rcclGenericKernel`: 256 threads, 19.7 KB of LDS, ~280 registers per thread -- read off the gfx950 code object of the librccl.so
    that ships with torch) cannot share a CU with four such workgroups; every CU it occupies displaces two of them, and a displaced
    workgroup starts when another one ends, i.e. the launch takes ~1.5 x as long for as long as the collective is resident.  With
    2.4 MB per rank and step coming in over seven links the collective would be resident most of the time.  The copy engines take
    nothing from the CUs.  RCCL stays in charge of what it is good at here: rendezvous, barriers, the handle exchange below and the
    max-over-ranks reduction of the timings.

    Protocol.  `recv[slot, r]` on every rank is rank r's block of the gather in flight in `slot` (two slots: the collective of one
    chunk overlaps the steps of the next); `flag[slot, r]` holds the sequence number of the last block that has completely arrived
    from r.  push(): on a side stream, one device-to-device copy of the block to each rank's recv[slot, me], then one 4-byte copy of
    the sequence word to each rank's flag[slot, me] -- copies on one stream complete in order, so the word never overtakes its
    block.  ready() / wait(): all `world` words of the slot show the sequence number.  The buffers are shared between the processes
    through HIP IPC handles (torch.multiprocessing.reductions: the same mechanism torch uses to pass CUDA tensors between
    processes; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack), exchanged once with all_gather_object.
    The consumer of slot s must be done with it before the push after next (the cadence of the caller; the bench consumes nothing).

    `verify()` checks one gather against `all_gather_into_tensor` and agrees on the outcome across ranks: a stack where peer access
    or IPC does not work falls back to the RCCL collective on every rank alike."""

    SEQ_TABLE = 1 << 16

    def __init__(self, numel: int, device: torch.device, group=None, dtype=torch.float32):
        from torch.multiprocessing.reductions import reduce_tensor

        self.group, self.device, self.numel = group, device, int(numel)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.recv = torch.zeros((2, self.world, self.numel), dtype=dtype, device=device)
        self.flag = torch.zeros((2, self.world), dtype=torch.int32, device=device)
        self._seqs = torch.arange(1, self.SEQ_TABLE + 1, dtype=torch.int32, device=device)
        self._seq_base = 0
        self.pushes = 0
        self.stream = torch.cuda.Stream(device=device)
        mine = (reduce_tensor(self.recv), reduce_tensor(self.flag))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self.peer_recv, self.peer_flag = [], []
        for r, ((f_recv, a_recv), (f_flag, a_flag)) in enumerate(handles):
            if r == self.rank:
                self.peer_recv.append(self.recv)
                self.peer_flag.append(self.flag)
            else:
                self.peer_recv.append(f_recv(*a_recv))   # rank r's buffer, mapped into this process
                self.peer_flag.append(f_flag(*a_flag))
        # the pushes themselves go through ONE library call (mir_p2p_push: 2 x world hipMemcpyAsync from C, ~2 us each) instead of
        # 2 x world torch copy_ calls (device guards and cross-device event traffic: ~10 us each) -- this is host time in front of
        # a launch.  Without the library (CPU tests of the module's import) the torch path is used.
        self._lib = None
        try:
            import ctypes as C

            from .backend.lib import load_library
            lib = load_library()
            lib.mir_p2p_enable.argtypes = [C.c_int32, C.c_int32]
            lib.mir_p2p_push.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
            for t in self.peer_recv:
                if t.device.index != device.index and lib.mir_p2p_enable(device.index, t.device.index) != 0:
                    raise RuntimeError(lib.mir_last_error().decode())
            esz = self.recv.element_size()
            # destination addresses of this rank's block in every rank's buffers, per slot
            self._dst = [(C.c_void_p * self.world)(*[self.peer_recv[p][slot, self.rank].data_ptr() for p in range(self.world)]) for slot in (0, 1)]
            self._fdst = [(C.c_void_p * self.world)(*[self.peer_flag[p][slot, self.rank:self.rank + 1].data_ptr() for p in range(self.world)]) for slot in (0, 1)]
            self._esz, self._seq_ptr, self._lib = esz, self._seqs.data_ptr(), lib
        except (ImportError, OSError, AttributeError):
            self._lib = None

    def push(self, block: torch.Tensor) -> int:
        """Send `block` (<= numel elements, contiguous, on this rank's device) to every rank; -> its sequence number (for wait())."""
        n = block.numel()
        if n > self.numel:
            raise ValueError(f"block of {n} elements, buffers hold {self.numel}")
        k = self.pushes
        slot = k & 1
        if k - self._seq_base >= self.SEQ_TABLE:  # (one small kernel every 65536 pushes)
            self.stream.synchronize()
            self._seq_base += self.SEQ_TABLE
            self._seqs.add_(self.SEQ_TABLE)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))   # the block's producer
        if self._lib is not None:
            if not block.is_contiguous() or block.dtype != self.recv.dtype:
                raise ValueError("push: contiguous block of the buffers' dtype expected")
            rc = self._lib.mir_p2p_push(self._dst[slot], self.world, block.data_ptr(), n * self._esz, self._fdst[slot],
                                        self._seq_ptr + 4 * (k - self._seq_base), self.stream.cuda_stream)
            if rc != 0:
                raise RuntimeError(f"mir_p2p_push: {self._lib.mir_last_error().decode()}")
        else:
            word = self._seqs[k - self._seq_base:k - self._seq_base + 1]
            with torch.cuda.stream(self.stream):
                for p in range(self.world):
                    self.peer_recv[p][slot, self.rank, :n].copy_(block.reshape(-1), non_blocking=True)
                for p in range(self.world):
                    self.peer_flag[p][slot, self.rank:self.rank + 1].copy_(word, non_blocking=True)
        block.record_stream(self.stream)
        self.pushes = k + 1
        return k + 1

    def ready(self, seq: int) -> bool:
        """Have the blocks of push number `seq` arrived from every rank?  (One tiny reduction + host read: for the consumer's side of
        the protocol and for the end of a timed region, not for the step loop.)"""
        slot = (seq - 1) & 1
        return bool((self.flag[slot] >= seq).all().item())

    def wait(self, seq: int, timeout_s: float = 30.0) -> None:
        import time

        t0 = time.perf_counter()
        while not self.ready(seq):
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f"copy-path gather: block {seq} did not arrive from every rank within {timeout_s} s (flags {self.flag.tolist()})")

    def gathered(self, seq: int, n: int) -> torch.Tensor:
        """(world, n) view of the blocks of push `seq` (after wait())."""
        return self.recv[(seq - 1) & 1, :, :n]

    def verify(self) -> bool:
        """One gather of a rank-specific pattern against all_gather_into_tensor; the verdict is the same on every rank."""
        ok = 1
        try:
            n = min(self.numel, 4096)
            block = torch.arange(n, dtype=torch.float32, device=self.device) + 1000.0 * (self.rank + 1)
            want = torch.empty((self.world, n), dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(want.reshape(-1), block, group=self.group)
            seq = self.push(block.to(self.recv.dtype))
            self.wait(seq, timeout_s=20.0)
            ok = int(torch.equal(self.gathered(seq, n).float(), want))
        except Exception:  # noqa: BLE001
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())


def make_copy_gather(numel: int, device: torch.device, group=None):
    """-> (CopyPathGather, "") when the copy path works on EVERY rank, else (None, reason): construction (IPC handle exchange, peer
    mappings) and verify() are each followed by an agreement across the ranks, so either all ranks use the copy path or none does."""
    cg, err = None, ""
    try:
        cg = CopyPathGather(numel, device, group)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    t = torch.tensor([1 if cg is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    if not bool(t.item()):
        return None, "peer mapping failed on some rank" + (f" (here: {err})" if err else "")
    if not cg.verify():
        return None, "verification against all_gather_into_tensor failed on some rank"
    return cg, ""
