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


import weakref

# block sizes of gather_rows(num_envs=None), per process group OBJECT (an id() can be reused after the group is collected) and local size
_COUNTS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_COUNTS_DEFAULT: dict = {}  # the default group (None) has no object to hang the cache on


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
        try:
            cache = _COUNTS_DEFAULT if group is None else _COUNTS.setdefault(group, {})
        except TypeError:  # (a group object that cannot be weakly referenced: no caching)
            cache = {}
        counts = cache.get((world, n))
        if counts is None:
            sizes = torch.tensor([n], dtype=torch.int64, device=rows.device)
            all_sizes = torch.empty((world,), dtype=torch.int64, device=rows.device)
            dist.all_gather_into_tensor(all_sizes, sizes, group=group)
            counts = cache[(world, n)] = [int(c) for c in all_sizes.tolist()]
            if counts[dist.get_rank(group)] != n:
                raise RuntimeError("gather_rows: inconsistent block sizes across ranks")
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

    Protocol.  `recv[slot, r]` on every rank is rank r's block of the gather in flight in `slot` (NSLOT = 2: the gather of one chunk
    overlaps the steps of the next); `flag[slot, r]` holds the sequence number of the last block that has completely arrived from r;
    `ack[r]` the number of the last gather rank r has RELEASED.
      * push(block) -> seq: destination p's copy of the block and, behind it, its sequence word go out on stream p of this rank (one
        stream per destination: the links run side by side; copies on one stream complete in order, so a word never overtakes its
        block).  The streams first wait for the producer of the block (the caller's current stream).  An event per push marks the
        moment the block has been READ by every copy: wait_source(seq) makes the caller's stream wait for it before the memory of
        that block is written again (a ring of step outputs), and lag() counts the pushes whose copies have not finished.
      * flow control (flow_control=True): push number k does not start before every rank has released gather k - NSLOT -- the slot
        it overwrites on the peers -- waiting on the host if it has to (back-pressure on the producer, with a timeout).
      * consume: ready(seq) / wait(seq) say that all `world` blocks of gather `seq` have landed here, gathered(seq, n) is the view,
        release(seq) tells every rank that this one is done with it (4-byte words on the same streams, ordered behind the caller's
        current stream, i.e. behind the consumer's reads).
    The buffers are shared between the processes through HIP IPC handles (torch.multiprocessing.reductions: the mechanism torch
    uses to pass CUDA tensors between processes; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack), exchanged once with
    all_gather_object.  Without flow control (bench.py: nothing consumes the blocks there) a slot is simply overwritten every NSLOT
    pushes; what the bench does instead is compare the LAST gather of every timed region with `all_gather_into_tensor` of the same
    chunk (check_against_collective) and report lag().

    `verify()` checks one gather against `all_gather_into_tensor` and agrees on the outcome across ranks: a stack where peer access
    or IPC does not work falls back to the RCCL collective on every rank alike (make_copy_gather)."""

    SEQ_TABLE = 1 << 16
    SEQ_KEEP = 64   # words kept BELOW the table's base after a rebase: a release() of a gather pushed just before the rebase finds its word
    NSLOT = 2
    EVENT_DEPTH = 64  # pushes whose read-completion events are kept (wait_source / lag look this far back)

    def __init__(self, numel: int, device: torch.device, group=None, dtype=torch.float32, flow_control: bool = False):
        from torch.multiprocessing.reductions import reduce_tensor

        self.group, self.device, self.numel = group, device, int(numel)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.flow_control = bool(flow_control)
        # ---- local work that can fail comes FIRST, and the ranks agree on its outcome before the first collective that moves
        # objects: a rank that failed must not skip a collective the others are inside of
        local_err = None
        try:
            self.recv = torch.zeros((self.NSLOT, self.world, self.numel), dtype=dtype, device=device)
            self.flag = torch.zeros((self.NSLOT, self.world), dtype=torch.int32, device=device)
            self.ack = torch.zeros((self.world,), dtype=torch.int32, device=device)
            # word i = sequence number _seq_base - SEQ_KEEP + 1 + i: the table reaches SEQ_KEEP numbers back behind its base, so that the
            # consumer's release(k) still finds its word when push k + 1 has just rebased the table (it was dropped, the ack
            # stayed at k - 1 and the next push timed out under flow control)
            self._seqs = torch.arange(1 - self.SEQ_KEEP, self.SEQ_TABLE + 1, dtype=torch.int32, device=device)
            mine = (reduce_tensor(self.recv), reduce_tensor(self.flag), reduce_tensor(self.ack))
        except Exception as e:  # noqa: BLE001
            local_err, mine = e, None
        self._agree(local_err, "allocating / exporting the gather buffers")
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self._seq_base = 0
        self.pushes = 0
        self._acked = 0           # every rank is known to have released gathers up to this number
        self._released = 0
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.world)]
        self._ack_stream = torch.cuda.Stream(device=device)
        self._events: dict = {}   # seq -> [event per stream]: the block of push `seq` has been read by every copy
        self._lib = None
        try:
            self.peer_recv, self.peer_flag, self.peer_ack = [], [], []
            for r, ((f_recv, a_recv), (f_flag, a_flag), (f_ack, a_ack)) in enumerate(handles):
                if r == self.rank:
                    self.peer_recv.append(self.recv); self.peer_flag.append(self.flag); self.peer_ack.append(self.ack)
                else:
                    self.peer_recv.append(f_recv(*a_recv))   # rank r's buffers, mapped into this process
                    self.peer_flag.append(f_flag(*a_flag))
                    self.peer_ack.append(f_ack(*a_ack))
            # the pushes themselves go through ONE library call (mir_p2p_push_streams: 2 x world hipMemcpyAsync from C, ~2 us each)
            # instead of 2 x world torch copy_ calls (device guards and cross-device event traffic: ~10 us each) -- this is host time
            # in front of a launch.  Without the library (CPU tests of the module's import) the torch path is used.
            import ctypes as C

            try:
                from .backend.lib import load_library
                lib = load_library()
                lib.mir_p2p_enable.argtypes = [C.c_int32, C.c_int32]
                lib.mir_p2p_push_streams.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
            except (ImportError, OSError, AttributeError):
                lib = None
            if lib is not None:
                for t in self.peer_recv:   # (a peer on this rank's own device -- ranks sharing a GPU in the plumbing tests -- needs nothing: the call says so)
                    if lib.mir_p2p_enable(device.index, t.device.index) != 0:
                        raise RuntimeError(lib.mir_last_error().decode())
                vp = lambda xs: (C.c_void_p * self.world)(*xs)  # noqa: E731
                # destination addresses of this rank's block / word in every rank's buffers, per slot; of its ack word
                self._dst = [vp([self.peer_recv[p][slot, self.rank].data_ptr() for p in range(self.world)]) for slot in range(self.NSLOT)]
                self._fdst = [vp([self.peer_flag[p][slot, self.rank:self.rank + 1].data_ptr() for p in range(self.world)]) for slot in range(self.NSLOT)]
                self._adst = vp([self.peer_ack[p][self.rank:self.rank + 1].data_ptr() for p in range(self.world)])
                self._sptr = vp([st.cuda_stream for st in self.streams])
                self._esz, self._seq_ptr, self._lib = self.recv.element_size(), self._seqs.data_ptr(), lib
        except Exception as e:  # noqa: BLE001
            local_err = e
        self._agree(local_err, "mapping the peers' buffers")

    def _agree(self, err, what: str) -> None:
        """every rank learns whether `what` worked everywhere; raises on ALL ranks alike if it did not"""
        t = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        if not bool(t.item()):
            raise RuntimeError(f"copy-path gather: {what} failed on some rank" + (f" (here: {type(err).__name__}: {err})" if err is not None else ""))

    # ---- producer -------------------------------------------------------------------------------------------------------------
    def push(self, block: torch.Tensor, timeout_s: float = 30.0) -> int:
        """Send `block` (<= numel elements, contiguous, on this rank's device) to every rank; -> its sequence number."""
        n = block.numel()
        if n > self.numel:
            raise ValueError(f"block of {n} elements, buffers hold {self.numel}")
        if not block.is_contiguous() or block.dtype != self.recv.dtype:
            raise ValueError("push: contiguous block of the buffers' dtype expected")
        k = self.pushes
        slot = k % self.NSLOT
        if self.flow_control and k + 1 > self.NSLOT:
            self._wait_acks(k + 1 - self.NSLOT, timeout_s)   # the gather whose slot this push overwrites has been released everywhere
        if k - self._seq_base >= self.SEQ_TABLE:  # (one small kernel every 65536 pushes)
            for st in self.streams:
                st.synchronize()
            self._seq_base += self.SEQ_TABLE
            self._seqs.add_(self.SEQ_TABLE)
            torch.cuda.current_stream(self.device).synchronize()
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)   # the block's producer
        if self._lib is not None:
            rc = self._lib.mir_p2p_push_streams(self._dst[slot], self.world, block.data_ptr(), n * self._esz, self._fdst[slot],
                                                self._seq_ptr + 4 * (k - self._seq_base + self.SEQ_KEEP), self._sptr)
            if rc != 0:
                raise RuntimeError(f"mir_p2p_push_streams: {self._lib.mir_last_error().decode()}")
        else:
            word = self._seqs[k - self._seq_base + self.SEQ_KEEP:k - self._seq_base + self.SEQ_KEEP + 1]
            for p, st in enumerate(self.streams):
                with torch.cuda.stream(st):
                    self.peer_recv[p][slot, self.rank, :n].copy_(block.reshape(-1), non_blocking=True)
                    self.peer_flag[p][slot, self.rank:self.rank + 1].copy_(word, non_blocking=True)
        evs = []
        for st in self.streams:
            block.record_stream(st)
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        seq = k + 1
        self._events[seq] = evs
        self._events.pop(seq - self.EVENT_DEPTH, None)
        self.pushes = seq
        return seq

    def wait_source(self, seq: int) -> None:
        """The caller's current stream waits (on the device, the host goes on) until the copies of push `seq` have read their block:
        call it before the memory that block occupied is written again."""
        for ev in self._events.get(seq, ()):
            torch.cuda.current_stream(self.device).wait_event(ev)

    def lag(self) -> int:
        """pushes whose copies have not all finished yet, counted back from the newest (every stream completes its copies in order,
        so the walk stops at the first finished push: one or two pushes' events in the steady state)"""
        n = 0
        for seq in range(self.pushes, max(self.pushes - self.EVENT_DEPTH, 0), -1):
            evs = self._events.get(seq)
            if evs is None or all(ev.query() for ev in evs):
                break
            n += 1
        return n

    def _min_ack(self) -> int:
        with torch.cuda.stream(self._ack_stream):
            return int(self.ack.min().item())   # (synchronises the ack stream only)

    def _wait_acks(self, need: int, timeout_s: float) -> None:
        import time

        if self._acked >= need:
            return
        t0 = time.perf_counter()
        while True:
            self._acked = self._min_ack()
            if self._acked >= need:
                return
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f"copy-path gather: gather {need} was not released by every rank within {timeout_s} s (acks {self.ack.tolist()})")

    # ---- consumer -------------------------------------------------------------------------------------------------------------
    def ready(self, seq: int) -> bool:
        """Have the blocks of push number `seq` arrived from every rank?  (One tiny reduction + host read: for the consumer's side of
        the protocol and for the end of a timed region, not for the step loop.)"""
        slot = (seq - 1) % self.NSLOT
        with torch.cuda.stream(self._ack_stream):
            return bool((self.flag[slot] >= seq).all().item())

    def wait(self, seq: int, timeout_s: float = 30.0) -> None:
        import time

        t0 = time.perf_counter()
        while not self.ready(seq):
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f"copy-path gather: block {seq} did not arrive from every rank within {timeout_s} s (flags {self.flag.tolist()})")

    def gathered(self, seq: int, n: int) -> torch.Tensor:
        """(world, n) view of the blocks of push `seq` (after wait())."""
        return self.recv[(seq - 1) % self.NSLOT, :, :n]

    def release(self, seq: int) -> None:
        """This rank is done with gather `seq` (and all earlier ones): its word goes to every rank, behind whatever the caller's
        current stream has queued (the consumer's reads)."""
        if seq <= self._released:
            return
        if seq <= self._seq_base - self.SEQ_KEEP:   # (older than the table reaches back: only a consumer more than SEQ_KEEP gathers behind)
            raise ValueError(f"release: gather {seq} is more than {self.SEQ_KEEP} behind the sequence table's base {self._seq_base}")
        if seq - self._seq_base > self.SEQ_TABLE:
            raise ValueError("release: sequence number ahead of the pushes")
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)
        if self._lib is not None:
            rc = self._lib.mir_p2p_push_streams(self._adst, self.world, self._seq_ptr, 0, self._adst, self._seq_ptr + 4 * (seq - 1 - self._seq_base + self.SEQ_KEEP), self._sptr)
            if rc != 0:
                raise RuntimeError(f"mir_p2p_push_streams: {self._lib.mir_last_error().decode()}")
        else:
            word = self._seqs[seq - 1 - self._seq_base + self.SEQ_KEEP:seq - self._seq_base + self.SEQ_KEEP]
            for p, st in enumerate(self.streams):
                with torch.cuda.stream(st):
                    self.peer_ack[p][self.rank:self.rank + 1].copy_(word, non_blocking=True)
        self._released = seq

    # ---- checks ---------------------------------------------------------------------------------------------------------------
    def check_against_collective(self, seq: int, block: torch.Tensor) -> bool:
        """Gather `seq` (this rank pushed `block`) against all_gather_into_tensor of the same blocks; every rank calls it with its
        own block of the same length, after wait(seq).  The verdict is the same on every rank."""
        n = block.numel()
        want = torch.empty((self.world, n), dtype=block.dtype, device=self.device)
        dist.all_gather_into_tensor(want.reshape(-1), block.reshape(-1).contiguous(), group=self.group)
        ok = int(torch.equal(self.gathered(seq, n), want.to(self.recv.dtype)))
        t = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def verify(self) -> bool:
        """One gather of a rank-specific pattern against all_gather_into_tensor; the verdict is the same on every rank."""
        ok = 1
        try:
            n = min(self.numel, 4096)
            block = (torch.arange(n, dtype=torch.float32, device=self.device) + 1000.0 * (self.rank + 1)).to(self.recv.dtype)
            seq = self.push(block)
            self.wait(seq, timeout_s=20.0)
            ok = -1   # (from here on every rank is inside the collectives of the check)
        except Exception:  # noqa: BLE001
            ok = 0
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        if not bool(t.item()):
            return False
        good = self.check_against_collective(seq, block)
        self.release(seq)
        return good


def make_copy_gather(numel: int, device: torch.device, group=None, flow_control: bool = False):
    """-> (CopyPathGather, "") when the copy path works on EVERY rank, else (None, reason): construction (buffer export, IPC handle
    exchange, peer mappings) agrees across the ranks after each fallible stage -- before any collective a failed rank would
    skip -- and so does verify(): either all ranks use the copy path or none does."""
    try:
        cg = CopyPathGather(numel, device, group, flow_control=flow_control)
    except RuntimeError as e:
        return None, str(e)
    if not cg.verify():
        return None, "verification against all_gather_into_tensor failed on some rank"
    return cg, ""
