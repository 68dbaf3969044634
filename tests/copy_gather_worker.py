"""Worker of tests/test_gpu_copy_gather.py: one rank of a torch.distributed.run job whose ranks all sit on cuda:0 (a 1-GPU box).
Pushes rank- and round-specific blocks through sharding.CopyPathGather and checks what arrives from every rank."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
from gym_genesis.sharding import CopyPathGather, make_copy_gather  # noqa: E402


def pattern(n, k, r, dev):
    return torch.arange(n, dtype=torch.float32, device=dev) * (k + 1) + 1e6 * (r + 1)


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    numel = 5000
    # ---- (A) lock-step use: push, wait, check, release, barrier
    cg, why = make_copy_gather(numel, dev)
    assert cg is not None, why
    base = cg.pushes                          # (make_copy_gather's verification was push number 1)
    assert base == 1
    for k in range(12):
        n = numel if k % 3 else 1234          # full and partial blocks, both slots
        block = pattern(n, k, rank, dev)
        seq = cg.push(block)
        assert seq == base + k + 1
        cg.wait(seq, timeout_s=20.0)
        got = cg.gathered(seq, n)
        for r in range(world):
            assert torch.equal(got[r], pattern(n, k, r, dev)), (rank, k, r)
        assert cg.check_against_collective(seq, block)
        cg.release(seq)
        dist.barrier()                        # (no flow control here: the ranks stay in step by themselves)
    torch.cuda.synchronize()
    assert cg.lag() == 0
    # ---- (B) flow control, no barriers: every rank produces as fast as it can and consumes at its own pace (rank r dawdles r x 3 ms
    # per gather); a producer may run NSLOT gathers ahead of the slowest consumer and no further, and nothing is ever torn
    fc, why = make_copy_gather(numel, dev, flow_control=True)
    assert fc is not None, why
    K, consumed = 40, fc.pushes
    ring = [torch.empty(numel, dtype=torch.float32, device=dev) for _ in range(4)]   # the producer's own ring of source blocks
    ahead_max = 0
    for k in range(K):
        src = ring[k % 4]
        if k >= 4:
            fc.wait_source(fc.pushes - 3)     # the push that last read this ring entry
        src.copy_(pattern(numel, k, rank, dev))
        seq = fc.push(src)
        ahead_max = max(ahead_max, seq - consumed)
        # consume what has arrived, oldest first, without waiting for more than the flow control forces
        while consumed < seq and (fc.ready(consumed + 1) or seq - consumed >= fc.NSLOT):
            c = consumed + 1
            fc.wait(c, timeout_s=20.0)
            got = fc.gathered(c, numel).clone()
            kk = c - 2                          # (push 1 was the verification, push c carries pattern number c - 2)
            for r in range(world):
                assert torch.equal(got[r], pattern(numel, kk, r, dev)), (rank, c, r)
            time.sleep(0.003 * rank)
            fc.release(c)
            consumed = c
    while consumed < fc.pushes:
        c = consumed + 1
        fc.wait(c, timeout_s=20.0)
        got = fc.gathered(c, numel).clone()
        for r in range(world):
            assert torch.equal(got[r], pattern(numel, c - 2, r, dev)), (rank, c, r)
        fc.release(c)
        consumed = c
    assert ahead_max <= fc.NSLOT, ahead_max
    torch.cuda.synchronize()
    dist.barrier()
    # ---- (C) the sequence table rebased under flow control, in the overlapped pattern NSLOT = 2 exists for: push k + 1, THEN consume
    # and release k.  With the table shrunk to 8 words the rebase comes every 8 pushes, right between a push and the release of the
    # gather before it (that release was dropped, the ack stayed behind and the next push timed out after 30 s)
    CopyPathGather.SEQ_TABLE, CopyPathGather.SEQ_KEEP = 8, 4
    try:
        rb, why = make_copy_gather(numel, dev, flow_control=True)
        assert rb is not None, why
        prev = None
        for k in range(40):
            seq = rb.push(pattern(numel, k, rank, dev), timeout_s=10.0)
            if prev is not None:
                rb.wait(prev, timeout_s=10.0)
                got = rb.gathered(prev, numel).clone()
                for r in range(world):
                    assert torch.equal(got[r], pattern(numel, prev - 2, r, dev)), (rank, prev, r)
                rb.release(prev)
            prev = seq
        rb.wait(prev, timeout_s=10.0)
        rb.release(prev)
        assert rb._seq_base >= 32, rb._seq_base   # (the table was rebased at least four times)
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        CopyPathGather.SEQ_TABLE, CopyPathGather.SEQ_KEEP = 1 << 16, 64
    dist.destroy_process_group()
    if rank == 0:
        print("COPY_GATHER_OK", world, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
