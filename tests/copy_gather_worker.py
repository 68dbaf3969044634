"""Worker of tests/test_gpu_copy_gather.py: one rank of a torch.distributed.run job whose ranks all sit on cuda:0 (a 1-GPU box).
Pushes rank- and round-specific blocks through sharding.CopyPathGather and checks what arrives from every rank."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
from gym_genesis.sharding import make_copy_gather  # noqa: E402


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    numel = 5000
    cg, why = make_copy_gather(numel, dev)
    assert cg is not None, why
    base = cg.pushes                          # (make_copy_gather's verification was push number 1)
    assert base == 1
    for k in range(12):
        n = numel if k % 3 else 1234          # full and partial blocks, both slots
        block = torch.arange(n, dtype=torch.float32, device=dev) * (k + 1) + 1e6 * (rank + 1)
        seq = cg.push(block)
        assert seq == base + k + 1
        cg.wait(seq, timeout_s=20.0)
        got = cg.gathered(seq, n)
        for r in range(world):
            want = torch.arange(n, dtype=torch.float32, device=dev) * (k + 1) + 1e6 * (r + 1)
            assert torch.equal(got[r], want), (rank, k, r)
        dist.barrier()                        # (the consumer's side of the protocol: done with the slot before it is reused)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("COPY_GATHER_OK", world, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
