"""Multi-GPU plumbing: frames / streams are independent, so they shard across ranks with no data-path
collective; the single collective is the gather of fixed-capacity result records to rank 0 (RCCL on GPU tensors,
gloo in the CPU tests).  torch.distributed is plumbing here, not part of the product library."""
import torch
import torch.distributed as dist


def frames_of_rank(total, rank, world):
    """Contiguous block of frame indices owned by `rank` (blocks differ in length by at most one)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def gather_records(tensors, dst=0):
    """Gather each tensor of `tensors` (same shape on every rank) to rank `dst`.

    Returns, on dst, a list over ranks of lists of tensors; elsewhere None.  One dist.gather per record array per
    batch of frames (never per frame): the records are a few hundred KB, so the gather is latency-bound."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    out = []
    for t in tensors:
        buf = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, buf, dst=dst)
        out.append(buf)
    if rank != dst:
        return None
    return [[out[k][r] for k in range(len(tensors))] for r in range(world)]
