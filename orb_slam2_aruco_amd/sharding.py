"""Stream-to-rank assignment of the multi-GPU mode, and bench.py's gloo transport.

Frames / streams are independent, so they shard across ranks with no data-path collective: `stream_seed` / `frames_of_rank` say which
rank owns what (bench.py, the tests).  The PRODUCT gather -- every batch's record set to rank 0 over RCCL -- is C++ inside the library
(csrc/pipeline.hip: orbfe_pipeline_comm_init / _gathered_set / _gathered_wait; two processes with a peer: tests/fake_rccl.cpp).
`RecordGather` / `gather_records` are NOT that path: they are the torch.distributed.gather bench.py uses when the records have to go
through gloo from the host -- its test hook for N ranks on a one-GPU box (ORBFE_BENCH_BACKEND=gloo) and the fallback when RCCL cannot
be initialised (then the line is a diagnostic) -- and what tests/test_multigpu_cpu.py runs with gloo on the CPU."""
import torch
import torch.distributed as dist


def frames_of_rank(total, rank, world):
    """Contiguous block of frame indices owned by `rank` (blocks differ in length by at most one)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def stream_seed(rank, base=1000, stride=2000):
    """Seed of the synthetic stream rank `rank` owns (seed base 2000 * s)."""
    return base + stride * rank


class RecordGather:
    """The batch's one collective: gather one record buffer per rank to `dst`.

    The receive buffers are allocated once (rank dst only) and reused by every batch; a record set is ONE contiguous
    buffer (pipeline.RecordLayout), so a batch is one dist.gather -- RCCL send/recv over xGMI on the GPU, 7 peers -> rank 0.
    Issued on whatever stream is current (FrontEndPipeline: its communication stream)."""

    def __init__(self, like, dst=0):
        self.dst = dst
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.blocks = [torch.empty_like(like) for _ in range(self.world)] if self.rank == dst else None

    def __call__(self, tensor):
        dist.gather(tensor, self.blocks, dst=self.dst)
        return self.blocks


def gather_records(tensors, dst=0):
    """Gather each tensor of `tensors` (same shape on every rank) to rank `dst`.

    Returns, on dst, a list over ranks of lists of tensors; elsewhere None.  One gather per record array per
    batch of frames (never per frame): the records are a few hundred KB, so the gather is latency-bound."""
    out = [RecordGather(t, dst)(t) for t in tensors]
    if dist.get_rank() != dst:
        return None
    return [[out[k][r] for k in range(len(tensors))] for r in range(dist.get_world_size())]
