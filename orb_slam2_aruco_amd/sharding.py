"""Stream-to-rank assignment of the multi-GPU mode, and bench.py's gloo transport.

Frames / streams are independent, so they shard across ranks with no data-path collective: `stream_seed` / `frames_of_rank` say which
rank owns what (bench.py, the tests).  The PRODUCT gather -- every batch's record set to rank 0 over RCCL -- is C++ inside the library
(csrc/pipeline.hip: orbfe_pipeline_comm_init / _gathered_set / _gathered_wait; two processes with a peer: tests/fake_rccl.cpp).
`gather_plan` / `run_gather_plan` drive that path's own control flow -- which rank receives what into which of the rotating blocks, from
orbfe_pipeline_gather_plan, the function the library executes -- over torch.distributed point-to-point operations: tests/test_multigpu_cpu.py
runs it between two gloo ranks without a GPU.  `RecordGather` / `gather_records` are NOT the product path: they are the
torch.distributed.gather bench.py uses when the records have to go through gloo from the host -- its test hook for N ranks on a one-GPU
box (ORBFE_BENCH_BACKEND=gloo) and the fallback when RCCL cannot be initialised (then the line is a diagnostic)."""
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


# ---- the library's own gather plan over any transport -------------------------------------------------------------------------------
import ctypes as _C


class _GatherOp(_C.Structure):
    _fields_ = [("kind", _C.c_int32), ("peer", _C.c_int32), ("offset", _C.c_ulonglong)]


GATHER_RECV, GATHER_SEND, GATHER_COPY_OWN = 0, 1, 2


def gather_plan(rank, world, dst, record_set, record_sets, nbytes):
    """The operations of one batch's gather on `rank`, from the LIBRARY (orbfe_pipeline_gather_plan = csrc/gather_plan.hpp, the function
    csrc/pipeline.hip executes over RCCL): a list of (kind, peer, byte offset into dst's block buffer).  No device is touched."""
    from . import binding
    L = binding.load()
    L.orbfe_pipeline_gather_plan.argtypes = [_C.c_int, _C.c_int, _C.c_int, _C.c_int, _C.c_int, _C.c_size_t, _C.POINTER(_GatherOp), _C.c_int]
    ops = (_GatherOp * (world + 2))()
    n = L.orbfe_pipeline_gather_plan(rank, world, dst, record_set, record_sets, nbytes, ops, world + 2)
    if n < 0:
        raise ValueError("orbfe_pipeline_gather_plan: %s" % L.orbfe_last_error().decode())
    return [(ops[i].kind, ops[i].peer, ops[i].offset) for i in range(n)]


def run_gather_plan(plan, record, blocks):
    """Execute a plan with torch.distributed point-to-point operations (any backend): `record` = this rank's record set (1-D uint8),
    `blocks` = dst's block buffer (1-D uint8, record_sets x world blocks) or None.  The messages of a batch form one group, as in the
    library (ncclGroupStart .. ncclGroupEnd): all are posted before any is waited for."""
    nb = record.numel()
    reqs = []
    for kind, peer, off in plan:
        if kind == GATHER_RECV:
            reqs.append(dist.irecv(blocks[off:off + nb], src=peer))
        elif kind == GATHER_SEND:
            reqs.append(dist.isend(record, dst=peer))
    for r in reqs:
        r.wait()
    for kind, peer, off in plan:
        if kind == GATHER_COPY_OWN:
            blocks[off:off + nb].copy_(record)
