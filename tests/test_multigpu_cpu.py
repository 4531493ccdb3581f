"""CPU test of the multi-rank path: streams shard across ranks with no data-path collective; the only
collective is the final gather of fixed-capacity result records to rank 0.  Runs with gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orb_slam2_aruco_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cap = 8
    # each rank owns its own stream; fake per-frame results derived from (rank, frame)
    frames = sharding.frames_of_rank(10, rank, world)
    n = torch.tensor([(3 + rank + f) % cap for f in frames], dtype=torch.int32)
    kps = torch.zeros((len(frames), cap, 7), dtype=torch.float32)
    desc = torch.zeros((len(frames), cap, 32), dtype=torch.uint8)
    for i, f in enumerate(frames):
        kps[i, :, 0] = rank * 1000 + f
        desc[i, :, :] = (rank * 16 + f) % 256
    out = sharding.gather_records([n, kps, desc], dst=0)
    if rank == 0:
        q.put([[t.numpy() for t in per_rank] for per_rank in out])
    dist.barrier()
    dist.destroy_process_group()


def test_frames_of_rank_partitions_everything():
    for total in (1, 7, 10, 300):
        for world in (1, 2, 3, 8):
            parts = [sharding.frames_of_rank(total, r, world) for r in range(world)]
            flat = [f for p in parts for f in p]
            assert sorted(flat) == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_gather_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(got) == 2
    for rank, (n, kps, desc) in enumerate(got):
        frames = sharding.frames_of_rank(10, rank, 2)
        assert n.tolist() == [(3 + rank + f) % 8 for f in frames]
        assert np.all(kps[:, :, 0] == (rank * 1000 + np.array(frames))[:, None])
        assert np.all(desc[:, 0, 0] == (rank * 16 + np.array(frames)) % 256)


# ---- the library's gather control flow, two ranks, no GPU ------------------------------------------------------------------------------
def _record(rank, batch, nb):
    """what rank `rank` would gather for batch `batch`: nb bytes that name both"""
    g = torch.Generator().manual_seed(1000 * rank + batch)
    return torch.randint(0, 256, (nb,), dtype=torch.uint8, generator=g)


def _plan_worker(rank, world, port, q, dst, record_sets, nbatches, nb):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = torch.zeros(record_sets * world * nb, dtype=torch.uint8) if rank == dst else None
    ok = True
    for b in range(nbatches):
        s = b % record_sets                       # the record set batch b is written to (orbfe_pipeline_step's *record_set)
        plan = sharding.gather_plan(rank, world, dst, s, record_sets, nb)
        sharding.run_gather_plan(plan, _record(rank, b, nb), blocks)
        if rank == dst:
            # the consumer's view after batch b: set s holds batch b of every rank, the other sets the newest batch written to them
            for t in range(record_sets):
                newest = max((x for x in range(b + 1) if x % record_sets == t), default=None)
                for r in range(world):
                    blk = blocks[(t * world + r) * nb:(t * world + r + 1) * nb]
                    ok = ok and (torch.equal(blk, _record(r, newest, nb)) if newest is not None else not bool(blk.any()))
    if rank == dst:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dst", [0, 1])
def test_library_gather_plan_between_two_gloo_ranks(dst):
    """The product's gather control flow without a GPU: orbfe_pipeline_gather_plan (csrc/gather_plan.hpp -- the function csrc/pipeline.hip
    executes with ncclRecv / ncclSend / a device copy) says what each rank sends and where rank dst receives it; two gloo processes execute
    it for ten batches over four rotating record sets (the blocks wrap twice): after every batch every block holds the newest batch that
    was written to its set, from the right rank."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q, dst, 4, 10, 4096)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok


def test_library_gather_plan_shapes():
    """The plan itself, every rank of 1, 2 and 8: dst receives one block per peer into the record set's block row and copies its own;
    everyone else sends once; world 1 sends to itself (the same kernels on a one-GPU box); bad arguments are refused."""
    R, nb = 4, 1000
    for world in (1, 2, 8):
        for dst in range(world):
            for s in range(R):
                recv_from = set()
                for rank in range(world):
                    plan = sharding.gather_plan(rank, world, dst, s, R, nb)
                    kinds = [k for k, _, _ in plan]
                    if rank != dst:
                        assert plan == [(sharding.GATHER_SEND, dst, 0)]
                        continue
                    for k, peer, off in plan:
                        if k == sharding.GATHER_RECV:
                            assert off == (s * world + peer) * nb and peer not in recv_from
                            recv_from.add(peer)
                    if world == 1:
                        assert kinds == [sharding.GATHER_RECV, sharding.GATHER_SEND]
                    else:
                        assert kinds.count(sharding.GATHER_COPY_OWN) == 1 and plan[-1] == (sharding.GATHER_COPY_OWN, dst, (s * world + dst) * nb)
                assert recv_from == (set(range(world)) - {dst} if world > 1 else {0})
    for bad in ((2, 2, 0, 0, 4), (0, 2, 2, 0, 4), (0, 2, 0, 4, 4), (-1, 2, 0, 0, 4)):
        with pytest.raises(ValueError):
            sharding.gather_plan(*bad, 100)
