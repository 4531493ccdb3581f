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
