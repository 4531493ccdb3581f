#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame front-end (ORB extract + ArUco detect + Hamming match) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: a synthetic 640x480 mono stream (BASELINE.json configs[1]:
nFeatures 1000, 8 levels, ARUCO dictionary) resident in HBM before the timed region.  Each rank owns an
independent stream (frames/streams shard with no data-path collective, SURVEY 8e: weak scaling); the only collective
is the final RCCL gather of the fixed-capacity result records to rank 0, inside the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--splits", type=int, default=1, help="process a batch as this many sub-batches on separate handles and "
                    "streams (their latency-bound and VALU-bound kernels overlap)")
    ap.add_argument("--frames", type=int, default=300, help="frames per step per GPU (the C2 stream length)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--dictionary", default="ARUCO")
    ap.add_argument("--marker-capacity", type=int, default=64, help="marker (+ pose) records per frame in the gathered result set")
    ap.add_argument("--cpu-frames", type=int, default=300, help="frames timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--no-aruco", action="store_true", help="diagnostic only: drop the ArUco leg (invalidates value)")
    ap.add_argument("--no-orb", action="store_true", help="diagnostic only: drop the ORB + matching legs (invalidates value)")
    return ap.parse_args()


def make_stream(args, rank):
    """Synthetic stream of this rank (seed base differs per rank), cached under /tmp across runs on one box."""
    from orb_slam2_aruco_amd import synth
    seed = 1000 + 2000 * rank
    path = "/tmp/orbfe_stream_%dx%d_%d_%d_%s.npy" % (args.cols, args.rows, args.frames, seed, args.dictionary)
    if os.path.exists(path):
        try:
            return np.load(path)
        except Exception:
            pass
    s = synth.stream(args.rows, args.cols, args.frames, seed, args.dictionary, n_markers=4)
    try:
        np.save(path + ".tmp.npy", s)
        os.replace(path + ".tmp.npy", path)
    except Exception:
        pass
    return s


TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]
TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
MARKER_SIZE = 0.187   # Frame.cc:131


def cpu_baseline(args, frames_u8):
    """The oracle (CPU port of the reference path) on the same stream: single thread -- the reference runs extractor and
    detector serially on the Tracking thread (Frame.cc:91,142) -- after 10 warm-up frames; and, as a second row, all host
    cores with the frames sharded into contiguous blocks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    n = min(args.cpu_frames, len(frames_u8))
    if n < 2:
        return None
    K = O.camera_resize(np.array(TUM1_K, np.float32), (1280, 720), (args.cols, args.rows))
    D = np.array(TUM1_DIST, np.float32)
    use_aruco = hasattr(O, "ArucoOracle") and not args.no_aruco

    def run(frames):                       # ctypes releases the GIL inside the oracle calls
        orb = O.OrbOracle(args.nfeatures, 1.2, args.nlevels, 20, 7)
        aruco = O.ArucoOracle(args.dictionary) if use_aruco else None
        prev = None
        for img in frames:
            k, d = orb.extract(img)
            if aruco is not None:
                for m in aruco.detect(img):
                    O.marker_pose(m["corners"], MARKER_SIZE, K, D)
            if prev is not None:
                O.knn2(prev[1], d, 256)
                O.search_for_initialization(prev[0], prev[1], k, d, args.cols, args.rows, None, 100, 0.9, True)
            prev = (k, d)

    run(frames_u8[:min(10, n)])            # warm-up
    t0 = time.perf_counter()
    run(frames_u8[:n])
    dt = time.perf_counter() - t0
    what = "ORB%s + knn2 + SearchForInitialization" % (" + ArUco incl. marker poses" if use_aruco else "")
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d stream after 10 warm-up frames, oracle/ single thread (%s)"
                     % (n, args.cols, args.rows, what)}
    cores = min(os.cpu_count() or 1, n // 4)      # blocks of >= 4 frames; `cores` = the threads actually used
    if cores > 1:
        blocks = [frames_u8[n * c // cores:n * (c + 1) // cores] for c in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(run, blocks))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": n / dt, "unit": "frames/s", "cores": cores,
                            "sample": "the same %d frames in %d contiguous blocks, one thread each" % (n, cores)}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (single-GPU box): ORBFE_BENCH_DEVICE pins every rank to one device, ORBFE_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("ORBFE_BENCH_DEVICE"):
        local_rank = int(os.environ["ORBFE_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if os.environ.get("ORBFE_BENCH_BACKEND"):
            dist.init_process_group(os.environ["ORBFE_BENCH_BACKEND"])
        else:
            dist.init_process_group("nccl", device_id=dev)

    from orb_slam2_aruco_amd import binding
    L = binding.load()
    B, rows, cols = args.frames, args.rows, args.cols

    frames_np = make_stream(args, rank)
    pitch = (cols + 63) // 64 * 64
    d_imgs = torch.zeros((B, rows, pitch), dtype=torch.uint8, device=dev)
    d_imgs[:, :, :cols] = torch.from_numpy(frames_np).to(dev)

    S = max(1, min(args.splits, B // 2))
    bounds = [B * k // S for k in range(S + 1)]          # sub-batch k = frames bounds[k] .. bounds[k+1]
    exs = [binding.ORBextractor(args.nfeatures, 1.2, args.nlevels, 20, 7, device=local_rank) for _ in range(S)]
    ex = exs[0]
    cap = ex.capacity
    # Two sets of result records: the matching (third stream) and, on N > 1, the gather of batch i (communication stream)
    # overlap with batch i+1, which writes the other set.  A set is ONE contiguous buffer -- the record SURVEY 8e gathers:
    # {n_kp, kp[cap] x 28 B, desc[cap] x 32 B, n_mk, markers[mcap] x 36 B, poses[mcap] x 56 B} per frame -- so a batch is one collective.
    use_aruco = not args.no_aruco
    big_frames = False
    # marker records per frame in the result set (the detector clamps a frame's count to it; its own limit is 256 candidates)
    mcap = min(binding.MarkerDetector(args.dictionary, device=local_rank).capacity, args.marker_capacity) if use_aruco else 0
    up = lambda v: (v + 255) // 256 * 256
    off_kps, off_desc = 0, up(B * cap * 28)
    off_n = off_desc + up(B * cap * 32)
    off_mk = off_n + up(B * 4)
    off_nmk = off_mk + up(B * mcap * 36)
    off_pose = off_nmk + up(B * 4)
    rec_bytes = off_pose + up(B * mcap * 56)
    # camera of the reference's monocular example (Examples/Monocular/TUM1.yaml); the detector is handed CamSize 1280x720
    # (Frame.cc:132), so the matrix is rescaled to the frame size before the marker poses (markerdetector_impl.cpp:1110-1172)
    cam_K = binding.camera_resize(np.array(TUM1_K, np.float32), (1280, 720), (cols, rows)) if use_aruco else None
    cam_D = np.array(TUM1_DIST, np.float32)
    recs = [torch.zeros(rec_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    rec_ptr = [r.data_ptr() for r in recs]
    d_n = recs[0][off_n:off_n + B * 4].view(torch.int32)
    d_bidx = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_bdist = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_sdist = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_m12 = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B - 1, dtype=torch.int32, device=dev)
    if use_aruco:
        dets = [binding.MarkerDetector(args.dictionary, device=local_rank) for _ in range(S)]
        det = dets[0]
    # three HIP streams: the ORB extractor, the ArUco detector, the matching.  The first two only read the resident
    # frames; the matching of batch i reads extractor output set i % 2 while batch i+1 is extracted into the other set.
    # So the latency-bound kernels (contours, quadtree, SearchForInitialization) overlap with the VALU-bound ones.
    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    stream2 = torch.cuda.Stream(dev)
    sp2 = ctypes.c_void_p(stream2.cuda_stream)
    stream3 = torch.cuda.Stream(dev)
    sp3 = ctypes.c_void_p(stream3.cuda_stream)
    # with --splits S > 1: sub-batch k of the extractor / detector runs on its own handle and stream
    orb_streams = [stream] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    aru_streams = [stream2] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    if S == 1:
        # ROCm maps streams onto 4 hardware queues, and two busy streams on one queue serialise.  The extractor's forked
        # launch (the blur) is lent the matching stream; measured against the handle's own fork stream and against one
        # shared fork stream for both engines: 2.02 vs 2.12 vs 2.14 ms per step.
        ex.set_aux_stream(sp3)
    ex_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(2)]
    match_done = [torch.cuda.Event() for _ in range(2)]
    comm_stream = torch.cuda.Stream(dev)
    det_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(2)]
    gather_done = [torch.cuda.Event() for _ in range(2)]
    step_no = [0]

    gathered = None
    if world > 1:
        gathered = [torch.empty_like(recs[0]) for _ in range(world)] if rank == 0 else None

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]  # around the matching launches (their stream)

    def step():
        i = step_no[0]
        step_no[0] += 1
        base = rec_ptr[i % 2]
        if use_aruco:
            # the detector streams only depend on the (resident) input frames and on their own previous batch, so they are
            # not joined with the ORB streams per step: consecutive batches of the two engines pipeline freely.  They are
            # joined where their results meet: before the RCCL gather (N > 1) and before the clock stops.
            for k in range(S):
                f0, nf = bounds[k], bounds[k + 1] - bounds[k]
                if world > 1 and i >= 2:
                    aru_streams[k].wait_event(gather_done[i % 2])   # batch i-2 has left this record set
                dets[k].detect_batch_device(d_imgs.data_ptr() + f0 * rows * pitch, nf, rows * pitch, rows, cols, pitch,
                                            base + off_mk + f0 * mcap * 36, mcap, base + off_nmk + f0 * 4,
                                            ctypes.c_void_p(aru_streams[k].cuda_stream))
                # detect(image, CameraParameters, 0.187): every marker gets its IPPE pose (markerdetector_impl.cpp:8720-8780)
                binding._check(L, L.orbfe_marker_poses_batch_device(
                    base + off_mk + f0 * mcap * 36, base + off_nmk + f0 * 4, mcap, nf, MARKER_SIZE,
                    cam_K.ctypes.data_as(ctypes.c_void_p), cam_D.ctypes.data_as(ctypes.c_void_p), len(cam_D),
                    base + off_pose + f0 * mcap * 56, ctypes.c_void_p(aru_streams[k].cuda_stream)), "orbfe_marker_poses_batch_device")
                det_done[i % 2][k].record(aru_streams[k])
        if not args.no_orb:
            for k in range(S):
                f0, nf = bounds[k], bounds[k + 1] - bounds[k]
                if i >= 2:
                    orb_streams[k].wait_event(match_done[i % 2])    # the matching of batch i-2 has read this record set
                    if world > 1:
                        orb_streams[k].wait_event(gather_done[i % 2])
                exs[k].extract_batch_device(d_imgs.data_ptr() + f0 * rows * pitch, nf, rows * pitch, rows, cols, pitch,
                                            base + off_kps + f0 * cap * 28, base + off_desc + f0 * cap * 32, cap,
                                            base + off_n + f0 * 4, ctypes.c_void_p(orb_streams[k].cuda_stream))
                ex_done[i % 2][k].record(orb_streams[k])
                stream3.wait_event(ex_done[i % 2][k])
            # frame t vs t-1: all-pairs knn2 + one SearchForInitialization-style windowed pass
            ev[0].record(stream3)
            binding._check(L, L.orbfe_knn2_batch_device(base + off_desc, base + off_n, cap * 32, cap,
                                                        base + off_desc + cap * 32, base + off_n + 4, cap * 32, cap,
                                                        B - 1, 256, d_bidx.data_ptr(), d_bdist.data_ptr(),
                                                        d_sdist.data_ptr(), sp3), "knn2")
            ev[1].record(stream3)
            binding._check(L, L.orbfe_search_for_initialization_batch_device(
                base + off_kps, base + off_desc, base + off_n, cap, B - 1, cols, rows, None, 100, 0.9, 1,
                d_m12.data_ptr(), d_nm.data_ptr(), sp3), "sfi")
            ev[2].record(stream3)
            match_done[i % 2].record(stream3)
        if world > 1:
            # the batch's one collective, on its own stream: it waits for the engines of THIS batch and runs
            # while the next batch is computed into the other record set
            with torch.cuda.stream(comm_stream):
                for k in range(S):
                    if not args.no_orb:
                        comm_stream.wait_event(ex_done[i % 2][k])
                    if use_aruco:
                        comm_stream.wait_event(det_done[i % 2][k])
                dist.gather(recs[i % 2], gathered, dst=0)
                gather_done[i % 2].record(comm_stream)

    if os.environ.get("ORBFE_ORB_SKIP"):    # diagnosis: what does a kernel cost the concurrent pipeline (results invalid)
        binding.debug_control("orb_skip", int(os.environ["ORBFE_ORB_SKIP"]))
    if os.environ.get("ORBFE_ARUCO_SKIP"):
        binding.debug_control("aruco_skip", int(os.environ["ORBFE_ARUCO_SKIP"]))
    ex.enable_kernel_timing(False)
    if args.no_orb:  # diagnostics still want the level geometry
        ex.extract_batch_device(d_imgs.data_ptr(), B, rows * pitch, rows, cols, pitch, rec_ptr[0] + off_kps,
                                rec_ptr[0] + off_desc, cap, rec_ptr[0] + off_n, sp)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_aruco:
        # the device-pointer entry point cannot return a capacity error: ask once after the warm-up and once after the run.
        # Frames with more long contours than the LDS-resident kernels hold (large, busy images) need the big-frame kernel.
        if any(d.batch_status()[0] for d in dets):
            for d in dets:
                d.set_big_frames(True)
            big_frames = True
            for _ in range(max(args.warmup, 1)):
                step()
            torch.cuda.synchronize()
            if any(d.batch_status()[0] for d in dets):
                raise SystemExit("ArUco detector capacity exceeded at this frame size (flags 0x%x)" % dets[0].batch_status()[1])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ex.enable_kernel_timing(True)
    if use_aruco:
        det.enable_kernel_timing(True)
    ktimes = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kt = {"orb": ex.kernel_times_us()} if False else None  # per-step collection would sync; done after the loop
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_aruco and any(d.batch_status()[0] for d in dets):
        raise SystemExit("ArUco detector capacity exceeded during the timed run: results incomplete, no number reported")
    # HIP-event timings of the LAST timed step's launches (events were recorded on the launch stream every step)
    orb_us = ex.kernel_times_us()
    aruco_us = det.kernel_times_us() if use_aruco else np.zeros(0, np.float32)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_frames = B * args.steps * world
    n_host = d_n.cpu().numpy()

    if rank == 0:
        orb_names = binding.ORBextractor.STAGES  # blur7 runs on a second stream next to fast_cells + distribute
        stages = {nm: float(v) for nm, v in zip(orb_names, orb_us)}
        if not args.no_orb:
            stages["knn2"] = ev[0].elapsed_time(ev[1]) * 1000.0
            stages["search_init"] = ev[1].elapsed_time(ev[2]) * 1000.0
        if use_aruco:
            for nm, v in zip(binding.MarkerDetector.STAGES, aruco_us):
                stages["aruco_" + nm] = float(v)
        # algorithmic bytes per frame of each stage (DESIGN.md "roofline": terms of SURVEY 8d's B_orb / B_aruco)
        sizes = ex.level_sizes()
        P = [w * h for (w, h) in sizes]
        sumP, P0 = sum(P), P[0]
        N = float(n_host.mean())
        alg = {"resize": (sumP - P[-1]) + (sumP - P0), "fast_cells": sumP, "blur7": 2 * sumP,
               "orient_describe": N * (749 + 512 + 60), "distribute": 0,
               "knn2": 2 * N * 32 + N * 12, "search_init": 2 * N * (32 + 28) + N * 4}
        if use_aruco:
            alg.update(binding.MarkerDetector.algorithmic_bytes(rows, cols))
        # frames one launch of a stage covers: the matching runs over the whole batch, the engines per sub-batch (--splits)
        fl = lambda k: B if k in ("knn2", "search_init") else bounds[1] - bounds[0]
        dom = max(stages, key=lambda k: stages[k]) if stages else None
        roof = None
        if dom is not None and stages[dom] > 0:
            ach = alg.get(dom, 0) * fl(dom) / (stages[dom] * 1e-6) / 1e9
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    traffic = json.load(open(tp)).get(dom)
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                    "frac": ach / 8000.0, "traffic": traffic, "launch_us": stages[dom],
                    "algorithmic_bytes_per_launch": alg.get(dom, 0) * fl(dom), "frames_per_launch": fl(dom),
                    "note": "dominant launch of the last timed step (HIP events on its launch stream, the other engine "
                            "running concurrently); k_contours is serial border following, latency- not HBM-bound",
                    # the same figure for every stage, so the HBM-bound image kernels can be read off too
                    "all_stages_GBps": {k: (alg.get(k, 0) * fl(k) / (v * 1e-6) / 1e9 if v > 0 else 0.0)
                                        for k, v in stages.items()}}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu = cpu_baseline(args, frames_np)
        out = {
            "metric": "frames/s (ORB+ArUco extract+match, 640x480 mono)" if (rows, cols) == (480, 640) else
                      "frames/s (ORB+ArUco extract+match, %dx%d mono)" % (cols, rows),
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d-frame %dx%d mono stream per GPU, nFeatures=%d, %d levels, %s dictionary; "
                                   "per frame: ORB extract%s + knn2 all-pairs + SearchForInitialization vs previous frame"
                                   % ({(480, 640, 300, 1000): "C2", (720, 1280, 300, 2000): "C3", (1080, 1920, 100, 4000): "C5 frames"}
                                      .get((rows, cols, B, args.nfeatures), "custom"), B, cols, rows, args.nfeatures, args.nlevels, args.dictionary,
                                      " + ArUco detect incl. IPPE marker poses" if use_aruco else " (ArUco leg DISABLED: diagnostic run)"),
                       "frames_per_step_per_gpu": B, "mean_keypoints_per_frame": N, "marker_records_per_frame": mcap,
                       "result_record_bytes_per_step_per_gpu": rec_bytes,
                       "sub_batches": S, "aruco_big_frame_kernel": big_frames,
                       "parallelism": "stream-per-gpu x%d, RCCL gather to rank 0" % world},
            "roofline": roof, "cpu_baseline": cpu, "stage_us_last_step": stages,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
