#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame front-end (ORB extract + ArUco detect + Hamming match) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
    (N > 1: either as above -- bench.py then starts the N ranks itself through torch.distributed.run -- or under a launcher:
     python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; WORLD_SIZE must equal N)

One step = one pass of the hot path (orb_slam2_aruco_amd/pipeline.py: FrontEndPipeline.step) over one batch: a synthetic
mono stream resident in HBM before the timed region.  Default = BASELINE.json configs[1] (C2: 300 frames 640x480,
nFeatures 1000, 8 levels, ARUCO dictionary).  Each rank owns an independent stream (frames / streams shard with no
data-path collective, SURVEY 8e: weak scaling); the only collective is the RCCL gather of the fixed-capacity result records
to rank 0, once per batch, inside the timed region.  After the clock stops rank 0 checks sampled frames of the last timed
step against the CPU oracle ("verified_frames").  Prints ONE JSON line on rank 0.

    python bench.py --latency        drop-in latency of one frame through the host-pointer ABI (one JSON line, separate metric)
"""
import argparse
import ctypes
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY 8d configurations: frames per step, rows, cols, nFeatures, levels, dictionary
CONFIGS = {
    "C2": dict(frames=300, rows=480, cols=640, nfeatures=1000, nlevels=8, dictionary="ARUCO", n_markers=4),
    "C3": dict(frames=300, rows=720, cols=1280, nfeatures=2000, nlevels=8, dictionary="ARUCO_MIP_25h7", n_markers=6),
    # C4 = one C3-like stream per GPU (seed base 2000 * rank) + the RCCL gather: run with --gpus 8
    "C4": dict(frames=300, rows=720, cols=1280, nfeatures=2000, nlevels=8, dictionary="ARUCO_MIP_25h7", n_markers=6),
    # C5 = 100 frames 1920x1080 + the 10k x 10k all-pairs knn2 ("c5_match" in the JSON line)
    "C5": dict(frames=100, rows=1080, cols=1920, nfeatures=4000, nlevels=12, dictionary="ARUCO", n_markers=6),
}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# integer VALU issue: 4 cycles per wave64 instruction per SIMD (what tools/pmc.py's VALUus assumes and r01's FAST launch
# confirmed: 526 us of issue in a 529 us launch) = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 3.9e13 lane-ops/s
VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9
I8_MFMA_PEAK_TOPS = 5000.0      # dense i8 = 2x the bf16 rate (guide: >= 3944 TOPS measured with 16x16x64)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="SURVEY 8d configuration (default: the one "
                    "BASELINE.json's metric is quoted on)")
    ap.add_argument("--frames", type=int, default=None, help="frames per step per GPU (overrides the configuration)")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--nfeatures", type=int, default=None)
    ap.add_argument("--nlevels", type=int, default=None)
    ap.add_argument("--dictionary", default=None)
    ap.add_argument("--marker-capacity", type=int, default=64, help="marker (+ pose) records per frame in the gathered result set")
    ap.add_argument("--resident-batches", type=int, default=0, help="copies of the stream resident in HBM (different time "
                    "offsets), one per step in rotation; 0 = as many as exceed the 256 MiB Infinity Cache (at most 8)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run check against the oracle")
    ap.add_argument("--no-extras", action="store_true", help="default configuration only: skip the records appended after the clock "
                    "(single-frame latency, short C3 / C5 legs)")
    ap.add_argument("--no-aruco", action="store_true", help="diagnostic only: drop the ArUco leg (value becomes null)")
    ap.add_argument("--no-orb", action="store_true", help="diagnostic only: drop the ORB + matching legs (value becomes null)")
    ap.add_argument("--force-gather", action="store_true", help="with --gpus 1: still initialise the RCCL process group (world size 1) "
                    "and run the batch's device-tensor gather on the communication stream, so the N > 1 branch executes on a one-GPU box")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, count them over gloo and print n_gpus (no GPU "
                    "needed: the CPU test of the --gpus N launcher)")
    ap.add_argument("--latency", action="store_true", help="single-frame latency through the host-pointer ABI instead")
    ap.add_argument("--from-host", action="store_true", help="frames start in page-locked HOST memory and the record sets end there: upload, "
                    "engines and read-back overlapped by the pipeline (orbfe_pipeline_step_host); a separate metric, never the headline value")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    for k, v in cfg.items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    args.custom = any(getattr(args, k) != v for k, v in cfg.items() if k != "frames")
    args.reduced = args.frames != cfg["frames"]        # the named configuration at another batch length (tests, quick runs)
    if args.cpu_frames is None:     # ~10-30 s of single-thread oracle work
        args.cpu_frames = {"C2": 300, "C3": 100, "C4": 100, "C5": 40}[args.config]
    return args


def make_stream(args, rank):
    """Synthetic stream of this rank (seed base differs per rank), cached under /tmp across runs on one box."""
    from orb_slam2_aruco_amd import synth, sharding
    seed = sharding.stream_seed(rank)
    path = "/tmp/orbfe_stream_%dx%d_%d_%d_%s.npy" % (args.cols, args.rows, args.frames, seed, args.dictionary)
    if os.path.exists(path):
        try:
            return np.load(path)
        except Exception:
            pass
    # (rendered by a pool of processes: 300 frames of 1280 x 720 take a minute on one core -- but in this process when a profiler is
    # attached: the pool's forkserver children inherit rocprofv3's preloaded tool and never come back, which is how a profile run that
    # did not find the stream cached hung for as long as it was given)
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    workers = 1 if profiled else max(1, min(16, (os.cpu_count() or 1) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    s = synth.stream(args.rows, args.cols, args.frames, seed, args.dictionary, n_markers=args.n_markers, workers=workers)
    try:
        np.save(path + ".tmp.npy", s)
        os.replace(path + ".tmp.npy", path)
    except Exception:
        pass
    return s


def oracle_module():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    return oracle_lib


def cpu_baseline(args, frames_u8):
    """The oracle (CPU port of the reference path) on the same stream: single thread -- the reference runs extractor and
    detector serially on the Tracking thread (Frame.cc:91,142) -- after 10 warm-up frames; and, as a second row, all host
    cores with the frames sharded into contiguous blocks."""
    O = oracle_module()
    from orb_slam2_aruco_amd.pipeline import TUM1_K, TUM1_DIST, MARKER_SIZE
    from concurrent.futures import ThreadPoolExecutor
    n = min(args.cpu_frames, len(frames_u8))
    if n < 2:
        return None
    K = O.camera_resize(np.array(TUM1_K, np.float32), (1280, 720), (args.cols, args.rows))
    D = np.array(TUM1_DIST, np.float32)
    use_aruco = not args.no_aruco
    use_orb = not args.no_orb

    def run(frames):                       # ctypes releases the GIL inside the oracle calls
        orb = O.OrbOracle(args.nfeatures, 1.2, args.nlevels, 20, 7)
        aruco = O.ArucoOracle(args.dictionary) if use_aruco else None
        prev = None
        for img in frames:
            if aruco is not None:
                for m in aruco.detect(img):
                    O.marker_pose(m["corners"], MARKER_SIZE, K, D)
            if not use_orb:
                continue
            k, d = orb.extract(img)
            if prev is not None:
                O.knn2(prev[1], d, 256)
                O.search_for_initialization(prev[0], prev[1], k, d, args.cols, args.rows, None, 100, 0.9, True)
            prev = (k, d)

    run(frames_u8[:min(10, n)])            # warm-up
    t0 = time.perf_counter()
    run(frames_u8[:n])
    dt = time.perf_counter() - t0
    what = "%s%s" % ("ORB + knn2 + SearchForInitialization" if use_orb else "", " + ArUco incl. marker poses" if use_aruco else "")
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d stream after 10 warm-up frames, oracle/ single thread (%s); the port is scalar "
                     "C++ at -O3 (no SIMD intrinsics): the reference's OpenCV runs SSE2 FAST / resize / blur and would be faster "
                     "by a small integer factor" % (n, args.cols, args.rows, what)}
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


def load_profile(name, config):
    """A committed profile of THIS configuration (profiles/<name>_<config>.json), or None: counters measured on another
    frame size are never reported."""
    p = os.path.join(ROOT, "profiles", "%s_%s.json" % (name, config))
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


# what bounds each stage (from the PMC counters of tools/pmc.py): "hbm" = streaming, traffic ~ algorithmic
# bytes; "valu" = the launch needs (almost) its whole duration just to issue its VALU instructions; "latency" = serial
# dependent chains (border following, quadtree rounds, the coupled accept loops) that neither bandwidth nor issue bounds
STAGE_BOUND = {"resize": "hbm", "blur7": "valu", "fast_cells": "valu", "fast_cells_l0": "valu", "distribute": "latency", "orient_describe": "valu",
               "knn2": "mfma", "search_init": "latency", "aruco_pyramid": "hbm", "aruco_threshold": "valu",
               "aruco_contours": "latency", "aruco_decode": "latency", "aruco_finalize": "latency"}


def c5_match_leg(binding, torch, dev, O):
    """C5's matching leg: D = u8[10000 x 32] i.i.d. uniform bits (seed 5) matched 10k x 10k all-pairs,
    best + second-best, resident on the device; both kernels, each against its own bound."""
    from orb_slam2_aruco_amd import synth
    L = binding.load()
    n = 10000
    D = synth.random_descriptors(n, 5)
    d_D = torch.from_numpy(D).to(dev)
    d_n = torch.tensor([n], dtype=torch.int32, device=dev)
    outs = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    st = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(st.cuda_stream)
    res = {"workload": "10000 x 10000 all-pairs Hamming best + second-best, 256-bit descriptors (seed 5), resident in HBM",
           "pairs": n * n}
    want = None
    for name, path in (("valu", 1), ("mfma_i8", 2)):
        binding.debug_control("knn2_path", path)
        call = lambda: binding._check(L, L.orbfe_knn2_batch_device(d_D.data_ptr(), d_n.data_ptr(), 0, n, d_D.data_ptr(), d_n.data_ptr(),
                                                                   0, n, 1, 256, outs[0].data_ptr(), outs[1].data_ptr(),
                                                                   outs[2].data_ptr(), sp), "orbfe_knn2_batch_device")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(st)
        for _ in range(reps):
            call()
        e1.record(st)
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1000.0 / reps
        got = [o.cpu().numpy() for o in outs]
        if want is None:            # oracle on a sample of the queries (the full 10^8 pairs take the scalar port ~1 s per 200 queries)
            sample = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
            want = (sample, O.knn2(D[sample], D, 256))
        sample, (bi, bd, sd) = want
        assert np.array_equal(got[0][sample], bi) and np.array_equal(got[1][sample], bd) and np.array_equal(got[2][sample], sd), name
        # self-match: the best of query i is i itself at distance 0 (first index wins ties)
        assert np.array_equal(got[0], np.arange(n)) and not got[1].any(), name
        if name == "valu":      # 8 XOR + 8 BCNT lane-ops per pair
            ops = n * n * 16.0
            res[name] = {"launch_us": us, "bound": "valu", "achieved": ops / (us * 1e-6) / 1e12, "peak": VALU_LANE_OPS / 1e12,
                         "unit": "T lane-ops/s", "frac": ops / (us * 1e-6) / VALU_LANE_OPS}
        else:                   # |a ^ b| = |a| + |b| - 2 a.b: a (nq x 256) . (256 x nt) int8 GEMM, 2 * 256 ops per pair
            ops = n * n * 512.0
            res[name] = {"launch_us": us, "bound": "mfma", "achieved": ops / (us * 1e-6) / 1e12, "peak": I8_MFMA_PEAK_TOPS,
                         "unit": "TOP/s", "frac": ops / (us * 1e-6) / 1e12 / I8_MFMA_PEAK_TOPS}
    binding.debug_control("knn2_path", 0)
    res["verified_queries"] = int(len(want[0]))
    return res


def latency_mode(args):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    out = latency_record(args, make_stream(args, 0)[:max(16, min(64, args.frames))])
    print(json.dumps(out))
    if args.out:
        open(args.out, "w").write(json.dumps(out) + "\n")


def latency_record(args, frames, calls=200):
    """The drop-in mode the reference's sequential Tracking thread uses (Frame.cc:91,142; Tracking.cc:531): ONE frame per
    call through the host-pointer ABI (H2D + kernels + D2H + sync inside each call), median of `calls` calls."""
    from orb_slam2_aruco_amd import binding
    from orb_slam2_aruco_amd.pipeline import TUM1_K, TUM1_DIST, MARKER_SIZE
    ex = binding.ORBextractor(args.nfeatures, 1.2, args.nlevels, 20, 7)
    det = binding.MarkerDetector(args.dictionary)
    mt = binding.ORBmatcher(0.9, True)
    K = binding.camera_resize(np.array(TUM1_K, np.float32), (1280, 720), (args.cols, args.rows))
    D = np.array(TUM1_DIST, np.float32)
    cam = (K, D, (args.cols, args.rows))

    def run(paired):
        ex.pair_detector(det if paired else None)
        t_ex, t_det, t_sfi, t_all = [], [], [], []
        prev = None
        for i in range(calls + 20):
            img = frames[i % len(frames)]
            t0 = time.perf_counter()
            k, d = ex(img)
            t1 = time.perf_counter()
            det.detect(img, cam, MARKER_SIZE)
            t2 = time.perf_counter()
            if prev is not None:
                mt.SearchForInitialization(prev[0], prev[1], k, d, args.cols, args.rows, None, 100)
            t3 = time.perf_counter()
            prev = (k, d)
            if i >= 20:
                t_ex.append(t1 - t0); t_det.append(t2 - t1); t_sfi.append(t3 - t2); t_all.append(t3 - t0)
        ex.pair_detector(None)
        return t_ex, t_det, t_sfi, t_all

    t_ex, t_det, t_sfi, t_all = run(False)
    p_ex, p_det, p_sfi, p_all = run(True)
    med = lambda v: float(np.median(v) * 1e3)
    out = {"metric": "ms per frame, single-frame drop-in calls through the host-pointer ABI (%dx%d mono)" % (args.cols, args.rows),
           "value": med(t_all), "unit": "ms", "higher_is_better": False, "n_gpus": 1, "calls": calls, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "%s frames one at a time: orbfe_extract + orbfe_aruco_detect + orbfe_marker_poses + "
                                  "orbfe_search_for_initialization, host pointers (H2D, kernels, D2H and sync inside every call; "
                                  "includes the ctypes marshalling of the test binding)" % args.config},
           "median_ms": {"orbfe_extract": med(t_ex), "orbfe_aruco_detect+poses": med(t_det),
                         "orbfe_search_for_initialization": med(t_sfi)},
           # the same call sequence with the detector paired to the extractor (orbfe_extractor_pair_detector: one line added where the
           # reference creates the two objects): the extractor's call starts the detector on the image it uploads, the detector's call
           # -- same image -- finds its work done.  Identical results (tests/test_shims_gpu.py, test_aruco_gpu.py)
           "paired": {"value": med(p_all), "median_ms": {"orbfe_extract": med(p_ex), "orbfe_aruco_detect+poses": med(p_det),
                                                          "orbfe_search_for_initialization": med(p_sfi)}}}
    out["c_abi"] = latency_from_cpp(args, frames, K, calls)
    if args.cpu_frames > 0:
        O = oracle_module()
        orb, aru = O.OrbOracle(args.nfeatures, 1.2, args.nlevels, 20, 7), O.ArucoOracle(args.dictionary)
        c_ex, c_det, c_sfi = [], [], []
        prev = None
        for i in range(min(30, len(frames))):
            img = frames[i]
            t0 = time.perf_counter(); k, d = orb.extract(img)
            t1 = time.perf_counter()
            for m in aru.detect(img):
                O.marker_pose(m["corners"], MARKER_SIZE, K, D)
            t2 = time.perf_counter()
            if prev is not None:
                O.search_for_initialization(prev[0], prev[1], k, d, args.cols, args.rows, None, 100, 0.9, True)
            t3 = time.perf_counter()
            prev = (k, d)
            if i >= 2:
                c_ex.append(t1 - t0); c_det.append(t2 - t1); c_sfi.append(t3 - t2)
        out["cpu_baseline"] = {"value": med(c_ex) + med(c_det) + med(c_sfi), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "%d frames, oracle/ single thread" % len(c_ex),
                               "median_ms": {"extract": med(c_ex), "aruco_detect+poses": med(c_det), "search_for_initialization": med(c_sfi)}}
    return out


def latency_from_cpp(args, frames, K, calls):
    """The same call sequence from C++ (tools/latency_driver.cpp, compiled here with g++ against include/orbfe.h): the library's own
    latency, without the ctypes marshalling the Python loop above includes."""
    import subprocess
    import tempfile
    from orb_slam2_aruco_amd import binding
    root = os.path.dirname(os.path.abspath(__file__))
    libdir = os.path.dirname(os.path.abspath(binding.LIB_PATH))
    try:
        with tempfile.TemporaryDirectory() as td:
            exe, raw = os.path.join(td, "latency_driver"), os.path.join(td, "frames.u8")
            libname = os.path.basename(binding.LIB_PATH)
            subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "tools", "latency_driver.cpp"), "-o", exe, "-L" + libdir,
                            "-l:" + libname, "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
            np.ascontiguousarray(frames, np.uint8).tofile(raw)
            r = subprocess.run([exe, raw, str(len(frames)), str(args.rows), str(args.cols), str(calls), str(args.nfeatures), str(args.nlevels),
                                args.dictionary] + ["%.9g" % float(v) for v in K], check=True, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            d["what"] = "tools/latency_driver.cpp: the same three calls per frame from C++ through include/orbfe.h (no Python in the loop)"
            return d
    except Exception as e:  # noqa: BLE001
        return {"error": (getattr(e, "stderr", None) or repr(e))[-400:]}


def from_host_mode(args):
    """The PCIe-inclusive rate: the stream sits in page-locked host memory (4 rotating copies), every batch is uploaded by the
    pipeline's copy stream ahead of the engines and its record set copied back behind its matching (orbfe_pipeline_step_host)."""
    from orb_slam2_aruco_amd.pipeline import FrontEndPipeline, PinnedFrames
    B, rows, cols = args.frames, args.rows, args.cols
    frames_np = make_stream(args, 0)
    pipe = FrontEndPipeline(B, rows, cols, args.nfeatures, args.nlevels, args.dictionary, marker_capacity=args.marker_capacity)
    R = 4
    shifts = [(r * B) // R for r in range(R)]
    host = [PinnedFrames(np.roll(frames_np, -s, axis=0)) for s in shifts]
    d0 = pipe.upload(frames_np)
    pipe.warmup(d0, args.warmup)                    # resident warm-up (capacity flags, scratch sizes)
    for r in range(R):
        pipe.step_host(host[r])
    pipe.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        cur = pipe.step_host(host[i % R])
    pipe.flush()
    t_enq = time.perf_counter() - t0
    pipe.synchronize()
    dt = time.perf_counter() - t0
    st = pipe.status()
    if any(st.values()):
        raise SystemExit("front-end capacity exceeded: %r" % (st,))
    up_us, rb_us = pipe.host_copy_us()
    rec = pipe.host_records(cur)                    # what arrived in host memory
    matches = pipe.read_matches()
    r_last, r_prev = (args.steps - 1) % R, (args.steps - 2) % R
    verified = None
    if not args.no_verify:
        O = oracle_module()          # (puts tests/ on the path)
        import pipeline_check  # tests/
        fids = sorted({0, B // 2, B - 1})
        verified = pipeline_check.check_against_oracle(O, np.roll(frames_np, -shifts[r_last], axis=0), fids, rec, matches, args.nfeatures,
                                                       args.nlevels, args.dictionary, cols, rows, pipe.cam_K, pipe.cam_D, pairs=sorted({0, B // 2, B - 2}),
                                                       prev_last=np.roll(frames_np, -shifts[r_prev], axis=0)[B - 1])
    in_bytes, out_bytes = B * rows * cols, pipe.layout.nbytes
    out = {"metric": "frames/s incl. PCIe both ways (frames from page-locked host memory, record sets back to it; %dx%d mono)" % (cols, rows),
           "value": B * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
           "higher_is_better": True, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "%s from host: %d-frame %dx%d batches in page-locked memory, nFeatures=%d, %d levels, %s" % (
               args.config, B, cols, rows, args.nfeatures, args.nlevels, args.dictionary), "host_input_copies": R,
               "device_input_ring": 3, "record_sets": pipe.R},
           "pcie": {"h2d_bytes_per_step": in_bytes, "d2h_bytes_per_step": out_bytes, "h2d_GBps": in_bytes * args.steps / dt / 1e9,
                    "d2h_GBps": out_bytes * args.steps / dt / 1e9,
                    "upload_us": up_us, "upload_GBps_while_copying": in_bytes / (up_us * 1e-6) / 1e9 if up_us > 0 else None,
                    "readback_us": rb_us, "readback_GBps_while_copying": out_bytes / (rb_us * 1e-6) / 1e9 if rb_us > 0 else None,
                    "note": "achieved = bytes moved / wall time of the timed steps; both directions run concurrently with the engines"},
           "host_enqueue_ms_per_step": 1000.0 * t_enq / args.steps, "verified_frames": verified}
    print(json.dumps(out))
    if args.out:
        open(args.out, "w").write(json.dumps(out) + "\n")


def fused_bytes_per_frame(level_sizes, N, rows, cols, use_orb=True, use_aruco=True):
    """SURVEY 8d's contract: the compulsory traffic of a FUSED pipeline, per frame:
    B_orb = 3 * sumP - P0 + 1321 * N,  B_aruco = 3.33 * W * H,  B_match = (Q + T) * 32 + Q * 12 (frame t vs t - 1)."""
    P = [w * h for (w, h) in level_sizes]
    b_orb = (3 * sum(P) - P[0] + 1321 * N) if use_orb else 0.0
    b_aruco = (10.0 / 3.0) * rows * cols if use_aruco else 0.0
    b_match = (2 * N * 32 + N * 12) if use_orb else 0.0
    return b_orb, b_aruco, b_match


def extra_leg(name, steps=10):
    """A short leg of another BASELINE configuration after the clock of the default run (the driver's record should
    carry C3 and C5 too): bench.py itself on that configuration in a process of its own -- in this one a dozen HIP streams have come
    and gone, and new pipelines were measured 10 - 25 % slower (hardware queue assignment) --, its full batch when the box has the
    cores to render the stream in seconds, else a reduced one (said in the record)."""
    import subprocess
    import tempfile
    full = (os.cpu_count() or 1) >= 32
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--cpu-frames", "0", "--no-extras"]
    if not full:
        cmd += ["--frames", str({"C3": 96, "C5": 32}[name])]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "leg.json")
        r = subprocess.run(cmd + ["--out", out], capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not os.path.exists(out):
            return {"workload": name, "error": (r.stderr or r.stdout)[-400:]}
        d = json.load(open(out))
    return {"workload": d["config"]["workload"], "full_batch": full, "frames_per_step": d["config"]["frames_per_step_per_gpu"], "steps": d["steps"],
            "ms_per_step": d["ms_per_step"], "frames_per_s": d["value"] if d["value"] else d.get("diagnostic_frames_per_s"),
            "fused": d["roofline"]["step"]["fused"] if d.get("roofline") else None, "verified_frames": d["verified_frames"],
            "mean_keypoints_per_frame": d["config"]["mean_keypoints_per_frame"], "aruco_big_frame_kernel": d["config"]["aruco_big_frame_kernel"],
            "c5_match": d.get("c5_match")}


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` started WITHOUT torchrun (the form the driver uses for N = 1): become the launcher of N ranks on
    this node -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <the same arguments>` -- and pass its exit code on.  Rank 0 of the children prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(args, world, rank):
    """--launch-check: the N-rank launch of `--gpus N` without a GPU (CPU test of the launcher): every rank joins a gloo group,
    the ranks are counted with an all-reduce and listed with an all-gather, rank 0 prints a line with n_gpus and no value."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        ranks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64))
        counted, ranks = int(one.item()), [int(r.item()) for r in ranks]
        dist.barrier()
        dist.destroy_process_group()
    else:
        counted, ranks = 1, [0]
    if rank == 0:
        line = json.dumps({"metric": "launch check (no GPU work)", "value": None, "n_gpus": world, "ranks_counted": counted, "ranks": ranks,
                           "launched_by": "bench.py --gpus %d" % args.gpus if os.environ.get("ORBFE_BENCH_SELF_LAUNCHED") else "torchrun"})
        print(line)
        if args.out:
            open(args.out, "w").write(line + "\n")


def main():
    args = parse()
    if args.latency:
        return latency_mode(args)
    if args.from_host:
        return from_host_mode(args)
    # --gpus N is the number of ranks.  Under torchrun (WORLD_SIZE set) it must agree with the launcher; without it and N > 1 this
    # process launches the N ranks itself, so that the plain command `python bench.py --gpus N` measures N GPUs.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.environ["ORBFE_BENCH_SELF_LAUNCHED"] = "1"
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): start it as `python bench.py --gpus N` "
                         "or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.launch_check:
        return launch_check(args, world, rank)
    import torch
    import torch.distributed as dist

    # test hooks (single-GPU box): ORBFE_BENCH_DEVICE pins every rank to one device, ORBFE_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("ORBFE_BENCH_DEVICE"):
        local_rank = int(os.environ["ORBFE_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The batch's gather runs INSIDE the library over RCCL (orbfe_pipeline_comm_init: ncclCommInitRank, ncclSend / ncclRecv on the
    # pipeline's matching stream).  torch.distributed is only the rendezvous: a gloo group carries the 128-byte RCCL id to the ranks,
    # the barriers and the max of the elapsed times.  ORBFE_BENCH_BACKEND=gloo (test hook of a one-GPU box, where several ranks share a
    # device and RCCL cannot run) gathers the record sets through gloo from the host after every step instead.
    backend = os.environ.get("ORBFE_BENCH_BACKEND") or "rccl"
    multi = world > 1 or args.force_gather          # the gather branch runs (RCCL with world size 1 under --force-gather)
    if multi and "RANK" not in os.environ:          # --force-gather started without torchrun
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
    if multi:
        dist.init_process_group("gloo")

    from orb_slam2_aruco_amd import binding, sharding
    from orb_slam2_aruco_amd import pipeline as pipeline_mod
    from orb_slam2_aruco_amd.pipeline import FrontEndPipeline, valid_records
    L = binding.load()
    version = L.orbfe_version().decode()
    import hashlib
    lib_sha16 = hashlib.sha256(open(binding.LIB_PATH, "rb").read()).hexdigest()[:16]
    # launch ablation needs the diagnosis build (tools/ablate.sh); its numbers are never a result
    skips = {}
    late_skips = []      # ORBFE_SKIP_AFTER_WARMUP=1: the launches are left out only after the warm-up steps have filled every buffer
    for key, env in (("orb_skip", "ORBFE_ORB_SKIP"), ("aruco_skip", "ORBFE_ARUCO_SKIP")):
        if os.environ.get(env):
            if os.environ.get("ORBFE_SKIP_AFTER_WARMUP"):
                binding.debug_control(key, 0)                    # ORBFE_ERR_INVALID in the shipped library
                late_skips.append((key, int(os.environ[env])))
            else:
                binding.debug_control(key, int(os.environ[env]))
            skips[key] = int(os.environ[env])
    if "+ablation" in version:
        skips["library"] = version
    # every ORBFE_* variable that is set is recorded; the ones that change which kernels run or how they are scheduled make the
    # line a diagnostic (value null) unless they spell the default.  The list of variables and defaults comes from the library
    # (orbfe_pipeline_env_defaults), so it cannot drift from the getenv calls; a variable the library does not list, or one whose
    # default depends on the frame size, makes the line a diagnostic whatever it is set to.
    env_set = {k: v for k, v in sorted(os.environ.items()) if k.startswith("ORBFE_")}
    harmless = {"ORBFE_BENCH_DEVICE", "ORBFE_BENCH_BACKEND", "ORBFE_LIB", "ORBFE_BENCH_SELF_LAUNCHED"}
    defaults = pipeline_mod.env_defaults()
    env_nondefault = {k: v for k, v in env_set.items() if k not in harmless and (defaults.get(k) in (None, "size") or defaults[k] != v)}
    B, rows, cols = args.frames, args.rows, args.cols
    use_aruco, use_orb = not args.no_aruco, not args.no_orb

    frames_np = make_stream(args, rank)
    mkpipe = lambda: FrontEndPipeline(B, rows, cols, args.nfeatures, args.nlevels, args.dictionary, device=local_rank,
                                      marker_capacity=args.marker_capacity, use_orb=use_orb, use_aruco=use_aruco)
    pipe = mkpipe()
    gloo_gather = None
    rccl_error = None
    if multi and backend == "rccl":
        # every rank must end up on the same transport: a rank whose ncclCommInitRank fails says so over the gloo group, and then all of
        # them gather through gloo from the host (a diagnostic line, flagged in config.parallelism and gather_check.transport)
        # Before the collective ncclCommInitRank (in which the healthy ranks would wait for ever for one that cannot join): every rank
        # says over gloo whether it can open librccl and which device it sits on; a missing library or two ranks on one device (RCCL
        # refuses that) sends all of them to the fallback together.
        try:
            my_uid = pipeline_mod.comm_unique_id()      # (opens librccl; only rank 0's id is used)
            pre = None
        except Exception as e:  # noqa: BLE001
            my_uid, pre = None, repr(e)
        seats = [None] * world
        dist.all_gather_object(seats, (socket.gethostname(), int(local_rank), pre))
        if any(p_ for _, _, p_ in seats):
            rccl_error = "; ".join("rank %d: %s" % (r_, p_) for r_, (_, _, p_) in enumerate(seats) if p_)
        elif len({(h_, d_) for h_, d_, _ in seats}) < world:
            rccl_error = "ranks share a device: %s" % ", ".join("rank %d on %s:%d" % (r_, h_, d_) for r_, (h_, d_, _) in enumerate(seats))
        uid = [my_uid if rank == 0 and rccl_error is None else None]
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None and rccl_error is None:
            try:
                pipe.comm_init(uid[0], rank, world, 0)
            except Exception as e:  # noqa: BLE001
                rccl_error = repr(e)
        elif rccl_error is None:
            rccl_error = "rank 0 could not create the RCCL id"
        bad = torch.tensor([1 if rccl_error else 0], dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            errs = [None] * world
            dist.all_gather_object(errs, rccl_error)
            rccl_error = "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e) or "unknown"
            if rank == 0:
                print("bench.py: RCCL communicator not available (%s): gathering through gloo from the host" % rccl_error, file=sys.stderr)
            backend = "gloo"
            pipe = mkpipe()          # (a pipeline without a communicator)
    if multi and backend != "rccl":
        gloo_gather = sharding.RecordGather(torch.zeros(pipe.layout.nbytes, dtype=torch.uint8))
    # resident input: R copies of the stream at different time offsets, one per step in rotation, so that no step finds its
    # frames in the 256 MiB Infinity Cache (copy r = the stream rolled by r * B / R frames)
    pitch = pipe.pitch
    R = args.resident_batches or max(1, min(8, (256 << 20) // (B * rows * pitch) + 2))
    shifts = [(r * B) // R for r in range(R)]
    d_batches = [pipe.upload(np.roll(frames_np, -s, axis=0)) for s in shifts]

    # experiment hook (tools/ballast.hip, tools/sweeps.md): a kernel that keeps ONE resource of the chip busy next to every step
    ballast = None
    if os.environ.get("ORBFE_BENCH_BALLAST"):
        kind, iters = os.environ["ORBFE_BENCH_BALLAST"].split(":")
        bl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "libballast.so"))
        bl.ballast_last_us.restype = ctypes.c_float
        ballast = (bl, kind.encode(), int(iters))

    def step(r):
        if ballast:
            ballast[0].ballast_launch(ballast[1], ballast[2])
        cur = pipe.step(d_batches[r])
        if gloo_gather is not None:                 # the test hook: the record set through the host and gloo, every step
            gloo_gather(torch.from_numpy(pipe.record_bytes(cur)))
        return cur

    pipe.warmup(d_batches[0], args.warmup)
    for r in range(1, R):           # touch every resident copy once
        step(r)
    pipe.synchronize()
    for key, v in late_skips:
        binding.debug_control(key, v)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    for e in pipe.exs:
        e.enable_kernel_timing(True)
    for d in pipe.dets:
        d.enable_kernel_timing(True)
    pipe.reset_timing_history()
    last = (0, 0)
    r_hist = [R - 1]                # the resident copy each step ran on (the step before the timed region ran on the last one)
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = i % R
        last = (step(r), r)
        r_hist.append(r)
    pipe.flush()                               # the last batch's matching / gather (held back one step, see orbfe_pipeline_step)
    t_enq = time.perf_counter() - t0           # host time to enqueue all steps (the GPU runs behind it)
    pipe.synchronize()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("ORBFE_BENCH_WAVE_TIMING"):      # diagnosis build (-DORBFE_WAVE_TIMING, tools/wave_timing.sh): where the waves' lives went
        wt = (ctypes.c_ulonglong * 64)()
        if binding.load().orbfe_timing_read(wt, 1) == 0:
            for kid, name in enumerate(("orient_describe", "fast_cells", "distribute_pyr levels >= 1 (wave 0)", "distribute_pyr level 0 (wave 0)")):
                row = [wt[kid * 16 + k] for k in range(16)]
                if row[15]:
                    print("wave_timing %s: waves %d, clocks per wave by phase %s, total %.0f" % (
                        name, row[15], [round(v / row[15]) for v in row[:8]], sum(row[:8]) / row[15]) +
                          (", second passes %.3f of the waves" % (row[14] / row[15]) if row[14] else ""), file=sys.stderr)
    status = pipe.status()
    if any(status.values()):
        raise SystemExit("front-end capacity exceeded during the timed run: results incomplete, no number reported (%r)" % (status,))
    # HIP-event timings of the timed steps' launches (events recorded on the launch stream every step, no synchronisation in
    # between): the per-stage MEDIAN over the timed steps (the newest 64), not one step's events
    ex_last, det_last = pipe.last_engines()
    orb_us = ex_last.kernel_times_us(median=True) if use_orb else np.zeros(0, np.float32)
    aruco_us = det_last.kernel_times_us(median=True) if use_aruco else np.zeros(0, np.float32)
    orb_us_last = ex_last.kernel_times_us() if use_orb else np.zeros(0, np.float32)
    aruco_us_last = det_last.kernel_times_us() if use_aruco else np.zeros(0, np.float32)

    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_frames = B * args.steps * world
    cur, r_last = last
    rec = pipe.read_records(cur)
    matches = pipe.read_matches() if use_orb else None
    last_frames = np.roll(frames_np, -shifts[r_last], axis=0)       # host images of the last timed step's batch
    prev_last = np.roll(frames_np, -shifts[r_hist[-2]], axis=0)[B - 1]   # the frame in front of it in the stream the pipeline saw
    match_med = pipe.matching_times_us(median=True) if use_orb else (0.0, 0.0)
    match_last = pipe.matching_times_us() if use_orb else (0.0, 0.0)
    gather_us = pipe.gather_times_us() if (multi and backend == "rccl") else None
    gathered = [valid_records(pipe.gathered(r), use_orb) for r in range(world)] if (multi and backend == "rccl" and rank == 0) else None
    if gloo_gather is not None and rank == 0:
        gathered = [valid_records(pipe.layout.unpack(b.numpy()), use_orb) for b in gloo_gather.blocks]

    # ---- the stages ALONE on the GPU (outside the clock): each engine by itself on a stream of its own, three batches, median of its
    # HIP events.  In the timed pipeline a stage's launch time is stretched by whatever else shares the chip -- two extractor batches,
    # the detector and the matching are in flight together -- so the in-pipeline figure says how long the launch was resident, this
    # one what it costs.  (Engines of a second pipeline object: the timed one may hold an RCCL communicator.)
    alone = {}
    if not multi and not skips:
        lay = pipe.layout
        cap, mcap = pipe.cap, pipe.mcap
        st = torch.cuda.Stream(dev)
        sp = ctypes.c_void_p(st.cuda_stream)
        scratch = torch.zeros(lay.nbytes, dtype=torch.uint8, device=dev)
        base = scratch.data_ptr()
        torch.cuda.synchronize()
        if use_orb:
            ex0 = pipe.ex
            ex0.enable_kernel_timing(True)          # clears the history
            for _ in range(3):
                ex0.extract_batch_device(d_batches[r_last].data_ptr(), B, rows * pitch, rows, cols, pitch, base + lay.kps + cap * 28,
                                         base + lay.desc + cap * 32, cap, base + lay.n + 4, sp)
                torch.cuda.synchronize()
            t = ex0.kernel_times_us(median=True)
            alone.update({nm: float(v) for nm, v in zip(binding.ORBextractor.stage_names(len(t)), t)})
            outs = [torch.zeros((B, cap), dtype=torch.int32, device=dev) for _ in range(4)]
            d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
            tk, ts = [], []
            for _ in range(3):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(st)
                binding._check(L, L.orbfe_knn2_batch_device(base + lay.desc, base + lay.n, cap * 32, cap, base + lay.desc + cap * 32, base + lay.n + 4,
                                                            cap * 32, cap, B, 256, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), sp),
                               "orbfe_knn2_batch_device")
                e[1].record(st)
                binding._check(L, L.orbfe_search_for_initialization_batch_device(base + lay.kps, base + lay.desc, base + lay.n, cap, B, cols, rows,
                                                                                 None, 100, 0.9, 1, outs[3].data_ptr(), d_nm.data_ptr(), sp),
                               "orbfe_search_for_initialization_batch_device")
                e[2].record(st)
                torch.cuda.synchronize()
                tk.append(e[0].elapsed_time(e[1]) * 1000.0); ts.append(e[1].elapsed_time(e[2]) * 1000.0)
            alone["knn2"], alone["search_init"] = float(np.median(tk)), float(np.median(ts))
        if use_aruco:
            det0 = pipe.det
            det0.enable_kernel_timing(True)
            for _ in range(3):
                det0.detect_batch_device(d_batches[r_last].data_ptr(), B, rows * pitch, rows, cols, pitch, base + lay.mk, mcap, base + lay.nmk, sp)
                torch.cuda.synchronize()
            t = det0.kernel_times_us(median=True)
            alone.update({"aruco_" + nm: float(v) for nm, v in zip(binding.MarkerDetector.STAGES, t)})
        del scratch

    # ---- multi-GPU: rank 0 checks every gathered block of the last step against that rank's own stream, recomputed here (on a second
    # pipeline without a communicator)
    gather_check = None
    if multi and rank == 0:
        checked = []
        chk = None
        for r in range(world):
            got = gathered[r]
            if r == 0:
                want = valid_records(rec, use_orb)
            else:
                chk = chk or mkpipe()
                d_other = chk.upload(np.roll(make_stream(args, r), -shifts[r_last], axis=0))
                want = valid_records(chk.read_records(chk.step(d_other)), use_orb)
                del d_other
            assert got == want, "gathered records of rank %d differ from that rank's stream recomputed on rank 0" % r
            checked.append(r)
        del chk
        gather_check = {"ranks": checked, "frames_per_rank": B, "what": "n, keypoints, descriptors, marker ids / corners / poses of "
                        "every frame, byte for byte", "transport": "RCCL send / recv inside liborbfe (orbfe_pipeline_comm_init)" if backend == "rccl"
                        else "gloo from the host (test hook)" if not rccl_error else "gloo from the host: RCCL was not available (%s)" % rccl_error}

    if rank == 0:
        stages = {nm: float(v) for nm, v in zip(binding.ORBextractor.stage_names(len(orb_us)), orb_us)}   # blur7 runs on a second stream
        stages_last = {nm: float(v) for nm, v in zip(binding.ORBextractor.stage_names(len(orb_us_last)), orb_us_last)}
        if use_orb:
            stages["knn2"], stages["search_init"] = match_med
            stages_last["knn2"], stages_last["search_init"] = match_last
        if use_aruco:
            for nm, v in zip(binding.MarkerDetector.STAGES, aruco_us):
                stages["aruco_" + nm] = float(v)
            for nm, v in zip(binding.MarkerDetector.STAGES, aruco_us_last):
                stages_last["aruco_" + nm] = float(v)
        # algorithmic bytes per frame of each stage (DESIGN.md "roofline": the terms of SURVEY 8d's B_orb / B_aruco)
        if pipe.ex is not None:
            sizes = pipe.ex.level_sizes()
        else:   # --no-orb: the level geometry from an extractor of its own (one frame through it)
            ex_geo = binding.ORBextractor(args.nfeatures, 1.2, args.nlevels, 20, 7, device=local_rank)
            ex_geo(last_frames[0])
            sizes = ex_geo.level_sizes()
        P = [w * h for (w, h) in sizes]
        sumP, P0 = sum(P), P[0]
        N = float(rec["n"].mean()) if use_orb else 0.0
        Ncand = 24.0          # rectangle candidates decoded per frame, upper end of what the synthetic streams produce
        l0_split = "fast_cells_l0" in stages
        alg = {"resize": (sumP - P[-1]) + (sumP - P0), "fast_cells": sumP - P0 if l0_split else sumP, "fast_cells_l0": P0, "blur7": 2 * sumP,
               "orient_describe": N * (749 + 512 + 60), "distribute": N * 8,
               "knn2": 2 * N * 32 + N * 12, "search_init": 2 * N * (32 + 28) + N * 4}
        if use_aruco:
            alg.update(binding.MarkerDetector.algorithmic_bytes(rows, cols))
            alg["aruco_decode"] = Ncand * 2 * 35 * 35
            alg["aruco_finalize"] = Ncand * 36
        fl = lambda k: B                                        # frames (or frame pairs) one launch covers
        # counters are NOT measured in this run: they come from the committed rocprofv3 --pmc profile of this configuration
        # (tools/pmc.py + tools/make_profiles.py) and carry that profile's id; a custom size reports none
        pmc = load_profile("pmc_stage", args.config) if not (args.custom or args.reduced) else None
        traffic = load_profile("traffic", args.config) if not (args.custom or args.reduced) else None
        prof_id = lambda d: ({"file": d.get("_file"), "profile": d.get("_profile"), "library_commit": d.get("_commit"),
                              "library_sha16": d.get("_library_sha16")} if d else None)
        per_stage = {}
        # the two FAST launches (level 0 early, levels >= 1 after the resize chain) are one kernel in the profiles: its counters are
        # shared out by pixels
        share = {"fast_cells_l0": ("fast_cells", P0 / float(sumP)), "fast_cells": ("fast_cells", 1.0 - P0 / float(sumP))} if l0_split else {}
        for k, us in stages.items():
            ab = alg.get(k, 0) * fl(k)
            ent = {"bound": STAGE_BOUND.get(k, "latency"), "launch_us": us, "algorithmic_bytes_per_launch": ab,
                   "GBps": ab / (us * 1e-6) / 1e9 if us > 0 else 0.0}
            ent["hbm_frac"] = ent["GBps"] / HBM_PEAK_GBPS
            if alone.get(k):
                ent["launch_us_alone"] = alone[k]
                ent["hbm_frac_alone"] = ab / (alone[k] * 1e-6) / 1e9 / HBM_PEAK_GBPS
            pk, frac_of = share.get(k, (k, 1.0))
            ent["traffic"] = (traffic.get(pk) * frac_of if traffic.get(pk) is not None else None) if traffic else None
            if pmc and pk in pmc:    # VALU issue time of the stage's launches (instruction counts x 4 cycles / 1024 SIMDs / 2.4 GHz)
                vu = pmc[pk].get("valu_us")
                ent["valu_us"] = vu * frac_of if vu is not None else None
                # profile-run numerator over this run's denominator: an estimate, valid while the kernels are the profiled ones
                ent["valu_frac"] = ent["valu_us"] / us if us > 0 and vu is not None else None
                ent["lane_utilisation"] = pmc[pk].get("lane_utilisation")
            per_stage[k] = ent
        # The dominant kernel.  The step is bound by VALU issue with all engines overlapped, so where the
        # committed PMC profile of this configuration is available it is the stage that takes most of that resource -- the same
        # stage whose removal shortens the step most in the ablation study (profiles/r02_ablation.txt: FAST) -- else the longest
        # launch.  "longest_launch" names the latter in any case.
        longest = max(stages, key=lambda k: stages[k]) if stages else None
        with_valu = [k for k in stages if per_stage[k].get("valu_us") is not None]
        dom = max(with_valu, key=lambda k: per_stage[k]["valu_us"]) if with_valu else longest
        roof = None
        if dom is not None and stages[dom] > 0:
            d = per_stage[dom]
            roof = {"bound": d["bound"], "kernel": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": d["hbm_frac"], "traffic": d["traffic"], "launch_us": d["launch_us"],
                    "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "frames_per_launch": fl(dom),
                    "longest_launch": longest,
                    "alone": {"launch_us": d.get("launch_us_alone"), "frac": d.get("hbm_frac_alone"),
                              "note": "the same stage with nothing else on the GPU (three batches after the clock, median)"} if d.get("launch_us_alone") else None,
                    "valu_us": d.get("valu_us"), "valu_frac": d.get("valu_frac"),
                    "note": "dominant kernel of the last timed step = the stage with the largest VALU issue time (the resource that "
                            "bounds the step; without a PMC profile of this configuration: the longest launch); launch_us = HIP events on the stream the stage is launched "
                            "on, the other engines running concurrently (standalone times: DESIGN.md section 6); achieved = "
                            "algorithmic bytes / launch time against the HBM peak also where the stage's bound is not HBM -- "
                            "'stages' carries every stage's bound, and valu_us / valu_frac where the committed PMC profile of "
                            "this configuration has them",
                    "stages": per_stage}
            step_s = elapsed / args.steps
            # SURVEY 8d's contract: the compulsory traffic of a FUSED pipeline, per frame
            #   B_orb = 3 * sumP - P0 + 1321 * N,   B_aruco = 3.33 * W * H,   B_match = (Q + T) * 32 + Q * 12 (frame t vs t-1)
            b_orb, b_aruco, b_match = fused_bytes_per_frame(sizes, N, rows, cols, use_orb, use_aruco)
            fused_bytes = (b_orb + b_aruco) * B
            fused_bytes_m = fused_bytes + b_match * B
            step_bytes = sum(alg.get(k, 0) * B for k in stages)
            mk = lambda nbytes: {"algorithmic_bytes_per_step": nbytes, "GBps": nbytes / step_s / 1e9,
                                 "frac": nbytes / step_s / 1e9 / HBM_PEAK_GBPS}
            roof["step"] = {
                "fused": dict(mk(fused_bytes), bytes_per_frame={"B_orb": b_orb, "B_aruco": b_aruco},
                              note="SURVEY 8d / BASELINE.md section 4 contract bytes: (3*sumP - P0 + 1321*N) + 3.33*W*H per frame, "
                                   "x frames per step, over the measured step time against the 8 TB/s HBM peak"),
                "fused_with_match": dict(mk(fused_bytes_m), bytes_per_pair=b_match),
                "per_stage": dict(mk(step_bytes), note="sum of the UNFUSED stages' algorithmic bytes (every pyramid level counted "
                                  "once per kernel that reads or writes it): what the kernels as built must move, 1.4x the contract"),
                "frac": fused_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
                "ms_per_step": 1000.0 * step_s}
            if traffic:
                tsum = sum(v for k, v in traffic.items() if not k.startswith("_") and isinstance(v, (int, float)) and k in stages)
                roof["step"]["counter_traffic"] = {"bytes_per_step": tsum, "over_fused": tsum / fused_bytes if fused_bytes else None,
                                                   "source": prof_id(traffic)}
            roof["counters_from"] = {"pmc_stage": prof_id(pmc), "traffic": prof_id(traffic),
                                     "note": "traffic / valu_us / lane_utilisation are read from committed profiles of this "
                                             "configuration, not measured in this run; launch_us and ms_per_step are this run's"}
            # the counters describe the library they were measured on: when that is not the library this run loaded, the record
            # says so and the headline roofline entry carries no counter-derived number (the per-stage table keeps them, tagged)
            prof_shas = {d.get("_library_sha16") for d in (pmc, traffic) if d}
            stale = bool(prof_shas) and prof_shas != {lib_sha16}
            roof["counters_stale"] = stale
            if stale:
                roof["counters_from"]["stale"] = "profiled library %s != this run's library %s: re-run tools/profile_round.sh" % (
                    ", ".join(sorted(x or "?" for x in prof_shas)), lib_sha16)
                for k_ in ("traffic", "valu_us", "valu_frac"):
                    roof[k_] = None
            roof["launch_us_is"] = "median over the %d timed steps (newest 64)" % args.steps

        # (a multi-GPU line whose gather fell back to gloo from the host measures another transport and another schedule: diagnostic)
        invalid = bool(skips) or args.no_aruco or args.no_orb or bool(env_nondefault) or bool(rccl_error) or bool(ballast)
        verified = None
        if not args.no_verify and not skips:
            # outside the clock: frames {0, B/2, B-1} and pairs {0, B/2, B-2} of the LAST timed step against the oracle
            O = oracle_module()
            import pipeline_check  # tests/
            fids = sorted({0, B // 2, B - 1})
            pairs = sorted({0, B // 2, B - 2}) if B >= 2 else []
            verified = pipeline_check.check_against_oracle(O, last_frames, fids, rec, matches, args.nfeatures, args.nlevels,
                                                           args.dictionary, cols, rows, pipe.cam_K, pipe.cam_D,
                                                           use_orb=use_orb, use_aruco=use_aruco, pairs=pairs,
                                                           prev_last=prev_last if use_orb and args.steps + R > 1 else None)
        cpu = None
        if world == 1 and args.cpu_frames > 0 and not skips:
            cpu = cpu_baseline(args, frames_np)
        c5 = None
        if args.config == "C5" and world == 1 and use_orb and not args.custom and not args.reduced:
            c5 = c5_match_leg(binding, torch, dev, oracle_module())
        cfg_name = "custom" if args.custom else args.config + (" at %d frames per step" % B if args.reduced else "")
        out = {
            "metric": "frames/s (ORB+ArUco extract+match, 640x480 mono)" if (rows, cols) == (480, 640) else
                      "frames/s (ORB+ArUco extract+match, %dx%d mono)" % (cols, rows),
            "value": None if invalid else total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d-frame %dx%d mono stream per GPU, nFeatures=%d, %d levels, %s dictionary; "
                                   "per frame: %s%s"
                                   % (cfg_name, B, cols, rows, args.nfeatures, args.nlevels, args.dictionary,
                                      "ORB extract + knn2 all-pairs + SearchForInitialization vs previous frame" if use_orb
                                      else "(ORB + matching legs DISABLED: diagnostic run)",
                                      " + ArUco detect incl. IPPE marker poses" if use_aruco else " (ArUco leg DISABLED: diagnostic run)"),
                       "frames_per_step_per_gpu": B, "mean_keypoints_per_frame": N, "marker_records_per_frame": pipe.mcap,
                       "result_record_bytes_per_step_per_gpu": pipe.layout.nbytes, "resident_batches": R,
                       "engine_sets": pipe.D, "record_sets": pipe.R, "engine_phase_lock_stage": pipe.phase_pin, "matching_deferred_one_step": pipe.defer_post,
                       "detector_pyramid_in_line": pipe.det_nofork, "detector_phase_stage": pipe.det_pin, "aruco_big_frame_kernel": pipe.big_frames,
                       "host": "C++ (orbfe_pipeline_* in liborbfe.so); Python only marshals", "library": version,
                       "library_sha16": lib_sha16, "env": env_set, "env_nondefault": env_nondefault or None,
                       "parallelism": "stream-per-gpu x%d, %s" % (world, ("%s gather to rank 0" % ("RCCL" if backend == "rccl" else backend))
                                                                  if multi else "no collective (one rank)")},
            "roofline": roof, "cpu_baseline": cpu, "verified_frames": verified, "skips": skips or None,
            "stage_us": stages, "stage_us_last_step": stages_last, "host_enqueue_ms_per_step": 1000.0 * t_enq / args.steps,
        }
        if invalid:
            out["diagnostic_frames_per_s"] = total_frames / elapsed
        if ballast:      # its duration next to the last step, and alone
            in_pipe = float(ballast[0].ballast_last_us())
            torch.cuda.synchronize()
            alone = []
            for _ in range(5):
                ballast[0].ballast_launch(ballast[1], ballast[2])
                alone.append(float(ballast[0].ballast_last_us()))
            out["ballast"] = {"kind": ballast[1].decode(), "iterations": ballast[2], "us_next_to_a_step": in_pipe, "us_alone": sorted(alone)[2]}
        if gather_check:
            out["gather_check"] = gather_check
            out["gather_us"] = gather_us
        if c5:
            out["c5_match"] = c5
        # ---- after the clock, default run only: the drop-in latency and short legs of the other configurations, so that the
        # driver's record carries them and not only the builder's profile files
        # (not in the measurement scripts' runs: those pass --no-verify / --cpu-frames 0)
        if (args.config == "C2" and not args.custom and not args.reduced and not multi and not invalid and not args.no_extras
                and not args.no_verify and args.cpu_frames > 0):
            del d_batches
            pipe = None
            extras = {}
            try:
                # in a process of its own: what a drop-in application looks like (here a dozen HIP streams have come and gone, and the
                # paired extractor / detector calls were measured to land on one hardware queue: 0.80 ms instead of 0.57)
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--latency", "--cpu-frames", "0"], capture_output=True, text=True, timeout=300)
                extras["latency"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:      # an extra must not take the headline line with it
                extras["latency"] = {"error": repr(e)}
            for leg in ("C3", "C5"):
                try:
                    extras[leg] = extra_leg(leg)
                except Exception as e:
                    extras[leg] = {"workload": leg, "error": repr(e)}
            out["extras"] = extras
        line = json.dumps(out)
        print(line)
        if args.out:
            open(args.out, "w").write(line + "\n")
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
