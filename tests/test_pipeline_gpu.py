"""GPU tests of the batched pipeline bench.py times (tests/pipeline_case.py runs in its own process: torch + the HIP library)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _case(*argv, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(HERE, "pipeline_case.py"), *map(str, argv)], capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_timed_path_c2_full_batch_against_oracle():
    """The 300-frame C2 batch exactly as bench.py runs it: all 300 frames and all 299 pairs."""
    _case("C2", 300, 3)


@pytest.mark.gpu
def test_timed_path_c3_64_frames_against_oracle():
    """A 64-frame C3 batch (1280x720, nFeatures 2000, ARUCO_MIP_25h7): all frames, all 63 pairs."""
    _case("C3", 64, 2)


@pytest.mark.gpu
def test_bench_line_is_self_checking(tmp_path):
    """bench.py's own JSON line: verified_frames present, no skips, value set; the ablation keys are refused by the shipped library."""
    import json
    out = tmp_path / "b.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "24", "--steps", "3", "--warmup", "1",
                        "--cpu-frames", "4", "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(out.read_text())
    assert d["value"] and d["skips"] is None and d["verified_frames"]["frames"] == [0, 12, 23] and d["verified_frames"]["pairs"] == [0, 12, 22]
    assert d["roofline"]["stages"] and d["roofline"]["step"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    env = dict(os.environ, ORBFE_ARUCO_SKIP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--steps", "1", "--warmup", "1", "--cpu-frames", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0 and "orbfe_debug_control" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gather_is_verified(tmp_path):
    """bench.py's N > 1 branch end to end on a single-GPU box: two ranks pinned to device 0, gloo instead of RCCL (the test
    hooks of bench.py), double-buffered communication stream included.  Rank 0 compares the gathered record block of each rank,
    byte for byte, with that rank's stream recomputed locally ("gather_check")."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "b2.json"
    env = dict(os.environ, ORBFE_BENCH_DEVICE="0", ORBFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--steps", "4",
                        "--warmup", "2", "--cpu-frames", "0", "--out", str(out)],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 2 and d["gather_check"]["ranks"] == [0, 1] and d["value"] > 0
    assert d["verified_frames"]["frames"] == [0, 8, 15]


@pytest.mark.gpu
def test_bench_gpus_2_plain_command_launches_two_ranks(tmp_path):
    """The driver's plain command, `python bench.py --gpus 2`, with no launcher around it: bench.py starts the two ranks itself
    (both pinned to device 0 over gloo here, the one-GPU box's test hooks) and the line says n_gpus 2 with both ranks' records
    gathered and checked."""
    import json
    out = tmp_path / "b2p.json"
    env = dict(os.environ, ORBFE_BENCH_DEVICE="0", ORBFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--steps", "3", "--warmup", "1",
                        "--cpu-frames", "0", "--out", str(out)], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 2 and d["gather_check"]["ranks"] == [0, 1] and d["value"] > 0


@pytest.mark.gpu
def test_bench_falls_back_to_gloo_when_rccl_refuses(tmp_path):
    """Two ranks on ONE device with the default transport: RCCL cannot be used (duplicate GPU) -- the ranks find that out over the gloo
    group BEFORE the collective ncclCommInitRank, all of them gather through gloo from the host, the gathered records are still checked,
    and the line is a DIAGNOSTIC (value null, diagnostic_frames_per_s): another transport and schedule than the one the metric names."""
    import json
    out = tmp_path / "bfb.json"
    env = dict(os.environ, ORBFE_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ORBFE_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--steps", "3", "--warmup", "1",
                        "--cpu-frames", "0", "--out", str(out)], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 2 and d["gather_check"]["ranks"] == [0, 1] and d["value"] is None and d["diagnostic_frames_per_s"] > 0
    assert "RCCL was not available" in d["gather_check"]["transport"] and "gloo" in d["config"]["parallelism"]
    assert "share a device" in d["gather_check"]["transport"]


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.gpu
def test_bench_rccl_gather_branch_world_size_1(tmp_path):
    """The RCCL branch of bench.py / FrontEndPipeline on the hardware that is there: `--gpus 1 --force-gather` initialises the
    nccl (= RCCL) process group with world size 1 and runs the batch's gather on the DEVICE record buffer, on the communication
    stream, with the gather_done waits of the double-buffered record sets -- the code path N > 1 takes, minus the peers.
    gather_check compares the gathered block with the records byte for byte."""
    import json
    out = tmp_path / "bg.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ORBFE_BENCH_BACKEND", "ORBFE_BENCH_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--frames", "24", "--steps", "6",
                        "--warmup", "2", "--cpu-frames", "0", "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 1 and d["gather_check"]["ranks"] == [0] and d["value"] > 0 and d["gather_us"] > 0
    assert "RCCL gather" in d["config"]["parallelism"]
    assert d["verified_frames"]["frames"] == [0, 12, 23]


@pytest.mark.gpu
def test_bench_c4_eight_ranks_on_one_gpu(tmp_path):
    """BASELINE configs[3] (C4: 8 independent 1280x720 / nFeatures 2000 / ARUCO_MIP_25h7 streams, one per rank, seeds 1000 + 2000 r) as far
    as a one-GPU box can run it: 8 ranks pinned to device 0, gloo instead of RCCL, 8 frames per step.  Rank 0 receives 8 blocks and
    recomputes every rank's stream (gather_check.ranks == [0..7])."""
    import json
    from orb_slam2_aruco_amd import sharding
    assert [sharding.stream_seed(r) for r in range(8)] == [1000 + 2000 * r for r in range(8)]
    out = tmp_path / "c4.json"
    env = dict(os.environ, ORBFE_BENCH_DEVICE="0", ORBFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "C4", "--frames", "8",
                        "--steps", "3", "--warmup", "1", "--cpu-frames", "0", "--out", str(out)],
                       capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 8 and d["gather_check"]["ranks"] == list(range(8)) and d["gather_check"]["frames_per_rank"] == 8
    assert d["config"]["workload"].startswith("C4 at 8 frames per step: 8-frame 1280x720") and "ARUCO_MIP_25h7" in d["config"]["workload"]
    assert d["value"] > 0 and d["verified_frames"]["frames"] == [0, 4, 7]


@pytest.mark.gpu
@pytest.mark.parametrize("force_gather", [0, 1])
def test_pipeline_from_cpp_without_python(tmp_path, force_gather):
    """The batched-video mode is C++ inside liborbfe.so (orbfe_pipeline_*): tests/pipeline_driver.cpp -- hipcc, include/orbfe.h, no
    Python, no PyTorch -- runs a 24-frame batch three times and writes the record set and the match counts; the Python wrapper run on
    the same frames the same way must give the same bytes (keypoints, descriptors, markers, poses of every frame; the halo slot; the
    match count of every pair incl. the one across the batch boundary).  force_gather = 1: with an RCCL communicator of one rank
    created by the library (orbfe_pipeline_comm_unique_id / _comm_init); the record set read back is the gathered block."""
    import numpy as np
    from orb_slam2_aruco_amd import synth, binding
    from orb_slam2_aruco_amd.pipeline import FrontEndPipeline, valid_records
    B, rows, cols = 24, 480, 640
    frames = synth.stream(rows, cols, B, 4242, "ARUCO", n_markers=4)
    raw = tmp_path / "frames.u8"
    frames.tofile(raw)
    exe = tmp_path / "pipeline_driver"
    libdir = os.path.dirname(binding.LIB_PATH)
    r = subprocess.run(["hipcc", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "pipeline_driver.cpp"), "-o", str(exe), "-L" + libdir, "-lorbfe",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "records.bin"
    r = subprocess.run([str(exe), str(raw), str(B), str(rows), str(cols), "3", str(out), str(force_gather)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok frames 24 " in r.stdout, r.stdout + r.stderr[-2000:]      # (RCCL prints its version banner first)
    pipe = FrontEndPipeline(B, rows, cols)
    d = pipe.upload(frames)
    for _ in range(3):
        cur = pipe.step(d)
    rec = pipe.read_records(cur)
    nm = pipe.read_matches()["nmatches"]
    blob = np.fromfile(out, np.uint8)
    got = pipe.layout.unpack(blob[:pipe.layout.nbytes])
    got_nm = blob[pipe.layout.nbytes:].view(np.int32)
    assert valid_records(got) == valid_records(rec)
    assert int(got["halo_n"][0]) == int(rec["halo_n"][0]) == int(rec["n"][B - 1]) > 0      # the stream went on: the previous step's last frame
    assert np.array_equal(got_nm, nm) and nm[0] > 0 and nm[1:].sum() > 0
    assert "keypoints %d " % int(rec["n"].sum()) in r.stdout


@pytest.mark.gpu
def test_bench_from_host_overlaps_upload_and_readback(tmp_path):
    """bench.py --from-host: the frames start in page-locked host memory and the record sets end there (orbfe_pipeline_step_host: a ring
    of device input buffers on a copy stream, read-back on another); the records that arrived on the host are checked against the
    oracle, incl. the pair across the batch boundary."""
    import json
    out = tmp_path / "fh.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--from-host", "--frames", "24", "--steps", "7", "--warmup", "1",
                        "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(out.read_text())
    assert d["value"] > 0 and d["pcie"]["h2d_GBps"] > 0 and d["pcie"]["d2h_GBps"] > 0
    v = d["verified_frames"]
    assert v["frames"] == [0, 12, 23] and v["pairs"] == [0, 12, 22] and v["boundary_pair_checked"] and v["keypoints_checked"] > 2500


def _build_driver_and_fake_rccl(tmp_path):
    from orb_slam2_aruco_amd import binding
    libdir = os.path.dirname(binding.LIB_PATH)
    exe, fake = tmp_path / "pipeline_driver", tmp_path / "libfake_rccl.so"
    r = subprocess.run(["hipcc", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "pipeline_driver.cpp"), "-o", str(exe), "-L" + libdir, "-lorbfe",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(["hipcc", "-O2", "-std=c++17", "-Wall", "-Werror", "-shared", "-fPIC", os.path.join(HERE, "fake_rccl.cpp"), "-o", str(fake), "-lrt"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe, fake


def _two_rank_streams(tmp_path, B, rows, cols, nb):
    from orb_slam2_aruco_amd import synth
    raws = []
    for r in range(2):
        frames = synth.stream(rows, cols, B * nb, 900 + 1000 * r, "ARUCO", n_markers=3)
        raw = tmp_path / ("frames%d.u8" % r)
        frames.tofile(raw)
        raws.append(raw)
    return raws


@pytest.mark.gpu
def test_gather_between_two_processes_on_one_gpu(tmp_path):
    """The N > 1 branch of the gather in C++, with a peer: two processes of tests/pipeline_driver.cpp on device 0, the library's RCCL entry
    points bound to tests/fake_rccl.cpp (ORBFE_RCCL_LIB; shared-memory mailboxes).  Rank 1 sends every batch's record set, rank 0 receives
    it next to its own (receive loop, own-block copy); six steps over three different batches with four record sets, so the sets and their
    receive blocks rotate and wrap; rank 0 reads the blocks of batch s while batch s + 1 is in flight (orbfe_pipeline_gathered_wait / _set /
    _release).  Every block of every step must equal, over the defined part of the records, what that rank computes for that step
    on its own (the same driver without a communicator, synchronised and read back after every step)."""
    import numpy as np
    from orb_slam2_aruco_amd.pipeline import FrontEndPipeline, valid_records
    B, rows, cols, nb, steps = 6, 480, 640, 3, 6
    exe, fake = _build_driver_and_fake_rccl(tmp_path)
    raws = _two_rank_streams(tmp_path, B, rows, cols, nb)
    for r in range(2):   # what each rank's record sets must be, step by step
        q = subprocess.run(list(map(str, [exe, raws[r], B, rows, cols, steps, tmp_path / ("ref%d" % r), "gather", 0, 1, "-", 1])), capture_output=True, text=True, timeout=600)
        assert q.returncode == 0 and "ok rank 0 of 1" in q.stdout, q.stdout + q.stderr[-2000:]
    env = dict(os.environ, ORBFE_RCCL_LIB=str(fake), FAKE_RCCL_TIMEOUT_S="120")
    idf = tmp_path / "comm.id"
    procs = [subprocess.Popen(list(map(str, [exe, raws[r], B, rows, cols, steps, tmp_path / "got", "gather", r, 2, idf])), env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r in range(2):
        assert procs[r].returncode == 0 and ("ok rank %d of 2" % r) in outs[r][0], outs[r][0] + outs[r][1][-2000:]
    lay = FrontEndPipeline(B, rows, cols).layout
    nkp = 0
    for s in range(steps):
        for r in range(2):
            got = lay.unpack(np.fromfile(tmp_path / ("got.step%d.rank%d" % (s, r)), np.uint8))
            want = lay.unpack(np.fromfile(tmp_path / ("ref%d.own.step%d" % (r, s)), np.uint8))
            assert valid_records(got) == valid_records(want), (s, r)
            assert int(got["halo_n"][0]) == int(want["halo_n"][0]) and got["halo_kps"][0, :int(got["halo_n"][0])].tobytes() == want["halo_kps"][0, :int(want["halo_n"][0])].tobytes()
            nkp += int(got["n"].sum())
        # the two ranks' streams differ, and so do consecutive batches of one rank: a block in the wrong place would not go unnoticed
        a = lay.unpack(np.fromfile(tmp_path / ("got.step%d.rank0" % s), np.uint8))
        b = lay.unpack(np.fromfile(tmp_path / ("got.step%d.rank1" % s), np.uint8))
        assert valid_records(a) != valid_records(b)
        if s:
            prev = lay.unpack(np.fromfile(tmp_path / ("got.step%d.rank1" % (s - 1)), np.uint8))
            assert valid_records(prev) != valid_records(b)
    assert nkp > 2 * steps * B * 100


@pytest.mark.gpu
def test_gather_error_paths_fail_loudly_and_do_not_hang(tmp_path):
    """A send that fails inside the RCCL group (the stub's second ncclSend on rank 1): that rank's step returns an error that names the
    call, the group having been closed; rank 0, whose peer never delivers, gets the stub's time-out through ncclGroupEnd as an error of
    its step instead of waiting for ever.  Both processes end by themselves."""
    B, rows, cols, nb, steps = 6, 480, 640, 2, 4
    exe, fake = _build_driver_and_fake_rccl(tmp_path)
    raws = _two_rank_streams(tmp_path, B, rows, cols, nb)
    env = dict(os.environ, ORBFE_RCCL_LIB=str(fake), FAKE_RCCL_TIMEOUT_S="10")
    idf = tmp_path / "comm.id"
    procs = [subprocess.Popen(list(map(str, [exe, raws[r], B, rows, cols, steps, tmp_path / "got", "gather", r, 2, idf])),
                              env=dict(env, FAKE_RCCL_FAIL_SEND_AT="2") if r == 1 else env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert procs[1].returncode != 0 and "ncclSend failed" in outs[1][1] and "injected failure" in outs[1][1], outs[1]
    assert procs[0].returncode != 0 and "timed out waiting for a peer" in outs[0][1], outs[0]


@pytest.mark.gpu
def test_pipeline_from_a_directory_of_pgm_frames(tmp_path):
    """Frames from disk: tests/pipeline_driver.cpp reads a directory of binary PGM files in name order (the reference reads its video
    frame by frame, Examples/Monocular/mono_cvcam.cc:128-148) and runs them in batches; the records equal those of the same frames
    handed over as one raw blob."""
    import numpy as np
    from orb_slam2_aruco_amd import synth
    B, rows, cols = 8, 480, 640
    exe, _ = _build_driver_and_fake_rccl(tmp_path)
    frames = synth.stream(rows, cols, 2 * B, 77, "ARUCO", n_markers=2)
    d = tmp_path / "seq"
    d.mkdir()
    for i, f in enumerate(frames):
        with open(d / ("%06d.pgm" % i), "wb") as fh:
            fh.write(b"P5\n# frame %d\n%d %d\n255\n" % (i, cols, rows) + f.tobytes())
    raw = tmp_path / "frames.u8"
    frames.tofile(raw)
    outs = []
    for src, name in ((d, "a.bin"), (raw, "b.bin")):
        r = subprocess.run(list(map(str, [exe, src, B, rows, cols, 4, tmp_path / name])), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok frames 8 " in r.stdout, r.stdout + r.stderr[-2000:]
        outs.append(np.fromfile(tmp_path / name, np.uint8))
    assert np.array_equal(outs[0], outs[1]) and outs[0].size > 0



@pytest.mark.gpu
def test_from_host_with_two_record_sets_keeps_the_halo_slot_intact(tmp_path):
    """with two record sets the halo write of batch i (slot 0 of the set batch i + 1 is written to) must wait for the host
    copy of that set's previous contents, which reads the whole set, halo included.  bench.py --from-host checks exactly the host copies
    -- keypoints, markers, the pair across the batch boundary (halo_n / halo_desc) -- against the oracle; here with ORBFE_RECORD_SETS=2."""
    import json
    out = tmp_path / "fh2.json"
    env = dict(os.environ, ORBFE_RECORD_SETS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--from-host", "--frames", "24", "--steps", "9", "--warmup", "1",
                        "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(out.read_text())
    v = d["verified_frames"]
    assert d["config"]["record_sets"] == 2 and v["frames"] == [0, 12, 23] and v["boundary_pair_checked"] and v["keypoints_checked"] > 2500
