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
