"""`python bench.py --gpus N` is the command the driver runs: without a launcher around it, it must start N ranks itself
(the flag was parsed and never read, so a scaling run would have produced eight N = 1 lines).  --launch-check stops
after the ranks have counted each other over gloo, so the launcher is testable without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    out = tmp_path / "l.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--out", str(out)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(out.read_text())
    assert d["n_gpus"] == 2 and d["ranks_counted"] == 2 and d["ranks"] == [0, 1] and d["launched_by"] == "bench.py --gpus 2"
    # exactly one JSON line on stdout (rank 0's)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_gpus_must_agree_with_the_launcher():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in r.stderr


def test_gpus_1_stays_in_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=_env())
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
