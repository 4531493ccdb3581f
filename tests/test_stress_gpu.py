"""The randomised cross-checks of tools/stress*.py inside the test suite (they ran only on the builder's lease): fixed
seeds, bounded case counts, every case GPU against oracle -- random frame sizes and aspect ratios, extractor parameters, all
dictionaries, error-correction rates, noise and flat frames, the detector's other modes over sequences, the matching entry points with
clustered descriptors.  The scripts print one summary line; any mismatch fails the test with the script's own report."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, *argv, timeout=600, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *map(str, argv)], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_extractor_and_detector_on_random_frames():
    out = _run("stress.py", 60, 404)
    m = re.search(r"(\d+) cases, (\d+) mismatches, (\d+) refused", out)
    assert m and int(m.group(1)) == 60 and int(m.group(2)) == 0 and int(m.group(3)) <= 6, out[-3000:]


def test_detector_on_random_frames_dictionaries_and_rates():
    out = _run("stress_aruco.py", 100, 405)
    m = re.search(r"(\d+) cases, (\d+) mismatches", out)
    assert m and int(m.group(1)) == 100 and int(m.group(2)) == 0, out[-3000:]


def test_detector_with_the_speck_passes_on_random_frames():
    """The same random frames (other seed) and the detector-mode sequences with the speck passes as a launch of their own (ORBFE_ARUCO_SPECKS=1: a
    shipped switch, off by default; inside the relay kernels of full batches: tests/test_aruco_gpu.py): what they clear never changes a marker, a corner or a rectangle candidate."""
    out = _run("stress_aruco.py", 100, 505, env={"ORBFE_ARUCO_SPECKS": "1"})
    m = re.search(r"(\d+) cases, (\d+) mismatches", out)
    assert m and int(m.group(1)) == 100 and int(m.group(2)) == 0, out[-3000:]
    out = _run("stress_modes.py", 20, 506, env={"ORBFE_ARUCO_SPECKS": "1"})
    m = re.search(r"(\d+) mismatches", out)
    assert m and int(m.group(1)) == 0, out[-3000:]


def test_detector_modes_on_random_sequences():
    out = _run("stress_modes.py", 40, 406)
    m = re.search(r"(\d+) mismatches", out)
    assert m and int(m.group(1)) == 0, out[-3000:]


def test_matching_entry_points_on_random_inputs():
    out = _run("stress_match.py", 30, 407)
    m = re.search(r"(\d+) mismatches", out)
    assert m and int(m.group(1)) == 0, out[-3000:]
