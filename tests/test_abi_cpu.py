"""CPU tests: the C-ABI library loads and exports every symbol include/orbfe.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from orb_slam2_aruco_amd import binding
    assert _header_symbols() == sorted(binding.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from orb_slam2_aruco_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(binding.LIB_PATH)
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the product path must refuse to run: there is no CPU fallback behind the ABI."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from orb_slam2_aruco_amd import binding
    with pytest.raises(binding.OrbfeError):
        binding.ORBextractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(binding.OrbfeError):
        binding.MarkerDetector("ARUCO")
    L = binding.load()
    assert L.orbfe_device_count() == 0
    assert b"no CPU fallback" in L.orbfe_last_error() or b"HIP" in L.orbfe_last_error()


def test_product_sources_never_touch_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and smoke() may use oracle/ (it is the checker, not the product)."""
    pkg = os.path.join(ROOT, "orb_slam2_aruco_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_" not in txt and "liborbfe_oracle" not in txt and "oracle/" not in txt, f


def test_host_scalar_helpers_need_no_device():
    """orbfe_hamming / orbfe_three_maxima / orbfe_epipolar_distance_ok are plain host functions (the reference's static and
    protected ORBmatcher helpers, ORBmatcher.cc:1651-1667, :1605-1646, :139-157): checked against the oracle's restatement and
    known answers here, without a GPU."""
    import numpy as np
    import oracle_lib
    from orb_slam2_aruco_amd import binding
    L = ctypes.CDLL(binding.LIB_PATH)
    O = oracle_lib.lib()
    vp = ctypes.c_void_p
    L.orbfe_three_maxima.argtypes = [vp, ctypes.c_int, vp]; L.orbfe_three_maxima.restype = None
    L.orbfe_epipolar_distance_ok.argtypes = [ctypes.c_float] * 4 + [vp, ctypes.c_float]
    rng = np.random.default_rng(3)
    for trial in range(400):
        kind = trial % 4
        h = (rng.integers(0, 4, 30) if kind == 0 else rng.integers(0, 200, 30) if kind == 1 else
             rng.integers(0, 2, 30) * rng.integers(0, 50, 30) if kind == 2 else np.zeros(30, np.int64)).astype(np.int32)
        if kind == 1 and trial % 8 == 1:
            h[5] = h[17] = h[29] = 150         # ties: the earliest bin wins
        got, want = np.zeros(3, np.int32), np.full(3, -1, np.int32)
        L.orbfe_three_maxima(h.ctypes.data_as(vp), 30, got.ctypes.data_as(vp))
        O.oracle_three_maxima(h.ctypes.data_as(vp), 30, want.ctypes.data_as(vp))
        assert got.tolist() == want.tolist(), (h, got, want)
    # epipolar band: F12 of a pure x translation -> epipolar lines are the rows y2 = y1; 3.84 * sigma2 is the gate on dy^2
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    ok = lambda y2, s2: L.orbfe_epipolar_distance_ok(10.0, 20.0, 55.0, y2, F.ctypes.data_as(vp), s2)
    assert ok(20.0, 1.0) == 1 and ok(21.9, 1.0) == 1 and ok(22.0, 1.0) == 0 and ok(23.0, 1.44) == 0 and ok(22.3, 1.44) == 1
    assert L.orbfe_epipolar_distance_ok(1.0, 2.0, 3.0, 4.0, np.zeros(9, np.float32).ctypes.data_as(vp), 1.0) == 0   # den == 0


def test_every_environment_switch_of_the_library_is_in_its_own_list():
    """bench.py decides from orbfe_pipeline_env_defaults() whether an ORBFE_* variable makes its line a diagnostic; the list is kept
    next to the getenv calls, and this test keeps the two from drifting apart."""
    import ctypes as C
    from orb_slam2_aruco_amd import binding
    L = binding.load()
    L.orbfe_pipeline_env_defaults.restype = C.c_char_p
    listed = {kv.split("=")[0] for kv in L.orbfe_pipeline_env_defaults().decode().split(";") if kv}
    read = set()
    csrc = os.path.join(ROOT, "orb_slam2_aruco_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            txt = open(os.path.join(csrc, f)).read()
            read |= set(re.findall(r'(?:getenv|env_int|env_or|pick)\((?:[^"()]*,\s*)?"(ORBFE_[A-Z0-9_]+)"', txt))
    assert read, "no getenv calls found: the pattern is stale"
    assert read <= listed, sorted(read - listed)
