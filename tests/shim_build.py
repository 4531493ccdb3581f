"""Builds tests/shim_driver.cpp + include/shims/ORBmatcher_orbfe.cc against the mock headers of tests/mock_cv/ (test infrastructure)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build(out_dir):
    inc = []
    for d in ("tests/mock_cv", "tests/mock_cv/orbslam", "tests/mock_cv/aruco", "include", "include/shims"):
        inc += ["-I", os.path.join(ROOT, d)]
    obj = os.path.join(out_dir, "shim_matcher.o")
    exe = os.path.join(out_dir, "shim_driver")
    flags = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-Wno-unused-function"]
    subprocess.check_call(flags + inc + ["-c", os.path.join(ROOT, "include", "shims", "ORBmatcher_orbfe.cc"), "-o", obj])
    lib_dir = os.path.join(ROOT, "orb_slam2_aruco_amd")
    subprocess.check_call(flags + inc + [os.path.join(HERE, "shim_driver.cpp"), obj, "-o", exe, "-L", lib_dir, "-lorbfe",
                                          "-Wl,-rpath," + lib_dir])
    return exe
