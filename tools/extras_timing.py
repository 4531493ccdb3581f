"""Timing of the widened rows' batch kernels on one MI355X (run ON the GPU box):  python tools/extras_timing.py
300-frame 640x480 stream resident in HBM, nFeatures 1000; a k = 10, L = 6 synthetic vocabulary (ORBvoc size).
Prints microseconds per launch, the algorithmic bytes of the launch and the rate they correspond to."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orb_slam2_aruco_amd import binding, synth

dev = torch.device("cuda:0")
L = binding.load()
B = 300
frames = synth.stream(480, 640, B, 1000)
ex = binding.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.capacity
imgs = torch.from_numpy(frames).to(dev)
u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
f64 = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
kps, dsc, n = u8(B * cap * 28), u8(B * cap * 32), i32(B)
ex.extract_batch_device(imgs.data_ptr(), B, 480 * 640, 480, 640, 640, kps.data_ptr(), dsc.data_ptr(), cap, n.data_ptr(), 0)
torch.cuda.synchronize()
nfeat = int(n.sum())


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def report(name, us, nbytes, note=""):
    print("%-44s %9.1f us  %8.1f MB algorithmic  %7.1f GB/s  %s" % (name, us, nbytes / 1e6, nbytes / us / 1e3, note))


# ---- vocabulary transform
k, Lv = 10, 6
rng = np.random.default_rng(0)
counts = [k ** l for l in range(1, Lv + 1)]
nn = sum(counts)
parent = np.zeros(nn, np.int32); is_leaf = np.zeros(nn, np.uint8)
start, prev_start = 0, -1
for l, c in enumerate(counts):
    ids = np.arange(start, start + c)
    parent[start:start + c] = 0 if l == 0 else (prev_start + 1 + (ids - start) // k)
    if l == Lv - 1:
        is_leaf[start:start + c] = 1
    prev_start, start = start, start + c
voc = binding.ORBVocabulary.from_arrays(k, Lv, 0, 0, parent, is_leaf, rng.integers(0, 256, (nn, 32), dtype=np.uint8),
                                        np.where(is_leaf > 0, rng.uniform(0.5, 9.0, nn), 0.0))
word, node, wgt, bw, bv, nb = i32(B * cap), i32(B * cap), f64(B * cap), i32(B * cap), f64(B * cap), i32(B)
fn_, fo, ff, nf = i32(B * cap), i32(B * (cap + 1)), i32(B * cap), i32(B)
def bow():
    assert L.orbfe_vocabulary_transform_batch_device(voc.h, dsc.data_ptr(), n.data_ptr(), cap, B, 4, word.data_ptr(), node.data_ptr(), wgt.data_ptr(),
                                                     bw.data_ptr(), bv.data_ptr(), nb.data_ptr(), fn_.data_ptr(), fo.data_ptr(), ff.data_ptr(),
                                                     nf.data_ptr(), None) == 0
us = timeit(bow)
# per feature: 32 B descriptor + L levels x k children x 32 B of tree (cache resident) + 16 B out; vectors: 2 x (16 B in, 12 B out)
report("vocabulary transform (descend + vectors)", us, nfeat * (32 + Lv * k * 32 + 16 + 56), "tree reads are L2 / MALL hits, not HBM")

# ---- SearchByBoW / SearchForTriangulation over frame pairs (t, t+1)
m12, m21, nm = i32((B - 1) * cap), i32((B - 1) * cap), i32(B)
def sbb():
    assert L.orbfe_search_by_bow_batch_device(kps.data_ptr(), dsc.data_ptr(), None, n.data_ptr(), fn_.data_ptr(), fo.data_ptr(), ff.data_ptr(),
                                              nf.data_ptr(), cap, None, None, B - 1, 0, 0.7, 1, 50, np.float32(30 / 360.0), m12.data_ptr(),
                                              m21.data_ptr(), nm.data_ptr(), None) == 0
us = timeit(sbb)
report("SearchByBoW, 299 frame pairs", us, (B - 1) * 2 * 1000 * (32 + 28 + 8), "matches/pair %.0f" % nm[:B - 1].float().mean().item())
F12 = torch.tensor(np.tile(np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32), B), device=dev)
epi = torch.tensor(np.tile(np.array([1e5, 1e5], np.float32), B), device=dev)
sf = np.array([1.2 ** i for i in range(8)], np.float32); sg = (sf * sf).astype(np.float32)
def tri():
    assert L.orbfe_search_for_triangulation_batch_device(kps.data_ptr(), dsc.data_ptr(), None, n.data_ptr(), fn_.data_ptr(), fo.data_ptr(),
                                                         ff.data_ptr(), nf.data_ptr(), cap, None, None, B - 1, F12.data_ptr(), epi.data_ptr(),
                                                         sf.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p), 8, 1, m12.data_ptr(),
                                                         m21.data_ptr(), nm.data_ptr(), None) == 0
us = timeit(tri)
report("SearchForTriangulation, 299 frame pairs", us, (B - 1) * 2 * 1000 * (32 + 28 + 8), "matches/pair %.0f" % nm[:B - 1].float().mean().item())

# ---- undistortion of the keypoint records, in place
K4 = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
D5 = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
kun = torch.zeros_like(kps)
def und():
    assert L.orbfe_undistort_keypoints_batch_device(kps.data_ptr(), n.data_ptr(), cap, B, K4.ctypes.data_as(C.c_void_p), D5.ctypes.data_as(C.c_void_p),
                                                    5, kun.data_ptr(), None) == 0
us = timeit(und)
report("UndistortKeyPoints (f64, 5 iterations)", us, nfeat * 56, "28 B in + 28 B out per keypoint")

# ---- marker poses
mk = torch.zeros(B * 64 * 36, dtype=torch.uint8, device=dev); nmk = i32(B); poses = torch.zeros(B * 64 * 56, dtype=torch.uint8, device=dev)
det = binding.MarkerDetector("ARUCO")
det.detect_batch_device(imgs.data_ptr(), B, 480 * 640, 480, 640, 640, mk.data_ptr(), 64, nmk.data_ptr(), 0)
torch.cuda.synchronize()
def pose():
    assert L.orbfe_marker_poses_batch_device(mk.data_ptr(), nmk.data_ptr(), 64, B, np.float32(0.187), K4.ctypes.data_as(C.c_void_p),
                                             D5.ctypes.data_as(C.c_void_p), 5, poses.data_ptr(), None) == 0
us = timeit(pose)
report("marker poses (IPPE, f64)", us, int(nmk.sum()) * (36 + 56), "%d markers: latency of one lane's serial f64 chain" % int(nmk.sum()))

# ---- keyframe feature records: 3000 keyframes x 1000 features (one segment per keyframe)
nkf, per = 3000, 1000
tot = nkf * per
g_k, g_d, g_m = u8(tot * 28), u8(tot * 32), torch.zeros(tot, dtype=torch.int64, device=dev)
g_k.random_(0, 256); g_d.random_(0, 256)
seg_off = torch.tensor(8 + 48 + np.arange(nkf, dtype=np.int64) * (48 + 68 * per), dtype=torch.int64, device=dev)
seg_first = torch.tensor(np.arange(nkf + 1, dtype=np.int32) * per, dtype=torch.int32, device=dev)
fileimg = u8(8 + nkf * (48 + 68 * per)); bad = i32(1)
def pack():
    assert L.orbfe_keyframe_features_pack_device(g_k.data_ptr(), g_d.data_ptr(), g_m.data_ptr(), seg_off.data_ptr(), seg_first.data_ptr(), nkf, per,
                                                 fileimg.data_ptr(), None) == 0
def unpack():
    assert L.orbfe_keyframe_features_unpack_device(fileimg.data_ptr(), seg_off.data_ptr(), seg_first.data_ptr(), nkf, per, g_k.data_ptr(),
                                                   g_d.data_ptr(), g_m.data_ptr(), bad.data_ptr(), None) == 0
us = timeit(pack)
report("keyframe records pack (3 M features)", us, tot * 136, "68 B in + 68 B out per feature; HBM peak 8000 GB/s -> frac %.2f" % (tot * 136 / us / 1e3 / 8000))
us = timeit(unpack)
report("keyframe records unpack (3 M features)", us, tot * 136, "frac %.2f" % (tot * 136 / us / 1e3 / 8000))
assert int(bad[0]) == 0
