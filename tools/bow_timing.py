"""Timing of the vocabulary transform at ORBvoc size (k = 10, L = 6: 1 111 110 nodes, 10^6 words) over a 300-frame batch of
extracted descriptors.  Synthetic complete tree (random descriptors: worst case for cache locality)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orb_slam2_aruco_amd import binding, synth

k, Lv = 10, int(os.environ.get("BOW_L", "6"))
rng = np.random.default_rng(0)
counts = [k ** l for l in range(1, Lv + 1)]
nn = sum(counts)
parent = np.zeros(nn, np.int32); is_leaf = np.zeros(nn, np.uint8)
start, prev_start = 0, -1
for l, c in enumerate(counts):
    ids = np.arange(start, start + c)                       # node ids are ids + 1 (root = 0)
    parent[start:start + c] = 0 if l == 0 else (prev_start + 1 + (ids - start) // k)
    if l == Lv - 1:
        is_leaf[start:start + c] = 1
    prev_start, start = start, start + c
desc = rng.integers(0, 256, (nn, 32), dtype=np.uint8)
weight = np.where(is_leaf > 0, rng.uniform(0.5, 9.0, nn), 0.0)
t0 = time.perf_counter()
voc = binding.ORBVocabulary.from_arrays(k, Lv, 0, 0, parent, is_leaf, desc, weight)
print("vocabulary: %s, upload %.2f s" % (voc.info(), time.perf_counter() - t0))

B = 300
frames = synth.stream(480, 640, B, 1000)
ex = binding.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.capacity
dev = torch.device("cuda:0")
imgs = torch.from_numpy(frames).to(dev)
kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); dsc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
n = torch.zeros(B, dtype=torch.int32, device=dev)
ex.extract_batch_device(imgs.data_ptr(), B, 480 * 640, 480, 640, 640, kps.data_ptr(), dsc.data_ptr(), cap, n.data_ptr(), 0)
i32 = lambda m: torch.zeros(m, dtype=torch.int32, device=dev)
f64 = lambda m: torch.zeros(m, dtype=torch.float64, device=dev)
word, node, wgt, bw, bv, nb = i32(B * cap), i32(B * cap), f64(B * cap), i32(B * cap), f64(B * cap), i32(B)
fn, fo, ff, nf = i32(B * cap), i32(B * (cap + 1)), i32(B * cap), i32(B)
L = voc.L
def run(vectors=True):
    z = None
    rc = L.orbfe_vocabulary_transform_batch_device(voc.h, dsc.data_ptr(), n.data_ptr(), cap, B, 4, word.data_ptr(), node.data_ptr(),
        wgt.data_ptr(), bw.data_ptr() if vectors else z, bv.data_ptr() if vectors else z, nb.data_ptr() if vectors else z,
        fn.data_ptr() if vectors else z, fo.data_ptr() if vectors else z, ff.data_ptr() if vectors else z,
        nf.data_ptr() if vectors else z, None)
    assert rc == 0
for vectors in (False, True):
    for _ in range(3):
        run(vectors)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run(vectors)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%s: %.1f us per %d-frame batch (%d features) -> %.0f frames/s" %
          ("descend + vectors" if vectors else "descend only", dt * 1e6, B, int(n.sum()), B / dt))
print("words per frame %.0f, nodes per frame %.0f" % (nb.float().mean().item(), nf.float().mean().item()))
