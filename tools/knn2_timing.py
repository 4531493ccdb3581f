"""ON THE GPU BOX:  python tools/knn2_timing.py
Both all-pairs kernels alone on the GPU: C5's 10 k x 10 k leg (bench.c5_match_leg: oracle sample + self-match check) and a C2-shaped
batch (299 pairs of 1000 x 1000 descriptors), the two kernels compared with each other."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from orb_slam2_aruco_amd import binding, synth

dev = torch.device("cuda:0")
L = binding.load()
out = {"c5": bench.c5_match_leg(binding, torch, dev, bench.oracle_module())}
npairs, n = 299, 1000
D = np.stack([synth.random_descriptors(n, 100 + i) for i in range(npairs + 1)])          # frame f's descriptors
d_D = torch.from_numpy(D).to(dev)
nn = torch.from_numpy(np.array([n - (i % 7) * 9 for i in range(npairs + 1)], np.int32)).to(dev)   # ragged counts
outs = {}
st = torch.cuda.current_stream(dev)
sp = ctypes.c_void_p(st.cuda_stream)
for name, path in (("valu", 1), ("mfma_i8", 2)):
    binding.debug_control("knn2_path", path)
    o = [torch.zeros(npairs, n, dtype=torch.int32, device=dev) for _ in range(3)]
    call = lambda: binding._check(L, L.orbfe_knn2_batch_device(d_D.data_ptr(), nn.data_ptr(), n * 32, n, d_D[1:].data_ptr(), nn[1:].data_ptr(),
                                                               n * 32, n, npairs, 256, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), sp),
                                  "orbfe_knn2_batch_device")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        call()
    e1.record(st)
    torch.cuda.synchronize(dev)
    outs[name] = [t.cpu().numpy() for t in o]
    out["c2_shape_%s_us" % name] = e0.elapsed_time(e1) * 1000.0 / 20
binding.debug_control("knn2_path", 0)
nq = nn.cpu().numpy()
for p in range(npairs):
    for a, b in zip(outs["valu"], outs["mfma_i8"]):
        assert np.array_equal(a[p, :nq[p]], b[p, :nq[p]]), p
out["c2_shape"] = "%d pairs, %d x %d (ragged), both kernels equal" % (npairs, n, n)
print(json.dumps(out, indent=1))
