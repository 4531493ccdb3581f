"""Buffer-layout stress (run ON the GPU box): host frames with row strides larger than the width (and not multiples of 4 / 64), batches
of 1 .. 40 frames, device batches with odd pitches -- results must equal the tightly packed single-frame results.
    python tools/stress_layout.py [n_cases] [seed]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from orb_slam2_aruco_amd import binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    cols, rows = int(rng.integers(200, 900)), int(rng.integers(160, 700))
    if rows > cols: cols, rows = rows, cols
    B = int(rng.integers(1, 12))
    frames = np.stack([synth.scene(rows, cols, int(rng.integers(1, 10 ** 6)), "ARUCO", 2, side_range=(30, max(31, min(rows, cols) // 4)))[0] for _ in range(B)])
    pad = int(rng.integers(0, 70))
    wide = np.zeros((B, rows, cols + pad), np.uint8); wide[:] = rng.integers(0, 256, wide.shape, dtype=np.uint8)
    wide[:, :, :cols] = frames
    why = []
    try:
        ex = binding.ORBextractor(800, 1.2, 6, 20, 7); det = binding.MarkerDetector("ARUCO")
        ref = [ex(frames[f].copy()) for f in range(B)]
        refm = [det.detect(frames[f].copy()) for f in range(B)]
        for f in range(B):                         # strided single frames (views into the padded array)
            k, d = ex(wide[f, :, :cols])
            if not (np.array_equal(k, ref[f][0]) and np.array_equal(d, ref[f][1])): why.append("strided extract frame %d pad %d" % (f, pad))
            m = det.detect(wide[f, :, :cols])
            if not np.array_equal(m, refm[f]): why.append("strided detect frame %d pad %d" % (f, pad))
        got = ex.extract_batch(frames)
        for f in range(B):
            if not (np.array_equal(got[f][0], ref[f][0]) and np.array_equal(got[f][1], ref[f][1])): why.append("batch extract frame %d of %d" % (f, B))
        gm = det.detect_batch(frames)
        for f in range(B):
            if not np.array_equal(gm[f], refm[f]): why.append("batch detect frame %d of %d" % (f, B))
        # device-pointer batch with the padded (odd) pitch
        import torch
        dev = torch.device("cuda:0")
        pitch = cols + pad
        d_img = torch.from_numpy(wide).to(dev)
        cap = ex.capacity
        d_k = torch.zeros(B * cap * binding.KP_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        ex.extract_batch_device(d_img.data_ptr(), B, rows * pitch, rows, cols, pitch, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(), 0)
        mcap = 32
        d_m = torch.zeros(B * mcap * binding.MARKER_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
        det.detect_batch_device(d_img.data_ptr(), B, rows * pitch, rows, cols, pitch, d_m.data_ptr(), mcap, d_nm.data_ptr(), 0)
        torch.cuda.synchronize()
        nk = d_n.cpu().numpy(); kk = d_k.cpu().numpy().view(binding.KP_DTYPE).reshape(B, cap); dd = d_d.cpu().numpy().reshape(B, cap, 32)
        nm = d_nm.cpu().numpy(); mm = d_m.cpu().numpy().view(binding.MARKER_DTYPE).reshape(B, mcap)
        for f in range(B):
            if not (nk[f] == len(ref[f][0]) and np.array_equal(kk[f, :nk[f]], ref[f][0]) and np.array_equal(dd[f, :nk[f]], ref[f][1])):
                why.append("device extract frame %d pitch %d" % (f, pitch))
            if not (nm[f] == len(refm[f]) and np.array_equal(mm[f, :nm[f]]["id"], refm[f]["id"]) and np.array_equal(mm[f, :nm[f]]["corners"], refm[f]["corners"])):
                why.append("device detect frame %d pitch %d" % (f, pitch))
    except Exception as e:
        why.append("exception %r" % (e,))
    if why:
        bad += 1
        print("case %d %dx%d B %d pad %d: %s" % (case, cols, rows, B, pad, why[:3]))
print("%d cases, %d mismatches" % (n, bad))
