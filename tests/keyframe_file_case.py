"""Device-resident map file image case (run by tests/test_keyframe_io_gpu.py in its own process: torch first, then the HIP
library): header | features | header | features ... as Map::Save lays keyframes out (Map.cc:236-241, :277-321)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch, ctypes as C
torch.cuda.init()
from orb_slam2_aruco_amd import binding as orbfe, synth
import oracle_lib as oracle

ex = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
frames = [ex.extract(img) for img in synth.stream(480, 640, 3, 1000)]
frames[1] = (frames[1][0][:300], frames[1][1][:300])
rng = np.random.default_rng(3)
blob, seg_off, seg_first, mps = bytearray(rng.integers(0, 256, 8, dtype=np.uint8).tobytes()), [], [0], []
for k, d in frames:
    hdr = np.zeros(48, np.uint8); hdr[:] = rng.integers(0, 256, 48)         # id, timestamp, quaternion, translation
    hdr[44:48] = np.array([len(k)], "<i4").view(np.uint8)                     # N
    blob += hdr.tobytes()
    mp = rng.integers(0, 5000, len(k)).astype(np.uint64)
    mps.append(mp)
    seg_off.append(len(blob)); seg_first.append(seg_first[-1] + len(k))
    blob += oracle.keyframe_features_pack(k, d, mp).tobytes()
total = seg_first[-1]
dev = torch.device("cuda:0")
file_np = np.frombuffer(bytes(blob), np.uint8).copy()
d_file = torch.from_numpy(file_np).to(dev)
d_off = torch.tensor(seg_off, dtype=torch.int64, device=dev); d_first = torch.tensor(seg_first, dtype=torch.int32, device=dev)
d_kps = torch.zeros(total * 28, dtype=torch.uint8, device=dev); d_desc = torch.zeros(total * 32, dtype=torch.uint8, device=dev)
d_mp = torch.zeros(total, dtype=torch.int64, device=dev); d_bad = torch.zeros(1, dtype=torch.int32, device=dev)
L = orbfe.load()
mx = max(len(k) for k, _ in frames)
rc = L.orbfe_keyframe_features_unpack_device(d_file.data_ptr(), d_off.data_ptr(), d_first.data_ptr(), 3, mx, d_kps.data_ptr(),
                                             d_desc.data_ptr(), d_mp.data_ptr(), d_bad.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
assert int(d_bad[0]) == 0
kps = d_kps.cpu().numpy().view(orbfe.KP_DTYPE); desc = d_desc.cpu().numpy().reshape(-1, 32); mp = d_mp.cpu().numpy().view(np.uint64)
for s, (k, d) in enumerate(frames):
    a, b = seg_first[s], seg_first[s + 1]
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(kps[a:b][f], k[f]), (s, f)
    assert np.array_equal(desc[a:b], d) and np.array_equal(mp[a:b], mps[s])
# and back: the feature runs of a zeroed image are rewritten byte for byte, the headers stay untouched
d_out = torch.zeros_like(d_file)
rc = L.orbfe_keyframe_features_pack_device(d_kps.data_ptr(), d_desc.data_ptr(), d_mp.data_ptr(), d_off.data_ptr(), d_first.data_ptr(), 3, mx,
                                           d_out.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
out = d_out.cpu().numpy()
for s, (k, d) in enumerate(frames):
    a = seg_off[s]; b = a + 68 * len(k)
    assert out[a:b].tobytes() == file_np[a:b].tobytes()
    assert not out[a - 48:a].any()
print("ok")
