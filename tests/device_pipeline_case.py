"""Device-resident Frame glue case run by tests/test_match_gpu.py::test_undistort_keypoints_batch_device in its own
process (torch first, then the HIP library): extract_batch_device -> undistort_keypoints_batch_device -> matching over
the undistorted image bounds, against the oracle."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch, ctypes as C
torch.cuda.init()
from orb_slam2_aruco_amd import binding as orbfe, synth
import oracle_lib as oracle

TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
TUM1_DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)

s = synth.stream(480, 640, 3, 1000)
ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
cap = ex.capacity
dev = torch.device("cuda:0")
imgs = torch.from_numpy(s).to(dev)
kps = torch.zeros(3 * cap * orbfe.KP_DTYPE.itemsize, dtype=torch.uint8, device=dev)
desc = torch.zeros(3 * cap * 32, dtype=torch.uint8, device=dev)
n = torch.zeros(3, dtype=torch.int32, device=dev)
ex.extract_batch_device(imgs.data_ptr(), 3, 480 * 640, 480, 640, 640, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), 0)
L = ex.L
d = np.ascontiguousarray(TUM1_DIST)
rc = L.orbfe_undistort_keypoints_batch_device(kps.data_ptr(), n.data_ptr(), cap, 3, TUM1_K.ctypes.data_as(C.c_void_p),
                                              d.ctypes.data_as(C.c_void_p), 5, kps.data_ptr(), None)
assert rc == 0
torch.cuda.synchronize()
nh = n.cpu().numpy()
kh = kps.cpu().numpy().view(orbfe.KP_DTYPE).reshape(3, cap)
dh = desc.cpu().numpy().reshape(3, cap, 32)
o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
frames = []
for f in range(3):
    k, dd = o.extract(s[f])
    xy = oracle.undistort_points(np.stack([k["x"], k["y"]], 1), TUM1_K, TUM1_DIST)
    k = k.copy(); k["x"] = xy[:, 0]; k["y"] = xy[:, 1]
    assert nh[f] == len(k)
    for fld in ("x", "y", "size", "octave"):
        assert np.array_equal(kh[f, :nh[f]][fld], k[fld]), (f, fld)
    assert np.array_equal(dh[f, :nh[f]], dd)
    frames.append((k, dd))
bounds = orbfe.ComputeImageBounds(640, 480, TUM1_K, TUM1_DIST)
(k1, d1), (k2, d2) = frames[0], frames[1]
wn, wm, wp = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, None, 100, 0.9, True, bounds)
gn, gm, gp = orbfe.ORBmatcher(0.9, True).SearchForInitialization(kh[0, :nh[0]], dh[0, :nh[0]], kh[1, :nh[1]], dh[1, :nh[1]],
                                                                 640, 480, None, 100, bounds=bounds)
assert gn == wn and np.array_equal(gm, wm) and np.array_equal(gp, wp)

# ---- ComputeBoW over the batch on the device (Frame.cc:348-355): per-frame vectors against the oracle
import voc_cases as vc
voc = vc.make(10, 4, 41, irregular=False)
ovoc = oracle.VocabularyOracle.from_arrays(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
gvoc = orbfe.ORBVocabulary.from_arrays(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
i32 = lambda *shape: torch.zeros(*shape, dtype=torch.int32, device=dev)
word, node, weight = i32(3 * cap), i32(3 * cap), torch.zeros(3 * cap, dtype=torch.float64, device=dev)
bw, bv, nb = i32(3 * cap), torch.zeros(3 * cap, dtype=torch.float64, device=dev), i32(3)
fn, fo, ff, nf = i32(3 * cap), i32(3 * (cap + 1)), i32(3 * cap), i32(3)
rc = L.orbfe_vocabulary_transform_batch_device(gvoc.h, desc.data_ptr(), n.data_ptr(), cap, 3, 4, word.data_ptr(), node.data_ptr(),
                                               weight.data_ptr(), bw.data_ptr(), bv.data_ptr(), nb.data_ptr(), fn.data_ptr(),
                                               fo.data_ptr(), ff.data_ptr(), nf.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
for f in range(3):
    want = ovoc.transform(frames[f][1], 4)
    k = int(nb[f]); j = int(nf[f])
    assert np.array_equal(bw[f * cap:f * cap + k].cpu().numpy().view(np.uint32), want["bow"][0])
    assert np.array_equal(bv[f * cap:f * cap + k].cpu().numpy().view(np.uint64), want["bow"][1].view(np.uint64))
    assert np.array_equal(fn[f * cap:f * cap + j].cpu().numpy().view(np.uint32), want["fv"][0])
    off = fo[f * (cap + 1):f * (cap + 1) + j + 1].cpu().numpy()
    assert np.array_equal(off, want["fv"][1])
    assert np.array_equal(ff[f * cap:f * cap + off[-1]].cpu().numpy().view(np.uint32), want["fv"][2])
print("bow ok")

# ---- SearchByBoW / SearchForTriangulation over explicit frame pairs of the batch (device pointers end to end)
pairs = [(0, 1), (2, 0)]
p1 = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device=dev); p2 = torch.tensor([b for _, b in pairs], dtype=torch.int32, device=dev)
m12, m21, nmm = i32(2 * cap), i32(2 * cap), i32(2)
rc = L.orbfe_search_by_bow_batch_device(kps.data_ptr(), desc.data_ptr(), None, n.data_ptr(), fn.data_ptr(), fo.data_ptr(), ff.data_ptr(),
                                        nf.data_ptr(), cap, p1.data_ptr(), p2.data_ptr(), 2, 0, 0.7, 1, 50, np.float32(30 / 360.0),
                                        m12.data_ptr(), m21.data_ptr(), nmm.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
fvs = [ovoc.transform(frames[f][1], 4)["fv"] for f in range(3)]
for p, (a, b) in enumerate(pairs):
    (ka, da), (kb, db) = frames[a], frames[b]
    wn, w12, w21 = oracle.search_by_bow(ka, da, fvs[a], kb, db, fvs[b], None, None, 0.7, True, 50, 30 / 360.0)
    assert int(nmm[p]) == wn and wn > 0
    assert np.array_equal(m12[p * cap:p * cap + len(ka)].cpu().numpy(), w12) and np.array_equal(m21[p * cap:p * cap + len(kb)].cpu().numpy(), w21)
F12 = torch.tensor(np.tile(np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32), 2), device=dev)
epi = torch.tensor(np.array([320, 240, -1e4, 240], np.float32), device=dev)
sfh = np.array([1.2 ** i for i in range(8)], np.float32); sgh = (sfh * sfh).astype(np.float32)
rc = L.orbfe_search_for_triangulation_batch_device(kps.data_ptr(), desc.data_ptr(), None, n.data_ptr(), fn.data_ptr(), fo.data_ptr(), ff.data_ptr(),
                                                   nf.data_ptr(), cap, p1.data_ptr(), p2.data_ptr(), 2, F12.data_ptr(), epi.data_ptr(),
                                                   sfh.ctypes.data_as(C.c_void_p), sgh.ctypes.data_as(C.c_void_p), 8, 1, m12.data_ptr(),
                                                   m21.data_ptr(), nmm.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
for p, (a, b) in enumerate(pairs):
    (ka, da), (kb, db) = frames[a], frames[b]
    ep = (320.0, 240.0) if p == 0 else (-1e4, 240.0)
    wn, w12 = oracle.search_for_triangulation(ka, da, fvs[a], kb, db, fvs[b], np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32), ep, sfh, sgh)
    assert int(nmm[p]) == wn and np.array_equal(m12[p * cap:p * cap + len(ka)].cpu().numpy(), w12)
print("pairs ok")

# ---- the projection searches over a resident batch (orbfe_search_by_projection_batch_device): frame f is searched with queries
# built from frame (f + 1) % 3's keypoints; every mode against the per-frame host-pointer entry points and the oracle
rng = np.random.default_rng(3)
QC = 700
qs = np.zeros((3, QC), orbfe.WINDOW_QUERY_DTYPE); qd = np.zeros((3, QC, 32), np.uint8); nq = np.zeros(3, np.int32)
qobs = np.zeros((3, QC), np.uint8); qang = np.zeros((3, QC), np.float32); tk = np.zeros((3, cap), np.uint8)
for f in range(3):
    ks, ds = frames[(f + 1) % 3]
    sel = rng.permutation(len(ks))[:QC - 50 * f]
    nq[f] = len(sel)
    qs[f, :len(sel)]["x"] = ks["x"][sel] + rng.normal(0, 1.5, len(sel)).astype(np.float32)
    qs[f, :len(sel)]["y"] = ks["y"][sel] + rng.normal(0, 1.5, len(sel)).astype(np.float32)
    qs[f, :len(sel)]["r"] = (6.0 * 1.2 ** ks["octave"][sel]).astype(np.float32)
    qs[f, :len(sel)]["min_level"] = ks["octave"][sel] - 1
    qs[f, :len(sel)]["max_level"] = ks["octave"][sel] + (f % 2)
    qd[f, :len(sel)] = ds[sel]
    qang[f, :len(sel)] = ks["angle"][sel]
    qobs[f, :len(sel)] = rng.random(len(sel)) < 0.7
    tk[f, :nh[f]] = rng.random(nh[f]) < 0.1
tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
d_qs, d_qd, d_nq, d_qobs, d_qang = tdev(qs), tdev(qd), torch.from_numpy(nq).to(dev), tdev(qobs), tdev(qang)
outs = [i32(3 * QC) for _ in range(6)]
d_mc, d_nm3 = i32(3 * cap), i32(3)
kun = kps  # the (undistorted) keypoint records of the batch, still on the device
bnd_p = bounds.ctypes.data_as(C.c_void_p)
for mode in (0, 1, 2):
    d_tk = tdev(tk)
    for attempt in range(2):
        rc = L.orbfe_search_by_projection_batch_device(kun.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, 3, 640, 480, bnd_p, d_qs.data_ptr(),
                                                       d_qd.data_ptr(), d_nq.data_ptr(), QC, d_tk.data_ptr(), d_qobs.data_ptr(), d_qang.data_ptr(),
                                                       mode, 100, np.float32(0.8), np.float32(1.0 / 30), 1, *[o.data_ptr() for o in outs[:5]],
                                                       outs[5].data_ptr(), d_mc.data_ptr(), d_nm3.data_ptr(), None)
        assert rc == 0, L.orbfe_last_error()
        ovf = C.c_int32(0)
        assert L.orbfe_search_by_projection_batch_status(None, C.byref(ovf)) == 0
        if not ovf.value:
            break
        d_tk = tdev(tk)
    assert ovf.value == 0
    torch.cuda.synchronize()
    got = [o.cpu().numpy().reshape(3, QC) for o in outs]
    gmc = d_mc.cpu().numpy().reshape(3, cap); gnm = d_nm3.cpu().numpy(); gtk = d_tk.cpu().numpy().reshape(3, cap)
    for f in range(3):
        kf, df = kh[f, :nh[f]], dh[f, :nh[f]]
        q, m = qs[f, :nq[f]], int(nq[f])
        if mode < 2:
            want = orbfe.search_by_projection(kf, df, 640, 480, q, qd[f, :m], tk[f, :nh[f]], mode, 100, 0.8, bounds=bounds, q_observed=qobs[f, :m])
            ora = oracle.search_by_projection(kf, df, 640, 480, q, qd[f, :m], tk[f, :nh[f]], mode, 100, 0.8, bounds=bounds, q_observed=qobs[f, :m])
            for j, name in enumerate(("best_idx", "best_dist", "best_level", "second_dist", "second_level")):
                assert np.array_equal(got[j][f, :m], want[name]) and np.array_equal(want[name], ora[name]), (mode, f, name)
            if mode == 1:
                assert gnm[f] == want["nmatches"] == ora["nmatches"] and np.array_equal(got[5][f, :m], want["match"])
                assert np.array_equal(gtk[f, :nh[f]], want["taken"]) and np.array_equal(want["match"], ora["match"])
        else:
            wn, wm = orbfe.search_by_projection_best(kf, df, 640, 480, q, qang[f, :m], qd[f, :m], 100, 1.0 / 30, q_blocks=qobs[f, :m],
                                                     taken=tk[f, :nh[f]], bounds=bounds)
            assert gnm[f] == wn and wn > 5 and np.array_equal(gmc[f, :nh[f]], wm), (f, gnm[f], wn)
print("projection batch ok")

# ---- Fuse over several resident keyframes (orbfe_fuse_search_batch_device; LocalMapping::SearchInNeighbors, LocalMapping.cc:850-858):
# the map points seen by frame 0 fused into the three frames under three poses, against the per-keyframe entry point and the oracle
sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
for l in range(1, 8):
    sf[l] = np.float32(sf[l - 1] * np.float32(1.2))
isg = (1.0 / (sf * sf)).astype(np.float32)
logsf = np.float32(np.log(np.float32(1.2)))
k0, d0 = kh[0, :nh[0]], dh[0, :nh[0]]
nmp = len(k0)
z = rng.uniform(2.0, 6.0, nmp).astype(np.float32)
x3 = np.stack([(k0["x"] - TUM1_K[2]) / TUM1_K[0] * z, (k0["y"] - TUM1_K[3]) / TUM1_K[1] * z, z], 1).astype(np.float32)
dist3 = np.linalg.norm(x3, axis=1).astype(np.float32)
max_d = (dist3 * sf[k0["octave"]]).astype(np.float32)
min_d = (max_d / sf[7]).astype(np.float32)
nrm = (x3 / dist3[:, None]).astype(np.float32)
Tcws = np.zeros((3, 12), np.float32); Ows = np.zeros((3, 3), np.float32)
for k in range(3):
    T = np.eye(3, 4, dtype=np.float32); T[:, 3] = [0.02 * k, -0.01 * k, 0.01 * k]
    Tcws[k] = T.reshape(-1); Ows[k] = -T[:, 3]
fvalid = (rng.random((3, nmp)) < 0.9).astype(np.uint8)
d_x3, d_min, d_max, d_nrm, d_mpd, d_fv = tdev(x3), tdev(min_d), tdev(max_d), tdev(nrm), tdev(d0), tdev(fvalid)
d_bi, d_bd = i32(3 * nmp), i32(3 * nmp)
cp = lambda a: a.ctypes.data_as(C.c_void_p)
for chi2 in (5.99, 0.0):
    for attempt in range(2):
        rc = L.orbfe_fuse_search_batch_device(kun.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, 3, 640, 480, bnd_p, d_x3.data_ptr(), d_fv.data_ptr(),
                                              d_min.data_ptr(), d_max.data_ptr(), d_nrm.data_ptr(), d_mpd.data_ptr(), nmp, cp(Tcws), cp(Ows),
                                              cp(TUM1_K), cp(sf), cp(isg), 8, logsf, np.float32(3.0), chi2, d_bi.data_ptr(), d_bd.data_ptr(), None)
        assert rc == 0, L.orbfe_last_error()
        ovf = C.c_int32(0)
        assert L.orbfe_search_by_projection_batch_status(None, C.byref(ovf)) == 0
        if not ovf.value:
            break
    assert ovf.value == 0
    torch.cuda.synchronize()
    gbi, gbd = d_bi.cpu().numpy().reshape(3, nmp), d_bd.cpu().numpy().reshape(3, nmp)
    for k in range(3):
        kf, df = kh[k, :nh[k]], dh[k, :nh[k]]
        want = oracle.fuse_search(kf, df, 640, 480, x3, fvalid[k], min_d, max_d, nrm, d0, Tcws[k].reshape(3, 4), Ows[k], TUM1_K, sf, isg, logsf, 3.0, chi2,
                                  bounds=bounds)
        host = orbfe.fuse_search(kf, df, 640, 480, x3, fvalid[k], min_d, max_d, nrm, d0, Tcws[k].reshape(3, 4), Ows[k], TUM1_K, sf, isg, logsf, 3.0, chi2,
                                 bounds=bounds)
        assert np.array_equal(gbi[k], want[0]) and np.array_equal(gbd[k], want[1]), (chi2, k)
        assert np.array_equal(host[0], want[0]) and np.array_equal(host[1], want[1])
        if k == 0:
            assert (gbd[0] == 0).sum() > 0.5 * nmp      # frame 0 under its own pose finds its own keypoints
print("fuse batch ok")

# ---- detector batch on device pointers + marker poses (MarkerDetector::detect with camera parameters, Frame.cc:142)
import pose_cases as pc
det = orbfe.MarkerDetector("ARUCO")
mcap = 32
mk = torch.zeros(3 * mcap * 36, dtype=torch.uint8, device=dev); nmk = i32(3)
poses = torch.zeros(3 * mcap * 56, dtype=torch.uint8, device=dev)
det.detect_batch_device(imgs.data_ptr(), 3, 480 * 640, 480, 640, 640, mk.data_ptr(), mcap, nmk.data_ptr(), 0)
Kc = orbfe.camera_resize(pc.K4, (1280, 720), (640, 480)); Dc = np.ascontiguousarray(pc.DIST)
rc = L.orbfe_marker_poses_batch_device(mk.data_ptr(), nmk.data_ptr(), mcap, 3, np.float32(0.187), Kc.ctypes.data_as(C.c_void_p),
                                       Dc.ctypes.data_as(C.c_void_p), 5, poses.data_ptr(), None)
assert rc == 0, L.orbfe_last_error()
torch.cuda.synchronize()
assert det.batch_status() == (0, 0)
mkh = mk.cpu().numpy().view(orbfe.MARKER_DTYPE).reshape(3, mcap); ph = poses.cpu().numpy().view(orbfe.POSE_DTYPE).reshape(3, mcap)
odet = oracle.ArucoOracle("ARUCO")
for f in range(3):
    want = odet.detect(s[f])
    assert int(nmk[f]) == len(want) > 0 and np.array_equal(mkh[f, :len(want)]["id"], want["id"])
    for j, w in enumerate(want):
        r1, t1, r2, t2, err = oracle.marker_pose(w["corners"], 0.187, Kc, Dc)
        assert np.allclose(ph[f, j]["rvec"], r1, rtol=1e-5, atol=1e-6) and np.allclose(ph[f, j]["tvec"], t1, rtol=1e-5, atol=1e-6)
print("markers ok")
print("ok")
