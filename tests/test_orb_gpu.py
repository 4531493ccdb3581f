"""GPU parity: HIP ORB extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end."""
import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # (rows, cols, nfeatures, nlevels, seed)
    (240, 320, 500, 4, 11),
    (480, 640, 1000, 8, 1),
    (480, 640, 2000, 8, 2),   # the 2*nFeatures initialisation extractor (Tracking.cc:127)
    (360, 636, 700, 6, 5),    # odd width, non-multiple-of-4 levels
    (196, 1641, 500, 5, 6),   # panorama: ten quadtree roots
    (240, 320, 500, 8, 12),   # QVGA with the default 8 levels: level 7 is 89 x 67, one row of FAST cells
    (222, 222, 300, 8, 13),   # level 7 is 62 x 62: the smallest level the reference's cell grid allows
]


def _cmp_kps(a, b, what):
    assert len(a) == len(b), "%s: count %d vs %d" % (what, len(a), len(b))
    for fld in ("x", "y", "response"):
        assert np.array_equal(a[fld], b[fld]), "%s: field %s differs" % (what, fld)


@pytest.mark.parametrize("rows,cols,nf,nl,seed", CASES)
def test_stages_and_end_to_end(orbfe, oracle, rows, cols, nf, nl, seed):
    img, _ = synth.scene(rows, cols, seed, n_markers=3, side_range=(40, 90))
    ex = orbfe.ORBextractor(nf, 1.2, nl, 20, 7)
    ora = oracle.OrbOracle(nf, 1.2, nl, 20, 7)
    kps, desc = ex(img)
    okps, odesc = ora.extract(img)
    for l in range(nl):
        assert np.array_equal(ex.level_image(0, l), ora.level_image(l)), "pyramid level %d" % l
        _cmp_kps(ex.level_keypoints(0, l, 0), ora.level_keypoints(l, 0), "FAST candidates level %d" % l)
        _cmp_kps(ex.level_keypoints(0, l, 1), ora.level_keypoints(l, 1), "quadtree level %d" % l)
        if len(ora.level_keypoints(l, 1)):
            assert np.array_equal(ex.level_image(0, l, True), ora.level_image(l, True)), "blur level %d" % l
    assert len(kps) == len(okps)
    for fld in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[fld], okps[fld]), fld
    # orientation: shared integer moments + shared fastAtan2 -> expected bit-exact; contract is 1e-4
    assert np.allclose(kps["angle"], okps["angle"], atol=1e-4)
    assert np.array_equal(kps["angle"], okps["angle"])
    assert np.array_equal(desc, odesc)


def test_batch_equals_single(orbfe):
    imgs = np.stack([synth.scene(240, 320, 100 + i, n_markers=2, side_range=(40, 70))[0] for i in range(5)])
    ex = orbfe.ORBextractor(500, 1.2, 4, 20, 7)
    batch = ex.extract_batch(imgs)
    for i in range(5):
        k, d = ex(imgs[i])
        assert np.array_equal(k, batch[i][0]) and np.array_equal(d, batch[i][1])


def test_empty_image(orbfe):
    ex = orbfe.ORBextractor(500, 1.2, 4, 20, 7)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)


def test_flat_image_no_keypoints(orbfe, oracle):
    img = np.full((240, 320), 128, np.uint8)
    k, d = orbfe.ORBextractor(500, 1.2, 4, 20, 7)(img)
    ok, od = oracle.OrbOracle(500, 1.2, 4, 20, 7).extract(img)
    assert len(k) == 0 and len(ok) == 0


def test_quadtree_fast_path_and_general_kernel_agree(orbfe, oracle):
    """The count-pyramid quadtree and the general (key-moving) kernel are two implementations of DistributeOctTree."""
    img, _ = synth.scene(480, 640, 21, n_markers=3)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    k1, d1 = ex(img)
    assert not any(ex.quadtree_fell_back(0, l) for l in range(8))   # textured scene: the fast path handles every level
    ex.force_general_quadtree(True)
    k2, d2 = ex(img)
    ok, od = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert np.array_equal(k1, ok) and np.array_equal(d1, od)


def test_quadtree_fallback_to_general_kernel(orbfe, oracle):
    """With a shallow count pyramid the tree wants to split leaf cells: those levels are redone by the general kernel."""
    img, _ = synth.scene(480, 640, 22, n_markers=2)
    ok, od = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    for depth in (2, 3, 4):
        ex.set_pyramid_depth(depth)
        k, d = ex(img)
        if depth <= 3:   # 1000 features need depth-4 nodes at level 0: a depth <= 3 pyramid must give up there
            assert ex.quadtree_fell_back(0, 0), depth
        assert np.array_equal(k, ok) and np.array_equal(d, od), depth


def test_clustered_keypoints(orbfe, oracle):
    """All texture in one small patch (the reference stops splitting as soon as a pass leaves the node count unchanged)."""
    img = np.full((480, 640), 120, np.uint8)
    rng = np.random.default_rng(3)
    img[200:248, 300:348] = rng.integers(0, 256, (48, 48), dtype=np.uint8)
    k, d = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)(img)
    ok, od = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    assert len(ok) > 10
    assert np.array_equal(k, ok) and np.array_equal(d, od)


@pytest.mark.parametrize("rows,cols,nf,nl", [(480, 640, 1000, 8), (1080, 1920, 4000, 12)])
def test_dense_noise_thousands_of_candidates_a_level(orbfe, oracle, rows, cols, nf, nl):
    """Uniform noise: every cell is full of FAST corners.  The quadtree's count pass (a thread per candidate, round 6) then takes a
    level's candidates in several windows of its LDS list and, at 1920 x 1080, the cells of level 0 in several rounds."""
    img = np.random.default_rng(rows).integers(0, 256, (rows, cols), dtype=np.uint8)
    ex = orbfe.ORBextractor(nf, 1.2, nl, 20, 7)
    k, d = ex(img)
    assert len(ex.level_keypoints(0, 0, 0)) > 4000      # the candidates of level 0 (before the quadtree)
    ok, od = oracle.OrbOracle(nf, 1.2, nl, 20, 7).extract(img)
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    ex.force_general_quadtree(True)
    k2, d2 = ex(img)
    assert np.array_equal(k2, ok) and np.array_equal(d2, od)


@pytest.mark.parametrize("rows,cols,nf,nl,seed,dic,K", [
    (720, 1280, 2000, 8, 3, "ARUCO_MIP_25h7", 6),      # BASELINE configs[2]
    (1080, 1920, 4000, 12, 4, "ARUCO_MIP_36h12", 6),   # BASELINE configs[4] (extraction part)
    (540, 960, 1000, 8, 8, "ARUCO", 4),                # what mono_cvcam really feeds (mono_cvcam.cc:124)
])
def test_baseline_configs_end_to_end(orbfe, oracle, rows, cols, nf, nl, seed, dic, K):
    img, _ = synth.scene(rows, cols, seed, dic, K)
    ex = orbfe.ORBextractor(nf, 1.2, nl, 20, 7)
    kps, desc = ex(img)
    okps, odesc = oracle.OrbOracle(nf, 1.2, nl, 20, 7).extract(img)
    assert len(kps) == len(okps) and len(kps) >= nf * 0.9
    assert np.array_equal(kps, okps)
    assert np.array_equal(desc, odesc)
    # a second call on the same handle (scratch reuse) and a 2-frame batch give the same answer
    batch = ex.extract_batch(np.stack([img, img[::-1].copy()]))
    assert np.array_equal(batch[0][0], okps) and np.array_equal(batch[0][1], odesc)


def test_full_size_properties_without_oracle(orbfe):
    """Size-independent properties at the bench batch size (64 frames of 640x480): determinism, level quotas,
    keypoints inside the valid band, descriptors not degenerate."""
    s = synth.stream(480, 640, 8, 4000)
    frames = np.concatenate([s] * 8)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    a = ex.extract_batch(frames)
    b = ex.extract_batch(frames)
    scale = ex.GetScaleFactors()
    quota = ex.features_per_level()
    for i, ((k, d), (k2, d2)) in enumerate(zip(a, b)):
        assert np.array_equal(k, k2) and np.array_equal(d, d2)                 # idempotent
        assert np.array_equal(k, a[i % 8][0]) and np.array_equal(d, a[i % 8][1])  # same frame -> same result anywhere in the batch
        assert np.all(np.diff(k["octave"]) >= 0)                                # levels concatenated in ascending order
        for l in range(8):
            m = k["octave"] == l
            assert m.sum() <= quota[l] + 2                                      # ORBextractor.cc:730 overshoot bound
            x, y = k["x"][m] / scale[l], k["y"][m] / scale[l]
            assert x.min() >= 18.99 and y.min() >= 18.99                        # EDGE_THRESHOLD band
        assert 100 < np.unpackbits(d, axis=1).sum(1).mean() < 156               # bits are balanced


def test_gaussian_tap_variants(orbfe, oracle):
    """The GaussianBlur taps are a stated, switchable choice (OpenCV release dependent, orbfe_extractor_set_gaussian_taps): both
    variants are bit-exact against the oracle, blurred levels and descriptors."""
    img, _ = synth.scene(480, 640, 11, n_markers=3, side_range=(40, 90))
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    ora = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    out = []
    for mode in (1, 0):
        ex.set_gaussian_taps(mode); ora.set_gaussian_taps(mode)
        kps, desc = ex(img)
        okps, odesc = ora.extract(img)
        for l in range(8):
            if len(ora.level_keypoints(l, 1)):
                assert np.array_equal(ex.level_image(0, l, True), ora.level_image(l, True)), "blur level %d, taps mode %d" % (l, mode)
        assert np.array_equal(kps, okps) and np.array_equal(desc, odesc)
        out.append(desc)
    assert not np.array_equal(out[0], out[1])
    with pytest.raises(orbfe.OrbfeError):
        ex.set_gaussian_taps(2)


def test_hipgraph_replay_of_the_host_pointer_call():
    """ORBFE_GRAPH=1: from the third call with one frame size on, orbfe_extract replays its upload, launches and result copies as a
    hipGraph captured inside the library.  The end-to-end cases (several frames of one size through one handle) in a process with the
    switch set: replayed calls must give the oracle's keypoints and descriptors like direct ones."""
    import os, subprocess, sys
    env = dict(os.environ, ORBFE_GRAPH="1", ORBFE_GRAPH_VERBOSE="1")
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; from orb_slam2_aruco_amd import binding, synth; import oracle_lib\n"
            "s = synth.stream(480, 640, 6, 4242)\n"
            "ex = binding.ORBextractor(1000, 1.2, 8, 20, 7); ora = oracle_lib.OrbOracle(1000, 1.2, 8, 20, 7)\n"
            "for img in list(s) + [s[0], s[3]]:\n"
            "    k, d = ex(img); ok, od = ora.extract(img)\n"
            "    assert np.array_equal(k, ok) and np.array_equal(d, od)\n"
            "print('replays ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "replays ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    assert "hipGraph capture ok" in r.stderr, r.stderr[-1000:]


def test_phase_lock_between_handles_changes_no_result(orbfe, oracle):
    """orbfe_extractor_follow: a handle's batches start behind a stage of another handle's latest batch (the engine sets of the
    batched pipeline).  Ordering only: results are those of the free-running handles; arguments outside the ranges are refused."""
    imgs = [synth.scene(480, 640, 300 + i, "ARUCO", 2)[0] for i in range(4)]
    free = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    want = [free(im) for im in imgs]
    a, b = orbfe.ORBextractor(1000, 1.2, 8, 20, 7), orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    for stage in (1, 2, 3, 4, 21, 20, 14):   # (+ 10 g: a second gate in front of the follower's FAST)
        a.follow(b, stage); b.follow(a, stage)
        for i, im in enumerate(imgs):
            kps, desc = (a if i % 2 == 0 else b)(im)
            assert np.array_equal(kps, want[i][0]) and np.array_equal(desc, want[i][1])
    a.follow(None, 0); b.follow(a, 0)
    kps, desc = a(imgs[0])
    assert np.array_equal(kps, want[0][0])
    with pytest.raises(orbfe.OrbfeError):
        a.follow(b, 5)
    with pytest.raises(orbfe.OrbfeError):
        a.follow(b, 52)
    L = a.L
    assert L.orbfe_extractor_stage_wait(a.h, 0, None) != 0 and L.orbfe_extractor_stage_wait(a.h, 5, None) != 0
    assert L.orbfe_extractor_stage_wait(a.h, 2, None) == 0      # the null stream waits for a's latest quadtree: harmless
    assert np.array_equal(b(imgs[1])[0], want[1][0])
