"""GPU parity: Hamming matching kernels vs the CPU oracle (bit-exact: integer work)."""
import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,nt,init", [(1, 1, 256), (7, 300, 256), (1000, 1000, 256), (1030, 999, 2**31 - 1),
                                        (513, 4100, 256), (300, 0, 256)])
def test_knn2_all_pairs(orbfe, oracle, nq, nt, init):
    Q = synth.random_descriptors(nq, 5)
    T = synth.random_descriptors(max(nt, 1), 6)[:nt]
    # duplicates and near-duplicates exercise the tie rules (first candidate wins)
    if nt > 20:
        T[10] = T[3]; T[11] = T[3]; Q[0] = T[3]
    got = orbfe.knn2(Q, T, init)
    want = oracle.knn2(Q, T, init)
    for g, w, nm in zip(got, want, ("best_idx", "best_dist", "second_dist")):
        assert np.array_equal(g, w), nm


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("nq,nt,init", [(1, 1, 256), (33, 65, 256), (1000, 1000, 256), (1030, 999, 2**31 - 1),
                                        (300, 257, 60), (5, 0, 256)])
def test_knn2_both_kernels(orbfe, oracle, path, nq, nt, init):
    """The VALU tile kernel (1) and the matrix-core kernel (2) implement the same rule, ties and `init` included."""
    Q = synth.random_descriptors(nq, nq * 1000 + nt)
    T = synth.random_descriptors(max(nt, 1), nq * 1000 + nt + 1)[:nt]
    if nt > 40:   # duplicates and near-duplicates: ties in best and in second
        T[7] = T[3]; T[40] = T[3]; Q[0] = T[3]
        if nq > 2: Q[2] = T[3] ^ np.uint8(1)
    orbfe.debug_control("knn2_path", path)
    try:
        got = orbfe.knn2(Q, T, init)
    finally:
        orbfe.debug_control("knn2_path", 0)
    want = oracle.knn2(Q, T, init)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_knn2_large_property(orbfe):
    """10k x 10k (config C5): checked through size-independent properties instead of the (slow) oracle."""
    n = 10000
    D = synth.random_descriptors(n, 5)
    bi, bd, sd = orbfe.knn2(D, D, 256)
    assert np.array_equal(bi, np.arange(n))            # every descriptor is its own nearest neighbour ...
    assert np.all(bd == 0) and np.all(sd > 0)          # ... at distance 0, the runner-up is farther
    # runner-up distance = min over t != q of d(q, t), spot-checked with numpy popcounts
    idx = np.array([0, 17, 4999, 9999])
    x = np.unpackbits(D[idx][:, None, :] ^ D[None, :, :], axis=2).sum(2)
    x[np.arange(len(idx)), idx] = 10**6
    assert np.array_equal(sd[idx], x.min(1))


def _frame_pair(orbfe, seed):
    s = synth.stream(480, 640, 2, seed)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    return ex(s[0]), ex(s[1])


@pytest.mark.parametrize("seed,window,ratio,ori", [(1000, 100, 0.9, True), (1234, 100, 0.9, False),
                                                   (77, 30, 0.8, True), (5, 400, 0.95, True)])
def test_search_for_initialization(orbfe, oracle, seed, window, ratio, ori):
    (k1, d1), (k2, d2) = _frame_pair(orbfe, seed)
    m = orbfe.ORBmatcher(ratio, ori)
    n, m12, prev = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, None, window)
    on, om12, oprev = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, None, window, ratio, ori)
    assert n == on and n > 0
    assert np.array_equal(m12, om12)
    assert np.array_equal(prev, oprev)


def test_search_for_initialization_second_call_uses_prev(orbfe, oracle):
    (k1, d1), (k2, d2) = _frame_pair(orbfe, 31)
    m = orbfe.ORBmatcher(0.9, True)
    n, m12, prev = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, None, 100)
    n2, m12b, prev2 = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, prev, 100)
    on2, om12b, oprev2 = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, prev, 100, 0.9, True)
    assert n2 == on2 and np.array_equal(m12b, om12b) and np.array_equal(prev2, oprev2)
