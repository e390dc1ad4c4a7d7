"""Host-side marshalling helpers that a relinearisation round uses every iteration (no GPU): poses rewritten in place in a make_pairs() array, and the
block-sparse normal equations unpacked into a dense system."""
import ctypes as C

import numpy as np

from deepfactors_amd.aligners import SfmAligner, SfmPair, _se3
from deepfactors_amd.dist import NormalEquations, PairGraph


def test_set_poses_all_equals_per_record_assignment():
    rng = np.random.default_rng(3)
    poses = rng.standard_normal((6, 7)).astype(np.float32)
    i0, i1 = rng.integers(0, 6, 17), rng.integers(0, 6, 17)
    a, b = (SfmPair * 17)(), (SfmPair * 17)()
    for arr in (a, b):                       # some image half that must survive
        for k in range(17):
            arr[k].img0.w, arr[k].img0.h, arr[k].cam.fx = 640 + k, 480, 500.0 + k
    SfmAligner.set_poses_all(a, poses, i0, i1)
    for k in range(17):
        b[k].pose0, b[k].pose1 = _se3(poses[i0[k]]), _se3(poses[i1[k]])
    assert bytes(a) == bytes(b)
    assert C.sizeof(SfmPair) * 17 == len(bytes(a))


def test_dense_from_vectorised_equals_block_loop():
    cs, K = 4, 5
    D = 6 + cs
    pairs = [(i, j) for i in range(K) for j in range(K) if i != j and abs(i - j) <= 2]
    g = PairGraph(K, pairs)
    neq = NormalEquations.__new__(NormalEquations)
    neq.graph, neq.cs, neq.D = g, cs, D
    P = len(pairs)
    rng = np.random.default_rng(5)
    buf = rng.standard_normal(K * D * D + P * D * 6 + K * D).astype(np.float32)
    M, grad = neq.dense_from(buf)
    Hd = buf[:K * D * D].reshape(K, D, D).astype(np.float64)
    Ho = buf[K * D * D:K * D * D + P * D * 6].reshape(P, D, 6).astype(np.float64)
    R = np.zeros((K * D, K * D))
    for k in range(K):
        R[k * D:(k + 1) * D, k * D:(k + 1) * D] = Hd[k]
    for p, (a, c) in enumerate(pairs):
        R[a * D:(a + 1) * D, c * D:c * D + 6] += Ho[p]
        R[c * D:c * D + 6, a * D:(a + 1) * D] += Ho[p].T
    assert np.array_equal(M, R)
    assert np.array_equal(grad, buf[K * D * D + P * D * 6:].astype(np.float64))


def test_gram_dense_unpacks_the_row_major_upper_triangle():
    """The packing dfx_sparse_geometric_gram_batch documents (include/dfx.h): entry (i, j), i <= j, at i NC - i (i - 1) / 2 + (j - i)."""
    from deepfactors_amd.aligners import SparseGeometricFactor
    cs = 16
    nc = 12 + 2 * cs + 1
    A = np.random.default_rng(1).standard_normal((50, nc))
    G = A.T @ A
    packed = np.empty(nc * (nc + 1) // 2)
    for i in range(nc):
        for j in range(i, nc):
            packed[i * nc - i * (i - 1) // 2 + (j - i)] = G[i, j]
    assert np.allclose(SparseGeometricFactor.gram_dense(packed, cs), G, rtol=0, atol=1e-12)
