"""N > 1 path on CPU: world_size-2 `gloo` processes shard the pair list, assemble their normal-equation blocks and
all-reduce them; the result must equal the single-process assembly of all pairs.  Per-pair items come from the
oracle here (test infrastructure) -- on the GPU box the same code consumes libdfx's device items (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _items(n_pairs, cs, seed=0):
    """Deterministic fake-but-structured items (random SPD-ish packed JtJ, Jtr, residual, inliers)."""
    from deepfactors_amd import item_size
    from deepfactors_amd._lib import item_inliers_offset, item_jtj_len
    NP = 12 + cs
    isz = item_size(NP)
    raw = np.zeros((n_pairs, isz), np.uint8)
    rng = np.random.default_rng(seed)
    for k in range(n_pairs):
        J = rng.standard_normal((50, NP)).astype(np.float32)
        M = J.T @ J
        f = raw[k, : (item_jtj_len(NP) + NP + 1) * 4].view(np.float32)
        f[: item_jtj_len(NP)] = M[np.triu_indices(NP)]
        f[item_jtj_len(NP): item_jtj_len(NP) + NP] = rng.standard_normal(NP).astype(np.float32)
        f[item_jtj_len(NP) + NP] = rng.random()
        raw[k, item_inliers_offset(NP):].view(np.uint64)[0] = 1000 + k
    return raw, isz


def _worker(rank, world, port, n_pairs, cs, out, mode="all_reduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfactors_amd.dist import NormalEquations, shard_range
    raw, isz = _items(n_pairs, cs)
    lo, hi = shard_range(n_pairs, rank, world)
    neq = NormalEquations(n_pairs + 1, cs, "cpu")
    neq.assemble(torch.from_numpy(raw[lo:hi].copy()).reshape(-1), lo, hi - lo, isz)
    if mode == "all_reduce":
        neq.all_reduce(dist)
    else:
        neq.reduce(dist, root=0)   # bench.py's exchange step: the sum lands on the rank that solves
    if rank == 0:
        out["H"] = neq.H.clone()
        out["g"] = neq.g.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce"])
@pytest.mark.parametrize("n_pairs", [5, 8])
def test_sharded_normal_equations_match_single_process(n_pairs, mode):
    from deepfactors_amd.dist import NormalEquations, shard_range
    cs = 32
    raw, isz = _items(n_pairs, cs)
    ref = NormalEquations(n_pairs + 1, cs, "cpu")
    ref.assemble(torch.from_numpy(raw.copy()).reshape(-1), 0, n_pairs, isz)

    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_pairs, cs, out, mode), nprocs=2, join=True)
    assert torch.allclose(out["H"], ref.H, rtol=0, atol=1e-5 * float(ref.H.abs().max()))
    assert torch.allclose(out["g"], ref.g, rtol=0, atol=1e-5 * float(ref.g.abs().max()))
    # shards tile the pair list
    spans = [shard_range(n_pairs, r, 2) for r in range(2)]
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n_pairs


def test_block_layout_matches_photometric_factor_slicing():
    """G11,G12,G13,G22,G23,G33 / g1,g2,g3 of photometric_factor.cpp:135-161 land in the right frame blocks."""
    from deepfactors_amd.dist import NormalEquations
    cs = 32
    raw, isz = _items(1, cs, seed=3)
    NP = 12 + cs
    f = raw[0, : (NP * (NP + 1) // 2 + NP) * 4].view(np.float32)
    M = np.zeros((NP, NP), np.float32)
    M[np.triu_indices(NP)] = f[: NP * (NP + 1) // 2]
    M = M + np.triu(M, 1).T
    gv = f[NP * (NP + 1) // 2:]
    neq = NormalEquations(2, cs, "cpu")
    neq.assemble(torch.from_numpy(raw.copy()).reshape(-1), 0, 1, isz)
    H, g = neq.H.numpy(), neq.g.numpy()
    assert np.array_equal(H[0, 0][:6, :6], M[:6, :6])          # G11 (pose0, pose0)
    assert np.array_equal(H[0, 0][:6, 6:], M[:6, 12:])         # G13 (pose0, code0)
    assert np.array_equal(H[0, 0][6:, 6:], M[12:, 12:])        # G33
    assert np.array_equal(H[0, 1][:6, :6], M[:6, 6:12])        # G12 (pose0, pose1) -> off-diagonal block
    assert np.array_equal(H[0, 1][6:, :6], M[12:, 6:12])       # G23^T (code0, pose1)
    assert np.array_equal(H[1, 0][:6, :6], M[6:12, 6:12])      # G22 on frame 1's diagonal
    assert np.array_equal(g[0][:6], gv[:6]) and np.array_equal(g[0][6:], gv[12:]) and np.array_equal(g[1][:6], gv[6:12])
    D = neq.dense().numpy()
    assert np.allclose(D, D.T)
