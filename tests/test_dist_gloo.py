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


def _graph(kind):
    from deepfactors_amd.dist import PairGraph
    if kind == "chain5":
        return PairGraph.chain(5)
    if kind == "chain8":
        return PairGraph.chain(8)
    if kind == "window16":
        return PairGraph.all_pairs(16, both_directions=True)       # BASELINE configs[2], both directions as the mapper links them
    return PairGraph.window(64, 16)                                # BASELINE configs[3]: 64 keyframes, 1024 directed pairs


def _worker(rank, world, port, kind, cs, out, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfactors_amd.dist import NormalEquations, shard_range
    graph = _graph(kind)
    n_pairs = graph.n_pairs
    raw, isz = _items(n_pairs, cs)
    lo, hi = shard_range(n_pairs, rank, world)
    neq = NormalEquations(graph, cs, "cpu")
    mine = torch.from_numpy(raw[lo:hi].copy()).reshape(-1)
    if mode == "gather":
        # gather mode (SURVEY 8e option 1): every rank receives all items in pair order and assembles the whole system itself
        allit = NormalEquations.gather_items(dist, mine, n_pairs, isz, world)
        assert allit.numel() == n_pairs * isz and torch.equal(allit, torch.from_numpy(raw).reshape(-1))
        neq.assemble(allit, 0, n_pairs, isz)
    else:
        neq.assemble(mine, lo, hi - lo, isz)
        if mode == "all_reduce":
            neq.all_reduce(dist)
        else:
            neq.reduce(dist, root=0)   # bench.py's exchange step: the sum lands on the rank that solves
    if rank == 0 or mode != "reduce":
        out[f"buf{rank}"] = neq.buf.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce", "gather"])
@pytest.mark.parametrize("kind", ["chain5", "chain8", "window16", "window64"])
def test_sharded_normal_equations_match_single_process(kind, mode):
    from deepfactors_amd.dist import NormalEquations, shard_range
    cs = 32
    graph = _graph(kind)
    n_pairs = graph.n_pairs
    raw, isz = _items(n_pairs, cs)
    ref = NormalEquations(graph, cs, "cpu")
    ref.assemble(torch.from_numpy(raw.copy()).reshape(-1), 0, n_pairs, isz)

    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, cs, out, mode), nprocs=2, join=True)
    if mode == "gather":   # same items, same order, same arithmetic: identical bits on every rank
        assert torch.equal(out["buf0"], ref.buf) and torch.equal(out["buf1"], ref.buf)
    else:
        assert torch.allclose(out["buf0"], ref.buf, rtol=0, atol=1e-5 * float(ref.buf.abs().max()))
        if mode == "all_reduce":
            assert torch.equal(out["buf0"], out["buf1"])
    # shards tile the pair list
    spans = [shard_range(n_pairs, r, 2) for r in range(2)]
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n_pairs


def test_block_layout_matches_photometric_factor_slicing():
    """G11,G12,G13,G22,G23,G33 / g1,g2,g3 of photometric_factor.cpp:135-161 land in the right node blocks, for a pair whose keyframe
    node index is HIGHER than its frame's (the mapper links both directions, mapper.cpp:308-311)."""
    from deepfactors_amd.dist import NormalEquations, PairGraph
    cs = 32
    raw, isz = _items(2, cs, seed=3)
    NP, D = 12 + cs, 6 + cs

    def dense_item(k):
        f = raw[k, : (NP * (NP + 1) // 2 + NP) * 4].view(np.float32)
        M = np.zeros((NP, NP), np.float32)
        M[np.triu_indices(NP)] = f[: NP * (NP + 1) // 2]
        return M + np.triu(M, 1).T, f[NP * (NP + 1) // 2:]

    M0, g0 = dense_item(0)
    M1, g1 = dense_item(1)
    neq = NormalEquations(PairGraph(3, [(0, 1), (2, 1)]), cs, "cpu")
    neq.assemble(torch.from_numpy(raw.copy()).reshape(-1), 0, 2, isz)
    Hd, Ho, g = neq.Hd.numpy(), neq.Ho.numpy(), neq.g.numpy()
    assert np.array_equal(Hd[0][:6, :6], M0[:6, :6])          # G11 (pose0, pose0)
    assert np.array_equal(Hd[0][:6, 6:], M0[:6, 12:])         # G13 (pose0, code0)
    assert np.array_equal(Hd[0][6:, 6:], M0[12:, 12:])        # G33
    assert np.array_equal(Ho[0][:6, :], M0[:6, 6:12])         # G12 (pose0, pose1) -> the pair's off-diagonal block
    assert np.array_equal(Ho[0][6:, :], M0[12:, 6:12])        # G23^T (code0, pose1)
    assert np.array_equal(Ho[1][:6, :], M1[:6, 6:12]) and np.array_equal(Hd[2][6:, 6:], M1[12:, 12:])
    assert np.allclose(Hd[1][:6, :6], M0[6:12, 6:12] + M1[6:12, 6:12], rtol=0, atol=1e-6 * np.abs(M0).max())   # G22 of both pairs on node 1
    assert float(np.abs(Hd[1][6:, :]).max()) == 0.0 and float(np.abs(Hd[1][:, 6:]).max()) == 0.0            # node 1 is nobody's keyframe
    assert np.array_equal(g[0][:6], g0[:6]) and np.array_equal(g[0][6:], g0[12:]) and np.array_equal(g[2][6:], g1[12:])
    assert np.allclose(g[1][:6], g0[6:12] + g1[6:12], atol=1e-6 * np.abs(g0).max())
    Dm = neq.dense().numpy()
    assert np.allclose(Dm, Dm.T)
    # node 1's pose rows couple to node 2's (pose | code) through pair 1 = (keyframe 2 -> frame 1)
    assert np.array_equal(Dm[2 * D:3 * D, D:D + 6].astype(np.float32), Ho[1])


def _pipe_worker(rank, world, port, cs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfactors_amd.dist import NormalEquations, PairGraph, PipelinedReduce, shard_range
    graph = PairGraph.chain(8)
    lo, hi = shard_range(graph.n_pairs, rank, world)
    pipe = PipelinedReduce(dist, [NormalEquations(graph, cs, "cpu") for _ in range(2)], root=0)
    for step in range(5):                       # five independent batches through two buffers
        raw, isz = _items(graph.n_pairs, cs, seed=100 + step)
        neq = pipe.next()
        neq.assemble(torch.from_numpy(raw[lo:hi].copy()).reshape(-1), lo, hi - lo, isz)
        pipe.submit()
        if rank == 0 and step == 3:
            keep3 = neq                          # batch 3's buffer is next written at batch 5: still intact after the loop
    pipe.drain()
    if rank == 0:
        out["last"] = pipe.last().buf.clone()
        out["prev"] = keep3.buf.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_reduce_overlaps_batches_without_mixing_them():
    """bench.py's N > 1 exchange: the reduce of batch k is in flight while batch k + 1 is assembled into the other buffer; every batch's
    sum on the root equals its single-process assembly."""
    from deepfactors_amd.dist import NormalEquations, PairGraph
    cs = 16
    graph = PairGraph.chain(8)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pipe_worker, args=(2, _free_port(), cs, out), nprocs=2, join=True)
    for key, step in (("last", 4), ("prev", 3)):
        raw, isz = _items(graph.n_pairs, cs, seed=100 + step)
        ref = NormalEquations(graph, cs, "cpu")
        ref.assemble(torch.from_numpy(raw.copy()).reshape(-1), 0, graph.n_pairs, isz)
        assert torch.allclose(out[key], ref.buf, rtol=0, atol=1e-5 * float(ref.buf.abs().max())), key
