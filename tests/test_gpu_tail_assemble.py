"""dfx_sfm_step_batch_assemble_async: the reduction tail of a batched launch as ONE kernel (k_sfm_tail_b3: a workgroup per pair sums the
partials and writes the item; the last of a node's pairs to arrive gathers the node's diagonal block and gradient).  Same bits as the
two calls dfx_sfm_step_batch_async + dfx_graph_assemble_async -- for whole graphs and shards of the pair list (the blocks no local pair
writes are zeroed), graphs with isolated nodes, launch after launch (the arrival counters rewind), pairs of several image sizes, both
schedules, the deferred-tail mode, and the launches that have no such tail kernel (fp32 chain, a single pair)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plist(dfx, al, ctx, sizes, cs, seed0):
    from deepfactors_amd import synth
    keep, plist = [], []
    for k, (w, h) in enumerate(sizes):
        p = synth.make_pair(w, h, cs, seed=seed0 + k, device="cuda", motion_scale=0.4 + 0.1 * (k % 4))
        p["valid0"] = ctx.alloc_image(w, h)
        keep.append(p)
        plist.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"],
                          prx0_jac=p["prx_jac"], grad1=p["grad1"], valid0=p["valid0"]))
    return plist, keep


def _both(al, arr, graph, cs, n, first, dirty=7.0):
    """(items, system) of the two-call path and of the one-call path for the pairs `arr` = graph pairs [first, first + n)"""
    from deepfactors_amd.dist import NormalEquations
    import deepfactors_amd as dfx
    isz = dfx.item_size(12 + cs)
    out = []
    for fused in (False, True):
        items = torch.full((n * isz,), 0, dtype=torch.uint8, device="cuda")
        neq = NormalEquations(graph, cs, "cuda")
        neq.buf.fill_(dirty)   # every entry must be overwritten
        al.RunStepBatchAssembleAsync(arr, items, neq, first, fused=fused)
        al.ctx.tail_join()
        al.ctx.sync()
        out.append((items.cpu().numpy(), neq.buf.cpu().numpy(), neq))
    return out


@pytest.mark.parametrize("cs", [32, 16, 64])
def test_one_call_equals_two_calls_on_graphs_and_shards(dfx, cs):
    from deepfactors_amd.dist import PairGraph
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    w, h = 128, 96
    graphs = [
        PairGraph(6, [(0, 1), (0, 2), (1, 0), (1, 2), (2, 4), (3, 2), (0, 4), (4, 3), (2, 1), (5, 0), (0, 5), (0, 3)]),   # node 0: 8 incident pairs
        PairGraph(9, [(0, 1), (0, 2), (1, 0), (1, 2), (2, 4), (3, 2), (0, 4), (4, 3), (2, 1), (6, 0), (0, 6), (0, 3)]),   # nodes 5, 7, 8 isolated
        PairGraph.chain(12),
    ]
    plist, keep = _plist(dfx, al, ctx, [(w, h)] * 12, cs, 0x5100 + cs)
    for gi, graph in enumerate(graphs):
        n = graph.n_pairs
        for first, cnt in ((0, n), (3, 6), (n - 2, 2), (0, 1)):
            arr = al.make_pairs(plist[first:first + cnt])
            (i0, s0, _), (i1, s1, neq) = _both(al, arr, graph, cs, cnt, first)
            assert np.array_equal(i0, i1), (gi, first, cnt)
            assert np.array_equal(s0, s1), (gi, first, cnt, float(np.abs(s0 - s1).max()))
            assert np.abs(s1).max() > 0 and np.isfinite(s1).all()
            if cnt == n:
                M = neq.dense()
                assert torch.allclose(M, M.T)
    # launch after launch on one context: the arrival counters are back at zero every time
    graph = graphs[0]
    arr = al.make_pairs(plist)
    (i0, s0, _), _ = _both(al, arr, graph, cs, 12, 0)
    from deepfactors_amd.dist import NormalEquations
    isz = dfx.item_size(12 + cs)
    items = torch.zeros(12 * isz, dtype=torch.uint8, device="cuda")
    neq = NormalEquations(graph, cs, "cuda")
    snaps = []
    for r in range(6):
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        snaps.append((items.clone(), neq.buf.clone()))
    ctx.sync()
    for it, sy in snaps:
        assert np.array_equal(it.cpu().numpy(), i0) and np.array_equal(sy.cpu().numpy(), s0)


def test_mixed_sizes_both_schedules_and_the_launches_without_a_tail_kernel(dfx):
    from deepfactors_amd import _lib
    from deepfactors_amd.dist import PairGraph
    cs = 32
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    # pyramid levels of four factor sets in one launch, on a graph where the levels of a set share its two nodes
    sizes = [(256, 192), (128, 96), (64, 48)] * 4
    graph = PairGraph(5, [(s, s + 1) for s in range(4) for _ in range(3)])
    plist, keep = _plist(dfx, al, ctx, sizes, cs, 0x5300)
    arr = al.make_pairs(plist)
    (i0, s0, _), (i1, s1, _) = _both(al, arr, graph, cs, 12, 0)
    assert np.array_equal(i0, i1) and np.array_equal(s0, s1)
    # one size, dynamic schedule: the items differ from the static ones by re-association only, and the one-call system is the gather of ITS items
    # (80 pairs: teams of 51 waves = 51 partials per pair -- few enough for the one-workgroup-per-pair tail kernel; the 8-pair batch below
    # has teams of 512 and takes the per-tile finalize + k_graph_assemble)
    from deepfactors_amd.dist import NormalEquations
    isz = dfx.item_size(12 + cs)
    for npr, (w_, h_) in ((80, (128, 96)), (8, (320, 240))):
        plist, keep = _plist(dfx, al, ctx, [(w_, h_)] * npr, cs, 0x5400)
        arr = al.make_pairs(plist)
        graph = PairGraph.chain(npr)
        ctx.set_schedule(_lib.DFX_SCHEDULE_DYNAMIC)
        items = torch.zeros(npr * isz, dtype=torch.uint8, device="cuda")
        neq = NormalEquations(graph, cs, "cuda")
        for _ in range(3):
            al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        ctx.sync()
        assert ctx.last_schedule_dynamic()
        want = NormalEquations(graph, cs, "cuda")
        want.assemble_native(ctx, items, 0, npr)
        ctx.sync()
        assert torch.equal(want.buf, neq.buf)
        ctx.set_schedule(_lib.DFX_SCHEDULE_STATIC)
    # the fp32 chain has no tail kernel: the call runs finalize + assembly as two kernels
    ctx.set_mfma_mode(_lib.DFX_MFMA_F32_CHAIN)
    (i0, s0, _), (i1, s1, _) = _both(al, arr, graph, cs, 8, 0)
    assert ctx.last_mfma_mode() == _lib.DFX_MFMA_F32_CHAIN
    assert np.array_equal(i0, i1) and np.array_equal(s0, s1)
    ctx.set_mfma_mode(_lib.DFX_MFMA_AUTO)
    # a single pair travels by value and keeps the per-tile finalize kernel
    one = al.make_pairs(plist[:1])
    (i0, s0, _), (i1, s1, _) = _both(al, one, graph, cs, 1, 5)
    assert np.array_equal(i0, i1) and np.array_equal(s0, s1)
    # deferred-tail mode: the tail kernel (with the assembly) runs on the second stream
    (i_ref, s_ref, _), _ = _both(al, arr, graph, cs, 8, 0)
    tail = torch.cuda.Stream(device=torch.device("cuda", 0))
    ctx.set_tail_stream(tail)
    snaps = []
    for r in range(4):
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        with torch.cuda.stream(tail):
            snaps.append((items.clone(), neq.buf.clone()))
    ctx.tail_join(); ctx.sync()
    for it, sy in snaps:
        assert np.array_equal(it.cpu().numpy(), i_ref) and np.array_equal(sy.cpu().numpy(), s_ref)
    ctx.set_tail_stream(None)


def test_a_changed_valid0_map_is_rebuilt_by_the_tail_kernel(dfx):
    """The tail kernel also maintains the 1-bit shadow of a library-owned valid0 map: after the first launch on a fresh map the shadow
    equals the map, and the second launch (which reads the shadow) returns the same item."""
    cs = 32
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    w, h = 320, 240
    plist, keep = _plist(dfx, al, ctx, [(w, h)] * 3, cs, 0x5500)
    arr = al.make_pairs(plist)
    first = al.RunStepBatch(arr)
    for k, p in enumerate(keep):
        v = p["valid0"].download()
        assert (v == 1.0).sum() > 0 and np.array_equal(p["valid0"].valid0_shadow(), v == 1.0), k
    again = al.RunStepBatch(arr)
    for a, b in zip(first, again):
        assert np.array_equal(a.raw, b.raw)


def test_step_batch_assemble_rejects_inconsistent_arguments(dfx):
    """dfx_sfm_step_batch_assemble_async fails loudly on a graph of another code size, a pair range outside the graph, a null system --
    and leaves the context usable (the next launch gives the usual result)."""
    import ctypes as C
    from deepfactors_amd import _lib
    from deepfactors_amd.dist import NormalEquations, PairGraph
    cs = 16
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    plist, keep = _plist(dfx, al, ctx, [(96, 64)] * 4, cs, 0x5800)
    arr = al.make_pairs(plist)
    isz = dfx.item_size(12 + cs)
    items = torch.zeros(4 * isz, dtype=torch.uint8, device="cuda")
    graph = PairGraph.chain(4)
    neq = NormalEquations(graph, cs, "cuda")
    al.RunStepBatchAssembleAsync(arr, items, neq, 0)
    ctx.sync()
    want_items, want_sys = items.cpu().numpy().copy(), neq.buf.cpu().numpy().copy()
    other = NormalEquations(graph, 32, "cuda")
    with pytest.raises(ValueError, match="code size"):
        al.RunStepBatchAssembleAsync(arr, items, other, 0)
    p = al._p()
    rc = _lib.lib().dfx_sfm_step_batch_assemble_async(ctx.handle, cs, C.byref(p), arr, 4, C.c_void_p(items.data_ptr()), other.native_handle(ctx), 0,
                                                      C.c_void_p(other.buf.data_ptr()))
    assert rc == _lib.DFX_E_INVALID and "code size" in _lib.lib().dfx_last_error().decode()
    with pytest.raises(dfx.DfxError, match="not inside the graph"):
        al.RunStepBatchAssembleAsync(arr, items, neq, 2)
    rc = _lib.lib().dfx_sfm_step_batch_assemble_async(ctx.handle, cs, C.byref(p), arr, 4, C.c_void_p(items.data_ptr()), neq.native_handle(ctx), 0, None)
    assert rc == _lib.DFX_E_INVALID
    rc = _lib.lib().dfx_sfm_step_batch_assemble_async(ctx.handle, cs, C.byref(p), arr, 4, C.c_void_p(items.data_ptr()), None, 0, C.c_void_p(neq.buf.data_ptr()))
    assert rc == _lib.DFX_E_INVALID
    items.zero_(); neq.buf.fill_(3.0)
    al.RunStepBatchAssembleAsync(arr, items, neq, 0)
    ctx.sync()
    assert np.array_equal(items.cpu().numpy(), want_items) and np.array_equal(neq.buf.cpu().numpy(), want_sys)


def test_arrival_protocol_stress_on_random_graphs(dfx):
    """1000 launches of the assembling tail kernel on 25 random graphs (high-degree hubs, isolated nodes, both directions, shards of the pair
    list), many small pairs per launch so that the pairs' workgroups arrive at their nodes in every order the dispatcher produces: the system
    every launch leaves behind equals, bit for bit, the assembly of the same launch's items by the separate k_graph_assemble kernel (which
    starts after the launch has completed) -- a hand-over that lost an item store would show as a stale block.  The counters rewind launch
    after launch.  (The protocol is release / acquire on the arrival counter since round 5; DFX_TAIL_ORDERED=0 is the relaxed form.)"""
    from deepfactors_amd.dist import NormalEquations, PairGraph
    cs = 16
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    rng = np.random.default_rng(0x7A11)
    n_max = 160
    plist, keep = _plist(dfx, al, ctx, [(64, 48)] * n_max, cs, 0x5900)
    isz = dfx.item_size(12 + cs)
    launches = 0
    for g in range(25):
        n_nodes = int(rng.integers(3, 40))
        n_pairs = int(rng.integers(8, n_max + 1))
        hubs = rng.integers(0, n_nodes, 2)
        pairs = []
        while len(pairs) < n_pairs:
            a = int(hubs[rng.integers(0, 2)]) if rng.random() < 0.5 else int(rng.integers(0, n_nodes - 1))   # the last node stays isolated in half of the graphs
            b = int(rng.integers(0, n_nodes - (g % 2)))
            if a != b:
                pairs.append((a, b))
        graph = PairGraph(n_nodes, pairs)
        first = int(rng.integers(0, n_pairs // 2)) if g % 3 == 0 else 0
        cnt = n_pairs - first - (int(rng.integers(0, 4)) if g % 3 == 0 else 0)
        arr = al.make_pairs(plist[first:first + cnt])
        items = torch.zeros(cnt * isz, dtype=torch.uint8, device="cuda")
        neq, want = NormalEquations(graph, cs, "cuda"), NormalEquations(graph, cs, "cuda")
        ref = None
        for r in range(40):
            neq.buf.fill_(float(r + 1))
            al.RunStepBatchAssembleAsync(arr, items, neq, first)
            launches += 1
            if r % 8 == 0:
                want.buf.fill_(-1.0)
                want.assemble_native(ctx, items, first, cnt)
                ctx.sync()
                assert torch.equal(want.buf, neq.buf), (g, r)
                if ref is None:
                    ref = neq.buf.clone()
            # back-to-back launches in between: compared at the end against the first
        ctx.sync()
        assert torch.equal(ref, neq.buf), g
    assert launches == 1000
