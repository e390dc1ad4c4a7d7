"""The exchange step behind the C ABI on real RCCL (world size 1 is what a 1-GPU box can show): librccl is resolved at run time -- the
copy PyTorch has already loaded is reused --, a communicator is created from a unique id, and reduce / all-reduce / all-gather run on the
context's exchange stream (its tail stream in deferred-tail mode), after the kernels that produce their inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_comm_world_of_one_reduce_and_gather_follow_the_step(dfx):
    from deepfactors_amd import _lib, synth
    from deepfactors_amd.dist import NormalEquations, PairGraph
    L = _lib.lib()
    ctx = dfx.Context(0)
    uid = (C.c_ubyte * 128)()
    _lib.check(L.dfx_comm_get_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(L.dfx_comm_create(ctx.handle, uid, 0, 1, C.byref(comm)))
    assert L.dfx_comm_rank(comm) == 0 and L.dfx_comm_world(comm) == 1
    first, count = C.c_int(), C.c_int()
    _lib.check(L.dfx_shard_range(1024, 0, 1, C.byref(first), C.byref(count)))
    assert (first.value, count.value) == (0, 1024)
    w, h, cs, P = 160, 120, 32, 6
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    plist, keep = [], []
    for k in range(P):
        p = synth.make_pair(w, h, cs, seed=40 + k, device="cuda")
        keep.append(p)
        plist.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"]))
    arr = al.make_pairs(plist)
    graph = PairGraph.chain(P)
    neq = NormalEquations(graph, cs, torch.device("cuda", 0))
    isz = dfx.item_size(12 + cs)
    items = torch.zeros(P * isz, dtype=torch.uint8, device="cuda")
    allitems = torch.zeros(P * isz, dtype=torch.uint8, device="cuda")
    for tail in (None, torch.cuda.Stream()):
        ctx.set_tail_stream(tail)
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        want = None
        _lib.check(L.dfx_graph_reduce_async(ctx.handle, comm, neq.native_handle(ctx), C.c_void_p(neq.buf.data_ptr()), 0))
        _lib.check(L.dfx_comm_reduce_f32_async(ctx.handle, comm, C.c_void_p(neq.buf.data_ptr()), neq.buf.numel(), -1))
        _lib.check(L.dfx_items_all_gather_async(ctx.handle, comm, C.c_void_p(items.data_ptr()), P * isz, C.c_void_p(allitems.data_ptr())))
        ctx.tail_join(); ctx.sync()
        # world of one: the sums are the rank's own system, the gathered items its own items -- and both were enqueued behind the step
        assert np.array_equal(allitems.cpu().numpy(), items.cpu().numpy())
        chk = NormalEquations(graph, cs, torch.device("cuda", 0))
        ctx.set_tail_stream(None)
        chk.assemble_native(ctx, items, 0, P)
        ctx.sync()
        assert np.array_equal(chk.buf.cpu().numpy(), neq.buf.cpu().numpy())
        its = al.items_from_bytes(allitems.cpu().numpy(), cs)
        assert all(it.inliers > 0.5 * w * h for it in its)
    with pytest.raises(dfx.DfxError):
        _lib.check(L.dfx_comm_reduce_f32_async(ctx.handle, comm, C.c_void_p(neq.buf.data_ptr()), neq.buf.numel(), 3))
    # keyframe replication (ncclBroadcast on real RCCL, world of one: the root's bytes stay; a root outside the world is refused), ordered on
    # the context's stream: a step enqueued behind it reads the broadcast buffer
    kfimg = keep[0]["img0"].clone()
    _lib.check(L.dfx_comm_broadcast_async(ctx.handle, comm, C.c_void_p(kfimg.data_ptr()), kfimg.numel() * 4, 0))
    ctx.sync()
    assert torch.equal(kfimg, keep[0]["img0"])
    with pytest.raises(dfx.DfxError):
        _lib.check(L.dfx_comm_broadcast_async(ctx.handle, comm, C.c_void_p(kfimg.data_ptr()), kfimg.numel() * 4, 1))
    L.dfx_comm_destroy(comm)


def test_rccl_abi_preflight_on_the_real_library():
    """tests/cpp/rccl_abi_check: the hand-declared RCCL subset of dfx_comm.cpp against the real header (static_asserts, at build time) and, here,
    against the real libraries at run time -- the system's librccl and the copy PyTorch ships (the one dfx_comm.cpp finds already loaded in a
    torch process): every entry point resolves, and a world of one runs all-reduce / reduce / all-gather / broadcast through the product's own
    typedefs (id by value, enumerator values, argument order) with the data intact."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "rccl_abi_check")
    assert os.path.exists(exe), "tests/cpp/rccl_abi_check not built: run __graft_entry__.build()"
    libs = [None]
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        libs.append(cand)
    for lib in libs:
        out = subprocess.run([exe] + ([lib] if lib else []), capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
        assert out.returncode == 0, (lib, out.stdout + out.stderr)
        assert "rccl_abi_check OK" in out.stdout and "world-1 collectives" in out.stdout, out.stdout
