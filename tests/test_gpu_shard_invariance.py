""""fixed pair -> GPU map and fixed in-kernel reduction tree so 1/2/4/8-GPU results are bit-identical per pair".  A pair's sums are
bit-reproducible for a launch SHAPE (workgroups per pair), and the automatic shape depends on the batch size -- so a rank of a sharded job pins
the shape of the WHOLE pair list (dfx_sfm_auto_step_blocks -> dfx_sfm_params.step_blocks).  Here: one keyframe window evaluated as one launch,
as two / four / eight contiguous shards, and pair by pair, all with the shape of the whole list: identical bytes per pair; and without the pin
the shards differ from the whole (which is why the pin exists)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _window(dfx, cs, w, h, n_kf, nbr):
    from deepfactors_amd import synth
    from deepfactors_amd.dist import PairGraph
    kfs = [synth.make_pair(w, h, cs, seed=900 + k, device="cuda") for k in range(n_kf)]
    graph = PairGraph.window(n_kf, nbr)
    plist = []
    for (a, b) in graph.pairs:
        ka, kb = kfs[a], kfs[b]
        pose1 = np.asarray(ka["pose1"], np.float32).copy()
        pose1[4] += 0.002 * (b - a)
        plist.append(dict(pose0=ka["pose0"], pose1=pose1, cam=ka["cam"], img0=ka["img0"], img1=kb["img0"], dpt0=ka["dpt0"], prx0_jac=ka["prx_jac"],
                          grad1=kb["grad1"], valid0=ka["valid0"]))
    return kfs, plist


@pytest.mark.parametrize("mode", ["auto", "f32"])
def test_items_do_not_depend_on_how_the_pair_list_is_split(dfx, mode):
    from deepfactors_amd.dist import shard_range
    cs, w, h = 32, 160, 120
    ctx = dfx.Context(0)
    ctx.set_mfma_mode({"auto": 2, "f32": 0}[mode])
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    kfs, plist = _window(dfx, cs, w, h, 24, 8)            # 24 keyframes x 8 neighbours: 192 directed pairs, 8 per keyframe
    n = len(plist)
    isz = dfx.item_size(12 + cs)
    whole = al.AutoStepBlocks(w, h, n)
    al.SetStepThreadsBlocks(256, whole)
    ref = torch.zeros(n * isz, dtype=torch.uint8, device="cuda")
    al.RunStepBatchAsync(al.make_pairs(plist), ref)
    ctx.sync()
    ref = ref.cpu().numpy().reshape(n, isz)
    for world in (2, 4, 8):
        for rank in range(world):
            lo, hi = shard_range(n, rank, world)
            out = torch.zeros((hi - lo) * isz, dtype=torch.uint8, device="cuda")
            al.RunStepBatchAsync(al.make_pairs(plist[lo:hi]), out)
            ctx.sync()
            assert np.array_equal(out.cpu().numpy().reshape(-1, isz), ref[lo:hi]), (world, rank)
    for p in (0, 17, n - 1):                               # and alone, through the blocking single-pair entry
        q = plist[p]
        it = al.RunStep(q["pose0"], q["pose1"], None, q["cam"], q["img0"], q["img1"], q["dpt0"], None, q["valid0"], q["prx0_jac"], q["grad1"])
        assert np.array_equal(np.frombuffer(it.raw, np.uint8), ref[p]), p
    # without the pin the automatic shape follows the batch size, and so do the last bits
    small = al.AutoStepBlocks(w, h, n // 8)
    if small != whole:
        al.SetStepThreadsBlocks(256, 0)
        lo, hi = shard_range(n, 0, 8)
        out = torch.zeros((hi - lo) * isz, dtype=torch.uint8, device="cuda")
        al.RunStepBatchAsync(al.make_pairs(plist[lo:hi]), out)
        ctx.sync()
        got = out.cpu().numpy().reshape(-1, isz)
        assert not np.array_equal(got, ref[lo:hi])
        a = np.frombuffer(got[0].tobytes(), np.float32, count=100)
        b = np.frombuffer(ref[lo].tobytes(), np.float32, count=100)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_the_shape_of_the_bench_window_is_the_same_on_every_rank(dfx):
    """configs[3]: 64 keyframes, 1024 pairs of 640x480 over 8 ranks -- every rank computes the same answer for the whole list."""
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=32, ctx=ctx)
    b = al.AutoStepBlocks(640, 480, 1024)
    assert 1 <= b <= 65535 and al.AutoStepBlocks(640, 480, 1024) == b
    al.SetStepThreadsBlocks(256, 7)                        # a pinned context value does not answer the question
    assert al.AutoStepBlocks(640, 480, 1024) == b
