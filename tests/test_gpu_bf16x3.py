"""DFX_MFMA_BF16X3 -- the exact three-way bf16 split of the step kernel (k_sfm_step<..., B3>, k_sfm_finalize_b3) against the fp64
oracle and against the default fp32 chain: same inliers, same valid0 writes, sums equal at fp32 accuracy (the split drops terms
below 2^-26 of a product; tests/test_zspace_model.py is the executable specification of its tiles and of the finalize scatter)."""
import numpy as np
import pytest
import torch

from helpers import assert_blocks_below, assert_item_close

pytestmark = pytest.mark.gpu


def _pair(w, h, cs, seed, **kw):
    from deepfactors_amd import synth
    p = synth.make_pair(w, h, cs, seed=seed, device="cpu", **kw)
    return synth.to_numpy(p), synth.to_device(p, "cuda")


def _ctx(dfx, mode):
    from deepfactors_amd import _lib
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_BF16X3 if mode == "bf16x3" else _lib.DFX_MFMA_F32_CHAIN)
    return ctx


@pytest.mark.parametrize("w,h,cs", [(160, 120, 32), (100, 77, 32), (64, 48, 16), (96, 64, 64), (640, 480, 32), (101, 67, 16), (320, 240, 64)])
def test_bf16x3_step_matches_the_oracle_and_the_fp32_chain(dfx, oracle, w, h, cs):
    n, g = _pair(w, h, cs, seed=0xB3 + w)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    res = {}
    for mode in ("f32", "bf16x3"):
        al = dfx.SfmAligner(code_size=cs, ctx=_ctx(dfx, mode))
        valid = torch.zeros_like(g["img0"])
        res[mode] = (al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, valid, g["prx_jac"], g["grad1"]), valid)
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], accum_f64=True)
    got, chain = res["bf16x3"][0], res["f32"][0]
    assert_item_close(got, ref, w, h, what=f"bf16x3 sfm_step {w}x{h} cs={cs}")
    assert got.inliers == chain.inliers and torch.equal(res["bf16x3"][1], res["f32"][1])
    # fp32 quality, not merely inside the 1e-4 tolerance: EVERY block (pose-pose, pose-code, code-code, both gradients), each entry at
    # its own Cauchy-Schwarz scale, below 1e-5 in both modes, and the split no worse than 4x the chain block by block
    e_split = assert_blocks_below(got, ref, 1e-5, what=f"bf16x3 {w}x{h} cs={cs}")
    e_chain = assert_blocks_below(chain, ref, 1e-5, what=f"f32 chain {w}x{h} cs={cs}")
    for k in e_split:
        assert e_split[k]["cs"] < 4 * e_chain[k]["cs"] + 1e-6, (k, e_split[k], e_chain[k])


def test_bf16x3_batch_static_and_dynamic(dfx, oracle):
    """A 24-pair batch (static partition, bit-reproducible) and a 130-pair batch on the dynamic item queues, both in the split mode."""
    from deepfactors_amd import _lib
    cs = 32
    for (w, h, npairs, dyn) in ((128, 96, 24, False), (128, 96, 130, True)):
        host, dev = [], []
        for k in range(5):
            nk, gk = _pair(w, h, cs, seed=0x3B00 + k, motion_scale=0.5 + 0.2 * k)
            host.append(nk); dev.append(gk)
        rng = np.random.default_rng(5)
        idx = [(int(rng.integers(0, 5)), int(rng.integers(0, 5))) for _ in range(npairs)]
        ctx = _ctx(dfx, "bf16x3")
        ctx.set_schedule(_lib.DFX_SCHEDULE_DYNAMIC if dyn else _lib.DFX_SCHEDULE_STATIC)
        al = dfx.SfmAligner(code_size=cs, ctx=ctx)
        plist = []
        for (i, j) in idx:
            pose1 = host[j]["pose1"].copy(); pose1[4] += 0.004 * j
            plist.append(dict(pose0=host[i]["pose0"], pose1=pose1, cam=host[i]["cam"], img0=dev[i]["img0"], img1=dev[j]["img1"], dpt0=dev[i]["dpt0"],
                              prx0_jac=dev[i]["prx_jac"], grad1=dev[j]["grad1"]))
        arr = al.make_pairs(plist)
        a = al.RunStepBatch(arr)
        b = al.RunStepBatch(arr)
        assert ctx.last_schedule_dynamic() == dyn
        for q in range(0, npairs, 5 if npairs < 64 else 19):
            i, j = idx[q]
            pose1 = host[j]["pose1"].copy(); pose1[4] += 0.004 * j
            ref = oracle.sfm_step(host[i]["pose0"], pose1, host[i]["cam"], host[i]["img0"], host[j]["img1"], host[i]["dpt0"], host[i]["prx_jac"], host[j]["grad1"])
            assert_item_close(a[q], ref, w, h, what=f"bf16x3 batch pair {q} dyn={dyn}")
        if not dyn:
            for x, y in zip(a, b):
                assert np.array_equal(x.raw, y.raw)           # the static partition stays bit-reproducible in this mode


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_bf16x3_depth_aligner(dfx, oracle, cs):
    w, h = 128, 96
    n, g = _pair(w, h, cs, seed=51)
    tgt = n["dpt0"] * 1.05 + 0.02
    code = n["code"] * 0.5
    al = dfx.DepthAligner(code_size=cs, ctx=_ctx(dfx, "bf16x3"))
    got = al.RunStep(code, torch.from_numpy(tgt).cuda(), g["prx_orig"], g["prx_jac"], 2.0)
    ref = oracle.depth_aligner_step(code, tgt, n["prx_orig"], n["prx_jac"], 2.0)
    assert got.inliers == ref.inliers == w * h
    assert_item_close(got, ref, w, h, what="bf16x3 depth_aligner")
