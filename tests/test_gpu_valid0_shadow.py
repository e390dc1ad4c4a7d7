"""valid0 maps owned through the library (dfx_img_alloc) carry a 1-bit-per-pixel shadow ("known to hold 1.0"): the SfM step reads
8 bytes per 64-pixel chunk of it instead of the map.  The observable contract is unchanged (dense_sfm.h:161: valid0(x, y) = 1 where a
pixel is an inlier, never cleared) -- these tests pin it on the shadow variant of the kernel: same sums as with a foreign (torch) map,
same map content as the oracle writes, shadow bits always a subset of the pixels that hold 1.0, fills / uploads keep it consistent."""
import numpy as np
import pytest
import torch

from helpers import assert_item_close

pytestmark = pytest.mark.gpu


def _pair(w, h, cs, seed, **kw):
    from deepfactors_amd import synth
    p = synth.make_pair(w, h, cs, seed=seed, device="cpu", **kw)
    return synth.to_numpy(p), synth.to_device(p, "cuda")


def _step(al, n, g, pose1, valid0):
    return al.RunStep(n["pose0"], pose1, n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, valid0, g["prx_jac"], g["grad1"])


@pytest.mark.parametrize("w,h,cs", [(160, 120, 32), (100, 77, 32), (640, 480, 32), (128, 96, 64), (64, 48, 16)])
def test_library_owned_valid0_matches_oracle_and_foreign_map(dfx, oracle, w, h, cs):
    n, g = _pair(w, h, cs, seed=0xDF02 + w)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    vld = ctx.alloc_image(w, h)                       # zero-filled, library-owned
    assert vld.valid0_shadow() is None                # no shadow before the first use as valid0
    foreign = torch.zeros_like(g["img0"])
    a = _step(al, n, g, pose1, foreign)               # reading variant
    b = _step(al, n, g, pose1, vld)                   # shadow variant, first step: writes 1.0 at every inlier, finalize rebuilds the bits
    assert np.array_equal(a.raw, b.raw), "the sums must not depend on where the valid map lives"
    valid_ref = np.zeros_like(n["img0"])
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=valid_ref, accum_f64=True)
    assert_item_close(b, ref, w, h)
    v = vld.download()
    assert np.array_equal(v, foreign.cpu().numpy())
    assert int((v != valid_ref).sum()) <= max(1, int(1e-5 * w * h))
    sh = vld.valid0_shadow()
    assert sh is not None and np.array_equal(sh, v == 1.0), "after a step the shadow describes the map exactly"
    # steady state: nothing written, bits unchanged, sums bit-identical
    c = _step(al, n, g, pose1, vld)
    assert np.array_equal(b.raw, c.raw) and np.array_equal(vld.download(), v) and np.array_equal(vld.valid0_shadow(), sh)


def test_newly_exposed_pixels_are_written_and_learned(dfx):
    """A pose update exposes pixels that were out of view: they get their 1.0, earlier ones keep theirs (never cleared), and the shadow
    follows (union of the inlier sets)."""
    w, h, cs = 320, 240, 32
    n, g = _pair(w, h, cs, seed=21)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    vld = ctx.alloc_image(w, h)
    far = n["pose1"].copy(); far[4] += 0.6              # large sideways motion: a wide band leaves the view
    f1, f2 = torch.zeros_like(g["img0"]), torch.zeros_like(g["img0"])
    r1 = _step(al, n, g, far, vld); _step(al, n, g, far, f1)
    v1 = vld.download()
    assert np.array_equal(v1, f1.cpu().numpy()) and int((v1 == 1).sum()) == r1.inliers < 0.95 * w * h
    r2 = _step(al, n, g, n["pose1"], vld); _step(al, n, g, n["pose1"], f2)
    v2 = vld.download()
    union = np.maximum(f1.cpu().numpy(), f2.cpu().numpy())
    assert np.array_equal(v2, union) and int((v2 == 1).sum()) > r1.inliers
    assert np.array_equal(vld.valid0_shadow(), v2 == 1.0)
    assert r2.inliers == int((f2.cpu().numpy() == 1).sum())


def test_fill_and_upload_keep_the_shadow_consistent(dfx):
    w, h, cs = 160, 120, 32
    n, g = _pair(w, h, cs, seed=4)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    # the keyframe build order of the reference (mapper.cpp:937): fill with 1.0 BEFORE the first step -> the shadow is born all ones and
    # the first step writes nothing
    vld = ctx.alloc_image(w, h).fill(1.0)
    a = _step(al, n, g, n["pose1"], vld)
    assert vld.valid0_shadow().all() and (vld.download() == 1.0).all()
    # a marker fill: the bits are forgotten, inliers get 1.0 again, the marker survives elsewhere (never cleared)
    vld.fill(7.0)
    assert not vld.valid0_shadow().any()
    b = _step(al, n, g, n["pose1"], vld)
    v = vld.download()
    assert np.array_equal(a.raw, b.raw)
    assert int((v == 1.0).sum()) == a.inliers and int((v == 7.0).sum()) == v.size - a.inliers
    assert np.array_equal(vld.valid0_shadow(), v == 1.0)
    # an upload of arbitrary content (some ones among zeros): bits forgotten, then relearned from the map by the next step that writes
    pat = np.zeros((h, w), np.float32); pat[::3, ::5] = 1.0
    vld.upload(pat)
    assert not vld.valid0_shadow().any()
    _step(al, n, g, n["pose1"], vld)
    v = vld.download()
    assert ((v == 1.0) >= (pat == 1.0)).all() and int((v == 1.0).sum()) >= a.inliers
    assert np.array_equal(vld.valid0_shadow(), v == 1.0)


def test_batch_with_shared_and_mixed_maps(dfx, oracle):
    """One launch: two pairs share a keyframe's library-owned map (concurrent 1.0 stores, both rebuild the shadow with the same values),
    and a second launch mixes a library-owned with a foreign map (the whole batch then takes the reading variant) -- same items either way."""
    w, h, cs = 192, 128, 32
    n, g = _pair(w, h, cs, seed=9)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    p1 = n["pose1"].copy(); p1[4] += 0.02
    p2 = n["pose1"].copy(); p2[5] -= 0.03
    shared = ctx.alloc_image(w, h)
    mk = lambda pose1, v: dict(pose0=n["pose0"], pose1=pose1, cam=n["cam"], img0=g["img0"], img1=g["img1"], dpt0=g["dpt0"],  # noqa: E731
                               prx0_jac=g["prx_jac"], grad1=g["grad1"], valid0=v)
    items = al.RunStepBatch(al.make_pairs([mk(p1, shared), mk(p2, shared)]))
    fa, fb = torch.zeros_like(g["img0"]), torch.zeros_like(g["img0"])
    want = al.RunStepBatch(al.make_pairs([mk(p1, fa), mk(p2, fb)]))
    for a, b in zip(items, want):
        assert np.array_equal(a.raw, b.raw)
    v = shared.download()
    assert np.array_equal(v, np.maximum(fa.cpu().numpy(), fb.cpu().numpy()))
    assert np.array_equal(shared.valid0_shadow(), v == 1.0)
    other = ctx.alloc_image(w, h)
    fc = torch.zeros_like(g["img0"])
    mixed = al.RunStepBatch(al.make_pairs([mk(p1, other), mk(p2, fc)]))
    for a, b in zip(mixed, want):
        assert np.array_equal(a.raw, b.raw)
    assert np.array_equal(other.download(), fa.cpu().numpy()) and np.array_equal(fc.cpu().numpy(), fb.cpu().numpy())
    sh = other.valid0_shadow()          # reading variant: the bits were not rebuilt, and whatever they say is a subset of the ones
    assert sh is None or not (sh & (other.download() != 1.0)).any()


def test_write_before_the_first_step_is_not_forgotten(dfx):
    """Advisor finding (round 3): fill(1.0) then upload(zeros) while NO shadow exists in the process, then the first use as valid0.  The record
    of the image must have learned about the upload: the shadow is created all-clear and the step writes its 1.0s (it once started as
    `every pixel known to hold 1.0` and the map stayed zero)."""
    w, h, cs = 160, 120, 32
    n, g = _pair(w, h, cs, seed=77)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=cs, ctx=ctx)
    vld = ctx.alloc_image(w, h)
    vld.fill(1.0)
    vld.upload(np.zeros((h, w), np.float32))
    assert vld.valid0_shadow() is None
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    foreign = torch.zeros_like(g["img0"])
    a = _step(al, n, g, pose1, foreign)
    b = _step(al, n, g, pose1, vld)
    assert np.array_equal(a.raw, b.raw)
    v = vld.download()
    assert np.array_equal(v, foreign.cpu().numpy()) and v.sum() > 0
    assert np.array_equal(vld.valid0_shadow(), v == 1.0)
