"""dfx_build_pyramid[_batch_async]: Frame::FillPyramids (core/mapping/frame.h:80-94) / UploadLiveFrame (core/deepfactors.cpp:616-630) as ONE enqueue --
a launch per pyramid level over all frames of the batch, the level read once (Sobel gradient and blur-down to the next level from the same LDS tile).
Against the per-level operators (dfx_gaussian_blur_down, dfx_sobel_gradients: the same bits) and the oracle (pinned to the reference's kernel bodies,
tests/test_oracle_vs_ref.py / ref_vectors_f1.npz: Sobel exact, blur within 1e-6) at even, odd and narrow sizes, pitched buffers, skipped gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _alloc(levels, w, h, ch=None, pad=0):
    out = []
    for i in range(levels):
        ww, hh = w, h
        for _ in range(i):
            ww, hh = ww // 2, hh // 2
        shape = (hh, ww + pad) if ch is None else (hh, ww + pad, ch)
        t = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
        out.append(t[:, :ww] if pad else t)
    return out


# even widths with 8 / 16-byte aligned rows take the row-streaming kernel (k_pyr_rows: strips of 128 columns, segments of >= 4 rows), everything else the
# LDS-tile kernel (k_pyr_level): odd widths, odd pitches; from the first level of at most 160 x 120 pixels on (when at least two levels are left) ONE launch does the rest
# (k_pyr_tail: bands of a frame in LDS, halo rows recomputed): (640, 480, 4) levels 2-3, (320, 240, 3 / 4) levels 1.., (640, 480, 5) and (1280, 960, 6) three levels,
# (600, 440, 4) 150 x 110 and 75 x 55 (odd widths inside the tail), (320, 240, 4, pad 3) unaligned rows inside the tail, (16, 12, 3) and (40, 30, 4) tails of one band
# (a last level of 3 rows), (444, 444, 4) an ODD height (111) in the middle of the tail; (130, 66): a strip with ONE active lane, then an odd level; (2, 2): one pixel pair; (256, 100): strip seams
# at x = 128 and segment seams every 4 rows; (386, 131): odd height, partial last strip, padded rows
@pytest.mark.parametrize("w,h,levels,pad", [(640, 480, 4, 0), (320, 240, 3, 0), (333, 217, 3, 0), (70, 50, 2, 3), (64, 16, 1, 0), (9, 7, 2, 0), (1280, 960, 4, 4),
                                            (256, 100, 3, 0), (130, 66, 2, 0), (2, 2, 1, 0), (128, 4, 2, 0), (386, 131, 2, 2), (644, 484, 3, 0),
                                            (640, 480, 5, 0), (1280, 960, 6, 0), (600, 440, 4, 0), (320, 240, 4, 3),
                                            (16, 12, 3, 0), (40, 30, 4, 0), (444, 444, 4, 0)])
def test_pyramid_build_equals_the_per_level_operators_and_the_oracle(dfx, oracle, w, h, levels, pad):
    rng = np.random.default_rng(w * 7 + h)
    n = 3
    imgs = [rng.random((h, w), dtype=np.float32) for _ in range(n)]
    pyr_i, pyr_g = [], []
    for k in range(n):
        pi, pg = _alloc(levels, w, h, pad=pad), _alloc(levels, w, h, ch=2, pad=pad)
        pi[0].copy_(torch.from_numpy(imgs[k]))
        pyr_i.append(pi); pyr_g.append(pg)
    dfx.BuildPyramids(pyr_i, pyr_g)
    torch.cuda.synchronize()
    # the pre-marshalled form (a frame ring's buffers described once): same bytes
    snap = [[t.clone() for t in p] for p in pyr_i], [[t.clone() for t in p] for p in pyr_g]
    arr = dfx.make_pyramids(pyr_i, pyr_g)
    for p in pyr_i:
        for t in p[1:]:
            t.fill_(-1.0)
    dfx.BuildPyramids(arr)
    torch.cuda.synchronize()
    for k in range(n):
        for i in range(levels):
            assert torch.equal(pyr_i[k][i], snap[0][k][i]) and torch.equal(pyr_g[k][i], snap[1][k][i])
    for k in range(n):
        ref_img = imgs[k]
        ri, rg = _alloc(levels, w, h), _alloc(levels, w, h, ch=2)
        ri[0].copy_(torch.from_numpy(imgs[k]))
        for i in range(levels):
            if i > 0:
                dfx.GaussianBlurDown(ri[i - 1], ri[i])
                ref_img = oracle.blur_down(ref_img)
            dfx.SobelGradients(ri[i], rg[i])
            got_i, got_g = pyr_i[k][i].cpu().numpy(), pyr_g[k][i].cpu().numpy()
            assert np.array_equal(got_i, ri[i].cpu().numpy()), (k, i)          # the per-level operators: same bits
            assert np.array_equal(got_g, rg[i].cpu().numpy()), (k, i)
            assert np.abs(got_i - ref_img).max() <= 1e-6 * max(1, i), (k, i)  # the oracle (blur: 1 ulp of the reference's own summation order per level)
            assert np.array_equal(got_g, oracle.sobel(got_i)), (k, i)          # Sobel of the level the GPU holds: exact
    # a single frame through the blocking entry; level 0's gradient skipped (UploadLiveFrame)
    one_i, one_g = _alloc(levels, w, h), _alloc(levels, w, h, ch=2)
    one_i[0].copy_(torch.from_numpy(imgs[0]))
    for g in one_g:
        g.fill_(-3.0)
    dfx.BuildPyramids([one_i], [[None] + one_g[1:]], blocking=True)
    assert float(one_g[0].min()) == -3.0 and float(one_g[0].max()) == -3.0
    for i in range(levels):
        assert np.array_equal(one_i[i].cpu().numpy(), pyr_i[0][i].cpu().numpy())
        if i > 0:
            assert np.array_equal(one_g[i].cpu().numpy(), pyr_g[0][i].cpu().numpy())


def test_pyramid_build_rejects_inconsistent_levels_and_keyframe_store_uses_it(dfx):
    a, g = _alloc(2, 64, 48), _alloc(2, 64, 48, ch=2)
    bad = [a[0], torch.zeros((20, 32), device="cuda")]
    with pytest.raises(dfx.DfxError, match="half of level"):
        dfx.BuildPyramids([bad], [g])
    with pytest.raises(dfx.DfxError, match="frame 1"):
        dfx.BuildPyramids([a, _alloc(2, 32, 24)], [g, _alloc(2, 32, 24, ch=2)])
    # deepfactors_amd.keyframe.Frame.FillPyramids goes through it
    rng = np.random.default_rng(5)
    img = rng.random((96, 128), dtype=np.float32)
    f = dfx.Frame(3, 128, 96)
    f.FillPyramids(img)
    ref = torch.from_numpy(img).cuda()
    for i in range(3):
        if i > 0:
            nxt = torch.empty((96 >> i, 128 >> i), device="cuda")
            dfx.GaussianBlurDown(ref, nxt)
            ref = nxt
        gr = torch.empty((96 >> i, 128 >> i, 2), device="cuda")
        dfx.SobelGradients(ref, gr)
        assert torch.equal(f.pyr_img[i], ref) and torch.equal(f.pyr_grad[i], gr)
    fs = [dfx.Frame(3, 128, 96) for _ in range(4)]
    dfx.Frame.FillPyramidsBatch(fs, [img] * 4)
    for q in fs:
        for i in range(3):
            assert torch.equal(q.pyr_img[i], f.pyr_img[i]) and torch.equal(q.pyr_grad[i], f.pyr_grad[i])



def test_large_build_same_bits(dfx):
    """Nine frames of 640x480 (the size from which a build's launches carry thousands of waves each): every output equals the per-level operators', repeated
    builds included."""
    rng = np.random.default_rng(11)
    n, w, h, levels = 9, 640, 480, 3
    imgs = [torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda() for _ in range(n)]
    pyr_i = [_alloc(levels, w, h) for _ in range(n)]
    pyr_g = [_alloc(levels, w, h, ch=2) for _ in range(n)]
    for k in range(n):
        pyr_i[k][0].copy_(imgs[k])
    arr = dfx.make_pyramids(pyr_i, pyr_g)
    for rep in range(3):
        for k in range(n):
            for t in pyr_i[k][1:] + pyr_g[k]:
                t.fill_(float("nan"))
        dfx.BuildPyramids(arr)
    torch.cuda.synchronize()
    for k in range(n):
        ref = imgs[k]
        for i in range(levels):
            if i > 0:
                nxt = torch.empty((h >> i, w >> i), device="cuda")
                dfx.GaussianBlurDown(ref, nxt)
                ref = nxt
                assert torch.equal(pyr_i[k][i], ref), (k, i)
            g = torch.empty((h >> i, w >> i, 2), device="cuda")
            dfx.SobelGradients(ref, g)
            assert torch.equal(pyr_g[k][i], g), (k, i)
