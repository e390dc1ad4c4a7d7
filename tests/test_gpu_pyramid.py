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


def test_back_to_back_builds_over_rotating_buffer_sets_with_other_batched_work_between(dfx):
    """The build's descriptors are read out of a ring of pinned staging slots by its first launch (which mirrors them to device memory for the later ones), and a
    slot is handed out again once a LATER build has reported that it is running -- no event behind a build.  Forty enqueue-only builds over three rotating
    buffer sets (so a slot that was reused too early, or a stale mirror, shows up as a pyramid built into / from the wrong buffers), with another batched operator
    that stages through the same ring between them (event-guarded slots next to build-guarded ones), then a last build with nothing behind it."""
    rng = np.random.default_rng(2026)
    w, h, levels, nf, sets = 320, 240, 3, 5, 3
    imgs = [[torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda() for _ in range(nf)] for _ in range(sets)]
    pyr_i = [[_alloc(levels, w, h) for _ in range(nf)] for _ in range(sets)]
    pyr_g = [[_alloc(levels, w, h, ch=2) for _ in range(nf)] for _ in range(sets)]
    arrs = []
    for s in range(sets):
        for k in range(nf):
            pyr_i[s][k][0].copy_(imgs[s][k])
        arrs.append(dfx.make_pyramids(pyr_i[s], pyr_g[s]))
    cs = 32
    prx = [torch.rand((h, w), device="cuda") + 0.5 for _ in range(2)]
    jac = [torch.rand((h, w * cs), device="cuda") * 0.01 for _ in range(2)]
    dpt = [torch.empty((h, w), device="cuda") for _ in range(2)]
    codes = rng.standard_normal((2, cs)).astype(np.float32)
    for rep in range(40):
        dfx.BuildPyramids(arrs[rep % sets])
        if rep % 3 == 1:
            dfx.UpdateDepthBatch(codes, prx, jac, 2.0, dpt)
    dfx.BuildPyramids(arrs[0])
    torch.cuda.synchronize()
    for s in range(sets):
        for k in range(nf):
            ref = imgs[s][k]
            for i in range(levels):
                if i > 0:
                    nxt = torch.empty((h >> i, w >> i), device="cuda")
                    dfx.GaussianBlurDown(ref, nxt)
                    ref = nxt
                    assert torch.equal(pyr_i[s][k][i], ref), (s, k, i)
                g = torch.empty((h >> i, w >> i, 2), device="cuda")
                dfx.SobelGradients(ref, g)
                assert torch.equal(pyr_g[s][k][i], g), (s, k, i)
    one = torch.empty((h, w), device="cuda")
    dfx.UpdateDepth(codes[0], prx[0], jac[0], 2.0, one)
    assert torch.equal(dpt[0], one)


def test_repeated_builds_into_the_same_buffers_follow_the_new_frame(dfx):
    """A live frame's pyramid is rebuilt in place frame after frame (UploadLiveFrame, core/deepfactors.cpp:616-630): from the second build on the library finds the
    previous build's descriptors still in device memory and stages nothing -- the result must follow the NEW level-0 image, a build over other buffers in between
    must not be served from the remembered block, and neither must a build of fewer levels over the same buffers."""
    rng = np.random.default_rng(77)
    w, h, levels = 640, 480, 4
    pi, pg = [_alloc(levels, w, h)], [_alloc(levels, w, h, ch=2)]
    qi, qg = [_alloc(levels, w, h)], [_alloc(levels, w, h, ch=2)]
    arr_p, arr_q = dfx.make_pyramids(pi, pg), dfx.make_pyramids(qi, qg)
    arr_p3 = dfx.make_pyramids([pi[0][:3]], [pg[0][:3]])

    def check(pyr_i, pyr_g, img, nl):
        ref = img
        for i in range(nl):
            if i > 0:
                nxt = torch.empty((h >> i, w >> i), device="cuda")
                dfx.GaussianBlurDown(ref, nxt)
                ref = nxt
                assert torch.equal(pyr_i[i], ref), i
            g = torch.empty((h >> i, w >> i, 2), device="cuda")
            dfx.SobelGradients(ref, g)
            assert torch.equal(pyr_g[i], g), i
    for frame in range(6):
        img = torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda()
        pi[0][0].copy_(img)
        if frame == 3:   # other buffers in between
            qi[0][0].copy_(img * 0.5)
            dfx.BuildPyramids(arr_q)
        nl = 3 if frame == 4 else levels   # (frame 4: three levels of the same buffers; frame 5: four again)
        if nl == 3:
            pi[0][3].fill_(-7.0); pg[0][3].fill_(-7.0)
        dfx.BuildPyramids(arr_p3 if nl == 3 else arr_p)
        torch.cuda.synchronize()
        check(pi[0], pg[0], img, nl)
        if nl == 3:
            assert bool((pi[0][3] == -7.0).all()) and bool((pg[0][3] == -7.0).all())
        if frame == 3:
            check(qi[0], qg[0], img * 0.5, levels)


@pytest.mark.parametrize("n", [64, 60, 32, 16])
def test_builds_whose_workgroups_tile_the_compute_units(dfx, n):
    """Frame counts for which the row-streaming launches are shaped so that every compute unit holds the same number of workgroups (pyr_rows_per_segment:
    64 frames of 640x480 -> segments of 60 rows at level 0 and 20 at level 1; 60 -> 60 / 16 (within 6 % of equal shares); 32 -> 30 / 10; 16 -> whatever balances or the ~4096-wave rule): other segment seams than
    the small builds above walk over -- every level of the first, a middle and the last frame equals the per-level operators bit for bit, and every frame's
    last level equals frame-by-frame builds (three frames per enqueue: the other rule)."""
    rng = np.random.default_rng(1000 + n)
    w, h, levels = 640, 480, 4
    base = torch.from_numpy(rng.random((h + 64, w + 64), dtype=np.float32)).cuda()
    pyr_i = [_alloc(levels, w, h) for _ in range(n)]
    pyr_g = [_alloc(levels, w, h, ch=2) for _ in range(n)]
    for k in range(n):
        pyr_i[k][0].copy_(base[k % 61: k % 61 + h, (7 * k) % 59: (7 * k) % 59 + w])   # n distinct windows of one random field
    dfx.BuildPyramids(pyr_i, pyr_g)
    torch.cuda.synchronize()
    for k in (0, n // 2 - 1, n - 1):
        ref = pyr_i[k][0]
        for i in range(levels):
            if i > 0:
                nxt = torch.empty((h >> i, w >> i), device="cuda")
                dfx.GaussianBlurDown(ref, nxt)
                ref = nxt
                assert torch.equal(pyr_i[k][i], ref), (k, i)
            g = torch.empty((h >> i, w >> i, 2), device="cuda")
            dfx.SobelGradients(ref, g)
            assert torch.equal(pyr_g[k][i], g), (k, i)
    qi = [_alloc(levels, w, h) for _ in range(3)]
    qg = [_alloc(levels, w, h, ch=2) for _ in range(3)]
    for k0 in range(0, n - 2, 3):
        for j in range(3):
            qi[j][0].copy_(pyr_i[k0 + j][0])
        dfx.BuildPyramids(qi, qg)
        torch.cuda.synchronize()
        for j in range(3):
            for i in range(1, levels):
                assert torch.equal(qi[j][i], pyr_i[k0 + j][i]) and torch.equal(qg[j][i], pyr_g[k0 + j][i]), (k0 + j, i)
            assert torch.equal(qg[j][0], pyr_g[k0 + j][0]), (k0 + j, 0)


def test_builds_on_a_side_stream_and_across_a_stream_switch(dfx):
    """The staging-slot bookkeeping of the build (a slot is free once a later build has reported that it runs on the context's stream) on a non-default stream, and
    across dfx_ctx_set_stream (which drains the old stream and forgets the ring's guards): 20 builds over two rotating buffer sets on a side stream, a switch to
    another stream in the middle of the sequence, every result checked."""
    rng = np.random.default_rng(31)
    w, h, levels, nf = 256, 192, 3, 4
    side, other = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = dfx.Context(0, stream=side)
    imgs = [[torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda() for _ in range(nf)] for _ in range(2)]
    pyr_i = [[_alloc(levels, w, h) for _ in range(nf)] for _ in range(2)]
    pyr_g = [[_alloc(levels, w, h, ch=2) for _ in range(nf)] for _ in range(2)]
    arrs = []
    for s in range(2):
        for k in range(nf):
            pyr_i[s][k][0].copy_(imgs[s][k])
        arrs.append(dfx.make_pyramids(pyr_i[s], pyr_g[s]))
    torch.cuda.synchronize()
    for rep in range(20):
        if rep == 11:
            ctx.set_stream(other)
        dfx.BuildPyramids(arrs[rep % 2], ctx=ctx)
    ctx.sync()
    torch.cuda.synchronize()
    for s in range(2):
        for k in range(nf):
            ref = imgs[s][k]
            for i in range(levels):
                if i > 0:
                    nxt = torch.empty((h >> i, w >> i), device="cuda")
                    dfx.GaussianBlurDown(ref, nxt)
                    ref = nxt
                    assert torch.equal(pyr_i[s][k][i], ref), (s, k, i)
                g = torch.empty((h >> i, w >> i, 2), device="cuda")
                dfx.SobelGradients(ref, g)
                assert torch.equal(pyr_g[s][k][i], g), (s, k, i)
