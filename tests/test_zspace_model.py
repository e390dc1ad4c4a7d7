"""Executable specification of the packed z-space of the SfM step (deepfactors_amd/csrc/dfx_sfm_step.hip): a numpy model of the
products the kernel accumulates (X, Pm on 16x16x4 MFMAs, Dd and the P P^T tiles on 4x4x1 MFMAs) and of k_sfm_finalize's scatter,
checked against the direct J^T J / J^T r of the same per-pixel rows.  Runs on the CPU: an index mistake in the layout shows up
here before any GPU time is spent."""
import numpy as np
import pytest


def _model_item(gC, wr, inl, s, jac, M, HM, cs):
    """gC [N][6], wr [N], inl [N], s [N], jac [N][cs]; returns (JtJ dense NPxNP, Jtr, residual, inliers) as the kernels build them."""
    N = len(wr)
    ncb = cs // 16
    NP = 12 + cs
    P8 = np.concatenate([gC, wr[:, None], inl[:, None]], axis=1)                   # LDS rows 0..7
    C = [s[:, None] * jac[:, b::ncb] for b in range(ncb)]                          # C_b[i] = s * jac[ncb * i + b]
    # ---- step kernel: accumulators
    X = {}
    for b in range(ncb):
        for b2 in range(b + 1, ncb):
            X[(b, b2)] = np.einsum("pi,pj->ij", C[b], C[b2])
    # Pm(b): even b: A = [P ; C_b rows 8..15], odd b: A = [C_b rows 0..7 ; P]  (lane selects, no cross-lane moves)
    Pm = [np.einsum("pi,pj->ij", np.concatenate([P8, C[b][:, 8:]] if b % 2 == 0 else [C[b][:, :8], P8], axis=1), C[b]) for b in range(ncb)]
    # Dd(q) on 4x4x1: operand M = [C_b0 rows 0..7 ; C_b1 rows 8..15] in the 16x16x4 lane layout = 16 blocks (pixel k, row group rg);
    # form 0 broadcasts the A rows of the first block of every block pair (CBSZ = 1, ABID = 0), form 1 is the plain product.
    # Accumulators: [form][block = 4 k + rg][i][j], pixels are grouped by fours as in the kernel (group g, k = pixel % 4).
    Dd = []
    for q in range((ncb + 1) // 2):
        b0, b1 = 2 * q, min(2 * q + 1, ncb - 1)
        Mop = np.concatenate([C[b0][:, :8], C[b1][:, 8:]], axis=1)          # [N][16]
        acc = np.zeros((2, 16, 4, 4))
        for px in range(N):
            k = px % 4
            for rg in range(4):
                Bg = Mop[px, 4 * rg:4 * rg + 4]
                A0 = Mop[px, 4 * (rg & ~1):4 * (rg & ~1) + 4]              # broadcast: block pair (0,1) -> group 0, (2,3) -> group 2
                acc[0, 4 * k + rg] += np.outer(A0, Bg)
                acc[1, 4 * k + rg] += np.outer(Bg, Bg)
        Dd.append(acc)
    # P P^T tiles on 4x4x1: instruction t handles pixels 5t .. 5t+4; block b < 15 holds tile b % 3 of pixel 5t + b // 3
    S0 = np.zeros((16, 4, 4))
    for t in range((N + 4) // 5):
        for b in range(15):
            px = 5 * t + b // 3
            if px >= N:
                continue
            typ = b % 3
            tr, tc = (1 if typ == 2 else 0), (1 if typ != 0 else 0)
            S0[b] += np.outer(P8[px, 4 * tr:4 * tr + 4], P8[px, 4 * tc:4 * tc + 4])

    # ---- finalize
    def pp(p, q):
        typ = 0 if q < 4 else (1 if p < 4 else 2)
        return sum(S0[b][p & 3, q & 3] for b in range(typ, 15, 3))

    T = np.zeros((12, 6))
    for n in range(12):
        j, grp = n % 3, n // 3
        for i in range(6):
            if grp == 0:
                T[n, i] = M[3 * i + j] if i < 3 else 0.0
            elif grp == 1:
                T[n, i] = M[3 * (i - 3) + j] if i >= 3 else 0.0
            elif grp == 2:
                T[n, i] = -M[3 * i + j] if i < 3 else 0.0
            else:
                T[n, i] = -HM[3 * i + j] if i < 3 else -M[3 * (i - 3) + j]
    H = np.full((NP, NP), np.nan)
    g = np.full(NP, np.nan)

    def put(a, b, v):
        lo, hi = min(a, b), max(a, b)
        assert np.isnan(H[lo, hi]), f"entry ({lo},{hi}) written twice"
        H[lo, hi] = v

    G = np.array([[pp(min(i, j), max(i, j)) for j in range(6)] for i in range(6)])
    TG = T @ G @ T.T
    for n in range(12):
        for m in range(n, 12):
            put(n, m, TG[n, m])
        g[n] = T[n] @ np.array([pp(i, 6) for i in range(6)])
    residual, inliers = pp(6, 6), pp(7, 7)
    for (b, b2), S in X.items():
        for i in range(16):
            for j in range(16):
                put(12 + ncb * i + b, 12 + ncb * j + b2, S[i, j])
    for b, S in enumerate(Pm):
        pofs = 8 if b % 2 else 0
        cofs = 8 - pofs
        for n in range(12):
            for j in range(16):
                put(n, 12 + ncb * j + b, T[n] @ S[pofs:pofs + 6, j])
        for j in range(16):
            assert np.isnan(g[12 + ncb * j + b])
            g[12 + ncb * j + b] = S[pofs + 6, j]
        for r in range(cofs, cofs + 8):
            for c in range(16):
                if c >= r or c < cofs:
                    put(12 + ncb * r + b, 12 + ncb * c + b, S[r, c])
    for q, acc in enumerate(Dd):
        Tl = acc.reshape(2, 4, 4, 4, 4).sum(axis=1)       # fold the 4 pixels of a group: [form][rg][i][j]
        for rg in range(4):
            bb = 2 * q if rg < 2 else 2 * q + 1
            if bb >= ncb:
                continue                                   # odd ncb: the upper half repeats rows Pm(b0) already covers
            base = 0 if rg < 2 else 8
            for i in range(4):
                for j in range(4):
                    if rg % 2 == 0:
                        if i <= j:
                            put(12 + ncb * (base + i) + bb, 12 + ncb * (base + j) + bb, Tl[0, rg, i, j])
                    else:
                        put(12 + ncb * (base + i) + bb, 12 + ncb * (base + 4 + j) + bb, Tl[0, rg, i, j])
                        if i <= j:
                            put(12 + ncb * (base + 4 + i) + bb, 12 + ncb * (base + 4 + j) + bb, Tl[1, rg, i, j])
    assert not np.isnan(H[np.triu_indices(NP)]).any(), "upper triangle not fully covered"
    assert not np.isnan(g).any()
    return H, g, residual, inliers


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_packed_zspace_covers_the_item_exactly_once(cs):
    rng = np.random.default_rng(cs)
    N = 23   # not a multiple of 5 or 4: exercises the padded pixel of the 4x4 tiles
    gC, wr, s = rng.normal(size=(N, 6)), rng.normal(size=N), rng.normal(size=N)
    inl = (rng.random(N) > 0.2).astype(np.float64)
    gC *= inl[:, None]; wr *= inl; s *= inl          # weights of invalid pixels are zero
    jac = rng.normal(size=(N, cs))
    M, HM = rng.normal(size=9), rng.normal(size=9)
    H, g, residual, inliers = _model_item(gC, wr, inl, s, jac, M, HM, cs)
    # direct: J = [gC * blkdiag(M, M) | gC * [[-M, -HM], [0, -M]] | s * jac]   (warping.h:119-134, dense_sfm.h:149-200)
    Mm, HMm = M.reshape(3, 3), HM.reshape(3, 3)
    J0 = np.concatenate([gC[:, :3] @ Mm, gC[:, 3:] @ Mm], axis=1)
    J1 = np.concatenate([-(gC[:, :3] @ Mm), -(gC[:, :3] @ HMm) - gC[:, 3:] @ Mm], axis=1)
    J = np.concatenate([J0, J1, s[:, None] * jac], axis=1)
    ref = J.T @ J
    iu = np.triu_indices(12 + cs)
    assert np.abs(H[iu] - ref[iu]).max() < 1e-9 * np.abs(ref).max()
    assert np.abs(g - J.T @ wr).max() < 1e-9 * max(1.0, np.abs(J.T @ wr).max())
    assert abs(residual - float(wr @ wr)) < 1e-9 and inliers == inl.sum()


def _wave_chunks(W, H, total_waves, wid, banded=True):
    """The chunk sequence of wave `wid` as k_sfm_step computes it (banded map with the interleaved fallback)."""
    nchunks = (W * H + 63) >> 6
    vs = (W + 32) >> 6
    if banded and vs >= 1 and total_waves >= vs:
        crows = (nchunks + vs - 1) // vs
        nbands = total_waves // vs
        per = (crows + nbands - 1) // nbands
        b, j = wid // vs, wid % vs
        if b >= nbands:
            return [], vs
        cend = min(nchunks, (b * per + per) * vs)
        return list(range(b * per * vs + j, cend, vs)), vs
    return list(range(wid, nchunks, total_waves)), total_waves


@pytest.mark.parametrize("W,H", [(640, 480), (1280, 960), (320, 240), (100, 77), (101, 67), (17, 9), (64, 1), (8, 8), (12288, 2), (96, 64)])
def test_chunk_map_visits_every_chunk_once(W, H):
    """The banded chunk -> wave map (and its fallback for tiny grids) is a partition of the chunks, and the per-lane (x, y)
    walk (one integer division for the first chunk, then a constant pixel stride with a single wrap test) stays exact."""
    nchunks = (W * H + 63) >> 6
    for bpp in (1, 2, 3, 5, 7, 16, 60, 80, 120, 240, 1000):
        total_waves = 4 * bpp
        seen = np.zeros(nchunks, np.int32)
        for wid in range(total_waves):
            seq, cstride = _wave_chunks(W, H, total_waves, wid)
            for c in seq:
                seen[c] += 1
            if len(seq) > 1:   # advance_xy: x += sdx, y += sdy, one wrap
                pstride = cstride << 6
                sdy, sdx = divmod(pstride, W)
                for lane in (0, 63):
                    p = seq[0] * 64 + lane
                    y, x = divmod(p, W)
                    for c in seq[1:]:
                        x += sdx; y += sdy
                        if x >= W:
                            x -= W; y += 1
                        assert (y, x) == divmod(c * 64 + lane, W)
        assert (seen == 1).all(), (W, H, bpp, np.flatnonzero(seen != 1)[:5])


# ---- DFX_MFMA_BF16X3: exact three-way bf16 split, plain 16x16 tiles (k_sfm_step<..., B3 = true>, k_sfm_finalize_b3) -----------------
def _rne_bf16(x):
    """float32 -> nearest-even bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does, tools/ubench/bf16x3_probe.cpp)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _split3(x):
    x = np.asarray(x, np.float32)
    h = _rne_bf16(x)
    r = (x - h).astype(np.float32)
    m = _rne_bf16(r)
    l = (r - m).astype(np.float32)
    return h, m, l


def test_three_way_bf16_split_is_exact():
    rng = np.random.default_rng(3)
    x = (rng.uniform(1, 2, 200000) * 2.0 ** rng.integers(-40, 41, 200000) * rng.choice([-1, 1], 200000)).astype(np.float32)
    h, m, l = _split3(x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    for p in (h, m, l):                                   # every piece is a bf16 (low 16 bits clear), the last one without rounding
        assert not (p.view(np.uint32) & 0xFFFF).any()


DIAG4_MAX_NCB = 4   # DFX_B3_DIAG4_MAX_NCB of dfx_sfm_step.hip


def _model_item_b3(gC, wr, inl, s, jac, M, HM, cs):
    """The bf16x3 step + k_sfm_finalize_b3 in numpy.  z blocks: P (8 rows), C_b (16 rows each).  What the step kernel accumulates
    (fp32 on the hardware, float64 here), with the P block STACKED as Pa = [P_h ; P_m], Pl0 = [P_l ; 0]:
      block 0       Pa Pa^T                          = [hh hm ; mh mm] quadrants
      block 1 + b   Pa (C_h + C_m + C_l)^T + Pl0 C_h^T : rows 0..7 = hh + hm + hl + lh, rows 8..15 = mh + mm + ml
      (C_b,C_b')    b < b': the six products; b = b': S = hh + mm with N = hm + hl in its own block (four-product diagonals, ncb <= 2),
                    else the six products
      N block of (P,P): Pl0 Pa^T: top-left quadrant = lh
    and what the finalize kernel makes of it: quadrant / row-half sums, lh + lh^T, Z = S + N + N^T."""
    ncb = cs // 16
    NP = 12 + cs
    N = len(wr)
    P8 = np.concatenate([gC, wr[:, None], inl[:, None]], axis=1).astype(np.float32)
    cblocks = [(s[:, None].astype(np.float32) * jac[:, b::ncb].astype(np.float32)).astype(np.float32) for b in range(ncb)]
    ein = lambda A, B: np.einsum("pi,pj->ij", A.astype(np.float64), B.astype(np.float64))
    ph, pm, pl = _split3(P8)
    Pa, Pl0 = np.concatenate([ph, pm], axis=1), np.concatenate([pl, np.zeros_like(pl)], axis=1)
    cp = [_split3(b) for b in cblocks]
    raw0, rawN = ein(Pa, Pa), ein(Pl0, Pa)
    t0 = np.zeros((16, 16))
    t0[:8, :8] = raw0[:8, :8] + raw0[:8, 8:] + raw0[8:, :8] + raw0[8:, 8:] + rawN[:8, :8] + rawN[:8, :8].T
    tiles = [t0]
    for b in range(ncb):
        h, m, l = cp[b]
        raw = ein(Pl0, h) + ein(Pa, l) + ein(Pa, m) + ein(Pa, h)
        t = np.zeros((16, 16))
        t[:8] = raw[:8] + raw[8:]
        tiles.append(t)
    for b in range(ncb):
        for b2 in range(b, ncb):
            (ha, ma, la), (hb, mb, lb) = cp[b], cp[b2]
            if b == b2 and ncb <= DIAG4_MAX_NCB:
                S, Nn = ein(ma, mb) + ein(ha, hb), ein(ha, lb) + ein(ha, mb)
                tiles.append(S + Nn + Nn.T)
            else:
                tiles.append(ein(ma, mb) + ein(ha, lb) + ein(la, hb) + ein(ha, mb) + ein(ma, hb) + ein(ha, hb))
    assert len(tiles) == 1 + ncb + ncb * (ncb + 1) // 2

    T = np.zeros((12, 6))
    for n in range(12):
        j, grp = n % 3, n // 3
        for i in range(6):
            if grp == 0:
                T[n, i] = M[3 * i + j] if i < 3 else 0.0
            elif grp == 1:
                T[n, i] = M[3 * (i - 3) + j] if i >= 3 else 0.0
            elif grp == 2:
                T[n, i] = -M[3 * i + j] if i < 3 else 0.0
            else:
                T[n, i] = -HM[3 * i + j] if i < 3 else -M[3 * (i - 3) + j]
    H = np.full((NP, NP), np.nan)
    g = np.full(NP, np.nan)

    def put(a, b, v):
        lo, hi = min(a, b), max(a, b)
        assert np.isnan(H[lo, hi]), f"entry ({lo},{hi}) written twice"
        H[lo, hi] = v

    # blk 0
    S = tiles[0]
    pp = lambda p, q: S[p, q]
    G = np.array([[pp(min(i, j), max(i, j)) for j in range(6)] for i in range(6)])
    TG = T @ G @ T.T
    for n in range(12):
        for m in range(n, 12):
            put(n, m, TG[n, m])
        g[n] = T[n] @ np.array([pp(i, 6) for i in range(6)])
    residual, inliers = pp(6, 6), pp(7, 7)
    # blk 1 + b
    for b in range(ncb):
        S = tiles[1 + b]
        for n in range(12):
            for j in range(16):
                put(n, 12 + ncb * j + b, T[n] @ S[0:6, j])
        for j in range(16):
            assert np.isnan(g[12 + ncb * j + b])
            g[12 + ncb * j + b] = S[6, j]
    # code-code tiles: index -> (b, b2) exactly as k_sfm_finalize_b3 decodes blockIdx.x
    for blk in range(1 + ncb, len(tiles)):
        q = blk - 1 - ncb
        for b in range(ncb):
            n = ncb - b
            if q < n:
                b2 = b + q
                break
            q -= n
        S = tiles[blk]
        for i in range(16):
            for j in range(16):
                if b != b2 or i <= j:
                    put(12 + ncb * i + b, 12 + ncb * j + b2, S[i, j])
    assert not np.isnan(H[np.triu_indices(NP)]).any(), "upper triangle not fully covered"
    assert not np.isnan(g).any()
    return H, g, residual, inliers


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_bf16x3_tiles_cover_the_item_exactly_once_at_fp32_accuracy(cs):
    rng = np.random.default_rng(100 + cs)
    N = 64
    gC, wr, s = rng.normal(size=(N, 6)), rng.normal(size=N), rng.normal(size=N)
    inl = (rng.random(N) > 0.2).astype(np.float64)
    gC *= inl[:, None]; wr *= inl; s *= inl
    jac = rng.normal(size=(N, cs))
    gC, wr, s, jac = (a.astype(np.float32).astype(np.float64) for a in (gC, wr, s, jac))
    M, HM = rng.normal(size=9), rng.normal(size=9)
    H, g, residual, inliers = _model_item_b3(gC, wr, inl, s, jac, M, HM, cs)
    Mm, HMm = M.reshape(3, 3), HM.reshape(3, 3)
    C = (s[:, None].astype(np.float32) * jac.astype(np.float32)).astype(np.float64)      # the kernel's fp32 product s * jac
    J0 = np.concatenate([gC[:, :3] @ Mm, gC[:, 3:] @ Mm], axis=1)
    J1 = np.concatenate([-(gC[:, :3] @ Mm), -(gC[:, :3] @ HMm) - gC[:, 3:] @ Mm], axis=1)
    J = np.concatenate([J0, J1, C], axis=1)
    ref = J.T @ J
    iu = np.triu_indices(12 + cs)
    # the dropped ml + lm + ll terms are < 2^-26 of a product
    assert np.abs(H[iu] - ref[iu]).max() < 3e-8 * np.abs(ref).max()
    assert np.abs(g - J.T @ wr).max() < 3e-8 * max(1.0, np.abs(J.T @ wr).max())
    assert abs(residual - float(wr @ wr)) < 3e-8 * float(wr @ wr) and abs(inliers - inl.sum()) < 1e-6


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_bf16x3_diag4_equals_the_six_product_tiles(cs):
    """Four-product diagonal tiles (DFX_B3_DIAG4_MAX_NCB): they keep S = hh + mm and N = hm + hl; Z = S + N + N^T equals the six-product tile."""
    rng = np.random.default_rng(cs)
    z = rng.normal(size=(64, 16)).astype(np.float32)
    h, m, l = (a.astype(np.float64) for a in _split3(z))
    ein = lambda A, B: np.einsum("pi,pj->ij", A, B)
    six = ein(m, m) + ein(h, l) + ein(l, h) + ein(h, m) + ein(m, h) + ein(h, h)
    S, N = ein(m, m) + ein(h, h), ein(h, l) + ein(h, m)
    assert np.abs(S + N + N.T - six).max() < 1e-12 * np.abs(six).max()
    # decode of a diagonal tile index as k_sfm_finalize_b3 does it
    ncb = cs // 16
    diag = {}
    for blk in range(1 + ncb + ncb * (ncb + 1) // 2):
        d = -1
        if blk == 0:
            d = 0
        elif blk > ncb:
            q = blk - 1 - ncb
            for b in range(ncb):
                if q == 0:
                    d = 1 + b
                    break
                q -= ncb - b
                if q < 0:
                    break
        diag[blk] = d
    tiles = [(0, 0)] + [(0, 1 + b) for b in range(ncb)] + [(1 + b, 1 + b2) for b in range(ncb) for b2 in range(b, ncb)]
    for blk, (ka, kb) in enumerate(tiles):
        assert diag[blk] == (ka if ka == kb else -1), (blk, ka, kb, diag[blk])
