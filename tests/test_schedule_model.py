"""Executable model (numpy, CPU) of the dynamic schedule's index arithmetic in k_sfm_step<..., DYN> (deepfactors_amd/csrc/dfx_sfm_step.hip)
and of its host-side parameters (dfx_api.cpp): wave -> (pair, team member), item -> chunks, chunk -> image row by multiply-high.
The kernel is tested against the oracle on the GPU (tests/test_gpu_configs.py); this pins the arithmetic it relies on."""
import numpy as np
import pytest

WAVES = 4 * 4 * 256          # resident grid: 4 workgroups per CU x 4 waves x 256 CUs


@pytest.mark.parametrize("npairs", [16, 20, 128, 130, 1000, 4096])
def test_every_pair_gets_a_full_team_spread_over_the_xcds(npairs):
    gid = np.arange(WAVES)
    member = gid // npairs
    pair = (gid - member * npairs + 4 * member) % npairs        # the rotation by 4 * member moves a pair's members across workgroups
    team = WAVES // npairs
    live = member < team                                         # surplus waves retire at once
    assert live.sum() == team * npairs
    seen = np.zeros((npairs, team), int)
    np.add.at(seen, (pair[live], member[live]), 1)
    assert (seen == 1).all()                                     # a bijection: every (pair, member) slot has exactly one wave
    if npairs == 128:                                            # workgroup b runs on XCD b mod 8: every team has 4 members on each
        xcd = (gid // 4) % 8
        per = np.zeros((npairs, 8), int)
        np.add.at(per, (pair[live], xcd[live]), 1)
        assert (per == team // 8).all()


@pytest.mark.parametrize("w,h,rows", [(640, 480, 6), (640, 480, 32), (128, 96, 2), (320, 240, 7), (4096, 100, 3)])
def test_items_cover_every_chunk_exactly_once(w, h, rows):
    vs = w // 64                                                 # chunk columns; the dynamic schedule needs W % 64 == 0
    nchunks = w * h // 64
    items = vs * ((h + rows - 1) // rows)
    count = np.zeros(nchunks, int)
    for t in range(items):
        band, col = divmod(t, vs)
        row0 = band * rows
        c, end = row0 * vs + col, min((row0 + rows) * vs, nchunks)
        while c < end:                                           # the wave walks down the image: one chunk row per step
            count[c] += 1
            c += vs
    assert (count == 1).all()


def test_chunk_row_by_multiply_high_is_exact():
    """dyn_row(c) = umulhi(c, 2^32 / vs + 1) == c // vs for every chunk id below 2^20 and every chunk-row length 1..64."""
    c = np.concatenate([np.arange(0, 1 << 20, 7), np.arange((1 << 20) - 4096, 1 << 20)]).astype(np.uint64)
    for vs in range(1, 65):
        magic = np.uint64((1 << 32) // vs + 1)
        assert np.array_equal((c * magic) >> np.uint64(32), c // np.uint64(vs)), vs
