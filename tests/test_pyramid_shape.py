"""Launch shape of the pyramid build's row-streaming kernel (k_pyr_rows; dfx_build_pyramid_batch_async) -- host arithmetic of the shipped library,
asked through dfx_debug_pyramid_launch_shape; no device needed.  Every workgroup of such a launch lives about as long as the launch, so the rule gives the compute
units EQUAL numbers of them wherever the frame count allows (832 workgroups on 256 CUs = 3.25 each cost level 0 of a 64-frame build 12 %, profiles/r06_pyramid.txt)."""
import ctypes as C

import pytest

from deepfactors_amd import _lib


def _shape(w, h, n, cus=256):
    r, g, v = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = _lib.lib().dfx_debug_pyramid_launch_shape(w, h, n, cus, C.byref(r), C.byref(g), C.byref(v))
    assert rc == 0
    return r.value, g.value, v.value


@pytest.mark.parametrize("w,h,n,rows,wgs_per_cu", [(640, 480, 64, 60, 2), (320, 240, 64, 20, 3), (640, 480, 32, 30, 2), (320, 240, 32, 10, 3),
                                                  (640, 480, 128, 120, 2), (1280, 960, 16, 60, 2), (256, 192, 64, 12, 4)])
def test_the_measured_shapes(w, h, n, rows, wgs_per_cu):
    r, g, v = _shape(w, h, n)
    assert r == rows and g == wgs_per_cu * 256 and 8 <= wgs_per_cu * v <= 16


@pytest.mark.parametrize("w,h,n,rows,wgs", [(640, 480, 60, 60, 480), (640, 480, 48, 30, 768), (320, 240, 48, 16, 720), (640, 480, 100, 96, 500)])
def test_frame_counts_that_do_not_divide_the_cus_get_nearly_equal_shares(w, h, n, rows, wgs):
    r, g, v = _shape(w, h, n)
    assert (r, g) == (rows, wgs)
    busiest = -(-g // 256)
    assert g / (busiest * 256) >= 0.9 and 8 <= busiest * v <= 16


def test_few_frames_keep_the_many_waves_rule():
    for n in (1, 2, 3, 4, 9):
        r, g, v = _shape(640, 480, n)
        assert r == max(4, (-(-480 * 5 * n // 4096) + 1) & ~1)


@pytest.mark.parametrize("cus", [0, 64, 256, 304])
def test_segments_tile_the_height_for_every_size_and_count(cus):
    for w in (2, 64, 126, 130, 320, 640, 1024, 1280, 2048):
        for h in (1, 3, 4, 7, 60, 100, 131, 240, 480, 960, 1080):
            for n in (1, 2, 3, 5, 9, 16, 63, 64, 100, 256, 1000):
                r, g, v = _shape(w, h, n, cus)
                nstrips = (w + 127) // 128
                assert r >= 4 and r % 2 == 0 and 1 <= v <= 8
                nsegs = (h + r - 1) // r
                gps = (nstrips + v - 1) // v
                assert g == nsegs * gps * n and gps * v >= nstrips
                assert (nsegs - 1) * r < h <= nsegs * r           # the segments cover every row, none is empty


def test_refuses_what_the_kernel_cannot_take():
    assert _lib.lib().dfx_debug_pyramid_launch_shape(641, 480, 1, 256, None, None, None) != 0
    assert _lib.lib().dfx_debug_pyramid_launch_shape(640, 0, 1, 256, None, None, None) != 0
