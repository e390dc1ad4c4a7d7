"""CPU-side checks of the product: the C-ABI library builds/loads and exports every symbol include/dfx.h declares,
struct layouts match the header, and calls fail loudly (never fall back) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dfx.h")).read()
    return sorted(set(re.findall(r"DFX_API\s+[\w\s\*]+?\b(dfx_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import deepfactors_amd
    assert os.path.exists(deepfactors_amd.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(deepfactors_amd.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # the Python binding table covers the header too
    assert set(syms) == set(deepfactors_amd.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    from deepfactors_amd import _lib
    assert C.sizeof(_lib.Img) == 24 and C.sizeof(_lib.SE3) == 28 and C.sizeof(_lib.Cam) == 24
    assert C.sizeof(_lib.SfmParams) == 20 and _lib.SfmParams.step_blocks.offset == 16 and C.sizeof(_lib.CorrItem) == 16
    assert _lib.CorrItem.inliers.offset == 8
    # JTJJrReductionItem<float,NP> sizes quoted in SURVEY.md (reduction_items.h:139-142): 120 B (NP=6), 4152 B (NP=44)
    assert _lib.item_size(6) == 120 and _lib.item_size(44) == 4152 and _lib.item_inliers_offset(44) == 4144
    assert C.sizeof(_lib.SfmPair) == 28 * 2 + 24 + 6 * 24 + 0 or C.sizeof(_lib.SfmPair) % 8 == 0


def test_no_cpu_fallback():
    """Without a HIP device the product path raises; it must never route through the oracle or any CPU code."""
    import torch
    import deepfactors_amd as dfx
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dfx.DfxError):
        dfx.Context(0)
    lib = C.CDLL(dfx.LIB_PATH)
    h = C.c_void_p()
    lib.dfx_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    rc = lib.dfx_ctx_create(0, None, C.byref(h))
    assert rc == -3 and not h.value   # DFX_E_NOGPU
    lib.dfx_last_error.restype = C.c_char_p
    assert b"HIP device" in lib.dfx_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "deepfactors_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libdfx_oracle" not in txt, f
                assert "dfx_oracle" not in txt, f


def test_result_item_views():
    from deepfactors_amd import JTJJrReductionItem, item_size
    raw = np.zeros(item_size(6), np.uint8)
    f = raw[:112].view(np.float32)
    f[:21] = np.arange(21)
    f[21:27] = 10 + np.arange(6)
    f[27] = 3.5
    raw[112:120].view(np.uint64)[0] = 1234
    it = JTJJrReductionItem(6, raw)
    M = it.toDenseMatrix()
    assert M[0, 5] == 5 and M[5, 0] == 5 and M[1, 1] == 6 and M[5, 5] == 20 and np.allclose(M, M.T)
    assert it.residual == 3.5 and it.inliers == 1234 and it.Jtr[5] == 15


def test_synth_camera_pyramid_matches_reference_rule():
    """camera_pyramid.h:41-46 + pinhole_camera_impl.h:126-136: integer-halved size, intrinsics scaled by the size ratio."""
    from deepfactors_amd import synth
    cams = synth.camera_pyramid(synth.scenenet_cam(640, 480), 3)
    assert [tuple(c[4:]) for c in cams] == [(640, 480), (320, 240), (160, 120)]
    assert np.allclose(cams[1][:4], cams[0][:4] / 2) and np.allclose(cams[2][:4], cams[0][:4] / 4)
    odd = synth.camera_pyramid(np.array([100, 100, 50.5, 37.5, 101, 75], np.float32), 2)
    assert tuple(odd[1][4:]) == (50, 37) and np.isclose(odd[1][0], 100 * np.float32(50 / 101))
