"""Generates tests/golden/se3_fixture_1047_1052.npz from the reference's own test images.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_fixtures.py

Source data: /root/reference/data/testimg/{1047.jpg,1052.jpg,1047.png} -- the fixture of
tests/ut_se3aligner.cpp:52-54 (320x240 SceneNet pair + 16-bit depth in millimetres).
Stored RAW (uint8 grayscale / uint16 depth); preprocessing (/255, /1000, 25x25 box blur) is redone
by the tests exactly as ut_se3aligner.cpp:78-89 does, so the fixture stays small (~300 KB).

Grayscale: the reference uses cv::imread(IMREAD_GRAYSCALE), i.e. libjpeg's luma plane.  PIL's
draft('L') asks libjpeg for the same plane; any residual +-1 grey-level difference is far below
the 25x25 blur used by the test.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/data/testimg"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "se3_fixture_1047_1052.npz")


def gray(path):
    im = Image.open(path)
    im.draft("L", im.size)
    return np.asarray(im.convert("L"), dtype=np.uint8)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference images not present; fixture can only be regenerated in the build container")
    img0 = gray(os.path.join(REF, "1047.jpg"))
    img1 = gray(os.path.join(REF, "1052.jpg"))
    dpt0 = np.asarray(Image.open(os.path.join(REF, "1047.png")))
    assert dpt0.dtype in (np.uint16, np.int32), dpt0.dtype
    dpt0 = dpt0.astype(np.uint16)
    assert img0.shape == img1.shape == dpt0.shape == (240, 320)
    np.savez_compressed(OUT, img0=img0, img1=img1, dpt0_mm=dpt0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
