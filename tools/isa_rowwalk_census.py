#!/usr/bin/env python3
"""Static census of the row-walk kernels' innermost loops (hipcc cross-compiles: no GPU needed): instruction mix, registers, occupancy.
usage: tools/isa_rowwalk_census.py [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_isa_census import _innermost_loop  # noqa: E402


def main():
    out = os.path.join(tempfile.mkdtemp(prefix="isa_", dir="/tmp"), "misc.s")
    src = os.path.join(ROOT, "deepfactors_amd", "csrc", "dfx_misc_kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S",
           "-o", out, src] + sys.argv[1:]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    isa = open(out).read()
    print("listing:", out)
    for m in re.finditer(r"^(_ZN3dfx\d+k_(?:se3_step|sfm_error)\w*):.*?\.end_amdhsa_kernel.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", isa, re.S | re.M):
        name = m.group(1)
        body = isa[m.start():m.end()]
        try:
            ops = _innermost_loop(body)
        except AssertionError:
            ops = []
        c = Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_"))
        print(f"{name[:44]:44s} loop {len(ops):4d} valu {valu:4d} salu {salu:4d} vgpr {m.group(2):>3s} scratch {m.group(3)} occ {m.group(4)} | "
              + " ".join(f"{k}={v}" for k, v in sorted(c.items()) if "load" in k or "dpp" in k or "waitcnt" in k or "cbranch" in k or "cndmask" in k))


if __name__ == "__main__":
    main()
