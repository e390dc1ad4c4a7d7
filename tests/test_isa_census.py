"""Static checks on the gfx950 code hipcc generates for the row-walk reductions (no GPU needed: hipcc cross-compiles).  Their speed rests on
properties of the instruction stream that no numerical test can see and that a compiler or source change could silently undo
(profiles/r04_tap_loads.txt, r04_launch_shape.txt):
  * the 2 x 2 taps of img1 are DWORD loads -- the compiler fuses adjacent dword loads into 8-byte ones unless the offset is opaque, and an
    8-byte load at a 4-byte lane stride costs the texture addresser 3.5 x a dword load;
  * the gradient taps are 16-byte loads, the row's ray-table entry is a SCALAR load;
  * no kernel of the file spills (a scratch access inside the pipelined loop sits in the in-order vmcnt queue: 3-5 x slower, measured)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def misc_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "misc.s"
    src = os.path.join(ROOT, "deepfactors_amd", "csrc", "dfx_misc_kernels.hip")
    # the flags of deepfactors_amd/csrc/Makefile for this file
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", str(out), src]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return out.read_text()


def _kernel(isa, mangled_prefix):
    m = re.search(r"^(%s\w*):.*?s_endpgm(.*?)\.end_amdhsa_kernel" % re.escape(mangled_prefix), isa, re.S | re.M)
    assert m, f"kernel {mangled_prefix}* not found in the listing"
    body = isa[m.start():m.end()]
    return body


def _innermost_loop(body):
    """instructions between the loop label that is the target of the LAST backward s_cbranch and that branch: the unrolled row steps + the item loop"""
    lines = body.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"^\s+s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    assert loops, "no loop found"
    a, b = max(loops, key=lambda s: s[1] - s[0])
    return [l.split()[0] for l in lines[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]


@pytest.mark.parametrize("prefix,grad", [("_ZN3dfx16k_se3_step_batch", True), ("_ZN3dfx17k_sfm_error_batch", False), ("_ZN3dfx14k_se3_step_dev", True)])
def test_row_walk_load_forms(misc_isa, prefix, grad):
    ops = _innermost_loop(_kernel(misc_isa, prefix))
    n = {k: sum(1 for o in ops if o == k) for k in ("buffer_load_dword", "buffer_load_dwordx2", "buffer_load_dwordx4", "s_load_dword")}
    assert n["buffer_load_dwordx2"] == 0, f"the img1 taps were fused back into 8-byte loads: {n}"
    steps = n["buffer_load_dwordx4"] // 2 if grad else None          # two 16-byte gradient taps per row step
    if grad:
        assert n["buffer_load_dwordx4"] >= 2 and n["buffer_load_dwordx4"] % 2 == 0, n
    else:
        assert n["buffer_load_dwordx4"] == 0, n
        steps = (n["buffer_load_dword"] - 1) // 6                    # per row step: depth + intensity + four taps; + the item's ray column
    assert steps >= 2, n
    assert n["buffer_load_dword"] >= 6 * steps, f"expected depth + intensity + four dword taps per row step: {n}"
    assert n["s_load_dword"] >= steps, f"the row's ray-table entry should be a scalar load: {n}"


def test_no_kernel_of_the_file_spills(misc_isa):
    spills = re.findall(r"^; ScratchSize: (\d+)", misc_isa, re.M)
    assert spills and all(int(s) == 0 for s in spills), spills
    assert "scratch_load" not in misc_isa and "scratch_store" not in misc_isa


@pytest.fixture(scope="module")
def step_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_step") / "step.s"
    src = os.path.join(ROOT, "deepfactors_amd", "csrc", "dfx_sfm_step.hip")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", str(out), src]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    return out.read_text()


def test_step_kernel_does_not_spill_on_the_dense_jacobian_path(step_isa):
    """k_sfm_step<NCB, MODE, JDENSE, ...>: every instantiation that streams a dense (unpitched) code Jacobian -- the reference's layout, the
    headline and every BASELINE config -- must be free of scratch: spills in its pipelined loop cost 37-41 % (profiles/r03_ab_occupancy.txt).
    (The pitched-Jacobian instantiations at CS = 64 carry 32-36 bytes of scratch: a rarely used fall-back, measured, tolerated.)"""
    seen = 0
    for m in re.finditer(r"^(_ZN3dfx10k_sfm_stepILi(\d)ELi(\d)ELb([01])E\w+):.*?^; ScratchSize: (\d+)", step_isa, re.S | re.M):
        name, ncb, mode, jdense, scratch = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4) == "1", int(m.group(5))
        if jdense:
            seen += 1
            assert scratch == 0, (name, scratch)
        else:
            assert scratch <= 64, (name, scratch)
    assert seen >= 8
    # the default kernel of the headline: CS = 32 (NCB 2), photometric (MODE 0), dense Jacobian, bf16 split -- three waves per SIMD need <= 168 VGPRs
    m = re.search(r"^_ZN3dfx10k_sfm_stepILi2ELi0ELb1ELb1ELb0ELb0ELb1ELb1E\w+:.*?^; NumVgprs: (\d+).*?^; Occupancy: (\d+)", step_isa, re.S | re.M)
    assert m, "the headline instantiation k_sfm_step<2, 0, true, true, false, false, true, true> is gone"
    assert int(m.group(1)) <= 168 and int(m.group(2)) >= 3, m.groups()
