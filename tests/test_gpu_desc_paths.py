"""The descriptor array of a batched launch reaches the kernels in one of two ways (dfx_api.cpp: simple_zerocopy / desc_zerocopy): read straight
out of the pinned staging slot (default for the batched SE3 step / EvaluateError / decoder, opt-in for the batched SfM step) or through a device
copy made on the context's copy stream (the other way round).  Both must give the same bytes.  The switches are per-context options
(dfx_ctx_configure: DFX_OPT_SIMPLE_DESC_ZEROCOPY / DFX_OPT_STEP_DESC_ZEROCOPY; environment variables until round 5); each setting still runs in its own
interpreter so that no allocation of one run can serve another."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib
ctx = dfx.Context(0)
for opt, val in (OPTIONS):
    ctx.configure(getattr(_lib, opt), val)
w, h, cs, n = 192, 144, 32, 9
al, se3 = dfx.SfmAligner(code_size=cs, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
plist, slist, keep = [], [], []
for k in range(n):
    p = synth.make_pair(w, h, cs, seed=4100 + k, device="cuda", motion_scale=0.5 + 0.1 * k)
    keep.append(p)
    plist.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"], valid0=p["valid0"]))
    slist.append(dict(se3=p["pose10_true"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]))
hsh = hashlib.sha256()
for rep in range(12):          # more calls than the staging ring has slots: slots are reused while earlier launches may still be in flight
    ed = torch.zeros(16 * n, dtype=torch.uint8, device="cuda")
    sd = torch.zeros(dfx.item_size(6) * n, dtype=torch.uint8, device="cuda")
    it = torch.zeros(dfx.item_size(12 + cs) * n, dtype=torch.uint8, device="cuda")
    al.EvaluateErrorBatch(al.make_pairs(plist), ed)
    se3.RunStepBatch(se3.make_pairs(slist), sd)
    al.RunStepBatchAsync(al.make_pairs(plist), it)
    outs = [torch.empty_like(p["img0"]) for p in keep]
    dfx.UpdateDepthBatch(np.stack([np.asarray(p["code"], np.float32) for p in keep]), [p["prx_orig"] for p in keep], [p["prx_jac"] for p in keep], 2.0, outs, ctx=ctx)
    ctx.sync()
    for t in (ed, sd, it, *outs):
        hsh.update(t.cpu().numpy().tobytes())
print("DIGEST", hsh.hexdigest())
''' % ROOT


def _run(options):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", WORKER.replace("(OPTIONS)", repr(tuple(options.items())))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")]
    assert lines, r.stdout[-500:]
    return lines[-1].split()[1]


def test_zero_copy_and_copied_descriptors_give_the_same_bytes():
    default = _run({})
    copied = _run({"DFX_OPT_SIMPLE_DESC_ZEROCOPY": 0})
    step_zero = _run({"DFX_OPT_STEP_DESC_ZEROCOPY": 1})
    assert default == copied == step_zero
    assert len(default) == len(hashlib.sha256().hexdigest())
