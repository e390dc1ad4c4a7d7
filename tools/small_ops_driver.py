#!/usr/bin/env python3
"""A few launches of the batched SE3 step and the batched EvaluateError over 128 distinct 640x480 pairs (bench.py's small-operator workload),
for counter collection: rocprofv3 --pmc ... --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" -- python tools/small_ops_driver.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    W, H, CS, P = 640, 480, 32, 128
    prs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev) for k in range(P)]
    al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    sarr = se3.make_pairs([dict(se3=(p["pose10_true"] if os.environ.get("POSE") == "true" else synth.IDENTITY), cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
    sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
    earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                               grad1=p["grad1"]) for p in prs])
    eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
    for _ in range(int(os.environ.get("REPS", "6"))):
        se3.RunStepBatch(sarr, sitems)
        al.EvaluateErrorBatch(earr, eitems)
    ctx.sync()


if __name__ == "__main__":
    main()
