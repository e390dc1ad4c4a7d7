#!/usr/bin/env python3
"""How much of the time between two step kernels is the measurement itself?  The bench workload (128 pairs, step + one-kernel tail with the
graph assembly) back to back on one stream, wall time per step with the library's profiling events around every step kernel
(dfx_set_profiling, what bench.py's timed region runs with) and without them."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import bench
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    from deepfactors_amd.dist import NormalEquations, PairGraph
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    P = 128
    al = dfx.SfmAligner(code_size=32, ctx=ctx)
    pairs, keep = bench.build_pairs(dfx, synth, dev, 0, P, 640, 480, 32, ctx=ctx)
    arr = al.make_pairs(pairs)
    items = torch.zeros(P * dfx.item_size(44), dtype=torch.uint8, device=dev)
    neq = NormalEquations(PairGraph.chain(P), 32, dev)
    for _ in range(1500):
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
    ctx.sync()
    n = 600
    for rep in range(3):
        for prof in (True, False):
            ctx.set_profiling(prof)
            for _ in range(100):
                al.RunStepBatchAssembleAsync(arr, items, neq, 0)
            ctx.sync()
            if prof:
                ctx.profile_read_ex()
            t0 = time.perf_counter()
            for _ in range(n):
                al.RunStepBatchAssembleAsync(arr, items, neq, 0)
            ctx.sync()
            t1 = time.perf_counter()
            k = ""
            if prof:
                nl, ms, mn, mx = ctx.profile_read_ex()
                k = f"  step kernel {1e3 * ms / nl:.1f} us"
            ctx.set_profiling(False)
            print(f"rep {rep} profiling events {'on ' if prof else 'off'}: wall {1e6 * (t1 - t0) / n:7.1f} us per step{k}", flush=True)


if __name__ == "__main__":
    main()
