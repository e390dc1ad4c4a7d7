#!/usr/bin/env python3
"""Within-one-GPU-call A/B of libdfx.so tuning variants (tools/ab_variants.sh).  Each variant runs in its own
process (DFX_LIB selects the .so); every worker generates one synthetic pair, clones it to `--pairs` distinct
allocations (same bytes, distinct HBM), and reports the step-kernel time from the library's own HIP events.
Usage: python tools/ab_bench.py --libs gpurun_build/libdfx_a.so,... [--blocks 0,48,64] [--rounds 3]"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    dev = torch.device("cuda", 0)
    base = synth.make_pair(a.width, a.height, a.cs, seed=0xDF02, device=dev)
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(a.mode)
    res = {}
    for blocks in [int(b) for b in a.blocks.split(",")]:
        al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=blocks), code_size=a.cs, ctx=ctx)
        pairs, keep = [], []
        for k in range(a.pairs):
            if a.distinct:   # bench.py's workload: every pair its own scene, motion and validity pattern
                t = synth.make_pair(a.width, a.height, a.cs, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
                t["valid0"] = ctx.alloc_image(a.width, a.height)   # library-owned map (shadowed), as bench.py keeps it
            else:
                t = {n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in base.items()}
            keep.append(t)
            pairs.append(dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"],
                              prx0_jac=t["prx_jac"], grad1=t["grad1"], **(dict(valid0=t["valid0"]) if a.distinct else {})))
        arr = al.make_pairs(pairs)
        items = torch.zeros(a.pairs * dfx.item_size(12 + a.cs), dtype=torch.uint8, device=dev)
        for _ in range(a.preroll):   # every configuration: building its inputs idles the GPU, and the clocks need ~50-100 launches after idle
            al.RunStepBatchAsync(arr, items)
        ctx.sync()
        ctx.set_profiling(True)
        for _ in range(a.steps):
            al.RunStepBatchAsync(arr, items)
        n, ms = ctx.profile_read()
        ctx.set_profiling(False)
        it = al.items_from_bytes(items.cpu().numpy(), a.cs)[0]
        res[blocks] = dict(kernel_us=ms / n * 1e3, inliers=it.inliers, residual=it.residual)
        del keep, pairs, arr
    print("ABRESULT " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--blocks", default="0")
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cs", type=int, default=32)
    ap.add_argument("--mode", type=int, default=0, help="0 = fp32 chain (DFX_MFMA_F32_CHAIN), 1 = exact bf16 split (DFX_MFMA_BF16X3)")
    ap.add_argument("--preroll", type=int, default=150, help="untimed launches before the first measurement of a process (clock ramp)")
    ap.add_argument("--distinct", action="store_true", help="distinct synthetic pairs with a valid0 image, as bench.py builds them")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    libs = [l for l in a.libs.split(",") if l]
    table = {}
    for r in range(a.rounds):          # interleaved rounds: variant order repeats, so drift hits all variants alike
        for lib in libs:
            env = dict(os.environ, DFX_LIB=os.path.abspath(lib))
            cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--blocks", a.blocks, "--pairs", str(a.pairs), "--steps", str(a.steps),
                   "--width", str(a.width), "--height", str(a.height), "--cs", str(a.cs), "--mode", str(a.mode), "--preroll", str(a.preroll)]
            if a.distinct:
                cmd.append("--distinct")
            try:
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180, stdin=subprocess.DEVNULL)
                line = [l for l in out.stdout.splitlines() if l.startswith("ABRESULT ")]
                res = json.loads(line[-1][9:]) if line else {"error": out.stderr[-300:]}
            except subprocess.TimeoutExpired:
                res = {"error": "timeout"}
            table.setdefault(os.path.basename(lib), []).append(res)
            print(os.path.basename(lib), "round", r, json.dumps(res), flush=True)
    print("ABSUMMARY " + json.dumps(table))


if __name__ == "__main__":
    main()
