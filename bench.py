#!/usr/bin/env python3
"""bench.py -- keyframe-pair residual+Jacobian evaluations per second (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM: a single batched launch of
SfmAligner::RunStep (reference cu_sfmaligner.cpp:149-185) over `--pairs` independent 640x480, 32-code keyframe->frame
pairs (BASELINE.json configs[1] geometry, batched; every pair has its own keyframe so the working set,
pairs x 47 MB = 6 GB, is far beyond the 256 MB Infinity Cache and the sweep is honestly HBM-resident), with the assembly
of the Gauss-Newton normal-equation blocks fused into its finalize kernel and -- for N > 1 -- their RCCL all-reduce over
xGMI.  The default, 128 pairs per GPU, is the per-GPU shard of BASELINE configs[3] ("~1k pairs sharded across 8 GPUs"):
`--gpus 8` IS that configuration, `--gpus 1` is one eighth of it (weak scaling).  `--pairs 16` gives the small-batch figure
quoted in DESIGN.md section 5.

Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL); pairs are independent units, sharded
contiguously across ranks (weak scaling: --pairs per GPU); the only exchange step is the RCCL reduce of the
block-tridiagonal normal equations onto rank 0, where the solve runs (SURVEY.md section 8e option 2).

Output: ONE JSON line on rank 0 (see the driver contract in the task statement), including
  roofline     -- step kernel only: algorithmic bytes (148 B/px x px x pairs per launch) / HIP-event duration
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's host path) timed on this box's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=128, help="keyframe pairs per GPU per step (one batched launch)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cs", type=int, default=32)
    ap.add_argument("--step-blocks", type=int, default=0, help="workgroups per pair (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(w, h, cs, seconds):
    """The oracle (a port of the reference's host loop over DenseSfm<...,TargetHost>, ut_sfmaligner.cpp:307-315), fp32
    accumulate, OpenMP over rows on all host cores.  Bounded sample: the same 640x480x32 pair, repeated ~`seconds`."""
    from deepfactors_amd import synth
    from oracle import dfx_oracle as orc   # cpu_baseline leg only
    orc.build()
    n = synth.to_numpy(synth.make_pair(w, h, cs, seed=0xDF02, device="cpu"))
    cores = orc.max_threads()
    args = (n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])
    orc.sfm_step(*args, accum_f64=False, threads=cores)   # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.sfm_step(*args, accum_f64=False, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 2000:
            break
    return dict(value=reps / dt, unit="pair-evals/s", cores=cores, kind="port",
                sample=f"{reps} repetitions of one {w}x{h} cs={cs} SfmAligner::RunStep pair, OpenMP over rows, fp32 accumulate, {dt:.1f} s")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs torch.distributed.run --nproc-per-node {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: use RCCL even for one rank (exercises the exchange step)
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    from deepfactors_amd.dist import NormalEquations

    W, H, CS, P = a.width, a.height, a.cs, a.pairs
    ctx = dfx.Context(local)
    al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=a.step_blocks), code_size=CS, ctx=ctx)

    # ---- synthetic, device-resident input: P distinct keyframe->frame pairs per rank
    pairs, keep = [], []
    for k in range(P):
        p = synth.make_pair(W, H, CS, seed=0xDF02 + (0 if os.environ.get("DFX_BENCH_SAME") else 1000 * rank + k), device=dev, motion_scale=(1.0 if os.environ.get("DFX_BENCH_SAME") else 0.6 + 0.05 * (k % 8)))
        keep.append(p)
        pairs.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"],
                          prx0_jac=p["prx_jac"], grad1=p["grad1"], **({} if os.environ.get("DFX_BENCH_NOVALID") else dict(valid0=p["valid0"]))))
    arr = al.make_pairs(pairs)
    isz = dfx.item_size(12 + CS)
    items = torch.zeros(P * isz, dtype=torch.uint8, device=dev)
    neq = NormalEquations(world * P + 1, CS, dev)

    def step():
        # hot path: one launch over P pairs; its finalize kernel also scatter-adds the items into the normal-equation
        # blocks of this rank's pairs (dfx_sfm_step_batch_neq_async = RunStepBatchAsync + assemble_native, fused)
        al.RunStepBatchAssembleAsync(arr, items, neq, rank * P)
        if dist is not None:
            neq.reduce(dist, root=0)                           # RCCL reduce over xGMI onto the rank that solves

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, untimed and not part of the W warm-up steps: the GPU's clocks take ~100 launches to settle after idle (measured:
    # the same kernel runs 4 % slower in the first 10 steps of a process than after 60), so ramp them before anything is counted
    for _ in range(40):
        step()
    barrier()
    for _ in range(a.warmup):
        step()
    barrier()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    n_launch, kern_ms = ctx.profile_read()
    ctx.set_profiling(False)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # sanity (untimed): results are real (inliers > half of the pixels on every pair) ...
    its = al.items_from_bytes(items.cpu().numpy(), CS)
    assert all(it.inliers > 0.5 * W * H for it in its), [it.inliers for it in its]
    # ... and the exchanged system is the sum of all ranks' pairs: every Jtr entry lands in exactly one slot of g, so the
    # checksum of the reduced g on rank 0 must equal the checksum of all ranks' items (catches stale or double-counted blocks)
    local = torch.tensor([sum(float(np.sum(it.Jtr.astype(np.float64))) for it in its),
                          sum(float(np.sum(np.abs(it.Jtr.astype(np.float64)))) for it in its)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(local)
    if rank == 0:
        got = float(neq.g.double().sum())
        assert abs(got - float(local[0])) <= 1e-4 * float(local[1]) + 1e-6, (got, local.tolist())

    if rank == 0:
        evals = world * P * a.steps
        bytes_per_launch = (20 + 4 * CS) * W * H * P          # 148 B/px compulsory at CS=32
        kern_s = kern_ms / 1e3 / max(n_launch, 1)
        achieved = bytes_per_launch / kern_s / 1e9
        flops_per_launch = 2.0 * ((12 + CS) * (13 + CS) / 2 + (12 + CS) + 1) * W * H * P   # JtJ + Jtr + r^2 (FMA = 2)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    tj = json.load(fh)
                if tj.get("pairs") == P and tj.get("width") == W and tj.get("height") == H and tj.get("cs") == CS:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "keyframe-pair residual+Jacobian evals/sec (640x480, 32-code)",
            "value": evals / elapsed,
            "unit": "pair-evals/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] geometry in the batch size of configs[3] (1k pairs / 8 GPUs): {P} independent {W}x{H} pairs per GPU per step, "
                                   f"CS={CS}, SfmAligner::RunStep (SE3+code Jacobians, JtJ/Jtr) in one launch, level 0; "
                                   "+ normal-equation block assembly" + (" + RCCL reduce to rank 0" if world > 1 else ""),
                       "pairs_per_gpu": P, "width": W, "height": H, "code_size": CS,
                       "parallelism": f"pairs sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_sfm_step<2,0>", "kernel_us": kern_s * 1e6, "launches": n_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "fp32_tflops": flops_per_launch / kern_s / 1e12},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, CS, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
