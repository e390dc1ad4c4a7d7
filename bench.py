#!/usr/bin/env python3
"""bench.py -- keyframe-pair residual+Jacobian evaluations per second (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM: a single batched launch of
SfmAligner::RunStep (reference cu_sfmaligner.cpp:149-185) over `--pairs` independent 640x480, 32-code keyframe->frame
pairs (BASELINE.json configs[1] geometry, batched; every pair has its own keyframe so the working set,
pairs x 47 MB = 6 GB, is far beyond the 256 MB Infinity Cache and the sweep is honestly HBM-resident), followed by the
assembly of the pairs' 44x44 systems into the block-sparse Gauss-Newton normal equations of the keyframe graph
(dfx_graph_assemble_async) and -- for N > 1 -- their RCCL reduce over xGMI onto the rank that solves.  The default,
128 pairs per GPU, is the per-GPU shard of BASELINE configs[3] ("~1k pairs sharded across 8 GPUs"); weak scaling.

`python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one process per GPU, RCCL); under an external
torchrun it uses the ranks it is given.

Output: ONE JSON line on rank 0 (the driver contract), with
  roofline      step kernel only: algorithmic bytes (148 B/px x px x pairs per launch) / HIP-event duration measured inside
                the library on the launch stream; `traffic` = memory-side bytes of the same kernel from rocprofv3 PMC request
                counters by size (a counters-only child run of this script; null + reason when rocprofv3 is unavailable)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's host path) built -O3 -march=native on this box:
                median of >= 10 repetitions on all cores (`value`) and on one thread
  configs       secondary, untimed-by-the-driver measurements of the other BASELINE configs: configs[1] as ONE blocking
                single-pair call (the reference's per-factor call pattern), configs[4] (1280x960, 64-code) batched, and -- with
                --window -- configs[3] as a real 64-keyframe / 1024-pair window sharded over the ranks.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

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
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child run (roofline.traffic = null)")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary configs[1] / configs[4] measurements")
    ap.add_argument("--window", action="store_true", help="also measure BASELINE configs[3]: 64 keyframes, 1024 pairs over the ranks")
    ap.add_argument("--schedule", choices=["auto", "static", "dynamic"], default="auto",
                    help="auto: the library's default (DFX_SCHEDULE_AUTO); static / dynamic: force one")
    ap.add_argument("--mfma", choices=["auto", "f32", "bf16x3"], default="auto",
                    help="evaluation mode of the step kernel: auto = the library's default (DFX_MFMA_AUTO); f32 = fp32 fmaf chain; bf16x3 = exact three-way bf16 split")
    ap.add_argument("--deferred-tail", action="store_true",
                    help="run the reduction tail of every step (finalize kernel, graph assembly) on a second stream beside the next step's kernel (dfx_set_tail_stream) instead "
                         "of in order on the launch stream.  Measured on MI355X (profiles/r03_bench_tail_modes.txt): the gap between step time and kernel time falls "
                         "from 38 to 27 us, but the 768 finalize workgroups take CU slots from the next step kernel (+15 us): 1.0233 vs 1.0200 ms per step -- so "
                         "the default stays in order (a higher stream priority for the launch stream does not change that: 1.044 vs 1.038 ms, r03_bench_tail_modes.txt)")
    ap.add_argument("--two-call-tail", action="store_true", help="issue the step and the graph assembly as two library calls (dfx_sfm_step_batch_async + "
                    "dfx_graph_assemble_async: two tail kernels) instead of dfx_sfm_step_batch_assemble_async (the assembly inside the launch's tail kernel)")
    ap.add_argument("--foreign-valid0", action="store_true", help="keep the valid0 maps in torch tensors (memory the library does not own: the step kernel "
                    "then re-reads the map every step, 4 B/px) instead of library-owned images with a 1-bit shadow")
    ap.add_argument("--pmc-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------------
def self_spawn(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(w, h, cs):
    """The oracle (a port of the reference's host loop over DenseSfm<...,TargetHost>, ut_sfmaligner.cpp:307-315), fp32
    accumulate, built -O3 -march=native -ffp-contract=off on THIS box.  Median of >= 10 repetitions of the same 640x480x32
    pair: OpenMP over rows on all host cores (`value`) and one thread (the reference's own host loop is single-threaded)."""
    from deepfactors_amd import synth
    from oracle import dfx_oracle as orc   # cpu_baseline leg only
    orc.build_native()
    n = synth.to_numpy(synth.make_pair(w, h, cs, seed=0xDF02, device="cpu"))
    cores = orc.max_threads()
    args = (n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"])

    def median_rate(threads, reps, budget):
        for _ in range(2):
            orc.sfm_step(*args, accum_f64=False, threads=threads)   # warm-up
        ts, t_start = [], time.perf_counter()
        while len(ts) < reps or (time.perf_counter() - t_start < budget and len(ts) < 10 * reps):
            t0 = time.perf_counter()
            orc.sfm_step(*args, accum_f64=False, threads=threads)
            ts.append(time.perf_counter() - t0)
        return 1.0 / float(np.median(ts)), len(ts), time.perf_counter() - t_start

    allc, n_all, t_all = median_rate(cores, 10, 1.5)
    one, n_one, t_one = median_rate(1, 10, 0.0)
    out = dict(value=allc, unit="pair-evals/s", cores=cores, kind="port", single_thread=one,
               sample=f"one {w}x{h} cs={cs} SfmAligner::RunStep pair, fp32 accumulate, g++ -O3 -march=native -ffp-contract=off: median of {n_all} "
                      f"repetitions with OpenMP over rows on {cores} threads ({t_all:.1f} s), median of {n_one} repetitions on 1 thread ({t_one:.1f} s)")
    # beside the port: the REFERENCE'S OWN per-pixel code (oracle/_ref: its unmodified dense_sfm.h / warping.h ... compiled against stand-in
    # Eigen / Sophus / VisionCore headers, g++ -O2, the host loop of ut_sfmaligner.cpp:307-315, single-threaded like the reference) where the
    # prebuilt library travelled with the snapshot; the same pair
    try:
        from oracle import dfx_ref
        if dfx_ref.available():
            dfx_ref.sfm_step(*args)
            ts = []
            while len(ts) < 7:
                t0 = time.perf_counter()
                dfx_ref.sfm_step(*args)
                ts.append(time.perf_counter() - t0)
            out["reference_code_single_thread"] = 1.0 / float(np.median(ts))
            out["sample"] += f"; reference_code_single_thread: oracle/_ref (the reference's own DenseSfm host loop, g++ -O2, stand-in Eigen), median of {len(ts)}"
    except Exception as e:   # noqa: BLE001 -- an extra figure, never the line
        out["reference_code_error"] = f"{type(e).__name__}: {e}"
    return out


def build_pairs(dfx, synth, dev, rank, P, W, H, CS, same=False, ctx=None):
    """P synthetic pairs.  With `ctx` every keyframe's valid map (kf->pyr_vld) is an image owned through the library, as the host layer's
    keyframes keep it (include/dfx_host.hpp, Keyframe::pyr_vld): zero-filled, so the first step writes 1.0 at every inlier and the
    steady state writes nothing -- and the library-owned map lets the step kernel consult its 1-bit shadow instead of re-reading it."""
    pairs, keep = [], []
    for k in range(P):
        p = synth.make_pair(W, H, CS, seed=0xDF02 + (0 if same else 1000 * rank + k), device=dev, motion_scale=(1.0 if same else 0.6 + 0.05 * (k % 8)))
        if ctx is not None:
            p["valid0"] = ctx.alloc_image(W, H)
        keep.append(p)
        pairs.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"],
                          prx0_jac=p["prx_jac"], grad1=p["grad1"], valid0=p["valid0"]))
    return pairs, keep


def pmc_traffic(a):
    """Memory-side traffic of the step kernel from the TCC request counters by size (rocprofv3, counters only, own process so that
    the timed run above is never profiled).  Returns (bytes per launch or None, detail dict)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"source": "unavailable: rocprofv3 not found"}
    out = tempfile.mkdtemp(prefix="dfx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    worker = [sys.executable, os.path.abspath(__file__), "--pmc-worker", "--pairs", str(a.pairs), "--width", str(a.width), "--height", str(a.height),
              "--cs", str(a.cs), "--step-blocks", str(a.step_blocks), "--schedule", a.schedule, "--mfma", a.mfma] + (["--foreign-valid0"] if a.foreign_valid0 else [])
    passes = {"rd": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
              "wr": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]}
    vals = {}
    try:
        for name, ctrs in passes.items():
            cmd = [exe, "--pmc"] + ctrs + ["--kernel-include-regex", "k_sfm_step", "--output-format", "csv", "-d", os.path.join(out, name), "-o", "pmc", "--"] + worker
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"source": f"unavailable: rocprofv3 pass '{name}' rc={r.returncode}"}
            acc, cnt = {}, {}
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if "k_sfm_step" not in row.get("Kernel_Name", ""):
                        continue
                    acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    cnt[row["Counter_Name"]] = cnt.get(row["Counter_Name"], 0) + 1
            for k in acc:
                vals[k] = acc[k] / cnt[k]
        rd = 32 * vals["TCC_EA0_RDREQ_32B_sum"] + 64 * vals["TCC_EA0_RDREQ_64B_sum"] + 128 * vals["TCC_EA0_RDREQ_128B_sum"]
        other = vals["TCC_EA0_RDREQ_sum"] - vals["TCC_EA0_RDREQ_32B_sum"] - vals["TCC_EA0_RDREQ_64B_sum"] - vals["TCC_EA0_RDREQ_128B_sum"]
        rd += 64 * max(other, 0.0)
        wr = 64 * vals["TCC_EA0_WRREQ_64B_sum"] + 32 * max(vals["TCC_EA0_WRREQ_sum"] - vals["TCC_EA0_WRREQ_64B_sum"], 0.0)
        return rd + wr, {"source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ_{,64B}_sum x request size, averaged over the step-kernel "
                                   "dispatches of a counters-only child run of this workload (Infinity-Cache hits are included: memory-side requests)",
                         "read_bytes": rd, "write_bytes": wr}
    except Exception as e:   # noqa: BLE001 -- measurement is optional, the bench line is not
        return None, {"source": f"unavailable: {type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(out, ignore_errors=True)


MFMA_NAMES = {0: "fp32 fmaf chain on v_mfma_f32_16x16x4_f32 (DFX_MFMA_F32_CHAIN)",
              1: "exact three-way bf16 split on v_mfma_f32_16x16x32_bf16, fp32 accumulate (DFX_MFMA_BF16X3; fp32-accurate, tests/test_gpu_bf16x3.py)"}


def mode_kernel_us(ctx, launch, bytes_per_launch, warm, steps, mode):
    """Step-kernel time of `launch` with the context pinned to evaluation mode `mode`; the context goes back to DFX_MFMA_AUTO whatever
    happens, and a failure is reported instead of raised (this is a secondary figure of the bench line)."""
    from deepfactors_amd import _lib
    try:
        ctx.set_mfma_mode(mode)
        for _ in range(warm):
            launch()
        ctx.sync()
        ctx.set_profiling(True)
        for _ in range(steps):
            launch()
        n, ms = ctx.profile_read()
        ctx.set_profiling(False)
        ks = ms / 1e3 / max(n, 1)
        return dict(kernel_us=ks * 1e6, algorithmic_gbs=bytes_per_launch / ks / 1e9, frac=bytes_per_launch / ks / 1e9 / HBM_PEAK_GBS, mfma=MFMA_NAMES[ctx.last_mfma_mode()])
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        try:
            ctx.set_profiling(False)
            ctx.set_mfma_mode(_lib.DFX_MFMA_AUTO)
        except Exception:   # noqa: BLE001
            pass


def event_time_us(torch, enqueue, reps, warm):
    """Average device time of `enqueue()` (kernels on the current torch stream = the context's stream) from a pair of events around
    `reps` back-to-back enqueues, after `warm` untimed ones."""
    for _ in range(warm):
        enqueue()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        enqueue()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def small_operator_rooflines(dfx, synth, ctx, dev):
    """HBM-resident sweeps of the operators beside the SfM step (SURVEY 8d byte counts): the code-Jacobian decoder UpdateDepth (136 B/px at
    CS = 32) over 64 distinct keyframes (2.7 GB), SE3Aligner::RunStep (20 B/px) and SfmAligner::EvaluateError (12 B/px) over 128 distinct
    pairs in one launch each.  Times are event pairs around back-to-back enqueues on the context's stream: they include the operator's
    finalize kernel and its 10-20 KB descriptor upload, i.e. they are a few microseconds pessimistic for the kernel itself."""
    import torch
    W, H, CS = 640, 480, 32
    out = {}
    K = 64
    kfs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev) for k in range(K)]
    codes = np.stack([np.asarray(k["code"], np.float32) for k in kfs])
    outs = [torch.empty_like(k["img0"]) for k in kfs]
    us = event_time_us(torch, lambda: dfx.UpdateDepthBatch(codes, [k["prx_orig"] for k in kfs], [k["prx_jac"] for k in kfs], 2.0, outs, ctx=ctx), reps=40, warm=150)
    byts = (8 + 4 * CS) * W * H * K
    out["update_depth_batch_64kf"] = dict(us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                          note="k_update_depth_batch<32>: dpt = a / (prx + jac . code) - a for 64 distinct 640x480 keyframes in one launch (2.7 GB)")
    P = 128
    al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    prs = [kfs[k % K] for k in range(P)]
    # 128 distinct (img0, img1, dpt0, grad1) sets: the 64 keyframes plus clones (distinct HBM)
    extra = [{n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in kfs[k].items()} for k in range(P - K)]
    prs = kfs + extra
    sarr = se3.make_pairs([dict(se3=synth.IDENTITY, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
    sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
    us = event_time_us(torch, lambda: se3.RunStepBatch(sarr, sitems), reps=60, warm=300)
    byts = 20 * W * H * P
    out["se3_step_batch_128pairs"] = dict(us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                          note="k_se3_step_batch + finalize: 128 distinct 640x480 pairs in one launch (786 MB: beyond the 256 MB Infinity Cache)")
    earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                               grad1=p["grad1"]) for p in prs])
    eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
    us = event_time_us(torch, lambda: al.EvaluateErrorBatch(earr, eitems), reps=60, warm=300)
    byts = 12 * W * H * P
    out["sfm_error_batch_128pairs"] = dict(us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                           note="k_sfm_error_batch + finalize: 128 distinct 640x480 pairs in one launch (472 MB)")
    return out


def secondary_configs(dfx, synth, ctx, dev):
    """configs[1] as the reference calls it (ONE pair per blocking call, photometric_factor.cpp:267-274), its 3-level pyramid variant, configs[4]
    (1280x960, 64-code; 16 pairs per launch, 5.5 GB working set)."""
    import torch
    out = {}
    # ---- configs[1]: single 640x480x32 pair, blocking dfx_sfm_step
    al = dfx.SfmAligner(code_size=32, ctx=ctx)
    p = synth.make_pair(640, 480, 32, seed=0xDF02, device=dev)
    call = lambda: al.RunStep(p["pose0"], p["pose1"], None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, p["valid0"], p["prx_jac"], p["grad1"])  # noqa: E731
    for _ in range(2000):
        call()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    n, ms = ctx.profile_read()
    ctx.set_profiling(False)
    out["configs1_single_pair_blocking"] = dict(call_us=dt * 1e6, kernel_us=ms / n * 1e3, evals_per_s=1.0 / dt,
                                                note="one 640x480 cs=32 pair per blocking SfmAligner::RunStep call (46 MB: Infinity-Cache resident, not an HBM figure)")
    del p
    # ---- SURVEY 8d "all pyramid levels" variant: levels 0, 1, 2 (640x480, 320x240, 160x120) of 128 factor sets -- 384 pairs -- in ONE launch
    # (pairs of several image sizes share a launch: workgroups in proportion to the pixel count, large pairs first), and level by level
    P3 = 128
    lv_pairs, lv_keep = [], []
    for (w, h) in ((640, 480), (320, 240), (160, 120)):
        pr, kp = build_pairs(dfx, synth, dev, 3, P3, w, h, 32, ctx=ctx)
        lv_pairs.append(pr); lv_keep.append(kp)
    items = torch.zeros(3 * P3 * dfx.item_size(12 + 32), dtype=torch.uint8, device=dev)

    def kernel_us(arr, warm, reps):
        for _ in range(warm):
            al.RunStepBatchAsync(arr, items)
        ctx.sync()
        ctx.set_profiling(True)
        for _ in range(reps):
            al.RunStepBatchAsync(arr, items)
        nl, ms = ctx.profile_read()
        ctx.set_profiling(False)
        return ms / nl * 1e3
    lv_us = [kernel_us(al.make_pairs(pr), 300 if k == 0 else 1200, 30) for k, pr in enumerate(lv_pairs)]
    one_us = kernel_us(al.make_pairs([p for pr in lv_pairs for p in pr]), 300, 30)
    px3 = (640 * 480 + 320 * 240 + 160 * 120) * P3
    out["configs1_pyramid3_128pairs"] = dict(one_launch_kernel_us=one_us, evals_per_s=P3 / (one_us * 1e-6), algorithmic_gbs=148 * px3 / (one_us * 1e-6) / 1e9,
                                             frac=148 * px3 / (one_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                             level_by_level_kernel_us=lv_us, level_by_level_frac=148 * px3 / (sum(lv_us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                             note="SfmAligner::RunStep over pyramid levels 0-2 of 128 factor sets: one 'evaluation' = all three levels; ONE launch over the 384 "
                                                  "pairs (dfx_sfm_step_batch_async accepts pairs of several image sizes) vs one launch per level; step kernels only")
    del lv_pairs, lv_keep, items
    # ---- configs[4]: 1280x960, cs = 64
    W, H, CS, P = 1280, 960, 64, 16
    al4 = dfx.SfmAligner(code_size=CS, ctx=ctx)
    pairs, keep = build_pairs(dfx, synth, dev, 7, P, W, H, CS, ctx=ctx)
    arr = al4.make_pairs(pairs)
    items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
    for _ in range(250):
        al4.RunStepBatchAsync(arr, items)
    ctx.sync()
    ctx.set_profiling(True)
    for _ in range(20):
        al4.RunStepBatchAsync(arr, items)
    n, ms = ctx.profile_read()
    ctx.set_profiling(False)
    kern_s = ms / 1e3 / n
    bpl = (20 + 4 * CS) * W * H * P
    out["configs4_1280x960_cs64"] = dict(pairs_per_launch=P, kernel_us=kern_s * 1e6, algorithmic_gbs=bpl / kern_s / 1e9, frac=bpl / kern_s / 1e9 / HBM_PEAK_GBS,
                                         evals_per_s=P / kern_s, mfma=MFMA_NAMES[ctx.last_mfma_mode()] + " -- the library default (DFX_MFMA_AUTO)")
    # the same batch pinned to the fp32 chain (matrix-bound at CS = 64: the reason DFX_MFMA_AUTO picks the split)
    from deepfactors_amd import _lib as _dl4
    out["configs4_1280x960_cs64"]["f32_chain"] = mode_kernel_us(ctx, lambda: al4.RunStepBatchAsync(arr, items), bpl, warm=150, steps=20, mode=_dl4.DFX_MFMA_F32_CHAIN)
    del pairs, keep, arr, items
    # ---- configs[2] as the reference's relinearisation round (PhotometricFactor::RunAlignmentStep, photometric_factor.cpp:225-293, for every
    # factor of a 16-keyframe window): UpdateDepth once per keyframe whose code moved + one batched RunStep over the 120 pairs
    from deepfactors_amd.dist import PairGraph
    W, H, CS, K = 640, 480, 32, 16
    graph = PairGraph.all_pairs(K, both_directions=False)
    al2 = dfx.SfmAligner(code_size=CS, ctx=ctx)
    kfs = [synth.make_pair(W, H, CS, seed=0x1600 + k, device=dev) for k in range(K)]
    plist, prx, codes = [], [], []
    for (i, j) in graph.pairs:
        a, b = kfs[int(i)], kfs[int(j)]
        plist.append(dict(pose0=a["pose0"], pose1=b["pose1"], cam=a["cam"], img0=a["img0"], img1=b["img0"], dpt0=a["dpt0"], valid0=a["valid0"],
                          prx0_jac=a["prx_jac"], grad1=b["grad1"]))
        prx.append(a["prx_orig"])
        codes.append(np.asarray(a["code"].cpu() if hasattr(a["code"], "cpu") else a["code"], np.float32))
    arr = al2.make_pairs(plist)
    items = torch.zeros(len(plist) * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
    codes = np.stack(codes)
    for _ in range(200):
        al2.LinearizeBatch(arr, prx, codes, items)
    ctx.sync()
    reps = 40
    t0 = time.perf_counter()
    for _ in range(reps):
        al2.LinearizeBatch(arr, prx, codes, items)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    byts = ((8 + 4 * CS) * K + (20 + 4 * CS) * len(plist)) * W * H
    out["configs2_linearize_16kf_120pairs"] = dict(round_us=dt * 1e6, evals_per_s=len(plist) / dt, algorithmic_gbs=byts / dt / 1e9, frac=byts / dt / 1e9 / HBM_PEAK_GBS,
                                                   note="UpdateDepth of the 16 keyframes (136 B/px each) + one batched RunStep of the 120 pairs (148 B/px each) per round, wall clock "
                                                        "of enqueue-to-completion; the 16 Jacobian images (630 MB) are shared by the pairs, so part of the stream is Infinity-Cache/L2 traffic")
    return out


def window_config(dfx, synth, ctx, dev, dist, rank, world):
    """BASELINE configs[3]: 64 keyframes (replicated on every rank), each linked to its 16 nearest -> 1024 directed pairs, sharded
    contiguously (by source keyframe) over the ranks; per step: one batched launch per rank + graph assembly + RCCL reduce."""
    import torch
    from deepfactors_amd.dist import NormalEquations, PairGraph, shard_range
    W, H, CS, K = 640, 480, 32, 64
    graph = PairGraph.window(K, 16)
    al = dfx.SfmAligner(code_size=CS, ctx=ctx)
    rng = np.random.default_rng(0xDF03)
    kfs = [synth.make_pair(W, H, CS, seed=0x6400 + k, device=dev) for k in range(K)]
    poses = []
    for k in range(K):
        R = synth.so3_exp(rng.normal(0, 0.004, 3))
        poses.append(synth.pose_qt(R, rng.normal(0, 0.008, 3)))
    lo, hi = shard_range(graph.n_pairs, rank, world)
    plist = []
    for (i, j) in graph.pairs[lo:hi]:
        a, b = kfs[int(i)], kfs[int(j)]
        plist.append(dict(pose0=poses[int(i)], pose1=poses[int(j)], cam=a["cam"], img0=a["img0"], img1=b["img0"], dpt0=a["dpt0"], valid0=a["valid0"],
                          prx0_jac=a["prx_jac"], grad1=b["grad1"]))
    arr = al.make_pairs(plist)
    items = torch.zeros((hi - lo) * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
    neq = NormalEquations(graph, CS, dev)

    def step():
        al.RunStepBatchAssembleAsync(arr, items, neq, lo)
        if dist is not None:
            neq.reduce(dist, root=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(30):
        step()
    barrier()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        step()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    return dict(keyframes=K, pairs=graph.n_pairs, pairs_per_rank=hi - lo, ms_per_step=el / steps * 1e3, evals_per_s=graph.n_pairs * steps / el,
                system_bytes=int(neq.buf.numel() * 4),
                note="keyframe pyramids replicated; 16 pairs share each keyframe's 39 MB Jacobian, so this configuration re-reads from L2 / Infinity Cache "
                     "(not an HBM-roofline figure)")


def run_protocol(a, dist, dev, ctx, step, barrier, P):
    """The measurement protocol around `step` (one pass of the hot path): clock ramp in windows until the kernel time has settled, W
    warm-up steps, K timed steps between barriers, MAX over ranks.  The decision that steers the control flow (leaving the ramp) is agreed
    between the ranks by a collective, so all ranks issue the same sequence of steps and collectives (tests/test_bench_protocol.py runs it
    on two gloo ranks with fakes).  Schedule and evaluation mode are the library's defaults unless forced on the command line: nothing is
    probed here.  Returns a dict."""
    import torch
    # setup, untimed and not part of the W warm-up steps: after idle the GPU needs ~0.15 s of sustained work to reach its steady clocks
    # (tools/clock_series.py, 128-pair steps: launches 0-49 average 1287 us, 50-99 1144 us, from 100 on 1065 +- 5 us for thousands
    # of launches) and some boxes take longer, so the same steps run in windows until the step kernel's time has settled: three
    # consecutive windows within 1 %, at least 6 windows (~0.35 s), at most 40 (~2.3 s).  All ranks leave the loop together.
    win = max(25, 6400 // max(P, 1))
    ramp_steps, hist = 0, []
    ctx.set_profiling(True)
    for w_i in range(40):
        for _ in range(win):
            step()
        barrier()
        n_w, ms_w = ctx.profile_read()
        hist.append(ms_w / max(n_w, 1))
        ramp_steps += win
        settled = w_i >= 5 and max(hist[-3:]) <= 1.01 * min(hist[-3:])
        go_on = torch.tensor([0 if settled else 1], dtype=torch.int32, device=dev)
        if dist is not None:
            dist.all_reduce(go_on, op=dist.ReduceOp.MAX)
        if int(go_on.item()) == 0:
            break
    ctx.set_profiling(False)
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
    n_launch, kern_ms, kern_min, kern_max = ctx.profile_read_ex()
    ctx.set_profiling(False)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    return dict(hist=hist, ramp_steps=ramp_steps, elapsed=elapsed, n_launch=n_launch, kern_ms=kern_ms, kern_min_ms=kern_min, kern_max_ms=kern_max)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and not a.pmc_worker:
        self_spawn(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
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
    from deepfactors_amd import _lib as _dl
    from deepfactors_amd import synth
    from deepfactors_amd.dist import NormalEquations, PairGraph, PipelinedReduce

    W, H, CS, P = a.width, a.height, a.cs, a.pairs
    ctx = dfx.Context(local)
    # library defaults unless forced: explicit calls, so that stray DFX_MFMA / DFX_SCHEDULE environment variables cannot steer the line
    ctx.set_mfma_mode({"auto": _dl.DFX_MFMA_AUTO, "f32": _dl.DFX_MFMA_F32_CHAIN, "bf16x3": _dl.DFX_MFMA_BF16X3}[a.mfma])
    ctx.set_schedule({"auto": _dl.DFX_SCHEDULE_AUTO, "static": _dl.DFX_SCHEDULE_STATIC, "dynamic": _dl.DFX_SCHEDULE_DYNAMIC}[a.schedule])
    al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=a.step_blocks), code_size=CS, ctx=ctx)

    # ---- synthetic, device-resident input: P distinct keyframe->frame pairs per rank
    pairs, keep = build_pairs(dfx, synth, dev, rank, P, W, H, CS, same=bool(os.environ.get("DFX_BENCH_SAME")), ctx=None if a.foreign_valid0 else ctx)
    arr = al.make_pairs(pairs)
    isz = dfx.item_size(12 + CS)
    items = torch.zeros(P * isz, dtype=torch.uint8, device=dev)

    if a.pmc_worker:   # counters-only child of pmc_traffic(): a few launches of the same workload, nothing else
        for _ in range(4):
            al.RunStepBatchAsync(arr, items)
        ctx.sync()
        return

    # --deferred-tail: consecutive steps are independent batches, so the reduction tail of step k (finalize kernel + graph assembly, ~35 us
    # of short dependent kernels) can run on a second stream beside the 1 ms step kernel of step k + 1 (dfx_set_tail_stream); for N > 1 the
    # RCCL reduce of step k is then issued on that stream too.  Every tail and every reduce has completed when the timed region ends.
    tail = torch.cuda.Stream(device=dev) if a.deferred_tail else None
    if tail is not None:
        ctx.set_tail_stream(tail)

    # the pairs of all ranks form one trajectory: pair p links keyframe node p -> frame node p + 1
    graph = PairGraph.chain(world * P)
    pipe = PipelinedReduce(dist, [NormalEquations(graph, CS, dev) for _ in range(2)], root=0, stream=tail) if world > 1 else None
    neq = NormalEquations(graph, CS, dev) if pipe is None else None

    fused = not a.two_call_tail

    def step():
        # hot path: one launch over P pairs (+ its finalize kernel), then this rank's items are summed into the block-sparse
        # normal equations of the graph; for N > 1 the ranks' buffers are reduced onto the rank that solves
        if pipe is not None:
            al.RunStepBatchAssembleAsync(arr, items, pipe.next(), rank * P, fused=fused)
            pipe.submit()                                      # RCCL reduce over xGMI, overlapped with the next step's kernels
            return
        al.RunStepBatchAssembleAsync(arr, items, neq, rank * P, fused=fused)
        if dist is not None:
            if tail is not None:
                with torch.cuda.stream(tail):
                    neq.reduce(dist, root=0)
            else:
                neq.reduce(dist, root=0)

    def barrier():
        if pipe is not None:
            pipe.drain()
        ctx.tail_join()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pr = run_protocol(a, dist, dev, ctx, step, barrier, P)
    hist, ramp_steps, elapsed, n_launch, kern_ms = pr["hist"], pr["ramp_steps"], pr["elapsed"], pr["n_launch"], pr["kern_ms"]

    # sanity (untimed): results are real (inliers > half of the pixels on every pair) ...
    its = al.items_from_bytes(items.cpu().numpy(), CS)
    assert all(it.inliers > 0.5 * W * H for it in its), [it.inliers for it in its]
    # ... and the exchanged system is the sum of all ranks' pairs: every Jtr entry lands in exactly one slot of g, so the
    # checksum of the reduced g on rank 0 must equal the checksum of all ranks' items (catches stale or double-counted blocks)
    chk = torch.tensor([sum(float(np.sum(it.Jtr.astype(np.float64))) for it in its),
                        sum(float(np.sum(np.abs(it.Jtr.astype(np.float64)))) for it in its)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(chk)
    if rank == 0:
        got = float((neq if pipe is None else pipe.last()).g.double().sum())
        assert abs(got - float(chk[0])) <= 1e-4 * float(chk[1]) + 1e-6, (got, chk.tolist())

    out = None
    mode_ran, dyn_ran = ctx.last_mfma_mode(), ctx.last_schedule_dynamic()
    kern_s = kern_ms / 1e3 / max(n_launch, 1)
    if rank == 0:
        evals = world * P * a.steps
        bytes_per_launch = (20 + 4 * CS) * W * H * P          # 148 B/px compulsory at CS=32
        achieved = bytes_per_launch / kern_s / 1e9
        flops_per_launch = 2.0 * ((12 + CS) * (13 + CS) / 2 + (12 + CS) + 1) * W * H * P   # JtJ + Jtr + r^2 (FMA = 2)
        out = {
            "metric": "keyframe-pair residual+Jacobian evals/sec (640x480, 32-code)",
            "value": evals / elapsed,
            "unit": "pair-evals/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ramp_steps": ramp_steps,
            "ramp_kernel_us": [round(h * 1e3, 1) for h in hist],
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] geometry in the batch size of configs[3] (1k pairs / 8 GPUs): {P} independent {W}x{H} pairs per GPU per step, "
                                   f"CS={CS}, SfmAligner::RunStep (SE3+code Jacobians, JtJ/Jtr) in one launch, level 0; "
                                   "+ block-sparse normal-equation assembly" + (" + RCCL reduce to rank 0" if world > 1 else "")
                                   + ("; the reduction tail of step k (finalize, assembly" + (", reduce" if world > 1 else "") + ") runs on a second stream beside the kernel of step k + 1"
                                      if tail is not None else ""),
                       "pairs_per_gpu": P, "width": W, "height": H, "code_size": CS,
                       "parallelism": f"pairs sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": f"k_sfm_step<NCB={CS // 16}, {'bf16x3' if mode_ran == _dl.DFX_MFMA_BF16X3 else 'f32 chain'}, {'dynamic' if dyn_ran else 'static'}>",
                         "kernel_us": kern_s * 1e6, "kernel_us_min": pr["kern_min_ms"] * 1e3, "kernel_us_max": pr["kern_max_ms"] * 1e3, "launches": n_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "fp32_tflops": flops_per_launch / kern_s / 1e12,
                         "schedule": ("dynamic item queues (results reproducible to fp32 re-association)" if dyn_ran else "static partition (bit-reproducible)")
                                     + (" -- the library default" if a.schedule == "auto" else " -- forced by --schedule"),
                         "mfma": MFMA_NAMES[mode_ran] + (" -- the library default (DFX_MFMA_AUTO)" if a.mfma == "auto" else " -- forced by --mfma")},
        }
    # the secondary measurements below run the library's defaults, in order on one stream
    ctx.set_tail_stream(None)
    ctx.set_schedule(_dl.DFX_SCHEDULE_AUTO)
    ctx.set_mfma_mode(_dl.DFX_MFMA_AUTO)
    configs = {}
    if world == 1 and not a.no_configs:
        # the timed workload once more, pinned to the evaluation mode the line did NOT run
        other = _dl.DFX_MFMA_F32_CHAIN if mode_ran == _dl.DFX_MFMA_BF16X3 else _dl.DFX_MFMA_BF16X3
        configs["headline_workload_other_mode"] = mode_kernel_us(ctx, lambda: al.RunStepBatchAsync(arr, items), (20 + 4 * CS) * W * H * P, warm=150, steps=30, mode=other)
    del keep, pairs, arr
    torch.cuda.empty_cache()
    if world == 1 and not a.no_configs:
        configs.update(secondary_configs(dfx, synth, ctx, dev))
        torch.cuda.empty_cache()
        configs.update(small_operator_rooflines(dfx, synth, ctx, dev))
        torch.cuda.empty_cache()
    if a.window or (world == 1 and not a.no_configs):
        configs["configs3_window64"] = window_config(dfx, synth, ctx, dev, dist, rank, world)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if configs:
            out["configs"] = configs
        if world == 1:
            del ctx
            torch.cuda.synchronize()
            if a.no_traffic:
                out["roofline"]["traffic_source"] = "skipped (--no-traffic)"
            else:
                traffic, detail = pmc_traffic(a)
                out["roofline"]["traffic"] = traffic
                out["roofline"]["traffic_source"] = detail.pop("source")
                out["roofline"].update({f"traffic_{k}": v for k, v in detail.items()})
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(W, H, CS)
        else:
            out["roofline"]["traffic_source"] = "not collected for N > 1 (per-rank kernels are identical to the N = 1 run)"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
