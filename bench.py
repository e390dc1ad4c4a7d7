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
                    help="run the reduction tail of every step (tail kernel with the graph assembly) on a second stream beside the next step's kernel (dfx_set_tail_stream) "
                         "instead of in order on the launch stream.  Measured on MI355X in interleaved windows of one process (tools/ab_tail_modes.py, profiles/r05_step_gap.txt): "
                         "step time - kernel time falls from 39 to 15-21 us, but the step kernel beside which the tail runs takes 13-19 us longer (the tail's 47 MB of "
                         "partials and its issue slots are paid there): 985-990 vs 991 us per step, -0.5 %%, at a roofline fraction 1.9 %% lower.  So the single-GPU line keeps "
                         "the tail in order; with the C-ABI exchange and N > 1 the mode is on (it is how the collective leaves the launch stream)")
    ap.add_argument("--two-call-tail", action="store_true", help="issue the step and the graph assembly as two library calls (dfx_sfm_step_batch_async + "
                    "dfx_graph_assemble_async: two tail kernels) instead of dfx_sfm_step_batch_assemble_async (the assembly inside the launch's tail kernel)")
    ap.add_argument("--foreign-valid0", action="store_true", help="keep the valid0 maps in torch tensors (memory the library does not own: the step kernel "
                    "then re-reads the map every step, 4 B/px) instead of library-owned images with a 1-bit shadow")
    ap.add_argument("--exchange", choices=["cabi", "torch"], default="cabi",
                    help="how the ranks' normal equations are summed when the script runs under torch.distributed.run: cabi (default) = the SHIPPED exchange, "
                         "dfx_comm_* of the C ABI (deepfactors_amd/csrc/dfx_comm.cpp: ncclReduce enqueued on the context's exchange stream; the communicator is created "
                         "from a unique id handed round over the process group that launched the ranks); torch = torch.distributed's collectives on the same buffers (A/B)")
    ap.add_argument("--workload", choices=["truth", "perturbed", "unrelated"], default="perturbed",
                    help="what the timed pairs look like: perturbed (default since round 6) = pose1 of every pair moved by N(0, 5 mm) / N(0, 0.3 deg) per axis "
                         "(SURVEY 8d cfg 3: what a relinearisation sees); truth = every pair at its generating pose (residual ~ 0, Huber never active, taps maximally "
                         "coherent: the line of rounds 1-5); unrelated = perturbed poses AND img1 / grad1 of ANOTHER scene (residuals of the order of huber_delta).  "
                         "The line's `config.workload` names it and `configs.headline_<other>` carry the two it did not run: they differ by < 1 %% on MI355X")
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


def parity_probe(dfx, synth, ctx, dev, W, H, CS):
    """GPU side of `parity_blocks`: ONE pair of the headline geometry (seed 0xDF02, pose1 moved by 1 cm so that the gradient is not ~0) through
    SfmAligner::RunStep in BOTH evaluation modes.  No oracle here -- the items are compared in the cpu_baseline leg (parity_blocks)."""
    from deepfactors_amd import _lib
    p = synth.make_pair(W, H, CS, seed=0xDF02, device=dev)
    pose1 = np.asarray(p["pose1"], np.float32).copy()
    pose1[4] += 0.01
    got = {}
    try:
        for name, mode in (("f32_chain", _lib.DFX_MFMA_F32_CHAIN), ("bf16x3", _lib.DFX_MFMA_BF16X3)):
            ctx.set_mfma_mode(mode)
            al = dfx.SfmAligner(code_size=CS, ctx=ctx)
            it = al.RunStep(p["pose0"], pose1, None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, None, p["prx_jac"], p["grad1"])
            got[name] = dict(JtJ=np.array(it.JtJ, np.float64), Jtr=np.array(it.Jtr, np.float64), residual=float(it.residual), inliers=int(it.inliers))
    finally:
        ctx.set_mfma_mode(_lib.DFX_MFMA_AUTO)
    return dict(items=got, pose1=pose1, w=W, h=H, cs=CS)


def parity_blocks(probe):
    """cpu_baseline leg: the probe's GPU items against the fp64-accumulating oracle on the same seeded pair, PER BLOCK of the 12 + CS system
    (the six G blocks and three gradients GTSAM receives, photometric_factor.cpp:135-161).  `cs` = max over the block's entries of
    |got_ij - ref_ij| / sqrt(ref_ii ref_jj) (Jtr: / sqrt(ref_ii * sum r^2)); `blk` = max|got - ref| / max|ref| over the block -- the
    comparison of tests/helpers.py, whose tolerance is 1e-4 per entry."""
    from types import SimpleNamespace
    from deepfactors_amd import synth
    from oracle import dfx_oracle as orc   # cpu_baseline leg only
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import block_errors
    W, H, CS = probe["w"], probe["h"], probe["cs"]
    n = synth.to_numpy(synth.make_pair(W, H, CS, seed=0xDF02, device="cpu"))
    ref = orc.sfm_step(n["pose0"], probe["pose1"], n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], accum_f64=True, threads=min(orc.max_threads(), 16))
    out = {"tolerance": 1e-4, "pair": f"{W}x{H} cs={CS} seed 0xDF02, pose1.tx + 1 cm", "oracle_inliers": int(ref.inliers)}
    for mode, it in probe["items"].items():
        e = block_errors(SimpleNamespace(**it), ref)
        out[mode] = {k: {"cs": float(f"{v['cs']:.3g}"), "blk": float(f"{v['blk']:.3g}"), "scale": float(f"{v['scale']:.4g}")} for k, v in e.items()}
        out[mode]["inliers"] = it["inliers"]
        out[mode]["worst_cs"] = max(v["cs"] for v in e.values())
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


def apply_workload(al, synth, pairs, kind, seed=0x5EED):
    """Turn the truth-pose batch into the `perturbed` / `unrelated` workload IN PLACE (no new device memory) and return the pair array.
    perturbed: pose1 <- (t + N(0, 5 mm), exp(N(0, 0.3 deg)) R) per pair, seeded (SURVEY 8d cfg 3).  unrelated: additionally pair k reads the
    img1 / grad1 of pair k + 1 -- every pair has its own texture seed, so that is another scene: residuals of the order of huber_delta (the synthetic
    textures are low-contrast: rms 0.1), the Huber branch on part of the pixels, gradients uncorrelated with the residual."""
    if kind == "truth":
        return al.make_pairs(pairs)
    rng = np.random.default_rng(seed)
    out = []
    for k, q in enumerate(pairs):
        q = dict(q)
        p1 = np.asarray(q["pose1"], np.float64)
        R = synth.so3_exp(rng.normal(0.0, np.deg2rad(0.3), 3)) @ synth.quat_to_R(p1[:4])
        q["pose1"] = synth.pose_qt(R, p1[4:] + rng.normal(0.0, 0.005, 3))
        if kind == "unrelated":
            o = pairs[(k + 1) % len(pairs)]
            q["img1"], q["grad1"] = o["img1"], o["grad1"]
        out.append(q)
    return al.make_pairs(out)


def pmc_traffic(a, workload=None):
    """Memory-side traffic of the step kernel from the TCC request counters by size (rocprofv3, counters only, own process so that
    the timed run above is never profiled).  Returns (bytes per launch or None, detail dict)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"source": "unavailable: rocprofv3 not found"}
    out = tempfile.mkdtemp(prefix="dfx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    worker = [sys.executable, os.path.abspath(__file__), "--pmc-worker", "--pairs", str(a.pairs), "--width", str(a.width), "--height", str(a.height),
              "--cs", str(a.cs), "--step-blocks", str(a.step_blocks), "--schedule", a.schedule, "--mfma", a.mfma, "--workload", workload or a.workload] + (["--foreign-valid0"] if a.foreign_valid0 else [])
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


def kernel_only_us(ctx, enqueue, reps=30):
    """Average duration of the bracketed main kernel of `enqueue` (dfx_set_profiling: HIP events on the context's stream around the reduction
    kernel alone -- what rocprofv3's kernel trace shows for it, without the finalize kernel and the two launch boundaries of a call)."""
    ctx.set_profiling(True)
    ctx.profile_read()
    for _ in range(reps):
        enqueue()
    n, ms = ctx.profile_read()
    ctx.set_profiling(False)
    return ms * 1e3 / max(n, 1)


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
    # frame ingest: Frame::FillPyramids for 64 frames of 640x480, 4 levels, in ONE enqueue (dfx_build_pyramid_batch_async: a launch per level
    # over all frames; a level is read once, its Sobel gradient and its blur-down come from the same LDS tile).  Algorithmic bytes per level-i pixel:
    # 4 read + 8 written (gradient) + 1 written (a quarter pixel of the next level; not for the last level)
    F, LV = 64, 4
    fh, fw = kfs[0]["img0"].shape
    pyr_i = [[torch.empty((fh >> i, fw >> i), dtype=torch.float32, device=dev) for i in range(LV)] for _ in range(F)]
    pyr_g = [[torch.empty((fh >> i, fw >> i, 2), dtype=torch.float32, device=dev) for i in range(LV)] for _ in range(F)]
    for k in range(F):
        pyr_i[k][0].copy_(kfs[k % K]["img0"])
    parr = dfx.make_pyramids(pyr_i, pyr_g)                   # (the buffers of a frame ring are marshalled once, not per frame)
    us = event_time_us(torch, lambda: dfx.BuildPyramids(parr, ctx=ctx), reps=40, warm=100)
    byts = sum((fw >> i) * (fh >> i) * (12 + (1 if i + 1 < LV else 0)) for i in range(LV)) * F
    lv0 = pyr_i[0][0]
    ref1 = torch.empty_like(pyr_i[0][1]); refg = torch.empty_like(pyr_g[0][0])
    dfx.GaussianBlurDown(lv0, ref1, ctx); dfx.SobelGradients(lv0, refg, ctx)
    parr1 = dfx.make_pyramids(pyr_i[:1], pyr_g[:1])
    one = event_time_us(torch, lambda: dfx.BuildPyramids(parr1, ctx=ctx), reps=100, warm=100)
    out["pyramid_build_64frames_4levels"] = dict(us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS, frames_per_s=F / (us * 1e-6),
                                                 single_frame_us=one, equals_per_level_operators=bool(torch.equal(pyr_i[0][1], ref1) and torch.equal(pyr_g[0][0], refg)),
                                                 note="k_pyr_rows (levels 0, 1) + k_pyr_tail (levels 2-3): image + gradient pyramids of 64 distinct 640x480 frames per enqueue (339 MB: beyond the Infinity Cache); "
                                                      "us = the whole enqueue (three launches, no copy command, no event) in back-to-back calls over the same buffers; single_frame_us = one frame per "
                                                      "enqueue (latency-bound: three dependent launches)")
    del pyr_i, pyr_g, parr, parr1
    P = 128
    al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    prs = [kfs[k % K] for k in range(P)]
    # 128 distinct (img0, img1, dpt0, grad1) sets: the 64 keyframes plus clones (distinct HBM)
    extra = [{n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in kfs[k].items()} for k in range(P - K)]
    prs = kfs + extra
    sarr = se3.make_pairs([dict(se3=synth.IDENTITY, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
    sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
    us = event_time_us(torch, lambda: se3.RunStepBatch(sarr, sitems), reps=60, warm=300)
    kus = kernel_only_us(ctx, lambda: se3.RunStepBatch(sarr, sitems))
    byts = 20 * W * H * P
    out["se3_step_batch_128pairs"] = dict(kernel="k_se3_step_batch", us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                          kernel_us=kus, kernel_frac=byts / kus / 1e3 / HBM_PEAK_GBS,
                                          note="k_se3_step_batch + finalize: 128 distinct 640x480 pairs in one launch (786 MB: beyond the 256 MB Infinity Cache); "
                                               "us / frac = the whole call in back-to-back enqueues (reduction kernel + finalize kernel + two launch boundaries), "
                                               "kernel_us / kernel_frac = the reduction kernel alone (HIP events around it: they read 2.4 - 3 us more than the kernel's "
                                               "duration in a rocprofv3 kernel trace, whatever its length; the warmed trace of these kernels -- tools/small_ops_trace.py, "
                                               "last 30 of 430 dispatches per phase -- is profiles/r05_small_ops_trace.csv)")
    # the same launch at the pairs' true relative poses: what a tracker evaluates from its second iteration on.  At the exact identity (the entry above,
    # kept for continuity with rounds 1-3) the outermost pixel columns / rows project ONTO the view border, where the fast geometry defers to the
    # reference-order evaluation: two of the ten 64-pixel bands take that path on every row
    sarr2 = se3.make_pairs([dict(se3=p["pose10_true"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
    us = event_time_us(torch, lambda: se3.RunStepBatch(sarr2, sitems), reps=60, warm=150)
    kus = kernel_only_us(ctx, lambda: se3.RunStepBatch(sarr2, sitems))
    out["se3_step_batch_128pairs_true_pose"] = dict(kernel="k_se3_step_batch", us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                                    kernel_us=kus, kernel_frac=byts / kus / 1e3 / HBM_PEAK_GBS,
                                                    note="as se3_step_batch_128pairs, evaluated at each pair's generating pose instead of the identity")
    earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                               grad1=p["grad1"]) for p in prs])
    eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
    us = event_time_us(torch, lambda: al.EvaluateErrorBatch(earr, eitems), reps=60, warm=300)
    kus = kernel_only_us(ctx, lambda: al.EvaluateErrorBatch(earr, eitems))
    byts = 12 * W * H * P
    out["sfm_error_batch_128pairs"] = dict(kernel="k_sfm_error_batch", us=us, algorithmic_bytes=byts, algorithmic_gbs=byts / us / 1e3, frac=byts / us / 1e3 / HBM_PEAK_GBS,
                                           kernel_us=kus, kernel_frac=byts / kus / 1e3 / HBM_PEAK_GBS,
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
    torch.cuda.empty_cache()
    # ---- configs[4] AS BASELINE STATES IT: "1280x960 input, 4-level pyramid, 64-dim code" -- levels 1280x960 ... 160x120 (the decoder emits one
    # Jacobian per level, core/network/decoder_network.cpp:258-259; pyramid_levels = 4 in core/deepfactors_options.h:43,83) of 16 factor sets: 64 pairs
    # in ONE launch (mixed image sizes: the per-tile finalize kernel in its mixed-size form), and level by level as the reference walks them
    # (tools/kernel_benchmark.cpp:192-203)
    lv4 = ((1280, 960), (640, 480), (320, 240), (160, 120))
    lv_pairs, lv_keep = [], []
    for (w, h) in lv4:
        pr, kp = build_pairs(dfx, synth, dev, 11, P, w, h, CS, ctx=ctx)
        lv_pairs.append(pr); lv_keep.append(kp)
    items = torch.zeros(len(lv4) * P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)

    def kernel_us4(arr, warm, reps):
        for _ in range(warm):
            al4.RunStepBatchAsync(arr, items)
        ctx.sync()
        ctx.set_profiling(True)
        for _ in range(reps):
            al4.RunStepBatchAsync(arr, items)
        nl, ms = ctx.profile_read()
        ctx.set_profiling(False)
        return ms / nl * 1e3
    one_us = kernel_us4(al4.make_pairs([p for pr in lv_pairs for p in pr]), 200, 20)
    lv_us = [kernel_us4(al4.make_pairs(pr), 200 if k == 0 else 600, 20) for k, pr in enumerate(lv_pairs)]
    px4 = sum(w * h for (w, h) in lv4) * P
    out["configs4_pyramid4"] = dict(pairs_per_launch=len(lv4) * P, factor_sets=P, levels=[list(s) for s in lv4], one_launch_kernel_us=one_us,
                                    evals_per_s=P / (one_us * 1e-6), algorithmic_bytes=(20 + 4 * CS) * px4, algorithmic_gbs=(20 + 4 * CS) * px4 / (one_us * 1e-6) / 1e9,
                                    frac=(20 + 4 * CS) * px4 / (one_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    level_by_level_kernel_us=lv_us, level_by_level_frac=(20 + 4 * CS) * px4 / (sum(lv_us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    mfma=MFMA_NAMES[ctx.last_mfma_mode()] + " -- the library default (DFX_MFMA_AUTO)",
                                    note="BASELINE configs[4] as stated: SfmAligner::RunStep over the FOUR pyramid levels 1280x960 ... 160x120 of 16 factor sets at CS = 64 "
                                         "(7.2 GB working set): one 'evaluation' = all four levels; ONE launch over the 64 pairs vs one launch per level; step kernels only")
    del lv_pairs, lv_keep, items
    torch.cuda.empty_cache()
    # ---- configs[2] as the reference's relinearisation round (PhotometricFactor::RunAlignmentStep, photometric_factor.cpp:225-293, for every
    # factor of a 16-keyframe window): UpdateDepth once per keyframe whose code moved + one batched RunStep over the 120 pairs
    from deepfactors_amd.dist import PairGraph
    W, H, CS, K = 640, 480, 32, 16
    al2 = dfx.SfmAligner(code_size=CS, ctx=ctx)
    kfs = [synth.make_pair(W, H, CS, seed=0x1600 + k, device=dev) for k in range(K)]
    # 120 pairs i < j (BASELINE's count), and the 240 DIRECTED pairs the mapper actually links: every keyframe pair gets a photometric factor in
    # both directions (core/mapping/mapper.cpp:308-311)
    for both, key in ((False, "configs2_linearize_16kf_120pairs"), (True, "configs2_linearize_16kf_240pairs_both_directions")):
        graph = PairGraph.all_pairs(K, both_directions=both)
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
        out[key] = dict(round_us=dt * 1e6, pairs=len(plist), evals_per_s=len(plist) / dt, algorithmic_gbs=byts / dt / 1e9, frac=byts / dt / 1e9 / HBM_PEAK_GBS,
                        note=f"UpdateDepth of the 16 keyframes (136 B/px each) + one batched RunStep of the {len(plist)} pairs (148 B/px each) per round, wall clock "
                             "of enqueue-to-completion; the 16 Jacobian images (630 MB) are shared by the pairs, so part of the stream is Infinity-Cache/L2 traffic"
                             + ("; both directions of every keyframe pair, as Mapper links them (mapper.cpp:308-311)" if both else ""))
        del arr, items
    return out



def _R_of(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gauss_newton_round(dfx, synth, ctx, al, kfs, graph, poses, reps=8, serial=True):
    """ONE Gauss-Newton iteration of a keyframe window, end to end, the way a mapper would wait for it (Mapper::MappingStep, core/mapping/mapper.cpp:450-552,
    minus iSAM2): UpdateDepth of every keyframe (batched) -> ONE batched RunStep + block-sparse assembly -> D2H of the system -> host Cholesky of the
    (gauge-fixed, damped) normal equations -> pose / code update.  Every stage is timed on its own (wall clock, synchronised); `serial` adds the
    reference's call pattern for the same linearisation: PhotometricFactor::linearize factor by factor -- cold (every call a blocking UpdateDepth +
    RunStep, photometric_factor.cpp:86-181) and warmed by ONE batched round that seeds every factor's cache (deepfactors_amd.factors.linearize_all)."""
    import torch
    import scipy.linalg
    from types import SimpleNamespace
    from deepfactors_amd.dist import NormalEquations
    from deepfactors_amd import factors as F
    cs, K, D = al.CS, len(kfs), 6 + al.CS
    n_pairs = graph.n_pairs
    neq = NormalEquations(graph, cs, kfs[0]["img0"].device)
    items = torch.zeros(n_pairs * dfx.item_size(12 + cs), dtype=torch.uint8, device=kfs[0]["img0"].device)
    codes = np.stack([np.asarray(k["code"], np.float32) for k in kfs])
    poses = [np.asarray(p, np.float32).copy() for p in poses]
    prx, jac, dpt = [k["prx_orig"] for k in kfs], [k["prx_jac"] for k in kfs], [k["dpt0"] for k in kfs]

    def pairs_of(ps):
        return al.make_pairs([dict(pose0=ps[int(i)], pose1=ps[int(j)], cam=kfs[int(i)]["cam"], img0=kfs[int(i)]["img0"], img1=kfs[int(j)]["img0"], dpt0=kfs[int(i)]["dpt0"],
                                   valid0=kfs[int(i)]["valid0"], prx0_jac=kfs[int(i)]["prx_jac"], grad1=kfs[int(j)]["grad1"]) for (i, j) in graph.pairs])
    keep = np.ones(K * D, bool)
    keep[0:6] = False                                    # gauge: pose of keyframe 0
    # LAPACK on 8 threads: with the pool's default (every host core: 128 on the GPU box) a 570- or 2426-unknown Cholesky is SLOWER (1.7 / 31-41 ms)
    # and the pool's idle workers keep spinning into the next iteration's enqueue (the GPU stage of the 64-keyframe round read 9 - 36 ms)
    try:
        from threadpoolctl import threadpool_limits
        blas_limit = lambda: threadpool_limits(limits=8, user_api="blas")   # noqa: E731
    except Exception:   # noqa: BLE001
        import contextlib
        blas_limit = contextlib.nullcontext
    stages = dict(marshal_ms=[], gpu_ms=[], d2h_ms=[], host_system_ms=[], host_solve_ms=[], update_ms=[], total_ms=[])
    arr = pairs_of(poses)                                # images, cameras: once per graph
    gi, gj = np.asarray(graph.pairs, np.int64)[:, 0], np.asarray(graph.pairs, np.int64)[:, 1]
    for rep in range(reps + 2):
        t = [time.perf_counter()]
        al.set_poses_all(arr, poses, gi, gj)             # poses: every round
        t.append(time.perf_counter())
        dfx.UpdateDepthBatch(codes, prx, jac, 2.0, dpt, ctx=ctx)
        al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        ctx.sync()
        t.append(time.perf_counter())
        buf = neq.buf.cpu()
        t.append(time.perf_counter())
        H, g = neq.dense_from(buf.numpy())
        if rep == 0:
            keep &= np.diag(H) > 0                       # unknowns no factor touches (the code of a keyframe that is never a pair's keyframe)
        A = H[np.ix_(keep, keep)]
        A[np.diag_indices_from(A)] *= 1.0 + 1e-4
        t.append(time.perf_counter())
        d = np.zeros(K * D)
        with blas_limit():
            d[keep] = -scipy.linalg.cho_solve(scipy.linalg.cho_factor(A, lower=True, overwrite_a=True, check_finite=False), g[keep], check_finite=False)
        t.append(time.perf_counter())
        # (the timed rounds re-evaluate the same point: the update is computed and retracted into copies)
        new_poses = [synth.pose_qt(synth.so3_exp(d[k * D + 3:k * D + 6]) @ _R_of(poses[k][:4]), poses[k][4:].astype(np.float64) + d[k * D:k * D + 3]) for k in range(K)]
        new_codes = (codes + d.reshape(K, D)[:, 6:]).astype(np.float32)
        del new_poses, new_codes
        t.append(time.perf_counter())
        if rep >= 2:
            for name, a, b in (("marshal_ms", 0, 1), ("gpu_ms", 1, 2), ("d2h_ms", 2, 3), ("host_system_ms", 3, 4), ("host_solve_ms", 4, 5), ("update_ms", 5, 6), ("total_ms", 0, 6)):
                stages[name].append((t[b] - t[a]) * 1e3)
    out = {k: float(np.median(v)) for k, v in stages.items()}
    out.update(keyframes=K, pairs=n_pairs, unknowns=int(keep.sum()), system_bytes=int(neq.buf.numel() * 4),
               note="one Gauss-Newton iteration end to end (median of %d): the round's poses into the pair descriptors (host; images marshalled once per graph) | batched UpdateDepth + batched RunStep + assembly (GPU, to sync) | D2H of the "
                    "block-sparse system | dense host system | scipy Cholesky (LAPACK, 8 threads) | retract" % reps)
    if serial:
        # the reference's per-factor pattern over the same factor set
        mk = lambda k: SimpleNamespace(pyr_img=[k["img0"]], pyr_grad=[k["grad1"]], pyr_dpt=[k["dpt0"]], pyr_vld=[k["valid0"]], pyr_stdev=[k["std0"]],   # noqa: E731
                                       pyr_prx_orig=[k["prx_orig"]], pyr_jac=[k["prx_jac"]])
        objs = [mk(k) for k in kfs]
        facs = [F.PhotometricFactor(kfs[int(i)]["cam"], objs[int(i)], objs[int(j)], ("p", int(i)), ("p", int(j)), ("c", int(i)), 0, al) for (i, j) in graph.pairs]
        vals = [(poses[int(i)], poses[int(j)], codes[int(i)]) for (i, j) in graph.pairs]
        for f, v in zip(facs, vals):
            f.linearize(*v)
        ctx.sync()
        cold, warm = [], []
        for rep in range(3):
            for f in facs:
                f.first_ = True                          # forget the cache: every linearize() is a blocking UpdateDepth + RunStep
            t0 = time.perf_counter()
            for f, v in zip(facs, vals):
                f.linearize(*v)
            cold.append((time.perf_counter() - t0) * 1e3)
            for f in facs:
                f.first_ = True
            t0 = time.perf_counter()
            F.linearize_all(facs, vals)                  # ONE batched round seeds every cache ...
            for f, v in zip(facs, vals):
                f.linearize(*v)                          # ... and the serial calls iSAM2 makes launch nothing
            warm.append((time.perf_counter() - t0) * 1e3)
        out.update(serial_linearize_cold_ms=float(np.median(cold)), serial_linearize_warmed_ms=float(np.median(warm)),
                   serial_note="PhotometricFactor::linearize called factor by factor over the same factors (Python mirror deepfactors_amd.factors, incl. the "
                               "G11..G33 slicing per factor): cold = every call a blocking UpdateDepth + RunStep (the reference's pattern); warmed = one "
                               "linearize_all round first, then the same serial calls hit their caches")
    return out


def tracker_and_geometric_configs(dfx, synth, ctx, dev):
    """BASELINE configs[0] (3-level SE3 tracker; the synthetic 640x480 pair of SURVEY 8d cfg 1 AND the reference's 320x240 fixture 1047 -> 1052 of
    tests/ut_se3aligner.cpp:173-211) and the geometric half of configs[2] (SparseGeometricFactor::linearize with geo_npoints = 500,
    data/flags/common.flags:29-34, for the 120 pairs of the 16-keyframe window)."""
    import torch
    out = {}
    # ---- configs[0], synthetic 640x480, levels 0..2, iterations 10,5,5 (flags tracking_iters = 5,5,10 are coarse-to-fine)
    p = synth.make_pair(640, 480, 16, seed=0xDF01, device=dev, with_decoder=False)
    cams = synth.camera_pyramid(p["cam"], 3)
    lv = [dict(img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"])]
    for _ in range(2):
        q = lv[-1]
        h, w = q["img0"].shape
        n = {}
        for k in ("img0", "img1"):
            n[k] = torch.empty((h // 2, w // 2), dtype=torch.float32, device=dev)
            dfx.GaussianBlurDown(q[k], n[k], ctx)
        n["dpt0"] = q["dpt0"][::2, ::2].contiguous()
        n["grad1"] = torch.empty((h // 2, w // 2, 2), dtype=torch.float32, device=dev)
        dfx.SobelGradients(n["img1"], n["grad1"], ctx)
        lv.append(n)
    iters = (10, 5, 5)
    trk = dfx.CameraTracker(cams, dfx.TrackerConfig(3, iters, 0.1), ctx)
    trk.SetKeyframe([l["img0"] for l in lv], [l["dpt0"] for l in lv])

    def frame():
        trk.Reset()
        return trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
    for _ in range(20):
        pose = frame()
    t0 = time.perf_counter()
    reps = 100
    for _ in range(reps):
        pose = frame()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gt = np.asarray(p["pose10_true"], np.float64)
    out["configs0_se3_tracker_3level"] = dict(ms_per_frame=ms, frames_per_s=1e3 / ms, iterations=list(iters), levels=3, width=640, height=480,
                                              pose_error_t=float(np.linalg.norm(pose[4:] - gt[4:])), pose_error_q=float(np.linalg.norm(pose[:4] - gt[:4])),
                                              residual_per_inlier=float(trk.GetError()), inliers_frac=float(trk.GetInliers()),
                                              note="CameraTracker::TrackFrame (core/system/camera_tracker.cpp:42-71) as ONE enqueue (dfx_track_frame: SE3 step + 6x6 LDL^T + retract "
                                                   "on the device, 20 iterations over 3 levels), from identity, incl. the blocking read-back of the pose")
    # ---- configs[0], the reference's fixture (ut_se3aligner.cpp:45-97,173-211): 320x240, 25x25 box blur, depth mm -> m, 40 iterations, criterion residual / inliers <= 1e-3
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "se3_fixture_1047_1052.npz")
    from scipy import ndimage   # (already imported by main(): see there)
    if os.path.exists(fx):
        d = np.load(fx)
        img0 = ndimage.uniform_filter(d["img0"].astype(np.float32) / np.float32(255), 25, mode="mirror")
        img1 = ndimage.uniform_filter(d["img1"].astype(np.float32) / np.float32(255), 25, mode="mirror")
        dpt0 = d["dpt0_mm"].astype(np.float32) / np.float32(1000)
        cam = np.array([np.float32(160 / 0.5773502691896257), np.float32(120 / 0.41421356237309503), 160, 120, 320, 240], np.float32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
        g1 = torch.empty((240, 320, 2), dtype=torch.float32, device=dev)
        i0, i1, d0 = t(img0), t(img1), t(dpt0)
        dfx.SobelGradients(i1, g1, ctx)
        tr = dfx.CameraTracker([cam], dfx.TrackerConfig(1, (40,), 0.1), ctx)
        tr.SetKeyframe([i0], [d0])
        for _ in range(10):
            tr.Reset(); tr.TrackFrame([i1], [g1])
        t0 = time.perf_counter()
        for _ in range(50):
            tr.Reset(); tr.TrackFrame([i1], [g1])
        ms = (time.perf_counter() - t0) / 50 * 1e3
        out["configs0_se3_fixture_1047_1052"] = dict(ms_per_alignment=ms, iterations=40, width=320, height=240, residual_per_inlier=float(tr.GetError()),
                                                     inliers_frac=float(tr.GetInliers()), passes_reference_criterion=bool(tr.GetError() <= 1e-3),
                                                     note="ut_se3aligner.cpp ImageAlignmentTest on the reference's own images (data/testimg/1047.jpg, 1052.jpg, 1047.png): 40 "
                                                          "Gauss-Newton iterations from identity in one enqueue; criterion residual / inliers <= 1e-3")
    # ---- configs[2], geometric half: 120 pairs x 500 points
    W, H, CS, K, NPTS = 640, 480, 32, 16, 500
    from deepfactors_amd.dist import PairGraph
    graph = PairGraph.all_pairs(K, both_directions=False)
    kfs = [synth.make_pair(W, H, CS, seed=0x1600 + k, device=dev) for k in range(K)]
    dgrad = []
    for k in kfs:
        g = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
        dfx.SobelGradients(k["dpt0"], g, ctx)            # mapper.cpp:998-1000: the keyframe's depth gradient
        dgrad.append(g)
    rng = np.random.default_rng(0xDF02)
    facs = []
    for (i, j) in graph.pairs:
        pts = np.stack([rng.integers(0, W, NPTS), rng.integers(0, H, NPTS)], 1).astype(np.int32)
        a, b = kfs[int(i)], kfs[int(j)]
        facs.append((dfx.SparseGeometricFactor(a["cam"], pts, dict(prx_orig=a["prx_orig"], prx_jac=a["prx_jac"]),
                                               dict(prx_orig=b["prx_orig"], prx_jac=b["prx_jac"], dpt_grad=dgrad[int(j)]), 0.1, code_size=CS, ctx=ctx), a, b))
    def geo_round():
        rows = None
        for f, a, b in facs:
            rows = f.linearize(a["pose0"], b["pose1"], a["code"], b["code"])
        return rows
    for _ in range(3):
        rows = geo_round()
    t0 = time.perf_counter()
    for _ in range(5):
        rows = geo_round()
    dt = (time.perf_counter() - t0) / 5
    out["configs2_sparse_geometric_500pts"] = dict(round_ms=dt * 1e3, factors=len(facs), points_per_factor=NPTS, us_per_factor=dt / len(facs) * 1e6,
                                                   nonzero_rows_last=int((np.abs(rows).sum(1) > 0).sum()),
                                                   note="SparseGeometricFactor::linearize (core/gtsam/sparse_geometric_factor.cpp:147-275) for the 120 pairs of the 16-keyframe window, "
                                                        "500 points each: one blocking dfx_sparse_geometric_linearize per factor (the reference's per-factor pattern; it is a CPU loop "
                                                        "there that first syncs the 39 MB code Jacobian to the host), rows read back to the host")
    # the same round as ONE launch (dfx_sparse_geometric_linearize_batch: CS / 4 lanes per point, grid.y = factor): rows left on the device (what a device-side
    # consumer -- a normal-equation assembly -- would read), and fetched to the host with one copy (what gtsam::JacobianFactor needs: 18.5 MB per round)
    gfs = [f for f, _, _ in facs]
    gvals = [(a["pose0"], b["pose1"], a["code"], b["code"]) for _, a, b in facs]
    ncol = 12 + 2 * CS + 1
    rows_dev = torch.empty((len(gfs) * NPTS, ncol), dtype=torch.float32, device=dev)
    batch_rows = dfx.SparseGeometricFactor.linearize_all(gfs, gvals)
    same = all(np.array_equal(batch_rows[k], gfs[k].linearize(*gvals[k])) for k in (0, 57, 119))
    for f in gfs:
        f.upload_points()                                    # the reference samples a factor's points once, in its constructor
    gbatch = dfx.SparseGeometricFactor.prepare(gfs)          # cameras, decoder images and points marshalled once; poses and codes per round
    enq = lambda: dfx.SparseGeometricFactor.linearize_all(gbatch, gvals, rows_dev=rows_dev)   # noqa: E731
    for _ in range(5):
        enq()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        enq()
    ctx.sync()
    dev_round = (time.perf_counter() - t0) / 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    enq(); e0.record(); enq(); e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dfx.SparseGeometricFactor.linearize_all(gbatch, gvals)
    host_round = (time.perf_counter() - t0) / 5
    gram_fn = getattr(dfx.SparseGeometricFactor, "gram_all", None)    # the round's normal equations formed on the device (dfx_sparse_geometric_gram_batch)
    if gram_fn is not None:
        gram_fn(gbatch, gvals)
        t0 = time.perf_counter()
        for _ in range(5):
            gram_fn(gbatch, gvals)
        out["configs2_sparse_geometric_500pts"]["batched_round_ms_normal_equations_to_host"] = (time.perf_counter() - t0) / 5 * 1e3
    out["configs2_sparse_geometric_500pts"].update(batched_round_ms_rows_on_device=dev_round * 1e3, batched_gpu_ms=e0.elapsed_time(e1), batched_round_ms_rows_to_host=host_round * 1e3,
                                                   batched_equals_per_factor_bits=bool(same), rows_bytes=int(rows_dev.numel() * 4),
                                                   batched_note="the 120 factors in ONE launch: rows_on_device = wall clock per round of back-to-back enqueues incl. the Python marshalling of "
                                                                "120 descriptors (points resident on the device); batched_gpu_ms = one round on the stream (H2D of the descriptors + kernel, HIP "
                                                                "events); rows_to_host = the blocking form with ONE device-to-host copy of all rows")
    # ---- configs[2]: one Gauss-Newton iteration of the 16-keyframe / 120-pair window end to end
    al = dfx.SfmAligner(code_size=CS, ctx=ctx)
    poses = []
    for k in range(K):
        R = synth.so3_exp(rng.normal(0, 0.004, 3))
        poses.append(synth.pose_qt(R, rng.normal(0, 0.008, 3)))
    out["configs2_gauss_newton_round_16kf_120pairs"] = gauss_newton_round(dfx, synth, ctx, al, kfs, graph, poses)
    del kfs, facs, dgrad
    torch.cuda.empty_cache()
    # the same relinearisation round from C++ (include/dfx_host.hpp; the Python mirror above pays ~0.1 ms of interpreter time per factor): the reference's serial
    # linearize() pattern, the batched seam, and the serial pattern warmed by the batched seam -- tools/cpp/gn_round_bench.cpp, built by tests/cpp/Makefile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "cpp", "gn_round_bench")
    for key, argv, what in (("configs2_relinearisation_round_cpp", ["16", "7"], "16 keyframes / 120 factors (all pairs i < j)"),
                            ("configs3_relinearisation_round_cpp", ["64", "3", "16"], "64 keyframes / 1024 factors (every keyframe linked to its 16 nearest: BASELINE configs[3])")):
        if not os.path.exists(exe):
            break
        try:
            r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=240, stdin=subprocess.DEVNULL)
            vals = {}
            for line in r.stdout.splitlines():
                tok = line.split()
                if len(tok) == 4 and tok[0].endswith("_ms"):
                    vals[tok[0]] = float(tok[1]); vals[tok[0][:-3] + "_per_factor_us"] = float(tok[3])
            if r.returncode == 0 and vals:
                vals["note"] = (f"C++ host layer (include/dfx_host.hpp, tools/cpp/gn_round_bench.cpp), {what} of 640x480x32 per round, median of {argv[1]}: serial_cold = "
                                "PhotometricFactor::linearize factor by factor (blocking UpdateDepth + RunStep each: what a header-swap build under iSAM2 delivers); batched = "
                                "dfx::LinearizeAll (one decode + one step launch); serial_warmed = LinearizeAll, then the serial calls hit their caches; every variant incl. the "
                                "G11..G33 slicing.  geometric_* = the sparse geometric factors of the same links (500 points each): all factors in ONE launch with the rows left on the "
                                "device / fetched with one copy (dfx::SparseGeometricLinearizeAll), and one blocking call per factor")
                out[key] = vals
            else:
                out[key] = {"error": (r.stdout + r.stderr)[-300:]}
        except Exception as e:   # noqa: BLE001 -- an extra figure, never the line
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    # blocking-call latencies as a C++ call site sees them (tools/cpp/latency_bench.cpp through include/dfx_shim.hpp), incl. dfx_track_frame without the Python wrapper
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "cpp", "latency_bench")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe], capture_output=True, text=True, timeout=120, stdin=subprocess.DEVNULL)
            lat = {}
            for line in r.stdout.splitlines():
                if " mean " in line and " us " in line:
                    name, rest = line.split(" mean ", 1)
                    lat[name.strip()] = float(rest.split()[0])
            if r.returncode == 0 and lat:
                out["blocking_call_latency_cpp_us"] = lat
                trk = [v for k, v in lat.items() if k.startswith("dfx_track_frame")]
                if trk and "configs0_se3_tracker_3level" in out:
                    out["configs0_se3_tracker_3level"]["cpp_ms_per_frame"] = trk[0] * 1e-3
        except Exception as e:   # noqa: BLE001
            out["blocking_call_latency_cpp_us"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_baseline_se3(levels_iters=(10, 5, 5)):
    """cpu_baseline leg of configs[0]: the oracle's SE3 step + the reference's solve-and-update (lucas_kanade_se3.h:85-95) in the coarse-to-fine
    schedule of CameraTracker::TrackFrame on the synthetic 640x480 pair, fp32 accumulate, OpenMP over rows on all host cores and on one thread."""
    from deepfactors_amd import synth
    from oracle import dfx_oracle as orc   # cpu_baseline leg only
    orc.build_native()
    n = synth.to_numpy(synth.make_pair(640, 480, 16, seed=0xDF01, device="cpu", with_decoder=False))
    cams = synth.camera_pyramid(n["cam"], 3)
    lv = [dict(img0=n["img0"], img1=n["img1"], dpt0=n["dpt0"], grad1=orc.sobel(n["img1"]))]
    for _ in range(2):
        q = lv[-1]
        i0, i1 = orc.blur_down(q["img0"]), orc.blur_down(q["img1"])
        lv.append(dict(img0=i0, img1=i1, dpt0=np.ascontiguousarray(q["dpt0"][::2, ::2][: i0.shape[0], : i0.shape[1]]), grad1=orc.sobel(i1)))

    def track(threads):
        qt = synth.IDENTITY.copy()
        for level in (2, 1, 0):
            for _ in range(levels_iters[level]):
                r = orc.se3_step(qt, cams[level], lv[level]["img0"], lv[level]["img1"], lv[level]["dpt0"], lv[level]["grad1"], 0.1, accum_f64=False, threads=threads)
                qt = orc.se3_solve_update(r.JtJ, r.Jtr, qt)
        return qt
    cores = orc.max_threads()
    res = {}
    for name, th, reps in (("all_cores", cores, 5), ("single_thread", 1, 2)):
        track(th)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            track(th)
            ts.append(time.perf_counter() - t0)
        res[name] = float(np.median(ts)) * 1e3
    return dict(ms_per_frame=res["all_cores"], cores=cores, single_thread_ms_per_frame=res["single_thread"], kind="port",
                sample="the oracle's SE3 step (a port of LucasKanadeSE3, lucas_kanade_se3.h:41-77) + SE3SolveAndUpdate in the 10/5/5 schedule over 3 levels of the synthetic "
                       "640x480 pair, g++ -O3 -march=native, OpenMP over rows")

def window_config(dfx, synth, ctx, dev, dist, rank, world, comm=None):
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
        if comm is not None:
            comm.reduce(ctx, neq.buf, root=0)   # the C-ABI exchange (dfx_comm_reduce_f32_async)
        elif dist is not None:
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
    res = dict(keyframes=K, pairs=graph.n_pairs, pairs_per_rank=hi - lo, ms_per_step=el / steps * 1e3, evals_per_s=graph.n_pairs * steps / el,
               system_bytes=int(neq.buf.numel() * 4),
               note="keyframe pyramids replicated; 16 pairs share each keyframe's 39 MB Jacobian, so this configuration re-reads from L2 / Infinity Cache "
                    "(not an HBM-roofline figure)")
    if dist is None or world == 1:
        # what a mapper waits for per Gauss-Newton iteration of this window: decode + step + assembly + D2H + host solve + update, and the
        # reference's serial linearize() pattern over the same 1024 factors (cold / warmed by one batched round)
        del items, neq, arr
        res["gauss_newton_round"] = gauss_newton_round(dfx, synth, ctx, al, kfs, graph, poses, reps=4)
    return res


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
    # Imported HERE, far from any timed loop: the first import of scipy.ndimage starts a thread pool whose start-up keeps the host cores busy for
    # a moment -- imported right in front of the tracker loop of tracker_and_geometric_configs() it made that host-driven loop read 0.97 ms
    # per frame instead of 0.26 (profiles/r04_bench_scipy_import.txt).
    import scipy.ndimage  # noqa: F401

    W, H, CS, P = a.width, a.height, a.cs, a.pairs
    ctx = dfx.Context(local)
    # library defaults unless forced: explicit calls, so that stray DFX_MFMA / DFX_SCHEDULE environment variables cannot steer the line
    ctx.set_mfma_mode({"auto": _dl.DFX_MFMA_AUTO, "f32": _dl.DFX_MFMA_F32_CHAIN, "bf16x3": _dl.DFX_MFMA_BF16X3}[a.mfma])
    ctx.set_schedule({"auto": _dl.DFX_SCHEDULE_AUTO, "static": _dl.DFX_SCHEDULE_STATIC, "dynamic": _dl.DFX_SCHEDULE_DYNAMIC}[a.schedule])
    al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=a.step_blocks), code_size=CS, ctx=ctx)

    # ---- synthetic, device-resident input: P distinct keyframe->frame pairs per rank
    pairs, keep = build_pairs(dfx, synth, dev, rank, P, W, H, CS, same=bool(os.environ.get("DFX_BENCH_SAME")), ctx=None if a.foreign_valid0 else ctx)
    arr = apply_workload(al, synth, pairs, a.workload)
    isz = dfx.item_size(12 + CS)
    items = torch.zeros(P * isz, dtype=torch.uint8, device=dev)

    if a.pmc_worker:   # counters-only child of pmc_traffic(): a few launches of the same workload, nothing else
        for _ in range(4):
            al.RunStepBatchAsync(arr, items)
        ctx.sync()
        return

    # The exchange.  cabi (default): the library's own collectives -- a dfx_comm created over RCCL from a unique id that rank 0 hands to the others over the
    # process group, then per step dfx_comm_reduce_f32_async on the context's exchange stream (what dfx_graph_reduce_async issues for the graph's system).
    comm, exchange_note = None, None
    if dist is not None and a.exchange == "cabi":
        from deepfactors_amd.dist import Comm
        try:
            comm = Comm.create(ctx, dist, rank, world, dev)
            err = ""
        except Exception as e:   # noqa: BLE001 -- e.g. an RCCL without an entry point dfx_comm.cpp resolves
            err = f"{type(e).__name__}: {e}"
        # the ranks agree: one rank without a communicator puts every rank on the process group's exchange (said in config.exchange), instead of N - 1 ranks
        # waiting in a collective the last one never joins
        okf = torch.tensor([0.0 if err else 1.0], device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if float(okf.item()) < 1.0:
            if comm is not None:
                comm.close()
            comm, exchange_note = None, "torch (dfx_comm_create failed on a rank" + (": " + err[:200] if err else "") + ")"
            print(f"[rank {rank}] C-ABI communicator unavailable, exchanging through torch.distributed {err}", file=sys.stderr)

    # Deferred tail: consecutive steps are independent batches, so the reduction tail of step k (the tail kernel with the graph assembly, and the RCCL reduce of
    # its system) can run on a second stream beside the 1 ms step kernel of step k + 1 (dfx_set_tail_stream).  Opt-in for one rank (--deferred-tail: -0.5 % step time
    # at a 1.9 % lower roofline fraction, profiles/r05_step_gap.txt); with the C-ABI exchange and N > 1 it is how the collective leaves the launch stream: the library
    # enqueues ncclReduce on the tail stream, behind the assembly.  Every tail and every reduce has completed when the timed region ends.
    tail = torch.cuda.Stream(device=dev) if (a.deferred_tail or (comm is not None and world > 1)) else None
    if tail is not None:
        ctx.set_tail_stream(tail)

    # the pairs of all ranks form one trajectory: pair p links keyframe node p -> frame node p + 1
    graph = PairGraph.chain(world * P)
    pipe = PipelinedReduce(dist, [NormalEquations(graph, CS, dev) for _ in range(2)], root=0, stream=tail) if (world > 1 and comm is None) else None
    # C-ABI exchange: two system buffers used round robin; the reduce of step k and the assembly of step k + 2 into the same buffer are ordered by the
    # stream they are both enqueued on
    systems = [NormalEquations(graph, CS, dev) for _ in range(2 if (comm is not None and world > 1) else 1)] if pipe is None else None
    neq = systems[0] if systems is not None else None
    step_no = [0]

    fused = not a.two_call_tail

    def step():
        # hot path: one launch over P pairs (+ its tail kernel), then this rank's items are summed into the block-sparse
        # normal equations of the graph; for N > 1 the ranks' buffers are reduced onto the rank that solves
        if pipe is not None:
            al.RunStepBatchAssembleAsync(arr, items, pipe.next(), rank * P, fused=fused)
            pipe.submit()                                      # RCCL reduce over xGMI, overlapped with the next step's kernels
            return
        sysb = systems[step_no[0] % len(systems)]
        step_no[0] += 1
        al.RunStepBatchAssembleAsync(arr, items, sysb, rank * P, fused=fused)
        if comm is not None:
            comm.reduce(ctx, sysb.buf, root=0)                 # ncclReduce on the context's exchange stream, behind the assembly
        elif dist is not None:
            if tail is not None:
                with torch.cuda.stream(tail):
                    sysb.reduce(dist, root=0)
            else:
                sysb.reduce(dist, root=0)

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
        got = float((systems[(step_no[0] - 1) % len(systems)] if pipe is None else pipe.last()).g.double().sum())
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
            "config": {"workload": {"truth": "every pair evaluated at its generating pose (residual ~ 0); ", "perturbed": "pose1 of every pair perturbed by N(0, 5 mm) / N(0, 0.3 deg) per axis (SURVEY 8d cfg 3); ",
                                    "unrelated": "perturbed poses and img1 / grad1 of another scene (Huber active); "}[a.workload]
                                   + f"BASELINE configs[1] geometry in the batch size of configs[3] (1k pairs / 8 GPUs): {P} independent {W}x{H} pairs per GPU per step, "
                                   f"CS={CS}, SfmAligner::RunStep (SE3+code Jacobians, JtJ/Jtr) in one launch, level 0; "
                                   "+ block-sparse normal-equation assembly" + ((" + RCCL reduce to rank 0 through " + ("the C ABI (dfx_comm_reduce_f32_async)" if comm is not None else "torch.distributed")) if dist is not None else "")
                                   + ("; the reduction tail of step k (finalize, assembly" + (", reduce" if world > 1 else "") + ") runs on a second stream beside the kernel of step k + 1"
                                      if tail is not None else ""),
                       "pairs_per_gpu": P, "width": W, "height": H, "code_size": CS, "poses": a.workload,
                       "parallelism": f"pairs sharded over {world} GPU(s)", "exchange": ("cabi" if comm is not None else ((exchange_note or "torch") if dist is not None else "none"))},
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
        # the same 128 pairs as the OTHER workloads (--workload): perturbed poses (what a relinearisation sees) and unrelated textures (Huber active)
        for kind in ("truth", "perturbed", "unrelated"):
            if kind == a.workload:
                continue
            arr_k = apply_workload(al, synth, pairs, kind)
            r = mode_kernel_us(ctx, lambda: al.RunStepBatchAsync(arr_k, items), (20 + 4 * CS) * W * H * P, warm=150, steps=30, mode=_dl.DFX_MFMA_AUTO)
            ctx.sync()
            its_k = al.items_from_bytes(items.cpu().numpy(), CS)
            r.update(mean_inliers_frac=float(np.mean([it.inliers for it in its_k]) / (W * H)),
                     mean_residual_per_inlier=float(np.mean([it.residual / max(it.inliers, 1) for it in its_k])))
            configs["headline_" + kind] = r
            del arr_k
    probe = parity_probe(dfx, synth, ctx, dev, W, H, CS) if (world == 1 and rank == 0 and not a.no_cpu_baseline) else None
    del keep, pairs, arr
    torch.cuda.empty_cache()
    if world == 1 and not a.no_configs:
        configs.update(secondary_configs(dfx, synth, ctx, dev))
        torch.cuda.empty_cache()
        configs.update(small_operator_rooflines(dfx, synth, ctx, dev))
        torch.cuda.empty_cache()
        configs.update(tracker_and_geometric_configs(dfx, synth, ctx, dev))
        torch.cuda.empty_cache()
    if a.window or (world == 1 and not a.no_configs):
        configs["configs3_window64"] = window_config(dfx, synth, ctx, dev, dist, rank, world, comm)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
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
            if not a.no_traffic and not a.no_configs:
                for kind in ("truth", "perturbed", "unrelated"):
                    if "headline_" + kind in configs and "kernel_us" in configs["headline_" + kind]:
                        tr_k, det_k = pmc_traffic(a, workload=kind)
                        configs["headline_" + kind].update(traffic=tr_k, traffic_ratio=(tr_k / ((20 + 4 * CS) * W * H * P) if tr_k else None), traffic_source=det_k.get("source"))
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(W, H, CS)
                try:
                    out["parity_blocks"] = parity_blocks(probe)
                except Exception as e:   # noqa: BLE001 -- a reported check, never the line
                    out["parity_blocks"] = {"error": f"{type(e).__name__}: {e}"}
                if "configs0_se3_tracker_3level" in configs:
                    configs["configs0_se3_tracker_3level"]["cpu_baseline"] = cpu_baseline_se3()
        else:
            out["roofline"]["traffic_source"] = "not collected for N > 1 (per-rank kernels are identical to the N = 1 run)"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
