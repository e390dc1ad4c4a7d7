#!/usr/bin/env python3
"""Warmed timing of the row-walk reductions (k_se3_step_batch at the identity and at the pairs' true poses, k_sfm_error_batch) over 128 distinct
640x480 pairs per launch -- bench.py's small-operator workload -- in a form a rocprofv3 kernel trace of the SAME process can be checked against.

Per phase: untimed windows of 50 launches until the reduction kernel's HIP-event time has settled (three consecutive windows within 1 %, at
least six: the clock ramp of bench.py's run_protocol), then 100 launches, then 30 launches whose event average is the figure.  The number of
launches per phase goes to --phases FILE, so that

    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python tools/small_ops_trace.py --phases ph.json > events.json
    python tools/small_ops_trace.py --summarise DIR/.../kt_kernel_trace.csv --phases ph.json --events events.json

slices the trace into the same phases and prints, per phase, the trace's average over the LAST 30 dispatches beside the event figure
("events and trace must agree within 2 % as they do for k_sfm_step")."""
import argparse
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HBM_PEAK_GBS = 8000.0
W, H, CS, P = 640, 480, 32, 128
PHASES = (("se3_step_batch_identity", "k_se3_step_batch", 20), ("se3_step_batch_true_pose", "k_se3_step_batch", 20), ("sfm_error_batch", "k_sfm_error_batch", 12))


def run(a):
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    prs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev) for k in range(P)]
    al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    mk = lambda key: se3.make_pairs([dict(se3=(p["pose10_true"] if key == "true" else synth.IDENTITY), cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"],  # noqa: E731
                                          grad1=p["grad1"]) for p in prs])
    s_id, s_tp = mk("ident"), mk("true")
    sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
    earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                               grad1=p["grad1"]) for p in prs])
    eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
    launch = {"se3_step_batch_identity": lambda: se3.RunStepBatch(s_id, sitems), "se3_step_batch_true_pose": lambda: se3.RunStepBatch(s_tp, sitems),
              "sfm_error_batch": lambda: al.EvaluateErrorBatch(earr, eitems)}
    out, counts = {}, {}
    for name, kernel, bpp in PHASES:
        fn = launch[name]
        n_launch, hist = 0, []
        ctx.set_profiling(True)
        ctx.profile_read()
        for w_i in range(30):
            for _ in range(50):
                fn()
            n, ms = ctx.profile_read()
            n_launch += n
            hist.append(ms / max(n, 1) * 1e3)
            if w_i >= 5 and max(hist[-3:]) <= 1.01 * min(hist[-3:]):
                break
        for _ in range(100):
            fn()
        n, _ = ctx.profile_read()
        n_launch += n
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        n, ms, lo, hi = ctx.profile_read_ex()
        n_launch += n
        torch.cuda.synchronize()
        call_us = e0.elapsed_time(e1) * 1e3 / 30
        ctx.set_profiling(False)
        byts = bpp * W * H * P
        kus = ms / n * 1e3
        out[name] = dict(kernel=kernel, events_kernel_us_last30=kus, events_kernel_us_min=lo * 1e3, events_kernel_us_max=hi * 1e3, events_kernel_frac=byts / kus / 1e3 / HBM_PEAK_GBS,
                         call_us_last30=call_us, call_frac=byts / call_us / 1e3 / HBM_PEAK_GBS, algorithmic_bytes=byts, ramp_kernel_us=[round(h, 1) for h in hist], launches=n_launch)
        counts[name] = n_launch
    ctx.sync()
    if a.phases:
        with open(a.phases, "w") as fh:
            json.dump(dict(order=[p[0] for p in PHASES], kernel={p[0]: p[1] for p in PHASES}, launches=counts), fh)
    out["_env"] = {k: os.environ.get(k) for k in ("DFX_RW_NX", "DFX_LIB", "DFX_BATCH_WGS_PER_CU") if os.environ.get(k) is not None}
    print(json.dumps(out))


def summarise(a):
    ph = json.load(open(a.phases))
    ev = json.loads(open(a.events).read().strip().splitlines()[-1]) if a.events else {}
    rows = {}
    with open(a.summarise) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            for k in set(ph["kernel"].values()):
                if k in name:
                    rows.setdefault(k, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for v in rows.values():
        v.sort()
    pos = {k: 0 for k in rows}
    print("phase,kernel,dispatches,trace_last30_avg_us,trace_last30_min_us,trace_last30_max_us,trace_frac_of_8TBs,events_last30_avg_us,events_frac,trace_over_events")
    for name in ph["order"]:
        k, n = ph["kernel"][name], ph["launches"][name]
        d = [x for _, x in rows.get(k, [])[pos[k]:pos[k] + n]]
        pos[k] += n
        if len(d) < 30:
            print(f"{name},{k},{len(d)},(fewer than 30 dispatches in the trace)")
            continue
        t = d[-30:]
        avg = sum(t) / len(t)
        byts = next(p[2] for p in PHASES if p[0] == name) * W * H * P
        e = ev.get(name, {})
        eus = e.get("events_kernel_us_last30")
        print(f"{name},{k},{len(d)},{avg:.2f},{min(t):.2f},{max(t):.2f},{byts / avg / 1e3 / HBM_PEAK_GBS:.4f}," + (f"{eus:.2f},{e['events_kernel_frac']:.4f},{avg / eus:.4f}" if eus else ",,"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--phases", default="")
    ap.add_argument("--summarise", default="")
    ap.add_argument("--events", default="")
    args = ap.parse_args()
    summarise(args) if args.summarise else run(args)
