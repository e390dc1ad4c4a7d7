#!/usr/bin/env python3
"""Does the step kernel run faster when the GPU idles between launches?  (Round 3, call 10: with a 120 us reduction tail behind every
launch the step kernel of bench.py took 930 us instead of 974 us.)  The bench workload (128 distinct 640x480 pairs, CS = 32, library
defaults) launch after launch on one stream, with `torch.cuda._sleep` spins of a given length between the launches; per phase the
step kernel's time from the library's events, the wall time per step, and rocm-smi's clocks / power sampled beside it.
usage: python tools/idle_gap_probe.py [--idle-us 0,60,120,240] [--seconds 3]"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def smi_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            c = d.get("card0", {})
            keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor memory)"))}
            out.append(keep)
        except Exception as e:   # noqa: BLE001
            out.append({"error": str(e)[:80]})
        stop.wait(0.4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--idle-us", default="0,60,120,240,0")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--pairs", type=int, default=128)
    a = ap.parse_args()
    import torch
    import bench
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(code_size=32, ctx=ctx)
    pairs, keep = bench.build_pairs(dfx, synth, dev, 0, a.pairs, 640, 480, 32, ctx=ctx)
    arr = al.make_pairs(pairs)
    items = torch.zeros(a.pairs * dfx.item_size(44), dtype=torch.uint8, device=dev)
    # calibrate torch.cuda._sleep: cycles per microsecond
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        torch.cuda._sleep(1_000_000)
    e1.record(); torch.cuda.synchronize()
    cyc_per_us = 20 * 1_000_000 / (e0.elapsed_time(e1) * 1e3)
    print(f"_sleep calibration: {cyc_per_us:.1f} cycles per us", flush=True)
    for _ in range(1500):   # clock ramp
        al.RunStepBatchAsync(arr, items)
    ctx.sync()
    for idle in [float(x) for x in a.idle_us.split(",")]:
        cyc = int(idle * cyc_per_us)
        n = max(200, int(a.seconds * 1e6 / (1000.0 + idle)))
        stop, samples = threading.Event(), []
        th = threading.Thread(target=smi_sampler, args=(stop, samples), daemon=True)
        for _ in range(n // 3):   # settle in the new duty cycle
            al.RunStepBatchAsync(arr, items)
            if cyc:
                torch.cuda._sleep(cyc)
        ctx.sync()
        ctx.set_profiling(True)
        th.start()
        t0 = time.perf_counter()
        for _ in range(n):
            al.RunStepBatchAsync(arr, items)
            if cyc:
                torch.cuda._sleep(cyc)
        ctx.sync()
        t1 = time.perf_counter()
        stop.set(); th.join()
        nl, ms, mn, mx = ctx.profile_read_ex()
        ctx.set_profiling(False)
        print(f"idle {idle:6.0f} us: kernel {1e3 * ms / nl:7.1f} us (min {1e3 * mn:.1f}, max {1e3 * mx:.1f}) over {nl} launches, wall {1e6 * (t1 - t0) / n:7.1f} us per step", flush=True)
        for s in samples[:6]:
            print("   ", json.dumps(s), flush=True)


if __name__ == "__main__":
    main()
