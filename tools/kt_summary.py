#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): per kernel count / avg / min / max / total in microseconds, plus the
average of the LAST n dispatches of each kernel (= the timed steps of bench.py; earlier ones are clock ramp and warm-up).
usage: kt_summary.py <kernel_trace.csv> [--like SUBSTR] [--last N]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else ""
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    rows = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if like and like not in name:
                continue
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            rows[name].append((s, (e - s) / 1e3))
    tot = sum(sum(d for _, d in v) for v in rows.values()) or 1.0
    print("kernel,calls,avg_us,min_us,max_us,total_us,share_pct" + (f",last{last}_avg_us,last{last}_min_us,last{last}_max_us" if last else ""))
    for name, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        v.sort()
        d = [x for _, x in v]
        line = f"\"{name.replace(',', ';')[:160]}\",{len(d)},{sum(d) / len(d):.3f},{min(d):.3f},{max(d):.3f},{sum(d):.3f},{100 * sum(d) / tot:.2f}"
        if last:
            t = d[-last:]
            line += f",{sum(t) / len(t):.3f},{min(t):.3f},{max(t):.3f}"
        print(line)


if __name__ == "__main__":
    main()
