#!/usr/bin/env python3
"""Timeline of the last N dispatches of a rocprofv3 --kernel-trace CSV: kernel, duration, idle gap since the previous kernel's end [us].
usage: kt_gaps.py <kernel_trace.csv> [--last 12] [--like SUBSTR] [--around SUBSTR]
(--like keeps only the kernels whose name contains SUBSTR; --around shows the window of `last` dispatches that ENDS with the last dispatch whose name contains SUBSTR)"""
import csv
import sys


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 12
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else ""
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if like and like not in (r.get("Kernel_Name") or ""):
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or "")[:60]))
    rows.sort()
    if "--around" in sys.argv:
        key = sys.argv[sys.argv.index("--around") + 1]
        idx = max(i for i, r in enumerate(rows) if key in r[2])
        rows = rows[: idx + 1]
    prev_end = None
    for s, e, n in rows[-last:]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else float("nan")
        print(f"{n:60s} dur {(e - s) / 1e3:8.2f} us   gap before {gap:7.2f} us")
        prev_end = e


if __name__ == "__main__":
    main()
