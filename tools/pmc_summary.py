#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection CSVs per kernel and counter.  Usage: pmc_summary.py <dir> [regex]   (kernels whose name matches; default k_sfm_step)"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
rx = re.compile(sys.argv[2] if len(sys.argv) > 2 else "k_sfm_step")
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    acc, cnt = defaultdict(float), defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            m = rx.search(name)
            if not m:
                continue
            short = re.sub(r"^.*?(k_\w+).*$", r"\1", name)
            acc[(short, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[(short, row["Counter_Name"])] += 1
    print(f"# {os.path.relpath(f, d)}")
    for k in sorted(acc):
        print(f"{k[0]},{k[1]},{acc[k] / cnt[k]:.1f},dispatches={cnt[k]}")
