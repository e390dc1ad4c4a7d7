#!/usr/bin/env python3
"""Per-level durations of the pyramid-build launches from a rocprofv3 kernel trace of tools/pyramid_bench.py.
usage (on the GPU box):  rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o pt -- python tools/pyramid_bench.py 64 ; python tools/pyramid_trace.py /tmp/pt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel trace under " + root
    by = defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        name = row["Kernel_Name"]
        if "k_pyr" not in name:
            continue
        grid = "x".join(str(row[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in row) or row.get("Grid_Size", "?")
        key = (name.split("(")[0].split("<")[0][-16:] + ("<" + name.split("<")[1].split(">")[0] + ">" if "<" in name else ""), grid, row.get("Workgroup_Size_X", row.get("Workgroup_Size", "?")))
        by[key].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    tot = 0.0
    for key, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        v.sort()
        last = v[-50:]
        d = sorted((e - s) / 1e3 for s, e in last)
        med = d[len(d) // 2]
        tot += med
        print(f"{key[0]:>24s} grid {key[1]:>9s} wg {key[2]:>4s}: {len(v):5d} dispatches, last {len(last)}: median {med:7.2f} us  min {d[0]:7.2f}  max {d[-1]:7.2f}")
    print(f"sum of medians {tot:.1f} us")


if __name__ == "__main__":
    main()
