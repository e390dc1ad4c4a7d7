#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run: per-kernel count / avg / min / max / total, as CSV.
Usage: python tools/rocpd_summary.py <results.db> [--like PATTERN] > profiles/<name>.csv"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else "%"
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0   # also: average of the last N dispatches per kernel
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) "
        "from kernels where name like ? group by name order by sum(end-start) desc", (like,)).fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("kernel,calls,avg_us,min_us,max_us,total_us,share_pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,grid_x,grid_y,wg_x")
    for r in rows:
        name = r[0].replace(",", ";")
        print(f"\"{name}\",{r[1]},{r[2]/1e3:.3f},{r[3]/1e3:.3f},{r[4]/1e3:.3f},{r[5]/1e3:.3f},{100*r[5]/tot:.2f},"
              f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]},{r[12]},{r[13]}")


    if last:
        print(f"# average over the last {last} dispatches of each kernel (= the timed steps of bench.py; earlier ones are clock ramp / warm-up)")
        for r in rows:
            d = [x[0] for x in cur.execute("select end-start from kernels where name = ? order by start desc limit ?", (r[0], last)).fetchall()]
            if d:
                print(f"\"{r[0].replace(',', ';')}\",last{len(d)},{sum(d) / len(d) / 1e3:.3f},{min(d) / 1e3:.3f},{max(d) / 1e3:.3f}")


if __name__ == "__main__":
    main()
