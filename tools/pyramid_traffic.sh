#!/bin/bash
# Memory-side request counters of the pyramid build's launches by request size (rocprofv3 --pmc, counters only, two passes):
#   bytes read = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B ; bytes written = 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B)   (as tools/profile_traffic.sh)
# usage (GPU box): tools/pyramid_traffic.sh OUT
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/pyr_traffic}; mkdir -p $O
for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wr:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pm_$name
  timeout 200 rocprofv3 --pmc $ctrs --kernel-include-regex "k_pyr" --output-format csv -d /tmp/pm_$name -o pm -- python tools/pyramid_bench.py 64 --build-only > /dev/null 2> $O/$name.err < /dev/null
done
python - $O <<'PY' > $O/pyramid_traffic.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("rd", "wr"):
    for f in glob.glob(f"/tmp/pm_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"].split("(")[0][-24:], r.get("Grid_Size", "?"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
W, H, F = 640, 480, 64
alg = {0: (4 * W * H * F, (8 + 1) * W * H * F), 1: (4 * W * H * F // 4, 9 * W * H * F // 4)}
tot_r = tot_w = 0.0
for key, d in sorted(acc.items(), key=lambda kv: -int(kv[0][1]) if kv[0][1].isdigit() else 0):
    m = {k: sum(v[-20:]) / len(v[-20:]) for k, v in d.items()}
    rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0))
    tot_r += rd; tot_w += wr
    print(f"{key[0]:>24s} grid {key[1]:>8s}: read {rd / 1e6:8.1f} MB  written {wr / 1e6:8.1f} MB   (L2 hits {m.get('TCC_HIT_sum', 0) / 1e6:.2f} M, misses {m.get('TCC_MISS_sum', 0) / 1e6:.2f} M; average of the last 20 dispatches)")
algr = sum(4 * (W >> i) * (H >> i) for i in range(4)) * F
algw = sum((8 + (1 if i < 3 else 0)) * (W >> i) * (H >> i) for i in range(4)) * F
print(f"build: read {tot_r / 1e6:.1f} MB (algorithmic {algr / 1e6:.1f}), written {tot_w / 1e6:.1f} MB (algorithmic {algw / 1e6:.1f}); total {(tot_r + tot_w) / (algr + algw):.3f} x the algorithmic {(algr + algw) / 1e6:.1f} MB")
PY
cat $O/pyramid_traffic.txt
