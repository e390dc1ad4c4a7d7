#!/bin/bash
# Memory-side request counters of the SfM step kernel by request size (rocprofv3 --pmc, counters only):
#   bytes read = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B  (TCC -> fabric requests; Infinity-Cache hits included)
# Usage: tools/profile_traffic.sh <outdir> <lib.so> [ab_bench worker args]
set -u
OUT=$1; LIB=$2; shift 2
ARGS=${@:-"--pairs 128 --distinct --steps 3 --preroll 5"}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local name=$1; shift
  DFX_LIB=$LIB timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "k_sfm_step" --output-format csv -d "$OUT/$name" -o pmc -- \
    python tools/ab_bench.py --worker $ARGS > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
run dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_REQ_sum TCC_READ_sum
timeout 60 python tools/pmc_summary.py "$OUT" < /dev/null
