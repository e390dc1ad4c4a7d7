#!/bin/bash
# PMC passes over the SfM step kernel (rocprofv3, counters only -- no trace domains combined with --pmc).
# Every pass is wrapped in `timeout`; nothing here reads stdin.  Usage: tools/profile_pmc.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:-"--steps 4 --warmup 1 --no-cpu-baseline"}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-include-regex "k_sfm_step" --output-format csv -d "$OUT/$name" -o pmc -- \
    python bench.py $ARGS > "$OUT/$name.bench.json" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*counter_collection.csv" < /dev/null | head -20
