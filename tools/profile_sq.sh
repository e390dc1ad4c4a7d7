#!/bin/bash
# SQ / TCC latency counters of the SfM step kernel (rocprofv3 --pmc, counters only).  Usage: tools/profile_sq.sh <outdir> <lib.so> [ab_bench worker args]
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
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE
run lat TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
timeout 60 python tools/pmc_summary.py "$OUT" < /dev/null
