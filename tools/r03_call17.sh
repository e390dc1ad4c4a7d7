#!/bin/bash
# Round 3, GPU call 17: what bounds the batched SE3 step / EvaluateError: SQ counters of the two kernels.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03q; mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" --output-format csv -d "$OUT/$name" -o pmc -- python tools/small_ops_driver.py > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES_LT_64 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
python - <<'P'
import csv,glob,collections
for name in ('sq1','sq2','sq3'):
    fs=glob.glob(f'gpurun_out/r03q/{name}/**/*counter_collection.csv', recursive=True)
    if not fs: print(name,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k='se3' if 'se3_step_batch' in r['Kernel_Name'] else 'err'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(name,k,{c:round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
P
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
