#!/bin/bash
# round 4, call 7: memory-pipeline counters of the row-walk kernels (<= 4 counters of a block per pass, 60 s cap per pass)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04q7; mkdir -p $OUT; export TMPDIR=/tmp
run() {
  local name=$1; shift
  REPS=8 timeout 60 rocprofv3 --pmc "$@" --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" --output-format csv -d "$OUT/$name" -o pmc -- python tools/small_ops_driver.py > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
run sq TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU
python - <<'P'
import csv,glob,collections
for name in ('tcp1','ta1','tcc1','tcp2','sq'):
    fs=glob.glob(f'gpurun_out/r04q7/{name}/**/*counter_collection.csv', recursive=True)
    if not fs: print(name,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k='se3' if 'se3_step_batch' in r['Kernel_Name'] else 'err'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(name,k,{c:round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
P
rm -rf $OUT/tcp1 $OUT/ta1 $OUT/tcc1 $OUT/tcp2 $OUT/sq
