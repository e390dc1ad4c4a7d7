#!/bin/bash
# round 4, call 4: what bounds the row-walk kernels -- SQ / TA / TCP / TCC counters of the two batched kernels (DT = 1 build)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC|SQ|GRBM)_[A-Z0-9_]+\b" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
export DFX_LIB=$PWD/gpurun_build/libdfx_s1e1.so
run() {
  local name=$1; shift
  REPS=12 timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" --output-format csv -d "$OUT/$name" -o pmc -- python tools/small_ops_driver.py > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum
run tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RD_UNCACHED_32B_sum
python - <<'P'
import csv,glob,collections
for name in ('sq1','sq2','tcp1','tcp2','ta','tcc'):
    fs=glob.glob(f'gpurun_out/r04q/{name}/**/*counter_collection.csv', recursive=True)
    if not fs: print(name,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k='se3' if 'se3_step_batch' in r['Kernel_Name'] else 'err'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(name,k,{c:round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
P
tail -3 $OUT/*.err | head -40
rm -rf $OUT/sq1 $OUT/sq2 $OUT/tcp1 $OUT/tcp2 $OUT/ta $OUT/tcc
