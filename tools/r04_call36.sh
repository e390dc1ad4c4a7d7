#!/bin/bash
# round 4, call 36: counters of the FINAL row-walk kernels (dword taps, snap, scalar ray load, zero-copy descriptors): two passes, <= 4 counters of a block each, 60 s cap
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04q36; mkdir -p $OUT; export TMPDIR=/tmp
run() {
  local name=$1; shift
  POSE=true REPS=8 timeout 60 rocprofv3 --pmc "$@" --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" --output-format csv -d "$OUT/$name" -o pmc -- python tools/small_ops_driver.py > "$OUT/$name.log" 2> "$OUT/$name.err" < /dev/null
  echo "$name rc=$?"
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sqw SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python - <<'P'
import csv,glob,collections
for name in ('tcc','sq','sqw'):
    fs=glob.glob(f'gpurun_out/r04q36/{name}/**/*counter_collection.csv', recursive=True)
    if not fs: print(name,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k='se3' if 'se3_step_batch' in r['Kernel_Name'] else 'err'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(name,k,{c:round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
P
rm -rf $OUT/tcc $OUT/sq $OUT/sqw
