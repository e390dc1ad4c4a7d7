#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum"; do
  rm -rf /tmp/pm
  timeout 120 rocprofv3 --pmc $ctrs --kernel-include-regex "k_pyr_rows" --output-format csv -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/tools/pyramid_bench.py 64 > /dev/null 2>/tmp/pm.err < /dev/null
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no output for: $ctrs"; tail -3 /tmp/pm.err; continue; }
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, d in sorted(acc.items(), key=lambda kv: -int(kv[0]) if kv[0].isdigit() else 0):
    print("grid", g, {k: round(sum(v[-20:]) / len(v[-20:])) for k, v in d.items()})
PY
done
