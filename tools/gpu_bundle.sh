#!/bin/bash
# One parameterised runner for everything a `gpurun` call measures (replaces the per-call scripts tools/r0N_callNN.sh of rounds 2-4).
# Every step is timeout-guarded, reads nothing from stdin and writes under OUT (default gpurun_out/bundle); steps run in the order given.
#
#   tools/gpu_bundle.sh [OUT=dir] step [step ...]
#
# steps (ENV=.. pairs in front of a step's arguments go to its environment, e.g. "smallops:DFX_RW_NX=0:tag=direct"):
#   suite                    python -m pytest tests -m gpu -q                                   -> pytest_gpu.log
#   tests:<pytest args>      python -m pytest -m gpu -q <args>  (e.g. tests:tests/test_gpu_rowwalk_nx.py)   -> pytest_<n>.log
#   bench[:tag=T][:args]     python bench.py <args>                                             -> bench[_T].json
#   ktrace[:args]            bench.py (short form) under rocprofv3 --kernel-trace --stats        -> bench_under_rocprof.json, kernel_trace_dfx.csv, kernel_stats.csv, launch_gaps.txt
#   smallops[:tag=T]         tools/small_ops_trace.py (warmed HIP-event timing of the row-walk reductions)   -> small_ops_events[_T].json
#   smallops_trace[:tag=T]   the same under rocprofv3 --kernel-trace, sliced into its phases     -> small_ops_trace[_T].csv (+ events json)
#   smallops_pmc             SQ / TCP / TCC counters of the row-walk reductions (tools/small_ops_driver.py) -> small_ops_pmc.txt
#   pmc_sq / pmc_traffic     counters of the headline step kernel (tools/profile_sq.sh, profile_traffic.sh)
#   rccl1[:args]             bench.py under torch.distributed.run with one rank (RCCL initialised, the exchange on real streams)
#   gnround                  tests/cpp/gn_round_bench 16 7 ; 64 3 16                             -> gn_round_cpp.txt
#   latency                  tests/cpp/latency_bench                                             -> latency_cpp.txt
#   cpp                      tests/cpp/{shim,host,comm,ref_callers}_test                          -> cpp_tests.txt
#   tracker                  tools/profile_tracker.py                                            -> tracker.json
#   pyramid                  tools/pyr_call.sh (pyramid tests, events for 64 frames / 1 frame, kernel-trace timeline) -> pyr/*
#   pyramid_traffic          tools/pyramid_traffic.sh (TCC_EA0 request counters of the build's launches)    -> pyr_traffic/pyramid_traffic.txt
#   clocks                   tools/idle_gap_probe.py --idle-us 0 --seconds 3                     -> clock_power_steady.txt
#   run:<command>            any command (timeout 600)                                           -> run_<n>.log
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/bundle
case "${1:-}" in OUT=*) OUT=${1#OUT=}; shift;; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
LIB=$PWD/deepfactors_amd/libdfx.so
t0=$(date +%s); lap() { echo "== $1 rc=$2 @ $(( $(date +%s) - t0 )) s"; }
n=0
for spec in "$@"; do
  n=$((n + 1))
  step=${spec%%:*}; rest=""; [ "$spec" != "$step" ] && rest=${spec#*:}
  envs=(); tag=""; args=""
  IFS=':' read -ra parts <<< "$rest"
  for p in "${parts[@]:-}"; do
    case "$p" in
      tag=*) tag=_${p#tag=};;
      [A-Z_]*=*) envs+=("$p");;
      *) args="$args${args:+:}$p";;   # (arguments of one step are separated by spaces inside the quoted spec, not by colons)
    esac
  done
  case "$step" in
    suite) env "${envs[@]}" timeout 1500 python -m pytest tests -m gpu -q $args > $OUT/pytest_gpu$tag.log 2>&1 < /dev/null; rc=$?; tail -4 $OUT/pytest_gpu$tag.log;;
    tests) env "${envs[@]}" timeout 900 python -m pytest -m gpu -q -x $args > $OUT/pytest_$n$tag.log 2>&1 < /dev/null; rc=$?; tail -6 $OUT/pytest_$n$tag.log;;
    bench) env "${envs[@]}" timeout 1200 python bench.py $args > $OUT/bench$tag.json 2> $OUT/bench$tag.err < /dev/null; rc=$?
           python - "$OUT/bench$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "kernel_us", round(r["kernel_us"], 1), round(r["kernel_us_min"], 1), round(r["kernel_us_max"], 1), "frac", round(r["frac"], 4),
          "traffic x", round(r["traffic"] / r["algorithmic_bytes_per_launch"], 4) if r.get("traffic") else None)
    for k, v in d.get("configs", {}).items():
        print(" ", k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("note", "serial_note", "batched_note", "mfma", "levels")})[:420])
    if "cpu_baseline" in d:
        print("  cpu", json.dumps({a: b for a, b in d["cpu_baseline"].items() if a != "sample"})[:300])
except Exception as e:
    print("bench line unreadable:", e)
PY
           ;;
    ktrace) rm -rf $OUT/kt; env "${envs[@]}" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-traffic --no-configs $args > $OUT/bench_under_rocprof$tag.json 2> $OUT/kt.err < /dev/null; rc=$?
            KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); ST=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
            [ -n "$KT" ] && { python tools/kt_summary.py $KT --like dfx --last 30 > $OUT/kernel_trace_dfx$tag.csv; python tools/kt_gaps.py $KT --last 8 > $OUT/launch_gaps$tag.txt 2>&1; cut -c1-230 $OUT/kernel_trace_dfx$tag.csv; tail -12 $OUT/launch_gaps$tag.txt; }
            [ -n "$ST" ] && grep -E "^\"Name\"|dfx::" $ST > $OUT/kernel_stats$tag.csv
            rm -rf $OUT/kt;;
    smallops) env "${envs[@]}" timeout 300 python tools/small_ops_trace.py > $OUT/small_ops_events$tag.json 2> $OUT/small_ops$tag.err < /dev/null; rc=$?
              python - "$OUT/small_ops_events$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k, v in d.items():
        if k != "_env":
            print(f"  {k:28s} kernel {v['events_kernel_us_last30']:7.2f} us = {v['events_kernel_frac']:.4f}   call {v['call_us_last30']:7.2f} us = {v['call_frac']:.4f}   ramp {v['ramp_kernel_us'][:3]}..{v['ramp_kernel_us'][-1]}")
    print("  env", d.get("_env"))
except Exception as e:
    print("small_ops line unreadable:", e)
PY
              ;;
    smallops_trace) rm -rf /tmp/kts; env "${envs[@]}" timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o kt -- python tools/small_ops_trace.py --phases $OUT/small_ops_phases$tag.json > $OUT/small_ops_events_under_trace$tag.json 2> $OUT/small_ops_trace$tag.err < /dev/null; rc=$?
              f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
              [ -n "$f" ] && { python tools/small_ops_trace.py --summarise $f --phases $OUT/small_ops_phases$tag.json --events $OUT/small_ops_events_under_trace$tag.json > $OUT/small_ops_trace$tag.csv; cat $OUT/small_ops_trace$tag.csv; }
              rm -rf /tmp/kts;;
    smallops_pmc) : > $OUT/small_ops_pmc$tag.txt
              for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum GRBM_GUI_ACTIVE"; do
                rm -rf /tmp/pmc; POSE=true REPS=8 env "${envs[@]}" timeout 90 rocprofv3 --pmc $ctrs --kernel-include-regex "k_se3_step_batch|k_sfm_error_batch" --output-format csv -d /tmp/pmc -o pmc -- python tools/small_ops_driver.py > /dev/null 2>&1 < /dev/null
                python tools/pmc_summary.py /tmp/pmc "k_se3_step_batch|k_sfm_error_batch" >> $OUT/small_ops_pmc$tag.txt 2>&1
              done; rc=0; cat $OUT/small_ops_pmc$tag.txt | head -40;;
    pmc_sq) timeout 400 tools/profile_sq.sh $OUT/sq $LIB --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq_cs32_128pairs.txt 2>&1 < /dev/null; rc=$?;;
    pmc_traffic) timeout 400 tools/profile_traffic.sh $OUT/tr $LIB --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_traffic_cs32_128pairs.txt 2>&1 < /dev/null; rc=$?;;
    rccl1) HSA_ENABLE_IPC_MODE_LEGACY=0 env "${envs[@]}" timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --no-cpu-baseline --no-traffic --no-configs $args > $OUT/bench_rccl_1rank$tag.json 2> $OUT/bench_rccl_1rank$tag.err < /dev/null; rc=$?
           tail -c 600 $OUT/bench_rccl_1rank$tag.json | cut -c1-600;;
    gnround) { timeout 200 tests/cpp/gn_round_bench 16 7; timeout 300 tests/cpp/gn_round_bench 64 3 16; } > $OUT/gn_round_cpp$tag.txt 2>&1 < /dev/null; rc=$?; cat $OUT/gn_round_cpp$tag.txt;;
    latency) timeout 120 tests/cpp/latency_bench > $OUT/latency_cpp$tag.txt 2>&1 < /dev/null; rc=$?; tail -14 $OUT/latency_cpp$tag.txt;;
    cpp) rc=0; for t in shim_test host_test host_logic_test comm_test ref_callers_test; do [ -x tests/cpp/$t ] && { e=(); [ $t = comm_test ] && e=(DFX_RCCL_LIB=$PWD/tests/cpp/librccl_stub.so); env "${e[@]}" timeout 300 tests/cpp/$t 2>&1 | tail -2; r=${PIPESTATUS[0]}; [ $r -ne 0 ] && rc=$r; }; done > $OUT/cpp_tests$tag.txt 2>&1 < /dev/null; cat $OUT/cpp_tests$tag.txt;;
    pyramid) timeout 600 tools/pyr_call.sh $OUT/pyr > $OUT/pyramid$tag.log 2>&1 < /dev/null; rc=$?; grep -E "build_pyramid|median|passed|failed" $OUT/pyramid$tag.log | cut -c1-150;;
    pyramid_traffic) timeout 400 tools/pyramid_traffic.sh $OUT/pyr_traffic > $OUT/pyramid_traffic$tag.log 2>&1 < /dev/null; rc=$?; tail -4 $OUT/pyramid_traffic$tag.log | cut -c1-200;;
    tracker) timeout 120 python tools/profile_tracker.py > $OUT/tracker$tag.json 2>/dev/null < /dev/null; rc=$?; tail -2 $OUT/tracker$tag.json;;
    clocks) timeout 120 python tools/idle_gap_probe.py --idle-us 0 --seconds 3 > $OUT/clock_power_steady$tag.txt 2>&1 < /dev/null; rc=$?; grep "^idle" $OUT/clock_power_steady$tag.txt;;
    run) env "${envs[@]}" timeout 600 bash -c "$args" > $OUT/run_$n$tag.log 2>&1 < /dev/null; rc=$?; tail -20 $OUT/run_$n$tag.log;;
    *) echo "unknown step $step"; rc=99;;
  esac
  lap "$spec" $rc
done
find $OUT -name "*.csv" -size +400k -delete; find $OUT -name "*.db" -delete
exit 0
