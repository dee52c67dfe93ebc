#!/bin/bash
# Round-5 GPU sessions (from the repo root on the GPU box): bash tools/gpu_r5.sh <what>
set -u
what=${1:-base}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
short="--no-cpu-baseline --pmc 0 --gpu-reference 0"
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("ms_per_step", "value", "host_enqueue_ms_per_step", "ms_per_step_library_default", "ms_per_step_async_wgrad_off", "ms_per_step_sustained", "collective")})
print({k: (round(v["ms"], 4), round(v.get("frac", 0), 3)) for k, v in d.get("kernels", {}).items()})
for k, v in d.get("ab", {}).items():
    print("A/B", k, v["median_default_ms"], v["median_toggled_ms"])
PY
}
case "$what" in
  base)   # tests at HEAD + the modes line + the uncontended per-call table
    timeout 2400 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --durations=10 > gpurun_out/pytest_gpu.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|pytest exit" gpurun_out/pytest_gpu.log | tail -8
    timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
    timeout 900 python bench.py $short --force-collective --sustained 200 > gpurun_out/bench_modes.json 2> gpurun_out/bench_modes.err; echo "bench exit $?"; summ gpurun_out/bench_modes.json
    MVS_ASYNC_WGRAD=0 timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/bench_k_sync.json 2> gpurun_out/bench_k_sync.err
    grep "ms/step" gpurun_out/bench_k_sync.err > gpurun_out/kernel_table_uncontended.txt; head -70 gpurun_out/kernel_table_uncontended.txt ;;
  modes)  # only the new mode tests
    timeout 1200 python -m pytest tests/test_gpu_modes.py -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_modes.log 2>&1
    echo "pytest exit $?"; tail -15 gpurun_out/pytest_modes.log ;;
  ab)     # bash tools/gpu_r5.sh ab "<spec>;<spec>" [reps]: interleaved A/B pairs of the config-2 step + the uncontended kernel table
    timeout 900 python bench.py --steps 20 --warmup 5 $short --ab "$2" --ab-reps ${3:-6} > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err; echo "bench exit $?"; summ gpurun_out/bench_ab.json
    MVS_ASYNC_WGRAD=0 timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/bench_k_sync.json 2> gpurun_out/bench_k_sync.err
    grep "ms/step" gpurun_out/bench_k_sync.err > gpurun_out/kernel_table_uncontended.txt; head -${4:-60} gpurun_out/kernel_table_uncontended.txt ;;
  pmcn)   # bash tools/gpu_r5.sh pmcn "<knob settings...>": SQ / TCC counters of the narrow-layer microbenchmark (tools/bench_narrow.py), one pass per counter group
    i=0
    if [ "${4:-sq}" = "mem" ]; then
      set -- "$1" "$2" "${3:-conv1 fwd}" mem "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
    else
      set -- "$1" "$2" "${3:-conv1 fwd}" sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
    fi
    for grp in "${@:5}"; do
      i=$((i+1)); rm -rf gpurun_out/pmcn_$i
      (cd /tmp && MVS_NARROW_ONLY="${3:-conv1 fwd}" timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmcn_$i" -o pmc -- \
          python "$OLDPWD/tools/bench_narrow.py" $2 > "$OLDPWD/gpurun_out/pmcn_$i.log" 2>&1); echo "pmc pass $i exit $?"
    done
    python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcn_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(k in name for k in ("conv_", "bn_")):
            continue
        acc[name][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        per = collections.defaultdict(float)
        for d, x in v:
            per[d] += x
        vals = sorted(per.values())
        print("   %-28s median %.4g  (n=%d)" % (c, vals[len(vals) // 2], len(vals)))
PY
    rm -rf gpurun_out/pmcn_[0-9] ;;
  configs)  # configs 3 / 4 / 5 short lines
    for cfg in 3 4 5; do
      timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --gpu-reference 0 --pmc 0 > gpurun_out/bench_c$cfg.json 2> gpurun_out/bench_c$cfg.err
      echo "config $cfg exit $?"; summ gpurun_out/bench_c$cfg.json
    done ;;
  final|final2)  # the closing sequence of the round: tools/gpu_round.sh final (final2: without configs 4 / 5) + what round 5 added
    bash tools/gpu_round.sh $what
    timeout 900 python bench.py $short --force-collective --sustained 200 > gpurun_out/final_bench_collective_sustained.json 2> gpurun_out/final_bench_cs.err; echo "collective+sustained exit $?"; summ gpurun_out/final_bench_collective_sustained.json
    MVS_ASYNC_WGRAD=0 timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/final_bench_k_sync.json 2> gpurun_out/final_bench_k_sync.err
    grep "ms/step" gpurun_out/final_bench_k_sync.err > gpurun_out/final_kernel_table_sync_mode.txt; head -8 gpurun_out/final_kernel_table_sync_mode.txt
    timeout 600 python tools/bench_narrow.py "" > gpurun_out/final_narrow_layers.log 2>&1; grep -v amdgpu gpurun_out/final_narrow_layers.log | tail -15
    timeout 900 python bench.py --steps 20 --warmup 5 $short --ab "wgrad8_gs=0;wgrad8_gs=1;wgrad8_groups=160;wgrad8_groups=256;c_entry" --ab-reps 6 > gpurun_out/final_bench_ab.json 2> gpurun_out/final_bench_ab.err; echo "ab exit $?"; summ gpurun_out/final_bench_ab.json ;;
  *) echo "unknown section $what"; exit 2 ;;
esac
