#!/bin/bash
# One GPU-box session (from the repo root on the GPU box): bash tools/gpu_round.sh <what>
#   tests    full -m gpu suite + smoke                       kernels  per-kernel microbenchmarks (tools/bench_kernels.py, bench_conv2d.py)
#   bench    default bench line + per-kernel HIP-event table  prof N   rocprofv3 --kernel-trace --stats of bench.py --config N
#   pmc      HBM traffic of the roofline kernels (tools/pmc_driver.py under rocprofv3 --pmc, separate passes)
#   final    the closing sequence of a round: tests, smoke, default line, kernel table, configs 3-5, rocprofv3 stats of configs 2-5
#   final2   the same without configs 4 / 5 (after a change to the training path only)
#   next     first session of the next round: the opt-in consumer-side BatchNorm of the 2-D extractor and two balance knobs, 8 A/B pairs each
# Outputs go to gpurun_out/ (merged back by gpurun); the ones worth keeping are copied to profiles/ by hand.
# (The per-experiment sections of rounds 1-4 -- A/B pairs of individual knobs -- are in the git history of this file; an A/B is now
#  `python bench.py --ab "<knob>=<value>;..."`: interleaved default / toggled runs inside one process.)
set -u
what=${1:-tests}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
short="--no-cpu-baseline --pmc 0 --gpu-reference 0"

run_tests() {
  timeout 2400 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --durations=10 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|pytest exit" gpurun_out/pytest_gpu.log | tail -6
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
}
run_prof() {   # $1 = config
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --config $1 --steps 40 --warmup 5 $short > "$OLDPWD/gpurun_out/prof_bench_c$1.json" 2> "$OLDPWD/gpurun_out/prof_c$1.err")
  echo "prof config $1 exit $?"; cut -c1-160 gpurun_out/prof_bench_c$1.json
  find gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rocprofv3_kernel_stats_c$1.csv \;
  rm -rf gpurun_out/prof
}
summarise() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("ms_per_step", "value", "host_enqueue_ms_per_step", "ms_per_step_async_wgrad_off", "wgrad_join")})
print({k: (round(v["ms"], 4), round(v.get("frac", 0), 3), v.get("traffic")) for k, v in d.get("kernels", {}).items()})
for k, v in d.get("ab", {}).items():
    print("A/B", k, v["median_default_ms"], v["median_toggled_ms"])
PY
}

case "$what" in
  tests) run_tests ;;
  bench)
    timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; summarise gpurun_out/bench.json
    timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err
    grep "ms/step" gpurun_out/bench_k.err > gpurun_out/bench_kernel_table.txt; head -30 gpurun_out/bench_kernel_table.txt ;;
  kernels)
    MVS_BENCH_SKIP_SWEEP=${MVS_BENCH_SKIP_SWEEP:-} timeout 900 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels.log
    timeout 600 python tools/bench_conv2d.py > gpurun_out/conv2d_layers.log 2>&1; echo "conv2d exit $?"; grep -v Warn gpurun_out/conv2d_layers.log | tail -28 ;;
  prof) run_prof "${2:-2}" ;;
  pmc)
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/pmc_$c
      (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$c" -o pmc -- \
          python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1); echo "pmc $c exit $?"
    done
    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json; cat gpurun_out/pmc_summary.json
    rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE ;;
  final)
    run_tests
    timeout 1800 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $?"; summarise gpurun_out/final_bench.json
    timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/final_bench_k.json 2> gpurun_out/final_bench_k.err
    grep "ms/step" gpurun_out/final_bench_k.err > gpurun_out/final_bench_kernel_table.txt; head -12 gpurun_out/final_bench_kernel_table.txt
    timeout 600 python bench.py --steps 20 --warmup 5 --step-events 1 $short > gpurun_out/final_bench_step_events.json 2> gpurun_out/final_bench_se.err
    python -c "import json; d = json.load(open('gpurun_out/final_bench_step_events.json')); print('per-step GPU ms', d['step_gpu_ms'], 'side-stream lag at the join', d.get('side_stream_lag_at_join_ms_median_max'))"
    for cfg in 3 4 5; do
      timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --gpu-reference 0 > gpurun_out/final_bench_c$cfg.json 2> gpurun_out/final_bench_c$cfg.err
      echo "config $cfg exit $?"; summarise gpurun_out/final_bench_c$cfg.json
    done
    for cfg in 2 3 4 5; do run_prof $cfg; done ;;
  final2)   # after a change that only touches the config-2 / config-3 training path: the closing sequence without configs 4 / 5
    run_tests
    timeout 1800 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $?"; summarise gpurun_out/final_bench.json
    timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels $short > gpurun_out/final_bench_k.json 2> gpurun_out/final_bench_k.err
    grep "ms/step" gpurun_out/final_bench_k.err > gpurun_out/final_bench_kernel_table.txt; head -12 gpurun_out/final_bench_kernel_table.txt
    timeout 600 python bench.py --steps 20 --warmup 5 --step-events 1 $short > gpurun_out/final_bench_step_events.json 2> gpurun_out/final_bench_se.err
    python -c "import json; d = json.load(open('gpurun_out/final_bench_step_events.json')); print('per-step GPU ms', d['step_gpu_ms'], 'side-stream lag at the join', d.get('side_stream_lag_at_join_ms_median_max'))"
    timeout 900 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --gpu-reference 0 > gpurun_out/final_bench_c3.json 2> gpurun_out/final_bench_c3.err
    echo "config 3 exit $?"; summarise gpurun_out/final_bench_c3.json
    for cfg in 2 3; do run_prof $cfg; done ;;
  next)     # first session of round 5: what round 4 built last and measured once
    run_tests
    timeout 900 python bench.py --steps 20 --warmup 5 --step-events 1 $short --ab "feature_fused_apply;wgrad2d_batch=3072;wgrad8_groups=224" --ab-reps 8 \
        > gpurun_out/next_bench.json 2> gpurun_out/next_bench.err; echo "bench exit $?"; summarise gpurun_out/next_bench.json
    timeout 600 python tools/bench_conv2d.py > gpurun_out/conv2d_layers.log 2>&1; grep -v Warn gpurun_out/conv2d_layers.log | tail -16 ;;
  *) echo "unknown section $what"; exit 2 ;;
esac
