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
  *) echo "unknown section $what"; exit 2 ;;
esac
