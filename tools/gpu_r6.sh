#!/bin/bash
# Round-6 GPU sessions (from the repo root on the GPU box): bash tools/gpu_r6.sh <what> [args]
set -u
what=${1:-green}
out=gpurun_out/${2:-r06}
mkdir -p $out
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
short="--no-cpu-baseline --pmc 0 --gpu-reference 0"
box() { echo "box: $(hostname) $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1) $(date -u +%FT%TZ)"; }
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("ms_per_step", "value", "host_enqueue_ms_per_step", "ms_per_step_library_default", "ms_per_step_async_wgrad_off")})
print({k: (round(v["ms"], 4), round(v.get("frac", 0), 3)) for k, v in d.get("kernels", {}).items()})
for k, v in d.get("ab", {}).items():
    print("A/B", k, v["median_default_ms"], v["median_toggled_ms"])
PY
}
case "$what" in
  green)  # N consecutive full GPU suites, exactly the driver's command line
    box > $out/green.log
    for i in $(seq 1 ${3:-5}); do
      python -m pytest tests -x -q -m gpu > $out/pytest_$i.log 2>&1; rc=$?
      echo "run $i: exit $rc: $(tail -1 $out/pytest_$i.log)" | tee -a $out/green.log
    done ;;
  ab)     # bash tools/gpu_r6.sh ab <dir> "<spec>;<spec>" [reps] [config]
    timeout 1200 python bench.py --config ${5:-2} --steps 20 --warmup 5 $short --ab "$3" --ab-reps ${4:-6} > $out/bench_ab.json 2> $out/bench_ab.err; echo "bench exit $?"; summ $out/bench_ab.json ;;
  ktable) # uncontended per-call kernel table (synchronous weight gradients)
    MVS_ASYNC_WGRAD=0 timeout 600 python bench.py --config ${3:-2} --steps 10 --warmup 3 --time-all-kernels $short > $out/bench_k_sync.json 2> $out/bench_k_sync.err
    grep "ms/step" $out/bench_k_sync.err > $out/kernel_table_sync_mode.txt; head -${4:-70} $out/kernel_table_sync_mode.txt ;;
  line)   # short bench line of a config
    timeout 900 python bench.py --config ${3:-2} --steps 20 --warmup 5 $short > $out/bench_c${3:-2}.json 2> $out/bench_c${3:-2}.err; echo "exit $?"; summ $out/bench_c${3:-2}.json ;;
  libab)  # bash tools/gpu_r6.sh libab <dir> [config] [reps]: two BUILDS of the library (tools/ab/lib_old.so, lib_new.so), alternating short bench lines
    lib=self-supervised-mvs_amd/libmvs_hip.so; cp $lib /tmp/lib_keep.so
    for i in $(seq 1 ${4:-3}); do
      for v in old new; do
        cp tools/ab/lib_$v.so $lib
        timeout 600 python bench.py --config ${3:-2} --steps 20 --warmup 5 $short > $out/bench_${v}_$i.json 2> $out/bench_${v}_$i.err
        python - $out/bench_${v}_$i.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 4), {k: round(v["ms"], 4) for k, v in d.get("kernels", {}).items()})
PY
      done
    done
    cp /tmp/lib_keep.so $lib ;;
  *) echo "unknown section $what"; exit 2 ;;
esac
