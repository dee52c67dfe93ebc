#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+ per-kernel HIP-event table), rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
if [ "$what" = "tbk" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -5
  timeout 900 python bench.py --steps 10 --warmup 3 --time-all-kernels --torch-profile gpurun_out/torch_profile.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json | cut -c1-300; grep "ms/step" gpurun_out/bench.err | head -12
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels.log
  [ -x tools/valu_rate.bin ] && timeout 60 tools/valu_rate.bin | tee gpurun_out/valu_rate.log
fi
if [ "$what" = "kb" ]; then
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels.log
  timeout 900 python bench.py --steps 10 --warmup 3 --time-all-kernels --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json | cut -c1-300; grep "ms/step" gpurun_out/bench.err | head -14
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_clean.json 2> gpurun_out/bench_clean.err
  echo "clean bench exit $?"; cut -c1-220 gpurun_out/bench_clean.json
fi
if [ "$what" = "pmc" ]; then
  # HBM traffic of the roofline kernels: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$c" -o pmc -- \
        python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1); echo "pmc $c exit $?"
  done
  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json; cat gpurun_out/pmc_summary.json
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
fi
if [ "$what" = "x" ]; then
  # A/B of the Cout==8 forward forms and tile orders: parity, kernel timings, HBM traffic (+ FETCH_SIZE calibration), bench
  timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider -k "cout8 or conv3d_family or costregnet" > gpurun_out/pytest_x.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_x.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_x.log | tail -8
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -E "conv0|calibration|sweep_fwd" gpurun_out/kernels.log
  rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/pmc_C
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_F" -o pmc -- \
      python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_F.log" 2>&1); echo "pmc F exit $?"
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_W" -o pmc -- \
      python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_W.log" 2>&1); echo "pmc W exit $?"
  (cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_C" -o pmc -- \
      "$OLDPWD/tools/fetch_calib.bin" > "$OLDPWD/gpurun_out/fetch_calib.log" 2>&1); echo "calib exit $?"; cat gpurun_out/fetch_calib.log | grep SEG
  python tools/pmc_summary.py gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/pmc_C > gpurun_out/pmc_summary.json; python -c "
import json; d=json.load(open('gpurun_out/pmc_summary.json'))
for k,v in d.items(): print(k, {c:x['per_dispatch'] for c,x in v.items()})"
  rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/pmc_C
  for t in "k8=1,xcd=1" "k8=5,xcd=1" "k8=7,xcd=1"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$t.json 2> gpurun_out/bench_$t.err
    echo "bench $t exit $?"; cut -c1-200 gpurun_out/bench_$t.json
  done
fi
if [ "$what" = "r2" ]; then
  # first GPU session of round 2: the variants written but not measured in round 1
  timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -8
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -E "sweep_fwd|fast staging|conv0 dgrad|conv2 fwd" gpurun_out/kernels.log
  for t in "" "sweep_fwd=6" "fs=1" "sweep_fwd=6,fs=1"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; cut -c1-200 "gpurun_out/bench_[$t].json"
  done
  MVS_HIP_FEATURE=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 > gpurun_out/bench_hipfeature.json 2> gpurun_out/bench_hipfeature.err
  echo "bench [MVS_HIP_FEATURE=1] exit $?"; cut -c1-200 gpurun_out/bench_hipfeature.json
fi
if [ "$what" = "runF" ]; then
  # K1 with LDS-staged depths (fwd_dl 0 / 1 / 2): parity subset, A/B in the training step and at config 5 (bf16, N = 7)
  MVS_SKIP_HEAVY=1 timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "sweep or homo or golden_mvsnet or config2_train or bf16" > gpurun_out/pytest_runF.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_runF.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_runF.log | tail -12
  for t in "fwd_dl=0" "fwd_dl=1" "fwd_dl=2"; do
    MVS_TUNING=$t timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'], {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_[$t].json"
  done
  for t in "fwd_dl=0" "fwd_dl=1"; do
    MVS_TUNING=$t timeout 200 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 > "gpurun_out/bench_c5_[$t].json" 2> "gpurun_out/bench_c5_[$t].err"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('config 5', sys.argv[1], d['ms_per_step'], d['value'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" "gpurun_out/bench_c5_[$t].json"
  done
fi
if [ "$what" = "sqk2" ]; then
  # what the waves of the sweep kernels wait for: average LDS / vector-memory / scalar-memory latency (INST_LEVEL / INSTS) and issue counts
  export MVS_PMC_SWEEP_ONLY=1
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_S$i
    (cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_S$i" -o pmc -- \
        python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_S$i.log" 2>&1); echo "pmc S$i exit $?"; tail -n 2 gpurun_out/pmc_S$i.log
  done
  python tools/pmc_summary.py gpurun_out/pmc_S1 gpurun_out/pmc_S2 > gpurun_out/pmc_sq_k2.json; python -c "
import json; d=json.load(open('gpurun_out/pmc_sq_k2.json'))
for k,v in d.items(): print(k, {c:round(x['mean']) for c,x in v.items()})"
  rm -rf gpurun_out/pmc_S1 gpurun_out/pmc_S2
fi
if [ "$what" = "runE" ]; then
  # K2 with group-ahead gradient requests + LDS-staged depths: parity subset, then A/B of the two forms in the training step
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "sweep or homo or golden_mvsnet or config2_train or config3_self" > gpurun_out/pytest_runE.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_runE.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_runE.log | tail -12
  for t in ${RUNE_SET:-"bwd_gd=0" "bwd_gd=2"}; do
    MVS_TUNING=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" "gpurun_out/bench_[$t].json"
  done
  for t in ${RUNE_SET3:-"bwd_gd=2" "bwd_pf=2"}; do
    MVS_TUNING=$t timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 > "gpurun_out/bench_c3_[$t].json" 2> "gpurun_out/bench_c3_[$t].err"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('config 3', sys.argv[1], d['ms_per_step'], d['value'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" "gpurun_out/bench_c3_[$t].json"
  done
fi
if [ "$what" = "runD" ]; then
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "conv3d_family or costregnet or golden_mvsnet or sweep or homo or config2_train" > gpurun_out/pytest_runD.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_runD.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_runD.log | tail -12
  for t in "tr2pw=0" "tr2pw=1"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --time-all-kernels > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" "gpurun_out/bench_[$t].json"; grep "ms/step" "gpurun_out/bench_[$t].err" | grep -E "16>8|8>16" | head -8
  done
  timeout 600 python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/bench_config_3.json 2> gpurun_out/bench_config_3.err; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('config 3', d['ms_per_step'], d['value'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" gpurun_out/bench_config_3.json
fi
if [ "$what" = "runC" ]; then
  for t in "conv_small_wgs=384" "conv_small_wgs=1024" "conv_small_wgs=2500" "conv_small_wgs=8000"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --time-all-kernels > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'])" "gpurun_out/bench_[$t].json"; grep "ms/step" "gpurun_out/bench_[$t].err" | grep -E "48x32x40|96x64x80" | grep -E "fwd|dgrad" | sort -k5 | head -24
  done
fi
if [ "$what" = "runB" ]; then
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --timeout 600 -k "conv3d_family or costregnet or two_ranks or golden_mvsnet or golden_cvp or cout8" > gpurun_out/pytest_runB.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_runB.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_runB.log | tail -12
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -n 3 gpurun_out/smoke.log
  for t in "conv_small=0" "conv_small=1"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --time-all-kernels > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'])" "gpurun_out/bench_[$t].json"; grep "ms/step" "gpurun_out/bench_[$t].err" | grep -E "24x16x20|48x32x40" | sort -k5 | head -30
  done
fi
if [ "$what" = "runA" ]; then
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep_bwd N=5" gpurun_out/kernels_k2.log
  timeout 900 python bench.py --config 5 --steps 10 --warmup 3 > "gpurun_out/bench_config_5.json" 2> "gpurun_out/bench_config_5.err"
  echo "bench --config 5 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()})" "gpurun_out/bench_config_5.json"
  MVS_HIP_FEATURE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 --time-all-kernels > gpurun_out/bench_hipfeature.json 2> gpurun_out/bench_hipfeature.err
  echo "bench MVS_HIP_FEATURE=1 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'])" gpurun_out/bench_hipfeature.json; grep "ms/step" gpurun_out/bench_hipfeature.err | grep -E "2d|bn_group" | head -40
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 --torch-profile gpurun_out/torch_profile.txt > /dev/null 2>&1; grep -E "miopen|Miopen|igemm|ck::|naive|batched_gemm|Name" gpurun_out/torch_profile.txt | head -30 | cut -c1-200
fi
if [ "$what" = "b2" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 --time-all-kernels > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench (default) exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], {k:(round(v['ms'],3), round(v['frac'],3), v.get('traffic')) for k,v in d['kernels'].items()}); print(d['roofline']); print(d.get('cpu_baseline')); print(d.get('reference_gpu_path'))" gpurun_out/bench.json; tail -3 gpurun_out/bench.err
  for t in "" "sweep_fwd=2"; do
    MVS_TUNING=$t timeout 600 python bench.py --config 3 --steps 20 --warmup 5 > "gpurun_out/bench_c3_[$t].json" 2> "gpurun_out/bench_c3_[$t].err"
    echo "bench config 3 [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()})" "gpurun_out/bench_c3_[$t].json"
    MVS_TUNING=$t timeout 600 python bench.py --config 5 --dtype f32 --steps 10 --warmup 3 > "gpurun_out/bench_c5f32_[$t].json" 2> "gpurun_out/bench_c5f32_[$t].err"
    echo "bench config 5 f32 [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()})" "gpurun_out/bench_c5f32_[$t].json"
  done
fi
if [ "$what" = "f34" ]; then
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --timeout 600 -k "bf16 or geo or conv2d or pyramid or featurenet or config1 or config5" -s > gpurun_out/pytest_f34.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_f34.log; grep -E "passed|failed|FAILED|Error|bf16 vs" gpurun_out/pytest_f34.log | tail -20
  for e in 0 1; do
    MVS_HIP_PYRAMID=$e timeout 900 python bench.py --config 4 --steps 10 --warmup 3 > "gpurun_out/bench_config_4_pyr$e.json" 2> "gpurun_out/bench_config_4_pyr$e.err"
    echo "bench --config 4 MVS_HIP_PYRAMID=$e exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'])" "gpurun_out/bench_config_4_pyr$e.json"
  done
  timeout 900 python bench.py --config 5 --steps 10 --warmup 3 --time-all-kernels > "gpurun_out/bench_config_5.json" 2> "gpurun_out/bench_config_5.err"
  echo "bench --config 5 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()})" "gpurun_out/bench_config_5.json"; grep "ms/step" "gpurun_out/bench_config_5.err" | head -6
fi
if [ "$what" = "bf16b" ]; then
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --timeout 600 -k "bf16 or golden_mvsnet or config1 or config5 or sweep or homo" -s > gpurun_out/pytest_bf16.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_bf16.log; grep -E "passed|failed|FAILED|Error|bf16 vs" gpurun_out/pytest_bf16.log | tail -20
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep_bwd" gpurun_out/kernels_k2.log | head -8
  for c in 5; do
    timeout 900 python bench.py --config $c --steps 10 --warmup 3 --time-all-kernels > "gpurun_out/bench_config_$c.json" 2> "gpurun_out/bench_config_$c.err"
    echo "bench --config $c exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['metric'], d['value'], d['ms_per_step'], d['dtype'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()}, d['roofline'])" "gpurun_out/bench_config_$c.json"; grep "ms/step" "gpurun_out/bench_config_$c.err" | head -12
  done
fi
if [ "$what" = "bf16" ]; then
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --timeout 600 -k "bf16 or golden_mvsnet or ms_homo" -s > gpurun_out/pytest_bf16.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_bf16.log; grep -E "passed|failed|FAILED|Error|bf16 vs" gpurun_out/pytest_bf16.log | tail -20
  for c in 5 "5 --dtype f32" 3 4; do
    timeout 900 python bench.py --config $c --steps 10 --warmup 3 --time-all-kernels > "gpurun_out/bench_config_$c.json" 2> "gpurun_out/bench_config_$c.err"
    echo "bench --config $c exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['metric'], d['value'], d['ms_per_step'], d['dtype'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items()})" "gpurun_out/bench_config_$c.json"; grep "ms/step" "gpurun_out/bench_config_$c.err" | head -24
  done
fi
if [ "$what" = "r2full" ]; then
  # full GPU parity suite (incl. the full-size config 2 / 4 / 5 cases), K2 A/B, bench
  timeout 2400 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --timeout 900 --durations=15 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|Error|worst HIP" gpurun_out/pytest_gpu.log | tail -20; grep -A16 "slowest" gpurun_out/pytest_gpu.log | head -18
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep_bwd" gpurun_out/kernels_k2.log
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --time-all-kernels > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" gpurun_out/bench.json; grep "ms/step" gpurun_out/bench.err | head -50
fi
if [ "$what" = "k2b" ]; then
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sweep or homo" > gpurun_out/pytest_k2.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_k2.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_k2.log | tail -12
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep_bwd" gpurun_out/kernels_k2.log
fi
if [ "$what" = "k2" ]; then
  # round 2: the rewritten backward of the sweep -- parity, A/B against the round-1 kernel, depth-slab sweep, bench
  timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider -k "sweep or homo or golden_mvsnet or cvp or config2 or config3" > gpurun_out/pytest_k2.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_k2.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_k2.log | tail -12
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep" gpurun_out/kernels_k2.log
  for t in "" "sweep_bwd=1" "sweep_fwd=6"; do
    MVS_TUNING=$t timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})" "gpurun_out/bench_[$t].json"
  done
fi
if [ "$what" = "sq" ] || [ "$what" = "sqsweep" ]; then
  [ "$what" = "sqsweep" ] && export MVS_PMC_SWEEP_ONLY=1
  # where the cycles of the big kernels go: SQ busy / wait / MFMA-busy / LDS counters + effective clock (GRBM_GUI_ACTIVE)
  (cd /tmp && timeout 120 rocprofv3 -L > "$OLDPWD/gpurun_out/rocprof_counters.txt" 2>&1); grep -c . gpurun_out/rocprof_counters.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_S$i
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_S$i" -o pmc -- \
        python "$OLDPWD/tools/pmc_driver.py" > "$OLDPWD/gpurun_out/pmc_S$i.log" 2>&1); echo "pmc S$i exit $?"; tail -n 2 gpurun_out/pmc_S$i.log
  done
  python tools/pmc_summary.py gpurun_out/pmc_S1 gpurun_out/pmc_S2 > gpurun_out/pmc_sq_summary.json; python -c "
import json; d=json.load(open('gpurun_out/pmc_sq_summary.json'))
for k,v in d.items(): print(k, {c:round(x['mean']) for c,x in v.items()})"
  rm -rf gpurun_out/pmc_S1 gpurun_out/pmc_S2
fi
if [ "$what" = "ks" ]; then
  MVS_BENCH_SWEEP_ONLY=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_sweep.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels_sweep.log
fi
if [ "$what" = "k" ]; then
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels.log
fi
if [ "$what" = "b" ]; then
  timeout 900 python bench.py --steps 10 --warmup 3 --time-all-kernels --torch-profile gpurun_out/torch_profile.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json | cut -c1-300; grep "ms/step" gpurun_out/bench.err | head -12
fi
if [ "$what" = "tb" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -5
  timeout 900 python bench.py --steps 10 --warmup 3 --time-all-kernels --torch-profile gpurun_out/torch_profile.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json | cut -c1-300; grep "ms/step" gpurun_out/bench.err | head -12
fi
if [ "$what" = "final" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -8
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -n 3 gpurun_out/smoke.log
  timeout 900 python bench.py --steps 20 --warmup 5 --time-all-kernels --gpu-reference 1 --torch-profile gpurun_out/torch_profile.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json; grep "ms/step" gpurun_out/bench.err | head -16
  for c in 3 4 5; do
    timeout 400 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err
    echo "bench config $c exit $?"; cut -c1-200 gpurun_out/bench_c$c.json
  done
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof exit $?"
  mkdir -p gpurun_out/prof_keep; find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  rm -rf gpurun_out/prof
  f=gpurun_out/prof_keep/trace_kernel_stats.csv; [ -f "$f" ] && grep -E "plane_sweep|conv_c8|conv_igemm|conv_wgrad|bn_" "$f" | cut -c1-160 | head -30
fi
if [ "$what" = "tests" ] || [ "$what" = "all" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -n 40 gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -n 5 gpurun_out/smoke.log
fi
if [ "$what" = "bench" ] || [ "$what" = "all" ]; then
  timeout 900 python bench.py --steps 10 --warmup 3 --time-all-kernels --gpu-reference 1 --torch-profile gpurun_out/torch_profile.txt > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json; tail -n 60 gpurun_out/bench.err
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --feature-channels-last 1 > gpurun_out/bench_featcl.json 2> gpurun_out/bench_featcl.err
  echo "bench(feature channels-last) exit $?"; cat gpurun_out/bench_featcl.json | cut -c1-260
fi
if [ "$what" = "kernels" ] || [ "$what" = "all" ]; then
  timeout 600 python tools/bench_kernels.py > gpurun_out/kernels.log 2>&1; echo "kernels exit $?"; cat gpurun_out/kernels.log | grep -v Warning
  [ -x tools/mfma_rate.bin ] && timeout 60 tools/mfma_rate.bin | tee gpurun_out/mfma_rate.log
  [ -x tools/valu_rate.bin ] && timeout 60 tools/valu_rate.bin | tee gpurun_out/valu_rate.log
fi
if [ "$what" = "prof" ] || [ "$what" = "all" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && MVS_ROCTX=1 timeout 900 rocprofv3 --kernel-trace --stats --selected-regions --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 10 --warmup 5 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof exit $?"
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 40 "$f"
  # keep the merged-back payload small: keep only the stats CSVs
  mkdir -p gpurun_out/prof_keep; find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  rm -rf gpurun_out/prof
fi
if [ "$what" = "r3a" ]; then
  # round 3, first session: the projection-table backward (K2) -- parity subset, A/B over its knobs at config-2 / N=5 shapes, step
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "sweep or homo_warp or golden_mvsnet or golden_unsup" > gpurun_out/pytest_r3a.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_r3a.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3a.log | tail -12
  MVS_BENCH_BWD_ONLY=1 timeout 600 python tools/bench_kernels.py --reps 10 > gpurun_out/kernels_k2.log 2>&1; echo "kernels exit $?"; grep -E "sweep_bwd|sweep_fwd\[cached8\]" gpurun_out/kernels_k2.log
  for t in "sweep_bwd=2" "sweep_bwd=0"; do
    MVS_TUNING=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_[$t].json" 2> "gpurun_out/bench_[$t].err"
    echo "bench [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['value'], d.get('ms_per_step_async_wgrad_off'), {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_[$t].json"
  done
fi
if [ "$what" = "r3b" ]; then
  # round 3, session B: the whole GPU suite (new: fusibile chain, tightened config-3 test), then the shared-projection forward (fwd_qs)
  timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|Error|config-3|worst" gpurun_out/pytest_gpu.log | tail -15
  for cfg in 2 3 5; do for t in "fwd_qs=0" "fwd_qs=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_c${cfg}_[$t].json" 2> "gpurun_out/bench_c${cfg}_[$t].err"
    echo "bench config $cfg [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_c${cfg}_[$t].json"
  done; done
fi
if [ "$what" = "r3c" ]; then
  # round 3, session C: persistent stride-1 implicit GEMM (conv_persist) -- parity subset + A/B at configs 2 and 4
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "conv3d_family or golden_costregnet or fusibile or smallest_volumes" > gpurun_out/pytest_r3c.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_r3c.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3c.log | tail -8
  for cfg in 2 4; do for t in "conv_persist=0" "conv_persist=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c${cfg}_[$t].json" 2> "gpurun_out/bench_c${cfg}_[$t].err"
    echo "bench config $cfg [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_c${cfg}_[$t].json"
    grep -E "s1:1x192x128x160|16>16:s1|64>64:s1|32>32:s1" "gpurun_out/bench_c${cfg}_[$t].err" | head -12
  done; done
fi
if [ "$what" = "r3d" ]; then
  # round 3, session D: the fused regulariser node (skip gradients in the dgrad epilogue, async weight gradients by default)
  MVS_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "golden_costregnet or golden_mvsnet or golden_cvp or config2_train_step or cvp_three_level or fusibile or two_ranks" > gpurun_out/pytest_r3d.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_r3d.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3d.log | tail -8
  for t in "1" "0"; do
    MVS_REG_FUSED=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_fused$t.json" 2> "gpurun_out/bench_fused$t.err"
    echo "bench MVS_REG_FUSED=$t exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), 'other wgrad mode:', d.get('ms_per_step_async_wgrad_off'), {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_fused$t.json"
  done
  MVS_REG_FUSED=1 timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 > gpurun_out/bench_c3_fused1.json 2> gpurun_out/bench_c3_fused1.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c3_fused1.json')); print('config 3', round(d['ms_per_step'],3), round(d['value'],1))"
fi
if [ "$what" = "r3e" ]; then
  for t in "1" "0" "1" "0"; do
    MVS_REG_FUSED=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_fused$t.json" 2> "gpurun_out/bench_fused$t.err"
    echo "bench MVS_REG_FUSED=$t exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), 'other wgrad mode:', d.get('ms_per_step_async_wgrad_off'), {k:round(v['ms'],4) for k,v in d['kernels'].items()})" "gpurun_out/bench_fused$t.json"
  done
fi
if [ "$what" = "r3f" ]; then
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "golden_mvsnet or config2_train_step or two_ranks" > gpurun_out/pytest_r3f.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_r3f.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3f.log | tail -8
  for t in "1" "0" "1" "0"; do
    MVS_SPLIT_CONV2D_BWD=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_split$t.json" 2> "gpurun_out/bench_split$t.err"
    echo "bench MVS_SPLIT_CONV2D_BWD=$t exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), 'other wgrad mode:', d.get('ms_per_step_async_wgrad_off'))" "gpurun_out/bench_split$t.json"
  done
fi
if [ "$what" = "r3final_a" ]; then
  timeout 1700 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|Error|config-3|worst" gpurun_out/pytest_gpu.log | tail -15
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
fi
if [ "$what" = "r3final_b" ]; then
  # round-3 validation, part B: the default bench line (cpu_baseline + reference_gpu_path + PMC), rocprofv3 stats of the timed region,
  # configs 3 / 4 / 5 with their own PMC roofline
  timeout 900 python bench.py --time-all-kernels > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cat gpurun_out/bench.json; grep "ms/step" gpurun_out/bench.err | head -14
  rm -rf gpurun_out/prof
  (cd /tmp && MVS_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --stats --selected-regions --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof exit $?"
  mkdir -p gpurun_out/prof_keep; find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  rm -rf gpurun_out/prof
  f=gpurun_out/prof_keep/trace_kernel_stats.csv; [ -f "$f" ] && head -n 14 "$f" | cut -c1-170
  for c in 3 4 5; do
    timeout 500 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --gpu-reference 0 --time-all-kernels > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err
    echo "bench config $c exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'], {k:(round(v['ms'],4), round(v.get('frac',0),3), v.get('traffic')) for k,v in d['kernels'].items()})" gpurun_out/bench_c$c.json
  done
fi
if [ "$what" = "r3final_c" ]; then
  MVS_PMC_CONFIG=5 MVS_PMC_DTYPE=bf16 timeout 200 python tools/pmc_driver.py 2>&1 | grep -v Warning | tail -12
  rm -rf gpurun_out/prof
  (cd /tmp && MVS_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --stats --selected-regions --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof exit $?"; cut -c1-120 gpurun_out/prof_bench.json
  mkdir -p gpurun_out/prof_keep; find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  rm -rf gpurun_out/prof
  f=gpurun_out/prof_keep/trace_kernel_stats.csv; [ -f "$f" ] && head -n 14 "$f" | cut -c1-170
  timeout 500 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --gpu-reference 0 --time-all-kernels > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "bench config 5 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'])" gpurun_out/bench_c5.json
fi
if [ "$what" = "r3final_d" ]; then
  rm -rf gpurun_out/prof gpurun_out/prof_keep
  (cd /tmp && MVS_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --stats --selected-regions --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof(selected regions) exit $?"; cut -c1-120 gpurun_out/prof_bench.json
  mkdir -p gpurun_out/prof_keep
  [ -d gpurun_out/prof ] && find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  rm -rf gpurun_out/prof
  if [ ! -f gpurun_out/prof_keep/trace_kernel_stats.csv ]; then
    echo "selected regions recorded nothing: whole run, 100 steps"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
        python "$OLDPWD/bench.py" --steps 100 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
    echo "prof(whole run) exit $?"; cut -c1-120 gpurun_out/prof_bench.json
    find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
    rm -rf gpurun_out/prof
  fi
  f=gpurun_out/prof_keep/trace_kernel_stats.csv; [ -f "$f" ] && head -n 16 "$f" | cut -c1-170
  timeout 500 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --gpu-reference 0 --time-all-kernels > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "bench config 5 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'], {k:(round(v['ms'],4), round(v.get('frac',0),3), v.get('traffic')) for k,v in d['kernels'].items()})" gpurun_out/bench_c5.json
fi
if [ "$what" = "r3g" ]; then
  # round 3: XCD-compact workgroup order of the sweep kernels (sweep_xcd) and the merged re-gather forward (fwd_dl=2)
  MVS_SKIP_HEAVY=1 timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "sweep or homo_warp or golden_mvsnet" > gpurun_out/pytest_r3g.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3g.log | tail -5
  for cfg in 5 3 2 4; do for t in "sweep_xcd=0" "sweep_xcd=1" "fwd_dl=2" "sweep_xcd=1,fwd_dl=2"; do
    MVS_TUNING=$t timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "gpurun_out/bench_c${cfg}_[$t].json" 2> "gpurun_out/bench_c${cfg}_[$t].err"
    echo "bench config $cfg [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:round(v['ms'],4) for k,v in d['kernels'].items() if 'sweep' in k})" "gpurun_out/bench_c${cfg}_[$t].json"
  done; done
fi
if [ "$what" = "r3h" ]; then
  # round 3: inference FeatureNet with BatchNorm folded into the csrc/conv2d.hip convolutions
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "featurenet or config1_eval or config5_shape or bf16_inference or refinenet or golden_mvsnet or geo" > gpurun_out/pytest_r3h.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3h.log | tail -8
  for cfg in 5; do for f in 0 1; do
    MVS_FOLD_EVAL_BN=$f timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c${cfg}_fold$f.json" 2> "gpurun_out/bench_c${cfg}_fold$f.err"
    echo "bench config $cfg fold=$f exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c${cfg}_fold$f.json"; grep "ms/step" "gpurun_out/bench_c${cfg}_fold$f.err" | head -24
  done; done
  MVS_FOLD_EVAL_BN=1 timeout 300 python bench.py --config 5 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_c5_f32_fold1.json 2>/dev/null; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('f32', round(d['ms_per_step'],3), round(d['value'],1))" gpurun_out/bench_c5_f32_fold1.json
fi
if [ "$what" = "r3i" ]; then
  # round 3: CVP pyramid with all views as one batch; SQ counters of the bf16 conv0 at config 5
  MVS_SKIP_HEAVY=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "cvp" > gpurun_out/pytest_r3i.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3i.log | tail -5
  for f in 0 1; do
    MVS_CVP_BATCH_VIEWS=$f timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c4_batch$f.json" 2> "gpurun_out/bench_c4_batch$f.err"
    echo "bench config 4 batch_views=$f exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c4_batch$f.json"; grep "fwd2d" "gpurun_out/bench_c4_batch$f.err" | head -12
  done
  export MVS_PMC_CONFIG=5 MVS_PMC_DTYPE=bf16; bash tools/gpu_round.sh sq 2>&1 | grep "conv_bf16"
fi
if [ "$what" = "r3j" ]; then
  # round 3: BatchNorm reductions finished by their last workgroup (no separate finalize launches)
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "bn or batchnorm or golden or featurenet or conv_bn or regulariser or two_rank or train_step" > gpurun_out/pytest_r3j.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3j.log | tail -8
  for cfg in 2 3; do
    timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c${cfg}_bnfin.json" 2> "gpurun_out/bench_c${cfg}_bnfin.err"
    echo "bench config $cfg exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), d.get('ms_per_step_async_wgrad_off'))" "gpurun_out/bench_c${cfg}_bnfin.json"; grep "mvs_bn" "gpurun_out/bench_c${cfg}_bnfin.err" | head -8
  done
fi
if [ "$what" = "r3final2" ]; then
  # closing validation of the round's final code: every GPU test except the three full-size oracle comparisons (those ran on the
  # head of r3final_a; the kernels they exercise are unchanged since), smoke, the default bench line, configs 3 / 4 / 5
  MVS_SKIP_HEAVY=1 timeout 1200 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -6
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
  timeout 900 python bench.py --time-all-kernels > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?"; cut -c1-400 gpurun_out/bench.json
  for c in 3 4 5; do
    timeout 500 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --gpu-reference 0 --time-all-kernels > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err
    echo "bench config $c exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:(round(v['ms'],4), round(v.get('frac',0),3), v.get('traffic')) for k,v in d['kernels'].items()})" gpurun_out/bench_c$c.json
  done
fi
if [ "$what" = "r3k" ]; then
  # round 3: the Cout = 1 (probability) layer, its input gradient and its bf16 form with four outputs per thread (knob cout1_d4)
  MVS_SKIP_HEAVY=1 MVS_TUNING=cout1_d4=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "conv3d_family or golden_costregnet or bf16_inference or conv3d_bf16 or smallest" > gpurun_out/pytest_r3k.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3k.log | tail -5
  for cfg in 2 4 5; do for t in "cout1_d4=0" "cout1_d4=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c${cfg}_[$t].json" 2> "gpurun_out/bench_c${cfg}_[$t].err"
    echo "bench config $cfg [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c${cfg}_[$t].json"; grep -E ">1:s1" "gpurun_out/bench_c${cfg}_[$t].err" | head -6
  done; done
fi
if [ "$what" = "r3l" ]; then
  # round 3: transposed bf16 convolution 16 -> 8 with both W parities in one MFMA (GEOM_TR2_PW), config 5
  MVS_SKIP_HEAVY=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "bf16" > gpurun_out/pytest_r3l.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3l.log | tail -5
  for t in "tr2pw=0" "tr2pw=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c5_[$t].json" 2> "gpurun_out/bench_c5_[$t].err"
    echo "bench config 5 [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c5_[$t].json"; grep -E "fwdT_bf16" "gpurun_out/bench_c5_[$t].err" | head -4
  done
fi
if [ "$what" = "r3m" ]; then
  # round 3: conv0 of the bf16 path with two output depth slices per MFMA (GEOM_S1_DP), config 5
  MVS_SKIP_HEAVY=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "bf16" > gpurun_out/pytest_r3m.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3m.log | tail -5
  for t in "bf16_dp=0" "bf16_dp=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c5_[$t].json" 2> "gpurun_out/bench_c5_[$t].err"
    echo "bench config 5 [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c5_[$t].json"; grep -E "fwd_bf16:32>8" "gpurun_out/bench_c5_[$t].err" | head -2
  done
fi
if [ "$what" = "r3n" ]; then
  # round 3: narrow 2-D convolutions (3 -> 8, 8 -> 8) as pixel-pair GEMMs (knob conv2d_pp), config 5
  MVS_SKIP_HEAVY=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "conv2d or featurenet or refinenet or pyramid or config1_eval" > gpurun_out/pytest_r3n.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r3n.log | tail -5
  for t in "conv2d_pp=0" "conv2d_pp=1"; do
    MVS_TUNING=$t timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c5_[$t].json" 2> "gpurun_out/bench_c5_[$t].err"
    echo "bench config 5 [$t] exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c5_[$t].json"; grep -E "fwd2d" "gpurun_out/bench_c5_[$t].err" | head -8
  done
fi
if [ "$what" = "r3final3" ]; then
  # closing validation after the inference-path kernel changes: every GPU test except the three full-size oracle comparisons,
  # smoke, config 2 (short line) and config 5 (with PMC)
  MVS_SKIP_HEAVY=1 timeout 600 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -6
  timeout 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
  timeout 200 python bench.py --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
  echo "bench exit $?"; cut -c1-160 gpurun_out/bench_short.json
  timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --gpu-reference 0 --time-all-kernels > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "bench config 5 exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:(round(v['ms'],4), round(v.get('frac',0),3), v.get('traffic')) for k,v in d['kernels'].items()})" gpurun_out/bench_c5.json
fi
if [ "$what" = "r3o" ]; then
  for f in 0 1; do
    MVS_HIP_FEATURE_FWD=$f timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_hipfwd$f.json 2> gpurun_out/bench_hipfwd$f.err
    echo "hip_fwd_train=$f exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), d['final_loss'])" gpurun_out/bench_hipfwd$f.json
  done
fi
if [ "$what" = "r4a" ]; then
  # first session of round 4: what round 3 built last and could not time -- the 2-D extractor's forward convolution through
  # conv2d.hip with BatchNorm statistics in its epilogue (MVS_HIP_FEATURE_FWD), configs 2 and 3; then the 2-D side-stream
  # mismatch (MVS_SPLIT_CONV2D_BWD=1 under the side-stream mode) as a parity question, not a timing one
  MVS_SKIP_HEAVY=1 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "featurenet" > gpurun_out/pytest_r4a.log 2>&1
  echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_r4a.log | tail -4
  for cfg in 2 3; do for f in 0 1; do
    MVS_HIP_FEATURE_FWD=$f timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --time-all-kernels > "gpurun_out/bench_c${cfg}_hipfwd$f.json" 2> "gpurun_out/bench_c${cfg}_hipfwd$f.err"
    echo "config $cfg MVS_HIP_FEATURE_FWD=$f exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1))" "gpurun_out/bench_c${cfg}_hipfwd$f.json"; grep -E "fwd2d|bn_group" "gpurun_out/bench_c${cfg}_hipfwd$f.err" | head -10
  done; done
fi
if [ "$what" = "r4a" ]; then
  # round 4, first session: statistic-slot BatchNorm + batch weight packing -- targeted parity, step time, kernel table, A/B of the HIP 2-D forward
  timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider \
    -k "conv3d_family or cout8 or smallest_volumes or costregnet or mvsnet_end_to_end or config2_train_step or featurenet_training or featurenet_hip or cvpmvsnet_end_to_end or config1" \
    > gpurun_out/pytest_r4a.log 2>&1
  echo "pytest exit $?"; tail -5 gpurun_out/pytest_r4a.log
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "feature_fwd" > gpurun_out/bench_r4a.json 2> gpurun_out/bench_r4a.err
  echo "bench exit $?"; cut -c1-400 gpurun_out/bench_r4a.json; grep "A/B" gpurun_out/bench_r4a.err
  timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4a_k.json 2> gpurun_out/bench_r4a_k.err
  echo "bench-k exit $?"; grep "ms/step" gpurun_out/bench_r4a_k.err | head -60
  MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_r4a.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels_r4a.log | tail -40
fi
if [ "$what" = "r4b" ]; then
  # round 4, second session: prologue / epilogue fixes, deferred join, side-input prefetch A/B
  timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider \
    -k "dgrad_with_summand or conv3d_family or cout8 or costregnet or mvsnet_end_to_end or config2_train_step or featurenet_training or cvpmvsnet_end_to_end" \
    > gpurun_out/pytest_r4b.log 2>&1
  echo "pytest exit $?"; tail -5 gpurun_out/pytest_r4b.log
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "defer_join;side_pre=0;feature_fwd" > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err
  echo "bench exit $?"; cut -c1-400 gpurun_out/bench_r4b.json; grep "A/B" gpurun_out/bench_r4b.err
  timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4b_k.json 2> gpurun_out/bench_r4b_k.err
  echo "bench-k exit $?"; grep "ms/step" gpurun_out/bench_r4b_k.err | head -60
  MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_r4b.log 2>&1; echo "kernels exit $?"; grep -v Warn gpurun_out/kernels_r4b.log | tail -32
fi
if [ "$what" = "r4c" ]; then
  # round 4, third session: clean line + A/B (side_pre, graph), rocprofv3 kernel trace of the step (durations and gaps)
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "side_pre=0;defer_join" > gpurun_out/bench_r4c.json 2> gpurun_out/bench_r4c.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4c.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","ms_per_step_async_wgrad_off","wgrad_join")}, d.get("ab"))
PY
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --graph 1 > gpurun_out/bench_r4c_graph.json 2> gpurun_out/bench_r4c_graph.err
  echo "graph bench exit $?"; cut -c1-330 gpurun_out/bench_r4c_graph.json; tail -3 gpurun_out/bench_r4c_graph.err
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err")
  echo "prof exit $?"; cut -c1-200 gpurun_out/prof_bench.json
  mkdir -p gpurun_out/prof_keep; find gpurun_out/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof_keep/ \;
  # the last 3 steps of the kernel trace (start/end per kernel): gaps and overlap
  python tools/trace_tail.py gpurun_out/prof gpurun_out/prof_keep/r4c_trace_tail.csv 700
  rm -rf gpurun_out/prof
fi
if [ "$what" = "r4d" ]; then
  # round 4, fourth session: launch-count items (materialised grads, 2-D batch pack, fused loss), branch-free cin1, scalar-weight cout1
  timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider \
    -k "dgrad_with_summand or conv3d_family or smallest_volumes or costregnet_mvs or mvsnet_end_to_end or config2_train_step or featurenet_training or mvsnet_loss" \
    > gpurun_out/pytest_r4d.log 2>&1
  echo "pytest exit $?"; tail -4 gpurun_out/pytest_r4d.log
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "defer_join" --ab-reps 3 > gpurun_out/bench_r4d.json 2> gpurun_out/bench_r4d.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4d.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","ms_per_step_async_wgrad_off","wgrad_join")}, d.get("ab"))
print({k:(round(v["ms"],4), round(v.get("frac",0),3)) for k,v in d["kernels"].items()}, d["roofline"])
PY
  timeout 600 python bench.py --steps 10 --warmup 3 --time-all-kernels --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4d_k.json 2> gpurun_out/bench_r4d_k.err
  echo "bench-k exit $?"; grep "ms/step" gpurun_out/bench_r4d_k.err | head -24
  MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_r4d.log 2>&1; echo "kernels exit $?"; grep -E "prob|conv1 dgrad" gpurun_out/kernels_r4d.log
fi
if [ "$what" = "r4e" ]; then
  # round 4, fifth session: several side streams, 2-D weight gradients on them, wide wgrad reduction, cout1 back on LDS weights
  timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider \
    -k "conv3d_family or costregnet_mvs or mvsnet_end_to_end or config2_train_step or featurenet_training" > gpurun_out/pytest_r4e.log 2>&1
  echo "pytest exit $?"; tail -3 gpurun_out/pytest_r4e.log
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "wgrad_streams=2;wgrad_streams=3;wgrad_streams=4;split_bwd" --ab-reps 3 > gpurun_out/bench_r4e.json 2> gpurun_out/bench_r4e.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4e.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","ms_per_step_async_wgrad_off","wgrad_join")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
  MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_r4e.log 2>&1; echo "kernels exit $?"; grep -E "prob|wgrad" gpurun_out/kernels_r4e.log
fi
if [ "$what" = "r4f" ]; then
  # round 4, sixth session: low-priority side stream A/B, cleanup regression (sweep variants removed), other configs
  timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider \
    -k "plane_sweep or filter_depth_scan or batch3 or mvsnet_loss or mvsnet_end_to_end or config2_train_step or cvpmvsnet_end_to_end or bf16_inference" > gpurun_out/pytest_r4f.log 2>&1
  echo "pytest exit $?"; tail -3 gpurun_out/pytest_r4f.log
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "side_low" --ab-reps 3 > gpurun_out/bench_r4f.json 2> gpurun_out/bench_r4f.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4f.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","ms_per_step_async_wgrad_off")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
  for cfg in 3 4 5; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4f_c$cfg.json 2> gpurun_out/bench_r4f_c$cfg.err
    echo "config $cfg exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:(round(v['ms'],4), round(v.get('frac',0),3)) for k,v in d['kernels'].items()})" gpurun_out/bench_r4f_c$cfg.json
  done
fi
if [ "$what" = "r4g" ]; then
  # round 4, session 7: quarter-size tiles in the generic weight-gradient kernels
  timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "conv3d_family or costregnet_mvs" > gpurun_out/pytest_r4g.log 2>&1
  echo "pytest exit $?"; tail -2 gpurun_out/pytest_r4g.log
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "wgrad_small=1;wgrad_small=2" --ab-reps 3 > gpurun_out/bench_r4g.json 2> gpurun_out/bench_r4g.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4g.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
  for m in 0 1 2; do echo "wgrad_small=$m"; MVS_TUNING="wgrad_small=$m" MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "conv1 wgrad|conv2 wgrad|prob wgrad"; done
fi
if [ "$what" = "r4h" ]; then
  # round 4, session 8: host-side trims (no stream switch for side-stream wgrads, cached pack plans), wgrad workgroup-count knobs
  timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "costregnet_mvs or mvsnet_end_to_end or config2_train_step or featurenet_training" > gpurun_out/pytest_r4h.log 2>&1
  echo "pytest exit $?"; tail -2 gpurun_out/pytest_r4h.log
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "wgrad_groups=256;wgrad_groups=512;wgrad8_groups=256" --ab-reps 3 > gpurun_out/bench_r4h.json 2> gpurun_out/bench_r4h.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4h.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","host_enqueue_ms_per_step_median_max","ms_per_step_async_wgrad_off")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4h_2.json 2> gpurun_out/bench_r4h_2.err
  python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4h_2.json"))
print("second process:", {k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","host_enqueue_ms_per_step_median_max")})
PY
fi
if [ "$what" = "r4i" ]; then
  timeout 600 python tools/bench_conv2d.py > gpurun_out/conv2d_layers_r4i.log 2>&1; echo "conv2d exit $?"; grep -v Warn gpurun_out/conv2d_layers_r4i.log | tail -32
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "wgrad8_groups=128;wgrad8_groups=256;wgrad8_groups=384" --ab-reps 3 > gpurun_out/bench_r4i.json 2> gpurun_out/bench_r4i.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4i.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
fi
if [ "$what" = "r4j" ]; then
  timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "featurenet or conv2d_family or mvsnet_end_to_end or config2_train_step" > gpurun_out/pytest_r4j.log 2>&1
  echo "pytest exit $?"; tail -2 gpurun_out/pytest_r4j.log
  timeout 600 python tools/bench_conv2d.py > gpurun_out/conv2d_layers_r4j.log 2>&1; echo "conv2d exit $?"; grep -E "weight grad|TOTAL" gpurun_out/conv2d_layers_r4j.log
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "feature_dgrad;feature_wgrad;wgrad8_groups=64;wgrad8_groups=192" --ab-reps 3 > gpurun_out/bench_r4j.json 2> gpurun_out/bench_r4j.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4j.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","host_enqueue_ms_per_step_median_max")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
fi
if [ "$what" = "r4k" ]; then
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "fork_early=1;fork_early=2;split_bwd;wgrad_groups=512" --ab-reps 3 > gpurun_out/bench_r4k.json 2> gpurun_out/bench_r4k.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4k.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","host_enqueue_ms_per_step_median_max")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
fi
if [ "$what" = "full" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu_full.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu_full.log; grep -E "passed|failed|FAILED|pytest exit" gpurun_out/pytest_gpu_full.log | tail -8
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
fi
if [ "$what" = "heavyfast" ]; then
  MIOPEN_FIND_MODE=FAST timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 -k "config5_shape_seven_views_eval" > gpurun_out/pytest_heavyfast.log 2>&1
  echo "pytest exit $?"; tail -12 gpurun_out/pytest_heavyfast.log
fi
if [ "$what" = "r4l" ]; then
  for pf in 0 2; do
    MVS_TUNING="bwd_pf=$pf" timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 > gpurun_out/bench_r4l_c3_pf$pf.json 2> gpurun_out/bench_r4l_c3_pf$pf.err
    echo "config 3 bwd_pf=$pf exit $?"; python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['ms_per_step'],3), round(d['value'],1), {k:(round(v['ms'],4), round(v.get('frac',0),3)) for k,v in d['kernels'].items()})" gpurun_out/bench_r4l_c3_pf$pf.json
  done
  timeout 300 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "wide_depth_range or config3_shape" 2>&1 | tail -2
fi
if [ "$what" = "r4m" ]; then
  timeout 300 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "cout8 or costregnet_mvs" 2>&1 | tail -2
  timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "wgrad8_nch=2;wgrad8_nch=2,wgrad8_groups=256" --ab-reps 3 > gpurun_out/bench_r4m.json 2> gpurun_out/bench_r4m.err
  echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r4m.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","host_enqueue_ms_per_step_median_max")})
for k,v in d.get("ab",{}).items(): print(k, v["median_default_ms"], v["median_toggled_ms"], v["default_ms"], v["toggled_ms"])
PY
  for m in 1 2; do echo "wgrad8_nch=$m"; MVS_TUNING="wgrad8_nch=$m,wgrad8_groups=256" MVS_BENCH_SKIP_SWEEP=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "conv0 wgrad \[4x4x1 broadcast operand, XCD"; done
fi
