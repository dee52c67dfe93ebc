set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "featurenet or golden_mvsnet or config2_train or jdacs" -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pmc 0 --gpu-reference 0 --ab "feature_one_node" --ab-reps 5 > gpurun_out/run14_bench.json 2> gpurun_out/run14_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/run14_bench.json"))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")})
for k,v in d.get("ab",{}).items(): print("A/B",k,v)
PY
