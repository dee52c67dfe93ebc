set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
for b in 1 0; do
MVS_FEATURE_WGRAD_BATCH=$b timeout 600 python bench.py --steps 30 --warmup 5 $S --step-events 1 > gpurun_out/run22_$b.json 2> gpurun_out/run22.err; echo "batch $b exit $?"
python - gpurun_out/run22_$b.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","side_stream_lag_at_join_ms_median_max")})
g=sorted(d["step_gpu_ms"]); print("gpu median", g[len(g)//2], "first", d["step_gpu_ms"][:4])
PY
done
