set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
for sp in 0 30 0 30; do
timeout 600 python bench.py --steps 20 --warmup 5 $S --step-events 1 --pre-spin-ms $sp > gpurun_out/run17_$sp.json 2> gpurun_out/run17.err; echo "spin $sp exit $?"
python - gpurun_out/run17_$sp.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")})
print("gpu", d["step_gpu_ms"][:8])
print("host", d["step_host_ms"][:8])
print("phases", d["step_host_fwd_bwd_rest_ms"])
PY
done
