set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
for w in 5 5 40; do
timeout 600 python bench.py --steps 40 --warmup $w $S --step-events 1 > gpurun_out/run16_w$w.json 2> gpurun_out/run16.err; echo "exit $?"
python - gpurun_out/run16_w$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step","warmup")})
print("gpu", d["step_gpu_ms"])
print("host", d["step_host_ms"])
PY
done
