set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
for pl in 0 1 8 0 1 8; do
timeout 600 python bench.py --steps 20 --warmup 5 $S --step-events 1 --pre-launch $pl > gpurun_out/run23_$pl.json 2> gpurun_out/run23.err; echo "prelaunch $pl exit $?"
python - gpurun_out/run23_$pl.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")}, "gpu", d["step_gpu_ms"][:4], "host", d["step_host_ms"][:3], d["step_host_fwd_bwd_rest_ms"][:2])
PY
done
