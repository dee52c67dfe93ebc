set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
for i in 1 2; do
for m in all dominant; do
timeout 600 python bench.py --steps 20 --warmup 5 $S --region-timers $m > gpurun_out/run15_$m$i.json 2> gpurun_out/run15_$m$i.err; echo "$m exit $?"
python - gpurun_out/run15_$m$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","value","host_enqueue_ms_per_step")}, d["roofline"]["kernel"], round(d["roofline"]["ms"],4), d["roofline"]["timed"])
print({k:round(v["ms"],4) for k,v in d["kernels"].items()})
PY
done; done
