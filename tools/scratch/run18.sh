set -u
export TMPDIR=/tmp
S="--no-cpu-baseline --pmc 0 --gpu-reference 0"
timeout 600 python bench.py --steps 20 --warmup 5 $S --step-events 2 > gpurun_out/run18.json 2> gpurun_out/run18.err; echo "exit $?"
python tools/scratch/trace_cmp.py gpurun_out/host_trace.json 6 > gpurun_out/run18_trace_cmp.txt
grep "<<<" gpurun_out/run18_trace_cmp.txt | head -60
