set -u
export TMPDIR=/tmp
timeout 600 python tools/bench_conv2d.py > gpurun_out/run21_conv2d_layers.log 2>&1; grep -v Warn gpurun_out/run21_conv2d_layers.log | tail -17
