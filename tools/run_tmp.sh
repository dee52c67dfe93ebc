cd /root/repo
MVS_NARROW_ONLY="conv0 wgrad" timeout 300 python tools/bench_narrow.py "wgrad8_gs=3,xcd=0" "wgrad8_gs=3,xcd=0" "wgrad8_gs=5,xcd=0" "wgrad8_gs=6,xcd=0" "wgrad8_gs=4,xcd=0" "wgrad8_gs=4,xcd=1" 2>&1 | grep -v amdgpu
