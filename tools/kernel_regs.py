#!/usr/bin/env python3
"""Register / LDS / scratch usage per kernel from hipcc's assembly (--cuda-device-only -S): python tools/kernel_regs.py file.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    try:
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt and flt not in name:
        continue
    vg = re.search(r'\.vgpr_count:\s+(\d+)', b).group(1)
    ag = b.split('\n')[0].strip()
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', b).group(1)
    sc = re.search(r'\.private_segment_fixed_size:\s+(\d+)', b).group(1)
    print("%-78s vgpr %3s agpr %3s lds %6s scratch %s" % (name[:78], vg, ag, lds, sc))
