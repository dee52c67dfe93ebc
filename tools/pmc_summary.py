#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: mean counter value per kernel name.
    python tools/pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout"""
import csv
import glob
import json
import os
import sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if not any(k in name for k in ("plane_sweep", "conv_c8", "conv_igemm", "conv_wgrad")):
                    continue
                short = name.split("(")[0].replace("void ", "")
                key = (short, row.get("Counter_Name", ""))
                out.setdefault(key, []).append(float(row.get("Counter_Value", 0)))
res = {}
for (k, c), v in sorted(out.items()):
    res.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(res, indent=1))
