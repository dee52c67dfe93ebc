#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per kernel name, the counter value of every dispatch (in dispatch order) and the mean.
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
                if not any(k in name for k in ("plane_sweep", "conv_c8", "conv_igemm", "conv_wgrad", "fetch_calib")):
                    continue
                short = name.split("(")[0].replace("void ", "")
                key = (short, row.get("Counter_Name", ""))
                out.setdefault(key, []).append((int(row.get("Dispatch_Id", 0) or 0), float(row.get("Counter_Value", 0))))
res = {}
for (k, c), v in sorted(out.items()):
    # one row per (dispatch, counter instance): sum the instances of a dispatch
    per = {}
    for d_id, val in v:
        per[d_id] = per.get(d_id, 0.0) + val
    vals = [per[d] for d in sorted(per)]
    res.setdefault(k, {})[c] = {"mean": sum(vals) / len(vals), "n": len(vals), "per_dispatch": [round(x, 1) for x in vals]}
print(json.dumps(res, indent=1))
