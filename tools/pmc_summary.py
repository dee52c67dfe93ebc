#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per kernel name, the counter value of every dispatch (in dispatch order) and the mean.
    python tools/pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout
bench.py imports summarise() for its roofline.traffic figure."""
import csv
import glob
import json
import os
import sys

KERNELS = ("plane_sweep", "conv_c8", "conv_igemm", "conv_pers", "conv_wgrad", "conv_bf16", "fetch_calib")


def summarise(dirs):
    out = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name", "")
                    if not any(k in name for k in KERNELS):
                        continue
                    short = name.split("(")[0].replace("void ", "")
                    key = (short, row.get("Counter_Name", ""))
                    out.setdefault(key, []).append((int(row.get("Dispatch_Id", 0) or 0), float(row.get("Counter_Value", 0))))
    res = {}
    for (k, c), v in sorted(out.items()):
        per = {}   # one row per (dispatch, counter instance): sum the instances of a dispatch
        for d_id, val in v:
            per[d_id] = per.get(d_id, 0.0) + val
        vals = [per[d] for d in sorted(per)]
        res.setdefault(k, {})[c] = {"mean": sum(vals) / len(vals), "n": len(vals), "per_dispatch": [round(x, 1) for x in vals]}
    return res


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1:]), indent=1))
