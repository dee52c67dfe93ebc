#!/usr/bin/env python3
"""Keep the last N kernel dispatches of a rocprofv3 --kernel-trace CSV (name, stream/queue, start, end in ns relative to the
first kept row): python tools/trace_tail.py <rocprof output dir> <out.csv> [N]"""
import csv
import glob
import os
import sys

src, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 600
files = [f for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)]
if not files:
    sys.exit("no kernel_trace.csv under %s" % src)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "queue", "stream", "start_us", "end_us", "dur_us"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        w.writerow([r["Kernel_Name"][:60], r.get("Queue_Id", ""), r.get("Stream_Id", ""), "%.2f" % (s / 1e3), "%.2f" % (e / 1e3), "%.2f" % ((e - s) / 1e3)])
print("kept", len(rows), "dispatches ->", out)
