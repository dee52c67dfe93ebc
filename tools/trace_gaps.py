"""Where a training step's GPU time goes BETWEEN kernels: reads a rocprofv3 --kernel-trace CSV of bench.py and reports, per
steady-state step (delimited by the fused Adam kernel), the wall time, each queue's busy time, the time no queue is busy, and the
kernels that most often start after an idle interval on their queue.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --steps 10 --warmup 5 ...
    python tools/trace_gaps.py $OUT > profiles/rNN_trace_gaps.txt
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.search(r"Adam|adam", name)
    return "fused_adam" if m else name.split("(")[0]


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adam" in r[3].lower()]
    # one optimiser launch may be several kernels: keep the LAST of each run of adam kernels
    ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 8]
    print("%d kernels, %d optimiser steps found" % (len(rows), len(ends)))
    if len(ends) < 6:
        return
    use = ends[-8:-1] if len(ends) >= 9 else ends[1:-1]     # the last steps of the timed region
    after_gap = collections.defaultdict(lambda: [0, 0.0])
    for a, b in zip(use[:-1], use[1:]):
        step = rows[a + 1:b + 1]
        t0, t1 = rows[a][1], rows[b][1]
        per_q = collections.defaultdict(list)
        for r in step:
            per_q[r[2]].append(r)
        # union of busy intervals over all queues
        ev = sorted((max(r[0], t0), min(r[1], t1)) for r in step)
        busy, cur_s, cur_e = 0, None, None
        for s, e in ev:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        line = "step %.3f ms: %d kernels, some queue busy %.3f ms, no queue busy %.3f ms" % ((t1 - t0) / 1e6, len(step), busy / 1e6, (t1 - t0 - busy) / 1e6)
        for q, rs in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
            qb = sum(r[1] - r[0] for r in rs)
            gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
            pos = [g for g in gaps if g > 0]
            line += "\n    queue %s: %d kernels, busy %.3f ms, %d gaps summing %.3f ms (median %.2f us)" % (
                q, len(rs), qb / 1e6, len(pos), sum(pos) / 1e6, (sorted(pos)[len(pos) // 2] / 1e3) if pos else 0.0)
            for i, g in enumerate(gaps):
                if g > 0:
                    k = after_gap[(q, rs[i + 1][3][:70], rs[i][3][:50])]
                    k[0] += 1
                    k[1] += g / 1e3
        print(line)
    print("\nlargest idle intervals by (queue, kernel that follows, kernel before): count, total us over %d steps" % (len(use) - 1))
    for k, v in sorted(after_gap.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %8.1f us  x%-3d  q%s  %-70s after %s" % (v[1], v[0], k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
