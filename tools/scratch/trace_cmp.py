import json, sys
d = json.load(open(sys.argv[1]))
marks, rows = d["marks"], d["rows"]
def step_rows(i):
    return [(t - marks[i], n, tag) for t, n, tag in rows if marks[i] <= t < marks[i + 1]]
a, b = step_rows(0), step_rows(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
print(len(a), len(b))
pa = pb = 0.0
for (ta, na, ga), (tb, nb, gb) in zip(a, b):
    da, db = ta - pa, tb - pb
    flag = " <<<" if da - db > 0.03e-3 else ""
    print("%-40s %-34s %8.1f %8.1f   cum %8.1f %8.1f%s" % (na[:40], ga[:34], da * 1e6, db * 1e6, ta * 1e6, tb * 1e6, flag))
    pa, pb = ta, tb
