"""Ordered kernel sequence between two launches of a marker kernel (one encoder layer's backward or forward) in a
`rocprofv3 --kernel-trace --output-format csv` run: start offset, duration, gap to the previous kernel of the same queue.

usage: python tools/layer_chain.py <kernel_trace.csv> <marker regex> [occurrence]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    trace, pat = sys.argv[1], re.compile(sys.argv[2])
    occ = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"]),
                         r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if pat.search(r[3])]
    a, b = marks[occ - 1], marks[occ]
    t0 = rows[a][0]
    last_end = {}
    print(f"{b - a} kernels, {(rows[b][0] - t0) / 1e3:.1f} us from marker to marker")
    for s, e, q, n, gx, wx in rows[a:b + 1]:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"q{q} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {n}  grid {gx}/{wx}")


if __name__ == "__main__":
    main()
