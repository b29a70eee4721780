"""Timeline view of the training steps in a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: per queue busy
time, the time during which NO kernel runs anywhere (GPU idle: the host or a dependency is the bottleneck there), the
time during which exactly one / two or more queues run, and the largest idle gaps with their neighbours.

usage: python tools/timeline.py <kernel_trace.csv> [--steps N]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")))
    rows.sort(key=lambda r: r[1])
    opt = [i for i, r in enumerate(rows) if "FusedSgd" in r[0] or "fused_sgd" in r[0].lower()]
    groups = []
    for i in opt:
        if groups and rows[i][1] - rows[groups[-1][-1]][2] < 1_000_000:
            groups[-1].append(i)
        else:
            groups.append([i])
    first, last = groups[-a.steps - 1][-1] + 1, groups[-1][-1]
    win = rows[first:last + 1]
    t0, t1 = win[0][1], win[-1][2]
    span = (t1 - t0) / 1e6 / a.steps
    per_q = collections.defaultdict(float)
    events = []
    for n, s, e, q in win:
        per_q[q] += e - s
        events.append((s, 1, q))
        events.append((e, -1, q))
    events.sort()
    depth, prev, hist = 0, t0, collections.defaultdict(float)
    active = collections.Counter()
    for t, d, q in events:
        nq = sum(1 for v in active.values() if v > 0)
        hist[min(nq, 3)] += t - prev
        prev = t
        active[q] += d
    print(f"{a.steps} steps, {span:.2f} ms per step wall; per step: idle {hist[0] / 1e6 / a.steps:.2f} ms, one queue "
          f"{hist[1] / 1e6 / a.steps:.2f} ms, two {hist[2] / 1e6 / a.steps:.2f} ms, three or more {hist[3] / 1e6 / a.steps:.2f} ms")
    for q, ns in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: {ns / 1e6 / a.steps:.2f} ms busy per step")
    # largest gaps with nothing running
    ends = sorted(win, key=lambda r: r[1])
    gaps, cur_end, cur_name = [], ends[0][2], ends[0][0]
    for n, s, e, q in ends[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, short(cur_name), short(n)))
        if e > cur_end:
            cur_end, cur_name = e, n
    gaps.sort(reverse=True)
    print("largest idle gaps (us): after -> before")
    for g, a_, b_ in gaps[:25]:
        print(f"  {g / 1e3:8.1f}  {a_}  ->  {b_}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for g, a_, b_ in gaps:
        agg[(a_, b_)][0] += 1
        agg[(a_, b_)][1] += g
    print("idle gaps summed by neighbour pair (per step):")
    for (a_, b_), (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {ns / 1e3 / a.steps:8.1f} us in {c / a.steps:5.1f} gaps  {a_}  ->  {b_}")


if __name__ == "__main__":
    main()
