"""Summarise a rocprofv3 rocpd database: per-kernel totals inside the training-step window of bench.py.

usage: python tools/rocpd_summary.py <results.db> [--steps N] [--csv out.csv]
The window is bounded by optimizer launches (multi_tensor_apply): the last N of them delimit N training steps.
"""
import argparse
import collections
import re
import sqlite3


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=50)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[0]]
    # group optimizer launches into steps (launches closer than 1 ms belong to one step)
    groups = []
    for i in opt:
        if groups and rows[i][1] - rows[groups[-1][-1]][2] < 1_000_000:
            groups[-1].append(i)
        else:
            groups.append([i])
    assert len(groups) > a.steps, (len(groups), a.steps)
    lo = groups[-a.steps - 1][-1] + 1
    hi = groups[-1][-1] + 1
    win = rows[lo:hi]
    wall = (win[-1][2] - win[0][1]) / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e in win:
        k = short(n)
        agg[k][0] += 1
        agg[k][1] += e - s
    tot = sum(v[1] for v in agg.values()) / 1e6
    print(f"{a.steps} training steps: wall {wall / a.steps:.2f} ms/step, kernel-busy {tot / a.steps:.2f} ms/step, "
          f"{len(win) / a.steps:.0f} launches/step")
    out = sorted(agg.items(), key=lambda x: -x[1][1])
    for n, (k, t) in out[:a.top]:
        print(f"{t / 1e6 / a.steps:8.3f} ms/step {100 * t / 1e6 / tot:5.1f}%  {k / a.steps:7.1f}/step {t / k / 1e3:8.1f} us  {n}")
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("kernel,launches_per_step,ms_per_step,avg_us,percent\n")
            for n, (k, t) in out:
                f.write(f"\"{n}\",{k / a.steps:.1f},{t / 1e6 / a.steps:.4f},{t / k / 1e3:.2f},{100 * t / 1e6 / tot:.2f}\n")


if __name__ == "__main__":
    main()
