"""Fabric traffic per launch of the kernels matching a regex, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
same command: python tools/pmc_kernel_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel regex>
Correction as in tools/pmc_conv_traffic.py (MI355X_MICROARCH.md, HBM section): counter unit KiB; gfx950 counts 128-B read
requests as 64 B -> FETCH_SIZE doubled; WRITE_SIZE as is."""
import collections
import csv
import re
import sys


def per_kernel(path, counter, pat):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not pat.search(n):
            continue
        m = re.search(r"(\w+<[^>]*>)", n.replace("(anonymous namespace)::", ""))
        per[m.group(1) if m else n[:60]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per


def main():
    fetch_csv, write_csv, pat = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
    f = per_kernel(fetch_csv, "FETCH_SIZE", pat)
    w = per_kernel(write_csv, "WRITE_SIZE", pat)
    print(f"{'kernel':36s} {'launches':>8s} {'read MB':>9s} {'written MB':>10s} {'traffic MB per launch':>22s}")
    for k in sorted(f):
        nf, nw = len(f[k]), max(len(w.get(k, {})), 1)
        rd = 2.0 * sum(f[k].values()) * 1024 / nf / 1e6
        wr = sum(w.get(k, {}).values()) * 1024 / nw / 1e6
        print(f"{k:36s} {nf:8d} {rd:9.1f} {wr:10.1f} {rd + wr:22.1f}")


if __name__ == "__main__":
    main()
