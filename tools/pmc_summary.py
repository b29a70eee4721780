"""Sum rocprofv3 --pmc counter_collection.csv per kernel name pattern: python tools/pmc_summary.py file.csv pattern"""
import csv
import collections
import re
import sys

rows = csv.DictReader(open(sys.argv[1]))
pat = re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    name = r["Kernel_Name"]
    if not pat.search(name):
        continue
    m = re.search(r"(\w+)<(\d+)[,>]", name)
    key = m.group(0) if m else name[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    did = (key, r["Dispatch_Id"])
    if did not in seen:
        seen.add(did)
        cnt[key] += 1
for k in sorted(agg):
    print(k, "dispatches", cnt[k])
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[k]:16.1f} per dispatch")
