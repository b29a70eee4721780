"""Build profiles/rNN_pmc_conv_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline` and the per-layer list bench.py writes.

usage: python tools/pmc_conv_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <bench_layers.json> <out.json>
Correction (MI355X_MICROARCH.md, HBM section): counter unit KiB; FETCH_SIZE counts 128-B read requests as 64 B on
gfx950 -> doubled; WRITE_SIZE taken as is.
"""
import csv
import re
import json
import sys


def conv_dispatches(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        # sparse instantiations: spconv_split_kernel<NBT, RB, DENSE = false, IO> and every spconv_tile_kernel
        if not re.search(r"spconv_split_kernel<\d+, \d+, false|spconv_tile_kernel<", n) or r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        per[d] = (n, per.get(d, (n, 0.0))[1] + float(r["Counter_Value"]))
    return [per[d] for d in sorted(per)]


def main():
    fetch_csv, write_csv, layers_json, out = sys.argv[1:5]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    layers = json.load(open(layers_json))
    n = len(layers)
    f = conv_dispatches(fetch_csv, "FETCH_SIZE")[-n:]
    w = conv_dispatches(write_csv, "WRITE_SIZE")[-n:]
    assert len(f) == n and len(w) == n, (len(f), len(w), n)
    rows, tot_t, tot_a = [], 0.0, 0.0
    for i, (lay, (kn, fk), (_, wk)) in enumerate(zip(layers, f, w)):
        fb, wb = fk * 1024.0, wk * 1024.0
        traffic = 2.0 * fb + wb
        algo = lay["pairs"] * (lay["cin"] + lay["cout"]) * 4 + 27 * lay["cin"] * lay["cout"] * 4 + lay["pairs"] * 8
        rows.append({"layer": i, "rows": lay["rows"], "pairs": lay["pairs"], "cin": lay["cin"], "cout": lay["cout"],
                     "kernel": kn[:60], "FETCH_SIZE_bytes": int(fb), "WRITE_SIZE_bytes": int(wb),
                     "traffic_bytes_corrected": int(traffic), "algorithmic_bytes": int(algo)})
        tot_t += traffic
        tot_a += algo
    doc = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two separate passes) --output-format csv -- python3 "
                      "bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline; tools/pmc_conv_traffic.py",
           "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> doubled (MI355X_MICROARCH.md section "
                         "HBM); WRITE_SIZE taken as is; counter unit KiB",
           "kernel": f"spconv_tile_kernel<*> / spconv_split_kernel<*, 2, false> ({n} sparse-conv launches of the scene-0 forward)",
           "lib_sha16": bench.lib_sha16(),
           "traffic_bytes_per_launch": int(tot_t / n), "algorithmic_bytes_per_launch": int(tot_a / n), "layers": rows}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"traffic {tot_t / n / 1e6:.1f} MB per launch, algorithmic {tot_a / n / 1e6:.1f} MB per launch")


if __name__ == "__main__":
    main()
