"""Per-kernel totals inside the training steps of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

usage: python tools/trace_summary.py <kernel_trace.csv> [--steps N] [--csv out.csv] [--top 60]
Training steps are delimited by the optimizer's launches (fused SGD): the last N of them bound N steps."""
import argparse
import collections
import csv
import re


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def family(n):
    if "spconv_split_kernel<" in n and ", true," in n:
        return "Linear forward / dX (dense rows through the gather-GEMM)"
    for key, fam in (("spconv_tile_kernel<", "sparse conv gather-GEMM (fwd + dgrad)"), ("spconv_split_kernel<", "sparse conv gather-GEMM (fwd + dgrad)"),
                     ("plan_kernel", "index build + rest"), ("morton_keys", "index build + rest"), ("spconv_fwd_kernel", "exact-fp32 point MLP"),
                     ("wgrad", "weight gradients"), ("attn", "window attention"), ("tau_reduce", "window attention"),
                     ("ln_", "LayerNorm / BatchNorm passes"), ("col_", "LayerNorm / BatchNorm passes"), ("bn_", "LayerNorm / BatchNorm passes"),
                     ("affine_act", "LayerNorm / BatchNorm passes"), ("lovasz", "criterion + kNN"), ("ce_", "criterion + kNN"),
                     ("knn", "criterion + kNN"), ("radix", "criterion + kNN / sorts"), ("pack_weight", "packing"),
                     ("at::native", "torch elementwise / other"), ("multi_tensor", "optimizer")):
        if key in n:
            return fam
    return "index build + rest"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    # the optimizer's own launches only: `_foreach_add_` on the BatchNorm step counters is a multi_tensor_apply kernel too
    opt = [i for i, r in enumerate(rows) if "FusedSgd" in r[0] or "fused_sgd" in r[0].lower()]
    groups = []
    for i in opt:
        if groups and rows[i][1] - rows[groups[-1][-1]][2] < 1_000_000:
            groups[-1].append(i)
        else:
            groups.append([i])
    assert len(groups) > a.steps, f"only {len(groups)} optimizer steps in the trace"
    first = groups[-a.steps - 1][-1] + 1
    last = groups[-1][-1]
    win = rows[first:last + 1]
    span = (win[-1][2] - win[0][1]) / 1e6 / a.steps
    tot = collections.defaultdict(lambda: [0, 0])
    fam = collections.defaultdict(float)
    for n, s, e in win:
        k = short(n)
        tot[k][0] += 1
        tot[k][1] += e - s
        fam[family(k)] += e - s
    busy = sum(v[1] for v in tot.values()) / 1e6 / a.steps
    print(f"{a.steps} training steps: {span:.2f} ms per step wall (first launch to last end), {busy:.2f} ms kernel-busy, "
          f"{len(win) / a.steps:.0f} launches per step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"  {v / 1e6 / a.steps:7.2f} ms {100 * v / 1e6 / a.steps / busy:5.1f} %  {k}")
    out = sorted(tot.items(), key=lambda kv: -kv[1][1])
    print("launches/step   ms/step   us/launch  kernel")
    for k, (c, ns) in out[: a.top]:
        print(f"{c / a.steps:10.1f} {ns / 1e6 / a.steps:10.3f} {ns / 1e3 / c:10.1f}  {k}")
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("kernel,launches_per_step,ms_per_step,us_per_launch,share\n")
            for k, (c, ns) in out:
                f.write(f"\"{k}\",{c / a.steps:.2f},{ns / 1e6 / a.steps:.4f},{ns / 1e3 / c:.2f},{ns / 1e6 / a.steps / busy:.4f}\n")


if __name__ == "__main__":
    main()
