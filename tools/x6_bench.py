"""Per-point MLP layers of the headline scene (174 633 points): the six-product split (csrc/linear_x6.hip) against the
exact-fp32 MFMA kernel it replaces and the three-product split, HIP events, median of 20.
    python tools/x6_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openseg3d_amd import _lib, ops  # noqa: E402

SHAPES = [(64, 128), (128, 256), (256, 64), (96, 256), (256, 128), (128, 64), (64, 64)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 174633
    dev = torch.device("cuda:0")
    tot = [0.0, 0.0, 0.0]
    for cin, cout in SHAPES:
        x = torch.randn(m, cin, device=dev)
        w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        p6 = torch.empty((_lib.query("seg3d_linear_packed_bytes_x6", cin, cout),), dtype=torch.uint8, device=dev)
        _lib.call("seg3d_linear_pack_weight_x6", ops._ptr(w), cin, cout, 0, ops._ptr(p6), ops._stream())
        p32 = ops._linear_pack_f32(w, 0)
        p3 = ops._linear_pack(w, False)
        t6 = timed(lambda: ops._linear_apply_x6(x, p6, None, cin, cout))
        t32 = timed(lambda: ops._linear_apply_f32(x, p32, None, cin, cout))
        t3 = timed(lambda: ops._linear_apply(x, p3, None, cin, cout))
        ref = x.double() @ w.double().t()
        mag = x.double().abs() @ w.double().abs().t()
        errs = [float(((f.double() - ref).abs() / mag).max()) for f in
                (ops._linear_apply_x6(x, p6, None, cin, cout), ops._linear_apply_f32(x, p32, None, cin, cout),
                 ops._linear_apply(x, p3, None, cin, cout))]
        gb = m * (cin + cout) * 4 / 1e9
        print(f"{cin:4d} -> {cout:4d} @{m}: x6 {t6:7.1f} us ({gb / t6 * 1e6 / 1e3:5.2f} TB/s)  fp32 {t32:7.1f} us  x3 {t3:7.1f} us   "
              f"err/mag x6 {errs[0]:.2e} fp32 {errs[1]:.2e} x3 {errs[2]:.2e}", flush=True)
        tot[0] += t6; tot[1] += t32; tot[2] += t3
    print(f"sum of the seven layers: x6 {tot[0]:.0f} us, fp32 {tot[1]:.0f} us, x3 {tot[2]:.0f} us")


if __name__ == "__main__":
    main()
