"""Timing of the dense Linear kernels through the C ABI (no autograd / allocator in the timed loop, so the GPU is the limit)
at the shapes of the headline scene's stages: forward (= input-gradient kernel) and weight gradient (+ its chunk reduce).
python tools/linear_bench.py   -> us per launch and the share of the HBM roofline (rows * (cin + cout) * 4 bytes / 8 TB/s)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, ops  # noqa: E402

SHAPES = [(108690, 48, 48), (108690, 48, 96), (108690, 96, 48), (121168, 96, 96), (121168, 96, 192), (121168, 192, 96),
          (58453, 192, 192), (58453, 192, 384), (58453, 384, 192), (19483, 384, 384), (19483, 384, 768), (19483, 768, 384)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    p, st = ops._ptr, ops._stream
    tot_f = tot_w = 0.0
    for m, cin, cout in SHAPES:
        x = torch.randn(m, cin, device=dev)
        w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        b = torch.zeros(cout, device=dev)
        dy = torch.randn(m, cout, device=dev)
        y = torch.empty(m, cout, device=dev)
        packed = ops._linear_pack(w, False)
        t_f = timeit(lambda: _lib.call("seg3d_linear_fwd", p(x), m, p(packed), p(b), p(None), cin, cout, p(y), st()))
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ws_bytes = _lib.query("seg3d_linear_wgrad_workspace_bytes", m, cin, cout)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        t_w = timeit(lambda: _lib.call("seg3d_linear_wgrad", p(x), p(dy), m, cin, cout, p(dw), p(db), p(ws), ws_bytes, st()))
        ideal = m * (cin + cout) * 4 / 8e12 * 1e6
        tot_f += t_f
        tot_w += t_w
        print(f"rows {m:7d} {cin:4d} -> {cout:4d}: fwd {t_f:7.1f} us ({ideal / t_f:5.2f} of HBM roofline)   "
              f"wgrad + reduce {t_w:7.1f} us ({ideal / t_w:5.2f})")
    print(f"sum fwd {tot_f:.1f} us, wgrad {tot_w:.1f} us")


if __name__ == "__main__":
    main()
