"""Time seg3d_linear_wgrad on the Linear shapes of the waymo_one_sweep workload (GPU box only)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from openseg3d_amd import _lib, ops

# (rows, cin, cout, calls per training step): per encoder layer qk C->2C, v C->C, out C->C, mlp C->2C->C
SHAPES = []
for m, c, depth in ((108690, 48, 3), (121168, 96, 4), (58453, 192, 8), (19483, 384, 3)):
    SHAPES += [(m, c, 2 * c, 2 * depth), (m, c, c, 2 * depth), (m, 2 * c, c, depth)]
SHAPES += [(174633, 64, 128, 1), (174633, 128, 256, 1), (174633, 256, 64, 1), (174633, 96, 256, 1), (174633, 256, 128, 1)]
dev = torch.device("cuda:0")
total = 0.0
for m, cin, cout, calls in SHAPES:
    x = torch.randn(m, cin, device=dev)
    dy = torch.randn(m, cout, device=dev)
    dw = torch.empty(cout, cin, device=dev)
    db = torch.empty(cout, device=dev)
    nb = _lib.query("seg3d_linear_wgrad_workspace_bytes", m, cin, cout)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    xb = "--xbf16" in sys.argv
    xq = x.to(torch.bfloat16) if xb else x
    chunks = ctypes.c_int32(0)
    def run():
        if xb or "--partials" in sys.argv:
            _lib.call("seg3d_linear_wgrad_partials_xbf16" if xb else "seg3d_linear_wgrad_partials", ops._ptr(xq), ops._ptr(dy), m, cin, cout,
                      1, ops._ptr(ws), nb, ctypes.byref(chunks), ops._stream())
            return
        _lib.call("seg3d_linear_wgrad", ops._ptr(x), ops._ptr(dy), m, cin, cout, ops._ptr(dw), ops._ptr(db), ops._ptr(ws), nb,
                  ops._stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = m * (cin + cout) * 4 / 1e9
    total += us * calls
    print(f"m={m:7d} {cin:4d}->{cout:4d}: {us:7.1f} us x{calls:2d}  {gb / us * 1e6:7.0f} GB/s algorithmic  ws {nb / 1e6:.1f} MB", flush=True)
print(f"sum over one training step: {total / 1e3:.2f} ms")
