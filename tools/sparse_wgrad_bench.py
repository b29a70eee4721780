"""Per-layer timing of the sparse-conv weight gradients of one training step on the headline scene (through the C ABI):
python tools/sparse_wgrad_bench.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, batch as B, config, ops, scene, segformer, spconv  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(v) for v in ds.grid_size[::-1]], 1)
    levels, strided = [level], []
    for _ in range(3):
        coarse, fwd, inv = levels[-1].down()
        strided.append((fwd, inv))
        levels.append(coarse)
    # (table, rows of x, cin, cout, name): the 20 sparse convs of PointTransformer
    L = levels
    layers = [(L[0].subm(), 64, 48, "L1 subm"), (L[0].subm(), 48, 48, "L1 subm"), (L[0].subm(), 48, 48, "L1 subm"),
              (strided[0][0], 48, 96, "L1->2 strided"), (L[1].subm(), 96, 96, "L2 subm"), (L[1].subm(), 96, 96, "L2 subm"),
              (strided[1][0], 96, 192, "L2->3 strided"), (L[2].subm(), 192, 192, "L3 subm"), (L[2].subm(), 192, 192, "L3 subm"),
              (strided[2][0], 192, 384, "L3->4 strided"), (L[3].subm(), 384, 384, "L4 subm"), (L[3].subm(), 384, 384, "L4 subm"),
              (L[3].subm(), 768, 384, "L4 subm"), (strided[2][1], 384, 192, "L4->3 inverse"), (L[2].subm(), 384, 192, "L3 subm"),
              (strided[1][1], 192, 96, "L3->2 inverse"), (L[1].subm(), 192, 96, "L2 subm"), (strided[0][1], 96, 48, "L2->1 inverse"),
              (L[0].subm(), 96, 48, "L1 subm"), (L[0].subm(), 48, 32, "L1 subm")]
    total = 0.0
    for nbr, cin, cout, name in layers:
        m_out = nbr.shape[1]
        m_in = int(nbr.max().item()) + 1
        x = torch.randn(m_in, cin, device=dev)
        xb = "--xbf16" in sys.argv  # the opt-in bf16 copies of the training mode: x rows as bf16, partial blocks only
        if xb:
            x = x.to(torch.bfloat16)
        chunks = ctypes.c_int32(0)
        dy = torch.randn(m_out, cout, device=dev)
        dw = torch.empty(cout, 27, cin, device=dev)
        nb = _lib.query("seg3d_spconv_wgrad_workspace_bytes", m_out, cin, cout)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        pairs = int((nbr >= 0).sum().item())

        def run():
            if xb or "--partials" in sys.argv:
                _lib.call("seg3d_spconv_wgrad_partials_xbf16" if xb else "seg3d_spconv_wgrad_partials", ops._ptr(x), ops._ptr(dy),
                          ops._ptr(nbr), m_out, m_in, cin, cout, ops._ptr(ws), nb, ctypes.byref(chunks), ops._stream())
                return
            _lib.call("seg3d_spconv_wgrad", ops._ptr(x), ops._ptr(dy), ops._ptr(nbr), m_out, m_in, cin, cout, 4, ops._ptr(dw),
                      ops._ptr(ws), nb, ops._stream())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        total += us
        frac = 2.0 * pairs * cin * cout / us / 1e6 / (2500.0 / 3.0)
        print(f"{name:14s} rows {m_out:7d} {cin:4d} -> {cout:4d}: {us:7.1f} us  {frac:.3f} of the split-bf16 ceiling  ws {nb / 1e6:.0f} MB", flush=True)
    print(f"sum {total / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
