"""Where a workgroup of the wide sparse weight-gradient kernel spends its time: per-phase s_memtime sums (100 MHz ticks).

Needs a library built with -DSEG3D_WGRAD_STAMP on wgrad_split.hip (tools/probes/build_wgrad_stamp_lib.sh -> csrc/libW.so;
copy it over libseg3d_hip.so on the GPU box for this probe only).  Runs the wide layers of the headline scene through the C ABI.
phases: compaction (+ its barriers), gather issue, fragment reads + MFMAs, wait for rows + split + image store, epilogue.
python tools/probes/wgrad_stamps.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, batch as B, config, ops, scene, spconv  # noqa: E402


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
    L = levels
    layers = [(L[1].subm(), 96, 96, "L2 subm"), (L[2].subm(), 384, 192, "L3 subm"), (L[3].subm(), 384, 384, "L4 subm"),
              (L[3].subm(), 768, 384, "L4 subm"), (strided[2][0], 192, 384, "L3->4 strided"), (strided[2][1], 384, 192, "L4->3 inverse")]
    lib = _lib.load()
    fn = lib.seg3d_debug_wgrad_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = torch.zeros((32,), dtype=torch.int64, device=dev)
    assert fn(buf.data_ptr()) == 0
    names = ["compact", "issue", "mfma", "wait+split", "epilogue", "steps", "total", "prologue"]
    for nbr, cin, cout, name in layers:
        m_out = nbr.shape[1]
        m_in = int(nbr.max().item()) + 1
        x = torch.randn(m_in, cin, device=dev)
        dy = torch.randn(m_out, cout, device=dev)
        nb = _lib.query("seg3d_spconv_wgrad_workspace_bytes", m_out, cin, cout)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        chunks = ctypes.c_int32(0)

        def run():
            _lib.call("seg3d_spconv_wgrad_partials", ops._ptr(x), ops._ptr(dy), ops._ptr(nbr), m_out, m_in, cin, cout, ops._ptr(ws), nb,
                      ctypes.byref(chunks), ops._stream())
        run()
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        s = buf.cpu().double()
        print(f"{name:14s} {cin:4d}->{cout:4d}  kernel {e0.elapsed_time(e1) * 1e3:7.1f} us")
        for tag, off, cnt in (("other offsets", 0, 16), ("centre offset", 8, 17)):
            waves = max(float(s[cnt]), 1.0)
            v = s[off:off + 8] / waves
            tot = max(float(v[6]), 1.0)
            steps = max(float(v[5]), 1.0)
            print(f"   {tag}: waves {int(waves):6d}  life {tot / 100:8.1f} us  steps/wave {steps:6.1f}  us/step {tot / 100 / steps:5.2f}   " +
                  "  ".join(f"{names[i]} {float(v[i]) / tot * 100:4.1f}%" for i in (7, 0, 1, 2, 3, 4)))


if __name__ == "__main__":
    main()
