"""Where a deep sparse-conv workgroup spends its cycles: per-phase s_memtime sums of spconv_split_kernel.

Needs a library built with -DSEG3D_CONV_STAMP on spconv_split.hip (tools/probes/build_stamp_lib.sh puts it in
csrc/libS.so; copy it over libseg3d_hip.so on the GPU box for this probe only).  One eval forward of the headline scene;
after each sparse conv the stamp buffer ([workgroup][wave][8] cycle sums) is read back and averaged.
phases: 0 issue next chunk's loads (table entries from LDS, W + row gathers)  1 MFMA block (LDS W reads + MFMA issue)
        2 wait for the row / W loads  3 split rows to bf16 hi/lo  4 W -> LDS + drain  5 barrier  7 prologue  6 whole wave
python tools/probes/conv_stamps.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, batch as B, config, ops, scene, segformer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).eval()
    b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
    with torch.no_grad():
        model(dict(b))
    lib = _lib.load()
    fn = lib.seg3d_debug_conv_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = torch.zeros((4096 * 4 * 8,), dtype=torch.int64, device=dev)
    assert fn(buf.data_ptr()) == 0
    orig = ops.conv_act

    def hooked(x, nbr, packed, bias, cin, cout, order=None, addend=None, relu=True):
        buf.zero_()
        y = orig(x, nbr, packed, bias, cin, cout, order, addend, relu)
        torch.cuda.synchronize()
        s = buf.view(-1, 4, 8).double()
        used = s[:, :, 6].sum(dim=1) > 0
        s = s[used]
        wg = s.shape[0]
        mean = s.mean(dim=(0, 1))
        tot = float(mean[6])
        names = ["issue", "mfma", "ldwait", "split", "commit", "barrier", "total", "prologue"]
        print(f"rows {x.shape[0]:7d} {cin:4d}->{cout:4d} wgs {wg:4d} total {tot:9.0f} cyc  " +
              "  ".join(f"{names[i]} {float(mean[i]) / tot * 100:4.1f}%" for i in (7, 0, 1, 2, 3, 4, 5)) +
              f"  | slowest wave {float(s[:, :, 6].max()):9.0f}")
        return y

    orig_lin = ops._linear_apply

    def hooked_lin(x, packed, bias, cin, cout, addend=None):
        buf.zero_()
        y = orig_lin(x, packed, bias, cin, cout, addend)
        torch.cuda.synchronize()
        s = buf.view(-1, 4, 8).double()
        s = s[s[:, :, 6].sum(dim=1) > 0]
        mean = s.mean(dim=(0, 1))
        tot = float(mean[6])
        names = ["issue", "mfma", "ldwait", "split", "commit", "barrier", "total", "prologue"]
        loop = sum(float(mean[i]) for i in (0, 1, 2, 3, 4, 5))
        print(f"LINEAR rows {x.shape[0]:7d} {cin:4d}->{cout:4d} wgs {s.shape[0]:4d} total {tot:9.0f} cyc  " +
              "  ".join(f"{names[i]} {float(mean[i]) / tot * 100:4.1f}%" for i in (7, 0, 1, 2, 3, 4, 5)) +
              f"  epilogue+rest {(tot - loop - float(mean[7])) / tot * 100:4.1f}%")
        return y

    ops.conv_act = hooked
    if "--linear" in sys.argv:
        ops._linear_apply = hooked_lin
    try:
        with torch.no_grad():
            model(dict(b))
    finally:
        ops.conv_act = orig
        ops._linear_apply = orig_lin


if __name__ == "__main__":
    main()
