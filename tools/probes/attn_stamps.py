"""Where a window-attention workgroup spends its cycles: per-phase s_memtime sums of attn_fused_fwd.

Needs a library built with -DSEG3D_ATTN_STAMP on attention_fused.hip (tools/probes/build_attn_stamp_lib.sh puts it in
csrc/libS.so; on the GPU box: cp libS.so libseg3d_hip.so for this probe only).
phases (persistent kernel, sums over the items a workgroup walks): 0 descriptors of the workgroup (once)  1 first item's
        token indices + row requests  3 rows: wait + convert + LDS store  4 barriers  5 tile compute (LDS reads, MFMAs,
        softmax) incl. the next item's index prefetch  6 next item's row requests + epilogue  7 whole wave; slot 2 = items
python tools/probes/attn_stamps.py [--workload dense2m]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, batch as B, config, ops, scene, spconv, swformer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    info = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    lib = _lib.load()
    fn = lib.seg3d_debug_attn_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = torch.zeros((1 << 20) * 4 * 8, dtype=torch.int64, device=dev)
    assert fn(buf.data_ptr()) == 0
    names = ["desc", "rowwait", "items", "convert", "barrier", "compute", "epilog", "total"]
    for stage, c in enumerate((48, 96, 192, 384)):
        part = swformer.SparseWindowPartitionLayer(info[stage], cfg.MODEL.WINDOW_SHAPE, [float(g) / 2 ** stage for g in ds.grid_size])
        plan = part.plan(level.coords, 1, c)
        m = level.coords.shape[0]
        tau = torch.ones((1, 1, 1), device=dev)
        wi = plan.index[0]
        qk = torch.randn(m, 2 * c, device=dev)
        v = torch.randn(m, c, device=dev)
        for _ in range(2):
            buf.zero_()
            ops.window_attention_packed(qk, v, tau, 0.01, 8, wi, 0.0, 0)
            torch.cuda.synchronize()
        s = buf.view(-1, 4, 8).double()
        s = s[s[:, :, 7].sum(dim=1) > 0]
        mean = s.mean(dim=(0, 1))
        tot = float(mean[7])
        wg_life = s[:, :, 7].max(dim=1).values
        print(f"stage {stage + 1} C={c} wgs {s.shape[0]:6d} items/wg {float(mean[2]):6.1f} mean wave {tot:8.0f} cyc  " +
              "  ".join(f"{names[i]} {float(mean[i]) / tot * 100:4.1f}%" for i in (0, 1, 3, 4, 5, 6)) +
              f"  | wg life mean {float(wg_life.mean()):8.0f} min {float(wg_life.min()):8.0f} max {float(wg_life.max()):8.0f}"
              f"  = {float(wg_life.max()) / 2.4e3:6.1f} us@2.4GHz", flush=True)
        if stage < 3:
            level = level.down()[0]


if __name__ == "__main__":
    main()
