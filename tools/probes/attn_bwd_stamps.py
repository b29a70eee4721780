"""Where the workgroups of the two attention-backward passes spend their time: s_memtime (100 MHz) spans per wave --
prologue (item record -> window geometry -> token indices -> stationary rows -> LDS -> first barrier), main loop, epilogue.

Needs a library built with -DSEG3D_ATTN_STAMP on attention_fused_bwd.hip (tools/probes/build_attn_bwd_stamp_lib.sh puts it in
csrc/libS.so; on the GPU box: cp libS.so libseg3d_hip.so for this probe only).
python tools/probes/attn_bwd_stamps.py [--drop 0.1]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openseg3d_amd import _lib, batch as B, config, ops, scene, spconv, swformer  # noqa: E402


def main():
    drop = float(sys.argv[sys.argv.index("--drop") + 1]) if "--drop" in sys.argv else 0.1
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    info = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    lib = _lib.load()
    fn = lib.seg3d_debug_attn_bwd_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = torch.zeros((1 << 18) * 4 * 8, dtype=torch.int64, device=dev)
    assert fn(buf.data_ptr()) == 0
    for stage, c in enumerate((48, 96, 192, 384)):
        if stage > 0:
            part = swformer.SparseWindowPartitionLayer(info[stage], cfg.MODEL.WINDOW_SHAPE, [float(g) / 2 ** stage for g in ds.grid_size])
            plan = part.plan(level.coords, 1, c)
            m = level.coords.shape[0]
            tau = torch.ones((1, 1, 1), device=dev)
            wi = plan.index[0]
            qk = torch.randn(m, 2 * c, device=dev, requires_grad=True)
            v = torch.randn(m, c, device=dev, requires_grad=True)
            g = torch.randn(m, c, device=dev)
            for _ in range(2):
                buf.zero_()
                ops.window_attention_packed(qk, v, tau, 0.01, 8, wi, drop, 1234).backward(g)
                qk.grad = v.grad = None
                torch.cuda.synchronize()
            s = buf.view(-1, 4, 8).double()
            # grid of both passes (attention_fused_bwd.hip: bwd_blocks): whole groups of 8 items x head groups; the tile loop's
            # phase sums sit in a second record block `grid` workgroups further
            n_items, hgn = (wi.n_tiles, 2) if c // 8 <= 12 else (wi.n_qgroups, 8)
            grid = (n_items + 7) // 8 * 8 * hgn
            for mode, name in ((0, "pass Q "), (1, "pass KV")):
                t = s[:grid, :, 4 * mode:4 * mode + 4]
                ran = t[:, :, 3].sum(dim=1) > 0
                ph = s[grid:2 * grid, :, 4 * mode:4 * mode + 4][ran]  # compute, wait for rows, convert + store, barrier
                tot = ph.sum(dim=(0, 1))
                print(f"stage {stage + 1} C={c} {name}: tile loop = compute {float(tot[0] / tot.sum()) * 100:4.1f}%  wait for rows {float(tot[1] / tot.sum()) * 100:4.1f}%  "
                      f"convert + LDS store {float(tot[2] / tot.sum()) * 100:4.1f}%  barrier {float(tot[3] / tot.sum()) * 100:4.1f}%")
                t = t[ran]  # workgroups that ran
                life = t[:, :, :3].sum(dim=2).max(dim=1).values  # longest wave of the workgroup, in 10 ns ticks
                pro, loop, epi = (t[:, :, i].mean(dim=1) for i in range(3))
                steps = t[:, 0, 3]
                print(f"stage {stage + 1} C={c} {name}: wgs {t.shape[0]:6d}  tiles/wg {float(steps.mean()):5.1f}  wg life {float(life.mean()) / 100:6.2f} us  "
                      f"prologue {float(pro.mean()) / 100:5.2f} us ({float(pro.sum() / life.sum()) * 100:4.1f}%)  loop {float(loop.mean()) / 100:6.2f} us "
                      f"({float(loop.sum() / steps.sum()) / 100:5.2f} us per tile)  epilogue {float(epi.mean()) / 100:5.2f} us ({float(epi.sum() / life.sum()) * 100:4.1f}%)", flush=True)
        if stage < 3:
            level = level.down()[0]


if __name__ == "__main__":
    main()
