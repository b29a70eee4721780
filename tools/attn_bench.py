"""Stand-alone timing of the window-attention core per stage of the headline scene (forward, optionally backward):
python tools/attn_bench.py [--bwd] [--iters 20] [--workload one_sweep|dense2m] [--drop 0.1]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import batch as B, config, ops, scene, spconv, swformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--workload", default="one_sweep")
    ap.add_argument("--drop", type=float, default=0.0)
    ap.add_argument("--stages", default="0,1,2,3")
    ap.add_argument("--tau", type=float, default=1.0, help="value of the learnable temperature (tau_min = 0.01 clamps below)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    if args.workload == "dense2m":
        cfg.DATASET.POINT_CLOUD_RANGE, cfg.DATASET.VOXEL_SIZE = list(scene.DENSE_RANGE), list(scene.DENSE_VOXEL)
        pts = scene.make_dense_scene(0)
    else:
        pts = scene.make_scene(0)
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    info = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    tot_ms, tot_fl = 0.0, 0.0
    depth_split = {0: (1, 2), 1: (2, 2), 2: (4, 4), 3: (1, 2)}  # layers on shift 0 / shift 1 (depths 3, 4, 8, 3)
    for stage, c in enumerate((48, 96, 192, 384)):
        if str(stage) in args.stages.split(","):
            part = swformer.SparseWindowPartitionLayer(info[stage], cfg.MODEL.WINDOW_SHAPE, [float(g) / 2 ** stage for g in ds.grid_size])
            plan = part.plan(level.coords, 1, c)
            m = level.coords.shape[0]
            tau = torch.full((1, 1, 1), args.tau, device=dev)
            for shift in (0, 1):
                wi = plan.index[shift]
                qk = torch.randn(m, 2 * c, device=dev, requires_grad=args.bwd)
                v = torch.randn(m, c, device=dev, requires_grad=args.bwd)
                g = torch.randn(m, c, device=dev)
                cnt = wi.win_count[: wi.n_windows].double()
                flop = 4.0 * c * float((cnt * cnt).sum())

                def run():
                    o = ops.window_attention_packed(qk, v, tau, 0.01, 8, wi, args.drop, 1234)
                    if args.bwd:
                        o.backward(g)
                        qk.grad = v.grad = None
                for _ in range(3):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / args.iters * 1e3
                layers = depth_split[stage][shift]
                tot_ms += us * layers / 1e3
                tot_fl += flop * layers
                print(f"stage {stage + 1} C={c} shift {shift}: m={m} windows={wi.n_windows} tiles={wi.n_tiles} chunks={wi.n_qgroups} "
                      f"{us:8.1f} us  {flop * (3.5 if args.bwd else 1.0) / us / 1e6:7.1f} TF/s (algorithmic)", flush=True)
        if stage < 3:
            level = level.down()[0]
    print(f"forward{'+backward' if args.bwd else ''} of the 18 layers: {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TF/s fwd-algorithmic")


if __name__ == "__main__":
    main()
