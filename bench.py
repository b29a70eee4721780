#!/usr/bin/env python3
"""Benchmark of the sparse-voxel segmentation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode fwd|fwdbwd] [--scenes S]

One "step" = one pass of the hot path over one batch of synthetic input per GPU: GPU voxelisation of
a resident Waymo-shaped scene (~180 k float32 points, 0.1 m voxels, grid 1440x1440x64, i.e.
configs/waymo_one_sweep.yaml = BASELINE.json configs[1]) followed by Segformer forward + loss +
backward + SGD step (default mode fwdbwd = the metric "points/sec fwd+bwd"; gradients all-reduced
over RCCL when N > 1).  The forward-only eval rate of the same scenes is timed in the same run and
reported as `fwd_only` (mode fwd makes it the headline instead).  Scenes shard data-parallel: every rank processes its own scenes,
`value` = points processed by all ranks / max-over-ranks wall time ("scaling": "weak").

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the dominant kernel (sparse-conv gather-GEMM): algorithmic bytes / live HIP-event time
  cpu_baseline -- the CPU oracle (port of the reference algorithm) timed on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (6290 GB/s measured copy), MI355X_MICROARCH.md:36
ATTENTION_REPORT = None  # filled by conv_roofline's instrumented forward
# single-rank rehearsal of the multi-GPU path (RCCL process group, DDP wrapper, barriers) on a one-GPU box:
# python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 with SEG3D_BENCH_DIST=1
DIST_REHEARSAL = os.environ.get("SEG3D_BENCH_DIST", "0") == "1" and "RANK" in os.environ


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["fwd", "fwdbwd"], default="fwdbwd")
    ap.add_argument("--scenes", type=int, default=4, help="distinct resident scenes per rank")
    ap.add_argument("--batch", type=int, default=1, help="scenes per step per GPU")
    ap.add_argument("--workload", choices=["one_sweep", "cylinder", "multi_sweeps"], default="one_sweep",
                    help="one_sweep = BASELINE configs[1] (the headline); cylinder = configs[2] geometry "
                         "(waymo_one_sweep_cylinder.yaml, use --batch 4); multi_sweeps = configs[3] "
                         "(waymo_multi_sweeps.yaml + image features, 3 sweeps, use --batch 2)")
    ap.add_argument("--segmentor", choices=["segformer", "spnet"], default="segformer",
                    help="MODEL.SEGMENTOR (builder.py:8-23); segformer is the headline, spnet = SparseUnet + OCR")
    ap.add_argument("--criterion", choices=["default", "ce"], default="default",
                    help="default = MODEL.LOSSES of the reference config (ohem_ce + lovasz on the point, voxel and "
                         "auxiliary heads, tools/train.py:71-110); ce = plain cross-entropy on the three heads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU-baseline sample (0 = whole scene)")
    return ap.parse_args()


def setup_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or DIST_REHEARSAL:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")  # RCCL over xGMI
    torch.cuda.set_device(local)
    return rank, world, local


def barrier(world):
    if world > 1 or DIST_REHEARSAL:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def conv_roofline(model, batch, dev):
    """One instrumented forward: HIP events around every seg3d_spconv_fwd launch (same stream)."""
    from openseg3d_amd import ops
    records = []
    orig = ops._conv_apply

    def timed(x, nbr, w_packed, bias, cin, cout, order=None):  # noqa: E306
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, nbr, w_packed, bias, cin, cout, order)
        e1.record()
        records.append((nbr, cin, cout, e0, e1))
        return y

    orig_act = ops.conv_act  # inference form of the conv blocks (BatchNorm folded, ReLU / residual in the epilogue)

    def timed_act(x, nbr, packed, bias, cin, cout, order=None, addend=None, relu=True):  # noqa: E306
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_act(x, nbr, packed, bias, cin, cout, order, addend, relu)
        e1.record()
        records.append((nbr, cin, cout, e0, e1))
        return y

    attn_records = []
    orig_attn = ops.window_attention_packed

    def timed_attn(qk, v, tau, tau_min, heads, wi):  # noqa: E306
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = orig_attn(qk, v, tau, tau_min, heads, wi)
        e1.record()
        attn_records.append((wi, v.shape[1], e0, e1))
        return o

    ops._conv_apply = timed
    ops.conv_act = timed_act
    ops.window_attention_packed = timed_attn
    passes = 3  # the figure is an average over 3 instrumented forwards (60 launches): one pass alone moves by +-5 %
    try:
        with torch.no_grad():
            for _ in range(passes):
                model(dict(batch))
        torch.cuda.synchronize()
    finally:
        ops._conv_apply = orig
        ops.conv_act = orig_act
        ops.window_attention_packed = orig_attn
    # window attention (prepare + core kernels of one layer): algorithmic FLOPs 4*C*sum_w n_w^2
    sq_cache, a_flop, a_ms = {}, 0.0, 0.0
    for wi, c, e0, e1 in attn_records:
        if id(wi) not in sq_cache:
            cnt = wi.win_count[: wi.n_windows].double()
            sq_cache[id(wi)] = float((cnt * cnt).sum().item())
        a_flop += 4.0 * c * sq_cache[id(wi)] / passes
        a_ms += e0.elapsed_time(e1) / passes
    attn_tf = a_flop / a_ms / 1e9 if a_ms > 0 else 0.0
    global ATTENTION_REPORT
    ATTENTION_REPORT = {
        "bound": "mfma", "kernel": "window attention of one forward, 18 encoder layers: attn_prepare_fwd + attn_core_fwd "
                                   "(dh 24/48, split-bf16 MFMA) and attn_small_fwd (dh 6/12, exact-fp32 vector ALU)",
        "achieved": round(attn_tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(attn_tf / 2500.0, 5),
        "note": "algorithmic 4*C*sum(n_w^2) FLOPs per layer / HIP-event time; the MFMA kernels execute 3 bf16 MFMAs "
                "per product (split-bf16) on 32x32-token tiles, so executed MFMA FLOPs are >= 3x the algorithmic; "
                "the two narrow-head stages do not use the matrix cores at all",
        "layers": len(attn_records) // passes, "gflop_per_forward": round(a_flop / 1e9, 2), "ms_per_forward": round(a_ms, 3)}
    pairs_cache, tot_bytes, tot_ms = {}, 0.0, 0.0
    per_layer = []
    n_layers = len(records) // passes
    for li in range(n_layers):
        nbr, cin, cout = records[li][:3]
        key = nbr.data_ptr()
        if key not in pairs_cache:
            pairs_cache[key] = int((nbr >= 0).sum().item())
        p = pairs_cache[key]
        algo = p * (cin + cout) * 4 + 27 * cin * cout * 4 + p * 8
        ms = sum(records[li + k * n_layers][3].elapsed_time(records[li + k * n_layers][4]) for k in range(passes)) / passes
        tot_bytes += algo
        tot_ms += ms
        per_layer.append({"rows": int(nbr.shape[1]), "pairs": p, "cin": cin, "cout": cout, "us": round(ms * 1e3, 1),
                          "GBs": round(algo / ms / 1e6, 1)})
    n = max(n_layers, 1)
    achieved = tot_bytes / tot_ms / 1e6 if tot_ms > 0 else 0.0
    # HBM traffic per launch from the committed PMC passes (FETCH_SIZE doubled per the gfx950 correction +
    # WRITE_SIZE, two separate rocprofv3 --pmc runs of this same workload); only quoted when the profiled
    # workload is the one just run (same algorithmic bytes), otherwise null.
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            p = json.load(f)
        if abs(p.get("algorithmic_bytes_per_launch", 0) - tot_bytes / n) <= 0.01 * tot_bytes / n:
            traffic = p["traffic_bytes_per_launch"]
    return {"bound": "hbm", "kernel": f"spconv_split_kernel (all {n_layers} sparse-conv launches of one forward, mean of {passes} forwards)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "launches": n_layers,
            "bytes_per_launch": int(tot_bytes / n), "us_per_launch": round(tot_ms * 1e3 / n, 2)}, per_layer


def cpu_baseline(scene_np, cfg, ds, model, n_points):
    """The oracle (CPU restatement of the reference algorithm) on the host cores: same scene (or its first
    n_points rows), same weights, eval forward."""
    from oracle import index_ops, model as omodel
    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe a larger affinity mask
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle forward on {cores} host threads ...", file=sys.stderr, flush=True)
    pts = scene_np if not n_points else scene_np[:n_points]
    t0 = time.time()
    coords, ids = index_ops.voxelize(pts, ds.voxel_size, ds.point_cloud_range)
    batch = {"points": torch.from_numpy(np.pad(pts, ((0, 0), (1, 0)))).float(),
             "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float(),
             "point_voxel_ids": torch.from_numpy(ids).long(), "batch_size": 1}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range),
            "batching_info": [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO],
            "window_shape": cfg.MODEL.WINDOW_SHAPE, "depths": cfg.MODEL.DEPTHS}
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        (omodel.spnet_forward if cfg.MODEL.SEGMENTOR == "spnet" else omodel.segformer_forward)(batch, sd, ocfg)
    dt = time.time() - t0
    return {"value": round(pts.shape[0] / dt, 1), "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"1 forward (voxelize + {cfg.MODEL.SEGMENTOR} eval) of {pts.shape[0]} points of scene seed 0, "
                      f"{coords.shape[0]} voxels, {dt:.1f} s, torch.set_num_threads({cores})"}


def main():
    args = parse()
    rank, world, local = setup_dist(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    from openseg3d_amd import batch as B, config, dist as D, scene, segformer
    dev = torch.device("cuda", local)
    cfg = config.default_cfg()  # == configs/waymo_one_sweep.yaml for every key the model path reads
    if args.workload == "cylinder":  # configs/waymo_one_sweep_cylinder.yaml:2-4
        cfg.DATASET.USE_CYLINDER = True
        cfg.DATASET.POINT_CLOUD_RANGE = [0, -3.1415926, -2, 75.2, 3.1415926, 5.2]
        cfg.DATASET.VOXEL_SIZE = [0.05, 0.012, 0.1]
    elif args.workload == "multi_sweeps":  # configs/waymo_multi_sweeps.yaml:1-4 + USE_IMAGE_FEATURE
        cfg.DATASET.USE_MULTI_SWEEPS = True
        cfg.DATASET.USE_IMAGE_FEATURE = True
    cfg.MODEL.SEGMENTOR = args.segmentor
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev)

    # resident synthetic scenes: seeds differ per rank (data-parallel shards of the scene stream)
    seeds = [rank * args.scenes * args.batch + i for i in range(args.scenes * args.batch)]
    if args.workload == "multi_sweeps":
        made = [scene.make_multi_sweep_scene(s, cfg.DATASET.NUM_SWEEPS) for s in seeds]
        scenes_np, n_cur = [m[0] for m in made], [m[1] for m in made]
    else:
        scenes_np = [scene.make_scene(s) for s in seeds]
        if args.workload == "cylinder":
            scenes_np = [scene.cart2polar_rows(s) for s in scenes_np]
        n_cur = [s.shape[0] for s in scenes_np]
    groups = [list(range(i * args.batch, (i + 1) * args.batch)) for i in range(args.scenes)]
    resident = [B.collate_points([scenes_np[j] for j in g], dev) for g in groups]
    offsets = [np.cumsum([n_cur[j] for j in g]).tolist() for g in groups]  # cumulative current-sweep rows
    images = [None] * len(groups)
    if args.workload == "multi_sweeps":
        images = [torch.from_numpy(np.concatenate([scene.make_image_features(seeds[j], n_cur[j]) for j in g])).to(dev)
                  for g in groups]
    pts_per_step = [int(o[-1]) for o in offsets]  # points that receive logits

    train = args.mode == "fwdbwd"
    # configs/waymo_one_sweep.yaml: SGD, momentum 0.9, weight decay 1e-4; torch's fused multi-tensor implementation
    # (same arithmetic, 3 launches instead of 24) unless SEG3D_BENCH_FOREACH_SGD=1
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4,
                          fused=os.environ.get("SEG3D_BENCH_FOREACH_SGD", "0") != "1")
    net = model
    if train and (world > 1 or DIST_REHEARSAL):
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False,
                                                        broadcast_buffers=False, gradient_as_bucket_view=True)
    labels = [torch.randint(0, 22, (n,), device=dev) for n in pts_per_step]
    from openseg3d_amd import losses as _losses, ops as _ops
    if args.criterion == "ce":
        cfg.MODEL.LOSSES = {"ce": 1.0}
    criterion = _losses.build_criterion(cfg, ds)  # builder.py:26-40
    # voxel labels are part of the collated batch in the reference (WaymoDataset.prepare_voxel_labels in the loader
    # workers): made once per resident scene group here, outside the timed region; the auxiliary-head labels are looked
    # up inside the step, as tools/train.py:86-104 does
    voxel_labels = []
    for j in range(len(resident)):
        b = B.batch_from_resident(resident[j], offsets[j], ds.voxel_size, ds.point_cloud_range, images[j])
        cur = torch.nonzero(b["points"][:, 4] == 0).view(-1) if args.workload == "multi_sweeps" else None
        voxel_labels.append(_ops.prepare_voxel_labels(b["point_voxel_ids"], labels[j], b["voxel_coords"].shape[0],
                                                      ignore_index=ds.ignore_index, cur_point_indices=cur).long())

    def fwd_step(i):
        j = i % len(resident)
        b = B.batch_from_resident(resident[j], offsets[j], ds.voxel_size, ds.point_cloud_range, images[j])
        with torch.no_grad():
            return model(b)

    def train_step(i):
        j = i % len(resident)
        b = B.batch_from_resident(resident[j], offsets[j], ds.voxel_size, ds.point_cloud_range, images[j])
        opt.zero_grad(set_to_none=True)
        res = net(b)
        data = {"point_labels": labels[j], "voxel_labels": voxel_labels[j], "batch_size": b["batch_size"]}
        loss = _losses.compute_loss(res, data, criterion, cfg)
        loss.backward()
        opt.step()
        return res

    def timed(step_fn):
        for i in range(args.warmup):
            step_fn(i)
        barrier(world)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_fn(args.warmup + i)
        barrier(world)
        sec = time.perf_counter() - t0
        pts = sum(pts_per_step[(args.warmup + i) % len(resident)] for i in range(args.steps))
        return D.aggregate_throughput(sec, pts, dev)

    if train:
        model.train()
        dt, n_pts = timed(train_step)
    model.eval()
    dt_f, n_pts_f = timed(fwd_step)  # forward-only eval (BASELINE configs[1] as literally worded)
    if not train:
        dt, n_pts = dt_f, n_pts_f

    out = None
    if rank == 0:
        model.eval()
        b0 = B.batch_from_resident(resident[0], offsets[0], ds.voxel_size, ds.point_cloud_range, images[0])
        roof, per_layer = conv_roofline(model, b0, dev)
        out = {
            "metric": "points/sec fwd+bwd, Waymo 1-sweep ~180k pts @0.1m voxel; logit parity",
            "value": round(n_pts / dt, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"one_sweep": "waymo_one_sweep (BASELINE configs[1])",
                                    "cylinder": "waymo_one_sweep_cylinder (BASELINE configs[2] geometry)",
                                    "multi_sweeps": "waymo_multi_sweeps + image features (BASELINE configs[3], 3 sweeps)"
                                    }[args.workload] + ": synthetic 64-beam scene, "
                                   f"{pts_per_step[0]} pts/step/GPU, voxel {ds.voxel_size}, grid {ds.grid_size.tolist()}, "
                                   f"{'forward-only eval' if not train else 'fwd + criterion (' + '+'.join(cfg.MODEL.LOSSES) + ' on 3 heads) + bwd + SGD step'}",
                       "mode": args.mode, "segmentor": args.segmentor, "scenes_per_step_per_gpu": args.batch,
                       "voxels": int(b0["voxel_coords"].shape[0]), "parallelism": f"dp{world}"},
            "fwd_only": {"value": round(n_pts_f / dt_f, 1), "unit": "points/s",
                         "ms_per_step": round(dt_f / args.steps * 1e3, 3)},
            "roofline": roof,
            "attention_roofline": ATTENTION_REPORT if args.segmentor == "segformer" else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scenes_np[0], cfg, ds, model, args.cpu_points)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_layers.json"), "w") as f:
            json.dump(per_layer, f, indent=1)
        print(json.dumps(out), flush=True)
    if world > 1 or DIST_REHEARSAL:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
