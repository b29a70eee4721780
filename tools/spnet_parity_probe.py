"""Where does SPNet's logit gap against the oracle come from once the bench's SGD has run?  

Round 4's record: `bench.py --segmentor spnet` read 2.7e-4 at the starting weights and 1.28 (aux head 21.75, rel 2.0e-3)
after 13 steps on random labels.  This probe separates the three candidates:

  conditioning  -- the trained network amplifies ANY fp32 round-off: then the oracle's own fp32 forward differs from the
                   oracle's fp64 forward by the same relative amount, and the GPU's exact-fp32 arm is no closer than
                   the split arm;
  split products-- bf16x3 loses bits in one specific op: then the exact-fp32 arm (SEG3D_CONV_PRECISION=fp32, a second
                   process: the switch is read at import) sits well below the split arm, from a specific stage on;
  a bug         -- a stage's error jumps by orders of magnitude in both arms.

    python tools/spnet_parity_probe.py train  DIR [steps]   # default arithmetic: train, save weights, oracle fp32 + fp64
    python tools/spnet_parity_probe.py eval   DIR TAG       # any arithmetic: GPU eval forward vs the saved oracle taps

Per stage (conv1..conv4, ocr, up4..up1, the three heads) it prints max |x|, and max |delta| / max |x| of
oracle-fp32, GPU against the fp64 oracle.  Uses oracle/ as the checker only; nothing here is a product path.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import batch as B, config, losses, ops, scene, segformer  # noqa: E402

STAGES = ("conv1", "conv2", "conv3", "conv4", "ocr", "up4", "up3", "up2", "up1")
HEADS = ("aux_voxel_out", "voxel_out", "point_out")


def setup():
    cfg = config.default_cfg()
    cfg.MODEL.SEGMENTOR = "spnet"
    return cfg, config.DatasetSpec(cfg)


def oracle_forward(pts, cfg, ds, sd, dtype):
    from oracle import index_ops, model as omodel
    coords, ids = index_ops.voxelize(pts, ds.voxel_size, ds.point_cloud_range)
    batch = {"points": torch.from_numpy(np.pad(pts, ((0, 0), (1, 0)))).to(dtype),
             "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float(),
             "point_voxel_ids": torch.from_numpy(ids).long(), "batch_size": 1,
             "point_id_offset": torch.tensor([float(pts.shape[0])])}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range), "use_multi_sweeps": False,
            "use_image_feature": False}
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    t0 = time.time()
    with torch.no_grad():
        res = omodel.spnet_forward(batch, p, ocfg)
    taps = {k: v.double() for k, v in res["_taps"].items()}
    taps.update({k: res[k].double() for k in HEADS})
    print(f"[probe] oracle {dtype} forward: {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    return taps


def gpu_forward(model, pts, ds, dev):
    got = {}
    hooks = []
    enc = model.voxel_encoder
    for name in STAGES:
        hooks.append(getattr(enc, name).register_forward_hook(
            lambda m, i, o, name=name: got.__setitem__(name, o.features.detach().double().cpu())))
    b = B.batch_from_resident(B.collate_points([pts], dev), [pts.shape[0]], ds.voxel_size, ds.point_cloud_range)
    with torch.no_grad():
        res = model(b)
    for h in hooks:
        h.remove()
    got.update({k: res[k].detach().double().cpu() for k in HEADS})
    return got


def table(ref, arms):
    rows = {}
    for name in STAGES + HEADS:
        mx = float(ref[name].abs().max())
        rows[name] = {"max_abs": mx}
        for tag, t in arms.items():
            d = float((t[name] - ref[name]).abs().max())
            rows[name][tag] = {"max_abs_diff": d, "rel": d / max(mx, 1e-30)}
    return rows


def show(rows, arms):
    print("%-14s %10s " % ("stage", "max|x|") + " ".join("%24s" % a for a in arms))
    for name, r in rows.items():
        print("%-14s %10.3g " % (name, r["max_abs"]) + " ".join("%12.3e (%8.2e)" % (r[a]["max_abs_diff"], r[a]["rel"]) for a in arms))


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda:0")
    cfg, ds = setup()
    pts = scene.make_scene(0)
    if mode == "train":
        steps = int(sys.argv[3]) if len(sys.argv) > 3 else 13
        torch.manual_seed(0)
        model = segformer.build_segmentor(cfg, ds).to(dev)
        sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        crit = losses.build_criterion(cfg, ds)
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, fused=True)  # bench.py's setting
        scenes = [scene.make_scene(s) for s in range(4)]
        res_dev = [B.collate_points([s], dev) for s in scenes]
        labels = [torch.randint(0, 22, (s.shape[0],), device=dev) for s in scenes]
        model.train()
        for i in range(steps):
            j = i % 4
            b = B.batch_from_resident(res_dev[j], [scenes[j].shape[0]], ds.voxel_size, ds.point_cloud_range)
            vl = ops.prepare_voxel_labels(b["point_voxel_ids"], labels[j], b["voxel_coords"].shape[0], ignore_index=ds.ignore_index).long()
            opt.zero_grad(set_to_none=True)
            res = model(b)
            loss = losses.compute_loss(res, {"point_labels": labels[j], "voxel_labels": vl, "batch_size": 1}, crit, cfg)
            loss.backward()
            opt.step()
            print(f"[probe] step {i} loss {float(loss):.4f}", file=sys.stderr, flush=True)
        model.eval()
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        torch.save({"start": sd0, "trained": sd, "steps": steps}, os.path.join(out_dir, "weights.pt"))
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
        for tag, w in (("start", sd0), ("trained", sd)):
            torch.save({"f64": oracle_forward(pts, cfg, ds, w, torch.float64), "f32": oracle_forward(pts, cfg, ds, w, torch.float32)},
                       os.path.join(out_dir, f"oracle_{tag}.pt"))
        return
    tag = sys.argv[3]
    w = torch.load(os.path.join(out_dir, "weights.pt"))
    report = {"arm": tag, "conv_precision": ops.CONV_PRECISION, "steps": w["steps"]}
    for which in ("start", "trained"):
        model = segformer.build_segmentor(cfg, ds).to(dev).eval()
        model.load_state_dict(w[which])
        orc = torch.load(os.path.join(out_dir, f"oracle_{which}.pt"))
        rows = table(orc["f64"], {"oracle_f32": orc["f32"], "gpu_" + tag: gpu_forward(model, pts, ds, dev)})
        print(f"--- {which} weights, GPU arithmetic {ops.CONV_PRECISION}: max |delta| (rel) against the fp64 oracle")
        show(rows, ["oracle_f32", "gpu_" + tag])
        report[which] = rows
    with open(os.path.join(ROOT, "gpurun_out", f"spnet_probe_{tag}.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
