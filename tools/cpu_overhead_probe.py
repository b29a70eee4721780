"""How much of a training step is host time?  Enqueue time of forward / criterion / backward / optimizer (no sync
inside; the host syncs of the index plan happen in `prepare`, measured separately) against the synchronised step time.
GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import batch as B, config, losses, ops, scene, segformer

dev = torch.device("cuda:0")
cfg = config.default_cfg()
ds = config.DatasetSpec(cfg)
torch.manual_seed(0)
model = segformer.build_segmentor(cfg, ds).to(dev).train()
fused = os.environ.get("FOREACH", "0") != "1"
opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, fused=fused)
crit = losses.build_criterion(cfg, ds)
pts = B.collate_points([scene.make_scene(0)], dev)
n = pts.shape[0]
labels = torch.randint(0, 22, (n,), device=dev)
b0 = B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range)
vox_labels = ops.prepare_voxel_labels(b0["point_voxel_ids"], labels, b0["voxel_coords"].shape[0]).long()
acc = {"batch": 0.0, "fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0, "total": 0.0}
steps = 12
for i in range(steps + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    res = model(b)
    t2 = time.perf_counter()
    loss = losses.compute_loss(res, {"point_labels": labels, "voxel_labels": vox_labels, "batch_size": 1}, crit, cfg)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    if i >= 3:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t0)):
            acc[k] += v * 1e3 / steps
print("fused SGD" if fused else "foreach SGD", {k: round(v, 2) for k, v in acc.items()},
      "host enqueue total", round(acc["batch"] + acc["fwd"] + acc["loss"] + acc["bwd"] + acc["opt"], 2), "ms; cores", os.cpu_count())
