"""List every implicit host synchronisation of one training step (torch.cuda.set_sync_debug_mode).  GPU box only."""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import batch as B, config, losses, ops, scene, segformer

dev = torch.device("cuda:0")
cfg = config.default_cfg()
cfg.MODEL.SEGMENTOR = os.environ.get("SEGMENTOR", "segformer")
ds = config.DatasetSpec(cfg)
model = segformer.build_segmentor(cfg, ds).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, fused=True)
crit = losses.build_criterion(cfg, ds)
nb = int(os.environ.get("BATCH", "1"))
samples = [scene.make_scene(s) for s in range(nb)]
pts = B.collate_points(samples, dev)
offs = list(__import__("numpy").cumsum([s.shape[0] for s in samples]))
n = pts.shape[0]
labels = torch.randint(0, 22, (n,), device=dev)
b0 = B.batch_from_resident(pts, offs, ds.voxel_size, ds.point_cloud_range)
vox = ops.prepare_voxel_labels(b0["point_voxel_ids"], labels, b0["voxel_coords"].shape[0]).long()


def step():
    b = B.batch_from_resident(pts, offs, ds.voxel_size, ds.point_cloud_range)
    opt.zero_grad(set_to_none=True)
    res = model(b)
    loss = losses.compute_loss(res, {"point_labels": labels, "voxel_labels": vox, "batch_size": nb}, crit, cfg)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter()
for x in w:
    if "synchroniz" in str(x.message).lower():
        c[(os.path.relpath(x.filename), x.lineno)] += 1
for (f, l), k in sorted(c.items()):
    print(f"{k:3d} x {f}:{l}")
print("total implicit syncs in one step:", sum(c.values()))
