"""Time the auxiliary-label lookup (tools/train.py:86-104: k = 1, coarse voxel centres against fine voxel centres) for
several grid configurations, on the headline scene (GPU box only)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import batch as B, config, ops, scene, spconv

dev = torch.device("cuda:0")
cfg = config.default_cfg()
ds = config.DatasetSpec(cfg)
nb = int(os.environ.get("BATCH", "1"))
b = B.make_batch([scene.make_scene(s) for s in range(nb)], ds.voxel_size, ds.point_cloud_range)
lvl = spconv.SiteLevel(b["voxel_coords"], ds.grid_size[::-1], nb)
for _ in range(3):
    lvl = lvl.down()[0]
centers = ops.get_voxel_centers(b["voxel_coords"][:, 1:], 1.0, ds.voxel_size, ds.point_cloud_range).contiguous()
aux = ops.get_voxel_centers(lvl.coords[:, 1:], 8.0, ds.voxel_size, ds.point_cloud_range).contiguous()
off = torch.cumsum(torch.bincount(b["voxel_coords"][:, 0].long(), minlength=nb), 0).int()
aoff = torch.cumsum(torch.bincount(lvl.coords[:, 0].long(), minlength=nb), 0).int()
print("fine", centers.shape[0], "coarse", aux.shape[0])
ops.KNN_GRID_MIN_POINTS = 1 << 40
ref_i, ref_d = ops.knn_query(1, centers, aux, off, aoff)
torch.cuda.synchronize()
print("nearest-distance quantiles", [round(float(v), 3) for v in torch.quantile(ref_d[:, 0], torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=dev))])
ops.KNN_GRID_MIN_POINTS = 1
for levels in (None, ((0.1, 2), (0.8, 3), (6.4, 4)), ((0.1, 1), (0.8, 2), (6.4, 3)), ((0.1, 0), (0.8, 2), (6.4, 3)),
               ((0.1, 3), (0.8, 2), (6.4, 3)), ((0.1, 1), (0.8, 1), (6.4, 2)), ((0.1, 1), (0.8, 4)), ((0.1, 0), (0.8, 3)),
               ((0.05, 0), (0.4, 2), (3.2, 2)), ((0.05, 0), (0.4, 4), (3.2, 2)), ((0.15, 1), (1.2, 2), (9.6, 2))):
    for _ in range(2):
        i, d = ops.knn_query(1, centers, aux, off, aoff, levels=levels)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        i, d = ops.knn_query(1, centers, aux, off, aoff, levels=levels)
    torch.cuda.synchronize()
    print(levels, f"{(time.time() - t0) * 200:.3f} ms", "exact" if torch.equal(i, ref_i) and torch.equal(d, ref_d) else "MISMATCH")
