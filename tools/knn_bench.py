"""Time ops.knn_query (k = 16, self-query) on the multi-sweep bench scene for several grid configurations (GPU box only)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import ops, scene

dev = torch.device("cuda:0")
pts, n_cur = scene.make_multi_sweep_scene(0)
cur = torch.from_numpy(pts[:n_cur, :3].copy()).to(dev)
xyz = torch.cat([cur, cur + 0.003], 0).contiguous()  # two samples
off = torch.tensor([n_cur, 2 * n_cur], dtype=torch.int32, device=dev)
print("points per sample", n_cur)


def run(levels, min_points):
    ops.KNN_GRID_LEVELS, ops.KNN_GRID_MIN_POINTS = levels, min_points
    for _ in range(2):
        i, d = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        i, d = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    return (time.time() - t0) / 3 * 1e3, i, d


ms, i0, d0 = run(ops.KNN_GRID_LEVELS, 1 << 40)
print(f"brute force: {ms:.2f} ms; mean 16th-neighbour distance {float(d0[:, 15].mean()):.3f} m, max {float(d0[:, 15].max()):.2f} m")
q = torch.quantile(d0[:, 15], torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
print("16th-neighbour distance quantiles 50/90/99/99.9 %:", [round(float(v), 3) for v in q])
for levels in (((0.02, 2), (0.16, 3), (1.28, 3), (10.24, 3)), ((0.4, 3), (1.6, 3), (6.4, 4)), ((0.1, 2), (0.5, 3), (3.0, 4)), ((0.2, 2), (1.0, 3), (6.4, 16)), ((0.25, 4),),
               ((1.0, 16),)):
    ms, i1, d1 = run(levels, 1)
    print(levels, f"{ms:.2f} ms", "exact" if torch.equal(i0, i1) and torch.equal(d0, d1) else "MISMATCH")
