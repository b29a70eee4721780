"""Time ops.knn_query on exactly what DeepFusionBlock hands it in the multi-sweep bench (GPU box only)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import ops, scene

dev = torch.device("cuda:0")
rows = []
offs = []
tot = 0
for seed in (0, 1):
    pts, n_cur = scene.make_multi_sweep_scene(seed)
    rows.append(torch.from_numpy(pts[:n_cur].copy()))
    tot += n_cur
    offs.append(tot)
xyz = torch.cat(rows, 0).to(dev).contiguous()  # [N, 6] current-sweep rows, handed over as is (stride-3 reinterpretation)
off = torch.tensor(offs, dtype=torch.int32, device=dev)
print(xyz.shape, offs)
for name, minp in (("brute", 1 << 40), ("grid", 1)):
    ops.KNN_GRID_MIN_POINTS = minp
    for _ in range(2):
        i, d = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    t0 = time.time()
    i, d = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    print(name, f"{(time.time() - t0) * 1e3:.2f} ms", "16th distance quantiles",
          [round(float(v), 4) for v in torch.quantile(d[:, 15], torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev))])
ops.KNN_GRID_MIN_POINTS = 1
for levels in (((0.05, 2), (0.4, 3), (3.2, 4)), ((0.4, 3), (1.6, 3), (6.4, 4)), ((0.04, 3), (0.32, 3), (2.56, 3), (20.48, 2)),
               ((0.1, 3), (0.8, 3), (6.4, 3)), ((0.01, 2), (0.08, 3), (0.64, 3), (5.12, 4))):
    ops.KNN_GRID_LEVELS = levels
    for _ in range(2):
        i1, d1 = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    t0 = time.time()
    i1, d1 = ops.knn_query(16, xyz, xyz, off, off)
    torch.cuda.synchronize()
    t_quirk = (time.time() - t0) * 1e3
    x3 = xyz[:, :3].contiguous()
    for _ in range(2):
        ops.knn_query(16, x3, x3, off, off)
    torch.cuda.synchronize()
    t0 = time.time()
    ops.knn_query(16, x3, x3, off, off)
    torch.cuda.synchronize()
    print(levels, f"[N,6] as handed over: {t_quirk:.2f} ms ({'exact' if torch.equal(i, i1) else 'MISMATCH'});  xyz only: {(time.time() - t0) * 1e3:.2f} ms")
