import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from openseg3d_amd import batch as B, config, losses, ops, scene, spconv, _lib
dev = torch.device("cuda:0")
cfg = config.default_cfg(); ds = config.DatasetSpec(cfg)
n = 174633
x = torch.randn(n, 22, device=dev, requires_grad=True)
y = torch.randint(0, 22, (n,), device=dev)
def host(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3 / reps
print("lovasz fwd enqueue ms", host(lambda: ops.lovasz_softmax(x, y)))
print("ohem fwd enqueue ms", host(lambda: ops.cross_entropy(x, y, ignore_index=255, keep_thresh=0.7)))
print("lovasz ws query ms", host(lambda: _lib.query("seg3d_lovasz_workspace_bytes", n, 22)))
print("knn ws query ms", host(lambda: _lib.query("seg3d_knn_level_workspace_bytes", n)))
b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
lvl = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
for _ in range(3): lvl = lvl.down()[0]
vl = torch.zeros(b["voxel_coords"].shape[0], dtype=torch.long, device=dev)
print("aux labels enqueue ms", host(lambda: ops.aux_voxel_labels(b["voxel_coords"].int(), lvl.coords, vl, 1, ds.voxel_size, ds.point_cloud_range)))
l = ops.lovasz_softmax(x, y)
print("lovasz bwd enqueue ms", host(lambda: torch.autograd.grad(ops.lovasz_softmax(x, y), x)) )
