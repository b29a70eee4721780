"""Is the eval forward host-bound?  Per forward of the headline scene: host enqueue time of prepare_batch (index plan, with
its read-backs) and of the feature phase, against the synchronised time of each -- and the same feature phase replayed from
a hipGraph (no host work at all).  GPU box only: python tools/fwd_host_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import batch as B, config, scene, segformer

dev = torch.device("cuda:0")
cfg = config.default_cfg()
ds = config.DatasetSpec(cfg)
torch.manual_seed(0)
model = segformer.build_segmentor(cfg, ds).to(dev).eval()
pts = B.collate_points([scene.make_scene(0)], dev)
n = pts.shape[0]
acc = {"batch+plan enqueue": 0.0, "batch+plan synced": 0.0, "features enqueue": 0.0, "features synced": 0.0}
steps = 20
with torch.no_grad():
    for i in range(steps + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = model.prepare_batch(B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out = model(b)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        if i >= 3:
            acc["batch+plan enqueue"] += t1 - t0
            acc["batch+plan synced"] += t2 - t0
            acc["features enqueue"] += t3 - t2
            acc["features synced"] += t4 - t2
    for k, v in acc.items():
        print(f"{k:22s} {v / steps * 1e3:7.2f} ms")
    # the same feature phase from a captured graph
    b = model.prepare_batch(B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(dict(b))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(dict(b))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    print(f"{'features, graph replay':22s} {(time.perf_counter() - t0) / steps * 1e3:7.2f} ms")
