"""In-situ duration of every dense Linear launch (ops._linear_apply / seg3d_linear_fwd_sum) inside a real eval forward of the
headline scene (HIP events around each launch, median of 5 forwards), next to the same launch repeated in isolation on the
same tensors right after the forward (warm caches).  GPU box only: python tools/linear_insitu.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseg3d_amd import _lib, batch as B, config, ops, scene, segformer

dev = torch.device("cuda:0")
cfg = config.default_cfg()
ds = config.DatasetSpec(cfg)
torch.manual_seed(0)
model = segformer.build_segmentor(cfg, ds).to(dev).eval()
pts = B.collate_points([scene.make_scene(0)], dev)
n = pts.shape[0]
rec = []
orig_apply, orig_call = ops._linear_apply, _lib.call


def timed_apply(x, packed, bias, cin, cout, addend=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = orig_apply(x, packed, bias, cin, cout, addend)
    b.record()
    rec.append(("linear", x.shape[0], cin, cout, a, b, (x, packed, bias, addend)))
    return y


ops._linear_apply = timed_apply
with torch.no_grad():
    runs = []
    for i in range(7):
        rec.clear()
        b = model.prepare_batch(B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range))
        torch.cuda.synchronize()
        model(b)
        torch.cuda.synchronize()
        if i >= 2:
            runs.append([(k, m, ci, co, a.elapsed_time(e) * 1e3) for k, m, ci, co, a, e, _ in rec])
    last = list(rec)
    by = collections.OrderedDict()
    for j in range(len(runs[0])):
        k, m, ci, co, _ = runs[0][j]
        med = sorted(r[j][4] for r in runs)[len(runs) // 2]
        by.setdefault((m, ci, co), []).append(med)
    iso = {}
    for k, m, ci, co, a, e, args in last:
        key = (m, ci, co)
        if key in iso:
            continue
        x, packed, bias, addend = args
        ts = []
        for _ in range(12):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig_apply(x, packed, bias, ci, co, addend)
            t.record()
            t.synchronize()
            ts.append(s.elapsed_time(t) * 1e3)
        iso[key] = sorted(ts)[len(ts) // 2]
    tot_in = tot_iso = 0.0
    for (m, ci, co), v in by.items():
        mean = sum(v) / len(v)
        tot_in += sum(v)
        tot_iso += iso[(m, ci, co)] * len(v)
        print(f"{ci:4d} -> {co:4d} @{m:7d}: {len(v):3d} launches, in the forward {mean:7.1f} us (min {min(v):6.1f}, max {max(v):6.1f}), repeated alone {iso[(m, ci, co)]:7.1f} us, "
              f"{m * (ci + co) * 4 / mean / 1e6:5.2f} TB/s in situ")
    print(f"sum per forward: in situ {tot_in / 1e3:.3f} ms, alone {tot_iso / 1e3:.3f} ms ({len(runs[0])} launches through _linear_apply)")
