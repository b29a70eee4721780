"""Which Python lines launch the SHORT kernels of a training step (torch.profiler with stacks; GPU box only):
python tools/small_kernel_sources.py [max_us=15]  ->  count / total us per (aten op, innermost repo source line)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import batch as B, config, losses, ops, scene, segformer  # noqa: E402


def main():
    max_us = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, fused=True)
    crit = losses.build_criterion(cfg, ds)
    pts = B.collate_points([scene.make_scene(0)], dev)
    n = pts.shape[0]
    labels = torch.randint(0, 22, (n,), device=dev)

    def step():
        b = B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range)
        vl = ops.prepare_voxel_labels(b["point_voxel_ids"], labels.to(torch.uint8), b["voxel_coords"].shape[0]).long()
        opt.zero_grad(set_to_none=True)
        res = model(b)
        loss = losses.compute_loss(res, {"point_labels": labels, "voxel_labels": vl, "batch_size": 1}, crit, cfg)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        us = sum(k.duration for k in e.kernels)
        if us / max(len(e.kernels), 1) > max_us:
            continue
        src = next((fr for fr in e.stack if "openseg3d_amd" in fr or "bench.py" in fr or "tools/" in fr), e.stack[0] if e.stack else "?")
        key = (e.name, src.strip()[-90:])
        agg[key][0] += len(e.kernels)
        agg[key][1] += us
    tot_n = sum(v[0] for v in agg.values())
    print(f"kernels <= {max_us} us: {tot_n} launches, {sum(v[1] for v in agg.values()) / 1e3:.2f} ms")
    for (name, src), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{cnt:4d} {us:8.1f} us  {name[:34]:34s} {src}")


if __name__ == "__main__":
    main()
