"""Per-layer timing of the 20 sparse convolutions of one forward on the headline scene (bench.py's conv_roofline):
python tools/conv_bench.py   (honours SEG3D_CONV_XCD / SEG3D_CONV_NBT)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from openseg3d_amd import batch as B, config, scene, segformer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).eval()
    b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
    with torch.no_grad():
        model(dict(b))
    roof, layers = bench.conv_roofline(model, b, dev)
    tot = 0.0
    for l in layers:
        tot += l["us"]
        print(f"rows {l['rows']:7d} {l['cin']:4d} -> {l['cout']:4d}  {l['us']:8.1f} us  {l['bound']:12s} frac {l['frac']:.3f}")
    print(f"total {tot / 1e3:.3f} ms, aggregate {roof['achieved']:.0f} GB/s = {roof['frac']:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
