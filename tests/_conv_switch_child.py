"""Child of tests/test_gpu_parity.py::test_opt_in_conv_schedules_change_no_bit: submanifold conv forward + input gradient on a
seeded scene under whatever SEG3D_* scheduling switches the parent set; writes the results for the parent to compare."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openseg3d_amd import batch as B, config, scene, spconv  # noqa: E402


def main():
    out_path = sys.argv[1]
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([scene.make_scene(3, n_side=2500)], ds.voxel_size, ds.point_cloud_range)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    res = {}
    for li, (cin, cout) in enumerate([(48, 48), (96, 96), (192, 192)]):
        torch.manual_seed(li)
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=True, indice_key=f"k{li}").to(dev)
        x = torch.randn(level.coords.shape[0], cin, device=dev, requires_grad=True)
        t = spconv.SparseConvTensor(x, level.coords, level.shape, 1, _level=level)
        y = conv(t).features
        g = torch.randn_like(y)
        y.backward(g)
        res[f"y{li}"], res[f"dx{li}"], res[f"dw{li}"] = y.detach().cpu(), x.grad.cpu(), conv.weight.grad.cpu()
        res[f"order{li}"] = torch.tensor(0 if level.mask_order() is None else int(level.mask_order().numel()))
        level = level.down()[0]
    torch.save(res, out_path)


if __name__ == "__main__":
    main()
