"""Does logit parity against the oracle survive a few optimiser steps?  Separates (a) stale caches in the GPU modules
(fresh module loaded from the same state_dict agrees with the oracle, the trained one does not) from (b) a model that the
bench's random-label SGD has driven into an ill-conditioned state (both disagree; activations / tau extreme)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from openseg3d_amd import batch as B, config, losses, ops, scene, segformer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev)
    pts = scene.make_scene(0)[:n_pts]
    crit = losses.build_criterion(cfg, ds)
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4, fused=True)
    res_dev = B.collate_points([pts], dev)
    labels = torch.randint(0, 22, (pts.shape[0],), device=dev)

    def check(tag, m):
        m.eval()
        rep, o_res, o_c, o_i = bench.cpu_baseline(pts, pts.shape[0], None, cfg, ds, m)
        par = bench.parity_report(pts, pts.shape[0], None, ds, m, dev, o_res, o_c, o_i)
        taus = [float(p.detach().min()) for k, p in m.named_parameters() if k.endswith("tau")]
        print(tag, "diff %.3e max|logit| %.2f  tau min %.4f max %.4f" % (par["max_abs_logit_diff"], par["max_abs_logit"],
                                                                          min(taus), max(taus)), flush=True)

    check("init        ", model)
    model.train()
    for i in range(steps):
        b = B.batch_from_resident(res_dev, [pts.shape[0]], ds.voxel_size, ds.point_cloud_range)
        vl = ops.prepare_voxel_labels(b["point_voxel_ids"], labels, b["voxel_coords"].shape[0], ignore_index=ds.ignore_index).long()
        opt.zero_grad(set_to_none=True)
        res = model(b)
        loss = losses.compute_loss(res, {"point_labels": labels, "voxel_labels": vl, "batch_size": 1}, crit, cfg)
        loss.backward()
        opt.step()
        print("step", i, "loss %.4f" % float(loss), flush=True)
        if i in (0, 2, steps - 1):
            check("after step %d (same module)" % i, model)
            model.train()
    fresh = segformer.build_segmentor(cfg, ds).to(dev)
    fresh.load_state_dict(model.state_dict())
    check("fresh module, trained weights", fresh)
    check("trained module again         ", model)


if __name__ == "__main__":
    main()
