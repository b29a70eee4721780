import sys, json, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import refcfg
from oracle import params
from openseg3d_amd import config, segformer, ops
dev = torch.device("cuda:0")
for prec in ("bf16x3", "fp32"):
    ops.CONV_PRECISION = prec
    for tag, cyl in (("cart", False), ("cyl", True)):
        cfg = config.default_cfg()
        if cyl:
            cfg.DATASET.USE_CYLINDER = True; cfg.DATASET.POINT_CLOUD_RANGE = refcfg.CYL_RANGE; cfg.DATASET.VOXEL_SIZE = refcfg.CYL_VOXEL
        ds = config.DatasetSpec(cfg)
        model = segformer.build_segmentor(cfg, ds); params.fill_by_name(model, seed=0); model = model.to(dev).eval()
        d = np.load(f"tests/golden/segformer_{tag}.npz")
        batch = {"points": torch.from_numpy(d["points"]).to(dev), "voxel_coords": torch.from_numpy(d["voxel_coords"]).to(dev),
                 "point_voxel_ids": torch.from_numpy(d["point_voxel_ids"]).to(dev), "point_id_offset": torch.from_numpy(d["point_id_offset"]).to(dev), "batch_size": int(d["batch_size"])}
        with torch.no_grad(): res = model(batch)
        print(prec, tag, {k: float(np.abs(res[k].cpu().numpy() - d[k]).max()) for k in ("point_out", "voxel_out", "aux_voxel_out")}, "max|logit|", float(np.abs(d["point_out"]).max()))
