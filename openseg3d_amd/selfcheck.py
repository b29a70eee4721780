"""smoke(): one small pass of the hot path on cuda:0, checked against the CPU oracle.

This is the only module of the package that touches ``oracle/`` (as the checker); it is called by
``__graft_entry__.smoke()`` and never by the product path."""
import sys

import torch


def run_smoke(n_points=6000, verbose=True):
    from . import _lib, batch, config, scene, segformer
    _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (the HIP path has no CPU fallback)")
    from oracle import index_ops, model as omodel, params  # checker only

    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    model = segformer.build_segmentor(cfg, ds)
    params.fill_by_name(model, seed=0)
    model = model.to(dev).eval()
    samples = [scene.make_small_scene(101, n_points, extent=8.0)]
    b = batch.make_batch(samples, ds.voxel_size, ds.point_cloud_range, device=dev)
    with torch.no_grad():
        res = model(dict(b))
    torch.cuda.synchronize()

    coords_ref, ids_ref = index_ops.voxelize(samples[0], ds.voxel_size, ds.point_cloud_range)
    assert (b["voxel_coords"].cpu().int().numpy()[:, 1:] == coords_ref).all(), "voxel coordinates differ"
    assert (b["point_voxel_ids"].cpu().numpy() == ids_ref).all(), "point->voxel ids differ"
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items() if k != "point_voxel_index"}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range),
            "batching_info": [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO],
            "window_shape": cfg.MODEL.WINDOW_SHAPE, "depths": cfg.MODEL.DEPTHS}
    with torch.no_grad():
        ref = omodel.segformer_forward(cpu, {k: v.cpu() for k, v in model.state_dict().items()}, ocfg)
    err = float((res["point_out"].cpu() - ref["point_out"]).abs().max())
    if verbose:
        print(f"smoke: {samples[0].shape[0]} points, {coords_ref.shape[0]} voxels, max |logit diff| vs oracle = {err:.3e}",
              file=sys.stderr)
    assert err < 1e-3, f"per-point logits differ from the oracle by {err}"
    return err
