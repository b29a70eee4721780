"""Oracle restatement of Segformer.forward in eval mode -- TEST INFRASTRUCTURE ONLY.

Functional torch-CPU forward driven by a state_dict keyed exactly like the
reference module tree (SURVEY.md section 8b "state-dict"):

  seg3d/models/segmentors/segformer.py:94-146        (wiring)
  seg3d/models/backbones/pointtransformer.py:47-219  (blocks, U-Net)
  seg3d/models/voxel_encoders/vfe.py:16-27, layers/se_layer.py:16-29

eval() semantics: BatchNorm uses running stats, Dropout/DropPath are identity.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import sparse_conv as sc
from . import window as win

BN_EPS_SPARSE = 1e-3  # pointtransformer.py:129
NUM_HEADS = 8         # pointtransformer.py:142-157


def _bn(x, p, prefix, eps):
    return F.batch_norm(x, p[prefix + "running_mean"], p[prefix + "running_var"],
                        p[prefix + "weight"], p[prefix + "bias"], False, 0.0, eps)


def _point_encoder(x, p, pre="point_encoder."):
    """segformer.py:21-32."""
    x = _bn(x, p, pre + "0.", 1e-5)
    x = F.relu(_bn(F.linear(x, p[pre + "1.weight"]), p, pre + "2.", 1e-5))
    x = F.relu(_bn(F.linear(x, p[pre + "4.weight"]), p, pre + "5.", 1e-5))
    x = F.relu(_bn(F.linear(x, p[pre + "7.weight"]), p, pre + "8.", 1e-5))
    return F.linear(x, p[pre + "10.weight"], p[pre + "10.bias"])


def _conv_module(x, nbr, p, pre):
    """ConvModule = conv(no bias) + BN + ReLU, spconv_utils.py:13-32."""
    y = sc.apply_rulebook(x, nbr, p[pre + "0.weight"])
    return F.relu(_bn(y, p, pre + "1.", BN_EPS_SPARSE))


def _basic_block(x, nbr, p, pre):
    """SparseBasicBlock.forward (no SE/SA in PointTransformer), pointtransformer.py:47-66."""
    y = sc.apply_rulebook(x, nbr, p[pre + "conv1.weight"], p[pre + "conv1.bias"])
    y = F.relu(_bn(y, p, pre + "bn1.", BN_EPS_SPARSE))
    y = sc.apply_rulebook(y, nbr, p[pre + "conv2.weight"], p[pre + "conv2.bias"])
    y = _bn(y, p, pre + "bn2.", BN_EPS_SPARSE)
    return F.relu(y + x)


def _up_block(x_bottom, x_lateral, nbr_subm, nbr_out, p, pre):
    """UpBlock.forward, pointtransformer.py:104-113 (channel_reduction :88-102)."""
    t = _basic_block(x_lateral, nbr_subm, p, pre + "transform.")
    cat = torch.cat([x_bottom, t], dim=1)
    m = _conv_module(cat, nbr_subm, p, pre + "bottleneck.")
    red = cat.view(cat.shape[0], m.shape[1], -1).sum(dim=2)
    return _conv_module(m + red, nbr_out, p, pre + "out.")


def segformer_forward(batch, params, cfg):
    """batch: dict of torch-CPU tensors with the load_data_to_gpu dtypes
    (seg3d/utils/data_utils.py:6-15): points f32 [N,1+D], point_voxel_ids i64 [N],
    voxel_coords f32 [M,4], batch_size int.  cfg: dict with grid_size (x,y,z),
    batching_info (list of 4 dicts with int keys), window_shape, depths.
    cfg may set use_multi_sweeps / use_image_feature (configs/waymo_multi_sweeps.yaml): then batch also
    carries point_id_offset (cumulative current-sweep rows) and point_image_features.
    Returns the reference's result OrderedDict plus the intermediates tests need.
    """
    p = params
    points = batch["points"][:, 1:]
    ids = batch["point_voxel_ids"]
    multi = bool(cfg.get("use_multi_sweeps", False))
    if multi:
        cur = points[:, 3] == 0  # segformer.py:98
        cur_points = points[cur]
    else:
        cur_points = points
    pf = _point_encoder(cur_points, p)

    ok = ids != -1
    if multi:
        vox = sc.scatter(points[ok], ids[ok], reduce="mean")  # VFE(dim_point, mean) over all sweeps, segformer.py:107
    else:
        vox = sc.scatter(pf[ok], ids[ok], reduce="max")  # VFE(max), vfe.py:24-25

    coords = batch["voxel_coords"].int().numpy()
    sparse_shape = np.asarray(cfg["grid_size"])[::-1]  # pointtransformer.py:120
    assert vox.shape[0] == coords.shape[0]
    lvl = [sc.Sites(coords, sparse_shape)]
    pre = "point_transformer."
    grid_xyz = np.asarray(cfg["grid_size"], dtype=np.float64)

    def stage(x, sites, k):
        c = x.shape[1]
        info = win.window_partition(torch.from_numpy(sites.coords), cfg["batching_info"][k],
                                    cfg["window_shape"], grid_xyz / (2 ** k), c)
        return win.swformer_block(x, info, p, f"{pre}swformer_block{k + 1}.1.",
                                  cfg["depths"][k], NUM_HEADS)

    x1 = _conv_module(vox, lvl[0].subm(), p, pre + "conv_input.")
    x1 = stage(x1, lvl[0], 0)
    feats = [x1]
    for k in range(1, 4):
        coarse, fwd, _ = lvl[k - 1].down()
        lvl.append(coarse)
        x = _conv_module(feats[-1], fwd, p, f"{pre}conv_down{k}.")
        feats.append(stage(x, coarse, k))

    aux = F.linear(feats[3], p[pre + "aux_voxel_classifier.0.weight"])

    x = _up_block(feats[3], feats[3], lvl[3].subm(), lvl[2].down()[2], p, pre + "up4.")
    x = _up_block(x, feats[2], lvl[2].subm(), lvl[1].down()[2], p, pre + "up3.")
    x = _up_block(x, feats[1], lvl[1].subm(), lvl[0].down()[2], p, pre + "up2.")
    x = _up_block(x, feats[0], lvl[0].subm(), lvl[0].subm(), p, pre + "up1.")
    voxel_out = F.linear(x, p[pre + "voxel_classifier.0.weight"])

    pv = sc.voxel_to_point(x, ids[cur] if multi else ids)
    f = torch.cat([pf, pv], dim=1)
    if cfg.get("use_image_feature", False):  # DeepFusionBlock.forward, deep_fusion.py:26-45 (eval: no dropout)
        from .knn import knn_query
        img = batch["point_image_features"]
        df = "deep_fusion."
        q = F.linear(f, p[df + "q_embedding.weight"], p[df + "q_embedding.bias"])
        kk = F.linear(img, p[df + "k_embedding.weight"], p[df + "k_embedding.bias"])
        vv = F.linear(img, p[df + "v_embedding.weight"], p[df + "v_embedding.bias"])
        off = batch["point_id_offset"].int()
        nn_ids, _ = knn_query(16, cur_points.contiguous(), cur_points.contiguous(), off, off)
        nn_ids = nn_ids.long()
        w = torch.einsum("nc,nkc->nk", q, kk[nn_ids]) / np.sqrt(q.shape[-1])
        w[(img.sum(dim=1) == 0)[nn_ids]] = float("-inf")
        w = torch.nan_to_num(torch.softmax(w, dim=-1))
        o = torch.einsum("nk,nkc->nc", w, vv[nn_ids])
        f = torch.cat([f, F.linear(o, p[df + "c_proj.weight"], p[df + "c_proj.bias"])], dim=1)
    fe = "fusion_encoder."
    f = F.relu(_bn(F.linear(f, p[fe + "0.weight"]), p, fe + "1.", 1e-5))
    f = F.relu(_bn(F.linear(f, p[fe + "3.weight"]), p, fe + "4.", 1e-5))
    f = F.relu(_bn(F.linear(f, p[fe + "6.weight"]), p, fe + "7.", 1e-5))

    bidx = (batch["points"][:, 0][cur] if multi else batch["points"][:, 0]).long()
    g = sc.scatter(f, bidx, reduce="mean")  # FlattenSELayer, se_layer.py:24-28
    g = torch.sigmoid(F.linear(F.relu(F.linear(g, p["se.fc.0.weight"])), p["se.fc.2.weight"]))
    f = f + f * g[bidx]

    h = F.relu(_bn(F.linear(f, p["classifier.0.weight"]), p, "classifier.1.", 1e-5))
    point_out = F.linear(h, p["classifier.4.weight"])

    res = OrderedDict()
    res["point_out"] = point_out
    res["voxel_out"] = voxel_out
    res["aux_voxel_out"] = aux
    res["voxel_coords"] = torch.from_numpy(lvl[0].coords)
    res["aux_voxel_coords"] = torch.from_numpy(lvl[3].coords)
    res["_levels"] = lvl
    res["_stage_feats"] = feats
    res["_voxel_in"] = vox
    return res
