"""Oracle restatement of Segformer.forward and SPNet.forward in eval mode -- TEST INFRASTRUCTURE ONLY.

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


def _flatten_se(x, bidx, p, pre):
    """FlattenSELayer.forward, se_layer.py:24-28: per-sample mean -> fc -> sigmoid gate."""
    g = sc.scatter(x, bidx, reduce="mean")
    g = torch.sigmoid(F.linear(F.relu(F.linear(g, p[pre + "fc.0.weight"])), p[pre + "fc.2.weight"]))
    return x * g[bidx]


def _basic_block(x, nbr, p, pre, se_batch=None):
    """SparseBasicBlock.forward, pointtransformer.py:47-66 (no SE/SA) and spconv_unet.py:46-66
    (``se_batch`` = the batch column of the sites when the block was built with_se=True)."""
    y = sc.apply_rulebook(x, nbr, p[pre + "conv1.weight"], p[pre + "conv1.bias"])
    y = F.relu(_bn(y, p, pre + "bn1.", BN_EPS_SPARSE))
    y = sc.apply_rulebook(y, nbr, p[pre + "conv2.weight"], p[pre + "conv2.bias"])
    y = _bn(y, p, pre + "bn2.", BN_EPS_SPARSE)
    if se_batch is not None:
        y = _flatten_se(y, se_batch, p, pre + "se.")
    return F.relu(y + x)


def _up_block(x_bottom, x_lateral, nbr_subm, nbr_out, p, pre):
    """UpBlock.forward, pointtransformer.py:104-113 (channel_reduction :88-102)."""
    t = _basic_block(x_lateral, nbr_subm, p, pre + "transform.")
    cat = torch.cat([x_bottom, t], dim=1)
    m = _conv_module(cat, nbr_subm, p, pre + "bottleneck.")
    red = cat.view(cat.shape[0], m.shape[1], -1).sum(dim=2)
    return _conv_module(m + red, nbr_out, p, pre + "out.")


def _fusion_head(batch, p, cfg, pf, x, cur, cur_points):
    """Voxel->point gather, optional DeepFusion, fusion MLP, SE and classifier: segformer.py:112-138, identical in
    spnet.py:113-140."""
    ids = batch["point_voxel_ids"]
    multi = cur is not None
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
    f = f + _flatten_se(f, bidx, p, "se.")

    h = F.relu(_bn(F.linear(f, p["classifier.0.weight"]), p, "classifier.1.", 1e-5))
    point_out = F.linear(h, p["classifier.4.weight"])

    return point_out


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

    point_out = _fusion_head(batch, p, cfg, pf, x, cur if multi else None, cur_points)

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


def _ocr(x, sites, aux, batch_size, p, pre):
    """OCRLayer.forward in eval mode, seg3d/models/layers/ocr.py:104-116 (SpatialGatherModule :19-36,
    ObjectAttentionBlock :69-82)."""
    f = F.relu(_bn(sc.apply_rulebook(x, sites.subm(), p[pre + "transform_input.0.weight"]), p,
                   pre + "transform_input.1.", 1e-5))
    bidx = torch.from_numpy(sites.coords[:, 0]).long()
    oc = pre + "object_context_block."

    def proj(t, name):
        return F.relu(_bn(F.linear(t, p[oc + name + ".0.weight"]), p, oc + name + ".1.", 1e-5))

    key_channels = p[oc + "query_project.0.weight"].shape[0]
    out = torch.zeros_like(f)
    for i in range(batch_size):
        sel = bidx == i
        fi = f[sel]
        prob = F.softmax(aux[sel].t(), dim=1)  # scale = 1, ocr.py:86
        proxy = prob @ fi  # [classes, C]
        sim = F.softmax((key_channels ** -0.5) * (proj(fi, "query_project") @ proj(proxy, "key_project").t()), dim=-1)
        out[sel] = proj(sim @ proj(proxy, "value_project"), "bottleneck")
    cat = torch.cat([out, f], dim=1)
    return F.relu(_bn(F.linear(cat, p[pre + "bottleneck.0.weight"]), p, pre + "bottleneck.1.", 1e-5))


def spnet_forward(batch, params, cfg):
    """SPNet.forward in eval mode: seg3d/models/segmentors/spnet.py:94-148 over
    SparseUnet.forward, seg3d/models/backbones/spconv_unet.py:186-233.  Same batch / cfg / result contract as
    segformer_forward (cfg needs grid_size only, plus the multi-sweep / image switches)."""
    p = params
    points = batch["points"][:, 1:]
    ids = batch["point_voxel_ids"]
    multi = bool(cfg.get("use_multi_sweeps", False))
    cur = points[:, 3] == 0 if multi else None  # spnet.py:97
    cur_points = points[cur] if multi else points
    pf = _point_encoder(cur_points, p)
    ok = ids != -1
    vox = sc.scatter(points[ok], ids[ok], reduce="mean") if multi else sc.scatter(pf[ok], ids[ok], reduce="max")

    coords = batch["voxel_coords"].int().numpy()
    lvl = [sc.Sites(coords, np.asarray(cfg["grid_size"])[::-1])]
    pre = "voxel_encoder."
    x = _conv_module(vox, lvl[0].subm(), p, pre + "conv_input.")
    for j in range(2):
        x = _basic_block(x, lvl[0].subm(), p, f"{pre}conv1.{j}.")
    feats = [x]
    for k in range(1, 4):  # conv2..conv4: strided conv + 3 blocks, SE on the last block of conv3 / conv4
        coarse, fwd, _ = lvl[k - 1].down()
        lvl.append(coarse)
        x = _conv_module(feats[-1], fwd, p, f"{pre}conv{k + 1}.0.")
        bcol = torch.from_numpy(coarse.coords[:, 0]).long()
        for j in (1, 2, 3):
            x = _basic_block(x, coarse.subm(), p, f"{pre}conv{k + 1}.{j}.", se_batch=bcol if (k >= 2 and j == 3) else None)
        feats.append(x)

    aux = F.linear(feats[3], p[pre + "aux_voxel_classifier.0.weight"])
    x4 = _ocr(feats[3], lvl[3], aux, int(batch["batch_size"]), p, pre + "ocr.")

    taps = OrderedDict((f"conv{k + 1}", f) for k, f in enumerate(feats))  # per-stage outputs (tools/spnet_parity_probe.py)
    taps["ocr"] = x4
    x = taps["up4"] = _up_block(x4, x4, lvl[3].subm(), lvl[2].down()[2], p, pre + "up4.")
    x = taps["up3"] = _up_block(x, feats[2], lvl[2].subm(), lvl[1].down()[2], p, pre + "up3.")
    x = taps["up2"] = _up_block(x, feats[1], lvl[1].subm(), lvl[0].down()[2], p, pre + "up2.")
    x = taps["up1"] = _up_block(x, feats[0], lvl[0].subm(), lvl[0].subm(), p, pre + "up1.")
    voxel_out = F.linear(x, p[pre + "voxel_classifier.0.weight"])

    res = OrderedDict()
    res["point_out"] = _fusion_head(batch, p, cfg, pf, x, cur, cur_points)
    res["voxel_out"] = voxel_out
    res["aux_voxel_out"] = aux
    res["voxel_coords"] = torch.from_numpy(lvl[0].coords)
    res["aux_voxel_coords"] = torch.from_numpy(lvl[3].coords)
    res["_levels"] = lvl
    res["_stage_feats"] = feats
    res["_voxel_in"] = vox
    res["_taps"] = taps
    return res
