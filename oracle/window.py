"""Oracle restatement of the sparse-window (SWFormer) stage -- TEST INFRASTRUCTURE ONLY.

torch-CPU, functional (parameters come in as a dict keyed like the reference's
state_dict).  Follows the reference's *padded, level-bucketed* formulation so
that every intermediate (window ids, levels, flat->window slots, positional
embedding, padding masks) can be compared bit-for-bit with the HIP path:

  seg3d/utils/swformer_utils.py               (index transforms)
  seg3d/models/layers/point_transformer_layer.py  (partition, attention wiring)
  seg3d/models/layers/cosine_msa.py           (cosine multi-head attention)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .index_ops import ingroup_rank


# --------------------------------------------------------------------------- a13
def window_coords(coors, sparse_shape_xyz, window_shape, do_shift):
    """get_window_coors, swformer_utils.py:108-154.

    coors: int64 [M,4] rows [b,z,y,x].  sparse_shape_xyz may be fractional
    (pointtransformer.py:146-154 passes grid/2**k as a float array); only
    ceil(S/win)+1 is taken from it (swformer_utils.py:119-121).
    Returns (batch_win_inds int64[M], coors_in_win int64[M,3] as z,y,x).
    """
    wx, wy, wz = window_shape
    sx, sy, sz = (float(s) for s in sparse_shape_xyz)
    nx = int(np.ceil(sx / wx) + 1)
    ny = int(np.ceil(sy / wy) + 1)
    nz = int(np.ceil(sz / wz) + 1)
    per_sample = nx * ny * nz
    if do_shift:
        shx, shy, shz = wx // 2, wy // 2, wz // 2
    else:
        shx, shy, shz = wx, wy, wz
    if sz == wz:  # swformer_utils.py:130-131
        shz = 0
    cx, cy, cz = coors[:, 3] + shx, coors[:, 2] + shy, coors[:, 1] + shz
    win = (
        coors[:, 0] * per_sample
        + torch.div(cx, wx, rounding_mode="floor") * (ny * nz)
        + torch.div(cy, wy, rounding_mode="floor") * nz
        + torch.div(cz, wz, rounding_mode="floor")
    )
    in_win = torch.stack([cz % wz, cy % wy, cx % wx], dim=-1)
    return win, in_win


def _rank(group):
    return torch.from_numpy(ingroup_rank(group.numpy()))


# --------------------------------------------------------------------------- a15
def batching_single_shift(win, batching_info):
    """batching_single_shift, point_transformer_layer.py:71-87."""
    rank = _rank(win)
    per_win = torch.bincount(win)[win]
    level = -torch.ones_like(win)
    cap = torch.zeros_like(win)
    for bl, info in batching_info.items():
        lo, hi = info["batching_range"]
        sel = (per_win >= lo) & (per_win < hi)
        cap[sel] = info["max_tokens"]
        level[sel] = bl
    return rank < cap, level


# --------------------------------------------------------------------------- a16
def continuous_inds(inds):
    """make_continuous_inds, swformer_utils.py:158-171: rank of each id among the sorted unique ids."""
    uniq, inv = torch.unique(inds, sorted=True, return_inverse=True)
    return inv


def flat2win_inds(win, level, batching_info):
    """get_flat2win_inds(_v2), swformer_utils.py:8-31,88-93."""
    out = {}
    for bl, info in batching_info.items():
        sel = level == bl
        if not sel.any():
            continue
        conti = continuous_inds(win[sel])
        slot = conti * info["max_tokens"] + _rank(conti)
        out[bl] = (slot, torch.where(sel))
    out["voxel_batching_level"] = level
    out["batching_info"] = batching_info
    return out


# --------------------------------------------------------------------------- a19
def flat2window(feat, inds):
    """flat2window(_v2), swformer_utils.py:34-64,101-105."""
    info, level = inds["batching_info"], inds["voxel_batching_level"]
    out = {}
    for bl in info:
        sel = level == bl
        if not sel.any():
            continue
        slot = inds[bl][0]
        t = info[bl]["max_tokens"]
        nwin = int(torch.div(slot, t, rounding_mode="floor").max()) + 1
        canvas = torch.zeros((nwin * t, feat.shape[-1]), dtype=feat.dtype)
        canvas[slot] = feat[sel]
        out[bl] = canvas.reshape(nwin, t, feat.shape[-1])
    return out


def window2flat(feat3d, inds):
    """window2flat(_v2), swformer_utils.py:67-85,96-98."""
    levels = [k for k in inds if not isinstance(k, str)]
    total = sum(inds[k][0].shape[0] for k in levels)
    first = feat3d[next(iter(feat3d))]
    flat = torch.zeros((total, first.shape[-1]), dtype=first.dtype)
    for bl, f in feat3d.items():
        slot, pos = inds[bl]
        flat[pos] = f.reshape(-1, f.shape[-1])[slot]
    return flat


# --------------------------------------------------------------------------- a17
def pos_embed_flat(coors_in_win, window_shape, feat_dim, pos_temperature=1000, dtype=torch.float32):
    """get_pos_embed up to (not including) the flat2window scatter,
    point_transformer_layer.py:151-203 (normalize_pos=False, 3-D window)."""
    wx, wy, wz = window_shape
    assert wz != 1 and len(window_shape) == 3
    z = coors_in_win[:, 0] - wz / 2
    y = coors_in_win[:, 1] - wy / 2
    x = coors_in_win[:, 2] - wx / 2
    plen = feat_dim // 3
    inv_freq = torch.arange(plen, dtype=torch.float32)
    inv_freq = pos_temperature ** (2 * torch.div(inv_freq, 2, rounding_mode="floor") / plen)

    def emb(v):
        e = v[:, None] / inv_freq[None, :]
        return torch.stack([e[:, ::2].sin(), e[:, 1::2].cos()], dim=-1).flatten(1)

    return torch.cat([emb(x), emb(y), emb(z)], dim=-1).to(dtype)


def key_padding_mask(inds):
    """get_key_padding_mask, point_transformer_layer.py:209-220 (True = padded slot)."""
    n = len(inds["voxel_batching_level"])
    d = flat2window(torch.ones((n, 1), dtype=torch.bool), inds)
    return {k: v.logical_not().squeeze(2) for k, v in d.items()}


def window_partition(coords_bzyx, batching_info, window_shape, sparse_shape_xyz, feat_dim,
                     pos_temperature=1000):
    """SparseWindowPartitionLayer.forward, point_transformer_layer.py:36-69.

    Voxel dropping (max_tokens < window occupancy, :131-137) changes the row
    count and breaks replace_feature at pointtransformer.py:193;
    it is unsupported here as in the build: asserted, not reproduced.
    """
    coors = coords_bzyx.long()
    info = {}
    for s in range(2):
        win, in_win = window_coords(coors, sparse_shape_xyz, window_shape, s == 1)
        keep, level = batching_single_shift(win, batching_info)
        assert bool(keep.all()), "voxel dropping is unsupported (SURVEY.md section 8 quirk 1)"
        inds = flat2win_inds(win, level, batching_info)
        pe = pos_embed_flat(in_win, window_shape, feat_dim, pos_temperature)
        info[f"batch_win_inds_shift{s}"] = win
        info[f"coors_in_win_shift{s}"] = in_win
        info[f"voxel_batching_level_shift{s}"] = level
        info[f"flat2win_inds_shift{s}"] = inds
        info[f"pos_flat_shift{s}"] = pe
        info[f"pos_dict_shift{s}"] = flat2window(pe, inds)
        info[f"key_mask_shift{s}"] = key_padding_mask(inds)
    return info


# --------------------------------------------------------------------------- a21
def cosine_attention(q_in, k_in, v_in, params, prefix, num_heads, key_padding, tau_min=0.01):
    """CosineMultiheadAttention.forward in eval mode, cosine_msa.py:434-501 ->
    cosine_multi_head_attention_forward :180-410 -> _scaled_cosine_attention :115-177.

    q_in,k_in,v_in: [T, W, C] (sequence-first, batch_first=False).  q_in is k_in
    but k_in is not v_in, so the packed in-projection takes the third branch
    (:58-63): three separate linears on W.chunk(3).  Returns [T, W, C].
    """
    w, b = params[prefix + "in_proj_weight"], params[prefix + "in_proj_bias"]
    t, nwin, c = q_in.shape
    dh = c // num_heads
    wq, wk, wv = w.chunk(3)
    bq, bk, bv = b.chunk(3)
    q = F.linear(q_in, wq, bq).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
    k = F.linear(k_in, wk, bk).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
    v = F.linear(v_in, wv, bv).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
    q = F.normalize(q, dim=2)
    k = F.normalize(k, dim=2)
    attn = torch.bmm(q, k.transpose(-2, -1)) / params[prefix + "tau"].clamp(min=tau_min)
    mask = torch.zeros((nwin, 1, 1, t), dtype=q.dtype)
    mask.masked_fill_(key_padding.view(nwin, 1, 1, t), float("-inf"))
    attn = attn + mask.expand(-1, num_heads, -1, -1).reshape(nwin * num_heads, 1, t)
    attn = torch.softmax(attn, dim=-1)
    out = torch.bmm(attn, v).transpose(0, 1).contiguous().view(t, nwin, c)
    return F.linear(out, params[prefix + "out_proj.weight"], params[prefix + "out_proj.bias"])


# --------------------------------------------------------------------------- a20
def window_attention(feat, pos_dict, inds, key_mask, params, prefix, num_heads):
    """WindowAttention.forward, point_transformer_layer.py:233-258."""
    feat3d = flat2window(feat, inds)
    out = {}
    for bl, f in feat3d.items():
        x = f.permute(1, 0, 2)
        qk = x + pos_dict[bl].permute(1, 0, 2)
        o = cosine_attention(qk, qk, x, params, prefix + "self_attn.", num_heads, key_mask[bl])
        out[bl] = o.permute(1, 0, 2)
    return window2flat(out, inds)


# --------------------------------------------------------------------------- a22
def encoder_layer(x, pos_dict, inds, key_mask, params, prefix, num_heads):
    """EncoderLayer.forward (eval: DropPath/Dropout are identity), point_transformer_layer.py:289-298."""
    c = x.shape[1]
    a = window_attention(x, pos_dict, inds, key_mask, params, prefix + "win_attn.", num_heads)
    x = x + F.layer_norm(a, (c,), params[prefix + "norm1.weight"], params[prefix + "norm1.bias"])
    h = F.linear(x, params[prefix + "mlp.fc1.weight"], params[prefix + "mlp.fc1.bias"])
    h = F.linear(F.gelu(h), params[prefix + "mlp.fc2.weight"], params[prefix + "mlp.fc2.bias"])
    return x + F.layer_norm(h, (c,), params[prefix + "norm2.weight"], params[prefix + "norm2.bias"])


# --------------------------------------------------------------------------- a23
def swformer_block(x, info, params, prefix, depth, num_heads):
    """SWFormerBlock.forward, point_transformer_layer.py:314-339: first depth//2 layers on shift 0."""
    for i in range(depth):
        s = 0 if i < int(depth / 2) else 1
        x = encoder_layer(
            x, info[f"pos_dict_shift{s}"], info[f"flat2win_inds_shift{s}"], info[f"key_mask_shift{s}"],
            params, f"{prefix}layers.{i}.", num_heads,
        )
    return x


def attention_core_flops(info, c):
    """4*C*sum_w n_w^2 per shift, for reporting."""
    out = []
    for s in range(2):
        cnt = torch.bincount(info[f"batch_win_inds_shift{s}"]).double()
        out.append(4.0 * c * float((cnt * cnt).sum()))
    return out


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("math",)]
