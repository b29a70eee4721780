"""Sparse-window transformer stage (SWFormer) on the HIP window-partition / attention kernels.

Mirrors the module tree -- hence the state_dict keys -- of
seg3d/models/layers/point_transformer_layer.py (SparseWindowPartitionLayer, WindowAttention, MLP,
EncoderLayer, SWFormerBlock) and seg3d/models/layers/cosine_msa.py (CosineMultiheadAttention):
``layers.{i}.win_attn.self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias,tau}``,
``layers.{i}.norm{1,2}``, ``layers.{i}.mlp.fc{1,2}``.

What changed underneath (MI355X-first):
  * windows stay ragged: the partition emits a CSR of non-empty windows and the attention kernel
    consumes it directly; the reference's padded [W, T, C] tensors, their -inf masks and the
    flat2window / window2flat copies in every encoder layer (swformer_utils.py:34-85) do not exist;
  * one device pass per (stage, shift) builds every index with no host sync; the counts the host
    needs (windows per shift) are fetched once per stage;
  * attention-probability dropout (attn_drop = 0.1, training only) is applied inside the attention kernels from a
    counter-based mask; the backward regenerates it (csrc/attn_dropout.hpp);
  * activation checkpointing (point_transformer_layer.py:321-337) is not used: 288 GB of HBM holds
    the activations of a 180 k-point scene many times over.
  * in training an encoder layer is ONE autograd node (ops._EncoderLayerFn): in-projection, attention, out-projection,
    residual + DropPath + LayerNorm and the MLP run as a fixed sequence of library kernels, the residual-path
    gradients join in GEMM epilogues; only GELU and ``x + pos`` are torch elementwise kernels.
"""
import math
import os

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def window_geometry(sparse_shape_xyz, window_shape, do_shift):
    """Host-side constants of get_window_coors (swformer_utils.py:108-131): windows per axis and shift."""
    win = [int(w) for w in window_shape]
    if len(win) != 3:
        raise NotImplementedError("3-D windows only (config.py:68 WINDOW_SHAPE)")
    s = [float(v) for v in sparse_shape_xyz]
    nwin = [int(np.ceil(s[i] / win[i]) + 1) for i in range(3)]
    shift = [w // 2 for w in win] if do_shift else list(win)
    if s[2] == win[2]:
        shift[2] = 0
    return win, nwin, shift


class WindowPlan:
    """Everything one SWFormer stage needs about the active sites: per shift a WindowIndex (CSR of
    windows) and the positional embedding in flat voxel order."""

    def __init__(self, index, pos):
        self.index = index  # [shift] -> ops.WindowIndex
        self.pos = pos      # [shift] -> float32 [M, C]


class SparseWindowPartitionLayer(nn.Module):
    """Window grouping + positional embedding (no parameters).  Called with a sparse tensor it returns a
    ``voxel_info`` dict shaped like the reference's (features + plan) for SWFormerBlock."""

    def __init__(self, batching_info, window_shape, sparse_shape, normalize_pos=False, pos_temperature=1000):
        super().__init__()
        if normalize_pos:
            raise NotImplementedError("normalize_pos=True is never enabled on the reference path")
        self.batching_info = batching_info
        self.window_shape = [int(w) for w in window_shape]
        self.sparse_shape = [float(s) for s in sparse_shape]  # (x, y, z), may be fractional
        self.pos_temperature = pos_temperature
        self._inv_freq = {}

    def levels(self):
        return [(v["batching_range"][0], v["batching_range"][1], v["max_tokens"])
                for _, v in sorted(self.batching_info.items())]

    def inv_freq(self, feat_dim, device):
        """pos_temperature ** (2 * (j // 2) / pos_length), evaluated by torch exactly as at
        point_transformer_layer.py:180-184 so the divisor is bit-identical."""
        key = (feat_dim, str(device))
        if key not in self._inv_freq:
            plen = feat_dim // 3
            t = torch.arange(plen, dtype=torch.float32)
            t = self.pos_temperature ** (2 * torch.div(t, 2, rounding_mode="floor") / plen)
            self._inv_freq[key] = t.to(device)
        return self._inv_freq[key]

    @torch.no_grad()
    def launch_plan(self, coords, batch_size, feat_dim, want_debug=False):
        """Queue the partition and positional-embedding kernels of both shifts; nothing is read back.  finish_plans()
        completes one or several launched plans with a single host read-back."""
        if feat_dim % 3 or (feat_dim // 3) % 2:
            raise ValueError("feature dim must be divisible by 6 (point_transformer_layer.py:176,196)")
        index, pos = [], []
        for s in range(2):
            win, nwin, shift = window_geometry(self.sparse_shape, self.window_shape, s == 1)
            wi = ops.window_partition(coords, batch_size, win, nwin, shift, self.levels(), want_debug=want_debug)
            index.append(wi)
            pos.append(ops.pos_embed(wi.in_win, win, self.inv_freq(feat_dim, coords.device), feat_dim))
        return WindowPlan(index, pos)

    @staticmethod
    @torch.no_grad()
    def finish_plans(plans):
        """One host sync for the window / tile counts of all given plans (every stage, both shifts)."""
        index = [wi for plan in plans for wi in plan.index]
        counts = torch.stack([wi.counts for wi in index]).tolist()
        for wi, (n_win, n_drop, n_tiles, n_qg) in zip(index, counts):
            wi.n_windows, wi.n_dropped, wi.n_tiles, wi.n_qgroups = int(n_win), int(n_drop), int(n_tiles), int(n_qg)
            if n_drop:
                raise RuntimeError(
                    f"{n_drop} voxels exceed max_tokens of their batching level: voxel dropping is "
                    "unsupported (it breaks replace_feature in the reference too, SURVEY.md 8 quirk 1)")
        return plans

    def plan(self, coords, batch_size, feat_dim, want_debug=False):
        return self.finish_plans([self.launch_plan(coords, batch_size, feat_dim, want_debug)])[0]

    def plan_level(self, level, feat_dim):
        """The plan of a spconv.SiteLevel, cached on the level: the backbone asks for every stage's plan up front
        (segformer.PointTransformer.prepare) so that the host syncs happen before any feature kernel is queued."""
        plan = level.window_plans.get(self)
        if plan is None:
            plan = level.window_plans[self] = self.plan(level.coords, level.batch_size, feat_dim)
        return plan

    def forward(self, x):
        feats = x.features
        return {"voxel_features": feats, "plan": self.plan_level(x.level, feats.shape[1])}


class CosineMultiheadAttention(nn.Module):
    """Parameters of cosine_msa.CosineMultiheadAttention (packed in-proj, out-proj, shared tau); the
    attention core runs in seg3d_window_attn_fwd/bwd."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, tau_min=0.01):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.tau_min, self.dropout = embed_dim, num_heads, tau_min, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.tau = nn.Parameter(torch.ones(1, 1, 1))
        nn.init.xavier_uniform_(self.in_proj_weight)  # nn.MultiheadAttention._reset_parameters
        nn.init.constant_(self.out_proj.bias, 0.0)

    def drop_args(self, seed):
        """(p, seed) of the attention-probability dropout: F.dropout on the softmax output in training mode only
        (cosine_msa.py:172-174, 394-395).  The mask is a pure function of (seed, window, head, query, key), regenerated by
        the backward kernels; ``seed`` comes from torch's CPU generator (SWFormerBlock.forward), so torch.manual_seed
        makes a step reproducible."""
        if self.training and self.dropout > 0.0:
            return float(self.dropout), int(seed)
        return 0.0, 0

    def forward(self, x, pos, wi, seed=0):
        """x [M, C] flat voxel features, pos [M, C]; q = k = (x + pos) W_qk, v = x W_v (cosine_msa.py:58-63)."""
        qk, v = ops.attn_in_proj(x, pos, self.in_proj_weight, self.in_proj_bias)
        drop_p, drop_seed = self.drop_args(seed)
        o = ops.window_attention_packed(qk, v, self.tau, self.tau_min, self.num_heads, wi, drop_p, drop_seed)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias)


class WindowAttention(nn.Module):
    def __init__(self, d_model, nhead, attn_drop, tau_min=0.01):
        super().__init__()
        self.self_attn = CosineMultiheadAttention(d_model, nhead, dropout=attn_drop, tau_min=tau_min)

    def forward(self, feat_2d, pos, wi, seed=0):
        return self.self_attn(feat_2d, pos, wi, seed)


class MLP(nn.Module):
    def __init__(self, in_features, hidden_features, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = drop

    def forward(self, x):
        x = F.gelu(ops.linear(x, self.fc1.weight, self.fc1.bias))
        if self.drop and self.training:
            x = F.dropout(x, self.drop)
        x = ops.linear(x, self.fc2.weight, self.fc2.bias)
        if self.drop and self.training:
            x = F.dropout(x, self.drop)
        return x


def attention_dropout_seed():
    """63-bit seed from torch's CPU generator: reproducible under torch.manual_seed, never touches the device."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def drop_path_scale(x, p):
    """Per-row stochastic-depth factor mask / keep_prob, [rows] (seg3d/models/layers/drop.py:6-19)."""
    keep = 1.0 - p
    mask = x.new_empty((x.shape[0],)).bernoulli_(keep)
    if keep > 0.0:
        mask.div_(keep)
    return mask


def drop_path(x, p, training):
    """Per-row stochastic depth (seg3d/models/layers/drop.py:6-19)."""
    if p == 0.0 or not training:
        return x
    return x * drop_path_scale(x, p).reshape((x.shape[0],) + (1,) * (x.dim() - 1))


# One autograd node per encoder layer in training (ops._EncoderLayerFn); 0 = compose the layer from its modules.
FUSED_LAYER = os.environ.get("SEG3D_FUSED_LAYER", "1") != "0"


class EncoderLayer(nn.Module):
    """Post-norm encoder layer: x + DP(LN1(attn(x))), then + DP(LN2(mlp(.))) (point_transformer_layer.py:289-298)."""

    def __init__(self, d_model, nhead, mlp_hidden_dim, drop=0.0, attn_drop=0.1, drop_path_rate=0.0):
        super().__init__()
        self.win_attn = WindowAttention(d_model, nhead, attn_drop)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.mlp = MLP(d_model, mlp_hidden_dim, drop=drop)
        self.drop_path_rate = float(drop_path_rate)

    def forward(self, x, pos, wi, scales=None, seed=None):
        """``scales``: optional (s1, s2) per-row DropPath factors made by the enclosing block for all its layers at once;
        None = draw them here (seg3d/models/layers/drop.py:6-19).  ``seed``: attention-dropout seed of this layer
        (None = draw one from torch's CPU generator)."""
        at, mlp = self.win_attn.self_attn, self.mlp
        if seed is None:
            seed = attention_dropout_seed() if self.training and at.dropout > 0.0 else 0
        drop = self.training and self.drop_path_rate > 0.0
        s1 = s2 = None
        if drop:
            s1, s2 = scales if scales is not None else (drop_path_scale(x, self.drop_path_rate),
                                                        drop_path_scale(x, self.drop_path_rate))
        if (FUSED_LAYER and torch.is_grad_enabled() and x.requires_grad and not mlp.drop and at.in_proj_bias is not None
                and ops.encoder_layer_fits(x, at.embed_dim, mlp.fc1.out_features, at.num_heads)):
            # training: the whole layer is one autograd node (ops._EncoderLayerFn)
            meta = (at.num_heads, at.tau_min, wi, self.norm1.eps, self.norm2.eps, s1, s2) + at.drop_args(seed)
            return ops._EncoderLayerFn.apply(x, pos, at.in_proj_weight, at.in_proj_bias, at.tau, at.out_proj.weight,
                                             at.out_proj.bias, self.norm1.weight, self.norm1.bias, mlp.fc1.weight,
                                             mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, self.norm2.weight,
                                             self.norm2.bias, meta)
        a = self.win_attn(x, pos, wi, seed)
        # fused residual + LayerNorm; stochastic depth rides in the same pass: x + mask/keep * LN(.)
        x = ops.layer_norm_residual(a, x, self.norm1, rowscale=s1)
        return ops.layer_norm_residual(self.mlp(x), x, self.norm2, rowscale=s2)


class SWFormerBlock(nn.Module):
    def __init__(self, d_model, nhead, depth=4, mlp_ratio=2.0, attn_drop=0.1, drop=0.0, drop_path=0.0):
        super().__init__()
        self.depth = depth
        rates = drop_path if isinstance(drop_path, (list, tuple)) else [drop_path] * depth
        self.layers = nn.ModuleList(
            EncoderLayer(d_model, nhead, int(d_model * mlp_ratio), drop=drop, attn_drop=attn_drop,
                         drop_path_rate=rates[i]) for i in range(depth))
        self._keep = None  # [2 * depth, 1] keep probabilities on the device (built on first use)

    def forward(self, voxel_info):
        x, plan = voxel_info["voxel_features"], voxel_info["plan"]
        if x.dtype == torch.bfloat16:  # a sparse-conv feature map of the bf16 storage mode: the residual stream is float32
            x = x.float()
        half = int(self.depth / 2)  # first depth//2 layers on the unshifted windows (:321-337)
        scales = self.drop_path_scales(x) if self.training else None
        # one draw from the CPU generator per block and forward (no device round trip); layer i uses seed + i
        seed = attention_dropout_seed() if self.training else 0
        for i, layer in enumerate(self.layers):
            s = 0 if i < half else 1
            x = layer(x, plan.pos[s], plan.index[s], None if scales is None else (scales[2 * i], scales[2 * i + 1]), seed + i)
        return x

    def drop_path_scales(self, x):
        """[2 * depth, rows] DropPath factors mask / keep_prob of all layers of the block from ONE uniform draw (4 launches
        per block instead of 4 per layer); same distribution as drop.py:6-19, rows of rate-0 layers are never read."""
        rates = [layer.drop_path_rate for layer in self.layers]
        if not any(r > 0.0 for r in rates):
            return None
        if self._keep is None or self._keep.device != x.device:
            self._keep = torch.tensor([1.0 - r for r in rates for _ in (0, 1)], dtype=torch.float32, device=x.device)[:, None]
        u = torch.rand((2 * self.depth, x.shape[0]), dtype=torch.float32, device=x.device)
        return (u < self._keep).to(torch.float32).div_(self._keep)
