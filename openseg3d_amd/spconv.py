"""``spconv.pytorch``-shaped sparse-convolution modules on the HIP rulebook / gather-GEMM kernels.

Only the surface the reference touches is provided (call sites
seg3d/utils/spconv_utils.py:13-32, seg3d/models/backbones/pointtransformer.py:13-113,184-189):
``SparseConvTensor``, ``SubMConv3d``, ``SparseConv3d``, ``SparseInverseConv3d``,
``SparseSequential``, ``SparseModule``.  INTEGRATION.md 2.2 shows the two-line ``sys.modules`` registration that
makes the reference's model files import this module as ``spconv.pytorch`` unchanged.

Design differences from spconv (MI355X-first):
  * a resolution level (``SiteLevel``) owns its coordinate hash and its neighbour tables; tables
    are pure functions of the active-site set, so ``indice_key`` only decides *which level* an
    inverse convolution returns to -- the cache the key names in spconv falls out of the level
    object being shared by every tensor that lives on it;
  * tables are output-stationary ([27][rows], -1 = inactive): forward, dgrad and wgrad all read the
    same two tables, nothing is scattered with atomics;
  * strided-conv output sites come out in ascending (b, z, y, x) order (build-defined; per-point
    logits do not depend on it).
Parameters: ``weight`` [Cout, 3, 3, 3, Cin] (+ ``bias`` [Cout]); default init = kaiming-uniform
(a=sqrt(5)) over fan_in = 27*Cin, bias uniform(+-1/sqrt(fan_in)), as torch's conv layers do.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops


# Inference: conv -> BatchNorm(eval) -> (+ residual) -> ReLU as one launch (seg3d_spconv_fwd_act); 0 = separate passes.
FUSE_EVAL_BN = os.environ.get("SEG3D_FUSE_EVAL_BN", "1") != "0"
# rows of a submanifold table processed grouped by neighbour mask inside buckets of this many rows (0 = table order,
# -1 = one bucket); SiteLevel.mask_order
SUBM_ORDER_BUCKET = int(os.environ.get("SEG3D_SUBM_ORDER", "0"))


class SiteLevel:
    """Active sites of one resolution level and everything derived from them (built lazily, once)."""

    def __init__(self, coords, spatial_shape, batch_size):
        self.coords = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
        self.coords = self.coords.contiguous()
        self.shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self._hash = None
        self._subm = None
        self._down = None
        self._parity = None
        self._mask_order = None
        self._subm_plan = None
        self._coarse = None  # [(coords, shape)] of the following strided levels when seeded by seed_chain()
        self._offsets = False
        self.window_plans = {}  # SparseWindowPartitionLayer -> WindowPlan of this level (built once per forward)

    def sample_offsets(self):
        """Cumulative row count per sample as python ints when every sample's rows are contiguous and in sample
        order (always true for collated voxel lists and for the sorted coarse levels), else None.  One host sync
        per level for batch_size > 1, none for a single sample."""
        if self._offsets is False:
            m = self.coords.shape[0]
            if self.batch_size == 1:
                self._offsets = [m]
            else:
                b = self.coords[:, 0]
                cnt = torch.bincount(b, minlength=self.batch_size)
                ordered = bool((b[1:] >= b[:-1]).all()) if m > 1 else True
                self._offsets = torch.cumsum(cnt, 0).tolist() if ordered and cnt.numel() == self.batch_size else None
        return self._offsets

    @property
    def hash(self):
        if self._hash is None:
            self._hash = ops.CoordHash(self.coords, self.shape)
        return self._hash

    def subm(self):
        """[27, M] table of SubMConv3d(k=3, padding=1)."""
        if self._subm is None:
            self._subm = ops.rulebook_subm(self.hash)
        return self._subm

    def subm_plan(self):
        """Tile plan of the submanifold table (ops.ConvPlan; None when the tiled schedule is switched off): Morton-ordered
        128-row tiles, their distinct input rows and image slots.  Built once, shared by every SubMConv3d on this level,
        forward and input gradient (the transposed table is the same table with the offsets mirrored).  A plan changes the
        schedule, and for 32 / 48-channel outputs also the (fixed) summation order: see ops.sparse_conv."""
        if not ops.CONV_TILED or ops.CONV_PRECISION != "bf16x3" or self.coords.shape[0] == 0:
            return None
        if self._subm_plan is None:
            self._subm_plan = ops.ConvPlan(self.coords, self.subm())
        return self._subm_plan

    def mask_order(self):
        """Processing order of the submanifold table's rows (int32 [M]; None = table order): inside buckets of
        SUBM_ORDER_BUCKET consecutive rows -- spatial neighbours, whose gathered rows share cache lines -- the rows are
        grouped by their 27-bit neighbour mask.  The gather-GEMM skips an offset for a whole 32-row wave tile / 128-row
        workgroup tile only when none of its rows has a neighbour there: in table order a wave executes 1.4 (deep levels)
        to 3.2 (level 1) row-products per useful one on the headline scene, grouped 1.15 - 2.2.  Scheduling only: results
        do not depend on it."""
        if SUBM_ORDER_BUCKET == 0:
            return None
        if self._mask_order is None:
            valid = self.subm() >= 0
            m = valid.shape[1]
            bits = (valid.to(torch.int64) << torch.arange(27, device=valid.device, dtype=torch.int64)[:, None]).sum(0)
            if SUBM_ORDER_BUCKET > 0:
                bits = bits | ((torch.arange(m, device=valid.device, dtype=torch.int64) // SUBM_ORDER_BUCKET) << 27)
            self._mask_order = torch.argsort(bits, stable=True).to(torch.int32)
        return self._mask_order

    def parity_order(self):
        """Rows grouped by the parity of (z, y, x), original order kept inside a group (int32 [M]).  Under a
        k=3, s=2, p=1 conv the kernel offsets that can reach a fine site are fixed by its parity (1, 2, 4 or 8 of
        the 27), so tiles of same-parity rows of the inverse table skip the other offsets wholesale."""
        if self._parity is None:
            self._parity = ops.parity_order(self.coords)  # stable 3-bit device radix sort (seg3d_parity_order)
        return self._parity

    def seed_chain(self, levels):
        """Build the site lists of the next ``levels`` strided levels with one host read-back (ops.downsample_chain) and
        hand them to the down() calls that follow."""
        if self._down is None and self._coarse is None:
            chain = ops.downsample_chain(self.coords, self.batch_size, self.shape, levels)
            self._coarse = chain

    def down(self):
        """(coarse SiteLevel, nbr_fwd [27, M_coarse], nbr_inv [27, M]) of SparseConv3d(k=3, s=2, p=1)."""
        if self._down is None:
            if self._coarse:
                (co, shape_out), rest = self._coarse[0], self._coarse[1:]
            else:
                (co, shape_out), rest = ops.downsample_coords(self.coords, self.batch_size, self.shape), None
            fwd, inv = ops.rulebook_strided(self.hash, co)
            coarse = SiteLevel(co, shape_out, self.batch_size)
            coarse._coarse = rest or None
            self._down = (coarse, fwd, inv)
            self._coarse = None
        return self._down


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, _level=None, _indice_dict=None):
        self.features = features
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.level = _level if _level is not None else SiteLevel(indices, self.spatial_shape, batch_size)
        self.indices = self.level.coords
        self.indice_dict = _indice_dict if _indice_dict is not None else {}

    def replace_feature(self, new_features):
        return SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.level,
                                self.indice_dict)

    def on_level(self, features, level):
        return SparseConvTensor(features, level.coords, level.shape, self.batch_size, level, self.indice_dict)


class SparseModule(nn.Module):
    """Marker base class: SparseSequential hands these the SparseConvTensor itself."""


class _Conv3x3x3(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, bias=True,
                 indice_key=None):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        if ks != 3 or dilation != 1:
            raise NotImplementedError("only 3x3x3, dilation 1 is on the OpenSeg3D path")
        if out_channels % 16:
            raise NotImplementedError("output channels must be a multiple of 16 (MFMA 16x16 tiles)")
        # narrow inputs (the 6/8 raw point channels of the multi-sweep config) are zero-padded per call
        self._pad_in = (-in_channels) % 16
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.padding, self.indice_key = stride, padding, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, 3, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packed = None  # (weight version, packed tensor) for no-grad forwards
        self._folded = None  # inference: (key, pack of W * bn_scale, folded bias) -- see forward_bn_act
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = 27 * self.in_channels
        nn.init.kaiming_uniform_(self.weight.view(self.out_channels, -1), a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _packed_weight(self):
        """Forward-operand pack, cached while the weight is unchanged (weights are static in eval)."""
        w = self.weight
        key = (ops._stamp(w), w.data_ptr(), ops.CONV_PRECISION)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                self._packed = (key, ops.pack_weight(w, ops.PACK_FWD))
        return self._packed[1]

    def _apply_tables(self, feats, nbr, nbr_t, t_flags, order=None, order_t=None, plan=None, plan_t=None):
        if self._pad_in:
            feats = torch.nn.functional.pad(feats, (0, self._pad_in))
            weight = torch.nn.functional.pad(self.weight, (0, self._pad_in))
            return ops.sparse_conv(feats, weight, self.bias, nbr, nbr_t, t_flags, None, order, order_t, plan, plan_t)
        return ops.sparse_conv(feats, self.weight, self.bias, nbr, nbr_t, t_flags, self._packed_weight(), order, order_t,
                               plan, plan_t)

    def plans(self, x):
        """(tile plan of tables()'s nbr, of its nbr_t) or (None, None): see SiteLevel.subm_plan."""
        return None, None

    def tables(self, x):
        """(nbr [27, rows_out], nbr_t, pack flags of the transposed operand, row order, row order of nbr_t, output
        SiteLevel) of this convolution on the sites of ``x``."""
        raise NotImplementedError

    def forward(self, x):
        nbr, nbr_t, t_flags, order, order_t, level = self.tables(x)
        plan, plan_t = self.plans(x)
        return x.on_level(self._apply_tables(x.features, nbr, nbr_t, t_flags, order, order_t, plan, plan_t), level)

    def fusable_with(self, bn, x):
        """Inference only: conv -> BatchNorm(eval) -> (+ residual) -> ReLU can run as one launch."""
        return (FUSE_EVAL_BN and not torch.is_grad_enabled() and not bn.training and bn.track_running_stats and bn.affine
                and not self._pad_in and ops.CONV_PRECISION == "bf16x3" and x.features.is_cuda
                and (x.features.dtype == torch.float32 or (x.features.dtype == torch.bfloat16 and ops.conv_storage_bf16())))

    def forward_bn_act(self, x, bn, relu=True, res=None):
        """act(bn(conv(x)) (+ res)) with the BatchNorm's running-statistics affine folded into the packed weights and the
        bias (seg3d_spconv_fwd_act); the fold is cached until the weights or the statistics change."""
        scale, shift, bn_key = ops.bn_eval_affine(bn)
        w = self.weight
        key = (ops._stamp(w), w.data_ptr(), None if self.bias is None else self.bias._version, bn_key)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                wf = (w * scale.view(-1, 1, 1, 1, 1)).contiguous()
                bf = shift if self.bias is None else shift + self.bias * scale
                self._folded = (key, ops.pack_weight(wf, ops.PACK_FWD, use_registry=False), bf.contiguous())
        _, packed, bias = self._folded
        nbr, _, _, order, _, level = self.tables(x)
        y = ops.conv_act(x.features, nbr, packed, bias, self.in_channels, self.out_channels, order, addend=res, relu=relu,
                         plan=self.plans(x)[0])
        return x.on_level(y, level)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size=3, stride={self.stride}, " \
               f"padding={self.padding}, indice_key={self.indice_key!r}"


class SubMConv3d(_Conv3x3x3):
    def tables(self, x):
        nbr = x.level.subm()
        order = x.level.mask_order()  # (the transposed table is the same table with the offsets mirrored: same grouping)
        return nbr, nbr, ops.PACK_T_FLIP, order, order, x.level

    def plans(self, x):
        plan = x.level.subm_plan()
        return plan, plan


class SparseConv3d(_Conv3x3x3):
    def tables(self, x):
        if self.stride != 2 or self.padding != 1:
            raise NotImplementedError("only SparseConv3d(k=3, stride=2, padding=1) is on the OpenSeg3D path")
        coarse, fwd, inv = x.level.down()
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = x.level
        return fwd, inv, ops.PACK_T, None, x.level.parity_order(), coarse


class SparseInverseConv3d(_Conv3x3x3):
    def __init__(self, in_channels, out_channels, kernel_size=3, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    def tables(self, x):
        fine = x.indice_dict.get(self.indice_key)
        if fine is None:
            raise KeyError(f"SparseInverseConv3d: no strided conv registered indice_key={self.indice_key!r}")
        coarse, fwd, inv = fine.down()
        if coarse is not x.level:
            raise RuntimeError("SparseInverseConv3d input does not live on the paired strided conv's output sites")
        return inv, fwd, ops.PACK_T, fine.parity_order(), None, fine


class SparseSequential(SparseModule):
    """Runs SparseModules on the sparse tensor and plain nn.Modules (BN, ReLU, ...) on ``.features``."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, idx):  # (spconv's SparseSequential is indexable like nn.Sequential)
        if not -len(self) <= idx < len(self):
            raise IndexError(f"index {idx} is out of range")
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x
