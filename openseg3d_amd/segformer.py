"""Segformer / PointTransformer segmentor on the HIP path.

Host-side mirror of the reference's default model (``cfg.MODEL.SEGMENTOR == 'segformer'``):
  seg3d/models/segmentors/segformer.py:12-146, seg3d/models/backbones/pointtransformer.py:13-219,
  seg3d/models/voxel_encoders/vfe.py, seg3d/models/layers/se_layer.py.
Same constructor arguments, same ``forward(batch_dict) -> OrderedDict`` contract and the same
state_dict keys / shapes (tests/golden/segformer_keys.json), so reference checkpoints load.

What runs where: voxel-feature reduce, every sparse convolution, window partition + attention, every Linear /
LayerNorm / BatchNorm(+ReLU) pass with MFMA-sized channels, the voxel->point gather, kNN and the training criterion run
in libseg3d_hip.so; torch supplies GELU, a few elementwise adds / concatenations, the 6 -> 64 and -> 22 GEMMs (rocBLAS),
the optimizer and autograd's bookkeeping.  Index structures (site levels, neighbour tables, window CSRs, point->voxel
CSR) are built once per batch, before the first feature kernel, and shared by all layers that use them.
Scope: all three shipped configs -- single sweep (cartesian / cylinder) and multi-sweep with optional
image-feature fusion (DeepFusionBlock over seg3d_knn_query).
"""
import os
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import ops
from . import spconv
from .swformer import SparseWindowPartitionLayer, SWFormerBlock


# Build every index structure of a forward before its first feature kernel (PointTransformer.prepare); 0 restores the
# lazy per-stage order for A/B timing.
PLAN_FIRST = os.environ.get("SEG3D_PLAN_FIRST", "1") != "0"


def replace_feature(out, new_features):
    return out.replace_feature(new_features)


def conv_module(cin, cout, norm_fn, act_fn, conv_type="subm", indice_key=None):
    """conv (no bias) + BN + ReLU with the reference's child names '0','1','2' (spconv_utils.py:13-32)."""
    if conv_type == "subm":
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key=indice_key)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv3d(cin, cout, 3, bias=False, indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return ConvBnAct(conv, norm_fn(cout), act_fn)


class ConvBnAct(spconv.SparseSequential):
    """conv -> BatchNorm1d -> ReLU with the reference's child names; BN + ReLU run as one fused pass."""

    def forward(self, x):
        conv, bn = self._modules["0"], self._modules["1"]
        if conv.fusable_with(bn, x):  # inference: one launch, BatchNorm folded into the packed weights
            return conv.forward_bn_act(x, bn, relu=True)
        y = conv(x)
        return y.replace_feature(ops.batch_norm_act(y.features, bn, relu=True))


class FusedMLP(nn.Sequential):
    """nn.Sequential of Linear / BatchNorm1d / ReLU / Dropout in which every (BatchNorm1d, ReLU) pair runs as
    one fused pass (same modules, same state_dict keys)."""

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            # eval: Linear -> BatchNorm1d (-> ReLU) as one launch (the BatchNorm's folded affine in the Linear's epilogue)
            if isinstance(m, RowLinear) and i + 1 < len(mods) and not self.training \
                    and isinstance(mods[i + 1], nn.BatchNorm1d) and not isinstance(mods[i + 1], NarrowBatchNorm1d):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                y = ops.linear_bn_act_eval(x, m, mods[i + 1], relu)
                if y is not None:
                    x = y
                    i += 3 if relu else 2
                    continue
            # SyncBatchNorm (convert_sync_batchnorm, tools/train.py:246-247) takes the same fused pass: ops.batch_norm_act
            # synchronises its statistics over the ranks; 6 / 8-channel inputs do not fit it and stay on torch's module
            if isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)) and not isinstance(m, NarrowBatchNorm1d) \
                    and m.num_features % 4 == 0:
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.batch_norm_act(x, m, relu=relu)
                i += 2 if relu else 1
            else:
                x = m(x)
                i += 1
        return x


class VFE(nn.Module):
    """Per-voxel reduce of point features (vfe.py:16-27); rows with id -1 are skipped by the CSR."""

    def __init__(self, voxel_feature_channel, reduce="mean"):
        super().__init__()
        self.voxel_feature_channel = voxel_feature_channel
        self.reduce = reduce

    def forward(self, features, seg):
        return ops.segment_reduce(features, seg, {"mean": ops.REDUCE_MEAN, "max": ops.REDUCE_MAX}[self.reduce])


class FlattenSELayer(nn.Module):
    """Squeeze-excite over each sample's rows (se_layer.py:16-29)."""

    def __init__(self, channel, reduction=4):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def forward(self, x, indices, row_offsets=None):
        """``row_offsets`` (python ints, cumulative rows per sample) lets the mean run on contiguous slices --
        samples are contiguous after collate_batch -- instead of a scatter over <= batch_size huge segments."""
        if row_offsets is not None:
            starts = [0] + list(row_offsets[:-1])
            spans = [(s, e) for s, e in zip(starts, row_offsets)]
            pooled = ops.span_mean(x, row_offsets)
            gate = self.fc(pooled)
            # broadcast per contiguous sample: the backward is a row sum, not an index_put over N rows
            return torch.cat([x[s:e] * gate[b] for b, (s, e) in enumerate(spans)], dim=0)
        indices = indices.long()
        pooled = ops.scatter(x, indices, reduce="mean")
        return x * self.fc(pooled)[indices]


class SparseBasicBlock(spconv.SparseModule):
    def __init__(self, inplanes, planes, norm_fn, act_fn, indice_key=None):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(inplanes, planes, 3, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.act = act_fn
        self.conv2 = spconv.SubMConv3d(planes, planes, 3, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)

    def forward(self, x):
        if self.conv1.fusable_with(self.bn1, x) and self.conv2.fusable_with(self.bn2, x):  # inference: two launches
            y = self.conv1.forward_bn_act(x, self.bn1, relu=True)
            return self.conv2.forward_bn_act(y, self.bn2, relu=True, res=x.features)
        y = self.conv1(x)
        y = y.replace_feature(ops.batch_norm_act(y.features, self.bn1, relu=True))
        y = self.conv2(y)
        return y.replace_feature(ops.batch_norm_act(y.features, self.bn2, relu=True, res=x.features))


class UpBlock(spconv.SparseModule):
    def __init__(self, inplanes, planes, norm_fn, act_fn, conv_type, layer_id):
        super().__init__()
        self.transform = SparseBasicBlock(inplanes, inplanes, norm_fn, act_fn, indice_key=f"subm{layer_id}")
        self.bottleneck = conv_module(2 * inplanes, inplanes, norm_fn, act_fn, "subm", f"subm{layer_id}")
        key = f"spconv{layer_id}" if conv_type == "inverseconv" else f"subm{layer_id}"
        self.out = conv_module(inplanes, planes, norm_fn, act_fn, conv_type, key)

    def forward(self, x_bottom, x_lateral):
        t = self.transform(x_lateral)
        cat = torch.cat([x_bottom.features, t.features], dim=1)
        m = self.bottleneck(t.replace_feature(cat)).features
        # channel_reduction (pointtransformer.py:88-102): sum adjacent channel pairs of the concat -- as an elementwise add of
        # the two strided halves (torch's generic reduce kernel over a size-2 dimension took 66 us per call, forward and backward)
        if cat.shape[1] == 2 * m.shape[1]:
            skip = cat[:, 0::2] + cat[:, 1::2]
        else:
            skip = cat.view(cat.shape[0], m.shape[1], -1).sum(dim=2)
        return self.out(t.replace_feature(m + skip))


class PointTransformer(nn.Module):
    def __init__(self, input_channels, output_channels, grid_size, voxel_size, point_cloud_range, batching_info,
                 window_shape, drop_path_rate, depths, num_classes):
        super().__init__()
        self.sparse_shape = [int(g) for g in grid_size][::-1]  # (z, y, x)
        self.voxel_size, self.point_cloud_range = voxel_size, point_cloud_range
        self.depths = list(depths)
        grid_xyz = [float(g) for g in grid_size]
        self.norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.act_fn = nn.ReLU(inplace=True)
        norm_fn, act_fn = self.norm_fn, self.act_fn

        self.conv_input = ConvBnAct(
            spconv.SubMConv3d(input_channels, 48, 3, padding=1, bias=False, indice_key="subm1"), norm_fn(48), act_fn)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.depths))]
        widths = (48, 96, 192, 384)
        for k, c in enumerate(widths):
            lo, hi = sum(self.depths[:k]), sum(self.depths[:k + 1])
            stage = nn.Sequential(
                SparseWindowPartitionLayer(batching_info[k], window_shape, [g / (2 ** k) for g in grid_xyz]),
                SWFormerBlock(c, 8, depth=self.depths[k], drop_path=dpr[lo:hi]))
            setattr(self, f"swformer_block{k + 1}", stage)
        for k in (1, 2, 3):
            setattr(self, f"conv_down{k}", conv_module(widths[k - 1], widths[k], norm_fn, act_fn, "spconv",
                                                       f"spconv{k + 1}"))
        self.up4 = UpBlock(384, 192, norm_fn, act_fn, "inverseconv", 4)
        self.up3 = UpBlock(192, 96, norm_fn, act_fn, "inverseconv", 3)
        self.up2 = UpBlock(96, 48, norm_fn, act_fn, "inverseconv", 2)
        self.up1 = UpBlock(48, output_channels, norm_fn, act_fn, "subm", 1)
        self.aux_voxel_classifier = nn.Sequential(RowLinear(384, num_classes, bias=False))
        self.voxel_classifier = nn.Sequential(RowLinear(output_channels, num_classes, bias=False))

    def prepare(self, batch_dict):
        """Index plan of the whole backbone -- site levels, rulebooks, row orders, window plans of the four stages --
        built before any feature kernel is queued.  Both host read-backs of a forward (the sizes of the three strided
        levels, then the window counts of all four stages) happen here, while only short index kernels are in flight; afterwards the feature pipeline (forward,
        loss, backward) is enqueued without a single wait, so the host runs ahead of the GPU instead of draining the
        queue once per stage."""
        level = spconv.SiteLevel(batch_dict["voxel_coords"].int(), self.sparse_shape, batch_dict["batch_size"])
        batch_dict["site_level"] = level
        widths = (48, 96, 192, 384)
        level.seed_chain(3)  # sites of the three strided levels: chained on the device, one read-back
        launched = []
        for k in range(4):
            level.subm()
            level.mask_order()
            level.subm_plan()
            part = getattr(self, f"swformer_block{k + 1}")[0]
            if part not in level.window_plans:  # partition kernels queued now, counts of all stages read back once below
                level.window_plans[part] = part.launch_plan(level.coords, level.batch_size, widths[k])
                launched.append(level.window_plans[part])
            if k < 3:
                level.parity_order()
                level = level.down()[0]
        if launched:
            SparseWindowPartitionLayer.finish_plans(launched)
        return batch_dict

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict["voxel_features"], indices=batch_dict["voxel_coords"].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"],
                                    _level=batch_dict.get("site_level"))
        x1 = self.conv_input(x)
        x1 = x1.replace_feature(self.swformer_block1(x1))
        x2 = self.conv_down1(x1)
        x2 = x2.replace_feature(self.swformer_block2(x2))
        x3 = self.conv_down2(x2)
        x3 = x3.replace_feature(self.swformer_block3(x3))
        x4 = self.conv_down3(x3)
        x4 = x4.replace_feature(self.swformer_block4(x4))

        batch_dict["aux_voxel_out"] = self.aux_voxel_classifier(x4.features)
        batch_dict["aux_voxel_coords"] = x4.indices

        y = self.up4(x4, x4)
        y = self.up3(y, x3)
        y = self.up2(y, x2)
        y = self.up1(y, x1)
        batch_dict["voxel_features"] = y.features
        batch_dict["voxel_coords"] = y.indices
        batch_dict["voxel_out"] = self.voxel_classifier(y.features)
        return batch_dict


class NarrowBatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d for very narrow inputs (the 6/8 raw point channels, segformer.py:22).  Same parameters, buffers and
    arithmetic as nn.BatchNorm1d.  torch's column reductions over a [175k, 6] tensor take 0.2-0.4 ms each (and its generic
    batch-norm backward ~7 ms): in training the rows are zero-padded to a multiple of 4 channels and go through the
    library's BatchNorm passes (ops.narrow_batch_norm)."""

    def forward(self, x):
        if not self.training or not self.track_running_stats:
            return super().forward(x)
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.affine and x.shape[0] >= 2:
            return ops.narrow_batch_norm(x, self)
        var, mean = torch.var_mean(x, dim=0, unbiased=False)
        with torch.no_grad():
            n = x.shape[0]
            self.num_batches_tracked += 1
            mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            self.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            self.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
        y = (x - mean) * torch.rsqrt(var + self.eps)
        return y * self.weight + self.bias if self.affine else y


class RowLinear(nn.Linear):
    """nn.Linear on [rows, C] activations whose weight gradient runs in libseg3d_hip.so (ops.linear)."""

    def forward(self, x):
        if x.dtype == torch.bfloat16:  # a sparse-conv feature map of the bf16 storage mode: arithmetic stays float32
            x = x.float()
        return ops.linear(x, self.weight, self.bias, exact=True)


def _bn_mlp(dims, first_bn=None, last_plain=False):
    """[BN(d0)] + (Linear(no bias) + BN + ReLU)* [+ Linear(bias)] with the reference's Sequential numbering."""
    mods = [NarrowBatchNorm1d(first_bn)] if first_bn is not None else []
    n = len(dims) - 1
    for i in range(n):
        if last_plain and i == n - 1:
            mods.append(RowLinear(dims[i], dims[i + 1]))
        else:
            mods += [RowLinear(dims[i], dims[i + 1], bias=False), nn.BatchNorm1d(dims[i + 1]), nn.ReLU(inplace=True)]
    return FusedMLP(*mods)


class DeepFusionBlock(nn.Module):
    """Point <-> image-feature cross attention over the k nearest current-sweep points
    (seg3d/models/layers/deep_fusion.py:10-45).  Neighbour search = seg3d_knn_query; the reference hands
    ``knn_query`` its [N, 6|8] point rows although the kernel strides by 3 floats: with
    ``faithful_stride=True`` (default) the same buffer is handed over, so outputs match the reference bit for
    bit in the neighbour sets; ``False`` searches on the xyz columns, which is what the code presumably meant."""

    def __init__(self, lidar_channel, image_channel, hidden_channel, n_neighbors, attn_pdrop=0.3, faithful_stride=True):
        super().__init__()
        self.n_neighbors, self.faithful_stride = n_neighbors, faithful_stride
        self.q_embedding = nn.Linear(lidar_channel, hidden_channel)
        self.k_embedding = nn.Linear(image_channel, hidden_channel)
        self.v_embedding = nn.Linear(image_channel, hidden_channel)
        self.attn_dropout = nn.Dropout(attn_pdrop)
        self.c_proj = nn.Linear(hidden_channel, image_channel)

    def neighbours(self, points, point_id_offset):
        """int32 [N, k] rows of the k nearest current-sweep points (deep_fusion.py:31).  Depends on the points alone: the
        batch builder / input pipeline may compute it ahead of the forward (Segformer.prepare_batch)."""
        xyz = points.contiguous() if self.faithful_stride else points[:, :3].contiguous()
        knn_ids, _ = ops.knn_query(self.n_neighbors, xyz, xyz, point_id_offset, point_id_offset)
        return knn_ids

    def forward(self, points, point_id_offset, lidar_features, image_features, knn_ids=None):
        q = self.q_embedding(lidar_features)
        k = self.k_embedding(image_features)
        v = self.v_embedding(image_features)
        if knn_ids is None:
            knn_ids = self.neighbours(points, point_id_offset)
        invalid = image_features.sum(dim=1) == 0
        if q.is_cuda and q.shape[1] == 32 and self.n_neighbors <= 16:
            # one kernel each way: dot products, masked softmax, dropout factors and the weighted sum over the 16 neighbour
            # rows, nothing of size [N, 16, 32] materialised (seg3d_knn_attention_fwd / _bwd)
            keep = None
            if self.training and self.attn_dropout.p > 0.0:  # F.dropout's factors, drawn by torch as the reference does
                keep = self.attn_dropout(torch.ones(knn_ids.shape, dtype=torch.float32, device=q.device))
            return self.c_proj(ops.knn_attention(q, k, v, knn_ids, invalid, keep))
        knn_ids = knn_ids.long()
        attn = (q.unsqueeze(1) * k[knn_ids]).sum(dim=-1) / (q.shape[-1] ** 0.5)
        attn = attn.masked_fill(invalid[knn_ids], float("-inf"))
        attn = torch.nan_to_num(torch.softmax(attn, dim=-1))
        attn = self.attn_dropout(attn)
        return self.c_proj((attn.unsqueeze(-1) * v[knn_ids]).sum(dim=1))


class Segformer(nn.Module):
    def __init__(self, dataset, batching_info, window_shape, depths, drop_path_rate):
        super().__init__()
        dim_point = dataset.dim_point + (2 if dataset.use_cylinder else 0)
        self.use_multi_sweeps = bool(dataset.use_multi_sweeps)
        self.use_image_feature = bool(dataset.use_image_feature)
        self.point_feature_channel = 64
        self.point_encoder = _bn_mlp([dim_point, 64, 128, 256, self.point_feature_channel], first_bn=dim_point,
                                     last_plain=True)
        # multi-sweep: the voxel features are the MEAN of the raw rows of all sweeps (segformer.py:34-37,105-107)
        self.vfe = VFE(dim_point, reduce="mean") if self.use_multi_sweeps else VFE(self.point_feature_channel, "max")
        self.scatter = VFE(3, reduce="mean")  # present (and unused) in the reference: segformer.py:39
        self.voxel_feature_channel = 32
        self.point_transformer = PointTransformer(
            self.vfe.voxel_feature_channel, self.voxel_feature_channel, dataset.grid_size, dataset.voxel_size,
            dataset.point_cloud_range, batching_info=batching_info, window_shape=window_shape, depths=depths,
            drop_path_rate=drop_path_rate, num_classes=dataset.num_classes)
        self.image_feature_channel = dataset.dim_image_feature if self.use_image_feature else 0
        if self.use_image_feature:
            self.deep_fusion = DeepFusionBlock(self.point_feature_channel + self.voxel_feature_channel,
                                               self.image_feature_channel, 32, 16)
        self.fusion_feature_channel = 64
        self.fusion_encoder = _bn_mlp([self.point_feature_channel + self.voxel_feature_channel
                                       + self.image_feature_channel, 256, 128, self.fusion_feature_channel])
        self.se = FlattenSELayer(self.fusion_feature_channel)
        self.classifier = FusedMLP(RowLinear(self.fusion_feature_channel, 64, bias=False), nn.BatchNorm1d(64),
                                   nn.ReLU(True), nn.Dropout(0.3),
                                   RowLinear(64, dataset.num_classes, bias=False))
        self.weight_initialization()

    def weight_initialization(self):
        """segformer.py:78-92: kaiming-normal Linear weights, unit norms; sparse convs keep their default init."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm1d, nn.LayerNorm)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def prepare_batch(self, batch_dict):
        """The forward's index plan (site levels, rulebooks, window partitions: everything that depends on the voxel
        coordinates only, and all of the forward's host read-backs), built ahead of time and carried by the batch -- an
        input pipeline calls this for batch i+1 on its own stream while batch i trains (bench.py)."""
        if "site_level" not in batch_dict:
            self.point_transformer.prepare(batch_dict)
        if self.use_image_feature and "fusion_knn" not in batch_dict:
            # DeepFusionBlock's neighbour search reads the points only (13 ms of a 63 ms multi-sweep forward): part of the plan
            cur_points, _ = self._current_sweep(batch_dict)
            batch_dict["fusion_knn"] = self.deep_fusion.neighbours(cur_points, batch_dict["point_id_offset"].int())
        return batch_dict

    def _current_sweep(self, batch_dict):
        """(rows of the current sweep without the batch column, their row numbers or None) -- segformer.py:96-100; the row
        numbers are found once per batch (torch.nonzero reads a count back) and travel with it."""
        points = batch_dict["points"][:, 1:]
        if not self.use_multi_sweeps:
            return points, None
        cur_rows = batch_dict.get("cur_rows")
        if cur_rows is None:
            cur_rows = batch_dict["cur_rows"] = torch.nonzero(points[:, 3] == 0).view(-1)  # time lag column == 0
        return points[cur_rows], cur_rows

    def forward(self, batch_dict):
        if self.training and torch.is_grad_enabled():
            ops.probe_deferred_join(batch_dict["points"].device)  # once per process: self-test of the deferred join
        with ops.deferred_bn_counters():  # one launch for all BatchNorm step counters
            return self._forward(batch_dict)

    def _forward(self, batch_dict):
        points = batch_dict["points"][:, 1:]
        ids = batch_dict["point_voxel_ids"]
        n_voxels = batch_dict["voxel_coords"].shape[0]
        # point <-> voxel CSR over ALL rows: built once, used by the VFE reduce (and the gather's backward)
        seg = batch_dict.get("point_voxel_index")
        if seg is None:
            seg = ops.SegmentIndex(ids, n_voxels)
        if PLAN_FIRST and "site_level" not in batch_dict:  # (a batch may arrive with its plan: prepare_batch below)
            # (Running the plan on a second stream under the point encoder changed nothing: 48.4 vs 48.5 ms.)
            self.point_transformer.prepare(batch_dict)

        if self.use_multi_sweeps:
            cur_points, cur_rows = self._current_sweep(batch_dict)  # rows whose time lag column == 0 (segformer.py:98)
            cur_ids = seg.ids[cur_rows]
            cur_seg = None
            batch_rows = batch_dict["points"][cur_rows, 0]
        else:
            cur_points, cur_ids, cur_seg, batch_rows = points, seg.ids, seg, batch_dict["points"][:, 0]
        knn_ids, knn_done = batch_dict.get("fusion_knn"), None
        if self.use_image_feature and knn_ids is None and points.is_cuda:
            # not planned ahead: the search depends on the points alone, so it runs on the second stream beside the point
            # encoder and the whole backbone and is awaited where DeepFusion needs it
            main, side = torch.cuda.current_stream(points.device), ops.side_stream(points.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                knn_ids = self.deep_fusion.neighbours(cur_points, batch_dict["point_id_offset"].int())
                knn_done = torch.cuda.Event()
                knn_done.record(side)
            knn_ids.record_stream(main)
            cur_points.record_stream(side)
        point_features = self.point_encoder(cur_points)

        batch_dict["voxel_features"] = self.vfe(points if self.use_multi_sweeps else point_features, seg)
        batch_dict = self.point_transformer(batch_dict)

        point_voxel_features = ops.gather_rows(batch_dict["voxel_features"], cur_ids, cur_seg)
        fused = torch.cat([point_features, point_voxel_features], dim=1)
        if self.use_image_feature:
            if knn_done is not None:
                torch.cuda.current_stream(points.device).wait_event(knn_done)
            img = self.deep_fusion(cur_points, batch_dict["point_id_offset"].int(), fused,
                                   batch_dict["point_image_features"], knn_ids)
            fused = torch.cat([fused, img], dim=1)
        fused = self.fusion_encoder(fused)

        # cumulative current-sweep rows per sample: python ints from the batch builder, else the collated tensor
        row_offsets = batch_dict.get("point_row_offsets")
        if row_offsets is None and batch_dict.get("point_id_offset") is not None:
            row_offsets = [int(v) for v in batch_dict["point_id_offset"].tolist()]
        fused = fused + self.se(fused, batch_rows, row_offsets)

        result = OrderedDict()
        result["point_out"] = self.classifier(fused)
        result["voxel_out"] = batch_dict["voxel_out"]
        result["aux_voxel_out"] = batch_dict["aux_voxel_out"]
        result["voxel_coords"] = batch_dict["voxel_coords"]
        result["aux_voxel_coords"] = batch_dict["aux_voxel_coords"]
        return result


def build_segmentor(cfg, dataset):
    """seg3d/models/builder.py:8-23: 'segformer' (the default every shipped config uses) or 'spnet'."""
    if cfg.MODEL.SEGMENTOR == "spnet":
        from .spnet import SPNet
        return SPNet(dataset=dataset)
    if cfg.MODEL.SEGMENTOR != "segformer":
        raise NotImplementedError(f"MODEL.SEGMENTOR={cfg.MODEL.SEGMENTOR!r}: the reference builds 'segformer' or 'spnet'")
    batching_info = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    return Segformer(dataset=dataset, batching_info=batching_info, window_shape=cfg.MODEL.WINDOW_SHAPE,
                     depths=cfg.MODEL.DEPTHS, drop_path_rate=cfg.MODEL.DROP_PATH_RATE)
