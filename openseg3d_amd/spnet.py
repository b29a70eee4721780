"""SPNet: the reference's second segmentor on the same kernels.

  seg3d/models/segmentors/spnet.py:12-148         SPNet (point MLPs + SparseUnet + fusion head)
  seg3d/models/backbones/spconv_unet.py:12-233    SparseBasicBlock (optional SE), UpBlock, SparseUnet
  seg3d/models/layers/ocr.py:10-116               SpatialGatherModule, ObjectAttentionBlock, OCRLayer

Same constructor arguments, ``forward(batch_dict) -> OrderedDict`` contract and state_dict keys / shapes as the
reference (tests/golden/spnet_keys.json).  Every sparse convolution, BatchNorm(+ReLU) pass, voxel reduce and
voxel->point gather runs in libseg3d_hip.so exactly as in Segformer; the object-context attention works on
[voxels of the stride-8 level] x [22 class proxies] matrices and stays on torch.
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import segformer as segformer_mod
from . import spconv
from .segformer import (ConvBnAct, DeepFusionBlock, FlattenSELayer, FusedMLP, RowLinear, UpBlock, VFE, _bn_mlp,
                        conv_module)


class SparseBasicBlock(spconv.SparseModule):
    """conv-BN-ReLU-conv-BN (+ squeeze-excite over each sample's voxels) + identity, ReLU (spconv_unet.py:12-66)."""

    def __init__(self, inplanes, planes, norm_fn, act_fn, with_se=False, indice_key=None):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(inplanes, planes, 3, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.act = act_fn
        self.conv2 = spconv.SubMConv3d(planes, planes, 3, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.se = FlattenSELayer(planes) if with_se else None
        self.sa = None  # SALayer is never enabled by the reference's SparseUnet

    def forward(self, x):
        if self.se is None and self.conv1.fusable_with(self.bn1, x) and self.conv2.fusable_with(self.bn2, x):
            y = self.conv1.forward_bn_act(x, self.bn1, relu=True)  # inference: BatchNorm folded, two launches
            return self.conv2.forward_bn_act(y, self.bn2, relu=True, res=x.features)
        y = self.conv1(x)
        y = y.replace_feature(ops.batch_norm_act(y.features, self.bn1, relu=True))
        y = self.conv2(y)
        if self.se is None:
            return y.replace_feature(ops.batch_norm_act(y.features, self.bn2, relu=True, res=x.features))
        f = ops.batch_norm_act(y.features, self.bn2, relu=False)
        f = self.se(f, y.indices[:, 0], y.level.sample_offsets())
        return y.replace_feature(torch.relu(f + x.features))


class SpatialGatherModule(nn.Module):
    """Class proxies of one sample: softmax of the coarse class scores over the sample's voxels, times the features
    (ocr.py:10-36)."""

    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, feats, probs, batch_size, batch_indices, offsets=None):
        if offsets is not None and feats.is_cuda and ops.class_context_fits(probs.shape[1], feats.shape[1]):
            # samples are row spans: one segmented softmax-matmul for the whole batch (seg3d_class_context_fwd / _bwd)
            return ops.class_context(feats, probs, offsets[:batch_size], self.scale)
        out = []
        for i in range(batch_size):
            sel = slice(offsets[i - 1] if i else 0, offsets[i]) if offsets is not None else batch_indices == i
            prob = F.softmax(self.scale * probs[sel].t(), dim=1)  # [classes, n_i]
            out.append(prob @ feats[sel])
        return torch.stack(out)  # [batch, classes, C]


class ObjectAttentionBlock(nn.Module):
    """Voxel-to-proxy attention (ocr.py:39-82); the BatchNorms see one sample at a time, as in the reference."""

    def __init__(self, in_channels, key_channels):
        super().__init__()
        self.key_channels = key_channels

        def proj(cin, cout):
            return FusedMLP(nn.Linear(cin, cout, bias=False), nn.BatchNorm1d(cout), nn.ReLU(inplace=True))

        self.query_project = proj(in_channels, key_channels)
        self.key_project = proj(in_channels, key_channels)
        self.value_project = proj(in_channels, key_channels)
        self.bottleneck = proj(key_channels, in_channels)

    def forward(self, x, proxy):
        query = self.query_project(x)
        key = self.key_project(proxy)
        value = self.value_project(proxy)
        sim = F.softmax((self.key_channels ** -0.5) * (query @ key.t()), dim=-1)
        return self.bottleneck(sim @ value)


class OCRLayer(nn.Module):
    """Object-contextual representation on the stride-8 level (ocr.py:85-116)."""

    def __init__(self, in_channels, mid_channels, key_channels, scale=1.0, drop=0.05):
        super().__init__()
        self.scale = scale
        self.transform_input = ConvBnAct(spconv.SubMConv3d(in_channels, mid_channels, 3, padding=1, bias=False),
                                         nn.BatchNorm1d(mid_channels), nn.ReLU(inplace=True))
        self.spatial_gather_module = SpatialGatherModule(self.scale)
        self.object_context_block = ObjectAttentionBlock(mid_channels, key_channels)
        self.bottleneck = FusedMLP(nn.Linear(mid_channels * 2, in_channels, bias=False), nn.BatchNorm1d(in_channels),
                                   nn.ReLU(inplace=True), nn.Dropout(drop))

    def forward(self, inputs, probs, batch_size):
        inputs = self.transform_input(inputs)
        feats = inputs.features
        batch_indices = inputs.indices[:, 0]
        offsets = inputs.level.sample_offsets()  # contiguous samples: slices instead of masks and scatters
        context = self.spatial_gather_module(feats, probs, batch_size, batch_indices, offsets)
        pieces, rows = [], []
        for i in range(batch_size):
            if offsets is not None:
                sel = slice(offsets[i - 1] if i else 0, offsets[i])
            else:
                sel = torch.nonzero(batch_indices == i).view(-1)
                rows.append(sel)
            pieces.append(self.object_context_block(feats[sel], context[i]))
        out = torch.cat(pieces) if batch_size > 1 else pieces[0]
        if rows:
            out = torch.empty_like(feats).index_copy(0, torch.cat(rows), out)
        feats = self.bottleneck(torch.cat([out, feats], dim=1))
        return inputs.replace_feature(feats)


class SparseUnet(nn.Module):
    """spconv_unet.py:115-233: 4-level sparse U-Net, channels 32/64/128/256, OCR on the coarsest level."""

    def __init__(self, input_channels, output_channels, grid_size, voxel_size, point_cloud_range, num_classes,
                 use_ocr=True):
        super().__init__()
        self.sparse_shape = grid_size[::-1]
        self.voxel_size, self.point_cloud_range, self.use_ocr = voxel_size, point_cloud_range, use_ocr
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        act_fn = nn.ReLU(inplace=True)
        block = partial(SparseBasicBlock, norm_fn=norm_fn, act_fn=act_fn)
        self.conv_input = conv_module(input_channels, 32, norm_fn, act_fn, "subm", "subm1")
        self.conv1 = spconv.SparseSequential(block(32, 32, indice_key="subm1"), block(32, 32, indice_key="subm1"))
        self.conv2 = spconv.SparseSequential(
            conv_module(32, 64, norm_fn, act_fn, "spconv", "spconv2"), block(64, 64, indice_key="subm2"),
            block(64, 64, indice_key="subm2"), block(64, 64, indice_key="subm2"))
        self.conv3 = spconv.SparseSequential(
            conv_module(64, 128, norm_fn, act_fn, "spconv", "spconv3"), block(128, 128, indice_key="subm3"),
            block(128, 128, indice_key="subm3"), block(128, 128, with_se=True, indice_key="subm3"))
        self.conv4 = spconv.SparseSequential(
            conv_module(128, 256, norm_fn, act_fn, "spconv", "spconv4"), block(256, 256, indice_key="subm4"),
            block(256, 256, indice_key="subm4"), block(256, 256, with_se=True, indice_key="subm4"))
        up = partial(_UpBlock, norm_fn=norm_fn, act_fn=act_fn)
        self.up4 = up(256, 128, conv_type="inverseconv", layer_id=4)
        self.up3 = up(128, 64, conv_type="inverseconv", layer_id=3)
        self.up2 = up(64, 32, conv_type="inverseconv", layer_id=2)
        self.up1 = up(32, output_channels, conv_type="subm", layer_id=1)
        if self.use_ocr:
            self.ocr = OCRLayer(256, 128, 64)
        self.aux_voxel_classifier = nn.Sequential(RowLinear(256, num_classes, bias=False))
        self.voxel_classifier = nn.Sequential(RowLinear(output_channels, num_classes, bias=False))

    def prepare(self, batch_dict):
        """All site levels, rulebooks, row orders and per-sample row offsets before the first feature kernel: the three
        host read-backs (strided level sizes) happen while only index kernels are in flight (see
        segformer.PointTransformer.prepare)."""
        level = spconv.SiteLevel(batch_dict["voxel_coords"].int(), self.sparse_shape, batch_dict["batch_size"])
        batch_dict["site_level"] = level
        level.seed_chain(3)
        for k in range(4):
            level.subm()
            level.mask_order()
            level.subm_plan()
            if k >= 2:
                level.sample_offsets()  # squeeze-excite of conv3 / conv4, OCR
            if k < 3:
                level.parity_order()
                level = level.down()[0]
        return batch_dict

    def forward(self, batch_dict):
        coords = batch_dict["voxel_coords"]
        x = spconv.SparseConvTensor(batch_dict["voxel_features"], coords.int(), self.sparse_shape, batch_dict["batch_size"],
                                    _level=batch_dict.get("site_level"))
        x = self.conv_input(x)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        aux = self.aux_voxel_classifier(x4.features)
        batch_dict["aux_voxel_out"], batch_dict["aux_voxel_coords"] = aux, x4.indices
        if self.use_ocr:
            x4 = self.ocr(x4, aux, batch_dict["batch_size"])
        y = self.up4(x4, x4)
        y = self.up3(y, x3)
        y = self.up2(y, x2)
        y = self.up1(y, x1)
        batch_dict["voxel_features"], batch_dict["voxel_coords"] = y.features, y.indices
        batch_dict["voxel_out"] = self.voxel_classifier(y.features)
        return batch_dict


class _UpBlock(UpBlock):
    """UpBlock whose lateral transform is this file's SparseBasicBlock (same parameters; spconv_unet.py:69-112)."""

    def __init__(self, inplanes, planes, norm_fn, act_fn, conv_type, layer_id):
        super().__init__(inplanes, planes, norm_fn, act_fn, conv_type, layer_id)
        self.transform = SparseBasicBlock(inplanes, inplanes, norm_fn, act_fn, indice_key=f"subm{layer_id}")


class SPNet(nn.Module):
    def __init__(self, dataset):
        super().__init__()
        dim_point = dataset.dim_point + (2 if dataset.use_cylinder else 0)
        self.use_multi_sweeps = bool(dataset.use_multi_sweeps)
        self.use_image_feature = bool(dataset.use_image_feature)
        self.point_feature_channel = 64
        self.point_encoder = _bn_mlp([dim_point, 64, 128, 256, self.point_feature_channel], first_bn=dim_point,
                                     last_plain=True)
        self.vfe = VFE(dim_point, reduce="mean") if self.use_multi_sweeps else VFE(self.point_feature_channel, "max")
        self.voxel_feature_channel = 64
        self.voxel_encoder = SparseUnet(self.vfe.voxel_feature_channel, self.voxel_feature_channel, dataset.grid_size,
                                        dataset.voxel_size, dataset.point_cloud_range, dataset.num_classes)
        self.image_feature_channel = dataset.dim_image_feature if self.use_image_feature else 0
        if self.use_image_feature:
            self.deep_fusion = DeepFusionBlock(self.point_feature_channel + self.voxel_feature_channel,
                                               self.image_feature_channel, 32, 16)
        self.fusion_feature_channel = 64
        self.fusion_encoder = _bn_mlp([self.point_feature_channel + self.voxel_feature_channel
                                       + self.image_feature_channel, 256, 128, self.fusion_feature_channel])
        self.se = FlattenSELayer(self.fusion_feature_channel)
        self.classifier = FusedMLP(RowLinear(self.fusion_feature_channel, 64, bias=False), nn.BatchNorm1d(64),
                                   nn.ReLU(True), nn.Dropout(0.3), RowLinear(64, dataset.num_classes, bias=False))
        self.weight_initialization()

    def weight_initialization(self):
        """spnet.py:77-92."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm1d, nn.LayerNorm)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def prepare_batch(self, batch_dict):
        """The forward's index plan, built ahead of time and carried by the batch (segformer.Segformer.prepare_batch)."""
        if "site_level" not in batch_dict:
            self.voxel_encoder.prepare(batch_dict)
        return batch_dict

    def forward(self, batch_dict):
        if self.training and torch.is_grad_enabled():
            ops.probe_deferred_join(batch_dict["points"].device)  # once per process: self-test of the deferred join
        with ops.deferred_bn_counters():  # one launch for all BatchNorm step counters
            return self._forward(batch_dict)

    def _forward(self, batch_dict):
        points = batch_dict["points"][:, 1:]
        ids = batch_dict["point_voxel_ids"]
        n_voxels = batch_dict["voxel_coords"].shape[0]
        seg = batch_dict.get("point_voxel_index")
        if seg is None:
            seg = ops.SegmentIndex(ids, n_voxels)
        if segformer_mod.PLAN_FIRST and "site_level" not in batch_dict:
            self.voxel_encoder.prepare(batch_dict)
        if self.use_multi_sweeps:
            cur_rows = torch.nonzero(points[:, 3] == 0).view(-1)  # spnet.py:97
            cur_points, cur_ids, cur_seg = points[cur_rows], seg.ids[cur_rows], None
            batch_rows = batch_dict["points"][cur_rows, 0]
        else:
            cur_points, cur_ids, cur_seg, batch_rows = points, seg.ids, seg, batch_dict["points"][:, 0]
        point_features = self.point_encoder(cur_points)
        batch_dict["voxel_features"] = self.vfe(points if self.use_multi_sweeps else point_features, seg)
        batch_dict = self.voxel_encoder(batch_dict)

        point_voxel_features = ops.gather_rows(batch_dict["voxel_features"], cur_ids, cur_seg)
        fused = torch.cat([point_features, point_voxel_features], dim=1)
        if self.use_image_feature:
            img = self.deep_fusion(cur_points, batch_dict["point_id_offset"].int(), fused,
                                   batch_dict["point_image_features"])
            fused = torch.cat([fused, img], dim=1)
        fused = self.fusion_encoder(fused)
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
