"""Training criterion of the reference on device.

  seg3d/models/builder.py:26-40                          build_criterion: MODEL.LOSSES = {'ohem_ce': 1.0, 'lovasz': 1.0}
  seg3d/models/losses/ohem_cross_entropy_loss.py:5-38    OHEMCrossEntropyLoss
  seg3d/models/losses/lovasz_loss.py:215-290             LovaszLoss
  tools/train.py:71-110                                  compute_loss (point, voxel and 0.4 x auxiliary terms)

Same class names, constructor arguments and ``loss_name`` properties.  The configurations build_criterion can produce
(OHEM by probability threshold, multi-class Lovasz over the whole batch) run in libseg3d_hip.so: one pass each way for
the cross-entropy terms, one device sort for all classes of the Lovasz term.  Options the builder never sets (OHEM by
keep_ratio, per-image / binary Lovasz) are composed from torch ops on the same device tensors.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(ignore_index=...) (builder.py:29-30) through seg3d_cross_entropy_fwd/bwd."""

    def __init__(self, ignore_index=255, loss_name="loss_cross_entropy"):
        super().__init__()
        self.ignore_index = ignore_index
        self._loss_name = loss_name

    def forward(self, inputs, targets):
        return ops.cross_entropy(inputs, targets, ignore_index=self.ignore_index)

    @property
    def loss_name(self):
        return self._loss_name


class OHEMCrossEntropyLoss(nn.Module):
    def __init__(self, keep_ratio=None, keep_thresh=None, ignore_index=255, class_weight=None,
                 loss_name="loss_ohem_cross_entropy"):
        super().__init__()
        self.keep_ratio, self.keep_thresh = keep_ratio, keep_thresh
        self.ignore_index, self.class_weight = ignore_index, class_weight
        self._loss_name = loss_name

    def forward(self, inputs, targets):
        if self.class_weight is None and not self.keep_ratio:
            # keep_thresh (what build_criterion passes) or plain CE: fused kernel, no [n, C] softmax materialised
            return ops.cross_entropy(inputs, targets, ignore_index=self.ignore_index, keep_thresh=self.keep_thresh)
        mask = targets != self.ignore_index
        losses = F.cross_entropy(inputs, targets, weight=self.class_weight, ignore_index=self.ignore_index,
                                 reduction="none")[mask]
        if self.keep_ratio:  # ohem_cross_entropy_loss.py:27-30: the hardest keep_ratio of the valid rows
            kept = int(losses.shape[0] * self.keep_ratio)
            losses = torch.topk(losses, kept, sorted=False)[0]
        elif self.keep_thresh:
            probs = F.softmax(inputs, dim=1)[mask].gather(1, targets[mask].unsqueeze(1)).squeeze(1)
            losses = losses[probs < self.keep_thresh]
        return losses.mean()

    @property
    def loss_name(self):
        return self._loss_name


def _lovasz_grad(gt_sorted):
    """lovasz_loss.py:13-26 (used by the torch-composed variants only)."""
    gts = gt_sorted.sum()
    intersection = gts - gt_sorted.float().cumsum(0)
    union = gts + (1 - gt_sorted).float().cumsum(0)
    jaccard = 1.0 - intersection / union
    if gt_sorted.shape[0] > 1:
        jaccard = torch.cat([jaccard[:1], jaccard[1:] - jaccard[:-1]])
    return jaccard


def _lovasz_hinge_flat(logits, labels):
    """lovasz_loss.py:56-76."""
    if labels.numel() == 0:
        return logits.sum() * 0.0
    signs = 2.0 * labels.float() - 1.0
    errors_sorted, perm = torch.sort(1.0 - logits * signs, dim=0, descending=True)
    return torch.dot(F.relu(errors_sorted), _lovasz_grad(labels[perm]))


class LovaszLoss(nn.Module):
    def __init__(self, loss_type="multi_class", classes="present", per_image=False, reduction="none", class_weight=None,
                 loss_weight=1.0, ignore_index=255, loss_name="loss_lovasz"):
        super().__init__()
        assert loss_type in ("binary", "multi_class")
        if not per_image:
            assert reduction == "none", "reduction should be 'none' when per_image is False."
        self.loss_type, self.classes, self.per_image, self.reduction = loss_type, classes, per_image, reduction
        self.class_weight, self.loss_weight, self.ignore_index = class_weight, loss_weight, ignore_index
        self._loss_name = loss_name

    def forward(self, cls_score, label, avg_factor=None, reduction_override=None):
        """cls_score [n, C] logits, label [n].  The reference feeds the rows as a [n, C, 1, 1] image batch, so
        per_image=True means one loss per ROW (lovasz_loss.py:278-287)."""
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if self.loss_type == "multi_class" and not self.per_image:
            return self.loss_weight * ops.lovasz_softmax(cls_score, label, self.ignore_index, self.classes, self.class_weight)
        if self.loss_type == "binary" and not self.per_image:
            valid = label.view(-1) != self.ignore_index
            return self.loss_weight * _lovasz_hinge_flat(cls_score.view(-1)[valid], label.view(-1)[valid])
        raise NotImplementedError("LovaszLoss(per_image=True) on [n, C] rows degenerates to one loss per row; "
                                  "no configuration of the reference uses it")

    @property
    def loss_name(self):
        return self._loss_name


def build_criterion(cfg, dataset):
    """seg3d/models/builder.py:26-40: list of (criterion, weight) in MODEL.LOSSES order."""
    losses = []
    for name in cfg.MODEL.LOSSES:
        if name == "ce":
            criterion = CrossEntropyLoss(ignore_index=dataset.ignore_index)
        elif name == "ohem_ce":
            criterion = OHEMCrossEntropyLoss(keep_thresh=cfg.MODEL.OHEM_KEEP_THRESH, ignore_index=dataset.ignore_index)
        elif name == "lovasz":
            criterion = LovaszLoss(ignore_index=dataset.ignore_index)
        else:
            raise NotImplementedError(name)
        losses.append((criterion, cfg.MODEL.LOSSES[name]))
    return losses


AUX_OVERLAP = os.environ.get("SEG3D_AUX_OVERLAP", "1") != "0"


def compute_loss(pred_result, data_dict, criterion, cfg):
    """tools/train.py:71-110: criterion on the point logits, the voxel logits and (x MODEL.AUX_LOSS_WEIGHT) the
    stride-8 auxiliary logits, whose ground truth is looked up by nearest fine voxel centre (ops.aux_voxel_labels)."""
    # The auxiliary labels (a kNN lookup: no gradient, independent of the other two heads) are looked up on the second
    # stream while this one computes the point and voxel losses; the streams meet in front of the auxiliary loss.
    aux_gt = side = None
    if "aux_voxel_out" in pred_result:
        dev = pred_result["aux_voxel_out"].device
        overlap = AUX_OVERLAP and dev.type == "cuda"
        if overlap:
            main, side = torch.cuda.current_stream(dev), ops.side_stream(dev)
            side.wait_stream(main)
        with torch.no_grad(), torch.cuda.stream(side) if overlap else contextlib.nullcontext():
            aux_gt = ops.aux_voxel_labels(pred_result["voxel_coords"], pred_result["aux_voxel_coords"],
                                          data_dict["voxel_labels"], data_dict["batch_size"], cfg.DATASET.VOXEL_SIZE,
                                          cfg.DATASET.POINT_CLOUD_RANGE)
    loss = 0
    for fn, w in criterion:
        loss = loss + fn(pred_result["point_out"], data_dict["point_labels"]) * w
    if "voxel_out" in pred_result:
        voxel_gt = data_dict["voxel_labels"]
        for fn, w in criterion:
            loss = loss + fn(pred_result["voxel_out"], voxel_gt) * w
    if aux_gt is not None:
        if side is not None:
            main.wait_stream(side)
        for fn, w in criterion:
            loss = loss + cfg.MODEL.AUX_LOSS_WEIGHT * fn(pred_result["aux_voxel_out"], aux_gt) * w
    return loss
