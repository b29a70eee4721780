"""Oracle restatement of the reference's training criterion -- TEST INFRASTRUCTURE ONLY.

torch-CPU, written from the algorithm (autograd supplies the gradients the device kernels are checked against):

  seg3d/models/losses/ohem_cross_entropy_loss.py:23-38   OHEM cross-entropy by probability threshold
  seg3d/models/losses/lovasz_loss.py:13-26, 118-158      Lovasz-softmax, classes 'present' | 'all' | list
  tools/train.py:71-110                                  sum over the three heads

Pinned by tests/golden/losses.npz, which tests/golden/make_golden.py produced with the reference's own loss modules.
"""
import torch
import torch.nn.functional as F


def ohem_cross_entropy(logits, labels, keep_thresh, ignore_index=255):
    valid = labels != ignore_index
    x, y = logits[valid], labels[valid]
    per_row = F.cross_entropy(x, y, reduction="none")
    if keep_thresh:
        p_target = F.softmax(x, dim=1).gather(1, y[:, None])[:, 0]
        per_row = per_row[p_target < keep_thresh]
    return per_row.mean()


def jaccard_steps(fg_sorted):
    """Increments of 1 - |fg ∩ top-k| / |fg ∪ top-k| along the descending-error order (lovasz_grad)."""
    total = fg_sorted.sum()
    inter = total - fg_sorted.cumsum(0)
    union = total + (1.0 - fg_sorted).cumsum(0)
    jac = 1.0 - inter / union
    steps = jac.clone()
    steps[1:] = jac[1:] - jac[:-1]
    return steps


def lovasz_softmax(logits, labels, ignore_index=255, classes="present", class_weight=None):
    valid = labels != ignore_index
    probs, y = F.softmax(logits, dim=1)[valid], labels[valid]
    if probs.numel() == 0:
        return logits.sum() * 0.0
    chosen = range(probs.shape[1]) if classes in ("present", "all") else classes
    terms = []
    for c in chosen:
        fg = (y == c).float()
        if classes == "present" and fg.sum() == 0:
            continue
        err, order = torch.sort((fg - probs[:, c]).abs(), descending=True)
        term = torch.dot(err, jaccard_steps(fg[order]))
        terms.append(term * class_weight[c] if class_weight is not None else term)
    return torch.stack(terms).mean()
