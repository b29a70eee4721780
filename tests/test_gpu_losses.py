"""device OHEM cross-entropy and Lovasz-softmax against the reference's loss modules (golden
vectors written by make_golden.gen_losses) and against the oracle restatement at the headline size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _case(name, d):
    from openseg3d_amd import losses
    return {"ohem": losses.OHEMCrossEntropyLoss(keep_thresh=0.7, ignore_index=255),
            "lovasz": losses.LovaszLoss(ignore_index=255),
            "lovasz_all": losses.LovaszLoss(classes="all", ignore_index=255),
            "lovasz_list": losses.LovaszLoss(classes=[0, 3, 7, 21], ignore_index=255),
            "lovasz_weighted": losses.LovaszLoss(class_weight=d["class_weight"].tolist(), ignore_index=255)}[name]


@pytest.mark.parametrize("name", ["ohem", "lovasz", "lovasz_all", "lovasz_list", "lovasz_weighted"])
def test_criterion_matches_reference_loss_modules(dev, golden_dir, name):
    """Value within 2e-6 relative, gradient within 1e-9 + 1e-5 relative of the reference module's autograd.  Rows 60..99
    and 100..139 of the fixture are identical: their errors tie, the Lovasz subgradient hands the tied ranks' steps out in
    sort order, and torch.sort leaves that order unspecified -- only the sum over each tied pair is defined."""
    d = np.load(os.path.join(golden_dir, "losses.npz"))
    x = torch.from_numpy(d["logits"]).to(dev).requires_grad_(True)
    loss = _case(name, d)(x, torch.from_numpy(d["labels"]).to(dev))
    (loss * 1.0).backward()
    ref, ref_grad = float(d[name]), d[name + "_grad"]
    assert abs(float(loss) - ref) <= 2e-6 * abs(ref), (float(loss), ref)
    got = x.grad.cpu().numpy()
    tol = 1e-9 + 1e-5 * np.abs(ref_grad).max()
    for a in (got, ref_grad):
        a[60:100] += a[100:140]
        a[100:140] = 0
    err = np.abs(got - ref_grad).max()
    assert err <= tol, err


@pytest.mark.parametrize("n", [174633, 1, 2049])
def test_criterion_matches_oracle_at_size(dev, n):
    """Headline size (one Waymo sweep), one row, and one row past a chunk boundary; upstream gradient != 1."""
    from oracle import losses as L
    from openseg3d_amd import losses
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 22, generator=g) * 2.5
    y = torch.randint(0, 22, (n,), generator=g)
    if n > 10:
        y[torch.rand(n, generator=g) < 0.1] = 255
    for fn_dev, fn_ref in ((losses.OHEMCrossEntropyLoss(keep_thresh=0.7), lambda t: L.ohem_cross_entropy(t, y, 0.7)),
                           (losses.LovaszLoss(), lambda t: L.lovasz_softmax(t, y))):
        xr = x.clone().requires_grad_(True)
        lr = fn_ref(xr)
        (lr * 0.4).backward()
        xg = x.to(dev).requires_grad_(True)
        lg = fn_dev(xg, y.to(dev))
        (lg * 0.4).backward()
        if not bool(torch.isfinite(lr)):  # nothing kept: torch's mean of an empty set is NaN, the device op returns 0
            assert float(lg) == 0.0 and float(xg.grad.abs().max()) == 0.0
            continue
        assert abs(float(lg) - float(lr)) <= 1e-5 * max(abs(float(lr)), 1e-3), (type(fn_dev).__name__, float(lg), float(lr))
        err = float((xg.grad.cpu() - xr.grad).abs().max())
        assert err <= 1e-9 + 2e-4 * float(xr.grad.abs().max()), (type(fn_dev).__name__, err)


def test_lovasz_properties(dev):
    """Size-independent properties: every row's logit gradient sums to 0 (softmax), ignored rows get exactly 0,
    permuting the rows permutes the gradient and leaves the loss unchanged, and a perfect prediction costs ~0."""
    from openseg3d_amd import ops
    g = torch.Generator().manual_seed(9)
    n = 60000
    x = (torch.randn(n, 22, generator=g) * 2).to(dev)
    y = torch.randint(0, 22, (n,), generator=g)
    y[::11] = 255
    y = y.to(dev)
    xa = x.clone().requires_grad_(True)
    la = ops.lovasz_softmax(xa, y)
    la.backward()
    assert float(xa.grad.sum(dim=1).abs().max()) < 1e-9
    assert float(xa.grad[y == 255].abs().max()) == 0.0
    perm = torch.randperm(n, generator=g).to(dev)
    xb = x[perm].clone().requires_grad_(True)
    lb = ops.lovasz_softmax(xb, y[perm])
    lb.backward()
    assert abs(float(la) - float(lb)) < 1e-6
    assert float((xb.grad - xa.grad[perm]).abs().max()) < 1e-9
    # run-to-run reproducibility (no atomics)
    xc = x.clone().requires_grad_(True)
    lc = ops.lovasz_softmax(xc, y)
    lc.backward()
    assert float(lc) == float(la) and torch.equal(xc.grad, xa.grad)
    perfect = torch.full((n, 22), -30.0, device=dev)
    valid = y != 255
    perfect[valid, y[valid]] = 30.0
    assert float(ops.lovasz_softmax(perfect, y)) < 1e-6


def test_criterion_edge_cases(dev):
    from openseg3d_amd import ops
    x = torch.randn(100, 22, device=dev, requires_grad=True)
    y = torch.full((100,), 255, device=dev)
    for loss in (ops.lovasz_softmax(x, y), ops.cross_entropy(x, y, ignore_index=255, keep_thresh=0.7)):
        assert float(loss) == 0.0
    (ops.lovasz_softmax(x, y) + ops.cross_entropy(x, y, ignore_index=255, keep_thresh=0.7)).backward()
    assert float(x.grad.abs().max()) == 0.0
    empty = torch.zeros((0, 22), device=dev, requires_grad=True)
    assert float(ops.lovasz_softmax(empty, y[:0])) == 0.0
    with pytest.raises(ValueError):
        ops.lovasz_softmax(torch.zeros(4, 100, device=dev), y[:4])  # more than 64 classes


def test_compute_loss_on_model_outputs(dev):
    """tools/train.py:71-110 end to end on a real forward: default criterion (ohem_ce + lovasz) over the three heads equals
    the oracle losses on the same logits and labels; the backward reaches every parameter."""
    from oracle import losses as L
    from openseg3d_amd import batch as B, config, losses, ops, scene, segformer
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    pts = scene.make_small_scene(3, 6000, extent=12.0)
    b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range)
    n = b["points"].shape[0]
    point_labels = (torch.arange(n, device=dev) * 7 % 23).long()
    point_labels[point_labels == 22] = 255
    b["point_labels"] = point_labels
    b["voxel_labels"] = ops.prepare_voxel_labels(b["point_voxel_ids"], point_labels.to(torch.uint8),
                                                 b["voxel_coords"].shape[0], ignore_index=255).long()
    crit = losses.build_criterion(cfg, ds)
    assert [type(f).__name__ for f, _ in crit] == ["OHEMCrossEntropyLoss", "LovaszLoss"]
    res = model(b)
    # BatchNorm step counters: bumped by one batched launch at the end of the forward (ops.deferred_bn_counters)
    counters = [int(m.num_batches_tracked) for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    assert counters and set(counters) == {1}
    loss = losses.compute_loss(res, b, crit, cfg)
    loss.backward()
    assert not [k for k, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    aux_gt = ops.aux_voxel_labels(res["voxel_coords"], res["aux_voxel_coords"], b["voxel_labels"], 1, cfg.DATASET.VOXEL_SIZE,
                                  cfg.DATASET.POINT_CLOUD_RANGE)
    want = 0.0
    for out, gt, w in ((res["point_out"], point_labels, 1.0), (res["voxel_out"], b["voxel_labels"], 1.0),
                       (res["aux_voxel_out"], aux_gt, cfg.MODEL.AUX_LOSS_WEIGHT)):
        o, t = out.detach().cpu(), gt.cpu()
        want += w * (float(L.ohem_cross_entropy(o, t, cfg.MODEL.OHEM_KEEP_THRESH)) + float(L.lovasz_softmax(o, t)))
    assert abs(float(loss) - want) <= 1e-4 * abs(want), (float(loss), want)
    # the auxiliary-label lookup runs on the second stream beside the other two heads' losses: same value, bit for bit,
    # as with everything on one stream
    with torch.no_grad():
        both = []
        for overlap in (True, False):
            losses.AUX_OVERLAP = overlap
            both.append(float(losses.compute_loss(res, b, crit, cfg)))
        losses.AUX_OVERLAP = True
    assert both[0] == both[1] == float(loss)
