"""GPU parity of the fused row-wise normalisation passes (rownorm.hip) against torch's fp64 reference."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,c", [(1000, 48), (4099, 96), (257, 192), (3000, 384), (5, 32), (70001, 96)])
@pytest.mark.parametrize("with_res", [True, False])
def test_layernorm_residual_forward_backward(m, c, with_res):
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m + c)
    x = torch.randn(m, c) * 2 + 0.5
    r = torch.randn(m, c)
    g = torch.randn(m, c)
    ln = nn.LayerNorm(c)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(c))
        ln.bias.copy_(0.1 * torch.randn(c))
    ref_ln = nn.LayerNorm(c).double()
    ref_ln.load_state_dict({k: v.double() for k, v in ln.state_dict().items()})
    xr, rr = x.double().requires_grad_(), r.double().requires_grad_()
    yr = ref_ln(xr) + (rr if with_res else 0)
    yr.backward(g.double())

    ln = ln.to(dev)
    xg, rg = x.to(dev).requires_grad_(), r.to(dev).requires_grad_()
    y = ops.layer_norm_residual(xg, rg if with_res else None, ln)
    assert "LayerNormResidual" in type(y.grad_fn).__name__
    y.backward(g.to(dev))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 2e-5
    if with_res:
        assert float((rg.grad.cpu().double() - rr.grad).abs().max()) < 1e-6
    for got, ref in ((ln.weight.grad, ref_ln.weight.grad), (ln.bias.grad, ref_ln.bias.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,c", [(1000, 48), (3001, 192)])
def test_layernorm_residual_with_drop_path_scale(m, c):
    """x + mask/keep * LN(a): the per-row stochastic-depth factor rides in the fused pass (drop.py:6-19)."""
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m)
    a, x, g = torch.randn(m, c), torch.randn(m, c), torch.randn(m, c)
    scale = (torch.rand(m) < 0.8).float() / 0.8
    ln = nn.LayerNorm(c)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(c))
        ln.bias.copy_(0.1 * torch.randn(c))
    ref_ln = nn.LayerNorm(c).double()
    ref_ln.load_state_dict({k: v.double() for k, v in ln.state_dict().items()})
    ar, xr = a.double().requires_grad_(), x.double().requires_grad_()
    yr = xr + ref_ln(ar) * scale.double()[:, None]
    yr.backward(g.double())
    ln = ln.to(dev)
    ag, xg = a.to(dev).requires_grad_(), x.to(dev).requires_grad_()
    y = ops.layer_norm_residual(ag, xg, ln, rowscale=scale.to(dev))
    y.backward(g.to(dev))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    assert float((ag.grad.cpu().double() - ar.grad).abs().max()) < 2e-5
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-6
    for got, ref in ((ln.weight.grad, ref_ln.weight.grad), (ln.bias.grad, ref_ln.bias.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,c", [(1000, 48), (4099, 96), (300, 256), (70001, 64), (2, 32)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("with_res", [True, False])
def test_batchnorm_act_training_and_eval(m, c, relu, with_res):
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m * 3 + c)
    x = torch.randn(m, c) * 1.5 + 3.0  # non-zero mean: exercises the shifted statistics
    r = torch.randn(m, c)
    g = torch.randn(m, c)
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(c))
        bn.bias.copy_(0.1 * torch.randn(c))
        bn.running_mean.copy_(torch.randn(c))
        bn.running_var.copy_(0.5 + torch.rand(c))
    ref = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    if relu:
        # keep every pre-activation away from the ReLU kink: the fp32 batch statistics differ from the fp64 reference by
        # rounding, so an element within rounding of zero may take the other branch and move dx / dgamma by O(1)
        for _ in range(4):
            with torch.no_grad():
                pre = torch.nn.functional.batch_norm(x.double(), None, None, ref.weight, ref.bias, True, 0.0, 1e-3)
                if with_res:
                    pre = pre + r.double()
                near = pre.abs() < 2e-3
                if not bool(near.any()):
                    break
                x[near] += 0.05

    pre_ref = []

    def ref_fwd(mod, xx, rr_):
        y = mod(xx)
        if with_res:
            y = y + rr_
        pre_ref.append(y.detach())
        return torch.relu(y) if relu else y

    # ---- training
    xr, rr = x.double().requires_grad_(), r.double().requires_grad_()
    yr = ref_fwd(ref.train(), xr, rr)
    yr.backward(g.double())
    bn = bn.to(dev).train()
    xg, rg = x.to(dev).requires_grad_(), r.to(dev).requires_grad_()
    y = ops.batch_norm_act(xg, bn, relu=relu, res=rg if with_res else None)
    assert "BatchNormAct" in type(y.grad_fn).__name__
    y.backward(g.to(dev))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 5e-5
    if relu:
        assert float(pre_ref[0].abs().min()) > 1e-3
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 5e-5
    if with_res:
        assert float((rg.grad.cpu().double() - rr.grad).abs().max()) < 1e-6
    for got, want in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        assert float((got.cpu().double() - want).abs().max()) < 1e-3 * max(1.0, float(want.abs().max()))
    assert float((bn.running_mean.cpu().double() - ref.running_mean).abs().max()) < 1e-5
    assert float((bn.running_var.cpu().double() - ref.running_var).abs().max()) < 1e-4
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)
    # ---- eval (running statistics)
    with torch.no_grad():
        ye = ops.batch_norm_act(x.to(dev), bn.eval(), relu=relu, res=r.to(dev) if with_res else None)
        yre = ref_fwd(ref.eval(), x.double(), r.double())
    assert float((ye.cpu().double() - yre).abs().max()) < 5e-5


@pytest.mark.parametrize("m,c", [(50001, 6), (3000, 8), (777, 10)])
def test_narrow_batchnorm_matches_torch(m, c):
    """The input BatchNorm of the point encoder (6 / 8 raw channels): padded rows through the library kernels vs
    nn.BatchNorm1d in fp64 -- output, parameter gradients, input gradient, running buffers."""
    from openseg3d_amd.segformer import NarrowBatchNorm1d
    dev = torch.device("cuda:0")
    torch.manual_seed(c)
    x = torch.randn(m, c) * torch.tensor([30.0, 30.0, 2.0, 1.0, 0.3, 0.1, 5.0, 1.0, 1.0, 1.0][:c]) + 3.0
    g = torch.randn(m, c)
    bn = NarrowBatchNorm1d(c).to(dev).train()
    ref = nn.BatchNorm1d(c).double().train()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(c)), bn.bias.copy_(0.1 * torch.randn(c))
        ref.weight.copy_(bn.weight.cpu().double()), ref.bias.copy_(bn.bias.cpu().double())
    xr = x.double().requires_grad_()
    yr = ref(xr)
    yr.backward(g.double())
    xg = x.to(dev).requires_grad_()
    y = bn(xg)
    y.backward(g.to(dev))
    assert y.shape == (m, c)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 5e-5
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 5e-5
    for got, want in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        assert float((got.cpu().double() - want).abs().max()) < 1e-3 * max(1.0, float(want.abs().max()))
    assert float((bn.running_mean.cpu().double() - ref.running_mean).abs().max()) < 1e-4
    assert float((bn.running_var.cpu().double() - ref.running_var).abs().max()) < 1e-3 * float(ref.running_var.max())
    assert int(bn.num_batches_tracked) == 1


def test_eval_batchnorm_fold_follows_training_and_checkpoint_loads():
    """The cached eval-mode affine (scale, shift) must track the running statistics: after training steps (buffers
    updated by the kernel through raw pointers) and after load_state_dict."""
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bn = nn.BatchNorm1d(32, eps=1e-3, momentum=0.1).to(dev)
    ref = nn.BatchNorm1d(32, eps=1e-3, momentum=0.1).to(dev)
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(500, 32, device=dev) * 2 + 1

    def check():
        bn.eval(), ref.eval()
        with torch.no_grad():
            assert float((ops.batch_norm_act(x, bn, relu=False) - ref(x)).abs().max()) < 1e-5

    check()
    check()  # second call comes from the cache
    for _ in range(3):
        bn.train(), ref.train()
        ops.batch_norm_act(x * 1.5 - 2.0, bn, relu=True)
        ref(x * 1.5 - 2.0)
    check()
    sd = {k: (v * 0.5 + 0.25 if v.is_floating_point() else v) for k, v in ref.state_dict().items()}
    bn.load_state_dict(sd), ref.load_state_dict(sd)
    check()


@pytest.mark.parametrize("n,c", [(50000, 22), (1, 22), (777, 5)])
def test_cross_entropy_matches_torch(n, c):
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 3
    y = torch.randint(0, c, (n,))
    y[::7] = 255  # ignored rows
    xr = x.double().requires_grad_()
    lr = torch.nn.functional.cross_entropy(xr, y, ignore_index=255)
    (lr * 1.7).backward()
    xg = x.to(dev).requires_grad_()
    lg = ops.cross_entropy(xg, y.to(dev), ignore_index=255)
    (lg * 1.7).backward()
    if bool((y != 255).any()):
        assert abs(float(lg) - float(lr)) < 1e-5 * max(1.0, abs(float(lr)))
        assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-7 + 1e-5 * float(xr.grad.abs().max())
    else:
        assert float(lg) == 0.0 and float(xg.grad.abs().max()) == 0.0
