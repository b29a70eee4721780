"""GPU parity of the dense-layer weight gradient (seg3d_linear_wgrad) against torch autograd in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,cin,cout", [(1000, 48, 96), (4097, 64, 64), (31, 192, 384), (70000, 96, 192),
                                        (513, 104, 48), (5000, 256, 64), (64, 768, 384)])
def test_linear_wgrad_matches_autograd(m, cin, cout):
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m + cin)
    x = torch.randn(m, cin)
    w = torch.randn(cout, cin) / cin ** 0.5
    b = torch.randn(cout)
    g = torch.randn(m, cout)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    torch.nn.functional.linear(xr, wr, br).backward(g.double())

    xg, wg, bg = x.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    y = ops.linear(xg, wg, bg)
    assert y.grad_fn is not None and "LinearFn" in type(y.grad_fn).__name__
    y.backward(g.to(dev))
    # split-bf16 products: ~2^-16 relative per term, sums of m terms of O(1) values
    scale = max(1.0, float(wr.grad.abs().max()))
    assert float((wg.grad.cpu().double() - wr.grad).abs().max()) < 1e-4 * scale
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-4
    assert float((bg.grad.cpu().double() - br.grad).abs().max()) < 1e-3 * max(1.0, float(br.grad.abs().max()))


def test_linear_wgrad_is_reproducible_and_handles_tiny_inputs():
    """Partial blocks are summed in a fixed order: two runs give the same bits.  m = 1 and m = 0 are legal."""
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    x = torch.randn(33333, 96, device=dev)
    w = torch.randn(288, 96, device=dev)
    b = torch.randn(288, device=dev)
    g = torch.randn(33333, 288, device=dev)
    grads = []
    for _ in range(2):
        wg, bg = w.clone().requires_grad_(), b.clone().requires_grad_()
        ops.linear(x, wg, bg).backward(g)
        grads.append((wg.grad.clone(), bg.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    for m in (1, 0):
        wg, bg = w.clone().requires_grad_(), b.clone().requires_grad_()
        ops.linear(x[:m], wg, bg).backward(g[:m])
        ref_w = g[:m].double().t() @ x[:m].double()
        assert float((wg.grad.double() - ref_w).abs().max()) < 1e-4
        assert float((bg.grad.double() - g[:m].double().sum(0)).abs().max()) < 1e-5


def test_linear_falls_back_for_unsupported_shapes():
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    x = torch.randn(100, 6, device=dev)
    w = torch.randn(64, 6, device=dev, requires_grad=True)  # cin % 4 != 0, few rows -> rocBLAS path
    y = ops.linear(x, w)
    assert "LinearFn" not in type(y.grad_fn).__name__
    y.sum().backward()
    assert torch.allclose(w.grad, x.sum(0).expand(64, 6), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("m,cin,cout,bias", [(20000, 6, 64, False), (50001, 64, 22, False), (8192, 10, 22, True)])
def test_odd_shape_linear_weight_gradient(m, cin, cout, bias):
    """6 -> 64 and -> 22 layers: rocBLAS forward / dX, padded split-bf16 weight gradient."""
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(cin * 100 + cout)
    x, w, g = torch.randn(m, cin), torch.randn(cout, cin) / cin ** 0.5, torch.randn(m, cout)
    b = torch.randn(cout) if bias else None
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    torch.nn.functional.linear(xr, wr, br).backward(g.double())
    xg, wg = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    bg = b.to(dev).requires_grad_() if bias else None
    y = ops.linear(xg, wg, bg, exact=True)
    assert "OddShape" in type(y.grad_fn).__name__
    y.backward(g.to(dev))
    scale = max(1.0, float(wr.grad.abs().max()))
    assert float((wg.grad.cpu().double() - wr.grad).abs().max()) < 1e-4 * scale
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-4
    if bias:
        assert float((bg.grad.cpu().double() - br.grad).abs().max()) < 1e-3 * max(1.0, float(br.grad.abs().max()))


def test_batched_pack_refresh_matches_single_packs():
    """After an in-place update of several parameters the registry repacks all of them in one launch; the streams
    must equal what the single-weight entry points produce."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    lin = [torch.nn.Parameter(torch.randn(co, ci, device=dev)) for co, ci in ((96, 48), (64, 128), (384, 192))]
    conv = [torch.nn.Parameter(torch.randn(co, 3, 3, 3, ci, device=dev)) for co, ci in ((48, 32), (96, 96))]
    for rnd in range(2):
        with torch.no_grad():
            for p in lin + conv:
                p.add_(0.5)  # bumps the version: every registered pack is stale
        got = []
        for p in lin:
            got += [ops._linear_pack(p, 0).clone(), ops._linear_pack(p[:32], 1).clone()]
        for p in conv:
            got += [ops.pack_weight(p, ops.PACK_FWD).data.clone(), ops.pack_weight(p, ops.PACK_T_FLIP).data.clone()]
        want = []
        for p in lin:
            for w, tr in ((p, 0), (p[:32], 1)):
                out = torch.empty(_lib.query("seg3d_linear_packed_bytes", w.shape[1], w.shape[0], tr), dtype=torch.uint8, device=dev)
                _lib.call("seg3d_linear_pack_weight", ops._ptr(w.contiguous()), w.shape[1], w.shape[0], tr, ops._ptr(out), ops._stream())
                want.append(out)
        for p in conv:
            for fl in (4, 4 | 3):
                out = torch.empty(_lib.query("seg3d_spconv_packed_bytes", p.shape[-1], p.shape[0], fl), dtype=torch.uint8, device=dev)
                _lib.call("seg3d_spconv_pack_weight", ops._ptr(p), p.shape[-1], p.shape[0], fl, ops._ptr(out), ops._stream())
                want.append(out)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.shape == b.shape and torch.equal(a, b), rnd
