"""GPU parity of the dense-layer weight gradient (seg3d_linear_wgrad) against torch autograd in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,cin,cout", [(1000, 48, 96), (4097, 64, 64), (31, 192, 384), (70000, 96, 192),
                                        (513, 104, 48), (5000, 256, 64), (64, 768, 384),
                                        # the wide kernel's block shapes: 2 x 3, 3 x 2, 2 x 2 waves, ragged last steps
                                        (40001, 192, 384), (33333, 384, 192), (20011, 384, 384), (9999, 384, 768),
                                        (20000, 768, 384), (12345, 128, 256)])
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


def test_batched_partial_sums_match_the_single_launch_entries():
    """seg3d_linear_wgrad_partials + seg3d_reduce_partials_batched (all parameter-gradient sums of a backward pass in ONE
    launch, ops._run_reduce_jobs) against the entries that sum at once: the same kernels and the same summation order, so
    Linear dw / db come out bit-identical; LayerNorm's dgamma / dbeta (whose own reduce kernel sums in another order) are
    compared with float64.  Includes an empty job (m = 0 -> zeros) and the 2-job path that launches singly."""
    import ctypes
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    p, st = ops._ptr, ops._stream
    fk = ops._WgradFork(dev)
    fk.on = False  # everything on the current stream: this test is about the sums, not the streams
    want, got = [], []
    for m, cin, cout in [(20000, 48, 96), (777, 192, 192), (0, 96, 48), (5001, 64, 16)]:
        x, dy = torch.randn(m, cin, device=dev), torch.randn(m, cout, device=dev)
        dw, db = torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev)
        nbytes = _lib.query("seg3d_linear_wgrad_workspace_bytes", m, cin, cout)
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
        _lib.call("seg3d_linear_wgrad", p(x), p(dy), m, cin, cout, p(dw), p(db), p(ws), nbytes, st())
        want.append((dw, db))
        dw2, db2 = torch.full_like(dw, float("nan")), torch.full_like(db, float("nan"))
        ops._linear_wgrad_into(fk, x, dy, cin, cout, dw2.data_ptr(), db2.data_ptr())
        got.append((dw2, db2))
    m, c = 30000, 96
    x, dy = torch.randn(m, c, device=dev), torch.randn(m, c, device=dev)
    gamma = torch.randn(c, device=dev)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    dx, dg, dbe = torch.empty_like(x), torch.full((c,), float("nan"), device=dev), torch.full((c,), float("nan"), device=dev)
    nbytes = _lib.query("seg3d_layernorm_bwd_workspace_bytes", m, c)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    nb = ctypes.c_int32(0)
    _lib.call("seg3d_layernorm_bwd_partials", p(dy), p(x), p(mean), p(rstd), p(gamma), p(None), m, c, p(dx), p(ws), nbytes,
              ctypes.byref(nb), st())
    assert nb.value > 0
    fk.add_reduce(ws, nb.value, 2 * c, c, dg.data_ptr(), dbe.data_ptr())
    assert len(fk.jobs) == 5
    fk.join()  # 5 jobs -> one seg3d_reduce_partials_batched launch
    assert not fk.jobs
    torch.cuda.synchronize()
    for (dw, db), (dw2, db2) in zip(want, got):
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert float(got[2][0].abs().max()) == 0.0 and float(got[2][1].abs().max()) == 0.0  # m = 0
    xh = ((x - mean[:, None]) * rstd[:, None]).double()
    assert float((dg.double() - (dy.double() * xh).sum(0)).abs().max()) < 1e-3
    assert float((dbe.double() - dy.double().sum(0)).abs().max()) < 1e-3
    # one and two jobs take the single-launch entry
    dw3 = torch.full_like(want[1][0], float("nan"))
    x, dy = torch.randn(777, 192, device=dev), torch.randn(777, 192, device=dev)
    ref = torch.empty_like(dw3)
    nbytes = _lib.query("seg3d_linear_wgrad_workspace_bytes", 777, 192, 192)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    _lib.call("seg3d_linear_wgrad", p(x), p(dy), 777, 192, 192, p(ref), p(None), p(ws), nbytes, st())
    ops._linear_wgrad_into(fk, x, dy, 192, 192, dw3.data_ptr(), 0)
    fk.join()
    torch.cuda.synchronize()
    assert torch.equal(ref, dw3)


@pytest.mark.parametrize("m,cin,cout", [(5000, 48, 96), (12345, 96, 192), (700, 192, 384), (300, 384, 768), (33, 16, 16)])
def test_linear_with_summed_operand_matches_the_materialised_sum(m, cin, cout):
    """seg3d_linear_fwd_sum (y = (x + x_add) W^T + b, the sum taken on the A operand's way into the split; the inference
    in-projection's q | k half) against seg3d_linear_fwd on the materialised x + x_add: the same float32 sum, the same split,
    the same products -- bit-identical."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m)
    x, pos = torch.randn(m, cin, device=dev), torch.randn(m, cin, device=dev)
    w, b = torch.randn(cout, cin, device=dev) / cin ** 0.5, torch.randn(cout, device=dev)
    packed = ops._linear_pack(w, 0)
    want = ops._linear_apply(x + pos, packed, b, cin, cout)
    got = torch.full_like(want, float("nan"))
    _lib.call("seg3d_linear_fwd_sum", ops._ptr(x), ops._ptr(pos), m, ops._ptr(packed), ops._ptr(b), cin, cout, ops._ptr(got),
              ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # and through the in-projection: no_grad takes the summed-operand path, grad mode the materialised one
    w_in, b_in = torch.randn(3 * cin, cin, device=dev) / cin ** 0.5, torch.randn(3 * cin, device=dev)
    if cin % 16 == 0:
        with torch.no_grad():
            qk0, v0 = ops.attn_in_proj(x, pos, w_in, b_in)
        qk1, v1 = ops.attn_in_proj(x.clone().requires_grad_(), pos, w_in, b_in)
        assert torch.equal(qk0, qk1.detach()) and torch.equal(v0, v1.detach())


def test_gelu_with_derivative_and_multiplying_epilogue():
    """seg3d_gelu_fwd (g and d g / d h in one pass) against torch's erf-form GELU and its autograd derivative in float64;
    seg3d_linear_fwd_mul (y = (x W^T) * factor) against the unfused pair, bit for bit; and the two together = what
    torch.ops.aten.gelu_backward computes in the encoder layer's backward."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    m, c, hid = 7001, 96, 192
    h = (torch.randn(m, hid, device=dev) * 3.0)
    h[0, :8] = torch.tensor([0.0, -0.0, 1e-8, -1e-8, 12.0, -12.0, 40.0, -40.0], device=dev)
    g, gp = torch.empty_like(h), torch.empty_like(h)
    _lib.call("seg3d_gelu_fwd", ops._ptr(h), h.numel(), ops._ptr(g), ops._ptr(gp), ops._stream())
    hr = h.double().cpu().requires_grad_()
    gr = torch.nn.functional.gelu(hr)
    gr.sum().backward()
    assert float((g.cpu().double() - gr.detach()).abs().max()) < 2e-6 * 40
    assert float((gp.cpu().double() - hr.grad).abs().max()) < 2e-6
    g_only = torch.empty_like(h)
    _lib.call("seg3d_gelu_fwd", ops._ptr(h), h.numel(), ops._ptr(g_only), ops._ptr(None), ops._stream())
    assert torch.equal(g_only, g)
    dm = torch.randn(m, c, device=dev)
    w2 = torch.randn(c, hid, device=dev) / hid ** 0.5  # fc2.weight [c, hid]; its input gradient is dm @ w2
    packed_t = ops._linear_pack(w2, 1)
    plain = ops._linear_apply(dm, packed_t, None, c, hid)
    fused = torch.full_like(plain, float("nan"))
    _lib.call("seg3d_linear_fwd_mul", ops._ptr(dm), m, ops._ptr(packed_t), ops._ptr(gp), c, hid, ops._ptr(fused), ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(fused, plain * gp)
    want = torch.ops.aten.gelu_backward(plain, h)
    assert float((fused - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("m,cin,cout", [(5000, 48, 48), (12345, 192, 96), (7000, 96, 96), (3001, 384, 192), (2500, 192, 192), (77, 16, 16)])
def test_linear_with_layernorm_epilogue(m, cin, cout):
    """seg3d_linear_layernorm_fwd (y = res + LayerNorm(x W^T + b), one launch: the inference encoder layer's out-projection +
    norm1 + residual and fc2 + norm2 + residual) against the two library passes it replaces (same split products, LayerNorm
    statistics summed in another order: 1e-5) and against float64; widths that do not fit one workgroup are refused."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(m + cout)
    x, res = torch.randn(m, cin, device=dev), torch.randn(m, cout, device=dev)
    w, b = torch.randn(cout, cin, device=dev) / cin ** 0.5, torch.randn(cout, device=dev)
    ln = torch.nn.LayerNorm(cout).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_()
    packed = ops._linear_pack(w, 0)
    for r in (res, None):
        got = ops._linear_layernorm(x, packed, b, r, ln.weight, ln.bias, ln.eps, cin, cout)
        with torch.no_grad():
            two = ops.layer_norm_residual(ops._linear_apply(x, packed, b, cin, cout), r, ln)
            ref = torch.nn.functional.layer_norm(x.double() @ w.double().t() + b.double(), (cout,), ln.weight.double(), ln.bias.double(), ln.eps)
            ref = ref if r is None else ref + r.double()
        assert float((got - two).abs().max()) < 1e-5 * max(1.0, float(two.abs().max()))
        assert float((got.double() - ref).abs().max()) < 2e-4
    y = torch.empty(m, 384, device=dev)
    w2 = torch.randn(384, cin, device=dev)
    with pytest.raises(_lib.Seg3dError):
        _lib.call("seg3d_linear_layernorm_fwd", ops._ptr(x), m, ops._ptr(ops._linear_pack(w2, 0)), ops._ptr(None), ops._ptr(None),
                  ops._ptr(torch.ones(384, device=dev)), ops._ptr(torch.zeros(384, device=dev)), 1e-5, cin, 384, ops._ptr(y),
                  ops._stream())


@pytest.mark.parametrize("m,cin,cout", [(5000, 48, 48), (5003, 48, 96), (10007, 96, 48), (70001, 96, 96), (33333, 96, 192),
                                        (121168, 192, 96), (58453, 192, 192), (58453, 192, 384), (4099, 192, 576), (15, 192, 192),
                                        (1, 48, 144), (20000, 128, 96), (20000, 64, 48)])
def test_row_streaming_linear_is_bit_identical_to_the_gather_gemm_case(m, cin, cout):
    """(The schedule is OFF by default -- it lost the A/B, csrc/linear_stream.hip -- and is pinned here through the debug
    switch so that the opt-in stays correct.)  Round 5's schedule of the dense Linear layers with cin <= 192 (csrc/linear_stream.hip: W block resident in LDS, every
    wave streams its own 16-row tiles with a rolling prefetch, quad-transposed 16-byte stores) against the schedule it
    replaces (the gather-GEMM's single-offset case, forced through the column-width switch): same products in the same order,
    so the results agree BIT FOR BIT -- plain, with bias, with a residual addend and with the elementwise factor of fc2's
    input gradient (seg3d_linear_fwd_mul) -- and both agree with fp64 within the split-bf16 tolerance."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(m * 7 + cin + cout)
    x = torch.randn(m, cin, generator=gen).to(dev)
    w = (torch.randn(cout, cin, generator=gen) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=gen).to(dev)
    add = torch.randn(m, cout, generator=gen).to(dev)
    packed = ops._linear_pack(w, False)
    old_nbt = {48: 3, 96: 6, 144: 3, 192: 12, 384: 12, 576: 12}[cout]

    def run(bias, addend, mul):
        y = torch.full((m, cout), float("nan"), device=dev)
        if mul:
            _lib.call("seg3d_linear_fwd_mul", ops._ptr(x), m, ops._ptr(packed), ops._ptr(addend), cin, cout, ops._ptr(y), ops._stream())
        else:
            _lib.call("seg3d_linear_fwd", ops._ptr(x), m, ops._ptr(packed), ops._ptr(bias), ops._ptr(addend), cin, cout, ops._ptr(y),
                      ops._stream())
        return y

    for bias, addend, mul in ((None, None, False), (b, None, False), (b, add, False), (None, add, True)):
        _lib.call("seg3d_debug_set_linear_stream", 1)
        try:
            new = run(bias, addend, mul)
        finally:
            _lib.call("seg3d_debug_set_linear_stream", -1)  # back to the environment's choice
        ops.debug_set_conv_nbt(old_nbt)  # a forced column width keeps the dense case on the gather-GEMM kernel
        try:
            old = run(bias, addend, mul)
        finally:
            ops.debug_set_conv_nbt(0)
        assert torch.equal(new, old), (bias is not None, addend is not None, mul, float((new - old).abs().max()))
        ref = x.double() @ w.double().t()
        if bias is not None:
            ref = ref + b.double()
        if addend is not None:
            ref = ref * add.double() if mul else ref + add.double()
        assert float((new.double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,cin,cout", [(174633, 128, 256), (70001, 256, 64), (5000, 64, 128), (4099, 96, 256), (257, 256, 128),
                                        (255, 64, 64), (1, 32, 64), (20000, 128, 64)])
def test_six_product_linear_is_fp32_grade(m, cin, cout):
    """The per-point MLPs' forward (csrc/linear_x6.hip): operands split three ways into bf16, six bf16 MFMAs per product.
    Against float64 its error is that of an fp32 evaluation -- the same grade as the v_mfma_f32_16x16x4_f32 kernel it
    replaces and two orders of magnitude under the three-product split -- with operand magnitudes spread over six decades
    (the split is exact whatever the exponent); the eval epilogue (column scale / shift, ReLU) rounds as the separate pass does."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(m + 3 * cin + cout)
    x = (torch.randn(m, cin, generator=gen) * torch.logspace(-3, 3, cin)[None, :]).to(dev)
    w = (torch.randn(cout, cin, generator=gen) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=gen).to(dev)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    mag = (x.double().abs() @ w.double().abs().t()) + b.double().abs()  # what the rounding errors scale with

    packed = torch.empty((_lib.query("seg3d_linear_packed_bytes_x6", cin, cout),), dtype=torch.uint8, device=dev)
    _lib.call("seg3d_linear_pack_weight_x6", ops._ptr(w), cin, cout, 0, ops._ptr(packed), ops._stream())
    y6 = ops._linear_apply_x6(x, packed, b, cin, cout)
    y32 = ops._linear_apply_f32(x, ops._linear_pack_f32(w, 0), b, cin, cout) if cout % 16 == 0 and cin % 16 == 0 else None
    y3 = ops._linear_apply(x, ops._linear_pack(w, False), b, cin, cout)
    e6 = float(((y6.double() - ref).abs() / mag).max())
    e3 = float(((y3.double() - ref).abs() / mag).max())
    assert e6 < 1e-6, e6  # fp32 grade: a few 2^-24 of the sum of magnitudes
    assert e3 > 10 * e6, (e3, e6)  # (the three-product split sits at 2^-16)
    if y32 is not None:
        e32 = float(((y32.double() - ref).abs() / mag).max())
        assert e6 < 4 * e32 + 1e-8, (e6, e32)
    # through ops.linear(exact=True), the route of the per-point MLPs
    assert ops._x6_fits(cin, cout)
    with torch.no_grad():
        via = ops.linear(x, torch.nn.Parameter(w), torch.nn.Parameter(b), exact=True)
    assert torch.equal(via, y6)
    # eval epilogue: y * scale + shift, ReLU -- bit for bit the separate passes
    s = (torch.rand(cout, generator=gen) + 0.5).to(dev)
    t = torch.randn(cout, generator=gen).to(dev)
    plain = ops._linear_apply_x6(x, packed, None, cin, cout)
    fused = ops._linear_apply_x6(x, packed, None, cin, cout, scale=s, shift=t, relu=True)
    assert torch.equal(fused, torch.relu(plain * s + t))
    # the transposed pack reads W^T: same stream as packing the materialised transpose
    if cin % 64 == 0 and cout % 32 == 0:
        pt = torch.empty((_lib.query("seg3d_linear_packed_bytes_x6", cout, cin),), dtype=torch.uint8, device=dev)
        pm = torch.empty_like(pt)
        _lib.call("seg3d_linear_pack_weight_x6", ops._ptr(w), cout, cin, 1, ops._ptr(pt), ops._stream())
        wt = w.t().contiguous()
        _lib.call("seg3d_linear_pack_weight_x6", ops._ptr(wt), cout, cin, 0, ops._ptr(pm), ops._stream())
        assert torch.equal(pt, pm)


def test_eval_point_mlp_runs_linear_batchnorm_relu_as_one_launch():
    """FusedMLP in eval mode hands Linear -> BatchNorm1d -> ReLU to the six-product kernel's epilogue: bit-identical to the
    Linear followed by the folded-affine pass, and the training-mode forward (batch statistics) is untouched."""
    from openseg3d_amd import ops, segformer
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    mlp = segformer._bn_mlp([96, 256, 128, 64]).to(dev)
    for mod in mlp.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_()
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_()
    x = torch.randn(30011, 96, device=dev)
    mlp.eval()
    with torch.no_grad():
        fused = mlp(x)
        ref = x
        mods = list(mlp)
        for i in range(0, len(mods), 3):
            ref = ops.batch_norm_act(ops.linear(ref, mods[i].weight, None, exact=True), mods[i + 1], relu=True)
    assert torch.equal(fused, ref)
    import copy
    want = torch.nn.Sequential(*[copy.deepcopy(m) for m in mlp]).double().eval()(x.double())
    assert float((fused.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    mlp.train()
    y = mlp(x.requires_grad_())
    y.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


@pytest.mark.parametrize("m,cin,cout", [(58453, 192, 192), (40001, 96, 192), (20011, 384, 384), (9999, 192, 384), (12345, 384, 768),
                                        (4096, 96, 192), (5000, 288, 576)])
def test_lds_shared_dense_weight_gradient_matches_fp64_and_is_reproducible(m, cin, cout):
    """The opt-in LDS-shared form of the Linear weight gradient (csrc/wgrad_dense_lds.hip: OFF by default, it did not win; pinned
    through the debug switch): dW and db against float64 within the split-bf16 tolerance, bit-identical run to run (partial
    blocks + fixed-order sum, no atomics), ragged last steps and a last chunk shorter than the others."""
    from openseg3d_amd import _lib, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(m + cin)
    x = torch.randn(m, cin, generator=gen).to(dev)
    dy = torch.randn(m, cout, generator=gen).to(dev)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)

    def run():
        dw = torch.full((cout, cin), float("nan"), device=dev)
        db = torch.full((cout,), float("nan"), device=dev)
        nb = _lib.query("seg3d_linear_wgrad_workspace_bytes", m, cin, cout)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.call("seg3d_linear_wgrad", ops._ptr(x), ops._ptr(dy), m, cin, cout, ops._ptr(dw), ops._ptr(db), ops._ptr(ws), nb,
                  ops._stream())
        return dw, db

    old_w, old_b = run()
    _lib.call("seg3d_debug_set_wgrad_lds", 1)
    try:
        dw, db = run()
        dw2, db2 = run()
    finally:
        _lib.call("seg3d_debug_set_wgrad_lds", -1)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    scale = float(ref_w.abs().max())
    assert float((dw.double() - ref_w).abs().max()) < 1e-4 * scale
    assert float((old_w.double() - ref_w).abs().max()) < 1e-4 * scale
    assert float((db.double() - ref_b).abs().max()) < 1e-4 * float(ref_b.abs().max())
