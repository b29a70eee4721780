"""a9: the "row image" schedule of the submanifold gather-GEMM (csrc/spconv_tile.hip, seg3d_conv_plan_build +
seg3d_spconv_fwd_tiled) against the per-pair gather kernel (bit for bit: same products, same order per output row) and
against the fp64 oracle, including the tile shapes the plan has to survive: fewer rows than a tile, several samples,
scattered sites whose distinct neighbour rows overflow the LDS image ("direct" tiles), isolated sites."""
import numpy as np
import pytest
import torch

import refcfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _scene_sites(n):
    from oracle import index_ops
    from openseg3d_amd import scene
    coords, _ = index_ops.voxelize(scene.make_scene(0), refcfg.CART_VOXEL, refcfg.CART_RANGE)
    return np.pad(coords[:n], ((0, 0), (1, 0))).astype(np.int32), refcfg.GRID_CART[::-1].tolist(), 1


def _random_sites(n, shape, batch, seed):
    """Uniformly scattered distinct sites."""
    rs = np.random.RandomState(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rs.choice(cells, size=n, replace=False)
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32), list(shape), batch


def _plan_arrays(plan, m):
    """Decode the plan buffer as csrc/spconv_tile.hip lays it out (256-B aligned pieces)."""
    raw = plan.data.cpu().numpy()
    nt = (m + 127) // 128
    off = 0

    def take(count, dtype):
        nonlocal off
        nbytes = count * np.dtype(dtype).itemsize
        a = raw[off:off + nbytes].view(dtype)
        off += (nbytes + 255) // 256 * 256
        return a

    return dict(row_order=take(nt * 128, np.int32).reshape(nt, 128), ucount=take(nt, np.int32),
                tilemask=take(nt, np.uint32), blkmask=take(nt * 32, np.uint8).reshape(nt, 32),
                lidx=take(nt * 27 * 128, np.uint16).reshape(nt, 27, 16, 8), uniq=take(nt * 512, np.int32).reshape(nt, 512))


def _image_slot(u):
    pair = u >> 1
    return pair * 16 + (((u & 1) << 3) ^ (pair & 15))


def _random_table(m, per_row, seed):
    """A table that is NOT a submanifold rulebook: `per_row` random input rows per output row at random offsets.  A
    128-row tile then touches ~128 * per_row distinct rows -- more than the 512-row LDS image holds ("direct" tiles),
    which compact Morton tiles of real site sets never do (their halo bounds the count)."""
    rs = np.random.RandomState(seed)
    nbr = np.full((27, m), -1, np.int32)
    for r in range(m):
        ks = rs.choice(27, size=per_row, replace=False)
        nbr[ks, r] = rs.randint(0, m, size=per_row)
    return nbr


_CASES = {"scene": lambda: _scene_sites(20000), "scattered": lambda: _random_sites(6000, [12, 40, 40], 1, 1),
          "tiny": lambda: _random_sites(37, [4, 6, 6], 1, 2), "batch3": lambda: _random_sites(3000, [6, 30, 30], 3, 3)}


@pytest.mark.parametrize("case", ["scene", "scattered", "tiny", "batch3", "random_table"])
def test_tile_plan_describes_the_table(dev, case):
    """row_order is a permutation of the rows; every tile's distinct-row list, image slots, block masks and tile mask say
    exactly what the neighbour table says."""
    from openseg3d_amd import ops, spconv
    if case == "random_table":
        coords, shape, bs = _random_sites(1500, [8, 30, 30], 1, 7)
        nbr = torch.from_numpy(_random_table(1500, 9, 8)).to(dev)
        lvl = spconv.SiteLevel(torch.from_numpy(coords).to(dev), shape, bs)
    else:
        coords, shape, bs = _CASES[case]()
        lvl = spconv.SiteLevel(torch.from_numpy(coords).to(dev), shape, bs)
        nbr = lvl.subm()
    plan = ops.ConvPlan(lvl.coords, nbr)
    torch.cuda.synchronize()
    m = coords.shape[0]
    p, tab = _plan_arrays(plan, m), nbr.cpu().numpy()
    order = p["row_order"].reshape(-1)
    assert np.array_equal(np.sort(order[order >= 0]), np.arange(m)) and (order >= 0).sum() == m
    n_direct = 0
    for t in range(p["row_order"].shape[0]):
        rows = p["row_order"][t]
        ent = np.where(rows[None, :] >= 0, tab[:, np.maximum(rows, 0)], -1)  # [27, 128] in tile-position order
        uniq_ref = np.unique(ent[ent >= 0])
        assert p["ucount"][t] == uniq_ref.size
        blk = (ent >= 0).reshape(27, 8, 16).any(2)
        assert np.array_equal(p["blkmask"][t, :27], (blk * (1 << np.arange(8))).sum(1).astype(np.uint8))
        assert p["tilemask"][t] == int((blk.any(1) * (1 << np.arange(27, dtype=np.int64))).sum())
        if uniq_ref.size > 512:
            n_direct += 1
            continue
        uniq = p["uniq"][t, :uniq_ref.size]
        assert np.array_equal(np.sort(uniq), uniq_ref)
        slot_of = {int(r): _image_slot(u) for u, r in enumerate(uniq)}
        want = np.array([[slot_of[int(v)] if v >= 0 else 4096 for v in ent[k]] for k in range(27)])  # [27, 128]
        got = p["lidx"][t].transpose(0, 2, 1).reshape(27, 128)  # [k][c16][rb] -> [k][rb * 16 + c16]
        assert np.array_equal(got, want)
        # rows of a tile are sorted by neighbour mask, padding last
        mask = ((ent >= 0) * (1 << np.arange(27, dtype=np.int64))[:, None]).sum(0)
        key = np.where(rows >= 0, mask, 1 << 40)
        assert np.all(np.diff(key) >= 0)
    assert (n_direct > 0) == (case == "random_table"), n_direct


@pytest.mark.parametrize("case,cin,cout", [("scene", 192, 192), ("scene", 384, 384), ("scene", 192, 96), ("scene", 96, 96),
                                           ("scene", 64, 48), ("scene", 48, 32), ("scene", 96, 48), ("scattered", 96, 192),
                                           ("scattered", 48, 48), ("tiny", 64, 96), ("batch3", 96, 192), ("batch3", 48, 32)])
def test_tiled_conv_is_bit_identical_to_the_per_pair_gather(dev, monkeypatch, case, cin, cout):
    """Forward (plain, and the inference block form with bias + residual + ReLU) and input gradient."""
    from openseg3d_amd import ops, spconv
    coords, shape, bs = _CASES[case]()
    m = coords.shape[0]
    torch.manual_seed(cin + 3 * cout)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=True).to(dev)
    x = torch.randn(m, cin, device=dev)
    res = torch.randn(m, cout, device=dev)
    g = torch.randn(m, cout, device=dev)
    bn = torch.nn.BatchNorm1d(cout, eps=1e-3).to(dev).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_()
        bn.bias.normal_()

    def run(tiled):
        monkeypatch.setattr(ops, "CONV_TILED", tiled)
        lvl = spconv.SiteLevel(torch.from_numpy(coords).to(dev), shape, bs)
        assert (lvl.subm_plan() is not None) == tiled
        xin = x.clone().requires_grad_()
        t = spconv.SparseConvTensor(xin, lvl.coords, shape, bs, _level=lvl)
        y = conv(t).features
        y.backward(g)
        with torch.no_grad():
            fused = conv.forward_bn_act(t.replace_feature(x), bn, relu=True, res=res).features
        return y.detach().clone(), xin.grad.clone(), fused.clone()

    ref, new, again = run(False), run(True), run(True)
    # 48- and 32-column outputs: the four waves of a workgroup split the tile's offsets and sum their partial tiles at the
    # end -- deterministic, but another summation order than the per-pair kernel's (a few ulp of the largest term)
    split = {"forward": cout in (32, 48), "input gradient": cin in (32, 48), "conv + bn + residual + relu": cout in (32, 48)}
    for a, b, c, name in zip(ref, new, again, ("forward", "input gradient", "conv + bn + residual + relu")):
        assert torch.equal(b, c), name
        if split[name]:
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), (name, float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("cin,cout", [(96, 192), (64, 96), (48, 48), (96, 32)])
def test_direct_tiles_are_bit_identical_too(dev, cin, cout):
    """Tiles whose distinct input rows overflow the LDS image are run offset by offset; same results."""
    from openseg3d_amd import ops
    m = 1500
    coords, _, _ = _random_sites(m, [8, 30, 30], 1, 7)
    nbr = torch.from_numpy(_random_table(m, 9, 8)).to(dev)
    plan = ops.ConvPlan(torch.from_numpy(coords).to(dev), nbr)
    torch.manual_seed(cin + cout)
    w = torch.randn(cout, 3, 3, 3, cin, device=dev) / (27 * cin) ** 0.5
    x, bias, res = torch.randn(m, cin, device=dev), torch.randn(cout, device=dev), torch.randn(m, cout, device=dev)
    packed = ops.pack_weight(w, ops.PACK_FWD, use_registry=False)
    for kw in (dict(addend=res, relu=True), dict(addend=None, relu=False)):
        b = bias if kw["relu"] else None
        old = ops.conv_act(x, nbr, packed, b, cin, cout, None, plan=None, **kw)
        new = ops.conv_act(x, nbr, packed, b, cin, cout, None, plan=plan, **kw)
        if cout in (32, 48):  # chunk-split layout: another summation order (see above)
            assert float((old - new).abs().max()) <= 2e-6 * float(old.abs().max())
        else:
            assert torch.equal(old, new), float((old - new).abs().max())


@pytest.mark.parametrize("cin,cout", [(96, 192), (64, 48), (48, 48), (48, 32), (96, 48), (192, 128), (384, 96)])
def test_tiled_conv_matches_the_fp64_oracle(dev, cin, cout):
    """The tile kernel against the fp64 oracle DIRECTLY (not against the per-pair kernel it replaces), forward and input
    gradient, one case per wave layout: 192-column (96 -> 192), the offset-split layouts of the narrow layers -- cout 48 / 32
    forward and, through the transposed pack, cin 48 / 32 in the input gradient: the layouts whose summation order is their
    own --, 128-column and 96-column workgroups."""
    from oracle import sparse_conv as sc
    from openseg3d_amd import ops, spconv
    assert ops.CONV_TILED
    coords, shape, bs = _random_sites(5000, [10, 36, 36], 2, 5)
    ref = sc.Sites(coords, shape)
    torch.manual_seed(cin * 1000 + cout)
    x = torch.randn(coords.shape[0], cin, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, 3, 3, 3, cin, dtype=torch.float64) / (27 * cin) ** 0.5).requires_grad_()
    y_ref = sc.subm_conv(x, ref, w)
    g = torch.randn(y_ref.shape, dtype=torch.float64)
    y_ref.backward(g)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w.detach().float())
    xin = x.detach().float().to(dev).requires_grad_()
    out = conv(spconv.SparseConvTensor(xin, torch.from_numpy(coords).to(dev), shape, bs))
    assert out.level.subm_plan() is not None
    from openseg3d_amd import _lib
    assert _lib.load().seg3d_spconv_tiled_supported(cin, cout)  # the forward takes the tile schedule (the input gradient too
    # wherever its own cout = cin is a supported width: 48, 96, 192, 384 here; 64 falls to the per-pair kernel)
    assert float((out.features.detach().cpu().double() - y_ref.detach()).abs().max()) < 1e-4
    out.features.backward(g.float().to(dev))
    assert float((xin.grad.cpu().double() - x.grad).abs().max()) < 1e-4
    assert float((conv.weight.grad.cpu().double() - w.grad).abs().max()) < 1e-4 * max(1.0, float(w.grad.abs().max()))


@pytest.mark.parametrize("cin,cout", [(96, 192), (192, 96), (64, 48), (96, 32), (384, 384)])
def test_tiled_bf16_storage_matches_the_fp32_storage_kernel(dev, monkeypatch, cin, cout):
    """Opt-in bf16 storage of the feature maps through the tile schedule (seg3d_spconv_fwd_tiled_bf16; BASELINE configs[4]):
    accumulation, bias and activation stay float32, so from float32 rows the output is the float32-storage output rounded
    to bf16, and from bf16 rows (their own hi half: only a hi image is staged, two MFMAs per product) it is the
    float32-storage kernel's result on the same, exactly representable rows -- within one bf16 ulp; and the tiled bf16
    kernels agree with the per-pair bf16 kernels bit for bit wherever the tiled layout keeps their summation order."""
    from openseg3d_amd import ops, spconv
    coords, shape, bs = _CASES["scene"]()
    lvl = spconv.SiteLevel(torch.from_numpy(coords).to(dev), shape, bs)
    nbr, plan = lvl.subm(), lvl.subm_plan()
    assert plan is not None
    m = coords.shape[0]
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(m, cin, generator=gen).to(dev)
    w = (torch.randn(cout, 3, 3, 3, cin, generator=gen) / (8 * cin) ** 0.5).to(dev)
    bias = torch.randn(cout, generator=gen).to(dev)
    res = torch.randn(m, cout, generator=gen).to(dev)
    packed = ops.pack_weight(w, ops.PACK_FWD, use_registry=False)
    with torch.no_grad():
        for xin, round_inputs in ((x, False), (x, True), (x.bfloat16(), True)):
            monkeypatch.setattr(ops, "STORAGE_ROUND_INPUTS", round_inputs)
            for addend, relu in ((None, False), (res, True)):
                monkeypatch.setattr(ops, "STORAGE", "fp32")
                x_ref = xin.bfloat16().float() if round_inputs else xin.float()
                ref = ops.conv_act(x_ref, nbr, packed, bias, cin, cout, None,
                                   None if addend is None else addend.bfloat16().float(), relu, plan=plan)
                monkeypatch.setattr(ops, "STORAGE", "bf16")
                got = ops.conv_act(xin, nbr, packed, bias, cin, cout, None, addend, relu, plan=plan)
                old = ops.conv_act(xin, nbr, packed, bias, cin, cout, None, addend, relu, plan=None)
                assert got.dtype == torch.bfloat16 and ref.dtype == torch.float32
                ulp = ref.abs().clamp(min=1e-30) * 2.0 ** -8
                assert bool(((got.float() - ref).abs() <= ulp + 1e-6).all()), (str(xin.dtype), relu)
                if cout not in (32, 48):  # (the chunk-split layouts sum in another order)
                    assert torch.equal(got, old), (str(xin.dtype), relu)
