"""Size-independent properties at BASELINE.json's full size (one synthetic Waymo sweep: 174 633 points, ~109 k voxels at
0.1 m), where the CPU oracle is too slow to be the checker: index structures against closed-form torch recomputation,
the sparse-conv kernels through linearity and adjointness (<conv(x), y> = <x, conv^T(y)> ties forward to dgrad,
<dW, W'> = <y, conv_W'(x)> ties it to wgrad), the attention through row-stochasticity and V-adjointness, kNN through
self-consistency and grid-vs-brute-force equality."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module", params=["one_sweep", "dense2m"])
def full(dev, request):
    """BASELINE configs[1] (one Waymo-shaped sweep, 174 633 points @0.1 m) and configs[4] (2 M points @0.02 m: 1.56 M
    voxels on the same 1440 x 1440 x 64 grid -- ten times the rows through every hash, table, scan and workspace)."""
    from openseg3d_amd import batch as B, config, scene
    cfg = config.default_cfg()
    if request.param == "dense2m":
        cfg.DATASET.POINT_CLOUD_RANGE = list(scene.DENSE_RANGE)
        cfg.DATASET.VOXEL_SIZE = list(scene.DENSE_VOXEL)
        pts = scene.make_dense_scene(0)
    else:
        pts = scene.make_scene(0)
    ds = config.DatasetSpec(cfg)
    b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range)
    yield cfg, ds, pts, b
    del b
    torch.cuda.empty_cache()


def test_fullsize_voxelizer_properties(dev, full):
    from openseg3d_amd import ops
    cfg, ds, pts, b = full
    n = pts.shape[0]
    assert n > 170000
    coords, ids = b["voxel_coords"].int(), b["point_voxel_ids"]
    m = coords.shape[0]
    xyz = b["points"][:, 1:4]
    lo = torch.tensor(ds.point_cloud_range[:3], device=dev)
    hi = torch.tensor(ds.point_cloud_range[3:], device=dev)
    vs = torch.tensor(ds.voxel_size, device=dev)
    cell = torch.floor((xyz - lo) / vs).int()  # the reference's float32 arithmetic (voxel_generator.py)
    grid = torch.tensor([int(g) for g in ds.grid_size], device=dev, dtype=torch.int32)
    inside = ((cell >= 0) & (cell < grid)).all(dim=1)
    assert torch.equal(ids >= 0, inside)                                   # exactly the in-range points get a voxel
    got = coords[ids[inside].long()][:, 1:]                                # (z, y, x) of every point's voxel
    assert torch.equal(got, cell[inside][:, [2, 1, 0]])                    # ... is its own cell: bit-exact
    key = (coords[:, 1].long() * grid[1] + coords[:, 2].long()) * grid[0] + coords[:, 3].long()
    assert torch.unique(key).numel() == m                                  # no voxel twice
    first = torch.full((m,), n, dtype=torch.long, device=dev).scatter_reduce(
        0, ids[inside].long(), torch.nonzero(inside).view(-1), "amin")
    assert bool((first[1:] > first[:-1]).all())                            # voxels numbered in first-appearance order
    # idempotence: the voxel centres voxelise to themselves, one point per voxel, same numbering
    centres = torch.zeros((m, b["points"].shape[1]), device=dev)
    centres[:, 1:4] = (coords[:, [3, 2, 1]].float() + 0.5) * vs + lo
    c2, i2 = ops.voxelize(centres, ds.voxel_size, ds.point_cloud_range, xyz_col=1, batch_col=0)
    assert torch.equal(c2, coords) and torch.equal(i2, torch.arange(m, device=dev, dtype=torch.int32))


def test_fullsize_rulebook_properties(dev, full):
    from openseg3d_amd import spconv
    cfg, ds, pts, b = full
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    for depth in range(3):
        nbr = level.subm()
        m = nbr.shape[1]
        rows = torch.arange(m, device=dev, dtype=torch.int32)
        assert torch.equal(nbr[13], rows)                                  # centre offset = identity
        for k in range(13):                                                # nbr[k][i] = j  <=>  nbr[26-k][j] = i
            i = torch.nonzero(nbr[k] >= 0).view(-1)
            assert torch.equal(nbr[26 - k][nbr[k][i].long()], i.int())
            assert int((nbr[k] >= 0).sum()) == int((nbr[26 - k] >= 0).sum())
        # a neighbour really sits at that offset
        k = 5
        dz, dy, dx = k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1
        i = torch.nonzero(nbr[k] >= 0).view(-1)
        delta = level.coords[nbr[k][i].long()] - level.coords[i]
        assert bool((delta == torch.tensor([0, dz, dy, dx], device=dev, dtype=torch.int32)).all())
        # parity order = stable sort by (z & 1, y & 1, x & 1)
        cz = level.coords
        key = (cz[:, 1] & 1) * 4 + (cz[:, 2] & 1) * 2 + (cz[:, 3] & 1)
        assert torch.equal(level.parity_order(), torch.sort(key, stable=True)[1].int())
        coarse, fwd, inv = level.down()
        # output sites of SparseConv3d(k=3, s=2, p=1): every o with 2o - 1 + k = c for some active c, sorted, unique
        c = level.coords[:, 1:].long()
        shape_out = torch.tensor(coarse.shape, device=dev)
        cand = []
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    o2 = c + 1 - torch.tensor([kz, ky, kx], device=dev)
                    ok = ((o2 % 2 == 0) & (o2 >= 0) & (o2 // 2 < shape_out)).all(dim=1)
                    o = o2[ok] // 2
                    cand.append((o[:, 0] * shape_out[1] + o[:, 1]) * shape_out[2] + o[:, 2])
        want = torch.unique(torch.cat(cand))
        cc = coarse.coords[:, 1:].long()
        got = (cc[:, 0] * shape_out[1] + cc[:, 1]) * shape_out[2] + cc[:, 2]
        assert torch.equal(got, want)                                      # bit-exact, ascending
        for k in (0, 7, 13, 20, 26):                                       # forward and inverse tables are transposes
            j = torch.nonzero(fwd[k] >= 0).view(-1)
            assert torch.equal(inv[k][fwd[k][j].long()], j.int())
            assert int((fwd[k] >= 0).sum()) == int((inv[k] >= 0).sum())
        level = coarse


# (kind, cin, cout, strided levels below the voxel grid the layer lives on): the level-1/2 layers and the deep layers
# of the real net -- conv_down2/3 (pointtransformer.py:159-166), the submanifold convs of up3 / up4 and up4's
# 768 -> 384 bottleneck (:69-113), up3's inverse conv -- each on the scene's own site level
@pytest.mark.parametrize("kind,cin,cout,depth", [("subm", 48, 48, 0), ("down", 48, 96, 0), ("up", 96, 48, 0),
                                                 ("subm", 192, 192, 2), ("subm", 384, 384, 3), ("subm", 768, 384, 3),
                                                 ("down", 192, 384, 2), ("up", 384, 192, 2), ("down", 96, 192, 1)])
def test_fullsize_sparse_conv_linearity_and_adjoints(dev, full, kind, cin, cout, depth):
    from openseg3d_amd import spconv
    cfg, ds, pts, b = full
    torch.manual_seed(7)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    for _ in range(depth):
        level = level.down()[0]
    base = spconv.SparseConvTensor(torch.zeros(level.coords.shape[0], 1, device=dev), level.coords, level.shape, 1, _level=level)
    if kind == "subm":
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="s").to(dev)
        src = base
    elif kind == "down":
        conv = spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key="d").to(dev)
        src = base
    else:
        down = spconv.SparseConv3d(16, 16, 3, stride=2, padding=1, bias=False, indice_key="d").to(dev)
        with torch.no_grad():
            src = down(base.replace_feature(torch.zeros(level.coords.shape[0], 16, device=dev)))
        conv = spconv.SparseInverseConv3d(cin, cout, 3, bias=False, indice_key="d").to(dev)
    m_in = src.features.shape[0]

    def run(x):
        return conv(src.replace_feature(x)).features

    x1 = torch.randn(m_in, cin, device=dev)
    x2 = torch.randn(m_in, cin, device=dev)
    with torch.no_grad():
        y1, y2, y12 = run(x1), run(x2), run(2.0 * x1 - 3.0 * x2)
    scale = float(y12.abs().max())
    assert float((y12 - (2.0 * y1 - 3.0 * y2)).abs().max()) <= 2e-4 * scale       # linear in x (split-bf16 rounding only)
    # adjoint identities through autograd: forward vs input gradient vs weight gradient
    x = x1.clone().requires_grad_(True)
    y = run(x)
    g = torch.randn_like(y)
    y.backward(g)
    lhs = float((y.detach().double() * g.double()).sum())
    # <y, g> is a random-sign sum: its typical size is |y| |g| / sqrt(N); the tolerance is 1e-4 of that (so a sum that
    # happens to land near zero does not fail the test) plus 2e-4 of the value itself
    typical = float(y.detach().double().norm() * g.double().norm()) / float(y.numel()) ** 0.5
    tol = 2e-4 * abs(lhs) + 1e-4 * typical
    assert abs(lhs - float((x1.double() * x.grad.double()).sum())) <= tol
    assert abs(lhs - float((conv.weight.detach().double() * conv.weight.grad.double()).sum())) <= tol
    # only the centre tap set: a submanifold conv degenerates to a per-row Linear layer
    if kind == "subm":
        with torch.no_grad():
            w = conv.weight.clone()
            conv.weight.zero_()
            conv.weight[:, 1, 1, 1, :] = w[:, 1, 1, 1, :]
            yc = run(x1)
            ref = x1.double() @ w[:, 1, 1, 1, :].double().t()
        assert float((yc.double() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


def test_fullsize_window_attention_properties(dev, full):
    """Stage 1 of the headline scene (108 690 tokens, 6 298 windows, narrow heads) and stage 3 shapes on a coarser level
    (wide heads): softmax rows sum to one (V = const -> out = const), <out(V), G> = <V, dV(G)>, zero rows for no token."""
    from openseg3d_amd import ops, spconv, swformer
    cfg, ds, pts, b = full
    torch.manual_seed(5)
    level = spconv.SiteLevel(b["voxel_coords"].int(), [int(g) for g in ds.grid_size][::-1], 1)
    info = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    for stage, c in ((0, 48), (2, 192)):
        lv = level
        for _ in range(stage):
            lv = lv.down()[0]
        part = swformer.SparseWindowPartitionLayer(info[stage], cfg.MODEL.WINDOW_SHAPE, [float(g) / 2 ** stage for g in ds.grid_size])
        plan = part.plan(lv.coords, 1, c)
        m = lv.coords.shape[0]
        tau = torch.full((1, 1, 1), 0.2, device=dev)
        for shift in (0, 1):
            wi = plan.index[shift]
            qk = torch.randn(m, 2 * c, device=dev)
            with torch.no_grad():
                out = ops.window_attention_packed(qk, torch.full((m, c), 1.5, device=dev), tau, 0.01, 8, wi)
            assert float((out - 1.5).abs().max()) <= 1e-4                  # every softmax row sums to 1
            v = torch.randn(m, c, device=dev, requires_grad=True)
            o = ops.window_attention_packed(qk, v, tau, 0.01, 8, wi)
            g = torch.randn_like(o)
            o.backward(g)
            lhs = float((o.detach().double() * g.double()).sum())
            rhs = float((v.detach().double() * v.grad.double()).sum())
            assert abs(lhs - rhs) <= 5e-4 * abs(lhs) + 0.5


def test_fullsize_knn_properties(dev, full, monkeypatch):
    from openseg3d_amd import ops
    cfg, ds, pts, b = full
    xyz = ops.get_voxel_centers(b["voxel_coords"][:, 1:], 1.0, ds.voxel_size, ds.point_cloud_range).contiguous()
    m = xyz.shape[0]
    off = torch.tensor([m], dtype=torch.int32, device=dev)
    idx, dist = ops.knn_query(4, xyz, xyz, off, off)
    assert torch.equal(idx[:, 0], torch.arange(m, device=dev, dtype=torch.int32))   # every site is its own nearest
    assert float(dist[:, 0].max()) == 0.0
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())                                # ascending
    assert float(dist[:, 1].min()) >= ds.voxel_size[0] * (1 - 1e-4)                 # distinct voxel centres are >= one pitch apart
    d_chk = (xyz[idx[:, 3].long()] - xyz).norm(dim=1)
    assert float((d_chk - dist[:, 3]).abs().max()) <= 1e-5                          # reported distance = actual distance
    monkeypatch.setattr(ops, "KNN_GRID_MIN_POINTS", 1 << 40)                        # brute force, same answer bit for bit
    sub = torch.arange(0, m, 9, device=dev)
    q = xyz[sub].contiguous()
    qoff = torch.tensor([q.shape[0]], dtype=torch.int32, device=dev)
    bi, bd = ops.knn_query(4, xyz, q, off, qoff)
    monkeypatch.setattr(ops, "KNN_GRID_MIN_POINTS", 1)
    gi, gd = ops.knn_query(4, xyz, q, off, qoff)
    assert torch.equal(bi, gi) and torch.equal(bd, gd)
    assert torch.equal(gi, idx[sub]) and torch.equal(gd, dist[sub])
