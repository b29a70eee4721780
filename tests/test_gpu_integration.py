"""The reference's own call patterns driven through this package after the registration INTEGRATION.md 2.2 / 2.5 shows:
``import spconv.pytorch as spconv`` / ``from torch_scatter import scatter`` resolve to openseg3d_amd, ``ConvModule`` and
``UpBlock`` are built exactly as seg3d/utils/spconv_utils.py:13-32 and seg3d/models/backbones/pointtransformer.py:69-113
build them, ``VoxelGenerator`` is constructed and called as waymo_dataset.py:275 / test_time_aug.py:33 do."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import refcfg

pytestmark = pytest.mark.gpu


@pytest.fixture()
def registered(monkeypatch):
    """INTEGRATION.md 2.2, verbatim: the third-party module names point at this package."""
    import openseg3d_amd.ops as _ops
    import openseg3d_amd.spconv as _sp
    pkg = types.ModuleType("spconv")
    pkg.pytorch = _sp
    monkeypatch.setitem(sys.modules, "spconv", pkg)
    monkeypatch.setitem(sys.modules, "spconv.pytorch", _sp)
    ts = types.ModuleType("torch_scatter")
    ts.scatter = _ops.scatter
    monkeypatch.setitem(sys.modules, "torch_scatter", ts)
    return _sp


def _conv_module(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, conv_type="subm", norm_fn=None,
                 act_fn=None, indice_key=None):
    """seg3d/utils/spconv_utils.py:13-32, argument for argument, on whatever `spconv.pytorch` resolves to."""
    import spconv.pytorch as spconv
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, bias=False,
                                 indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                                   bias=False, indice_key=indice_key)
    else:
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    return spconv.SparseSequential(conv, norm_fn(out_channels), act_fn)


def _replace_feature(out, new_features):  # spconv_utils.py:4-10
    if "replace_feature" in out.__dir__():
        return out.replace_feature(new_features)
    out.features = new_features
    return out


def test_reference_conv_modules_through_registered_spconv(registered, golden_dir):
    """ConvModule(subm) -> ConvModule(spconv, stride 2) -> SubM block at the coarse level -> ConvModule(inverseconv) with
    the shared `indice_key` pairing of UpBlock (pointtransformer.py:79-81) -> skip connection by replace_feature, in eval
    mode (BatchNorm with running statistics), against the oracle's rulebook application in fp64; plus the
    `from torch_scatter import scatter` surface (vfe.py:25, se_layer.py:25)."""
    from functools import partial
    from oracle import sparse_conv as sc
    import spconv.pytorch as spconv
    from torch_scatter import scatter
    assert spconv is registered
    dev = torch.device("cuda:0")
    d = np.load(os.path.join(golden_dir, "segformer_cart.npz"))
    coords, bs = d["voxel_coords"].astype(np.int32), int(d["batch_size"])
    shape = refcfg.GRID_CART[::-1].tolist()
    norm_fn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)  # pointtransformer.py:129
    torch.manual_seed(5)
    stem = _conv_module(16, 48, 3, padding=1, norm_fn=norm_fn, act_fn=torch.nn.ReLU(), indice_key="subm1")
    down = _conv_module(48, 96, 3, stride=2, padding=1, conv_type="spconv", norm_fn=norm_fn, act_fn=torch.nn.ReLU(),
                        indice_key="spconv2")
    mid = _conv_module(96, 96, 3, padding=1, norm_fn=norm_fn, act_fn=torch.nn.ReLU(), indice_key="subm2")
    up = _conv_module(96, 48, 3, conv_type="inverseconv", norm_fn=norm_fn, act_fn=torch.nn.ReLU(), indice_key="spconv2")
    mods = [stem, down, mid, up]
    for m in mods:
        assert isinstance(m, spconv.SparseModule)
        bn = m[1]
        with torch.no_grad():  # running statistics that differ from the initial (0, 1)
            bn.running_mean.normal_(0.0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0.0, 0.1)
        m.to(dev).eval()
    x0 = torch.randn(coords.shape[0], 16)
    x = spconv.SparseConvTensor(features=x0.to(dev), indices=torch.from_numpy(coords).to(dev), spatial_shape=shape,
                                batch_size=bs)
    with torch.no_grad():
        a = stem(x)
        b = down(a)
        c = mid(b)
        e = up(c)
        out = _replace_feature(e, e.features + a.features)  # UpBlock-style skip onto the fine level
    assert tuple(out.features.shape) == (coords.shape[0], 48)
    assert torch.equal(out.indices.cpu(), torch.from_numpy(coords))  # the inverse conv restores the fine sites, in order

    def bn_act(y, bn):
        bn = bn.cpu().double()
        return torch.relu((y - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps) * bn.weight + bn.bias)

    sites = sc.Sites(coords, shape)
    w = [m[0].weight.detach().cpu().double() for m in mods]
    ra = bn_act(sc.subm_conv(x0.double(), sites, w[0]), stem[1])
    rb, coarse = sc.strided_conv(ra, sites, w[1])
    rb = bn_act(rb, down[1])
    assert np.array_equal(b.indices.cpu().numpy(), coarse.coords)
    rc = bn_act(sc.subm_conv(rb, coarse, w[2]), mid[1])
    re = bn_act(sc.inverse_conv(rc, sites, w[3]), up[1]) + ra
    err = float((out.features.cpu().double() - re).abs().max())
    assert err < 2e-4 * max(1.0, float(re.abs().max())), err

    # torch_scatter.scatter as the reference calls it (vfe.py:25: mean over points per voxel; se_layer.py:25)
    ids = torch.from_numpy(d["point_voxel_ids"]).to(dev) if "point_voxel_ids" in d else torch.randint(0, 500, (4000,), device=dev)
    ok = ids >= 0
    src = torch.randn(int(ok.sum()), 8, device=dev)
    for red in ("mean", "max"):
        got = scatter(src, ids[ok], dim=0, reduce=red)
        want = sc.scatter(src.cpu().double(), ids[ok].cpu(), reduce=red)
        assert got.shape == want.shape and float((got.cpu().double() - want).abs().max()) < 1e-5


@pytest.mark.parametrize("tag,rng,vs", [("cart", refcfg.CART_RANGE, refcfg.CART_VOXEL), ("cyl", refcfg.CYL_RANGE, refcfg.CYL_VOXEL)])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_voxel_generator_drop_in(golden_dir, tag, rng, vs, dt):
    """`VoxelGenerator(voxel_size=..., point_cloud_range=...).generate(points)` as waymo_dataset.py:22-23,275 and
    test_time_aug.py:33 use it: numpy in -> (coors int32 [M, 3] zyx, point_voxel_ids int32 [N]) numpy out, bit-exact
    against the reference's own outputs (tests/golden/voxelize.npz), attributes typed as voxel_generator.py:15-22;
    a CUDA tensor stays on the device."""
    from openseg3d_amd.batch import VoxelGenerator
    d = np.load(os.path.join(golden_dir, "voxelize.npz"))
    k = f"{tag}_{dt}"
    gen = VoxelGenerator(voxel_size=vs, point_cloud_range=rng)
    assert gen.voxel_size.dtype == np.float32 and gen.point_cloud_range.dtype == np.float32 and gen.grid_size.dtype == np.int64
    assert gen.grid_size.tolist() == d[tag + "_grid"].tolist()
    coors, ids = gen.generate(d[k + "_points"])
    assert isinstance(coors, np.ndarray) and coors.dtype == np.int32 and ids.dtype == np.int32
    assert np.array_equal(coors, d[k + "_coors"]) and np.array_equal(ids, d[k + "_ids"])
    t_coors, t_ids = gen.generate(torch.from_numpy(d[k + "_points"]).cuda())
    assert t_coors.is_cuda and t_ids.is_cuda
    assert np.array_equal(t_coors.cpu().numpy(), coors) and np.array_equal(t_ids.cpu().numpy(), ids)
    assert "VoxelGenerator(voxel_size=" in repr(gen)
