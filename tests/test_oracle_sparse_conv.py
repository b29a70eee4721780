"""Independent check of the sparse-conv restatement (spconv itself is unavailable offline):
densify a small grid and compare with torch's dense conv3d / conv_transpose3d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_conv as sc


def _random_sites(rs, batch, shape, n):
    z, y, x = shape
    lin = rs.choice(batch * z * y * x, size=n, replace=False)
    b, r = np.divmod(lin, z * y * x)
    zz, r = np.divmod(r, y * x)
    yy, xx = np.divmod(r, x)
    return np.stack([b, zz, yy, xx], 1).astype(np.int32)


def _densify(feats, coords, batch, shape):
    c = feats.shape[1]
    dense = torch.zeros((batch, c) + tuple(shape), dtype=feats.dtype)
    co = torch.from_numpy(coords).long()
    dense[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = feats
    return dense


def _dense_weight(w):  # [Cout,3,3,3,Cin] -> conv3d layout [Cout,Cin,3,3,3]
    return w.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("shape", [(8, 10, 12), (7, 9, 6)])
def test_subm_equals_dense_conv_on_active_sites(shape):
    rs = np.random.RandomState(0)
    torch.manual_seed(0)
    coords = _random_sites(rs, 2, shape, 150)
    sites = sc.Sites(coords, shape)
    x = torch.randn(coords.shape[0], 5, dtype=torch.float64)
    w = torch.randn(7, 3, 3, 3, 5, dtype=torch.float64)
    b = torch.randn(7, dtype=torch.float64)
    y = sc.subm_conv(x, sites, w, b)
    dense = F.conv3d(_densify(x, coords, 2, shape), _dense_weight(w), b, padding=1)
    co = torch.from_numpy(coords).long()
    ref = dense[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]]
    assert torch.allclose(y, ref, atol=1e-10)


@pytest.mark.parametrize("shape", [(8, 10, 12), (7, 9, 6)])
def test_strided_and_inverse_equal_dense(shape):
    rs = np.random.RandomState(1)
    torch.manual_seed(1)
    coords = _random_sites(rs, 2, shape, 90)
    sites = sc.Sites(coords, shape)
    x = torch.randn(coords.shape[0], 4, dtype=torch.float64)
    w = torch.randn(6, 3, 3, 3, 4, dtype=torch.float64)
    y, coarse = sc.strided_conv(x, sites, w)
    oshape = tuple((s + 2 - 3) // 2 + 1 for s in shape)
    assert tuple(coarse.shape) == oshape
    dense = F.conv3d(_densify(x, coords, 2, shape), _dense_weight(w), stride=2, padding=1)
    assert dense.shape[2:] == oshape
    # active output set == receptive-field reachability of the active inputs
    occ = F.conv3d(_densify(torch.ones(coords.shape[0], 1, dtype=torch.float64), coords, 2, shape),
                   torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), stride=2, padding=1)[:, 0] > 0
    got = torch.zeros_like(occ)
    co = torch.from_numpy(coarse.coords).long()
    got[co[:, 0], co[:, 1], co[:, 2], co[:, 3]] = True
    assert torch.equal(got, occ)
    # canonical order: ascending (b, z, y, x)
    key = ((co[:, 0] * oshape[0] + co[:, 1]) * oshape[1] + co[:, 2]) * oshape[2] + co[:, 3]
    assert bool((key[1:] > key[:-1]).all())
    assert torch.allclose(y, dense[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]], atol=1e-10)
    # everything outside the active set is exactly zero in the dense result
    dz = dense.clone()
    dz[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = 0
    assert float(dz.abs().max()) == 0.0

    # inverse conv == transposed conv evaluated on the fine active sites
    wi = torch.randn(3, 3, 3, 3, 6, dtype=torch.float64)  # [Cout=3, k, k, k, Cin=6]
    z = sc.inverse_conv(y, sites, wi)
    wt = wi.permute(4, 0, 1, 2, 3).contiguous()  # conv_transpose3d layout [Cin, Cout, 3,3,3]
    op = tuple(s - ((o - 1) * 2 - 2 + 3) for s, o in zip(shape, oshape))
    dt = F.conv_transpose3d(_densify(y, coarse.coords, 2, oshape), wt, stride=2, padding=1, output_padding=op)
    assert dt.shape[2:] == tuple(shape)
    ci = torch.from_numpy(coords).long()
    assert torch.allclose(z, dt[ci[:, 0], :, ci[:, 1], ci[:, 2], ci[:, 3]], atol=1e-10)


def test_inverse_table_is_transpose_of_forward():
    rs = np.random.RandomState(2)
    coords = _random_sites(rs, 1, (6, 8, 8), 60)
    sites = sc.Sites(coords, (6, 8, 8))
    coarse, fwd, inv = sites.down()
    pairs_f = {(int(fwd[k, o]), o, k) for k in range(27) for o in range(fwd.shape[1]) if fwd[k, o] >= 0}
    pairs_i = {(i, int(inv[k, i]), k) for k in range(27) for i in range(inv.shape[1]) if inv[k, i] >= 0}
    assert pairs_f == pairs_i and len(pairs_f) > 0


def test_subm_table_symmetry_and_centre():
    rs = np.random.RandomState(3)
    coords = _random_sites(rs, 2, (5, 6, 7), 120)
    nbr = sc.Sites(coords, (5, 6, 7)).subm()
    m = coords.shape[0]
    assert np.array_equal(nbr[13], np.arange(m))
    for k in range(27):
        i = np.nonzero(nbr[k] >= 0)[0]
        assert np.array_equal(nbr[26 - k][nbr[k][i]], i)


def test_scatter_and_gather_restatements():
    src = torch.tensor([[1.0, -2.0], [3.0, 4.0], [-5.0, 6.0], [7.0, -8.0]])
    idx = torch.tensor([2, 0, 2, 0])
    assert sc.scatter(src, idx, "max").tolist() == [[7.0, 4.0], [0.0, 0.0], [1.0, 6.0]]
    assert sc.scatter(src, idx, "mean").tolist() == [[5.0, -2.0], [0.0, 0.0], [-2.0, 2.0]]
    ids = torch.tensor([1, -1, 0])
    assert sc.voxel_to_point(src[:2], ids).tolist() == [[3.0, 4.0], [0.0, 0.0], [1.0, -2.0]]
    cnt = torch.tensor([2, 0, 2], dtype=torch.int32)
    assert torch.allclose(sc.voxel_avg_pooling(src, idx.int(), cnt), sc.scatter(src, idx, "mean"))
