"""Oracle restatement of the spconv call sites + torch_scatter -- TEST INFRASTRUCTURE ONLY.

``spconv`` (requirements.txt:4, "spconv v2.x" docs/INSTALL.md:9, version
unpinned) and ``torch_scatter`` (requirements.txt:8) are third-party packages
absent from /root/reference: PARITY UNPINNED against the packages themselves.
The restatement follows the reference's call sites
(seg3d/utils/spconv_utils.py:13-32, pointtransformer.py:13-113,184-189) and the
published definitions: submanifold conv keeps the active set; SparseConv3d is a
regular strided conv restricted to active inputs; SparseInverseConv3d swaps
in/out of the paired rulebook.  It is cross-checked against dense
torch.nn.functional.conv3d / conv_transpose3d in tests/test_oracle_sparse_conv.py.

Weight layout (build-defined, believed to be spconv-2.x KRSC): [Cout, 3, 3, 3, Cin];
kernel offset k = (kz*3 + ky)*3 + kx, neighbour site = site + (kz-1, ky-1, kx-1)
(cross-correlation, as torch's dense conv3d).
"""
import numpy as np
import torch

from . import index_ops


class Sites:
    """Active sites of one resolution level: coords int32 [M,4] (b,z,y,x) + spatial shape (z,y,x)."""

    def __init__(self, coords, spatial_shape):
        self.coords = np.ascontiguousarray(coords, dtype=np.int32)
        self.shape = np.asarray(spatial_shape, dtype=np.int32)
        self._subm = None
        self._down = None

    def subm(self):
        if self._subm is None:
            self._subm = index_ops.rulebook_subm(self.coords, self.shape)
        return self._subm

    def down(self):
        """-> (Sites of the stride-2 level, nbr_fwd [27,M_out], nbr_inv [27,M_in])."""
        if self._down is None:
            co, so = index_ops.downsample_coords(self.coords, self.shape)
            fwd, inv = index_ops.rulebook_strided(self.coords, self.shape, co, so)
            self._down = (Sites(co, so), fwd, inv)
        return self._down


def kernel_matrices(weight):
    """[Cout,3,3,3,Cin] -> [27, Cin, Cout]."""
    cout, cin = weight.shape[0], weight.shape[-1]
    return weight.reshape(cout, 27, cin).permute(1, 2, 0).contiguous()


def apply_rulebook(x, nbr, weight, bias=None):
    """out[r] = sum_k x[nbr[k, r]] @ W_k  (+ bias), accumulated in ascending k."""
    wk = kernel_matrices(weight)
    nbr = torch.as_tensor(nbr, dtype=torch.int64)
    out = torch.zeros((nbr.shape[1], wk.shape[2]), dtype=x.dtype)
    for k in range(27):
        rows = torch.nonzero(nbr[k] >= 0).view(-1)
        if rows.numel():
            out[rows] += x[nbr[k, rows]] @ wk[k]
    if bias is not None:
        out = out + bias
    return out


def subm_conv(x, sites, weight, bias=None):
    """SubMConv3d(k=3, padding=1) -- a9."""
    return apply_rulebook(x, sites.subm(), weight, bias)


def strided_conv(x, sites, weight, bias=None):
    """SparseConv3d(k=3, stride=2, padding=1) -- a10.  Returns (features, Sites of the coarse level)."""
    coarse, fwd, _ = sites.down()
    return apply_rulebook(x, fwd, weight, bias), coarse


def inverse_conv(x_coarse, fine_sites, weight, bias=None):
    """SparseInverseConv3d(k=3) paired with fine_sites.down() -- a11."""
    _, _, inv = fine_sites.down()
    return apply_rulebook(x_coarse, inv, weight, bias)


# ----------------------------------------------------------------- torch_scatter.scatter (a7, a25)
def scatter(src, index, reduce="mean", dim_size=None):
    """torch_scatter.scatter(src, index, dim=0, reduce=...): rows = index.max()+1, empty rows 0."""
    index = index.long()
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n, src.shape[1]), dtype=src.dtype)
    if reduce in ("mean", "sum"):
        out.index_add_(0, index, src)
        if reduce == "mean":
            cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
            out = out / cnt[:, None]
        return out
    if reduce == "max":
        neg = torch.full_like(out, float("-inf"))
        red = neg.index_reduce(0, index, src, "amax", include_self=True)
        return torch.where(torch.isinf(red) & (red < 0), torch.zeros_like(red), red)
    raise NotImplementedError(reduce)


def voxel_to_point(feats, ids):
    """VoxelToPoint.__call__, seg3d/ops/voxel_to_point/voxel_to_point.py:4-17 (a24)."""
    out = torch.zeros((ids.shape[0], feats.shape[-1]), dtype=feats.dtype)
    ok = torch.nonzero(ids != -1).view(-1)
    out[ok] = feats[ids[ok]]
    return out


def voxel_avg_pooling(feats, coords, counts):
    """voxel_pooling_forward_cpu, seg3d/ops/voxel_pooling/src/voxel_pooling.cpp:5-24 (a26)."""
    m = counts.shape[0]
    out = torch.zeros((m, feats.shape[1]), dtype=torch.float32)
    ok = (coords >= 0) & (coords < m)
    idx = coords[ok].long()
    out.index_add_(0, idx, feats[ok] / counts[idx].to(feats.dtype)[:, None])
    return out
