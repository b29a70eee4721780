"""CPU stand-ins for ``spconv.pytorch`` and ``torch_scatter`` backed by the oracle
restatement -- TEST INFRASTRUCTURE ONLY.

Used by tests/golden/make_golden.py (build container only) so that the
reference's *own* model code (segformer.py, pointtransformer.py, ...) can run
on the CPU and produce golden vectors; never shipped in, or imported by, the
product package.  Only the API surface the reference touches is provided
(SURVEY.md section 2.3 / 8b).
"""
import math
import types

import numpy as np
import torch
import torch.nn as nn

from . import sparse_conv as sc


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, _sites=None, _indice_dict=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = list(int(s) for s in spatial_shape)
        self.batch_size = batch_size
        self.sites = _sites if _sites is not None else sc.Sites(indices.numpy(), self.spatial_shape)
        self.indice_dict = _indice_dict if _indice_dict is not None else {}

    def replace_feature(self, new_features):
        return SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size,
                                self.sites, self.indice_dict)


class SparseModule(nn.Module):
    pass


class _Conv(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 bias=True, indice_key=None):
        super().__init__()
        assert kernel_size == 3 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.padding, self.indice_key = stride, padding, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, 3, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))


class SubMConv3d(_Conv):
    def forward(self, x):
        assert self.padding == 1
        y = sc.subm_conv(x.features, x.sites, self.weight, self.bias)
        return x.replace_feature(y)


class SparseConv3d(_Conv):
    def forward(self, x):
        assert self.stride == 2 and self.padding == 1
        y, coarse = sc.strided_conv(x.features, x.sites, self.weight, self.bias)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = x
        return SparseConvTensor(y, torch.from_numpy(coarse.coords), coarse.shape, x.batch_size,
                                coarse, x.indice_dict)


class SparseInverseConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    def forward(self, x):
        fine = x.indice_dict[self.indice_key]
        y = sc.inverse_conv(x.features, fine.sites, self.weight, self.bias)
        return SparseConvTensor(y, fine.indices, fine.spatial_shape, x.batch_size, fine.sites,
                                x.indice_dict)


class SparseSequential(SparseModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x


def scatter(src, index, dim=0, reduce="sum"):
    assert dim == 0
    return sc.scatter(src, index, reduce=reduce)


def install(sys_modules):
    """Register ``spconv``, ``spconv.pytorch`` and ``torch_scatter`` stand-ins."""
    sp = types.ModuleType("spconv")
    spt = types.ModuleType("spconv.pytorch")
    for name in ("SparseConvTensor", "SparseModule", "SubMConv3d", "SparseConv3d",
                 "SparseInverseConv3d", "SparseSequential"):
        setattr(spt, name, globals()[name])
    sp.pytorch = spt
    ts = types.ModuleType("torch_scatter")
    ts.scatter = scatter
    sys_modules["spconv"] = sp
    sys_modules["spconv.pytorch"] = spt
    sys_modules["torch_scatter"] = ts
