"""Deterministic, name-keyed parameter fill -- TEST INFRASTRUCTURE ONLY.

A 33 M-parameter state_dict is too large to commit as a fixture, so golden
runs of the reference model (tests/golden/make_golden.py) and the models under
test are both filled by this function: every tensor is generated from a seed
derived from its state_dict key, so two modules with the same key set get the
same weights regardless of construction order.
"""
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def tensor_for(key, shape, seed=0):
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "tau":
        return 0.3 + 0.7 * torch.rand(shape, generator=g)
    if leaf in ("bias", "in_proj_bias"):
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 1:  # norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / fan_in ** 0.5)


@torch.no_grad()
def fill_by_name(module, seed=0):
    """In-place fill of every parameter and buffer of ``module``; returns its state_dict."""
    sd = module.state_dict()
    for key, t in sd.items():
        t.copy_(tensor_for(key, t.shape, seed).to(t.dtype))
    return sd


def state_dict_for(keys_and_shapes, seed=0):
    """Build a CPU state_dict from a {key: shape} mapping (e.g. tests/golden/segformer_keys.json)."""
    return {k: tensor_for(k, s, seed) for k, s in keys_and_shapes.items()}
