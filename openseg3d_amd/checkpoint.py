"""Loading reference checkpoints (tools/train.py:53-68, 251-273: ``torch.save({'model': state_dict, ...})``).

Everything but the sparse-conv weights loads by name: the module tree mirrors the reference's, so the state_dict keys
and shapes are identical (tests/golden/*_keys.json).  The sparse-conv weights are ``spconv`` parameters, and spconv has
shipped two memory layouts for them:

  KRSC  [Cout, kz, ky, kx, Cin]   spconv >= 2.1 (its default since the implicit-GEMM kernels) -- the layout this package keeps
  RSCK  [kz, ky, kx, Cin, Cout]   spconv 1.x and the 2.x "native" algorithm

``convert_spconv_state_dict`` recognises either by shape and returns a state_dict in this package's layout.

UNTESTED against a real spconv checkpoint: spconv is a third-party dependency absent from the reference tree and from this
image (requirements.txt:4, version unpinned), so neither layout -- nor spconv's kernel-offset order (assumed kz-major,
cross-correlation, as torch's conv3d), nor its tap orientation for SparseInverseConv3d -- can be verified offline.  What
is tested (tests/test_host_logic.py) is the shape recognition and that the permutation round-trips.
"""
import os
import warnings

import torch


def _to_krsc(name, w, want_shape):
    cout, kz, ky, kx, cin = want_shape
    if tuple(w.shape) == tuple(want_shape):
        return w
    if tuple(w.shape) == (kz, ky, kx, cin, cout):  # RSCK
        return w.permute(4, 0, 1, 2, 3).contiguous()
    raise ValueError(f"{name}: shape {tuple(w.shape)} is neither KRSC {tuple(want_shape)} nor RSCK {(kz, ky, kx, cin, cout)}")


def convert_spconv_state_dict(state_dict, model):
    """-> new dict: every 5-D sparse-conv weight of ``model`` taken from ``state_dict`` in whichever spconv layout it is
    stored and returned as [Cout, 3, 3, 3, Cin]; all other entries are passed through."""
    from . import spconv
    conv_keys = {f"{prefix}.weight" if prefix else "weight": tuple(m.weight.shape)
                 for prefix, m in model.named_modules() if isinstance(m, spconv._Conv3x3x3)}
    out = {}
    for k, v in state_dict.items():
        out[k] = _to_krsc(k, v, conv_keys[k]) if k in conv_keys and torch.is_tensor(v) and v.dim() == 5 else v
    return out


def load_reference_checkpoint(model, path_or_dict, strict=True):
    """``model.load_state_dict`` for a checkpoint written by the reference's tools/train.py (key 'model') or a bare
    state_dict, with the sparse-conv weights converted to this package's layout first."""
    is_path = isinstance(path_or_dict, (str, bytes, os.PathLike))
    ckpt = torch.load(os.fspath(path_or_dict), map_location="cpu") if is_path else path_or_dict
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and not torch.is_tensor(ckpt["model"]) else ckpt
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}  # DDP-wrapped saves
    converted = convert_spconv_state_dict(sd, model)
    if any(torch.is_tensor(v) and v.dim() == 5 for v in converted.values()):
        # say so at load time rather than only in the module docstring: wrong logits from a real checkpoint would be silent
        warnings.warn("openseg3d_amd.checkpoint: sparse-conv weights are ASSUMED to be stored KRSC [Cout,kz,ky,kx,Cin] or "
                      "RSCK [kz,ky,kx,Cin,Cout] with kz-major offsets (spconv 2.x); this converter has never been checked "
                      "against a file written by spconv itself -- verify a few logits against the source framework",
                      stacklevel=2)
    return model.load_state_dict(converted, strict=strict)
