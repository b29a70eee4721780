"""What would bf16 STORAGE of activations cost in logit accuracy (BASELINE configs[4] asks for bf16)?  Emulation: every
tensor an op of the voxel path hands to the next op (conv / Linear / attention / norm / reduce / gather outputs) is rounded
to bf16 and widened again, arithmetic unchanged (fp32 accumulate, split-bf16 products).  Compared with the fp32-storage
path on the same weights: python tools/bf16_storage_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from openseg3d_amd import batch as B, config, ops, scene, segformer  # noqa: E402
from oracle import params  # noqa: E402


def rounded(fn):
    def wrap(*a, **k):
        out = fn(*a, **k)
        if isinstance(out, tuple):
            return tuple(o.bfloat16().float() if torch.is_tensor(o) and o.dtype == torch.float32 else o for o in out)
        return out.bfloat16().float() if torch.is_tensor(out) and out.dtype == torch.float32 else out
    return wrap


NAMES = ["_conv_apply", "conv_act", "_linear_apply", "_linear_apply_f32", "window_attention_packed", "layer_norm_residual",
         "batch_norm_act", "segment_reduce", "gather_rows", "attn_in_proj", "linear"]


def main():
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    for tag, fill in (("golden-style weights (fill_by_name, |logit| <= 25)", True), ("default init, seed 0 (|logit| <= 210)", False)):
        torch.manual_seed(0)
        model = segformer.build_segmentor(cfg, ds)
        if fill:
            params.fill_by_name(model, seed=0)
        model = model.to(dev).eval()
        b = B.make_batch([scene.make_scene(0)], ds.voxel_size, ds.point_cloud_range)
        with torch.no_grad():
            ref = model(dict(b))["point_out"].clone()
        saved = {n: getattr(ops, n) for n in NAMES}
        try:
            for n in NAMES:
                setattr(ops, n, rounded(saved[n]))
            with torch.no_grad():
                got = model(dict(b))["point_out"]
        finally:
            for n, f in saved.items():
                setattr(ops, n, f)
        err = (got - ref).abs()
        print(f"{tag}: max |logit| {float(ref.abs().max()):.1f}; bf16 activation storage moves the logits by max {float(err.max()):.3e}, "
              f"mean {float(err.mean()):.3e}; argmax changes on {float((got.argmax(1) != ref.argmax(1)).float().mean()) * 100:.2f} % of the points")


if __name__ == "__main__":
    main()
