"""What would bf16 STORAGE of activations cost in logit accuracy (BASELINE configs[4] asks for bf16)?  Emulation: every
tensor an op of the voxel path hands to the next op (conv / Linear / attention / norm / reduce / gather outputs) is rounded
to bf16 and widened again, arithmetic unchanged (fp32 accumulate, split-bf16 products).  Compared with the fp32-storage
path on the same weights, in total and PER OP FAMILY (which family could take bf16 storage, which cannot):
python tools/bf16_storage_probe.py [--dense]"""
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


GROUPS = {
    "sparse convs (conv / fused conv + BatchNorm + ReLU outputs)": ["_conv_apply", "conv_act"],
    "BatchNorm / LayerNorm outputs": ["layer_norm_residual", "batch_norm_act"],
    "attention (in-projection q|k, v and attention output)": ["attn_in_proj", "window_attention_packed"],
    "Linear layers of the voxel path (out-proj, MLP)": ["_linear_apply"],
    "point path (per-point MLPs, voxel <-> point reduce / gather)": ["_linear_apply_f32", "linear", "segment_reduce", "gather_rows"],
}
NAMES = sorted({n for g in GROUPS.values() for n in g})


def run_with(model, b, names):
    saved = {n: getattr(ops, n) for n in names}
    try:
        for n in names:
            setattr(ops, n, rounded(saved[n]))
        with torch.no_grad():
            return model(dict(b))["point_out"]
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)


def main():
    dev = torch.device("cuda:0")
    dense = "--dense" in sys.argv
    cfg = config.default_cfg()
    if dense:  # BASELINE configs[4]: 2 M points @0.02 m
        cfg.DATASET.POINT_CLOUD_RANGE, cfg.DATASET.VOXEL_SIZE = list(scene.DENSE_RANGE), list(scene.DENSE_VOXEL)
    ds = config.DatasetSpec(cfg)
    pts = scene.make_dense_scene(0) if dense else scene.make_scene(0)
    for tag, fill in (("golden-style weights (fill_by_name)", True), ("default init, seed 0", False)):
        torch.manual_seed(0)
        model = segformer.build_segmentor(cfg, ds)
        if fill:
            params.fill_by_name(model, seed=0)
        model = model.to(dev).eval()
        b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range)
        with torch.no_grad():
            ref = model(dict(b))["point_out"].clone()
        print(f"== {'dense 2 M-point scene' if dense else 'headline scene'}, {tag}: max |logit| {float(ref.abs().max()):.1f}")
        for name, names in [("EVERYTHING", NAMES)] + list(GROUPS.items()):
            got = run_with(model, b, names)
            err = (got - ref).abs()
            flips = float((got.argmax(1) != ref.argmax(1)).float().mean()) * 100
            print(f"   bf16 storage of {name:68s}: max |dlogit| {float(err.max()):.3e}  mean {float(err.mean()):.3e}  "
                  f"arg-max changes {flips:.3f} %", flush=True)


if __name__ == "__main__":
    main()
