"""CPU experiment, zero GPU minutes (VERDICT r4 item 3 i): which operand splits could the window-attention kernels afford?

The GPU path multiplies fp32 operands as split bf16 (x = hi + lo, three bf16 MFMAs per product).  In the attention core
the split of P (and dS in the backward) is paid per ELEMENT and is a third of the kernels' vector work.  fp16 has the bf16
MFMA rate on gfx950 and 11 significant bits instead of 8; the attention operands are range-bounded (q^, k^ in [-1, 1],
P in [0, 1]), so this script emulates on the CPU oracle -- same scene, same weights as bench.py's parity run -- what the
headline logits would do under

    bf16x3        today's arithmetic everywhere (convs, voxel-path Linear layers, attention): the emulation's own check
                  against the 6.0e-4 the GPU measures
    p_bf16_hi     P as ONE bf16 (round 3's falsified two-product P.V: the GPU read 1.4e-3) -- second check of the emulation
    attn_fp16x3   q^, k^, P, V as fp16 hi | lo, three products (V under a per-window power-of-two scale)
    p_fp16        q^, k^ fp16x3; P as ONE fp16; V fp16 hi | lo: two products in P.V, no split of P
    p_fp16_qk_bf16  only P / V change (scores stay bf16x3): isolates the P term

Convs and the voxel-path Linear layers stay bf16x3 in every arm (activations are not range-bounded: fp16 would need a
per-tensor scale the gather-GEMM cannot apply per row).  Prints max |logit - exact oracle| per arm.

    python tools/attn_split_emulation.py [n_points (0 = whole scene)] [arms...]
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import index_ops, model as omodel, sparse_conv as sc, window as win  # noqa: E402
from openseg3d_amd import config, scene, segformer  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def split(x, kind):
    hi = x.to(DT[kind]).float()
    lo = (x - hi).to(DT[kind]).float()
    return hi, lo


def mm3(a, b, kind="bf16", bmm=False):
    """a @ b as hi.hi + hi.lo + lo.hi of the split operands (fp32 accumulation, as the MFMAs do)."""
    ah, al = split(a, kind)
    bh, bl = split(b, kind)
    mm = torch.bmm if bmm else torch.matmul
    return mm(ah, bh) + mm(ah, bl) + mm(al, bh)


def linear3(x, w, b=None):
    y = mm3(x, w.t())
    return y if b is None else y + b


class _SplitF:
    """torch.nn.functional with `linear` on split products (the voxel-path Linear layers of oracle.window)."""

    def __getattr__(self, name):
        return linear3 if name == "linear" else getattr(F, name)


def apply_rulebook3(x, nbr, weight, bias=None):
    wk = sc.kernel_matrices(weight)
    nbr = torch.as_tensor(nbr, dtype=torch.int64)
    out = torch.zeros((nbr.shape[1], wk.shape[2]), dtype=x.dtype)
    xh, xl = split(x, "bf16")
    for k in range(27):
        rows = torch.nonzero(nbr[k] >= 0).view(-1)
        if rows.numel():
            wh, wl = split(wk[k], "bf16")
            src = nbr[k, rows]
            out[rows] += xh[src] @ wh + xh[src] @ wl + xl[src] @ wh
    if bias is not None:
        out = out + bias
    return out


def make_attention(arm):
    qk_kind = {"bf16x3": "bf16", "p_bf16_hi": "bf16", "attn_fp16x3": "fp16", "p_fp16": "fp16", "p_fp16_qk_bf16": "bf16"}[arm]

    def cosine_attention(q_in, k_in, v_in, params, prefix, num_heads, key_padding, tau_min=0.01):
        w, b = params[prefix + "in_proj_weight"], params[prefix + "in_proj_bias"]
        t, nwin, c = q_in.shape
        dh = c // num_heads
        wq, wk, wv = w.chunk(3)
        bq, bk, bv = b.chunk(3)
        q = linear3(q_in, wq, bq).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
        k = linear3(k_in, wk, bk).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
        v = linear3(v_in, wv, bv).contiguous().view(t, nwin * num_heads, dh).transpose(0, 1)
        q = F.normalize(q, dim=2)
        k = F.normalize(k, dim=2)
        scale = 1.0 / params[prefix + "tau"].clamp(min=tau_min)
        # the kernels fold log2e / tau into q before the split and start the score accumulator at -bound (fixed maximum)
        s = mm3(q * (scale * 1.4426950408889634), k.transpose(-2, -1), qk_kind, bmm=True)
        bound = float(scale) * 1.4426950408889634
        p = torch.exp2(s - bound)
        pad = key_padding.view(nwin, 1, 1, t).expand(-1, num_heads, -1, -1).reshape(nwin * num_heads, 1, t)
        p = p.masked_fill(pad, 0.0)
        if arm in ("bf16x3", "attn_fp16x3"):
            kind = "bf16" if arm == "bf16x3" else "fp16"
            vs = v
            sc_ = None
            if kind == "fp16":  # per-window power-of-two scale: max |v| -> [2^13, 2^14)
                mx = v.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-30)
                sc_ = torch.exp2(13 - torch.floor(torch.log2(mx)))
                vs = v * sc_
            ph, pl = split(p, kind)
            vh, vl = split(vs, kind)
            o = torch.bmm(ph, vh) + torch.bmm(ph, vl) + torch.bmm(pl, vh)
            rs = (ph + pl).sum(-1, keepdim=True)  # the row sum rides in the same product (a column of ones behind V)
            if sc_ is not None:
                o = o / sc_
        else:
            pk = "bf16" if arm == "p_bf16_hi" else "fp16"
            vk = "bf16" if arm == "p_bf16_hi" else "fp16"
            vs, sc_ = v, None
            if vk == "fp16":
                mx = v.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-30)
                sc_ = torch.exp2(13 - torch.floor(torch.log2(mx)))
                vs = v * sc_
            p1 = p.to(DT[pk]).float()
            vh, vl = split(vs, vk)
            o = torch.bmm(p1, vh) + torch.bmm(p1, vl)
            rs = p1.sum(-1, keepdim=True)
            if sc_ is not None:
                o = o / sc_
        out = (o / rs.clamp(min=1e-30)).transpose(0, 1).contiguous().view(t, nwin, c)
        return linear3(out, params[prefix + "out_proj.weight"], params[prefix + "out_proj.bias"])

    return cosine_attention


def forward(pts, cfg, ds, sd):
    coords, ids = index_ops.voxelize(pts, ds.voxel_size, ds.point_cloud_range)
    batch = {"points": torch.from_numpy(np.pad(pts, ((0, 0), (1, 0)))).float(),
             "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float(),
             "point_voxel_ids": torch.from_numpy(ids).long(), "batch_size": 1,
             "point_id_offset": torch.tensor([float(pts.shape[0])])}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range),
            "batching_info": [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO],
            "window_shape": cfg.MODEL.WINDOW_SHAPE, "depths": cfg.MODEL.DEPTHS, "use_multi_sweeps": False, "use_image_feature": False}
    with torch.no_grad():
        return omodel.segformer_forward(batch, sd, ocfg)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    arms = sys.argv[2:] or ["bf16x3", "p_bf16_hi", "attn_fp16x3", "p_fp16", "p_fp16_qk_bf16"]
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).eval()  # the weights bench.py's parity run starts from
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    pts = scene.make_scene(0)
    if n:
        pts = pts[:n]
    t0 = time.time()
    exact = forward(pts, cfg, ds, sd)
    print(f"exact oracle: {time.time() - t0:.0f} s, max |logit| {float(exact['point_out'].abs().max()):.1f}", flush=True)
    keep = (win.cosine_attention, win.F, sc.apply_rulebook)
    report = {"n_points": int(pts.shape[0]), "max_abs_logit": float(exact["point_out"].abs().max())}
    for arm in arms:
        win.cosine_attention, win.F, sc.apply_rulebook = make_attention(arm), _SplitF(), apply_rulebook3
        try:
            t0 = time.time()
            got = forward(pts, cfg, ds, sd)
        finally:
            win.cosine_attention, win.F, sc.apply_rulebook = keep
        row = {k: float((got[k] - exact[k]).abs().max()) for k in ("point_out", "voxel_out", "aux_voxel_out")}
        report[arm] = row
        print(f"{arm:16s} point {row['point_out']:.3e}  voxel {row['voxel_out']:.3e}  aux {row['aux_voxel_out']:.3e}   ({time.time() - t0:.0f} s)",
              flush=True)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
