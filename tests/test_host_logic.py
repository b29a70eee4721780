"""Host-side logic that needs no GPU: config loading, model construction / state_dict contract,
window geometry, the synthetic scene generator."""
import json
import os

import numpy as np

import dropout_ref
import pytest
import torch

import refcfg

YAML_CYL = """
DATASET:
  USE_CYLINDER: True
  POINT_CLOUD_RANGE: [0, -3.1415926, -2, 75.2, 3.1415926, 5.2]
  VOXEL_SIZE: [0.05, 0.012, 0.1]
  CLASS_NAMES: ['Car', 'Truck']
TRAIN:
  LR: 0.05
  OPTIMIZER: 'sgd'
  WEIGHT_DECAY: 0.0001
  MOMENTUM: 0.9
"""


def test_yaml_merges_like_the_reference(tmp_path):
    from openseg3d_amd import config
    p = tmp_path / "c.yaml"
    p.write_text(YAML_CYL)
    cfg = config.cfg_from_file(str(p))
    assert cfg.DATASET.USE_CYLINDER is True and cfg.DATASET.VOXEL_SIZE == [0.05, 0.012, 0.1]
    assert cfg.TRAIN.OPTIMIZER == "sgd" and cfg.MODEL.SEGMENTOR == "segformer"
    assert cfg.MODEL.BATCHING_INFO[0]["3"]["max_tokens"] == 800
    bad = tmp_path / "bad.yaml"
    bad.write_text("DATASET:\n  NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        config.cfg_from_file(str(bad))
    bad.write_text("TRAIN:\n  LR: 'fast'\n")
    with pytest.raises(ValueError):
        config.cfg_from_file(str(bad))


def test_default_batching_info_equals_reference_defaults():
    from openseg3d_amd import config
    cfg = config.default_cfg()
    got = [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO]
    assert got == refcfg.BATCHING_INFO
    assert cfg.MODEL.WINDOW_SHAPE == refcfg.WINDOW_SHAPE and cfg.MODEL.DEPTHS == refcfg.DEPTHS


@pytest.mark.parametrize("cyl", [False, True])
def test_state_dict_contract(golden_dir, cyl):
    """Same keys and shapes as the reference Segformer (list written by make_golden.py)."""
    from openseg3d_amd import config, segformer
    cfg = config.default_cfg()
    if cyl:
        cfg.DATASET.USE_CYLINDER = True
        cfg.DATASET.POINT_CLOUD_RANGE, cfg.DATASET.VOXEL_SIZE = refcfg.CYL_RANGE, refcfg.CYL_VOXEL
    ds = config.DatasetSpec(cfg)
    assert ds.grid_size.tolist() == (refcfg.GRID_CYL if cyl else refcfg.GRID_CART).tolist()
    model = segformer.build_segmentor(cfg, ds)
    keys = refcfg.segformer_key_shapes(json.load(open(os.path.join(golden_dir, "segformer_keys.json"))), 8 if cyl else 6)
    sd = model.state_dict()
    assert set(sd) == set(keys)
    assert all(list(sd[k].shape) == list(keys[k]) for k in keys)
    assert abs(sum(p.numel() for p in model.parameters()) - 32.93e6) < 0.05e6


def test_multi_sweep_image_fusion_state_dict_contract(golden_dir):
    """configs/waymo_multi_sweeps.yaml + USE_IMAGE_FEATURE: same keys/shapes as the reference model."""
    from openseg3d_amd import config, segformer
    cfg = config.default_cfg()
    cfg.DATASET.USE_MULTI_SWEEPS = True
    cfg.DATASET.USE_IMAGE_FEATURE = True
    model = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
    keys = json.load(open(os.path.join(golden_dir, "segformer_ms_keys.json")))
    sd = model.state_dict()
    assert set(sd) == set(keys)
    assert all(list(sd[k].shape) == list(keys[k]) for k in keys)
    assert sd["point_transformer.conv_input.0.weight"].shape[-1] == 6  # raw point rows feed the voxel encoder


@pytest.mark.parametrize("tag", ["cart", "ms"])
def test_spnet_state_dict_contract(golden_dir, tag):
    """MODEL.SEGMENTOR='spnet' (builder.py:17-18): same keys and shapes as the reference SPNet/SparseUnet/OCRLayer."""
    from openseg3d_amd import config, segformer
    cfg = config.default_cfg()
    cfg.MODEL.SEGMENTOR = "spnet"
    cfg.DATASET.USE_MULTI_SWEEPS = cfg.DATASET.USE_IMAGE_FEATURE = tag == "ms"
    model = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
    keys = json.load(open(os.path.join(golden_dir, f"spnet_{tag}_keys.json")))
    sd = model.state_dict()
    assert set(sd) == set(keys)
    assert all(list(sd[k].shape) == list(keys[k]) for k in keys)
    assert list(sd) == list(keys)  # registration order too: optimizers index parameters by position


def test_unsupported_configs_are_refused():
    from openseg3d_amd import config, segformer
    cfg = config.default_cfg()
    cfg.MODEL.SEGMENTOR = "minkunet"
    with pytest.raises(NotImplementedError):
        segformer.build_segmentor(cfg, config.DatasetSpec(cfg))


def test_window_geometry_matches_reference_quirks():
    from openseg3d_amd.swformer import window_geometry
    # stage 1 of the cartesian grid: 145 x 145 x 9 windows, full-window "no shift" offset (quirk 4)
    assert window_geometry([1440, 1440, 64], [10, 10, 8], False) == ([10, 10, 8], [145, 145, 9], [10, 10, 8])
    assert window_geometry([1440, 1440, 64], [10, 10, 8], True) == ([10, 10, 8], [145, 145, 9], [5, 5, 4])
    # stage 4: S_z == win_z -> z never shifted (quirk 3)
    assert window_geometry([180, 180, 8], [10, 10, 8], True)[2] == [5, 5, 0]
    assert window_geometry([180, 180, 8], [10, 10, 8], False)[2] == [10, 10, 0]
    # cylinder stage 4 gets a fractional shape (quirk 2): only ceil(S/win)+1 is used
    assert window_geometry([188.0, 65.5, 9.0], [10, 10, 8], False)[1] == [20, 8, 3]


def test_synthetic_scene_is_waymo_shaped_and_seeded():
    from openseg3d_amd import scene
    from oracle import index_ops
    a, b = scene.make_scene(0), scene.make_scene(0)
    assert a.dtype == np.float32 and a.shape[1] == 6 and np.array_equal(a, b)
    assert 150_000 < a.shape[0] < 200_000
    assert not np.array_equal(a[:100], scene.make_scene(1)[:100])
    coors, ids = index_ops.voxelize(a, refcfg.CART_VOXEL, refcfg.CART_RANGE)
    assert 60_000 < coors.shape[0] < 140_000 and (ids >= 0).mean() > 0.95
    assert scene.cart2polar_rows(a).shape == (a.shape[0], 8)


def test_optimizer_steps_invalidate_weight_caches():
    """Fused optimizers leave Tensor._version alone: the operand caches key on ops._stamp = (version, optimizer epoch)."""
    import torch
    from openseg3d_amd import ops
    for fused in (True, False):
        p = torch.nn.Parameter(torch.randn(4, 4))
        opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, fused=fused)
        before = ops._stamp(p)
        p.grad = torch.ones_like(p)
        opt.step()
        assert ops._stamp(p) != before, fused
    before = ops._stamp(p)
    ops.invalidate_weight_caches()
    assert ops._stamp(p) != before


def test_spconv_checkpoint_layout_converter():
    """Both spconv weight layouts are recognised by shape; RSCK -> KRSC -> load round-trips (the layouts themselves are
    unverifiable offline: openseg3d_amd/checkpoint.py)."""
    import torch
    from openseg3d_amd import checkpoint, config, segformer
    cfg = config.default_cfg()
    torch.manual_seed(1)
    model = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    rsck = {("module." + k): (v.permute(1, 2, 3, 4, 0).contiguous() if v.dim() == 5 else v) for k, v in sd.items()}
    n_conv = sum(v.dim() == 5 for v in sd.values())
    assert n_conv == 20  # 14 submanifold + 3 strided + 3 inverse convs (pointtransformer.py:132-179)
    other = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
    import pathlib
    import tempfile
    import warnings
    with warnings.catch_warnings(record=True) as seen:  # the assumed spconv layout is announced at load time
        warnings.simplefilter("always")
        res = checkpoint.load_reference_checkpoint(other, {"model": rsck, "epoch": 3})
    assert any("ASSUMED" in str(w.message) for w in seen)
    assert not res.missing_keys and not res.unexpected_keys
    with tempfile.TemporaryDirectory() as d:  # a pathlib.Path is a path, not a state dict
        f = pathlib.Path(d) / "epoch_3.pth"
        torch.save({"model": rsck, "epoch": 3}, f)
        third = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res3 = checkpoint.load_reference_checkpoint(third, f)
        assert not res3.missing_keys and not res3.unexpected_keys
        assert all(torch.equal(v, sd[k]) for k, v in third.state_dict().items())
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k
    same = checkpoint.convert_spconv_state_dict(sd, model)  # already KRSC: untouched
    assert all(same[k] is sd[k] for k in sd)
    bad = dict(sd)
    key = "point_transformer.conv_input.0.weight"
    bad[key] = sd[key].permute(0, 4, 1, 2, 3).contiguous()
    try:
        checkpoint.convert_spconv_state_dict(bad, model)
    except ValueError as e:
        assert "neither KRSC" in str(e)
    else:
        raise AssertionError("a weight in neither layout must be refused")


def test_dropout_hash_statistics_on_host():
    """The mask generator as a random source, on its host restatement (640-token windows, 6 seeds x 3 heads): drop rate of
    the 26 / 256 threshold, lag-1 correlation of the drop indicator along keys, along queries and along the diagonal, and the
    chi-square of the byte histogram, all within what an ideal generator gives (4 sigma)."""
    n, thr = 640, 26
    for seed in (1, 0x1234_5678_9ABC_DEF, 77 << 33):
        for head in (0, 3, 7):
            by = dropout_ref.dropout_bytes(seed, 5, head, n).astype(np.int64)
            drop = (by < thr).astype(np.float64)
            rate, cells = thr / 256.0, float(n * n)
            assert abs(drop.mean() - rate) < 4.0 * (rate * (1 - rate) / cells) ** 0.5
            d = drop - drop.mean()
            var = float((d * d).mean())
            for a, b in ((d[:, 1:], d[:, :-1]), (d[1:, :], d[:-1, :]), (d[1:, 1:], d[:-1, :-1]), (d[1:, :-1], d[:-1, 1:])):
                assert abs(float((a * b).mean()) / var) < 4.0 / cells ** 0.5
            h = np.bincount(by.reshape(-1), minlength=256).astype(np.float64)
            chi2 = float(((h - cells / 256) ** 2 / (cells / 256)).sum())
            assert abs(chi2 - 255.0) < 4.0 * (2 * 255.0) ** 0.5


def test_deferred_join_bookkeeping_is_safe_by_construction(monkeypatch):
    """ops._defer_join on the CPU autograd engine with stand-in streams (no kernel runs): the join is deferred to the
    engine's final callback in the plain case; NEVER when a process group exists in the process (DDP / FSDP / comm hooks
    hang their work on the AccumulateGrad node where no Python check sees it); and the state a pass leaves behind when it
    raises mid-backward (the engine then skips its callbacks) is joined and dropped by the next pass instead of pinning
    tensors or being compared with the next pass's gradients."""
    from types import SimpleNamespace
    from openseg3d_amd import ops

    class FakeStream:
        def __init__(self):
            self.waits = 0

        def wait_stream(self, other):
            self.waits += 1

    log = []
    ctx_jobs = []  # pending partial-block sums a backward function hands over with its deferral

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w = w
            return x * w.sum()

        @staticmethod
        def backward(ctx, dy):
            gr = torch.full_like(ctx.w, 2.0)
            fk = SimpleNamespace(main=FakeStream(), side=FakeStream(), keep=[gr], jobs=list(ctx_jobs), device=SimpleNamespace(index=0))
            log.append((ops._defer_join(fk, [(ctx.w, gr)]), fk))
            return dy * ctx.w.sum(), gr

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, dy):
            raise RuntimeError("boom")

    monkeypatch.setattr(ops, "WGRAD_DEFER", True)
    monkeypatch.setitem(ops._DEFER_PROBE, "ok", True)
    x = torch.ones(3, requires_grad=True)
    w = torch.ones(4, requires_grad=True)
    Fn.apply(x, w).sum().backward()
    deferred, fk = log.pop()
    assert deferred and fk.main.waits == 1 and not ops._DEFERRED
    assert torch.equal(w.grad, torch.full((4,), 2.0))

    # a process group exists -> joined at the end of the backward function, whoever wrapped the model
    monkeypatch.setattr(ops, "_process_group_exists", lambda: True)
    w.grad = None
    Fn.apply(x, w).sum().backward()
    assert log.pop()[0] is False and not ops._DEFERRED
    monkeypatch.setattr(ops, "_process_group_exists", lambda: False)

    # the probe said no -> never deferred
    monkeypatch.setitem(ops._DEFER_PROBE, "ok", False)
    w.grad = None
    Fn.apply(x, w).sum().backward()
    assert log.pop()[0] is False
    monkeypatch.setitem(ops._DEFER_PROBE, "ok", True)

    # a pass that dies after a deferral leaves its record behind (the engine skips the callbacks): the next pass has its
    # own record and is not disturbed; dead records are completed once more than _MAX_IDLE_STATES pile up, or on reset
    w.grad = None
    with pytest.raises(RuntimeError, match="boom"):
        Fn.apply(Boom.apply(x), w).sum().backward()
    deferred, dead = log.pop()
    assert deferred and len(ops._DEFERRED) == 1 and dead.main.waits == 0  # callback skipped
    w2 = torch.ones(4, requires_grad=True)
    Fn.apply(x, w2).sum().backward()
    deferred, fk = log.pop()
    assert deferred and fk.main.waits == 1 and len(ops._DEFERRED) == 1
    assert torch.equal(w2.grad, torch.full((4,), 2.0))
    for _ in range(ops._MAX_IDLE_STATES + 2):
        wd = torch.ones(4, requires_grad=True)
        with pytest.raises(RuntimeError, match="boom"):
            Fn.apply(Boom.apply(x), wd).sum().backward()
        log.pop()
    assert len(ops._DEFERRED) <= ops._MAX_IDLE_STATES + 1 and dead.main.waits == 1  # the oldest were completed
    ops._reset_deferred()
    assert not ops._DEFERRED

    # a NESTED pass (reentrant torch.utils.checkpoint runs torch.autograd.backward inside a backward function,
    # point_transformer_layer.py:321-337) is another graph task: it completes its own record at its own end and leaves the
    # enclosing pass's pending sums and aliases alone
    ran = []
    monkeypatch.setattr(ops, "_run_reduce_jobs", lambda jobs, device, stream: ran.append(list(jobs)))

    class Nest(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w, ctx.x = w, x.detach()
            return x.clone()

        @staticmethod
        def backward(ctx, dy):
            assert len(ops._DEFERRED) == 1  # the outer pass's record, made by the Fn that ran before us
            outer = next(iter(ops._DEFERRED.values()))
            assert outer["jobs"] == ["outer sum"]
            with torch.enable_grad():
                xi = ctx.x.clone().requires_grad_()
                inner = Fn.apply(xi, ctx.w).sum()
            ctx_jobs[:] = ["inner sum"]
            torch.autograd.backward(inner)
            ctx_jobs[:] = []
            deferred_inner, fki = log.pop()
            assert deferred_inner and fki.main.waits == 1 and ran == [["inner sum"]]  # the nested task finished its own record
            assert len(ops._DEFERRED) == 1 and outer["jobs"] == ["outer sum"] and len(outer["fix"]) == 1
            return dy, None

    w_out, w_in = torch.ones(4, requires_grad=True), torch.ones(4, requires_grad=True)
    ctx_jobs[:] = ["outer sum"]
    Fn.apply(Nest.apply(x, w_in), w_out).sum().backward()
    deferred, fk = log.pop()
    assert deferred and fk.main.waits == 1 and ran == [["inner sum"], ["outer sum"]] and not ops._DEFERRED
    assert torch.equal(w_out.grad, torch.full((4,), 2.0)) and torch.equal(w_in.grad, torch.full((4,), 2.0))
