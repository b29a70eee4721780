"""GPU checks of the training path and of bench.py's output contract."""
import json
import os
import subprocess
import sys

import pytest
import torch

import refcfg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_step_reaches_every_parameter():
    """DDP runs with find_unused_parameters=False: every parameter must receive a gradient each step."""
    from openseg3d_amd import batch as B, config, scene, segformer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    samples = [scene.make_small_scene(7, 6000, extent=9.0), scene.make_small_scene(8, 4000, extent=6.0)]
    ce = torch.nn.functional.cross_entropy
    losses = []
    for _ in range(3):
        b = B.make_batch(samples, ds.voxel_size, ds.point_cloud_range)
        n = b["points"].shape[0]
        labels = torch.arange(n, device=dev) % 22
        opt.zero_grad(set_to_none=True)
        res = model(b)
        loss = (ce(res["point_out"], labels) + ce(res["voxel_out"], labels[:1].expand(res["voxel_out"].shape[0]))
                + 0.4 * ce(res["aux_voxel_out"], labels[:1].expand(res["aux_voxel_out"].shape[0])))
        loss.backward()
        missing = [k for k, p in model.named_parameters() if p.grad is None]
        assert not missing, missing
        bad = [k for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
        assert not bad, bad
        opt.step()
        losses.append(float(loss))
    assert all(l == l and l < 1e4 for l in losses)
    assert losses[-1] < losses[0], losses  # same batch three times: SGD must reduce the loss
    # BatchNorm running statistics moved (training mode) and stay finite
    bn = getattr(model.point_transformer.conv_input, "1")
    assert int(bn.num_batches_tracked) == 3 and torch.isfinite(bn.running_var).all()


@pytest.mark.parametrize("segmentor", ["segformer", "spnet"])
def test_training_step_is_bit_reproducible(segmentor):
    """No kernel on the single-sweep training path accumulates with atomics (weight gradients, LayerNorm / BatchNorm sums,
    tau gradient, losses all reduce per-block partials in a fixed order): the same step from the same state yields the
    same loss and the same gradients bit for bit, DropPath masks included once the generator is re-seeded -- and whether
    the weight gradients run on the second stream or not."""
    from openseg3d_amd import batch as B, config, losses, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    cfg.MODEL.SEGMENTOR = segmentor
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(1)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    crit = losses.build_criterion(cfg, ds)
    samples = [scene.make_small_scene(17, 9000, extent=10.0), scene.make_small_scene(18, 5000, extent=6.0)]
    runs = []
    for rep in range(3):
        # the third run puts every kernel on ONE stream: the second stream (weight gradients with their deferred join,
        # aux-label lookup) must not change a bit either
        ops.WGRAD_STREAM = losses.AUX_OVERLAP = rep < 2
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(123)
        b = B.make_batch(samples, ds.voxel_size, ds.point_cloud_range)
        n = b["points"].shape[0]
        b["point_labels"] = (torch.arange(n, device=dev) * 5 % 22).long()
        b["voxel_labels"] = ops.prepare_voxel_labels(b["point_voxel_ids"], b["point_labels"].to(torch.uint8),
                                                     b["voxel_coords"].shape[0]).long()
        res = model(b)
        loss = losses.compute_loss(res, b, crit, cfg)
        loss.backward()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    ops.WGRAD_STREAM = losses.AUX_OVERLAP = True
    for other in (1, 2):
        assert torch.equal(runs[0][0], runs[other][0])
        differing = [k for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[other][1][k])]
        assert not differing, (other, differing[:8])


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--scenes", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "points/s" and d["value"] > 0 and d["data"] == "synthetic" and d["dtype"].startswith("f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] == 20 and d["attention_roofline"]["layers"] == sum(refcfg.DEPTHS)
    # what the driver cannot see otherwise: the counter-based fraction beside the formula's and the footprint's, whether
    # the quoted counters were read from the library that just ran, and the same step in the reference's own arithmetic
    for key in ("frac_by_counters", "frac_of_hbm_by_footprint", "traffic_stale", "lib_sha16", "traffic_lib_sha16"):
        assert key in r, key
    if r["traffic"] is not None:
        assert abs(r["frac_by_counters"] - r["traffic"] / (r["us_per_launch"] * 1e-6) / 1e9 / r["peak"]) < 2e-3
        assert r["traffic_stale"] == (r["traffic_lib_sha16"] != r["lib_sha16"])
    f = d["fp32_exact"]
    assert f["ms_per_step"] > 0 and f["value"] > 0 and f["dtype"] == "f32" and f["steps"] == 5
    assert f["fwd_only_ms_per_step"] > 20.0  # exact-fp32 products run at 1/16 of the bf16 MFMA rate (default path: ~13 ms)
    assert len(d["conv_layers"]) == 20 and {l["bound"] for l in d["conv_layers"]} == {"hbm", "mfma_bf16x3"}
    assert d["fwd_only"]["value"] > d["value"]
    # round 5: the un-profiled answer to "does the GPU wait for the host inside a step" (HIP events on the main stream;
    # steady consecutive steps against steps enqueued whole behind a device-side sleep)
    idle = d["idle"]
    for key in ("host_enqueue_ms", "gpu_idle_ms", "gpu_step_ms_steady", "gpu_step_ms_fed", "gpu_opt_step_ms_steady",
                "gpu_opt_step_ms_fed", "gpu_between_steps_ms", "host_in_prefetch_ms"):
        assert key in idle and idle[key] >= 0.0, key
    assert idle["gpu_step_ms_fed"] > 10.0 and idle["host_enqueue_ms"] > 1.0


def test_bench_distributed_path_single_rank_rehearsal():
    """The N > 1 code path of bench.py (RCCL process group, DistributedDataParallel around the model with
    find_unused_parameters=False, barriers, rank-0 reporting) driven by torch.distributed.run with one rank: everything
    but the cross-GPU traffic itself.  The driver launches the same command line with --nproc-per-node N."""
    env = dict(os.environ, SEG3D_BENCH_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1",
                          "--steps", "2", "--warmup", "1", "--scenes", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no outer launcher (the driver's command form): the parent spawns two fresh ranks
    before touching the GPU; both run the DistributedDataParallel training step and rank 0 reports the aggregate.  On this
    one-GPU box the two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device); on
    the 8-GPU node the same entry runs one rank per GPU over RCCL."""
    env = dict(os.environ, SEG3D_BENCH_BACKEND="gloo", SEG3D_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--scenes", "1", "--no-cpu-baseline", "--sync-bn"], capture_output=True, text=True, timeout=1200, env=env,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["config"]["collective"] == "gloo world 2, SyncBatchNorm"


def test_sync_batchnorm_matches_full_batch():
    """--sync_bn (tools/train.py:246-247): the fused BatchNorm passes on a SyncBatchNorm module, two ranks with different
    row counts, against one fp64 BatchNorm1d over the concatenated rows -- outputs, input / residual gradients, running
    statistics, parameter gradients (tests/_syncbn_rank.py)."""
    from openseg3d_amd import dist as D  # the launcher only: this process's GPU state is irrelevant to the children
    drv = ("import sys; sys.path.insert(0, %r); from openseg3d_amd import dist as D; "
           "sys.exit(D.launch_local_ranks([sys.executable, %r], 2))" % (ROOT, os.path.join(ROOT, "tests", "_syncbn_rank.py")))
    out = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert "SYNCBN OK" in out.stdout
    assert D is not None


def test_headline_scene_logits_match_oracle():
    """BASELINE configs[1] at full size: the 174 633-point scene through the GPU path and through the CPU oracle with the
    same weights -- per-point logits within 1e-3, voxel ids and every rulebook (4 submanifold tables, 3 strided / inverse
    pairs) bit-exact.  This is bench.py's `parity` object, computed by the same functions (~25 s of CPU oracle)."""
    import bench
    from openseg3d_amd import config, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).eval()
    pts = scene.make_scene(0)
    report, o_res, o_coords, o_ids = bench.cpu_baseline(pts, pts.shape[0], None, cfg, ds, model)
    par = bench.parity_report(pts, pts.shape[0], None, ds, model, dev, o_res, o_coords, o_ids)
    assert par["n_points"] == 174633 and report["kind"] == "port"
    assert par["voxel_ids_bit_exact"] is True and par["rulebook_bit_exact"] is True
    assert par["max_abs_logit_diff"] < 1e-3, par
    assert par["max_abs_voxel_logit_diff"] < 1e-3 and par["max_abs_aux_logit_diff"] < 1e-3, par


@pytest.mark.parametrize("workload,extra,n_expect", [
    ("cylinder", ["--batch", "4", "--scenes", "1"], 174633),        # configs/waymo_one_sweep_cylinder.yaml:2-4, BASELINE configs[2]
    ("multi_sweeps", ["--batch", "2", "--scenes", "1", "--sweeps", "5"], 30000),  # configs/waymo_multi_sweeps.yaml:1-4 + image at
    # DATASET.MAX_NUM_SWEEPS = 5 (~800 k rows per scene: BASELINE configs[3] as worded; the YAML's own NUM_SWEEPS is 3)
    ("dense2m", ["--scenes", "1"], 150000),                           # BASELINE configs[4]: 2 M points @0.02 m, 150 k-point crop
])
def test_full_size_configs_logits_match_oracle(workload, extra, n_expect):
    """BASELINE configs[2] and configs[3] at their bench sizes, driver-visible: `bench.py --workload ...` builds the
    batch (4 cylinder scenes with the device cart2polar / 2 scenes of 3 sweeps with image features and the DeepFusion
    kNN), and its `parity` object compares scene 0 with the CPU oracle on the same weights: voxel ids and every rulebook
    bit-exact, per-point logits within 1e-3 (the multi-sweep sample is the first 30 000 current-sweep rows + 15 000 rows of
    each of the four history sweeps: the oracle's brute-force kNN sets that limit, bench.py `n_cpu`)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--mode", "fwd", "--steps", "1",
                          "--warmup", "1"] + extra, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    par = line["parity"]
    assert {"cylinder": "cylinder", "multi_sweeps": "multi_sweeps", "dense2m": "dense scene"}[workload] in line["config"]["workload"]
    if workload == "cylinder":
        # the device's phi is the correctly rounded atan2, numpy's float32 arctan2 (what the reference's loader computes)
        # is up to 4 ulp off: the points that fall into ANOTHER voxel because of it are counted, and they are few
        moved = par["cylinder_phi"]
        assert moved["n_points"] == n_expect and moved["frac"] <= 1e-4, moved
    if workload == "multi_sweeps":
        assert "5 sweeps" in line["config"]["workload"], line["config"]["workload"]
    assert par["n_points"] == n_expect, par
    assert par["voxel_ids_bit_exact"] is True and par["rulebook_bit_exact"] is True, par
    assert par["max_abs_logit_diff"] < 1e-3 and par["max_abs_voxel_logit_diff"] < 1e-3 and par["max_abs_aux_logit_diff"] < 1e-3, par
    assert par["max_rel_logit_diff"] < 1e-4 and "tests/golden" in par["pinned_by"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1


def test_eval_after_fused_optimizer_step_sees_the_new_weights():
    """torch's fused optimizers update parameters without bumping Tensor._version; the packed / folded operands cached by
    the conv, Linear and BatchNorm wrappers must follow them anyway (ops._stamp: version + optimizer-step epoch).
    Trained module in eval mode == a fresh module loaded from its state_dict, bit for bit."""
    from openseg3d_amd import batch as B, config, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(3)
    model = segformer.build_segmentor(cfg, ds).to(dev)
    samples = [scene.make_small_scene(21, 7000, extent=9.0)]
    ce = torch.nn.functional.cross_entropy
    for fused in (True, False):
        opt = torch.optim.SGD(model.parameters(), lr=0.02, momentum=0.9, fused=fused)
        model.eval()
        with torch.no_grad():
            model(B.make_batch(samples, ds.voxel_size, ds.point_cloud_range))  # fills every eval-mode cache
        model.train()
        for _ in range(2):
            b = B.make_batch(samples, ds.voxel_size, ds.point_cloud_range)
            lab = torch.arange(b["points"].shape[0], device=dev) % 22
            opt.zero_grad(set_to_none=True)
            res = model(b)
            (ce(res["point_out"], lab) + res["voxel_out"].square().mean() + res["aux_voxel_out"].square().mean()).backward()
            opt.step()
        fresh = segformer.build_segmentor(cfg, ds).to(dev).eval()
        fresh.load_state_dict(model.state_dict())
        model.eval()
        with torch.no_grad():
            a = model(B.make_batch(samples, ds.voxel_size, ds.point_cloud_range))["point_out"]
            c = fresh(B.make_batch(samples, ds.voxel_size, ds.point_cloud_range))["point_out"]
        assert torch.equal(a, c), (fused, float((a - c).abs().max()))


@pytest.mark.parametrize("variant", ["segformer", "spnet", "segformer_multi_sweep_fusion"])
def test_every_parameter_gradient_matches_oracle_autograd(variant):
    """End-to-end BACKWARD parity: the gradient of a loss on all three heads w.r.t. every parameter (400+ for Segformer), GPU
    path (the training-mode kernels: sparse-conv dgrad / wgrad, fused encoder layer, window-attention backward, LayerNorm /
    BatchNorm backward, gather / scatter backward, kNN-indexed DeepFusion) against fp64 autograd through the CPU oracle.
    Eval statistics (running BatchNorm moments, no dropout / DropPath) make the two forward passes the same function;
    reference: the module trees of seg3d/models/segmentors/segformer.py:94-146 + backbones/pointtransformer.py:181-219,
    segmentors/spnet.py:94-148 + backbones/spconv_unet.py:186-233 (BASELINE configs[3]) and the multi-sweep / image-fusion
    switches of configs/waymo_multi_sweeps.yaml with layers/deep_fusion.py:26-45 (BASELINE configs[2])."""
    import numpy as np
    from oracle import index_ops, model as omodel, params
    from openseg3d_amd import batch as B, config, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    multi = variant == "segformer_multi_sweep_fusion"
    if variant == "spnet":
        cfg.MODEL.SEGMENTOR = "spnet"
    if multi:
        cfg.DATASET.USE_MULTI_SWEEPS = cfg.DATASET.USE_IMAGE_FEATURE = True
    ds = config.DatasetSpec(cfg)
    model = segformer.build_segmentor(cfg, ds)
    params.fill_by_name(model, seed=0)
    model = model.to(dev).eval()
    pts = scene.make_small_scene(41, 3000, extent=7.0)
    n_cur = pts.shape[0]
    img = None
    if multi:  # current sweep (time lag 0) + one jittered, shifted history sweep (lag 0.1), rows of waymo_dataset.py:156-202
        rs = np.random.RandomState(5)
        hist = pts[::2].copy()
        hist[:, :3] += rs.normal(0.0, 0.02, size=(hist.shape[0], 3)).astype(np.float32)
        hist[:, 0] += 0.4
        hist[:, 3] = 0.1
        pts = np.concatenate([pts, hist], axis=0)
        img = scene.make_image_features(3, n_cur)
    # GPU: eval-mode forward with autograd on
    dev_pts = B.collate_points([pts], dev)
    b = B.batch_from_resident(dev_pts, [n_cur], ds.voxel_size, ds.point_cloud_range,
                              None if img is None else torch.from_numpy(img).to(dev))
    res = model(b)
    w_pt = torch.linspace(0.5, 1.5, 22, device=dev)
    loss = (res["point_out"] * w_pt).square().mean() + res["voxel_out"].square().mean() + res["aux_voxel_out"].square().mean()
    loss.backward()
    # oracle: fp64 autograd through the functional restatement
    coords, ids = index_ops.voxelize(pts, ds.voxel_size, ds.point_cloud_range)
    ob = {"points": torch.from_numpy(np.pad(pts, ((0, 0), (1, 0)))).double(),
          "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float(),
          "point_voxel_ids": torch.from_numpy(ids).long(), "batch_size": 1,
          "point_id_offset": torch.tensor([float(n_cur)])}
    if multi:
        ob["point_image_features"] = torch.from_numpy(img).double()
    sd = model.state_dict()
    trainable = {k for k, _ in model.named_parameters()}
    p = {k: (v.detach().cpu().double().requires_grad_() if k in trainable else (v.cpu().double() if v.dtype.is_floating_point else v.cpu()))
         for k, v in sd.items()}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range),
            "batching_info": [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO],
            "window_shape": cfg.MODEL.WINDOW_SHAPE, "depths": cfg.MODEL.DEPTHS,
            "use_multi_sweeps": multi, "use_image_feature": multi}
    ref = (omodel.spnet_forward if variant == "spnet" else omodel.segformer_forward)(ob, p, ocfg)
    assert float((res["point_out"].detach().cpu().double() - ref["point_out"].detach()).abs().max()) < 1e-3
    oloss = ((ref["point_out"] * w_pt.cpu().double()).square().mean() + ref["voxel_out"].square().mean()
             + ref["aux_voxel_out"].square().mean())
    oloss.backward()
    assert abs(float(loss) - float(oloss)) <= 1e-4 * abs(float(oloss))
    # the yardstick: the SAME graph in plain torch float32 on the CPU (the oracle's functional forward with float32
    # parameters and inputs) against its float64 self -- what fp32 arithmetic alone does to each parameter's gradient
    # through the network's discrete switches (ReLU masks, max-pool arg-max).  The GPU path's bar is derived from it.
    p32 = {k: (v.detach().float().requires_grad_() if k in trainable else (v.float() if v.dtype.is_floating_point else v))
           for k, v in p.items()}
    ob32 = {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.float64 else v) for k, v in ob.items()}
    ref32 = (omodel.spnet_forward if variant == "spnet" else omodel.segformer_forward)(ob32, p32, ocfg)
    loss32 = ((ref32["point_out"] * w_pt.cpu().float()).square().mean() + ref32["voxel_out"].square().mean()
              + ref32["aux_voxel_out"].square().mean())
    loss32.backward()
    worst, ratios = [], []
    for k, prm in model.named_parameters():
        if k.startswith("scatter."):
            continue
        g_ref = p[k].grad
        assert prm.grad is not None and g_ref is not None, k
        scale = float(g_ref.abs().max())
        err = float((prm.grad.cpu().double() - g_ref).abs().max())
        yard = float((p32[k].grad.double() - g_ref).abs().max()) / max(scale, 1e-12)
        worst.append((err / max(scale, 1e-12), k, err, scale))
        ratios.append((err / max(scale, 1e-12), yard, k))
    worst.sort(reverse=True)
    # What the yardstick shows (measured): torch's float32 reproduces the float64 gradients to ~3e-7 (median) / 2e-4
    # (worst) on this graph, the split-bf16 path (16 significant bits per operand) to ~2e-4 (median) -- the 2^8 between
    # the two mantissas, times the three products of a split multiply.  The bars below are set from that distribution:
    # the median within 5e-4, every parameter within 5e-3 of its largest gradient entry except a handful (<= 5 %) of
    # parameters of the stride-8 level (`up4.*` / `ocr.*`: their weight gradients on this 3 000-point scene are sums over
    # a few dozen rows behind ~40 layers of backward, where ONE activation that crosses a ReLU or max-pool tie within
    # the forward's 1e-5 moves the sum by a percent), which must stay within 5 %; tau gradients (one number summed over
    # every (query, key) pair with cancellation, test_gpu_attention.py) within 5 %.  An indexing or transposition error
    # is O(100 %) on the parameter it touches.  Which parameters land in the handful moves with ANY perturbation of the forward at
    # the 1e-7 level: SPNet counts 6 of 208 with the per-point MLPs on the fp32 MFMA and 8 with the six-product split (both
    # fp32-grade, tests/test_gpu_dense.py), the same `up4.*` / `ocr.*` family, all under 1.1 %.
    live = [(rel, yard, k) for rel, yard, k in ratios if not k.endswith(".tau")]
    loose = sorted([(rel, k) for rel, _, k in live if rel > 5e-3 and p[k].grad.abs().max() > 1e-12], reverse=True)
    assert len(loose) <= max(6, len(live) // 20), loose[:12]
    assert all(rel <= 5e-2 for rel, _ in loose), loose[:8]
    for rel, k, err, scale in worst:
        if k.endswith(".tau"):
            assert rel <= 5e-2 or err <= 1e-7, (k, rel, err, scale)
    med_gpu = sorted(r for r, _, _ in ratios)[len(ratios) // 2]
    med_f32 = sorted(y for _, y, _ in ratios)[len(ratios) // 2]
    print(f"[{variant}] median relative gradient error: GPU path {med_gpu:.2e}, torch float32 on the same graph {med_f32:.2e}; "
          f"parameters beyond 5e-3: {[(round(r, 4), k) for r, k in loose]}")
    assert med_gpu < 5e-4


def test_side_stream_weight_gradients_keep_autograd_semantics(monkeypatch):
    """Weight gradients run on a second stream and, in the plain case, are only awaited at the end of the backward pass
    (ops._WgradFork / _defer_join).  The cases where autograd touches the gradient earlier must give exactly what the
    single-stream path gives (the kernels are deterministic, so: bit-identical): a weight and bias used twice in one
    graph (the engine adds the two gradients), accumulation into an existing .grad, a tensor hook on the parameter,
    torch.autograd.grad, create_graph, and a sparse-conv weight; plus fp64 autograd as the sanity bound."""
    from openseg3d_amd import ops
    assert ops.WGRAD_STREAM and ops.WGRAD_DEFER  # the defaults under test
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    m, c = 40000, 96
    x = torch.randn(m, c, generator=gen).to(dev)
    w0 = (torch.randn(c, c, generator=gen) / c ** 0.5).to(dev)
    b0 = (torch.randn(c, generator=gen) * 0.1).to(dev)
    g = torch.randn(m, c, generator=gen).to(dev)

    def expr(lin, xx, w, b):  # the same weight and bias twice, a nonlinearity in between
        return lin(torch.relu(lin(xx, w, b)), w, b)

    def run():
        out = {}
        w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
        expr(ops.linear, x, w, b).backward(g)
        out["shared_w"], out["shared_b"] = w.grad.clone(), b.grad.clone()
        expr(ops.linear, x, w, b).backward(g)  # accumulates into the existing .grad
        out["acc_w"], out["acc_b"] = w.grad.clone(), b.grad.clone()
        w3, seen = w0.clone().requires_grad_(), []
        w3.register_hook(lambda gr: seen.append(gr.clone()))
        ops.linear(x, w3, None).backward(g)
        out["hook"], out["plain_w"] = seen[0], w3.grad.clone()
        w4 = w0.clone().requires_grad_()
        out["functional"] = torch.autograd.grad(ops.linear(x, w4, None), w4, g)[0].clone()
        out["create_graph"] = torch.autograd.grad(ops.linear(x, w4, None), w4, g, create_graph=True)[0].detach().clone()
        w5, b5 = w0.clone().requires_grad_(), b0.clone().requires_grad_()  # the plain deferred case, weight and bias
        ops.linear(x, w5, b5).backward(g)
        out["plain2_w"], out["plain2_b"] = w5.grad.clone(), b5.grad.clone()
        torch.cuda.synchronize()
        return out

    got = run()
    monkeypatch.setattr(ops, "WGRAD_STREAM", False)
    want = run()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # sanity against fp64 autograd (ReLU masks that flip within the forward's rounding set the floor: torch's own fp32
    # gradient differs from fp64 by 0.6 % of the largest entry on this expression)
    w64, b64 = w0.double().cpu().requires_grad_(), b0.double().cpu().requires_grad_()
    expr(torch.nn.functional.linear, x.double().cpu(), w64, b64).backward(g.double().cpu())
    assert float((got["shared_w"].double().cpu() - w64.grad).abs().max()) <= 2e-2 * float(w64.grad.abs().max())
    assert float((got["shared_b"].double().cpu() - b64.grad).abs().max()) <= 2e-2 * float(b64.grad.abs().max())


def test_batch_prepared_on_the_pipeline_stream_gives_the_same_logits():
    """bench.py's input pipeline: batch i+1 is voxelized, collated and planned (Segformer.prepare_batch) on its own stream
    while batch i runs.  A batch built that way -- beside a forward in flight on the main stream, then handed over by an
    event -- gives bit-identical logits to the same scene built and run on one stream; and a model run twice on a
    prepared batch does not rebuild the plan."""
    from openseg3d_amd import batch as B, config, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).eval()
    a = B.collate_points([scene.make_scene(5)[::3]], dev)
    b = B.collate_points([scene.make_small_scene(9, 20000, extent=20.0)], dev)
    na, nb = a.shape[0], b.shape[0]
    with torch.no_grad():
        plain = model(B.batch_from_resident(b, [nb], ds.voxel_size, ds.point_cloud_range))["point_out"].clone()
        main, pipe = torch.cuda.current_stream(dev), ops.side_stream(dev, 1)
        for _ in range(3):
            mark = torch.cuda.Event()
            mark.record(main)
            res_a = model(B.batch_from_resident(a, [na], ds.voxel_size, ds.point_cloud_range))  # in flight on main
            pipe.wait_event(mark)
            with torch.cuda.stream(pipe):
                bb = model.prepare_batch(B.batch_from_resident(b, [nb], ds.voxel_size, ds.point_cloud_range))
                done = torch.cuda.Event()
                done.record(pipe)
            main.wait_event(done)
            level = bb["site_level"]
            out = model(bb)["point_out"]
            assert bb["site_level"] is level  # the plan the batch carried was used, not rebuilt
            assert torch.equal(out, plain)
            del res_a


def test_bench_streams_and_pipeline_train_the_same_weights():
    """bench.py's default run (weight gradients + aux-label lookup on a second stream with the deferred join, next batch's
    voxelization and index plan on a third) and the same steps with everything on ONE stream (`--no-pipeline`,
    SEG3D_WGRAD_STREAM=0, SEG3D_AUX_OVERLAP=0) end with the same weights to the last digit of an fp64 fingerprint: the
    kernels are deterministic, so any hand-over that came too early or too late would show here."""
    def run(extra, env):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--scenes", "2",
                              "--mode", "fwdbwd", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=900,
                             cwd=ROOT, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    a = run([], {})
    b = run(["--no-pipeline"], {"SEG3D_WGRAD_STREAM": "0", "SEG3D_AUX_OVERLAP": "0"})
    assert "third stream" in a["config"]["streams"] and "--no-pipeline" in b["config"]["streams"]
    assert a["trained_weights_l1"] == b["trained_weights_l1"] and a["trained_weights_l1"] > 0
    # the same steps as ONE rank of a data-parallel job (RCCL group of one, dist.SceneParallel: gradients packed into the
    # arena, averaged over one rank, .grad rebound to arena views; the deferred join stays on): the same weights again
    c = run([], {"SEG3D_BENCH_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": "29531"})
    assert c["config"]["collective"].startswith("nccl world 1")
    assert c["trained_weights_l1"] == a["trained_weights_l1"]


def test_deferred_join_probe_passes_and_catches_a_broken_engine(monkeypatch):
    """ops.probe_deferred_join is the start-up self-test behind SEG3D_WGRAD_DEFER: on this torch build the deferred and the
    joined gradients are bit-identical (True); with the engine's final callback disabled -- what a torch release that
    changed `queue_callback` would look like -- the probe must notice (the side stream is kept busy and the buffers are
    poisoned) and switch deferral off for the process."""
    from openseg3d_amd import ops
    dev = torch.device("cuda:0")
    monkeypatch.setattr(ops, "WGRAD_DEFER", True)
    monkeypatch.setitem(ops._DEFER_PROBE, "done", False)
    monkeypatch.setitem(ops._DEFER_PROBE, "ok", None)
    assert ops.probe_deferred_join(dev) is True and ops.WGRAD_DEFER
    assert ops.probe_deferred_join(dev) is True  # cached: one probe per process
    monkeypatch.setitem(ops._DEFER_PROBE, "done", False)
    monkeypatch.setitem(ops._DEFER_PROBE, "ok", None)
    engine = torch.autograd.Variable._execution_engine
    finish = ops._finish_state
    monkeypatch.setattr(ops, "_finish_state", lambda key: None)  # the callback runs but no longer joins the streams
    with pytest.warns(UserWarning, match="deferred weight-gradient join"):
        assert ops.probe_deferred_join(dev) is False
    assert ops.WGRAD_DEFER is False
    del engine
    torch.cuda.synchronize()
    monkeypatch.setattr(ops, "_finish_state", finish)
    ops._reset_deferred()
    assert not ops._DEFERRED


def _ddp_ranks(world, backend, env=None):
    from openseg3d_amd import dist as D
    drv = ("import sys\nsys.path.insert(0, %r)\nfrom openseg3d_amd import dist as D\n"
           "sys.exit(D.launch_local_ranks([sys.executable, %r], %d))\n" % (ROOT, os.path.join(ROOT, "tests", "_ddp_rank.py"), world))
    out = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, SEG3D_DDP_BACKEND=backend, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    dec = json.JSONDecoder()  # the ranks write to one pipe: two records may share a line
    recs = [dec.raw_decode(part.lstrip())[0] for part in out.stdout.split("DDPRANK ")[1:]]
    assert len(recs) == world
    return sorted(recs, key=lambda r: r["rank"])


def _check_ddp_mean(recs, singles, deferred=False):
    import numpy as np
    assert all(r["n_nonfinite"] == 0 and r["deferred_in_this_process"] == deferred for r in recs)
    a, b = recs
    assert a["probe"] == b["probe"] and a["total"] == b["total"]  # every rank holds the same reduced gradient
    want = (np.asarray(singles[0]["probe"]) + np.asarray(singles[1]["probe"])) / 2
    got = np.asarray(a["probe"])
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-30)


def test_plain_ddp_wrapper_gets_finished_gradients():
    """A model wrapped in torch's DistributedDataParallel DIRECTLY (not via dist.wrap_data_parallel): DDP's bucket hooks
    read the gradients during the backward pass, so nothing may be deferred in a process that has a process group --
    safe by construction (ops._process_group_exists).  Two gloo ranks share the card; every rank must end with the mean
    of the two single-rank gradients, and the weight-gradient stream on / off must not change a bit."""
    both = _ddp_ranks(2, "gloo")
    off = _ddp_ranks(2, "gloo", {"SEG3D_WGRAD_STREAM": "0"})
    assert [r["probe"] for r in both] == [r["probe"] for r in off] and [r["total"] for r in both] == [r["total"] for r in off]
    singles = [_ddp_ranks(1, "gloo", {"SEG3D_DDP_ONLY_SCENE": str(r), "SEG3D_BENCH_DIST": "1"})[0] for r in range(2)]
    _check_ddp_mean(both, singles)


def test_scene_parallel_wrapper_exchanges_after_the_pass():
    """dist.wrap_data_parallel's default wrapper (dist.SceneParallel): only arrival counters hang on the AccumulateGrad
    nodes, the backward pass keeps its deferred weight-gradient join, and the exchange runs out of a flat arena -- after the
    first pass, slice by slice during the second.  Two gloo ranks share the card: both end with the mean of the two
    single-rank gradients, with deferral ON in their processes."""
    env = {"SEG3D_DDP_WRAPPER": "native", "SEG3D_DDP_PASSES": "2", "SEG3D_DDP_BUCKET_MB": "8"}
    both = _ddp_ranks(2, "gloo", env)
    singles = [_ddp_ranks(1, "gloo", {"SEG3D_DDP_ONLY_SCENE": str(r), "SEG3D_BENCH_DIST": "1"})[0] for r in range(2)]
    _check_ddp_mean(both, singles, deferred=True)
    # round 5: the second pass exchanged most slices from the weight-gradient stream while the pass was still running
    # (flushed sums + one multi-tensor copy + all-reduce per slice, ordered behind the gradient kernels on that stream)
    for r in both:
        assert r["slices"] >= 8 and r["early_slices"] >= r["slices"] // 2, r


def test_two_gpus_rccl_all_reduce():
    """First contact with a second GPU (skipped on a one-GPU box): the same two-rank step over `nccl` (= RCCL over xGMI),
    one card per rank -- equal reduced gradients on both ranks, equal to the mean of the two single-rank gradients -- and
    `bench.py --gpus 2` end to end (fresh children via launch_local_ranks, never an exec): one contract line,
    `config.collective == "nccl world 2"`, per-rank step times reported, both ranks trained the same weights."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    both = _ddp_ranks(2, "nccl")
    assert all(r["backend"] == "nccl" for r in both)
    singles = [_ddp_ranks(1, "gloo", {"SEG3D_DDP_ONLY_SCENE": str(r), "SEG3D_BENCH_DIST": "1"})[0] for r in range(2)]
    _check_ddp_mean(both, singles)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "nccl world 2"
    ranks = line["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["ms_per_step"] > 0 for r in ranks)
    assert ranks[0]["trained_weights_l1"] == ranks[1]["trained_weights_l1"] > 0


def test_eval_forward_replays_from_a_hip_graph():
    """Every C-ABI entry point is sync-free and allocation-free (include/seg3d_hip.h), and a batch can carry its index plan
    (Segformer.prepare_batch: all host read-backs happen there): the eval forward of a prepared batch -- point encoder,
    VFE, 20 sparse convs, 18 window-attention layers, gathers, classifier: ~400 launches -- is captured into ONE hipGraph
    and replayed.  The replay must reproduce the eager logits bit for bit, and again after the input rows are overwritten
    in place with another scene's features (same geometry): the graph reads the buffers, not values frozen at capture."""
    from openseg3d_amd import batch as B, config, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).eval()
    pts = B.collate_points([scene.make_scene(3)[::4]], dev)
    n = pts.shape[0]
    with torch.no_grad():
        batch = model.prepare_batch(B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range))
        eager = model(dict(batch))["point_out"].clone()  # also warms every cache (packed weights, folded BatchNorm)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # capture needs a non-default stream; a few warm-up runs on it first
            for _ in range(2):
                model(dict(batch))
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = model(dict(batch))["point_out"]
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        # new feature values in the same buffers (intensity / elongation columns; the geometry and hence the plan stay)
        old = batch["points"][:, 5:].clone()
        batch["points"][:, 5:] = torch.rand_like(old)
        want = model(dict(batch))["point_out"].clone()
        assert not torch.equal(want, eager)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        batch["points"][:, 5:] = old
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)


def test_bf16_storage_mode_stays_within_its_stated_tolerance(monkeypatch):
    """The opt-in storage mode of BASELINE configs[4] (bench.py --storage bf16 / SEG3D_STORAGE=bf16): sparse-conv feature
    maps stored in bf16 in the inference forward, everything else (accumulation, residual stream, attention, norms, point
    MLPs) float32.  Its tolerance is stated against this build's own float32-storage forward: max |dlogit|
    <= 1e-1 and arg-max agreement >= 99 % -- measured 1.3e-2 / 99.85 % with golden-style weights, 3.4e-2 / 99.98 % with
    the default initialisation for the conv outputs alone (tools/bf16_storage_probe.py), 4e-2 / 99.7 % after a few training
    steps (bench.py --storage bf16).  The default path is untouched: switching the mode off
    reproduces the float32-storage logits bit for bit."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import params
    from openseg3d_amd import batch as B, config, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    pts = scene.make_scene(2)[::2]
    for fill in (True, False):
        torch.manual_seed(0)
        model = segformer.build_segmentor(cfg, ds)
        if fill:
            params.fill_by_name(model, seed=0)
        model = model.to(dev).eval()
        b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range)
        with torch.no_grad():
            monkeypatch.setattr(ops, "STORAGE", "fp32")
            ref = model(dict(b))["point_out"].clone()
            monkeypatch.setattr(ops, "STORAGE", "bf16")
            res = model(dict(b))
            got = res["point_out"]
            assert got.dtype == torch.float32 and res["voxel_out"].dtype == torch.float32
            monkeypatch.setattr(ops, "STORAGE", "fp32")
            again = model(dict(b))["point_out"]
        err = float((got - ref).abs().max())
        agree = float((got.argmax(1) == ref.argmax(1)).float().mean())
        assert 0.0 < err <= 1e-1 and agree >= 0.99, (fill, err, agree)
        assert torch.equal(again, ref)


def test_parameter_gradients_match_oracle_when_the_switches_are_linear(monkeypatch):
    """The whole-backward parity test above tolerates a few parameters beyond 5e-3 and explains them by discrete switches
    (ReLU masks, the voxel max-pool's arg-max) that flip within the forward's 1e-5.  Here the explanation is put to the
    test: the SAME graph with every ReLU replaced by the identity and the voxel max-pool by the mean, on both sides (the
    device path: batch_norm_act's relu flag, nn.ReLU; the oracle: F.relu, scatter(reduce='max')) -- no switch is left, and
    every one of the 400+ parameter gradients must agree with fp64 autograd within 1e-3 of its largest entry (tau: 2e-2,
    one number summed over every (query, key) pair with cancellation).  A transposition or indexing error anywhere in the
    backward kernels would show here as O(1) on the parameter it touches."""
    import numpy as np
    from oracle import index_ops, model as omodel, params, sparse_conv as osc
    from openseg3d_amd import batch as B, config, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    model = segformer.build_segmentor(cfg, ds)
    params.fill_by_name(model, seed=0)
    model = model.to(dev).eval()
    # linear switches, device side
    real_bn_act = ops.batch_norm_act
    monkeypatch.setattr(ops, "batch_norm_act", lambda x, bn, relu=True, res=None: real_bn_act(x, bn, relu=False, res=res))
    monkeypatch.setattr(torch.nn.functional, "relu", lambda x, inplace=False: x)  # nn.ReLU modules and the oracle's F.relu
    model.vfe.reduce = "mean"
    # ... and oracle side
    real_scatter = osc.scatter
    monkeypatch.setattr(osc, "scatter", lambda src, index, reduce="mean", dim_size=None:
                        real_scatter(src, index, "mean" if reduce == "max" else reduce, dim_size))
    pts = scene.make_small_scene(41, 3000, extent=7.0)
    n_cur = pts.shape[0]
    b = B.batch_from_resident(B.collate_points([pts], dev), [n_cur], ds.voxel_size, ds.point_cloud_range, None)
    res = model(b)
    w_pt = torch.linspace(0.5, 1.5, 22, device=dev)
    loss = (res["point_out"] * w_pt).square().mean() + res["voxel_out"].square().mean() + res["aux_voxel_out"].square().mean()
    loss.backward()
    coords, ids = index_ops.voxelize(pts, ds.voxel_size, ds.point_cloud_range)
    ob = {"points": torch.from_numpy(np.pad(pts, ((0, 0), (1, 0)))).double(),
          "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float(),
          "point_voxel_ids": torch.from_numpy(ids).long(), "batch_size": 1, "point_id_offset": torch.tensor([float(n_cur)])}
    trainable = {k for k, _ in model.named_parameters()}
    p = {k: (v.detach().cpu().double().requires_grad_() if k in trainable else (v.cpu().double() if v.dtype.is_floating_point else v.cpu()))
         for k, v in model.state_dict().items()}
    ocfg = {"grid_size": index_ops.grid_size_of(ds.voxel_size, ds.point_cloud_range),
            "batching_info": [{int(k): v for k, v in lvl.items()} for lvl in cfg.MODEL.BATCHING_INFO],
            "window_shape": cfg.MODEL.WINDOW_SHAPE, "depths": cfg.MODEL.DEPTHS, "use_multi_sweeps": False, "use_image_feature": False}
    ref = omodel.segformer_forward(ob, p, ocfg)
    scale_out = float(ref["point_out"].detach().abs().max())
    assert float((res["point_out"].detach().cpu().double() - ref["point_out"].detach()).abs().max()) < 1e-4 * max(scale_out, 1.0)
    oloss = ((ref["point_out"] * w_pt.cpu().double()).square().mean() + ref["voxel_out"].square().mean()
             + ref["aux_voxel_out"].square().mean())
    oloss.backward()
    worst = []
    for k, prm in model.named_parameters():
        if k.startswith("scatter."):
            continue
        g_ref = p[k].grad
        assert prm.grad is not None and g_ref is not None, k
        scale = float(g_ref.abs().max())
        if scale < 1e-12:
            continue
        worst.append((float((prm.grad.cpu().double() - g_ref).abs().max()) / scale, k))
    worst.sort(reverse=True)
    assert len(worst) > 300  # (the rest: parameters whose gradient is identically zero in this graph)
    for rel, k in worst:
        assert rel <= (2e-2 if k.endswith(".tau") else 1e-3), worst[:8]


def test_spnet_error_after_training_is_conditioning_not_the_split_products():
    """Round 4's open parity question (row f3): after 13 bench steps on random labels SPNet's logits stand 1.3 apart from
    the oracle's (rel 2e-3) where they stood 2.7e-4 (rel 2.6e-6) at the starting weights.  tools/spnet_parity_probe.py
    localises it (profiles/r05_spnet_parity_probe_*.txt): the bench's SGD drives the stride-8 features to |7.6e5| and the
    auxiliary logits to |2.4e6|, and OCR's SpatialGatherModule takes a softmax over the voxels of THOSE logits (ocr.py:22-31)
    -- one fp32 ulp there is 0.25, i.e. a factor e^0.25 on a softmax weight.  The oracle's own fp32 forward is 1e-2 (relative)
    away from its fp64 forward at that stage; the GPU in exact-fp32 products 2.5e-3; the split products 4.5e-2.  Up to that
    softmax every stage stays within 4e-6 of fp64 on the trained weights in BOTH arithmetics.  Asserted here, against the
    fp64 oracle: the well-conditioned stages at 2e-5, the OCR stage within 30x of the fp32 oracle's own error, the point
    logits at rel 5e-3 (VERDICT r4 item 1's bar)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spnet_parity_probe as P
    from openseg3d_amd import batch as B, losses, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg, ds = P.setup()
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev)
    crit = losses.build_criterion(cfg, ds)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, fused=True)  # bench.py's setting
    scenes = [scene.make_scene(s) for s in range(4)]
    res_dev = [B.collate_points([s], dev) for s in scenes]
    labels = [torch.randint(0, 22, (s.shape[0],), device=dev) for s in scenes]
    model.train()
    for i in range(13):
        j = i % 4
        b = B.batch_from_resident(res_dev[j], [scenes[j].shape[0]], ds.voxel_size, ds.point_cloud_range)
        vl = ops.prepare_voxel_labels(b["point_voxel_ids"], labels[j], b["voxel_coords"].shape[0], ignore_index=ds.ignore_index).long()
        opt.zero_grad(set_to_none=True)
        loss = losses.compute_loss(model(b), {"point_labels": labels[j], "voxel_labels": vl, "batch_size": 1}, crit, cfg)
        loss.backward()
        opt.step()
    model.eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    f64 = P.oracle_forward(scenes[0], cfg, ds, sd, torch.float64)
    f32 = P.oracle_forward(scenes[0], cfg, ds, sd, torch.float32)
    rows = P.table(f64, {"oracle_f32": f32, "gpu": P.gpu_forward(model, scenes[0], ds, dev)})
    assert rows["aux_voxel_out"]["max_abs"] > 1e4, rows["aux_voxel_out"]  # the regime the record was taken in
    for name in ("conv1", "conv2", "conv3", "conv4", "aux_voxel_out"):  # in front of the ill-conditioned softmax
        assert rows[name]["gpu"]["rel"] <= 2e-5, (name, rows[name])
    assert rows["ocr"]["oracle_f32"]["rel"] >= 50 * rows["conv2"]["oracle_f32"]["rel"], rows["ocr"]  # fp32 itself loses it there
    assert rows["ocr"]["gpu"]["rel"] <= 30 * rows["ocr"]["oracle_f32"]["rel"], rows["ocr"]
    for name in ("voxel_out", "point_out"):
        assert rows[name]["gpu"]["rel"] <= 5e-3, (name, rows[name])


@pytest.mark.parametrize("segmentor", ["segformer", "spnet"])
def test_bf16_training_copies_stay_within_their_stated_tolerance(segmentor, monkeypatch):
    """BASELINE configs[4] names bf16; the reference has no reduced-precision mode, so the TRAINING mode is
    build-defined and opt-in (SEG3D_TRAIN_STORAGE=bf16, bench.py --storage bf16): the sparse-conv / Linear / encoder-layer
    functions keep a bf16 COPY of the rows their weight gradient multiplies with, the graph's tensors -- and therefore
    every gradient that flows through the graph -- stay fp32.  Stated tolerance against this repo's own fp32-copy step:
    the loss and every input-path quantity are IDENTICAL (nothing in the forward or in the input gradients reads the
    copies: bias, LayerNorm, BatchNorm and tau gradients equal bit for bit), and every conv / Linear weight gradient is
    within 1e-2 of its largest entry (one operand of each product carries 8 significant bits instead of 16; the sum over
    1e3 .. 1e5 rows averages the rounding errors: measured 1e-4 .. 2e-3)."""
    from openseg3d_amd import batch as B, config, losses, ops, scene, segformer
    dev = torch.device("cuda:0")
    cfg = config.default_cfg()
    cfg.MODEL.SEGMENTOR = segmentor
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(3)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    crit = losses.build_criterion(cfg, ds)
    samples = [scene.make_small_scene(27, 9000, extent=10.0), scene.make_small_scene(28, 6000, extent=7.0)]
    runs = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setattr(ops, "TRAIN_STORAGE", mode)
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(321)
        b = B.make_batch(samples, ds.voxel_size, ds.point_cloud_range)
        n = b["points"].shape[0]
        b["point_labels"] = (torch.arange(n, device=dev) * 7 % 22).long()
        b["voxel_labels"] = ops.prepare_voxel_labels(b["point_voxel_ids"], b["point_labels"].to(torch.uint8),
                                                     b["voxel_coords"].shape[0]).long()
        loss = losses.compute_loss(model(b), b, crit, cfg)
        loss.backward()
        runs[mode] = (loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert torch.equal(runs["fp32"][0], runs["bf16"][0])  # the forward does not read the copies
    changed, worst = 0, ("", 0.0)
    for k, g in runs["fp32"][1].items():
        h = runs["bf16"][1][k]
        if torch.equal(g, h):
            continue
        changed += 1
        assert g.dim() >= 2, k  # only weight matrices / conv kernels may move: vectors (bias, norms, tau) are untouched
        rel = float((g - h).abs().max()) / max(float(g.abs().max()), 1e-30)
        worst = max(worst, (k, rel), key=lambda t: t[1])
        assert rel <= 1e-2, (k, rel)
    assert changed >= 40, changed  # the mode really took the bf16 kernels (20+ convs, 70+ Linear layers in segformer)
    print("worst weight-gradient deviation", worst)
