"""One rank of tests/test_gpu_training.py::test_plain_ddp_wrapper_gets_finished_gradients (started by
dist.launch_local_ranks).  The model is wrapped in torch's DistributedDataParallel DIRECTLY -- not through
dist.wrap_data_parallel, which used to be the only thing that switched the deferred weight-gradient join off -- and
trained one step; every rank prints an fp64 fingerprint of its gradients.  With SEG3D_DDP_BACKEND=gloo all ranks share
cuda:0 (a one-GPU box), with nccl every rank takes its own card (RCCL over xGMI)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openseg3d_amd import batch as B, config, dist as D, ops, scene, segformer  # noqa: E402


def main():
    backend = os.environ.get("SEG3D_DDP_BACKEND", "gloo")
    rank, world, local = D.init_job(backend=backend, share_device=backend == "gloo")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    assert ops.WGRAD_DEFER or os.environ.get("SEG3D_WGRAD_DEFER") == "0"
    cfg = config.default_cfg()
    ds = config.DatasetSpec(cfg)
    torch.manual_seed(0)
    model = segformer.build_segmentor(cfg, ds).to(dev).train()
    if os.environ.get("SEG3D_DDP_WRAPPER") == "native":  # dist.SceneParallel: exchange after the pass, deferred join kept
        ddp = D.wrap_data_parallel(model, dev)
        assert isinstance(ddp, D.SceneParallel)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], broadcast_buffers=False)
    only = os.environ.get("SEG3D_DDP_ONLY_SCENE")  # single-rank reference runs: the scene of that rank
    seed = int(only) if only is not None else rank
    pts = scene.make_small_scene(20 + seed, 5000 + 1500 * seed, extent=8.0 + seed)
    b = B.make_batch([pts], ds.voxel_size, ds.point_cloud_range, device=dev)
    labels = torch.arange(b["points"].shape[0], device=dev) % 22
    # SEG3D_DDP_PASSES=2: the same step twice (gradients dropped in between) -- SceneParallel learns the arrival order in its
    # first synchronised pass and exchanges slice by slice from the weight-gradient stream DURING the second
    for _ in range(int(os.environ.get("SEG3D_DDP_PASSES", "1"))):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(5)  # DropPath / dropout seeds equal on every rank, in every pass and in the reference runs
        res = ddp(b)
        loss = torch.nn.functional.cross_entropy(res["point_out"], labels) + res["voxel_out"].square().mean() \
            + 0.4 * res["aux_voxel_out"].square().mean()
        loss.backward()
    torch.cuda.synchronize(dev)
    fp = {k: float(p.grad.double().abs().sum()) for k, p in model.named_parameters()}
    probe = model.point_transformer.swformer_block3[1].layers[0].mlp.fc1.weight.grad
    out = {"rank": rank, "world": world, "backend": dist.get_backend(), "total": float(sum(fp.values())),
           "n_nonfinite": int(sum(1 for v in fp.values() if v != v)), "probe": probe.double().flatten()[:64].tolist(),
           "deferred_in_this_process": bool(ops._DEFERRED) or ops.DEFER_COUNT > 0,
           "early_slices": getattr(ddp, "early_slices", None), "slices": len(getattr(ddp, "_slices", None) or [])}
    sys.stdout.write("\nDDPRANK " + json.dumps(out) + "\n")  # one write: the ranks share the pipe
    sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
