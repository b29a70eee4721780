"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` over RCCL/xGMI (backend "nccl";
"gloo" for the CPU rehearsal tests).  Mirrors seg3d/utils/distributed.py:8-34 (init_dist / get_dist_info)
and the scene sharding of seg3d/datasets/samplers/distributed_sampler.py:52-58.

Scenes are independent in forward and backward: the only data-path exchange is the gradient all-reduce
(DDP buckets, overlapped with backward); throughput is aggregated as sum(points) / max(time)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Returns (rank, world_size, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """DistributedSampler rule: pad to a multiple of world, then stride by rank (distributed_sampler.py:52-58)."""
    total = (n_items + world - 1) // world * world
    idx = list(range(n_items)) + list(range(total - n_items))
    return idx[rank:total:world]


def scene_seeds(rank, per_rank):
    """Disjoint seeded scene streams per rank for the synthetic bench."""
    return [rank * per_rank + i for i in range(per_rank)]


def aggregate_throughput(seconds, units, device):
    """(max over ranks of seconds, sum over ranks of units) -- the contract's whole-job aggregate."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds), float(units)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def broadcast_parameters(module, src=0):
    """What DDP's constructor does (tools/train.py:277): every rank starts from rank 0's weights."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def allreduce_gradients(module, bucket_bytes=25 * 1024 * 1024):
    """Bucketed flat gradient averaging (the exchange DDP performs during backward); used where DDP's hooks
    are not wanted.  Buckets are sized for xGMI's per-link bandwidth rather than NVSwitch."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
