"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` over RCCL/xGMI (backend "nccl";
"gloo" for the CPU rehearsal tests).  Mirrors seg3d/utils/distributed.py:8-34 (init_dist / get_dist_info).

Scenes are independent in forward and backward: the only data-path exchange is the gradient all-reduce
(DDP buckets, overlapped with backward); throughput is aggregated as sum(points) / max(time)."""
import os
import socket
import subprocess
import time

import torch
import torch.distributed as dist


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


# ---------------------------------------------------------------------------------------------- job launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local_ranks(argv, n_ranks, extra_env=None, poll_s=0.2):
    """What tools/dist_train.sh:7-13 does with torch.distributed.launch: start ``n_ranks`` fresh child processes of
    ``argv`` (one per GPU of this node) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, wait for
    them, return 0 if all succeeded.  The caller must not have touched the GPU (a process that has initialised HIP may
    neither fork workers onto the card nor be replaced by exec); children inherit stdout / stderr, so rank 0's report
    line is the job's.  If one rank fails the others are stopped by PID and its exit code is returned."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    code = 0
    live = list(procs)
    while live:
        time.sleep(poll_s)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in live:  # a dead rank leaves the others waiting in a collective: stop exactly those PIDs
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return code


def init_job(backend=None, share_device=False):
    """(rank, world, device index) of this process; joins the default process group when WORLD_SIZE > 1 (or when
    ``SEG3D_BENCH_DIST=1`` asks for a one-rank rehearsal).  backend: "nccl" (= RCCL over xGMI, the default on GPUs) or
    "gloo" (CPU tests; also lets several ranks share one card on a one-GPU box, which RCCL refuses)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if share_device else int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("SEG3D_BENCH_DIST", "0") == "1" and "RANK" in os.environ
    if (world > 1 or rehearsal) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def convert_sync_bn(model):
    """tools/train.py:246-247 (--sync_bn): every BatchNorm of the model becomes a torch.nn.SyncBatchNorm; the fused
    BatchNorm passes of this package recognise the type and exchange their statistics (ops.batch_norm_act)."""
    return torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)


def wrap_data_parallel(model, device, sync_bn=False):
    """The reference's multi-GPU training wrapper (tools/train.py:246-247, 276-279): optional
    convert_sync_batchnorm, then DistributedDataParallel -- gradient all-reduce in buckets overlapped with backward.
    find_unused_parameters=False: every parameter of the path receives a gradient each step
    (tests/test_gpu_training.py); broadcast_buffers=False: BatchNorm statistics stay per rank unless sync_bn."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if sync_bn:
        model = convert_sync_bn(model)
    ids = [device.index] if device.type == "cuda" else None
    # DDP's bucket hooks read .grad while the backward pass is still running: weight gradients launched on the side stream
    # must be complete when their backward function returns (ops._WgradFork), not only at the end of the pass
    from . import ops
    ops.WGRAD_DEFER = False
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=False,
                                                     broadcast_buffers=False, gradient_as_bucket_view=True)


def job_barrier(device):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
