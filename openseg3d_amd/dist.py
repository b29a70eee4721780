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


class _PassEnd(torch.autograd.Function):
    """Identity on the outputs of SceneParallel.forward.  Its backward is the FIRST function of a backward pass: it asks
    the autograd engine to call the wrapper back when the pass is complete (what DistributedDataParallel's own sink does)."""

    @staticmethod
    def forward(ctx, owner, *tensors):
        ctx.owner = owner
        ctx.set_materialize_grads(False)
        return tensors

    @staticmethod
    def backward(ctx, *grads):
        owner = ctx.owner
        if owner._sync and not owner._queued:
            owner._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(owner._after_backward)
        return (None,) + grads


class SceneParallel(torch.nn.Module):
    """Data-parallel wrapper with the surface tools/train.py:276-279 uses of DistributedDataParallel (``.module``, call
    through, ``no_sync()``; parameters and buffers broadcast from rank 0 at construction), built for this path's backward:

    torch's DDP copies every gradient into its bucket view with one scaled-copy kernel per parameter (332 launches, 1.7 ms of
    the main stream per step), makes the fused optimizer walk bucket views, and -- because its hooks read ``.grad`` DURING
    the pass -- forbids the deferred weight-gradient join (ops._defer_join): +3.9 ms per step on ONE rank before a byte
    crosses xGMI.  Here nothing hangs on the AccumulateGrad nodes.  The backward pass runs exactly as in single-process
    training (weight gradients on the side stream, one join at the end); when the engine reports the pass complete
    (_PassEnd -> queue_callback) the gradients are packed into ONE flat fp32 arena with a multi-tensor copy, the arena is
    all-reduced in a few large slices (RCCL: ReduceOp.AVG, one ring pass per slice -- 131 MB in four 33 MB slices is
    per-link bound on xGMI, not latency bound), and every ``.grad`` becomes a view of the arena.  The exchange is not
    overlapped with the backward pass (~1.5 ms exposed at 8 GPUs against the 3.9 ms the hooks cost).  Gradients of
    parameters that took no part in the pass are reduced as zeros (every rank must issue the same collectives)."""

    def __init__(self, module, bucket_bytes=None):
        super().__init__()
        self.module = module
        self._sync = True
        self._queued = False
        self._params = [p for p in module.parameters() if p.requires_grad]
        if len({(p.dtype, p.device) for p in self._params}) > 1:  # one flat arena: one dtype, one device
            raise ValueError("SceneParallel needs every trainable parameter on one device in one dtype; "
                             "SEG3D_DDP=torch selects torch's DistributedDataParallel")
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("SEG3D_DDP_BUCKET_MB", "32")) * (1 << 20))
        self._bucket = max(int(bucket_bytes) // 4, 1)
        self._world = dist.get_world_size()
        self._avg = dist.get_backend() == "nccl"  # RCCL averages in the ring; gloo sums, the division is one pass here
        self._arena = None
        self._views = None
        self.exchanges = 0  # (tests) completed gradient exchanges
        with torch.no_grad():  # DDP's _sync_module_states: rank 0's parameters and buffers everywhere
            state = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
            by_type = {}
            for t in state:
                by_type.setdefault((t.dtype, t.device), []).append(t)
            for (_dtype, _device), group in by_type.items():
                flat = torch.cat([t.detach().reshape(-1) for t in group])
                dist.broadcast(flat, 0)
                torch._foreach_copy_([t.detach() for t in group], [c.view_as(t) for c, t in zip(flat.split([t.numel() for t in group]), group)])
        from . import ops
        ops.DEFER_WITH_GROUP = True  # this wrapper reads gradients only after the pass: the deferred join stays on

    def _make_arena(self):
        p0 = self._params[0]
        sizes = [(p.numel() + 63) // 64 * 64 for p in self._params]  # 256-byte aligned slots
        self._arena = torch.zeros((sum(sizes),), dtype=p0.dtype, device=p0.device)
        self._views = [c[: p.numel()].view_as(p) for c, p in zip(self._arena.split(sizes), self._params)]

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if not (self.training and torch.is_grad_enabled() and self._sync):
            return out
        self._queued = False
        # every differentiable tensor of the result goes through ONE _PassEnd node (dicts / lists / tuples one level deep:
        # the segmentors return a dict of logits)
        if isinstance(out, torch.Tensor):
            return _PassEnd.apply(self, out)[0] if out.requires_grad else out
        if isinstance(out, dict):
            keys = [k for k, v in out.items() if isinstance(v, torch.Tensor) and v.requires_grad]
            if keys:
                new = _PassEnd.apply(self, *[out[k] for k in keys])
                out = dict(out)
                out.update(zip(keys, new))
            return out
        if isinstance(out, (list, tuple)):
            idx = [i for i, v in enumerate(out) if isinstance(v, torch.Tensor) and v.requires_grad]
            if idx:
                new = _PassEnd.apply(self, *[out[i] for i in idx])
                seq = list(out)
                for i, v in zip(idx, new):
                    seq[i] = v
                out = type(out)(seq) if isinstance(out, tuple) else seq
            return out
        return out

    class _NoSync:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev = self.owner._sync
            self.owner._sync = False

        def __exit__(self, *exc):
            self.owner._sync = self.prev

    def no_sync(self):
        """Backward passes inside the context accumulate local gradients without an exchange (DDP.no_sync)."""
        return SceneParallel._NoSync(self)

    def _after_backward(self):
        """Final callback of the backward pass (runs on the stream the caller's backward() was issued on)."""
        self._queued = False
        from . import ops
        if self._params and self._params[0].is_cuda:
            ops.finish_deferred(self._params[0].device)  # the pass's own final join first (its callback may be queued behind ours)
        self.reduce_gradients()

    def reduce_gradients(self):
        if self._arena is None:
            self._make_arena()
        src, dst, zero = [], [], []
        for p, v in zip(self._params, self._views):
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():
                src.append(g.detach().view_as(v) if g.is_contiguous() else g.detach().contiguous().view_as(v))
                dst.append(v)
        if zero:
            torch._foreach_zero_(zero)
        if src:
            torch._foreach_copy_(dst, src)
        flat = self._arena
        works = []
        for start in range(0, flat.numel(), self._bucket):
            piece = flat[start:start + self._bucket]
            works.append(dist.all_reduce(piece, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        if not self._avg and self._world > 1:
            flat.div_(self._world)
        for p, v in zip(self._params, self._views):
            p.grad = v
        self.exchanges += 1


def wrap_data_parallel(model, device, sync_bn=False):
    """The reference's multi-GPU training wrapper (tools/train.py:246-247, 276-279): optional
    convert_sync_batchnorm, then the data-parallel wrapper.  Default: SceneParallel above (gradients exchanged once,
    after the pass, out of a flat arena; the deferred weight-gradient join stays on).  SEG3D_DDP=torch: torch's
    DistributedDataParallel -- gradient all-reduce in buckets overlapped with backward, find_unused_parameters=False
    (every parameter of the path receives a gradient each step, tests/test_gpu_training.py), broadcast_buffers=False
    (BatchNorm statistics stay per rank unless sync_bn)."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if sync_bn:
        model = convert_sync_bn(model)
    if os.environ.get("SEG3D_DDP", "native") != "torch":
        return SceneParallel(model)
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
