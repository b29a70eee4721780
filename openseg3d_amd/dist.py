"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` over RCCL/xGMI (backend "nccl";
"gloo" for the CPU rehearsal tests).  Mirrors seg3d/utils/distributed.py:8-34 (init_dist / get_dist_info).

Scenes are independent in forward and backward: the only data-path exchange is the gradient all-reduce
(DDP buckets, overlapped with backward); throughput is aggregated as sum(points) / max(time)."""
import contextlib
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
        # (the node exists only for a forward made with synchronisation on: like DistributedDataParallel, what counts is the
        # state at FORWARD time -- a backward() issued inside no_sync() for such a forward still exchanges)
        if not owner._queued:
            owner._begin_pass()
            torch.autograd.Variable._execution_engine.queue_callback(owner._after_backward)
        return (None,) + grads


def _map_tensors(obj, fn):
    """obj with every tensor leaf replaced by fn(leaf); dicts / lists / tuples (named tuples too) at any depth, as DDP's
    _find_tensors walks them.  Other objects are returned as they are."""
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        new = [(k, _map_tensors(v, fn)) for k, v in obj.items()]
        try:
            out = type(obj)()
            out.update(new)
            return out
        except TypeError:
            return dict(new)
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):
        return type(obj)(*[_map_tensors(v, fn) for v in obj])
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class SceneParallel(torch.nn.Module):
    """Data-parallel wrapper with the surface tools/train.py:276-279 uses of DistributedDataParallel (``.module``, call
    through, ``no_sync()``; parameters and buffers broadcast from rank 0 at construction), built for this path's backward:

    torch's DDP copies every gradient into its bucket view with one scaled-copy kernel per parameter (332 launches, 1.7 ms of
    the main stream per step), makes the fused optimizer walk bucket views, and -- because its reducer reads ``.grad`` on
    the main stream DURING the pass -- forbids the deferred weight-gradient join (ops._defer_join): +4.5 ms per step on ONE
    rank before a byte crosses xGMI.  Here the backward pass runs exactly as in single-process training (weight gradients
    on the side stream, one join at the end) and the exchange rides on the SIDE stream:

      * gradients live in ONE flat fp32 arena laid out in the order the gradients ARRIVE in a backward pass (learned in
        the first synchronised pass, rank 0's order broadcast to every rank: what DDP's bucket rebuild does), cut into
        slices of SEG3D_DDP_BUCKET_MB;
      * a post-accumulate hook per parameter counts arrivals.  When every gradient of a slice (and of all slices in front
        of it: every rank must issue the same collectives in the same order) has been handed to autograd, the slice is
        closed ON THE SIDE STREAM: the pending fixed-order sums of the pass so far are flushed there
        (ops.flush_deferred_jobs), one multi-tensor copy packs the slice's gradients into the arena, and the all-reduce
        is issued from the side stream's context -- RCCL's stream then waits for exactly the weight-gradient kernels in front
        of it, not for the main chain.  A hook whose gradient kernel has not run yet is fine: the copy is ordered behind it
        on the same stream (the deferred-join protocol hands autograd the buffer before its kernel; ops.deferred_alias
        names that buffer);
      * the final callback of the pass (_PassEnd -> queue_callback) completes the pass's own deferred join, closes the
        slices that are still open on the current stream, waits for all exchanges once and makes every ``.grad`` a view of
        the arena.  RCCL: ReduceOp.AVG in the ring; gloo sums and one pass divides.

    Only the last slice (the point encoder's small layers, which arrive last) is exposed; SEG3D_DDP_OVERLAP=0 restores the
    single exchange after the pass.  Gradients of parameters that took no part in the pass are reduced as zeros (every
    rank must issue the same collectives).  NOT measured on more than one GPU: no multi-GPU node was reachable in any round
    (two gloo ranks on CPU and on one card, one RCCL rank); construction runs a self-test of the collective pattern on the
    live group and wrap_data_parallel falls back to torch's DistributedDataParallel if it fails."""

    def __init__(self, module, bucket_bytes=None, overlap=None):
        super().__init__()
        self.module = module
        self._sync = True
        self._queued = False
        self._params = [p for p in module.parameters() if p.requires_grad]
        if len({(p.dtype, p.device) for p in self._params}) > 1:  # one flat arena: one dtype, one device
            raise ValueError("SceneParallel needs every trainable parameter on one device in one dtype; "
                             "SEG3D_DDP=torch selects torch's DistributedDataParallel")
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("SEG3D_DDP_BUCKET_MB", "16")) * (1 << 20))
        self._bucket = max(int(bucket_bytes) // 4, 1)
        self._overlap = (os.environ.get("SEG3D_DDP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self._world = dist.get_world_size()
        self._avg = dist.get_backend() == "nccl"  # RCCL averages in the ring; gloo sums, the division is one pass here
        self._order = list(range(len(self._params)))  # arena slot k holds parameter _order[k]
        self._learned = False
        self._arena = None
        self._views = None      # per parameter index
        self._slices = None     # [(first element, end element, [parameter indices])] in arena order
        self._slice_of = None   # per parameter index
        self._fired, self._fired_set = [], set()
        self._pending, self._next, self._works = [], 0, []
        self.exchanges = 0      # (tests) completed gradient exchanges
        self.early_slices = 0   # (tests) slices whose exchange was issued before the pass had ended
        self._self_test()
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
        # this wrapper reads a gradient only behind its kernel on the side stream, or after the pass: the deferred join stays on
        # for ITS parameters (anything else that owns a process group -- torch DDP, FSDP, comm hooks -- is not in the set)
        self._handles = []
        for i, p in enumerate(self._params):
            ops.DEFER_OWNED.add(p)
            self._handles.append(p.register_post_accumulate_grad_hook(self._arrival_hook(i)))

    def close(self):
        """Detach from the module: hooks removed, parameters no longer exempt from the process-group rule of the deferred join."""
        from . import ops
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self._params:
            ops.DEFER_OWNED.discard(p)

    def __del__(self):  # a wrapper that is dropped gives its parameters back to the by-construction rule
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _self_test(self):
        """The collective pattern of reduce_gradients on the live group, before a gradient depends on it: slices of one flat
        tensor, asynchronous, averaged (or summed and divided)."""
        if not self._params:
            return
        p0 = self._params[0]
        rank = dist.get_rank()
        t = torch.full((1024,), float(rank + 1), dtype=p0.dtype, device=p0.device)
        works = [dist.all_reduce(piece, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, async_op=True)
                 for piece in (t[:512], t[512:])]
        for w in works:
            w.wait()
        if not self._avg:
            t.div_(self._world)
        want = (self._world + 1) / 2.0
        if not bool(((t - want).abs() <= 1e-5 * want).all()):
            raise RuntimeError(f"SceneParallel self-test: all-reduce of slices gave {float(t[0])}, expected {want}")

    # ------------------------------------------------------------------------------------------ arena
    def _make_arena(self):
        p0 = self._params[0]
        sizes = [(self._params[i].numel() + 63) // 64 * 64 for i in self._order]  # 256-byte aligned slots
        self._arena = torch.zeros((sum(sizes),), dtype=p0.dtype, device=p0.device)
        self._views = [None] * len(self._params)
        self._slices, self._slice_of = [], [0] * len(self._params)
        start, members, pos = 0, [], 0
        for i, c, n in zip(self._order, self._arena.split(sizes), sizes):
            self._views[i] = c[: self._params[i].numel()].view_as(self._params[i])
            members.append(i)
            self._slice_of[i] = len(self._slices)
            pos += n
            if pos - start >= self._bucket:
                self._slices.append((start, pos, members))
                start, members = pos, []
        if members:
            self._slices.append((start, pos, members))

    def _adopt_order(self):
        """After the first synchronised pass: the arena follows the order in which gradients arrived (rank 0's, so that
        every rank cuts the same slices); parameters that did not fire go last."""
        order = self._fired + [i for i in range(len(self._params)) if i not in self._fired_set]
        t = torch.tensor(order, dtype=torch.int64, device=self._params[0].device)
        dist.broadcast(t, 0)
        order = [int(v) for v in t.tolist()]
        self._learned = True
        if sorted(order) != list(range(len(self._params))):  # (cannot happen; never trade a layout for a wrong exchange)
            return
        self._order = order
        self._arena = None

    # ------------------------------------------------------------------------------------------ pass
    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if not (self.training and torch.is_grad_enabled() and self._sync):
            return out
        self._queued = False
        # every differentiable tensor of the result goes through ONE _PassEnd node, whatever container it sits in
        leaves = []
        _map_tensors(out, lambda t: leaves.append(t) if t.requires_grad else None)
        if not leaves:
            if self._params:
                raise RuntimeError("SceneParallel: training forward returned no differentiable tensor in a dict / list / tuple "
                                   "result -- the gradient exchange could not be attached (return tensors, or train under no_sync())")
            return out
        new = iter(_PassEnd.apply(self, *leaves))
        return _map_tensors(out, lambda t: next(new) if t.requires_grad else t)

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

    def _begin_pass(self):
        self._queued = True
        self._fired, self._fired_set = [], set()
        self._works, self._next = [], 0
        if self._learned and self._overlap:
            if self._arena is None:
                self._make_arena()
            self._pending = [len(m) for _, _, m in self._slices]

    def _arrival_hook(self, i):
        def hook(param):
            if not self._queued or i in self._fired_set:
                return
            self._fired.append(i)
            self._fired_set.add(i)
            if self._learned and self._overlap:
                b = self._slice_of[i]
                self._pending[b] -= 1
                while self._next < len(self._slices) and self._pending[self._next] == 0:
                    self._close_slice(self._next, early=True)
                    self._next += 1
        hook._seg3d_scene_parallel = True  # ops._defer_join: this hook does not read the gradient on the main stream
        return hook

    def _close_slice(self, b, early):
        """Pack slice b's gradients into the arena and start its all-reduce.  early: during the pass, on the side stream
        behind the weight-gradient kernels enqueued so far; otherwise on the current stream (everything is joined)."""
        from . import ops
        start, end, members = self._slices[b]
        dev = self._params[0].device
        side = None
        if early and dev.type == "cuda":
            side = ops.side_stream(dev)
            ops.flush_deferred_jobs(dev)  # fixed-order sums pending for gradients handed over so far (on the side stream)
            side.wait_stream(torch.cuda.current_stream(dev))  # gradients autograd itself produced on the main stream
        src, dst, zero = [], [], []
        for i in members:
            p, v = self._params[i], self._views[i]
            g = ops.deferred_alias(p) if early else None  # the buffer a pending side-stream kernel writes (== .grad normally)
            if g is None:
                g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():
                src.append(g.detach().view_as(v) if g.is_contiguous() else g.detach().contiguous().view_as(v))
                dst.append(v)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if zero:
                torch._foreach_zero_(zero)
            if src:
                torch._foreach_copy_(dst, src)
            self._works.append(dist.all_reduce(self._arena[start:end], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM,
                                               async_op=True))
        if early:
            self.early_slices += 1

    def _after_backward(self):
        """Final callback of the backward pass (runs on the stream the caller's backward() was issued on)."""
        self._queued = False
        from . import ops
        if self._params and self._params[0].is_cuda:
            ops.finish_deferred(self._params[0].device)  # the pass's own final join first (its callback may be queued behind ours)
        self.reduce_gradients()

    def reduce_gradients(self):
        """Close every slice that is still open, wait for all exchanges, point ``.grad`` at the arena."""
        if self._arena is None:
            self._make_arena()
            self._works, self._next = [], 0
        for b in range(self._next, len(self._slices)):
            self._close_slice(b, early=False)
        self._next = len(self._slices)
        for w in self._works:
            w.wait()
        self._works = []
        if not self._avg and self._world > 1:
            self._arena.div_(self._world)
        for p, v in zip(self._params, self._views):
            p.grad = v
        self.exchanges += 1
        if not self._learned:
            self._adopt_order()


def wrap_data_parallel(model, device, sync_bn=False):
    """The reference's multi-GPU training wrapper (tools/train.py:246-247, 276-279): optional
    convert_sync_batchnorm, then the data-parallel wrapper.  Default: SceneParallel above (gradients exchanged slice by
    slice from the weight-gradient stream while the pass runs, out of a flat arena; the deferred weight-gradient join stays
    on) -- if its construction-time self-test of the collective pattern fails on the live group, torch's wrapper is used
    instead, with a warning.  SEG3D_DDP=torch: torch's DistributedDataParallel -- gradient all-reduce in buckets overlapped
    with backward, find_unused_parameters=False (every parameter of the path receives a gradient each step,
    tests/test_gpu_training.py), broadcast_buffers=False (BatchNorm statistics stay per rank unless sync_bn)."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if sync_bn:
        model = convert_sync_bn(model)
    if os.environ.get("SEG3D_DDP", "native") != "torch":
        try:
            return SceneParallel(model)
        except (RuntimeError, ValueError) as e:
            import warnings
            warnings.warn(f"openseg3d_amd: SceneParallel refused ({e}); using torch.nn.parallel.DistributedDataParallel")
    ids = [device.index] if device.type == "cuda" else None
    # DDP's bucket hooks read .grad while the backward pass is still running: weight gradients launched on the side stream
    # must be complete when their backward function returns (ops._WgradFork), not only at the end of the pass.  Nothing
    # needs to be switched off for that: a process group exists and DDP's parameters are not in ops.DEFER_OWNED, so
    # ops._defer_join refuses them by construction.
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=False,
                                                     broadcast_buffers=False, gradient_as_bucket_view=True)


def job_barrier(device):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
