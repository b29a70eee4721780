"""Torch-facing wrappers of the C ABI (include/seg3d_hip.h): tensors in, tensors out.

PyTorch is plumbing here -- device memory (caching allocator), the current HIP stream and
autograd bookkeeping; all compute on this path happens in libseg3d_hip.so.  Every function
requires CUDA(HIP) tensors and raises if the library is missing: there is no CPU fallback.

The names exported for reference compatibility mirror ``seg3d.ops`` (seg3d/ops/__init__.py:1-6):
``get_inner_win_inds``, ``voxel_to_point``, ``voxel_avg_pooling``, ``voxel_max_pooling``; plus
``scatter`` with the ``torch_scatter.scatter(src, index, dim=0, reduce=...)`` signature used at
vfe.py:25 and se_layer.py:25.
"""
import ctypes
import weakref
import os

import torch

from . import _lib
from ._lib import REDUCE_MAX, REDUCE_MEAN, REDUCE_SUM

_F3 = ctypes.c_float * 3
_F6 = ctypes.c_float * 6
_I3 = ctypes.c_int32 * 3


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.Seg3dError("openseg3d_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)


# Weight gradients on a second HIP stream.  In a backward function the input gradient (needed by the next function of the
# chain) and the weight gradient (needed by the optimizer) are independent; the kernels of both leave matrix pipes, CUs
# and memory pipes idle (tails of 459 workgroups on 512 slots, barrier-bound chunks), so the two co-run.  Protocol per
# backward function: fork() = the side stream waits for everything enqueued so far on the current stream (the producers of
# dy) and hands out its handle; the weight-gradient launches go there, everything else stays on the current stream;
# join() = the current stream waits for the side stream, BEFORE the function returns -- autograd, DDP hooks and the
# optimizer only ever see finished gradients.  Buffers handed to side-stream kernels are allocated on the current stream's
# pool and kept alive until join(), after which the current stream's order covers them.  SEG3D_WGRAD_STREAM=0 turns it off.
WGRAD_STREAM = os.environ.get("SEG3D_WGRAD_STREAM", "1") != "0"
_SIDE_STREAMS = {}


def side_stream(device, which=0):
    """The process's extra streams on that device: 0 = weight gradients in backward and the aux-label lookup of the
    criterion; 1 = the input pipeline's (next batch's voxelization and index plan, bench.py)."""
    side = _SIDE_STREAMS.get((device.index, which))
    if side is None:
        # SEG3D_SIDE_PRIORITY: HIP stream priority of the extra streams (the main chain is the critical path; a lower
        # priority lets its workgroups go first where both streams have some ready)
        prio = int(os.environ.get("SEG3D_SIDE_PRIORITY", "0"))
        side = _SIDE_STREAMS[(device.index, which)] = torch.cuda.Stream(device=device, priority=prio)
    return side


class _WgradFork:
    def __init__(self, device):
        self.on = WGRAD_STREAM and device.type == "cuda"
        self.keep = []
        self.jobs = []  # pending fixed-order sums of this function's partial blocks (see add_reduce)
        self.used = False
        self.device = device
        if self.on:
            self.main = torch.cuda.current_stream(device)
            self.side = side_stream(device)

    def fork(self, *keep_alive):
        """Stream handle for a weight-gradient launch whose operands are already enqueued on the current stream."""
        if not self.on:
            # (bf16 training copies are written on the side stream, _saved_rows: a launch that stays on the current stream
            # -- the stream was switched off between forward and backward -- must wait for them)
            if self.device.type == "cuda" and any(torch.is_tensor(t) and t.dtype == torch.bfloat16 for t in keep_alive):
                torch.cuda.current_stream(self.device).wait_stream(side_stream(self.device))
            return _stream()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self.used = True
        self.keep.extend(keep_alive)
        return ctypes.c_void_p(self.side.cuda_stream)

    def hold(self, *tensors):
        self.keep.extend(tensors)

    def add_reduce(self, part, chunks, n, nw, dw_ptr, db_ptr, *keep):
        """Queue the fixed-order sum of part[chunks][n] into dw[0, nw) / db[0, n - nw) (raw device addresses; db_ptr 0 =
        none).  It runs at join(): on its own when the function joins at once, in ONE launch with every other pending
        sum of the backward pass when the join is deferred to the pass's end (~160 launches of 5-7 us otherwise)."""
        self.jobs.append((part, int(chunks), int(n), int(nw), int(dw_ptr), int(db_ptr or 0), keep))

    def join(self, *grads):
        """grads: (parameter, gradient launched on the side stream) pairs -- decides whether the wait may be deferred to
        the end of the backward pass (below); gradients that are None are ignored."""
        if self.on and self.used:
            if WGRAD_DEFER and _defer_join(self, [(w, gr) for w, gr in grads if gr is not None and w is not None]):
                return
            # joining now: this function's sums, and every sum still pending from functions of THIS pass (graph task) that
            # deferred earlier -- what joins here may be ADDED to one of their gradients by the engine (a parameter used twice)
            st = _DEFERRED.get(_state_key(self.device))
            jobs = (st["jobs"] if st is not None else []) + self.jobs
            if st is not None:
                st["jobs"] = []
            if jobs:  # (partial blocks written on the current stream, e.g. LayerNorm's, must be visible to the side stream)
                self.side.wait_stream(self.main)
                _run_reduce_jobs(jobs, self.device, self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)
        elif self.jobs:
            _run_reduce_jobs(self.jobs, self.device, None)
        self.keep = []
        self.jobs = []


# Deferred join (the default in single-process training): instead of waiting at the end of every backward function, the
# current stream waits for the side stream ONCE, in a callback the autograd engine runs when the whole backward pass is
# done -- the weight gradients then also overlap the memory-bound passes between the GEMMs (norm backward, activations,
# gathers), not only their sibling input gradient: 50.1 -> 47.3 ms per step (per-function joins alone: 50.1 -> 49.9).
# The gradient tensor is handed to autograd before its kernel has run, so deferral is taken only while nothing on the
# current stream can touch it before that callback:
#   * NO process group exists in this process (torch.distributed not initialised): DistributedDataParallel, FSDP, comm
#     hooks and optimizer-in-backward all hang their work on the AccumulateGrad node (C++ post hooks a Python check
#     cannot see) and copy or reduce the gradient during the pass -- safe by construction, whoever wraps the model;
#   * the parameter has no .grad yet (AccumulateGrad then stores the tensor, no kernel; zero_grad(set_to_none=True)),
#     no tensor or post-accumulate hooks, and appears for the first time in this pass (a second use would make the engine ADD the two
#     gradients on the current stream -- the second appearance joins at once, which also covers the first);
#   * no double backward (grad mode off inside the pass);
#   * the start-up probe (probe_deferred_join, run once per process before the first training backward) found the
#     deferred result bit-identical to the joined one: the protocol leans on autograd internals (queue_callback, the
#     identity of the tensor AccumulateGrad stores), and a torch release that changes them must turn deferral off, not
#     train on unfinished buffers.
# Everything a pending launch touches (operands autograd frees when their node completes, workspaces, index tables) stays
# referenced until the callback.  The bookkeeping is stamped with the engine's graph-task id: a pass that died in an
# exception (the engine then skips its final callbacks) leaves state that the first deferral of the NEXT pass drops after
# joining the streams.  SEG3D_WGRAD_DEFER=0 turns deferral off.
WGRAD_DEFER = os.environ.get("SEG3D_WGRAD_DEFER", "1") != "0"
# Pending state per (graph task, device).  A backward pass is one graph task of the engine; a NESTED pass -- the reference
# trains its encoder layers through reentrant torch.utils.checkpoint (point_transformer_layer.py:321-337), whose backward
# runs torch.autograd.backward inside a backward function -- is another task with a larger id, has its own final callback,
# and must not touch the outer pass's pending sums and aliases: it gets its own record.  One record per device: a model
# split over two GPUs of one process has two side streams and two job tables.
_DEFERRED = {}
_DEFER_PROBE = {"done": False, "ok": None}
_MAX_IDLE_STATES = 4  # records of other graph tasks tolerated beside the current one (see _state_for)
DEFER_COUNT = 0       # joins deferred so far in this process (tests: a process whose group belongs to plain DDP must stay at 0)


def _graph_task_id():
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return fn() if fn is not None else -1


def _state_key(device):
    return (_graph_task_id(), device.index if device.index is not None else torch.cuda.current_device())


def _new_state(task, device):
    return {"keep": [], "main": None, "side": None, "seen": set(), "twice": set(), "fix": [], "task": task, "jobs": [],
            "jobs_device": device, "alias": {}}


def _reset_deferred():
    """Forget every pending record (tests; after the probe)."""
    for key in list(_DEFERRED):
        _finish_state(key)


def _finish_state(key):
    """What the engine's final callback does for the record `key`: run its pending sums, let the current stream wait for the
    side stream, repair gradients the engine cloned too early, drop the references."""
    st = _DEFERRED.pop(key, None)
    if st is None or st["side"] is None:
        return
    if st["jobs"]:  # every pending partial-block sum of the pass in one launch, behind the last producer on either stream
        st["side"].wait_stream(st["main"])
        _run_reduce_jobs(st["jobs"], st["jobs_device"], st["side"])
    st["main"].wait_stream(st["side"])
    # AccumulateGrad stores the incoming tensor itself when nobody else references it (the normal case: the gradient
    # buffers below are aliases with their own TensorImpl, made for this check) and CLONES it otherwise -- a clone taken
    # before the side stream had written the buffer.  Whatever the engine did, .grad holds the finished values from here on.
    # (A parameter that appeared a second time in the pass was joined on the spot and summed by the engine: its .grad is a
    # new tensor by right.)
    for w, alias in st["fix"]:
        g = w.grad
        if g is not None and g.data_ptr() != alias.data_ptr() and w.data_ptr() not in st["twice"]:
            g.copy_(alias.view_as(g))


def _state_for(fk):
    """The record of the current graph task on fk's device, made on first use.  Records of OTHER tasks are left alone
    while they may be alive: a smaller id is an enclosing pass (or a pass that died in an exception -- the engine then
    skips its callbacks); a LARGER id can only be a finished-or-dead nested pass (a nested pass ends before its parent
    runs another function).  Dead records hold references but nothing anybody reads; they are completed (sums run, streams
    joined -- harmless for a live one too, it only gives up overlap) once more than _MAX_IDLE_STATES pile up."""
    key = _state_key(fk.device)
    st = _DEFERRED.get(key)
    if st is None:
        task, dev = key
        for other in [k for k in _DEFERRED if k[1] == dev and k[0] > task]:
            _finish_state(other)
        older = sorted(k for k in _DEFERRED if k[1] == dev and k[0] < task)
        for other in older[: max(0, len(older) - _MAX_IDLE_STATES)]:
            _finish_state(other)
        st = _DEFERRED[key] = _new_state(task, fk.device)
    return key, st


_REDUCE_JOB = None


def _run_reduce_jobs(jobs, device, stream):
    """The queued partial-block sums (see _WgradFork.add_reduce) on `stream` (a torch stream; None = the current one): one
    launch each for up to two, otherwise ONE launch over a device job table (seg3d_reduce_partials_batched)."""
    global _REDUCE_JOB
    if not jobs:
        return
    sp = _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)
    if len(jobs) <= 2:
        for part, chunks, n, nw, dw, db, _ in jobs:
            _lib.call("seg3d_reduce_partials", _ptr(part), chunks, n, nw, ctypes.c_void_p(dw), ctypes.c_void_p(db) if db else None, sp)
        return
    import contextlib
    import numpy as np
    if _REDUCE_JOB is None:
        _REDUCE_JOB = np.dtype([("part", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("n", "<i8"), ("nw", "<i8"), ("chunks", "<i4"),
                                ("reserved", "<i4"), ("first_block", "<i8")])
        assert _REDUCE_JOB.itemsize == 56
    rec = np.zeros((len(jobs),), dtype=_REDUCE_JOB)
    first = 0
    for i, (part, chunks, n, nw, dw, db, _) in enumerate(jobs):
        rec[i] = (part.data_ptr() if chunks else 0, dw, db, n, nw, chunks, 0, first)
        first += (n + 255) // 256
    # pinned staging block from torch's caching host allocator: it is not handed out again before the copy below has run
    host = torch.empty((rec.nbytes,), dtype=torch.uint8, pin_memory=True)
    host.numpy()[:] = rec.view(np.uint8)
    with (contextlib.nullcontext() if stream is None else torch.cuda.stream(stream)):
        table = host.to(device, non_blocking=True)
        _lib.call("seg3d_reduce_partials_batched", _ptr(table), len(jobs), first, sp)


# Parameters owned by a live dist.SceneParallel: that wrapper reads a gradient only behind its kernel on the side stream or
# after the backward pass (its post-accumulate hooks are marked, see _foreign_hooks), so deferral is safe for THESE
# parameters although a process group exists.  Anything else that owns a process group (torch DDP built directly, FSDP, comm
# hooks) is not in the set and keeps the by-construction rule -- also when it is built later in the same process.
class _IdentitySet:
    """Weak set of tensors by IDENTITY (a WeakSet would compare members with ==, which is elementwise on tensors)."""

    def __init__(self):
        self._refs = {}

    def add(self, t):
        key = id(t)
        self._refs[key] = weakref.ref(t, lambda _r, key=key, refs=self._refs: refs.pop(key, None))

    def discard(self, t):
        r = self._refs.get(id(t))
        if r is not None and r() is t:
            del self._refs[id(t)]

    def __contains__(self, t):
        r = self._refs.get(id(t))
        return r is not None and r() is t

    def __len__(self):
        return len(self._refs)


DEFER_OWNED = _IdentitySet()


def _process_group_exists():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _foreign_hooks(base):
    """True if the parameter carries a post-accumulate hook that is not SceneParallel's arrival counter."""
    hooks = getattr(base, "_post_accumulate_grad_hooks", None)
    return bool(hooks) and not all(getattr(h, "_seg3d_scene_parallel", False) for h in hooks.values())


def deferred_alias(w):
    """The buffer a pending deferred weight gradient of parameter `w` is being written to on the side stream (None if
    nothing is pending for it): what a reader on the SIDE stream may consume before the pass ends."""
    for st in _DEFERRED.values():
        alias = st["alias"].get(id(w))
        if alias is not None:
            return alias
    return None


def flush_deferred_jobs(device):
    """Run the fixed-order sums queued so far by the device's pending records NOW, on the side stream (behind every producer
    enqueued on either stream): the gradients handed to autograd so far are then complete in side-stream order.  The
    streams are not joined; the partial blocks stay referenced until the record completes."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    for key, st in list(_DEFERRED.items()):
        if key[1] == dev and st["side"] is not None and st["jobs"]:
            st["side"].wait_stream(st["main"])
            _run_reduce_jobs(st["jobs"], st["jobs_device"], st["side"])
            st["keep"].append(st["jobs"])
            st["jobs"] = []


def finish_deferred(device):
    """Complete every pending deferred-join record of the device NOW (sums run, streams joined): for a caller that needs the
    pass's gradients inside a final callback queued before the pass's own (dist.SceneParallel)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    for key in [k for k in _DEFERRED if k[1] == dev]:
        _finish_state(key)


def _defer_join(fk, grads):
    if torch.is_grad_enabled() or not grads or _DEFER_PROBE["ok"] is False:
        return False
    st = _DEFERRED.get(_state_key(fk.device))  # (made below, only once something is really deferred)
    bases = [w if w._base is None else w._base for w, _ in grads]
    if _process_group_exists() and not all(base in DEFER_OWNED for base in bases):
        return False
    if st is not None and any(base.data_ptr() in st["seen"] for base in bases):
        st["twice"].update(base.data_ptr() for base in bases)  # second use: this call joins now, the engine sums
        return False
    for base in bases:
        if (not base.is_leaf or base.grad is not None or base._backward_hooks or not base.is_contiguous()
                or _foreign_hooks(base)):
            return False
    if st is None:  # first deferral of this graph task: its final callback completes the record
        key, st = _state_for(fk)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda key=key: _finish_state(key))
        except RuntimeError:  # not inside a backward pass of the engine (a direct call): join now
            _DEFERRED.pop(key, None)
            return False
    for base, (w, gr) in zip(bases, grads):
        st["seen"].add(base.data_ptr())
        if w is base:
            st["fix"].append((base, gr.detach()))
            st["alias"][id(base)] = st["fix"][-1][1]
        else:
            st["keep"].append(gr.detach())  # gradient of a view of the parameter: autograd scatters it, no .grad to check
    global DEFER_COUNT
    DEFER_COUNT += 1
    st["main"], st["side"] = fk.main, fk.side
    st["keep"].append(fk.keep)
    st["jobs"].extend(fk.jobs)
    fk.keep = []
    fk.jobs = []
    return True


def probe_deferred_join(device):
    """Start-up self-test of the deferred join (once per process; the segmentors call it before their first training
    forward).  Two Linear layers are differentiated twice from the same inputs: with every weight gradient joined at the end
    of its backward function, and deferred -- with the side stream held busy by a long kernel so that a gradient read before
    the final join cannot be complete, and the freed gradient buffers poisoned so that it cannot be right by accident.
    Bit-identical results keep deferral on; anything else (a torch release whose engine no longer runs the callback, or
    clones the gradient where it used to keep it) turns it off for the process with a warning.  Returns the verdict."""
    global WGRAD_DEFER
    pr = _DEFER_PROBE
    if pr["done"] or not (WGRAD_STREAM and WGRAD_DEFER) or device.type != "cuda":
        return pr["ok"]
    pr["done"] = True
    if _process_group_exists() and not len(DEFER_OWNED):  # never deferred in such a process anyway
        return None
    before = WGRAD_DEFER
    try:
        with torch.cuda.device(device), torch.enable_grad():
            gen = torch.Generator().manual_seed(11)
            m, c = 16384, 64
            x = torch.randn(m, c, generator=gen).to(device)
            g = torch.randn(m, c, generator=gen).to(device)
            w0 = [(torch.randn(c, c, generator=gen) / 8.0).to(device) for _ in range(2)]
            b0 = [(torch.randn(c, generator=gen) * 0.1).to(device) for _ in range(2)]

            def grads(defer):
                global WGRAD_DEFER
                ws = [t.clone().requires_grad_() for t in w0]
                bs = [t.clone().requires_grad_() for t in b0]
                if _process_group_exists():  # a dist.SceneParallel owns this process's deferral: probe under its rule
                    for t in ws + bs:
                        DEFER_OWNED.add(t)
                WGRAD_DEFER = defer
                try:
                    if defer:
                        side = side_stream(device)
                        side.wait_stream(torch.cuda.current_stream(device))
                        with torch.cuda.stream(side):  # ~30 ms of work in front of the weight gradients -- far longer than
                            # the host needs to enqueue the pass and the copies below (no use of the global generators:
                            # the probe must not shift the run's DropPath / dropout draws)
                            if hasattr(torch.cuda, "_sleep"):
                                torch.cuda._sleep(3_000_000)  # spins on s_memtime: 100 MHz on gfx950
                            else:
                                big = torch.full((4096, 4096), 1e-4, device=device)
                                for _ in range(24):
                                    big = big @ big
                        junk = [torch.full((c, c), float("nan"), device=device) for _ in range(4)]
                        del junk  # the blocks the gradient buffers are most likely to be carved from
                    y = linear(torch.relu(linear(x, ws[0], bs[0])), ws[1], bs[1])
                    y.backward(g)
                finally:
                    WGRAD_DEFER = before
                # what the optimizer would read: copies enqueued on the CURRENT stream right behind the pass, no device
                # synchronisation in between -- a missing join shows as the poison (or as stale bytes) in these copies
                return [None if t.grad is None else t.grad.clone() for t in ws + bs]
            want = grads(False)
            got = grads(True)
            torch.cuda.synchronize(device)
            ok = all(a is not None and torch.equal(a, b) for a, b in zip(got, want))
    except Exception as e:  # noqa: BLE001 -- any failure of the probe means: do not defer
        ok = False
        import warnings
        warnings.warn(f"openseg3d_amd: deferred weight-gradient join probe raised {type(e).__name__}: {e}")
    pr["ok"] = bool(ok)
    if not ok:
        WGRAD_DEFER = False
        import warnings
        warnings.warn("openseg3d_amd: the deferred weight-gradient join did not reproduce the joined gradients on this "
                      "torch build; falling back to per-function joins (SEG3D_WGRAD_DEFER=0)")
    return pr["ok"]


def _i3(v):
    return _I3(*[int(x) for x in v])


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _i32c(t):
    return t.contiguous() if t.dtype == torch.int32 else t.to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------ a1/a2/a4
def grid_size(voxel_size, point_cloud_range):
    """VoxelGenerator.__init__ grid (x, y, z) -- voxel_generator.py:15-18 (host helper)."""
    out = _I3()
    _lib.call("seg3d_grid_size", _F3(*[float(v) for v in voxel_size]),
              _F6(*[float(v) for v in point_cloud_range]), out)
    return [int(v) for v in out]


def voxelize(points, voxel_size, point_cloud_range, xyz_col=0, batch_col=-1):
    """Hard voxelisation of a (collated) point tensor on the GPU.

    points: float32/float64 [N, D] CUDA.  Returns (voxel_coords int32 [M,4] as (b,z,y,x) in
    first-seen order, point_voxel_ids int32 [N], -1 = out of range).  One host sync (reads M).
    """
    _need_gpu(points)
    if points.dim() != 2 or points.dtype not in (torch.float32, torch.float64):
        raise _lib.Seg3dError("points must be a float32/float64 [N, D] tensor")
    points = points.contiguous()
    n, d = points.shape
    dev = points.device
    coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    ids = torch.empty((n,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = _lib.query("seg3d_voxelize_workspace_bytes", n)
    ws = _workspace(nbytes, dev)
    fn = "seg3d_voxelize_f32" if points.dtype == torch.float32 else "seg3d_voxelize_f64"
    _lib.call(fn, _ptr(points), n, d, int(xyz_col), int(batch_col), _F3(*[float(v) for v in voxel_size]),
              _F6(*[float(v) for v in point_cloud_range]), _ptr(coords), _ptr(ids), _ptr(count), _ptr(ws),
              ws.numel(), _stream())
    m = int(count.item())
    return coords[:m], ids


def voxelize_host(points, voxel_size, point_cloud_range, xyz_col=0, batch_col=-1):
    """The CPU entry of the voxelizer (seg3d_voxelize_host_f32 / _f64; include/seg3d_hip.h): what VoxelGenerator.generate
    does in a DataLoader worker or in test-time augmentation, where there is no GPU context (voxel_generator.py:24-26,
    waymo_dataset.py:275, test_time_aug.py:33).  points: float32 / float64 numpy [N, D].  Returns numpy
    (voxel_coords int32 [M, 4] as (b, z, y, x) in first-seen order, point_voxel_ids int32 [N], -1 = out of range).
    Host code of the same library; it makes no HIP call."""
    import numpy as np
    pts = np.ascontiguousarray(points)
    if pts.ndim != 2 or pts.dtype not in (np.float32, np.float64):
        raise _lib.Seg3dError("points must be a float32/float64 [N, D] array")
    n, d = pts.shape
    coords = np.empty((max(n, 1), 4), dtype=np.int32)
    ids = np.empty((n,), dtype=np.int32)
    count = np.zeros((1,), dtype=np.int32)
    ws = np.empty((max(int(_lib.query("seg3d_voxelize_host_workspace_bytes", n)), 8),), dtype=np.uint8)

    def host(a):
        return ctypes.c_void_p(a.ctypes.data)

    fn = "seg3d_voxelize_host_f32" if pts.dtype == np.float32 else "seg3d_voxelize_host_f64"
    _lib.call(fn, host(pts), n, d, int(xyz_col), int(batch_col), _F3(*[float(v) for v in voxel_size]),
              _F6(*[float(v) for v in point_cloud_range]), host(coords), host(ids), host(count), host(ws), ws.size)
    return coords[: int(count[0])], ids


def cart2polar(points, xyz_col=0):
    """``cart2polar`` + the row re-assembly of the cylinder configs (pointops_utils.py:8-11, waymo_dataset.py:270-273)
    on the device: float32/float64 [N, D] rows [.., x, y, z, f..] -> [N, D + 2] rows [.., rho, phi, z, x, y, f..]."""
    _need_gpu(points)
    if points.dim() != 2 or points.dtype not in (torch.float32, torch.float64):
        raise _lib.Seg3dError("points must be a float32/float64 [N, D] tensor")
    points = points.contiguous()
    n, d = points.shape
    out = torch.empty((n, d + 2), dtype=points.dtype, device=points.device)
    fn = "seg3d_cart2polar_f32" if points.dtype == torch.float32 else "seg3d_cart2polar_f64"
    _lib.call(fn, _ptr(points), n, d, int(xyz_col), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------------------ a14
def group_index(group_ids, n_groups, rank=True, order=True, offsets=True):
    """Deterministic in-group rank and CSR of ``group_ids`` (int32 [n], -1 = skip)."""
    _need_gpu(group_ids)
    g = _i32c(group_ids)
    n, dev = g.shape[0], g.device
    r = torch.empty((n,), dtype=torch.int32, device=dev) if rank else None
    o = torch.empty((n,), dtype=torch.int32, device=dev) if order else None
    off = torch.empty((n_groups + 1,), dtype=torch.int32, device=dev) if offsets else None
    ws = _workspace(_lib.query("seg3d_group_index_workspace_bytes", n, n_groups), dev)
    _lib.call("seg3d_group_index", _ptr(g), n, int(n_groups), _ptr(r), _ptr(o), _ptr(off), _ptr(ws), ws.numel(),
              _stream())
    return r, o, off


def get_inner_win_inds(group_inds):
    """``seg3d.ops.get_inner_win_inds`` (ingroup_inds.py:7-20): int64 [N] -> int64 [N], non-differentiable.
    Like the reference launcher (ingroup_inds.cpp:36) it reads max(group)+1 on the host."""
    _need_gpu(group_inds)
    if group_inds.numel() == 0:
        return torch.zeros_like(group_inds)
    ng = int(group_inds.max().item()) + 1
    r, _, _ = group_index(group_inds.to(torch.int32), ng, rank=True, order=False, offsets=False)
    return r.to(group_inds.dtype)


class SegmentIndex:
    """CSR of rows grouped by segment id (ids int32 [n], -1 = in no segment)."""

    def __init__(self, ids, n_segments):
        self.ids = _i32c(ids)
        self.n_segments = int(n_segments)
        _, self.order, self.offsets = group_index(self.ids, self.n_segments, rank=False)


# ------------------------------------------------------------------------------------------ a8-a11 rulebooks
class CoordHash:
    def __init__(self, coords, spatial_shape):
        _need_gpu(coords)
        self.coords = _i32c(coords)
        self.shape = [int(s) for s in spatial_shape]
        m = self.coords.shape[0]
        self.table = _workspace(_lib.query("seg3d_coord_hash_bytes", m), coords.device)
        _lib.call("seg3d_coord_hash_build", _ptr(self.coords), m, _i3(self.shape), _ptr(self.table),
                  self.table.numel(), _stream())


def rulebook_subm(h):
    m = h.coords.shape[0]
    nbr = torch.empty((27, m), dtype=torch.int32, device=h.coords.device)
    _lib.call("seg3d_rulebook_subm", _ptr(h.coords), m, _i3(h.shape), _ptr(h.table), h.table.numel(), _ptr(nbr),
              _stream())
    return nbr


# a9 "row image" schedule of the submanifold convs (csrc/spconv_tile.hip): 0 = off, 1 = every layer the schedule takes.
# (Bit-identical to the per-pair gather except in the 32 / 48-channel layers, whose offset-split layouts sum in another
# fixed order: the switch moves those layers' results by fp32 round-off.)
# SEG3D_CONV_TILED_MIN_C: smallest max(cin, cout) that takes it (narrower layers keep the per-pair gather).
CONV_TILED = os.environ.get("SEG3D_CONV_TILED", "1") != "0"
CONV_TILED_MIN_C = int(os.environ.get("SEG3D_CONV_TILED_MIN_C", "0"))


class ConvPlan:
    """Tile plan of one neighbour table (seg3d_conv_plan_build): Morton-ordered 128-row tiles with their distinct input
    rows and per-(offset, row) image slots.  Built once per site level, used by every layer on it, forward and dgrad."""

    def __init__(self, coords, nbr):
        _need_gpu(coords, nbr)
        coords = _i32c(coords)
        m = nbr.shape[1]
        if coords.shape[0] != m or nbr.shape[0] != 27 or nbr.dtype != torch.int32 or not nbr.is_contiguous():
            raise _lib.Seg3dError("ConvPlan: coords [m, 4] and a contiguous int32 table [27, m] of the same sites expected")
        self.nbr, self.m = nbr, m
        self.data = torch.empty((max(_lib.query("seg3d_conv_plan_bytes", m), 1),), dtype=torch.uint8, device=nbr.device)
        ws_bytes = _lib.query("seg3d_conv_plan_workspace_bytes", m)
        ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=nbr.device)
        _lib.call("seg3d_conv_plan_build", _ptr(coords), _ptr(nbr), m, _ptr(self.data), _ptr(ws), ws_bytes, _stream())


def _tiled_fits(plan, packed, cin, cout):
    return (plan is not None and CONV_TILED and bool(packed.flags & PACK_SPLIT_BF16) and max(cin, cout) >= CONV_TILED_MIN_C
            and bool(_lib.load().seg3d_spconv_tiled_supported(int(cin), int(cout))))


def _conv_tiled(x, plan, packed, bias, addend, relu, cin, cout, y):
    _lib.call("seg3d_spconv_fwd_tiled", _ptr(x), _ptr(plan.nbr), _ptr(plan.data), plan.m, x.shape[0], _ptr(packed.data),
              packed.flags, _ptr(bias), _ptr(addend), int(bool(relu)), cin, cout, _ptr(y), _stream())
    return y


def downsample_launch(coords, batch_size, spatial_shape, rows_dev=None):
    """Queue the output-site build of SparseConv3d(k=3, s=2, p=1) without reading anything back: returns (buffer [cap, 4],
    count int32 [1] on the device, shape_out).  ``rows_dev``: device count of valid rows when ``coords`` is itself such a
    buffer (chained levels)."""
    _need_gpu(coords)
    coords = _i32c(coords)
    m, dev = coords.shape[0], coords.device
    shape_out = [(int(s) + 2 - 3) // 2 + 1 for s in spatial_shape]
    cells = int(batch_size) * shape_out[0] * shape_out[1] * shape_out[2]
    cap = max(min(8 * m, cells), 1)
    out = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = _workspace(_lib.query("seg3d_downsample_workspace_bytes", int(batch_size), _i3(spatial_shape)), dev)
    _lib.call("seg3d_downsample_coords", _ptr(coords), m, _ptr(rows_dev), int(batch_size), _i3(spatial_shape), _ptr(out), cap,
              _ptr(count), _ptr(ws), ws.numel(), _stream())
    return out, count, shape_out


def downsample_coords(coords, batch_size, spatial_shape):
    """Active sites of SparseConv3d(k=3, s=2, p=1), ascending (b,z,y,x).  One host sync (reads M_out)."""
    out, count, shape_out = downsample_launch(coords, batch_size, spatial_shape)
    return out[: int(count.item())], shape_out


def downsample_chain(coords, batch_size, spatial_shape, levels):
    """Output sites of ``levels`` successive strided convs with ONE host read-back: level k + 1 is built from level k's
    buffer and its device-side count.  Returns [(coords_k, shape_k)] for k = 1 .. levels."""
    bufs, counts, shapes = [], [], []
    cur, cur_shape, rows = coords, spatial_shape, None
    for _ in range(levels):
        out, count, shape_out = downsample_launch(cur, batch_size, cur_shape, rows)
        bufs.append(out), counts.append(count), shapes.append(shape_out)
        cur, cur_shape, rows = out, shape_out, count
    ns = torch.cat(counts).tolist()
    return [(b[: int(n)], sh) for b, n, sh in zip(bufs, ns, shapes)]


def parity_order(coords):
    """int32 [M] permutation: rows of coords [M, 4] grouped by the parity of (z, y, x), stable inside a group."""
    _need_gpu(coords)
    coords = _i32c(coords)
    m = coords.shape[0]
    order = torch.empty((m,), dtype=torch.int32, device=coords.device)
    ws_bytes = _lib.query("seg3d_knn_level_workspace_bytes", m)
    ws = _workspace(ws_bytes, coords.device)
    _lib.call("seg3d_parity_order", _ptr(coords), m, _ptr(order), _ptr(ws), ws_bytes, _stream())
    return order


def rulebook_strided(h_in, coords_out):
    m_in, m_out = h_in.coords.shape[0], coords_out.shape[0]
    dev = coords_out.device
    fwd = torch.empty((27, m_out), dtype=torch.int32, device=dev)
    inv = torch.empty((27, m_in), dtype=torch.int32, device=dev)
    _lib.call("seg3d_rulebook_strided", _ptr(coords_out), m_out, m_in, _i3(h_in.shape), _ptr(h_in.table),
              h_in.table.numel(), _ptr(fwd), _ptr(inv), _stream())
    return fwd, inv


# ------------------------------------------------------------------------------------------ a9-a11 conv
PACK_FWD, PACK_T, PACK_T_FLIP = 0, 1, 3
PACK_SPLIT_BF16 = 4

# Arithmetic of the sparse-conv GEMMs: "bf16x3" = every fp32 operand split into bf16 hi + lo, products as
# three bf16 MFMAs with fp32 accumulation (~2^-16 relative, 16/3 the fp32-MFMA rate); "fp32" = exact fp32
# MFMA (v_mfma_f32_16x16x4_f32).  Both pass the 1e-3 logit parity test (tests/test_gpu_parity.py).
CONV_PRECISION = os.environ.get("SEG3D_CONV_PRECISION", "bf16x3")


def _precision_flag():
    if CONV_PRECISION == "bf16x3":
        return PACK_SPLIT_BF16
    if CONV_PRECISION == "fp32":
        return 0
    raise _lib.Seg3dError(f"SEG3D_CONV_PRECISION must be 'bf16x3' or 'fp32', got {CONV_PRECISION!r}")


# ---- when is a cached operand stale?  A tensor's ``_version`` counts in-place autograd-visible writes (copy_,
# load_state_dict, foreach optimizers) but NOT the fused optimizers (torch._fused_sgd_ / _fused_adamw_ update the
# parameters without touching the counter) nor writes through ``.data``.  Every cache below therefore keys on
# (version, WEIGHT_EPOCH): the epoch is bumped by a global optimizer post-step hook -- any torch optimizer, fused or
# not -- and by invalidate_weight_caches() for code that writes parameters behind autograd's back.
_WEIGHT_EPOCH = [0]


def invalidate_weight_caches():
    """Call after changing parameters in a way neither ``Tensor._version`` nor an optimizer step reveals (writes
    through ``.data``, raw pointers): every packed / folded operand is rebuilt at its next use."""
    _WEIGHT_EPOCH[0] += 1


def _stamp(t):
    return (t._version, _WEIGHT_EPOCH[0])


def _register_optimizer_hook():
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(lambda *_: invalidate_weight_caches())


_register_optimizer_hook()


# ---- pack registry: every (parameter [slice], operand form) that has been packed once is remembered; when a pack
# is requested and the parameter has changed since (every optimizer step in training), ALL stale packs are
# refreshed by one seg3d_pack_weights_batched launch instead of ~220 small ones per step.
class _PackJob:
    __slots__ = ("owner", "src_ptr", "out", "cin", "cout", "kk", "transpose", "flip", "version", "blocks")


_PACK_JOBS = {}
_PACK_DESC = {}  # tuple of job keys -> (device descriptor tensor, total blocks)


def _cached_pack(weight, kk, transpose, flip):
    """Split-bf16 pack of a contiguous fp32 parameter (or row slice of one) through the registry."""
    owner = weight._base if weight._base is not None else weight
    key = (id(owner), weight.storage_offset(), tuple(weight.shape), kk, transpose, flip)
    job = _PACK_JOBS.get(key)
    if job is None or job.owner() is not owner or job.src_ptr != weight.data_ptr():
        cout, cin = weight.shape[0], weight.shape[-1]
        job = _PackJob()
        job.owner, job.src_ptr = weakref.ref(owner), weight.data_ptr()
        job.cin, job.cout, job.kk, job.transpose, job.flip = cin, cout, kk, int(transpose), int(flip)
        nbytes = (_lib.query("seg3d_spconv_packed_bytes", cin, cout, 4 | job.transpose) if kk == 27
                  else _lib.query("seg3d_linear_packed_bytes", cin, cout, job.transpose))
        job.out = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
        job.blocks = (nbytes // 32 + 255) // 256  # one work item = 16 packed bf16 (a lane's hi and lo fragments)
        job.version = None
        _PACK_JOBS[key] = job
        _PACK_DESC.clear()
    if job.version != _stamp(owner):
        _refresh_packs(weight.device)
    return job.out


def _refresh_packs(device):
    stale, dead = [], []
    for key, job in _PACK_JOBS.items():
        owner = job.owner()
        if owner is None:
            dead.append(key)
        elif job.version != _stamp(owner) and job.out.device == device:
            stale.append((key, job, owner))
    for key in dead:
        del _PACK_JOBS[key]
    if dead:
        _PACK_DESC.clear()
    if not stale:
        return
    ident = tuple(k for k, _, _ in stale)
    hit = _PACK_DESC.get(ident)
    if hit is None:
        import numpy as np
        rec = np.zeros((len(stale),), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("cin", "<i4"), ("cout", "<i4"),
                                                       ("kk", "<i4"), ("transpose", "<i4"), ("flip", "<i4"),
                                                       ("reserved", "<i4"), ("first_block", "<i8")]))
        first = 0
        for i, (_, job, _) in enumerate(stale):
            rec[i] = (job.src_ptr, job.out.data_ptr(), job.cin, job.cout, job.kk, job.transpose, job.flip, 0, first)
            first += job.blocks
        desc = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
        hit = (desc, first)
        if len(_PACK_DESC) > 8:
            _PACK_DESC.clear()
        _PACK_DESC[ident] = hit
    _lib.call("seg3d_pack_weights_batched", _ptr(hit[0]), len(stale), hit[1], _stream())
    for _, job, owner in stale:
        job.version = _stamp(owner)


def _registry_ok(weight):
    return (CONV_PRECISION == "bf16x3" and weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()
            and not weight.is_inference())


class PackedWeight:
    __slots__ = ("data", "flags")

    def __init__(self, data, flags):
        self.data, self.flags = data, flags


def pack_weight(weight, flags, use_registry=True):
    """weight [Cout,3,3,3,Cin] (or [Cout,27,Cin]) -> MFMA B-fragment stream for seg3d_spconv_fwd."""
    flags = int(flags) | _precision_flag()
    if use_registry and _registry_ok(weight) and (weight if weight._base is None else weight._base).is_leaf and weight.shape[-1] % 16 == 0:
        return PackedWeight(_cached_pack(weight, 27, flags & 1, (flags >> 1) & 1), flags)
    w = _f32c(weight)
    cout, cin = w.shape[0], w.shape[-1]
    nbytes = _lib.query("seg3d_spconv_packed_bytes", cin, cout, flags)
    out = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    _lib.call("seg3d_spconv_pack_weight", _ptr(w), cin, cout, flags, _ptr(out), _stream())
    return PackedWeight(out, flags)


def debug_set_conv_nbt(nbt):
    """Test hook (seg3d_debug_set_conv_nbt): force the column-block width (x16) of the split-bf16 gather-GEMM;
    0 restores the automatic choice.  Results never depend on it."""
    _lib.call("seg3d_debug_set_conv_nbt", int(nbt))


# Wide layers (cin >= 192), alternative schedule: operands pre-split once per layer, gather-GEMM fed by LDS-DMA
# (csrc/spconv_dma.hip).  Measured on the headline scene it ties the register-staged kernel (-7 % on the 58 k-row layers,
# +2 % on the 19 k-row layers including its conversion pass: both are bound by the 1.2 rounds of workgroups those levels
# give 256 CUs, not by the operand path), so it is opt-in: SEG3D_CONV_DMA=1
CONV_DMA = os.environ.get("SEG3D_CONV_DMA", "0") == "1"


def _conv_dma_fits(packed, cin, cout):
    return CONV_DMA and (packed.flags & PACK_SPLIT_BF16) and cin >= 192 and cin % 32 == 0 and cout % 96 == 0


def _conv_dma(x, nbr, packed, bias, addend, relu, cin, cout, order, y):
    m_in, m_out = x.shape[0], nbr.shape[1]
    xs = torch.empty((_lib.query("seg3d_spconv_presplit_bytes", m_in, cin),), dtype=torch.uint8, device=x.device)
    _lib.call("seg3d_spconv_presplit", _ptr(x), m_in, cin, _ptr(xs), _stream())
    _lib.call("seg3d_spconv_fwd_presplit", _ptr(xs), _ptr(nbr), m_out, m_in, _ptr(packed.data), packed.flags, _ptr(bias),
              _ptr(addend), int(bool(relu)), cin, cout, _ptr(y), _ptr(order), _stream())
    return y


def _conv_apply(x, nbr, packed, bias, cin, cout, order=None, plan=None):
    m_out = nbr.shape[1]
    y = torch.empty((m_out, cout), dtype=torch.float32, device=x.device)
    if _tiled_fits(plan, packed, cin, cout):
        return _conv_tiled(x, plan, packed, bias, None, False, cin, cout, y)
    if _conv_dma_fits(packed, cin, cout):
        return _conv_dma(x, nbr, packed, bias, None, False, cin, cout, order, y)
    _lib.call("seg3d_spconv_fwd", _ptr(x), _ptr(nbr), m_out, x.shape[0], _ptr(packed.data), packed.flags, _ptr(bias),
              cin, cout, _ptr(y), _ptr(order), _stream())
    return y


# Storage of the sparse-conv feature maps in INFERENCE: "fp32" (default: the path the 1e-3 logit bar is stated for) or "bf16"
# (opt-in, BASELINE configs[4] names bf16; SEG3D_STORAGE=bf16 or bench.py --storage bf16).  Which tensors may be rounded was
# decided per op family from tools/bf16_storage_probe.py (max |dlogit| at max |logit| 207 / arg-max changes, headline
# scene): sparse-conv outputs 3.4e-2 / 0.02 %; attention 1.1e-2; voxel-path Linear 1.8e-2; the residual stream behind
# the LayerNorms 1.95 / 0.44 %; the per-point MLPs 2.4 / 0.51 % -- so the mode covers the conv feature maps (the
# gather-bound tensors of the dense configuration) and nothing else.
STORAGE = os.environ.get("SEG3D_STORAGE", "fp32")
STORAGE_ROUND_INPUTS = os.environ.get("SEG3D_STORAGE_ROUND_INPUTS", "1") != "0"


# Opt-in bf16 COPIES for the backward pass (BASELINE configs[4] names bf16; SURVEY D7: build-defined).  With
# SEG3D_TRAIN_STORAGE=bf16 the sparse-conv / Linear / encoder-layer functions save a bf16 copy of the rows their WEIGHT
# gradient will multiply with instead of the fp32 tensor (which is then free to be released as soon as the forward has moved
# on): every tensor of the autograd graph stays fp32, so every gradient stays fp32 -- what changes is the precision of one
# operand of the weight-gradient products (8 significant bits instead of 16) and their cost (a bf16 row is its own high
# half: 8-byte gathers, no split, two MFMAs per product).  Tolerance stated against this repo's own fp32-copy path in
# tests/test_gpu_training.py::test_bf16_training_copies_stay_within_their_stated_tolerance.
TRAIN_STORAGE = os.environ.get("SEG3D_TRAIN_STORAGE", "fp32")


def _saved_rows(x):
    """What a training forward keeps of the rows `x` for its weight gradient: x itself, or its bf16 copy (opt-in).  The
    copy is made on the WEIGHT-GRADIENT stream: that stream idles during the forward pass and is the only reader of the copy
    (the weight-gradient kernels of the backward pass are launched there, in order behind it), so the conversion leaves the
    forward's main chain and needs no join (on the main chain the ~150 conversion passes of a dense 2 M-point scene cost
    19 ms per step: 288 -> 307 ms)."""
    if TRAIN_STORAGE not in ("fp32", "bf16"):
        raise _lib.Seg3dError(f"SEG3D_TRAIN_STORAGE must be 'fp32' or 'bf16', got {TRAIN_STORAGE!r}")
    if not (TRAIN_STORAGE == "bf16" and CONV_PRECISION == "bf16x3" and x.dtype == torch.float32 and x.is_cuda):
        return x
    if not WGRAD_STREAM:
        return x.to(torch.bfloat16)
    main, side = torch.cuda.current_stream(x.device), side_stream(x.device)
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        y = x.to(torch.bfloat16)  # (allocated from the side stream's pool: the backward's readers run on that stream)
    x.record_stream(side)  # x's block must not be handed to a later main-stream tensor before the conversion has run
    return y


def conv_storage_bf16():
    if STORAGE not in ("fp32", "bf16"):
        raise _lib.Seg3dError(f"SEG3D_STORAGE must be 'fp32' or 'bf16', got {STORAGE!r}")
    return STORAGE == "bf16" and CONV_PRECISION == "bf16x3" and not torch.is_grad_enabled()


def conv_act(x, nbr, packed, bias, cin, cout, order=None, addend=None, relu=True, plan=None):
    """act(conv(x) + bias (+ addend)) in one launch (inference form of a conv block, seg3d_spconv_fwd_act); in the bf16
    storage mode the output (and the residual it adds) is bf16, the input float32 or bf16 as it comes."""
    if conv_storage_bf16() and not _conv_dma_fits(packed, cin, cout):
        # a float32 input (a SWFormer stage's output, the VFE rows, a concatenation) is rounded once here: the conv gathers
        # every row ~7-17 times, so the one conversion pass is repaid by the halved gather (dense scene, 192 -> 96 @1.2 M
        # rows: 2.9 -> 1.9 ms with bf16 rows); SEG3D_STORAGE_ROUND_INPUTS=0 keeps float32 inputs as they come
        if x.dtype == torch.float32 and STORAGE_ROUND_INPUTS:
            xin = x.to(torch.bfloat16).contiguous()
        else:
            xin = x.contiguous() if x.dtype in (torch.float32, torch.bfloat16) else x.float().contiguous()
        m_out = nbr.shape[1]
        y = torch.empty((m_out, cout), dtype=torch.bfloat16, device=x.device)
        res = None if addend is None else addend.to(torch.bfloat16).contiguous()
        if _tiled_fits(plan, packed, cin, cout):
            _lib.call("seg3d_spconv_fwd_tiled_bf16", _ptr(xin), int(xin.dtype == torch.bfloat16), _ptr(plan.nbr), _ptr(plan.data),
                      plan.m, xin.shape[0], _ptr(packed.data), packed.flags, _ptr(bias), _ptr(res), int(bool(relu)), cin, cout,
                      _ptr(y), _stream())
            return y
        _lib.call("seg3d_spconv_fwd_act_bf16", _ptr(xin), int(xin.dtype == torch.bfloat16), _ptr(nbr), m_out, xin.shape[0],
                  _ptr(packed.data), packed.flags, _ptr(bias), _ptr(res), int(bool(relu)), cin, cout, _ptr(y), _ptr(order),
                  _stream())
        return y
    x = _f32c(x)
    m_out = nbr.shape[1]
    y = torch.empty((m_out, cout), dtype=torch.float32, device=x.device)
    addend = None if addend is None else _f32c(addend)
    if _tiled_fits(plan, packed, cin, cout):
        return _conv_tiled(x, plan, packed, bias, addend, relu, cin, cout, y)
    if _conv_dma_fits(packed, cin, cout):
        return _conv_dma(x, nbr, packed, bias, addend, relu, cin, cout, order, y)
    _lib.call("seg3d_spconv_fwd_act", _ptr(x), _ptr(nbr), m_out, x.shape[0], _ptr(packed.data), packed.flags, _ptr(bias),
              _ptr(addend), int(bool(relu)), cin, cout, _ptr(y), _ptr(order), _stream())
    return y


class _SparseConvFn(torch.autograd.Function):
    """y[r] = bias + sum_k x[nbr[k][r]] . W_k ; nbr_t is the same pair list keyed by input row."""

    @staticmethod
    def forward(ctx, x, weight, bias, nbr, nbr_t, t_flags, packed, order, order_t, plan=None, plan_t=None):
        x = _f32c(x)
        cout, cin = weight.shape[0], weight.shape[-1]
        if packed is None:
            packed = pack_weight(weight, PACK_FWD)
        y = _conv_apply(x, nbr, packed, None if bias is None else _f32c(bias), cin, cout, order, plan)
        ctx.save_for_backward(_saved_rows(x) if (ctx.needs_input_grad[1] and cin % 16 == 0 and cout % 16 == 0) else x, weight)
        ctx.nbr, ctx.nbr_t, ctx.t_flags, ctx.has_bias, ctx.order_t = nbr, nbr_t, t_flags, bias is not None, order_t
        ctx.plan_t = plan_t
        ctx.bias_param = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32c(dy)
        cout, cin = weight.shape[0], weight.shape[-1]
        dx = dw = db = None
        fk = _WgradFork(dy.device)
        if ctx.needs_input_grad[1]:  # weight gradient first, on the side stream; the input gradient co-runs with it
            dw = torch.empty_like(weight, dtype=torch.float32)
            ws_bytes = _lib.query("seg3d_spconv_wgrad_workspace_bytes", dy.shape[0], cin, cout)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dy.device)
            if _precision_flag() & PACK_SPLIT_BF16:
                # partial blocks now; their fixed-order sum joins the pass's other parameter-gradient sums (one launch)
                chunks = ctypes.c_int32(0)
                _lib.call("seg3d_spconv_wgrad_partials_xbf16" if x.dtype == torch.bfloat16 else "seg3d_spconv_wgrad_partials",
                          _ptr(x), _ptr(dy), _ptr(ctx.nbr), dy.shape[0], x.shape[0], cin, cout,
                          _ptr(ws), ws_bytes, ctypes.byref(chunks), fk.fork(ws, dy, x, ctx.nbr))
                n = 27 * cin * cout
                fk.add_reduce(ws, chunks.value, n, n, dw.data_ptr(), 0, ws)
            else:
                x = _f32c(x)
                _lib.call("seg3d_spconv_wgrad", _ptr(x), _ptr(dy), _ptr(ctx.nbr), dy.shape[0], x.shape[0], cin, cout,
                          _precision_flag(), _ptr(dw), _ptr(ws), ws_bytes, fk.fork(ws, dy, x, ctx.nbr))
        if ctx.needs_input_grad[0]:
            wt = pack_weight(weight, ctx.t_flags)
            dx = _conv_apply(dy, ctx.nbr_t, wt, None, cout, cin, ctx.order_t, ctx.plan_t)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if fk.on:  # a parameter gradient like dw: column sums of dy on the side stream
                fk.fork(dy)
                with torch.cuda.stream(fk.side):
                    db = dy.sum(0)
            else:
                db = dy.sum(0)
        fk.join((weight, dw), (ctx.bias_param, db))
        return dx, dw, db, None, None, None, None, None, None, None, None


def sparse_conv(x, weight, bias, nbr, nbr_t, t_flags, packed=None, order=None, order_t=None, plan=None, plan_t=None):
    """order / order_t: optional processing orders (int32 permutations) of the rows of nbr / nbr_t -- see
    SiteLevel.parity_order; plan / plan_t: optional tile plans (ConvPlan) of nbr / nbr_t -- see SiteLevel.subm_plan.
    Orders change scheduling only, never results.  Plans likewise for cout (forward) / cin (input gradient) that are
    multiples of 96 or 128; at 32 / 48 channels the tiled kernel splits a tile's 27 offsets over its four waves and sums
    the partial tiles in a fixed order of its own -- deterministic, but SEG3D_CONV_TILED=0 and =1 then differ by fp32
    round-off in those layers (include/seg3d_hip.h, seg3d_spconv_fwd_tiled)."""
    _need_gpu(x, weight, nbr)
    return _SparseConvFn.apply(x, weight, bias, nbr, nbr_t, t_flags, packed, order, order_t, plan, plan_t)


# ------------------------------------------------------------------------------------------ a6/a22 dense layers
def _linear_pack(weight, transpose):
    """MFMA fragment stream of a Linear weight (or of a row slice of one, e.g. in_proj_weight[:2C]); parameters go
    through the pack registry (validated by the tensor version, refreshed in one batched launch)."""
    if _registry_ok(weight) and (weight if weight._base is None else weight._base).is_leaf:
        return _cached_pack(weight, 1, int(transpose), 0)
    w = _f32c(weight)
    cout, cin = w.shape
    out = torch.empty((_lib.query("seg3d_linear_packed_bytes", cin, cout, int(transpose)),), dtype=torch.uint8,
                      device=w.device)
    _lib.call("seg3d_linear_pack_weight", _ptr(w), cin, cout, int(transpose), _ptr(out), _stream())
    return out


def _linear_pack_f32(weight, transpose):
    """Exact-fp32 MFMA fragment stream of a Linear weight, cached on the parameter while its version is unchanged."""
    cache = weight.__dict__.setdefault("_seg3d_f32_packs", {})
    hit = cache.get(int(transpose))
    if hit is not None and hit[0] == (_stamp(weight), weight.data_ptr()):
        return hit[1]
    w = _f32c(weight)
    cout, cin = w.shape
    out = torch.empty((_lib.query("seg3d_linear_packed_bytes_f32", cin, cout),), dtype=torch.uint8, device=w.device)
    _lib.call("seg3d_linear_pack_weight_f32", _ptr(w), cin, cout, int(transpose), _ptr(out), _stream())
    cache[int(transpose)] = ((_stamp(weight), weight.data_ptr()), out)
    return out


def _linear_apply_f32(x, packed, bias, cin, cout):
    y = torch.empty((x.shape[0], cout), dtype=torch.float32, device=x.device)
    _lib.call("seg3d_linear_fwd_f32", _ptr(x), x.shape[0], _ptr(packed), _ptr(bias), cin, cout, _ptr(y), _stream())
    return y


# Arithmetic of the per-point MLPs' forward ("exact" Linear layers): "x6" = three-way bf16 split, six bf16 MFMAs per product,
# fp32-grade results at 6/16 of the fp32 MFMA's time (csrc/linear_x6.hip; shapes with cin % 32 == 0 and cout % 64 == 0);
# "fp32" = v_mfma_f32_16x16x4_f32 everywhere (rounds 1 - 4; also what SEG3D_CONV_PRECISION=fp32 selects).
POINT_MLP = os.environ.get("SEG3D_POINT_MLP", "x6")


def _x6_fits(cin, cout):
    return POINT_MLP == "x6" and CONV_PRECISION == "bf16x3" and cin % 32 == 0 and cout % 64 == 0


def _linear_pack_x6(weight):
    """Three-plane bf16 fragment stream of a Linear weight, cached on the parameter while its version is unchanged."""
    cache = weight.__dict__.setdefault("_seg3d_x6_packs", {})
    hit = cache.get(0)
    if hit is not None and hit[0] == (_stamp(weight), weight.data_ptr()):
        return hit[1]
    w = _f32c(weight)
    cout, cin = w.shape
    out = torch.empty((_lib.query("seg3d_linear_packed_bytes_x6", cin, cout),), dtype=torch.uint8, device=w.device)
    _lib.call("seg3d_linear_pack_weight_x6", _ptr(w), cin, cout, 0, _ptr(out), _stream())
    cache[0] = ((_stamp(weight), weight.data_ptr()), out)
    return out


def _linear_apply_x6(x, packed, bias, cin, cout, scale=None, shift=None, relu=False):
    y = torch.empty((x.shape[0], cout), dtype=torch.float32, device=x.device)
    _lib.call("seg3d_linear_fwd_x6", _ptr(x), x.shape[0], _ptr(packed), _ptr(bias), _ptr(scale), _ptr(shift), int(relu),
              cin, cout, _ptr(y), _stream())
    return y


LINEAR_LN_FUSED = os.environ.get("SEG3D_LINEAR_LN", "1") != "0"  # 0 = inference encoder layers run Linear and LayerNorm apart


def _linear_layernorm(x, packed, bias, res, gamma, beta, eps, cin, cout):
    """res + LayerNorm(x W^T + b) in one launch (inference; cout <= 192)."""
    y = torch.empty((x.shape[0], cout), dtype=torch.float32, device=x.device)
    _lib.call("seg3d_linear_layernorm_fwd", _ptr(x), x.shape[0], _ptr(packed), _ptr(bias), _ptr(res), _ptr(gamma), _ptr(beta),
              float(eps), cin, cout, _ptr(y), _stream())
    return y


def _linear_apply(x, packed, bias, cin, cout, addend=None):
    y = torch.empty((x.shape[0], cout), dtype=torch.float32, device=x.device)
    _lib.call("seg3d_linear_fwd", _ptr(x), x.shape[0], _ptr(packed), _ptr(bias), _ptr(addend), cin, cout, _ptr(y), _stream())
    return y


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on [rows, C] activations: forward and input gradient as the single-offset case of the
    split-bf16 gather-GEMM kernel, weight gradient through the tall-skinny split-bf16 kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, exact):
        x = _f32c(x)
        cout, cin = weight.shape
        ctx.save_for_backward(_saved_rows(x) if ctx.needs_input_grad[1] else x, weight)
        ctx.has_bias, ctx.exact = bias is not None, exact
        ctx.bias_param = bias  # (a reference for the deferred-join bookkeeping of the weight-gradient stream)
        if exact:  # fp32-grade forward (six-product split, or the fp32 MFMA; rocBLAS for odd shapes); the gradients use the split kernels
            if _x6_fits(cin, cout):
                return _linear_apply_x6(x, _linear_pack_x6(weight), None if bias is None else _f32c(bias), cin, cout)
            if cin % 16 == 0 and cout % 16 == 0:
                return _linear_apply_f32(x, _linear_pack_f32(weight, 0), None if bias is None else _f32c(bias), cin, cout)
            return torch.nn.functional.linear(x, weight, bias)
        return _linear_apply(x, _linear_pack(weight, 0), None if bias is None else _f32c(bias), cin, cout)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32c(dy)
        cout, cin = weight.shape
        dx = dw = db = None
        fk = _WgradFork(dy.device)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:  # on the side stream, co-running with the input gradient below
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            if want_db:  # column sums of dy ride along with the weight-gradient pass
                db = torch.empty((cout,), dtype=torch.float32, device=dy.device)
            _linear_wgrad_into(fk, x, dy, cin, cout, dw.data_ptr(), db.data_ptr() if want_db else 0)
        elif want_db:
            db = dy.sum(0)
        if ctx.needs_input_grad[0]:
            # ``exact`` protects the forward logits; the input gradient of an exact layer takes the split-bf16 kernel
            # like every other gradient of the step (16/3 the fp32-MFMA rate)
            if cout % 8 == 0 and cin % 16 == 0:
                dx = _linear_apply(dy, _linear_pack(weight, 1), None, cout, cin)
            elif ctx.exact and cout % 16 == 0 and cin % 16 == 0:
                dx = _linear_apply_f32(dy, _linear_pack_f32(weight, 1), None, cout, cin)
            else:
                dx = dy @ weight
        fk.join((weight, dw), (ctx.bias_param, db))
        return dx, dw, db, None


class _LinearOddShapeFn(torch.autograd.Function):
    """Linear layers whose shape does not fit the MFMA tiles (6 -> 64 input layer, -> 22 classifiers): forward and
    input gradient stay on rocBLAS, the weight gradient -- a 64 x 6 or 22 x 64 output reduced over 1e5 rows, for which
    rocBLAS takes 340-470 us -- runs on the split-bf16 tall-skinny kernel over zero-padded (multiple-of-4) columns."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _f32c(x)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32c(dy)
        cout, cin = weight.shape
        dx = dw = db = None
        fk = _WgradFork(dy.device)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            pi, po = (-cin) % 4, (-cout) % 4
            xp = torch.nn.functional.pad(x, (0, pi)) if pi else x
            dyp = torch.nn.functional.pad(dy, (0, po)) if po else dy
            dwp = torch.empty((cout + po, cin + pi), dtype=torch.float32, device=dy.device)
            dbp = torch.empty((cout + po,), dtype=torch.float32, device=dy.device) if want_db else None
            _linear_wgrad_into(fk, xp, dyp, cin + pi, cout + po, dwp.data_ptr(), dbp.data_ptr() if want_db else 0)
            dw = dwp[:cout, :cin]
            if want_db:
                db = dbp[:cout]
        elif want_db:
            db = dy.sum(0)
        if ctx.needs_input_grad[0]:
            dx = dy @ weight
        # (padded shapes return VIEWS of the padded gradient: autograd copies those, so this function joins at once unless
        # nothing was padded)
        if (cin % 4) or (cout % 4):
            fk.join()
        else:
            fk.join((weight, dw), (ctx.bias_param, db))
        return dx, dw, db


def linear(x, weight, bias=None, exact=False):
    """F.linear for [rows, C] activations.  Layers whose shape fits the MFMA tiles (cin % 8 == 0,
    cout % 16 == 0) run in libseg3d_hip.so in split-bf16 arithmetic; the rest (6 -> 64 input layer, -> 22
    classifiers) stay on rocBLAS.  ``exact=True`` keeps the FORWARD at fp32 grade -- the six-product three-way split of
    csrc/linear_x6.hip (``SEG3D_POINT_MLP=x6``, default, cin % 32 == 0 and cout % 64 == 0), else the v_mfma_f32_16x16x4_f32
    kernel, rocBLAS for shapes neither takes -- and takes the gradients from the split kernels: used by the per-point MLPs,
    whose ~2^-16 relative forward error would land directly on the logits (measured 1e-3 absolute) instead of being
    washed out by the LayerNorms of the voxel path."""
    fits = (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.shape[0] % 16 == 0
            and weight.shape[1] % 8 == 0 and CONV_PRECISION == "bf16x3")
    if exact and fits and weight.shape[1] % 16 == 0:
        return _LinearFn.apply(x, weight, bias, True)  # exact-fp32 MFMA forward / input gradient, split wgrad
    if not fits and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and CONV_PRECISION == "bf16x3" \
            and torch.is_grad_enabled() and weight.requires_grad and x.shape[0] >= 4096:
        return _LinearOddShapeFn.apply(x, weight, bias)
    if not fits or (exact and not (torch.is_grad_enabled() and weight.requires_grad)):
        return torch.nn.functional.linear(x, weight, bias)
    return _LinearFn.apply(x, weight, bias, exact)


# ------------------------------------------------------------------------------------------ a12/a22 norms
class _LayerNormResidualFn(torch.autograd.Function):
    """y = res + rowscale * LayerNorm(x) (res, rowscale optional; rowscale [rows] carries no gradient)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, rowscale):
        x = _f32c(x)
        m, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((m,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((m,), dtype=torch.float32, device=x.device)
        r = None if res is None else _f32c(res)
        rs = None if rowscale is None else _f32c(rowscale.reshape(-1))
        _lib.call("seg3d_layernorm_fwd", _ptr(x), _ptr(r), _ptr(gamma), _ptr(beta), _ptr(rs), float(eps), m, c, _ptr(y),
                  _ptr(mean), _ptr(rstd), _stream())
        ctx.save_for_backward(x, mean, rstd, gamma, rs)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, rs = ctx.saved_tensors
        dy = _f32c(dy)
        m, c = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty((c,), dtype=torch.float32, device=x.device)
        db = torch.empty((c,), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_layernorm_bwd_workspace_bytes", m, c)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
        fork = getattr(ctx, "fork", None)
        if fork is not None:  # an enclosing layer function joins for us: dgamma / dbeta's fixed-order sum is queued on its fork
            nb = ctypes.c_int32(0)
            _lib.call("seg3d_layernorm_bwd_partials", _ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(rs), m, c,
                      _ptr(dx), _ptr(ws), ws_bytes, ctypes.byref(nb), _stream())
            fork.add_reduce(ws, nb.value, 2 * c, c, dg.data_ptr(), db.data_ptr())
            return dx, (dy if ctx.has_res else None), dg, db, None, None
        _lib.call("seg3d_layernorm_bwd", _ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(rs), m, c, _ptr(dx),
                  _ptr(dg), _ptr(db), _ptr(ws), ws_bytes, _stream())
        return dx, (dy if ctx.has_res else None), dg, db, None, None


def layer_norm_residual(x, res, ln, rowscale=None):
    """res + rowscale * ln(x) for an nn.LayerNorm over the last dim of [rows, C] (C % 4 == 0, C <= 512);
    rowscale [rows] is the per-row DropPath factor (mask / keep_prob), None = 1."""
    c = x.shape[1]
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and c % 4 == 0 and c <= 512
            and ln.elementwise_affine and ln.bias is not None):
        y = ln(x)
        if rowscale is not None:
            y = y * rowscale.reshape(-1, 1)
        return y if res is None else res + y
    return _LayerNormResidualFn.apply(x, res, ln.weight, ln.bias, ln.eps, rowscale)


class _BatchNormActFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over rows (+ residual) (+ ReLU); statistics are computed in forward."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, mean, rstd, scale, shift, relu):
        x = _f32c(x)
        m, c = x.shape
        y = torch.empty_like(x)
        r = None if res is None else _f32c(res)
        _lib.call("seg3d_affine_act", _ptr(x), _ptr(r), _ptr(scale), _ptr(shift), int(relu), m, c, _ptr(y), _stream())
        ctx.save_for_backward(x, y, mean, rstd, gamma)
        ctx.relu, ctx.has_res = relu, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        m, c = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", m, c)
        ws = _workspace(ws_bytes, x.device)
        _lib.call("seg3d_batchnorm_bwd", _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), int(ctx.relu),
                  m, c, _ptr(dx), _ptr(dres), _ptr(sums), _ptr(ws), ws_bytes, _stream())
        return dx, dres, sums[1], sums[0], None, None, None, None, None


# ---- torch.nn.SyncBatchNorm (tools/train.py:246-247, --sync_bn): statistics over the rows of ALL ranks
def _sync_bn_group(bn):
    """Process group to synchronise over, or None: only a SyncBatchNorm in training mode inside an initialised
    job with more than one rank synchronises (torch's own rule, nn/modules/batchnorm.py SyncBatchNorm.forward)."""
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not bn.training:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


def combine_moments(mean, m2, count):
    """Chan's parallel-variance merge.  mean, m2: [ranks, c] per-rank mean and sum of squared deviations from it;
    count: [ranks, 1] rows per rank.  Returns (mean [c], m2 [c], total [1]) of the union, on the inputs' device."""
    total = count.sum(dim=0)
    g_mean = (mean * count).sum(dim=0) / total
    g_m2 = (m2 + count * (mean - g_mean) ** 2).sum(dim=0)
    return g_mean, g_m2, total


def _all_gather_rows(t, group):
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":  # the CPU-side rehearsal backend has no all_gather_into_tensor on devices
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return torch.stack(parts)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


class _SyncBatchNormActFn(torch.autograd.Function):
    """act(x * scale + shift (+ res)) with statistics of all ranks; backward all-reduces {sum g, sum g * xhat} between
    the reduce and apply halves (seg3d_batchnorm_bwd_reduce / _apply).  dgamma / dbeta stay rank-local, as in torch's
    SyncBatchNorm: DDP averages them with every other parameter gradient."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, mean, rstd, scale, shift, inv_count, relu, group):
        m, c = x.shape
        y = torch.empty_like(x)
        r = None if res is None else _f32c(res)
        _lib.call("seg3d_affine_act", _ptr(x), _ptr(r), _ptr(scale), _ptr(shift), int(relu), m, c, _ptr(y), _stream())
        ctx.save_for_backward(x, y, mean, rstd, gamma, inv_count)
        ctx.relu, ctx.has_res, ctx.group = relu, res is not None, group
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, y, mean, rstd, gamma, inv_count = ctx.saved_tensors
        dy = _f32c(dy)
        m, c = x.shape
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", m, c)
        ws = _workspace(ws_bytes, x.device)
        if m > 0:
            _lib.call("seg3d_batchnorm_bwd_reduce", _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), int(ctx.relu), m, c,
                      _ptr(sums), _ptr(ws), ws_bytes, _stream())
        else:
            sums.zero_()
        total = sums.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=ctx.group)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        if m > 0:
            _lib.call("seg3d_batchnorm_bwd_apply", _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma),
                      _ptr(total), _ptr(inv_count), int(ctx.relu), m, c, _ptr(dx), _ptr(dres), _stream())
        return dx, dres, sums[1], sums[0], None, None, None, None, None, None, None


def _sync_batch_norm_act(x, bn, relu, res, group):
    xc = _f32c(x)
    m, c = xc.shape
    with torch.no_grad():
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", m, c)
        ws = _workspace(ws_bytes, x.device)
        if m > 0:
            _lib.call("seg3d_colstats", _ptr(xc), m, c, _ptr(sums), _ptr(ws), ws_bytes, _stream())
            d = sums[0] / m  # shifted sums: mean of (x - x[0]) and the squared deviations from the local mean
            local = torch.cat([xc[0] + d, sums[1] - sums[0] * d, torch.full((1,), float(m), device=x.device)])
        else:  # a rank without rows still takes part in the exchange, with weight 0
            local = torch.zeros((2 * c + 1,), dtype=torch.float32, device=x.device)
        g = _all_gather_rows(local, group)
        mean, m2, total = combine_moments(g[:, :c], g[:, c:2 * c], g[:, 2 * c:])
        var = (m2 / total).clamp_(min=0.0)
        rstd = torch.rsqrt(var + bn.eps)
        scale = (bn.weight * rstd).contiguous()
        shift = (bn.bias - mean * scale).contiguous()
        if bn.track_running_stats:
            bn._seg3d_stats_epoch = getattr(bn, "_seg3d_stats_epoch", 0) + 1
            _bump_batches_tracked(bn)
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(m2 / (total - 1).clamp_(min=1.0), alpha=mom)
        inv_count = (1.0 / total).to(torch.float32).contiguous()
    return _SyncBatchNormActFn.apply(xc, res, bn.weight, bn.bias, mean.contiguous(), rstd.contiguous(), scale, shift, inv_count,
                                     bool(relu), group)


class _SpanMeanFn(torch.autograd.Function):
    """Column means of contiguous row spans: out[b] = mean(x[offsets[b-1] : offsets[b]], dim=0)."""

    @staticmethod
    def forward(ctx, x, offsets):
        x = _f32c(x)
        c = x.shape[1]
        out = torch.zeros((len(offsets), c), dtype=torch.float32, device=x.device)
        lo = 0
        for b, hi in enumerate(offsets):
            rows = hi - lo
            if rows > 0:
                sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
                ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", rows, c)
                ws = _workspace(ws_bytes, x.device)
                _lib.call("seg3d_colstats", ctypes.c_void_p(x.data_ptr() + 4 * lo * c), rows, c, _ptr(sums), _ptr(ws), ws_bytes,
                          _stream())
                out[b] = x[lo] + sums[0] / rows  # shifted sums: mean = x[first row] + mean of (x - x[first row])
            lo = hi
        ctx.offsets, ctx.rows = tuple(offsets), x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        pieces, lo = [], 0
        for b, hi in enumerate(ctx.offsets):
            if hi > lo:
                pieces.append((g[b] / (hi - lo)).expand(hi - lo, g.shape[1]))
            lo = hi
        if lo < ctx.rows:
            pieces.append(g.new_zeros((ctx.rows - lo, g.shape[1])))
        return (torch.cat(pieces) if len(pieces) != 1 else pieces[0].contiguous()), None


def span_mean(x, row_offsets):
    """[B, C] column means of the contiguous row spans given by cumulative python-int offsets (FlattenSELayer's scatter-mean
    over the batch index, se_layer.py:24-25, on collated rows).  Per-block partial sums in a fixed order (seg3d_colstats):
    deterministic, and -- unlike torch's two-stage reduce over ~1e5 rows, which did not replay correctly from a captured
    hipGraph once its input had new values (tools/probes/graph_dbg.py) -- a plain launch sequence."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[1] <= 1024):
        lo, out = 0, []
        for hi in row_offsets:
            out.append(x[lo:hi].mean(dim=0) if hi > lo else x.new_zeros(x.shape[1]))
            lo = hi
        return torch.stack(out)
    return _SpanMeanFn.apply(x, [int(o) for o in row_offsets])


def narrow_batch_norm(x, bn):
    """Training-mode BatchNorm1d over [rows, C] with C not a multiple of 4 (the raw 6 / 8 point channels): the rows and
    the affine parameters are zero-padded to the next multiple of 4 and run through the same statistics / affine /
    backward kernels as the wide layers; the running buffers are updated from the kernel's shifted sums."""
    m, c = x.shape
    pad = (-c) % 4
    xp = torch.nn.functional.pad(_f32c(x), (0, pad))
    w = torch.nn.functional.pad(bn.weight, (0, pad), value=1.0)
    b = torch.nn.functional.pad(bn.bias, (0, pad))
    cp = c + pad
    with torch.no_grad():
        stats = torch.empty((6, cp), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", m, cp)
        ws = _workspace(ws_bytes, x.device)
        _lib.call("seg3d_batchnorm_stats", _ptr(xp), m, cp, float(bn.eps), _ptr(w), _ptr(b), 0.0, None, None, _ptr(stats),
                  _ptr(ws), ws_bytes, _stream())
        _bump_batches_tracked(bn)
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        d = stats[0, :c] / m  # mean of (x - x[0]); biased variance from the shifted sums, as the kernel forms it
        var = (stats[1, :c] / m - d * d).clamp_(min=0.0)
        bn.running_mean.mul_(1 - mom).add_(stats[2, :c], alpha=mom)
        bn.running_var.mul_(1 - mom).add_(var, alpha=mom * (m / max(m - 1, 1)))
    y = _BatchNormActFn.apply(xp, None, w, b, stats[2], stats[3], stats[4], stats[5], False)
    return y[:, :c]


# BatchNorm's num_batches_tracked += 1 is a kernel launch per layer and forward (28 on the training step's critical path).
# Inside a segmentor's forward the counters are collected and bumped by ONE foreach launch at the end; with momentum=None
# (cumulative average: the value is needed at once) and outside that scope the increment happens on the spot.
_BN_COUNTERS = None


class deferred_bn_counters:
    def __enter__(self):
        global _BN_COUNTERS
        self.outer = _BN_COUNTERS
        if _BN_COUNTERS is None:
            _BN_COUNTERS = []
        return self

    def __exit__(self, *exc):
        global _BN_COUNTERS
        if self.outer is None:
            pending, _BN_COUNTERS = _BN_COUNTERS, None
            if pending:
                torch._foreach_add_(pending, 1)
        return False


def _bump_batches_tracked(bn):
    if _BN_COUNTERS is not None and bn.momentum is not None:
        _BN_COUNTERS.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def bn_eval_affine(bn):
    """(scale, shift, key) with bn(x) = x * scale + shift in eval mode.  Cached on the module, keyed on the tensors'
    versions so that loading a checkpoint invalidates it; training steps reset it (batch_norm_act) because the kernel
    updates the running buffers through raw pointers."""
    key = (_stamp(bn.weight), bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_var.data_ptr(), bn.eps, getattr(bn, "_seg3d_stats_epoch", 0))
    cached = getattr(bn, "_seg3d_eval_affine", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
        bn._seg3d_eval_affine = cached = (key, scale.contiguous(), shift.contiguous())
    return cached[1], cached[2], key


def batch_norm_act(x, bn, relu=True, res=None):
    """act(bn(x) (+ res)) for an nn.BatchNorm1d over [rows, C] features (C % 4 == 0), relu optional.
    Training: batch statistics (biased variance for the normalisation, unbiased into running_var, as torch);
    eval: running statistics folded into one affine pass."""
    c = x.shape[1]
    fits = x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and c % 4 == 0 and c <= 1024 and bn.affine
    use_batch = bn.training or not bn.track_running_stats
    if fits and use_batch:
        group = _sync_bn_group(bn)
        if group is not None:  # --sync_bn: statistics over the rows of all ranks (every rank must take this branch, so
            return _sync_batch_norm_act(x, bn, relu, res, group)  # it comes before the small-batch fallback below)
    if not fits or (not use_batch and torch.is_grad_enabled() and x.requires_grad) or x.shape[0] < 2:
        y = bn(x)
        if res is not None:
            y = y + res
        return torch.relu(y) if relu else y
    if use_batch:
        xc = _f32c(x)
        m = xc.shape[0]
        with torch.no_grad():  # statistics, folded affine and running buffers in one call (two small launches)
            stats = torch.empty((6, c), dtype=torch.float32, device=x.device)
            track = bn.training and bn.track_running_stats
            mom = 0.0
            if track:
                # the kernel updates the running buffers through raw pointers: tell the eval-affine cache
                bn._seg3d_stats_epoch = getattr(bn, "_seg3d_stats_epoch", 0) + 1
                _bump_batches_tracked(bn)
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            ws_bytes = _lib.query("seg3d_batchnorm_workspace_bytes", m, c)
            ws = _workspace(ws_bytes, x.device)
            _lib.call("seg3d_batchnorm_stats", _ptr(xc), m, c, float(bn.eps), _ptr(bn.weight), _ptr(bn.bias), float(mom),
                      _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None, _ptr(stats),
                      _ptr(ws), ws_bytes, _stream())
        return _BatchNormActFn.apply(xc, res, bn.weight, bn.bias, stats[2], stats[3], stats[4], stats[5], bool(relu))
    with torch.no_grad():
        # eval: the folded affine is a constant of the module; keyed on the tensors' versions so that loading a checkpoint
        # or resuming training invalidates it (5 tiny launches per BatchNorm per forward otherwise: ~1 ms on 27 layers)
        scale, shift, _ = bn_eval_affine(bn)
        xc = _f32c(x)
        y = torch.empty_like(xc)
        r = None if res is None else _f32c(res)
        _lib.call("seg3d_affine_act", _ptr(xc), _ptr(r), _ptr(scale), _ptr(shift), int(relu), xc.shape[0], c, _ptr(y),
                  _stream())
    return y


def linear_bn_act_eval(x, lin, bn, relu):
    """act(bn(lin(x))) in ONE launch for the eval forward of the per-point MLPs (Linear -> BatchNorm1d -> ReLU,
    seg3d/models/segmentors/segformer.py:21-32): the six-product Linear kernel applies the BatchNorm's folded affine and the
    ReLU in its epilogue, rounding as the separate seg3d_affine_act pass does (bit-identical to linear + batch_norm_act).
    Returns None when the shapes / modes are not this path's (the caller runs the two passes)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
        return None
    cout, cin = lin.weight.shape
    if not _x6_fits(cin, cout) or bn.training or not bn.track_running_stats or not bn.affine or bn.num_features != cout:
        return None
    if torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad):
        return None
    with torch.no_grad():
        scale, shift, _ = bn_eval_affine(bn)
        return _linear_apply_x6(_f32c(x), _linear_pack_x6(lin.weight), None if lin.bias is None else _f32c(lin.bias), cin, cout,
                                scale=scale, shift=shift, relu=relu)


# ------------------------------------------------------------------------------------------ a13-a18 windows
class WindowIndex:
    """Outputs of seg3d_window_partition for one (stage, shift)."""
    __slots__ = ("win_id", "in_win", "rank", "level", "slot", "tok", "win_start", "win_count", "win_tile0",
                 "tile_item", "qg_item", "counts", "n_windows", "n_dropped", "n_tiles", "n_qgroups", "m")


def window_partition(coords, batch_size, window_shape, nwin_xyz, shift_xyz, levels, want_debug=False):
    """levels: list of (lo, hi, max_tokens).  No host sync: n_windows is bounded by min(m, canvas) and the
    exact count stays on the device in ``counts`` until ``finish_window_index`` reads it."""
    _need_gpu(coords)
    coords = _i32c(coords)
    m, dev = coords.shape[0], coords.device
    canvas = int(batch_size) * int(nwin_xyz[0]) * int(nwin_xyz[1]) * int(nwin_xyz[2])
    cap = max(min(m, canvas), 1)
    wi = WindowIndex()
    wi.m = m
    i32 = dict(dtype=torch.int32, device=dev)
    wi.in_win = torch.empty((m, 3), **i32)
    wi.win_id = torch.empty((m,), **i32) if want_debug else None
    wi.rank = torch.empty((m,), **i32) if want_debug else None
    wi.level = torch.empty((m,), **i32) if want_debug else None
    wi.slot = torch.empty((m,), **i32) if want_debug else None
    wi.tok = torch.empty((max(m, 1),), **i32)
    wi.win_start = torch.empty((cap,), **i32)
    wi.win_count = torch.empty((cap,), **i32)
    wi.win_tile0 = torch.empty((cap,), **i32)
    wi.tile_item = torch.empty((m // 32 + cap + 1, 4), **i32)  # {window, tile, first token slot, tokens}
    wi.qg_item = torch.empty((m // 16 + cap + 1, 4), **i32)
    wi.counts = torch.zeros((4,), **i32)
    nl = len(levels)
    arr = ctypes.c_int32 * nl
    lo, hi, capt = arr(*[int(l[0]) for l in levels]), arr(*[int(l[1]) for l in levels]), arr(*[int(l[2]) for l in levels])
    ws = _workspace(_lib.query("seg3d_window_partition_workspace_bytes", m, int(batch_size), _i3(nwin_xyz)), dev)
    _lib.call("seg3d_window_partition", _ptr(coords), m, int(batch_size), _i3(window_shape), _i3(nwin_xyz),
              _i3(shift_xyz), nl, lo, hi, capt, _ptr(wi.win_id), _ptr(wi.in_win), _ptr(wi.rank), _ptr(wi.level),
              _ptr(wi.slot), _ptr(wi.tok), _ptr(wi.win_start), _ptr(wi.win_count), _ptr(wi.win_tile0),
              _ptr(wi.tile_item), _ptr(wi.qg_item), _ptr(wi.counts), _ptr(ws), ws.numel(), _stream())
    wi.n_windows = wi.n_dropped = wi.n_tiles = wi.n_qgroups = None  # filled by the caller's one sync
    return wi


def pos_embed(in_win, window_shape, inv_freq, c):
    m = in_win.shape[0]
    pos = torch.empty((m, c), dtype=torch.float32, device=in_win.device)
    _lib.call("seg3d_pos_embed", _ptr(in_win), m, _i3(window_shape), _ptr(inv_freq), int(c), _ptr(pos), _stream())
    return pos


# ------------------------------------------------------------------------------------------ a19-a21 attention
class _WindowAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, tau, tau_min, heads, wi):
        m, c = v.shape
        dh = c // heads
        dev = v.device
        out = torch.empty((m, c), dtype=torch.float32, device=dev)
        lse = torch.empty((m, heads), dtype=torch.float32, device=dev)
        tau_f = tau.reshape(-1)
        _lib.call("seg3d_window_attn_fwd", _ptr(q), _ptr(k), _ptr(v), q.stride(0), k.stride(0), v.stride(0),
                  _ptr(wi.tok), _ptr(wi.win_start), _ptr(wi.win_count), _ptr(wi.win_tile0), _ptr(wi.tile_item),
                  int(wi.n_tiles), _ptr(wi.qg_item), int(wi.n_qgroups), m, int(wi.n_windows), heads, dh,
                  _ptr(tau_f), float(tau_min), 0.0, 0, _ptr(out), _ptr(lse), None, 0, _stream())  # (the forward needs no workspace)
        ctx.save_for_backward(q, k, v, tau, out, lse)
        ctx.wi, ctx.heads, ctx.tau_min = wi, heads, tau_min
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, tau, out, lse = ctx.saved_tensors
        wi, heads = ctx.wi, ctx.heads
        m, c = v.shape
        dev = v.device
        dout = _f32c(dout)
        dq, dk, dv = (torch.empty((m, c), dtype=torch.float32, device=dev) for _ in range(3))
        dtau = torch.empty((1,), dtype=torch.float32, device=dev)  # written (not accumulated) by the kernels
        ws = _workspace(_lib.query("seg3d_window_attn_workspace_bytes", m, int(wi.n_tiles), heads, c // heads), dev)
        _lib.call("seg3d_window_attn_bwd", _ptr(q), _ptr(k), _ptr(v), q.stride(0), k.stride(0), v.stride(0),
                  _ptr(out), _ptr(dout), _ptr(lse), _ptr(wi.tok), _ptr(wi.win_start), _ptr(wi.win_count),
                  _ptr(wi.win_tile0), _ptr(wi.tile_item), int(wi.n_tiles), _ptr(wi.qg_item), int(wi.n_qgroups), m,
                  int(wi.n_windows), heads, c // heads, _ptr(tau.reshape(-1)), float(ctx.tau_min), 0.0, 0, _ptr(dq), _ptr(dk),
                  _ptr(dv), c, c, c, _ptr(dtau), _ptr(ws), ws.numel(), _stream())
        return dq, dk, dv, dtau.reshape(tau.shape), None, None, None


class _WindowAttnPackedFn(torch.autograd.Function):
    """Same kernels on the packed in-projection output qk [m, 2C] (q | k): the gradient comes back as one [m, 2C]
    tensor written in place by the backward kernels (row stride 2C), so autograd never materialises the two
    zero-filled slice gradients of q = qk[:, :C], k = qk[:, C:] and their sum."""

    @staticmethod
    def forward(ctx, qk, v, tau, tau_min, heads, wi, drop_p=0.0, drop_seed=0):
        m, c = v.shape
        dh = c // heads
        dev = v.device
        out = torch.empty((m, c), dtype=torch.float32, device=dev)
        lse = torch.empty((m, heads), dtype=torch.float32, device=dev)
        ctx.drop_p, ctx.drop_seed = drop_p, drop_seed
        kp = ctypes.c_void_p(qk.data_ptr() + 4 * c)
        _lib.call("seg3d_window_attn_fwd", _ptr(qk), kp, _ptr(v), 2 * c, 2 * c, c,
                  _ptr(wi.tok), _ptr(wi.win_start), _ptr(wi.win_count), _ptr(wi.win_tile0), _ptr(wi.tile_item),
                  int(wi.n_tiles), _ptr(wi.qg_item), int(wi.n_qgroups), m, int(wi.n_windows), heads, dh,
                  _ptr(tau.reshape(-1)), float(tau_min), float(drop_p), int(drop_seed), _ptr(out), _ptr(lse), None, 0,
                  _stream())  # (no workspace: seg3d_window_attn_workspace_bytes sizes the backward's)
        ctx.save_for_backward(qk, v, tau, out, lse)
        ctx.wi, ctx.heads, ctx.tau_min = wi, heads, tau_min
        return out

    @staticmethod
    def backward(ctx, dout):
        qk, v, tau, out, lse = ctx.saved_tensors
        wi, heads = ctx.wi, ctx.heads
        m, c = v.shape
        dev = v.device
        dout = _f32c(dout)
        dqk = torch.empty((m, 2 * c), dtype=torch.float32, device=dev)
        dv = torch.empty((m, c), dtype=torch.float32, device=dev)
        dtau = torch.empty((1,), dtype=torch.float32, device=dev)  # written (not accumulated) by the kernels
        ws = _workspace(_lib.query("seg3d_window_attn_workspace_bytes", m, int(wi.n_tiles), heads, c // heads), dev)
        kp = ctypes.c_void_p(qk.data_ptr() + 4 * c)
        dkp = ctypes.c_void_p(dqk.data_ptr() + 4 * c)
        _lib.call("seg3d_window_attn_bwd", _ptr(qk), kp, _ptr(v), 2 * c, 2 * c, c,
                  _ptr(out), _ptr(dout), _ptr(lse), _ptr(wi.tok), _ptr(wi.win_start), _ptr(wi.win_count),
                  _ptr(wi.win_tile0), _ptr(wi.tile_item), int(wi.n_tiles), _ptr(wi.qg_item), int(wi.n_qgroups), m,
                  int(wi.n_windows), heads, c // heads, _ptr(tau.reshape(-1)), float(ctx.tau_min), float(ctx.drop_p),
                  int(ctx.drop_seed), _ptr(dqk), dkp, _ptr(dv), 2 * c, 2 * c, c, _ptr(dtau), _ptr(ws), ws.numel(), _stream())
        return dqk, dv, dtau.reshape(tau.shape), None, None, None, None, None


def window_attention_packed(qk, v, tau, tau_min, heads, wi, drop_p=0.0, drop_seed=0):
    """qk: contiguous float32 [m, 2C] (q | k), v: contiguous float32 [m, C].  drop_p / drop_seed: attention-probability
    dropout of training mode (cosine_msa.py:172-174), mask = f(seed, window, head, query, key)."""
    _need_gpu(qk, v, tau)
    if qk.dtype != torch.float32 or v.dtype != torch.float32 or not qk.is_contiguous() or not v.is_contiguous() \
            or qk.shape[1] != 2 * v.shape[1]:
        raise _lib.Seg3dError("qk must be contiguous float32 [m, 2C] and v contiguous float32 [m, C]")
    if v.shape[1] % heads or not _lib.load().seg3d_window_attn_supported(int(heads), v.shape[1] // int(heads)):
        raise _lib.Seg3dError(
            f"window attention: {heads} heads on {v.shape[1]} channels is not a head geometry of this path (head widths "
            "6 / 12 with a head count that is a multiple of 4, 24 / 48 with up to 16 heads; the reference builds 8 heads "
            "on 48 / 96 / 192 / 384 channels, pointtransformer.py:143-155)")
    return _WindowAttnPackedFn.apply(qk, v, tau, tau_min, heads, wi, float(drop_p), int(drop_seed))


INPROJ_SUM = os.environ.get("SEG3D_INPROJ_SUM", "1") != "0"  # 0 = the inference in-projection materialises x + pos (A/B)


class _AttnInProjFn(torch.autograd.Function):
    """In-projection of the cosine attention (cosine_msa.py:58-63): qk = (x + pos) W_qk^T + b_qk, v = x W_v^T + b_v with
    the packed parameters in_proj_weight [3C, C] / in_proj_bias [3C].  One backward produces dx (both paths summed),
    dpos and the full dW / db (the two weight-gradient calls write the two row blocks in place): autograd sees no
    parameter slices, hence no zero-filled slice gradients and no extra adds."""

    @staticmethod
    def forward(ctx, x, pos, w_in, b_in):
        x = _f32c(x)
        c = x.shape[1]
        if (INPROJ_SUM and not any(ctx.needs_input_grad) and pos.shape == x.shape and pos.dtype == torch.float32
                and pos.is_contiguous()):
            # inference: x + pos is summed on the GEMM's A operand and never written (in training it is kept: the weight
            # gradient of the q | k rows multiplies with it)
            qk = torch.empty((x.shape[0], 2 * c), dtype=torch.float32, device=x.device)
            _lib.call("seg3d_linear_fwd_sum", _ptr(x), _ptr(pos), x.shape[0], _ptr(_linear_pack(w_in[: 2 * c], 0)),
                      _ptr(b_in[: 2 * c]), c, 2 * c, _ptr(qk), _stream())
            return qk, _linear_apply(x, _linear_pack(w_in[2 * c:], 0), b_in[2 * c:], c, c)
        xp = x + pos
        qk = _linear_apply(xp, _linear_pack(w_in[: 2 * c], 0), b_in[: 2 * c], c, 2 * c)
        v = _linear_apply(x, _linear_pack(w_in[2 * c:], 0), b_in[2 * c:], c, c)
        keep = ctx.needs_input_grad[2]
        ctx.save_for_backward(_saved_rows(x) if keep else x, _saved_rows(xp) if keep else xp, w_in)
        ctx.b_in = b_in
        return qk, v

    @staticmethod
    def backward(ctx, dqk, dv):
        x, xp, w_in = ctx.saved_tensors
        c = x.shape[1]
        m = x.shape[0]
        dqk, dv = _f32c(dqk), _f32c(dv)
        dx = dpos = dw = db = None
        fk = _WgradFork(x.device)
        if ctx.needs_input_grad[2]:  # both weight-gradient launches on the side stream, the input gradients co-run
            dw = torch.empty((3 * c, c), dtype=torch.float32, device=x.device)
            db = torch.empty((3 * c,), dtype=torch.float32, device=x.device)
            for src, dy, r0, rows in ((xp, dqk, 0, 2 * c), (x, dv, 2 * c, c)):
                _linear_wgrad_into(fk, src, dy, c, rows, dw.data_ptr() + 4 * r0 * c, db.data_ptr() + 4 * r0)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # an enclosing layer function may hand over the residual path's gradient: it rides in the same epilogue
            extra = None if ctx.needs_input_grad[1] else getattr(ctx, "dx_addend", None)
            d_xp = _linear_apply(dqk, _linear_pack(w_in[: 2 * c], 1), None, 2 * c, c, addend=extra)
            if ctx.needs_input_grad[1]:
                dpos = d_xp
            if ctx.needs_input_grad[0]:
                dx = _linear_apply(dv, _linear_pack(w_in[2 * c:], 1), None, c, c, addend=d_xp)  # both paths in one pass
        fk.join((w_in, dw), (ctx.b_in, db if ctx.needs_input_grad[3] else None))
        return dx, dpos, dw, (db if ctx.needs_input_grad[3] else None)


def attn_in_proj(x, pos, w_in, b_in):
    """(qk [m, 2C], v [m, C]) of the packed cosine-attention in-projection; falls back to three F.linear-style calls
    when the shapes do not fit the MFMA tiles."""
    c = x.shape[1]
    fits = (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and c % 16 == 0 and CONV_PRECISION == "bf16x3"
            and w_in.is_contiguous() and b_in is not None)
    if not fits:
        return linear(x + pos, w_in[: 2 * c], None if b_in is None else b_in[: 2 * c]), \
            linear(x, w_in[2 * c:], None if b_in is None else b_in[2 * c:])
    return _AttnInProjFn.apply(x, pos, w_in, b_in)


def window_attention(q, k, v, tau, tau_min, heads, wi):
    """q, k, v: float32 [m, C] row-strided views (last dim contiguous) of the in-projection output."""
    _need_gpu(q, k, v, tau)
    for t in (q, k, v):
        if t.dtype != torch.float32 or t.stride(1) != 1:
            raise _lib.Seg3dError("q/k/v must be float32 with a contiguous last dimension")
    return _WindowAttnFn.apply(q, k, v, tau, tau_min, heads, wi)


def _linear_wgrad(x, dy, cin, cout, want_db=True, fork=None):
    """(dW [cout, cin], db [cout] or None) of y = x W^T + b from the tall-skinny split-bf16 kernel (deterministic).
    fork: a _WgradFork of the calling backward function -- the launch goes to the side stream, the caller joins."""
    dw = torch.empty((cout, cin), dtype=torch.float32, device=dy.device)
    db = torch.empty((cout,), dtype=torch.float32, device=dy.device) if want_db else None
    if fork is not None:
        _linear_wgrad_into(fork, x, dy, cin, cout, dw.data_ptr(), db.data_ptr() if want_db else 0)
        return dw, db
    ws_bytes = _lib.query("seg3d_linear_wgrad_workspace_bytes", x.shape[0], cin, cout)
    ws = _workspace(ws_bytes, dy.device)
    _lib.call("seg3d_linear_wgrad", _ptr(x), _ptr(dy), x.shape[0], cin, cout, _ptr(dw), _ptr(db), _ptr(ws), ws_bytes, _stream())
    return dw, db


def _linear_wgrad_into(fk, x, dy, cin, cout, dw_ptr, db_ptr, *keep):
    """Dense weight gradient in two halves: the partial blocks per row chunk now, on fk's stream; their fixed-order sum
    queued on fk (it runs at fk.join(): alone, or batched with every other pending sum at the end of the backward pass
    when the join is deferred).  dw_ptr / db_ptr: device addresses of [cout, cin] / [cout] (0 = no bias gradient); the
    caller keeps those tensors alive until its join (queued jobs hold addresses only: a second reference to a gradient
    tensor would make AccumulateGrad clone it instead of keeping it)."""
    m = x.shape[0]
    ws_bytes = _lib.query("seg3d_linear_wgrad_workspace_bytes", m, cin, cout)
    ws = _workspace(ws_bytes, dy.device)
    chunks = ctypes.c_int32(0)
    _lib.call("seg3d_linear_wgrad_partials_xbf16" if x.dtype == torch.bfloat16 else "seg3d_linear_wgrad_partials",
              _ptr(x), _ptr(dy), m, cin, cout, 1 if db_ptr else 0, _ptr(ws), ws_bytes,
              ctypes.byref(chunks), fk.fork(ws, x, dy, *keep))
    fk.add_reduce(ws, chunks.value, cin * cout + cout, cin * cout, dw_ptr, db_ptr, *keep)


class _Ctx:
    """Stand-in for autograd's ctx when an enclosing Function drives another Function's forward / backward."""

    def __init__(self, *needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


GELU_SAVED_GRAD = os.environ.get("SEG3D_GELU_SAVED_GRAD", "1") != "0"  # 0 = torch gelu + gelu_backward passes (A/B)


class _EncoderLayerFn(torch.autograd.Function):
    """One post-norm encoder layer (point_transformer_layer.py:289-298) as a single autograd node:
        a  = out_proj(window_attention(in_proj(x, pos)))          x1 = x  + s1 * LN1(a)
        m  = fc2(gelu(fc1(x1)))                                   x2 = x1 + s2 * LN2(m)
    Same kernels as the composed modules.  What the single node buys: the two residual-path gradients are added in the
    epilogue of the branch's last input-gradient GEMM (no autograd accumulation adds) and the engine walks 1 node
    instead of 9.  (The GELU stays a separate elementwise pass: folded into the fc1 / fc2 GEMM epilogues its erf ran
    serialised after the MFMA loop -- 96 evaluations per lane -- and cost 1.6 ms per step more than the two passes it
    saved, which hide the erf behind their memory traffic.)"""

    @staticmethod
    def forward(ctx, x, pos, w_in, b_in, tau, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, meta):
        heads, tau_min, wi, eps1, eps2, s1, s2, drop_p, drop_seed = meta
        x = _f32c(x)
        c = x.shape[1]
        live = any(ctx.needs_input_grad)  # False under no_grad: nothing of this forward is kept
        c_in = _Ctx(live, False, live, live)
        qk, v = _AttnInProjFn.forward(c_in, x, pos, w_in, b_in)
        c_at = _Ctx(True, True, True, False, False, False)
        o = _WindowAttnPackedFn.forward(c_at, qk, v, tau, tau_min, heads, wi, drop_p, drop_seed)
        # inference: out-projection + LayerNorm + residual (and fc2 + LayerNorm + residual below) are ONE launch each --
        # the normalised row fits a workgroup's column blocks up to C = 192 (seg3d_linear_layernorm_fwd)
        fuse_ln = (LINEAR_LN_FUSED and not live and s1 is None and s2 is None and (c // 16) in (1, 2, 3, 4, 6, 8, 12)
                   and b_out is not None and b2 is not None)
        c_n1 = _Ctx(True, True, True, True, False, False)
        if fuse_ln:
            x1 = _linear_layernorm(o, _linear_pack(w_out, 0), b_out, x, g1, be1, eps1, c, c)
        else:
            a = _linear_apply(o, _linear_pack(w_out, 0), b_out, c, c)
            x1 = _LayerNormResidualFn.forward(c_n1, a, x, g1, be1, eps1, s1)
        hid = w1.shape[0]
        h = _linear_apply(x1, _linear_pack(w1, 0), b1, c, hid)
        if live and GELU_SAVED_GRAD:
            # training: the GELU pass also writes its derivative (in h's storage's place: h is not needed again), and the
            # backward multiplies by it in the epilogue of fc2's input-gradient GEMM -- no gelu_backward pass
            g, gp = torch.empty_like(h), torch.empty_like(h)
            _lib.call("seg3d_gelu_fwd", _ptr(h), h.numel(), _ptr(g), _ptr(gp), _stream())
            h = gp
        else:
            g = torch.nn.functional.gelu(h)
        c_n2 = _Ctx(True, True, True, True, False, False)
        if fuse_ln:
            x2 = _linear_layernorm(g, _linear_pack(w2, 0), b2, x1, g2, be2, eps2, hid, c)
        else:
            m = _linear_apply(g, _linear_pack(w2, 0), b2, hid, c)
            x2 = _LayerNormResidualFn.forward(c_n2, m, x1, g2, be2, eps2, s2)
        ctx.parts = (c_in, c_at, c_n1, c_n2)
        ctx.save_for_backward(_saved_rows(o), _saved_rows(x1), h, _saved_rows(g), w_out, w1, w2)
        ctx.bias_params = (b_out, b1, b2)
        ctx.norm_params = (g1, be1, g2, be2)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        c_in, c_at, c_n1, c_n2 = ctx.parts
        o, x1, h, g, w_out, w1, w2 = ctx.saved_tensors
        dx2 = _f32c(dx2)
        c, hid = x1.shape[1], w1.shape[0]
        # MLP branch: LN2 -> fc2 -> gelu' -> fc1, the residual gradient dx2 joins in fc1's epilogue
        # the three weight gradients of this function go to the side stream as soon as their operands are enqueued and
        # co-run with the chain below (the attention backward is the long pole); one join at the end
        fk = _WgradFork(dx2.device)
        c_n1.fork = c_n2.fork = fk
        dm, _, dg2, dbe2, _, _ = _LayerNormResidualFn.backward(c_n2, dx2)
        dw2, db2 = _linear_wgrad(g, dm, hid, c, fork=fk)
        if GELU_SAVED_GRAD:  # h holds gelu'(fc1 output), saved by the forward
            dh = torch.empty((dm.shape[0], hid), dtype=torch.float32, device=dm.device)
            _lib.call("seg3d_linear_fwd_mul", _ptr(dm), dm.shape[0], _ptr(_linear_pack(w2, 1)), _ptr(h), c, hid, _ptr(dh),
                      _stream())
        else:
            dh = torch.ops.aten.gelu_backward(_linear_apply(dm, _linear_pack(w2, 1), None, c, hid), h)
        dw1, db1 = _linear_wgrad(x1, dh, c, hid, fork=fk)
        d_x1 = _linear_apply(dh, _linear_pack(w1, 1), None, hid, c, addend=dx2)
        # attention branch: LN1 -> out_proj -> attention -> in_proj, d_x1 joins in the in-projection's epilogue
        da, _, dg1, dbe1, _, _ = _LayerNormResidualFn.backward(c_n1, d_x1)
        dw_out, db_out = _linear_wgrad(o, da, c, c, fork=fk)
        do = _linear_apply(da, _linear_pack(w_out, 1), None, c, c)
        dqk, dv, dtau = _WindowAttnPackedFn.backward(c_at, do)[:3]
        c_in.dx_addend = d_x1
        dx, _, dw_in, db_in = _AttnInProjFn.backward(c_in, dqk, dv)
        b_out, b1, b2 = ctx.bias_params
        g1, be1, g2, be2 = ctx.norm_params
        fk.join((w_out, dw_out), (w1, dw1), (w2, dw2), (b_out, db_out), (b1, db1), (b2, db2), (g1, dg1), (be1, dbe1),
                (g2, dg2), (be2, dbe2))
        c_n1.fork = c_n2.fork = None
        ctx.parts = None
        return dx, None, dw_in, db_in, dtau, dw_out, db_out, dg1, dbe1, dw1, db1, dw2, db2, dg2, dbe2, None


def encoder_layer_fits(x, c, hid, heads):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and c % 16 == 0 and hid % 16 == 0 and c <= 512
            and c % heads == 0 and CONV_PRECISION == "bf16x3")


# ------------------------------------------------------------------------------------------ a7/a24-a26
class _SegmentReduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg, mode):
        x = _f32c(x)
        n, c = x.shape
        out = torch.empty((seg.n_segments, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((seg.n_segments, c), dtype=torch.int32, device=x.device) if mode == REDUCE_MAX else None
        _lib.call("seg3d_segment_reduce_fwd", _ptr(x), c, _ptr(seg.order), _ptr(seg.offsets), seg.n_segments, mode,
                  _ptr(out), _ptr(arg), _stream())
        ctx.seg, ctx.mode, ctx.arg, ctx.n = seg, mode, arg, n
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        c = dout.shape[1]
        seg = ctx.seg
        if ctx.mode == REDUCE_MAX:
            dx = torch.zeros((ctx.n, c), dtype=torch.float32, device=dout.device)
        else:
            dx = torch.empty((ctx.n, c), dtype=torch.float32, device=dout.device)
        _lib.call("seg3d_segment_reduce_bwd", _ptr(dout), c, _ptr(seg.ids), ctx.n, _ptr(seg.offsets), _ptr(ctx.arg),
                  seg.n_segments, ctx.mode, _ptr(dx), _stream())
        return dx, None, None


def segment_reduce(x, seg, mode):
    _need_gpu(x)
    return _SegmentReduceFn.apply(x, seg, mode)


_MODES = {"sum": REDUCE_SUM, "add": REDUCE_SUM, "mean": REDUCE_MEAN, "max": REDUCE_MAX}


def scatter(src, index, dim=0, reduce="sum", dim_size=None):
    """``torch_scatter.scatter(src, index, dim=0, reduce='mean'|'max'|'sum')``: rows = index.max()+1
    (one host sync unless ``dim_size`` is given), empty rows are 0, differentiable w.r.t. src."""
    if dim != 0 or src.dim() != 2:
        raise _lib.Seg3dError("scatter: only dim=0 on [N, C] tensors is on this path")
    _need_gpu(src, index)
    n_seg = int(index.max().item()) + 1 if dim_size is None else int(dim_size)
    return segment_reduce(src, SegmentIndex(index, n_seg), _MODES[reduce])


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, ids, seg):
        feats = _f32c(feats)
        n, c = ids.shape[0], feats.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=feats.device)
        _lib.call("seg3d_gather_rows", _ptr(feats), _ptr(ids), n, c, _ptr(out), _stream())
        ctx.ids, ctx.seg, ctx.m = ids, seg, feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        seg = ctx.seg if ctx.seg is not None else SegmentIndex(ctx.ids, ctx.m)
        c = dout.shape[1]
        dfeat = torch.empty((ctx.m, c), dtype=torch.float32, device=dout.device)
        _lib.call("seg3d_segment_reduce_fwd", _ptr(dout), c, _ptr(seg.order), _ptr(seg.offsets), ctx.m, REDUCE_SUM,
                  _ptr(dfeat), None, _stream())
        return dfeat, None, None


def gather_rows(feats, ids, seg=None):
    """out[i] = feats[ids[i]], zeros where ids[i] == -1.  ``seg`` (SegmentIndex of ids over the feature
    rows) may be passed to reuse an existing CSR in the backward."""
    _need_gpu(feats, ids)
    return _GatherRowsFn.apply(feats, _i32c(ids), seg)


def voxel_to_point(feats, coords):
    """``seg3d.ops.voxel_to_point`` (voxel_to_point.py:4-17)."""
    return gather_rows(feats, coords)


def voxel_max_pooling(feats, coords):
    """``seg3d.ops.voxel_max_pooling`` (voxel_pooling.py:62-73): rows with coords == -1 are ignored."""
    return scatter(feats, coords, reduce="max")


def voxel_avg_pooling(feats, coords, counts):
    """``seg3d.ops.voxel_avg_pooling`` (voxel_pooling.py:10-60): out[M, C] with M = len(counts); rows whose
    id is outside [0, M) are skipped (voxel_pooling.cpp:14); the divisor is the number of rows actually
    pooled, which equals ``counts`` for every caller-side construction of it."""
    _need_gpu(feats, coords, counts)
    m = counts.shape[0]
    ids = _i32c(coords)
    ids = torch.where((ids >= 0) & (ids < m), ids, torch.full_like(ids, -1))
    return segment_reduce(feats, SegmentIndex(ids, m), REDUCE_MEAN)


# ------------------------------------------------------------------------------------------ kNN
KNN_GRID_MIN_POINTS = int(os.environ.get("SEG3D_KNN_GRID_MIN", "8192"))  # below this the brute-force kernel is faster
# (cell size in metres, shells walked) fine -> coarse; exactness does not depend on the choice
# cells in a ratio of exactly 8 let the kernel search dense coarse cells through their sub-cells
KNN_GRID_LEVELS = ((0.05, 2), (0.4, 3), (3.2, 4))  # measured best on the multi-sweep bench (tools/knn_bench_model.py)
if os.environ.get("SEG3D_KNN_LEVELS"):  # e.g. "0.1:2,0.5:3,3.0:4"
    KNN_GRID_LEVELS = tuple((float(a), int(b)) for a, b in (t.split(":") for t in os.environ["SEG3D_KNN_LEVELS"].split(",")))


class _KnnLevel(ctypes.Structure):  # seg3d_knn_level
    _fields_ = [("sorted_xyz", ctypes.c_void_p), ("src_index", ctypes.c_void_p), ("cell_end", ctypes.c_void_p),
                ("table_keys", ctypes.c_void_p), ("table_vals", ctypes.c_void_p), ("capacity", ctypes.c_int64),
                ("cell", ctypes.c_float), ("max_ring", ctypes.c_int32)]


def knn_query(nsample, xyz, new_xyz, offset, new_offset, levels=None):
    """``seg3d.ops.knn_query`` (knn_query.py:7-24): (idx int32 [m, nsample], dist float32 [m, nsample] = sqrt(d2)).
    xyz / new_xyz: contiguous float32 [*, 3]; offset / new_offset: cumulative int32 counts per sample.
    Non-differentiable, like the reference Function (it defines no backward).  ``levels``: ((cell, shells), ...) of the
    grid search, fine to coarse (default KNN_GRID_LEVELS); the result does not depend on it."""
    if new_xyz is None:
        new_xyz = xyz
    _need_gpu(xyz, new_xyz, offset, new_offset)
    if xyz.dim() != 2 or new_xyz.dim() != 2 or not xyz.is_contiguous() or not new_xyz.is_contiguous():
        raise _lib.Seg3dError("knn_query: xyz and new_xyz must be contiguous 2-D tensors (knn_query.py:16)")
    xyz, new_xyz = xyz.detach(), new_xyz.detach()
    if xyz.dtype != torch.float32 or new_xyz.dtype != torch.float32:
        raise _lib.Seg3dError("knn_query: float32 coordinates expected")
    # The CUDA kernel reads both buffers with a stride of 3 floats whatever their second dimension is
    # (knn_query_cuda.cu:97-99): n and m are taken from shape[0] and the memory is reinterpreted, which is
    # what happens when DeepFusionBlock passes [N, 6] points (SURVEY 2.2 "latent bug").  Reproduced as is.
    n, m = xyz.shape[0], new_xyz.shape[0]
    off, noff = _i32c(offset), _i32c(new_offset)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    if n >= KNN_GRID_MIN_POINTS and nsample <= 64 and off.shape[0] <= 255:
        # exact grid search (same results): per level one launch sequence (keys, device sort, gather + cell table),
        # then one query kernel; no device value is read on the host anywhere
        dev = xyz.device
        grid = tuple(levels) if levels is not None else KNN_GRID_LEVELS
        ws_bytes = _lib.query("seg3d_knn_level_workspace_bytes", max(n, m))
        if ws_bytes == 0:
            raise _lib.Seg3dError("seg3d_knn_level_workspace_bytes = 0: no device visible or too many points")
        ws = _workspace(ws_bytes, dev)
        cap = 1 << max(4, (2 * n - 1).bit_length())
        keep, lv = [], (_KnnLevel * len(grid))()
        for li, (cell, max_ring) in enumerate(grid):
            sorted_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
            src_index = torch.empty((n,), dtype=torch.int32, device=dev)
            cell_end = torch.empty((n,), dtype=torch.int32, device=dev)
            tkeys = torch.empty((cap,), dtype=torch.int64, device=dev)
            tvals = torch.empty((cap,), dtype=torch.int32, device=dev)
            _lib.call("seg3d_knn_level_build", _ptr(xyz), n, _ptr(off), off.shape[0], float(cell), _ptr(sorted_xyz),
                      _ptr(src_index), _ptr(cell_end), _ptr(tkeys), _ptr(tvals), cap, _ptr(ws), ws_bytes, _stream())
            keep.append((sorted_xyz, src_index, cell_end, tkeys, tvals))
            lv[li] = _KnnLevel(sorted_xyz.data_ptr(), src_index.data_ptr(), cell_end.data_ptr(), tkeys.data_ptr(),
                               tvals.data_ptr(), cap, float(cell), int(max_ring))
        if new_xyz.data_ptr() == xyz.data_ptr() and m == n:
            qorder = keep[0][1]  # self-query: the points' own finest-level cell order
        else:
            qorder = torch.empty((m,), dtype=torch.int32, device=dev)
            _lib.call("seg3d_knn_query_order", _ptr(new_xyz), m, _ptr(noff), noff.shape[0], float(grid[0][0]), _ptr(qorder),
                      _ptr(ws), ws_bytes, _stream())
        _lib.call("seg3d_knn_grid_query", ctypes.cast(lv, ctypes.c_void_p), len(grid), _ptr(new_xyz),
                  _ptr(qorder), m, _ptr(off), _ptr(noff), off.shape[0], int(nsample), _ptr(idx), _ptr(d2), _stream())
        return idx, torch.sqrt(d2)
    _lib.call("seg3d_knn_query", _ptr(xyz), n, _ptr(new_xyz), m, _ptr(off), _ptr(noff), off.shape[0], int(nsample),
              _ptr(idx), _ptr(d2), _stream())
    return idx, torch.sqrt(d2)


class _KnnAttentionFn(torch.autograd.Function):
    """DeepFusionBlock's attention over the kNN rows (deep_fusion.py:31-43) in one kernel each way."""

    @staticmethod
    def forward(ctx, q, k, v, idx, invalid, keep, scale):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        n, d = q.shape
        kk = idx.shape[1]
        out = torch.empty((n, d), dtype=torch.float32, device=q.device)
        need = any(ctx.needs_input_grad[:3])
        prob = torch.empty((n, kk), dtype=torch.float32, device=q.device) if need else None
        _lib.call("seg3d_knn_attention_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(idx), _ptr(invalid), _ptr(keep), n, k.shape[0], kk, d,
                  float(scale), _ptr(out), _ptr(prob), _stream())
        if need:
            ctx.save_for_backward(q, k, v, idx, keep, prob)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, idx, keep, prob = ctx.saved_tensors
        dout = _f32c(dout)
        n, d = q.shape
        kk, n_src = idx.shape[1], k.shape[0]
        # inverse neighbour lists: pairs (i, j) grouped by the source row they read, ascending pair index inside a row
        _, order, offsets = group_index(idx.reshape(-1), n_src, rank=False)
        dq = torch.empty_like(q)
        dk = torch.empty_like(k)
        dv = torch.empty_like(v)
        scratch = torch.empty((2 * n * kk,), dtype=torch.float32, device=q.device)
        _lib.call("seg3d_knn_attention_bwd", _ptr(q), _ptr(k), _ptr(v), _ptr(idx), _ptr(keep), _ptr(prob), _ptr(dout),
                  _ptr(order), _ptr(offsets), n, n_src, kk, d, float(ctx.scale), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(scratch),
                  _stream())
        return dq, dk, dv, None, None, None, None


def knn_attention(q, k, v, idx, invalid=None, keep=None, scale=None):
    """out_i = sum_j dropout(nan_to_num(softmax_j(<q_i, k[idx_ij]> * scale, masked where invalid[idx_ij]))) * v[idx_ij]
    (seg3d/models/layers/deep_fusion.py:31-43) without the [n, K, d] gathers.  q [n, 32], k / v [n_src, 32] float32,
    idx int32 [n, K <= 16]; invalid: bool / uint8 [n_src] or None; keep: float32 [n, K] dropout factors or None."""
    _need_gpu(q, k, v, idx)
    if q.shape[1] != 32 or k.shape[1] != 32 or v.shape[1] != 32 or idx.dim() != 2 or idx.shape[1] > 16:
        raise _lib.Seg3dError("knn_attention: 32 channels and at most 16 neighbours (DeepFusionBlock, segformer.py:51-53)")
    idx = _i32c(idx)
    inv = None if invalid is None else invalid.to(torch.uint8).contiguous()
    kp = None if keep is None else _f32c(keep)
    return _KnnAttentionFn.apply(q, k, v, idx, inv, kp, (q.shape[1] ** -0.5) if scale is None else float(scale))


class _ClassContextFn(torch.autograd.Function):
    """OCR's SpatialGatherModule over row spans (ocr.py:10-36) -- seg3d_class_context_fwd / _bwd."""

    @staticmethod
    def forward(ctx, feats, probs, plan, scale):
        feats, probs = _f32c(feats), _f32c(probs)
        offsets, chunks, chunk_offsets, batch = plan
        m, c = feats.shape
        k = probs.shape[1]
        dev = feats.device
        n_chunks = chunks.shape[0]
        weights = torch.empty((m, k), dtype=torch.float32, device=dev)
        partials = torch.empty((max(n_chunks, 1), k, c), dtype=torch.float32, device=dev)
        stats = torch.empty((batch, k, 2), dtype=torch.float32, device=dev)
        context = torch.empty((batch, k, c), dtype=torch.float32, device=dev)
        _lib.call("seg3d_class_context_fwd", _ptr(feats), _ptr(probs), _ptr(offsets), _ptr(chunks), _ptr(chunk_offsets),
                  n_chunks, batch, m, k, c, float(scale), _ptr(weights), _ptr(partials), _ptr(stats), _ptr(context), _stream())
        ctx.save_for_backward(feats, weights)
        ctx.plan, ctx.scale = plan, scale
        return context

    @staticmethod
    def backward(ctx, dcontext):
        feats, weights = ctx.saved_tensors
        offsets, chunks, chunk_offsets, batch = ctx.plan
        m, c = feats.shape
        k = weights.shape[1]
        dcontext = _f32c(dcontext)
        dfeats = torch.empty_like(feats)
        dprobs = torch.empty_like(weights)
        scratch = torch.empty((m * k + max(chunks.shape[0], 1) * k,), dtype=torch.float32, device=feats.device)
        _lib.call("seg3d_class_context_bwd", _ptr(feats), _ptr(weights), _ptr(dcontext), _ptr(offsets), _ptr(chunks),
                  _ptr(chunk_offsets), chunks.shape[0], batch, m, k, c, float(ctx.scale), _ptr(dfeats), _ptr(dprobs),
                  _ptr(scratch), _stream())
        return dfeats, dprobs, None, None


_CONTEXT_PLANS = {}


def class_context_plan(row_offsets, device):
    """Device tables of seg3d_class_context_* for cumulative per-sample row counts (python ints): offsets [B],
    chunks [n, 2] = (sample, first row) of every 128-row chunk, chunk_offsets [B]; cached per offsets tuple."""
    key = (tuple(int(o) for o in row_offsets), str(device))
    plan = _CONTEXT_PLANS.get(key)
    if plan is None:
        chunks, chunk_offsets, lo = [], [], 0
        for b, hi in enumerate(key[0]):
            chunks += [(b, r) for r in range(lo, hi, 128)]
            chunk_offsets.append(len(chunks))
            lo = hi
        i32 = dict(dtype=torch.int32, device=device)
        plan = (torch.tensor(key[0], **i32), torch.tensor(chunks, **i32).reshape(-1, 2), torch.tensor(chunk_offsets, **i32),
                len(key[0]))
        if len(_CONTEXT_PLANS) > 16:
            _CONTEXT_PLANS.clear()
        _CONTEXT_PLANS[key] = plan
    return plan


def class_context_fits(classes, channels):
    """Shapes the fused kernels take: the backward keeps d context [classes, C] + a 128 x 32 tile in 64 KiB of LDS."""
    return 1 <= classes <= 32 and channels % 4 == 0 and 4 <= channels <= 1024 and (classes * channels + 128 * 32) * 4 <= 64 * 1024


def class_context(feats, probs, row_offsets, scale=1.0):
    """context [B, classes, C] = per sample (rows offsets[b-1] .. offsets[b]) softmax over the sample's rows of
    scale * probs[:, k], times feats (SpatialGatherModule, seg3d/models/layers/ocr.py:10-36), any batch size in one
    launch sequence, differentiable w.r.t. feats and probs.  row_offsets: cumulative row counts per sample (python ints)."""
    _need_gpu(feats, probs)
    if not class_context_fits(probs.shape[1], feats.shape[1]):
        raise _lib.Seg3dError("class_context: at most 32 classes, channels a multiple of 4, classes * channels <= 12288")
    return _ClassContextFn.apply(feats, probs, class_context_plan(row_offsets, feats.device), float(scale))


# ------------------------------------------------------------------------------------------ labels
def prepare_voxel_labels(point_voxel_ids, point_labels, n_voxels, ignore_index=255, cur_point_indices=None):
    """``WaymoDataset.prepare_voxel_labels`` (waymo_dataset.py:213-246) on the device: uint8 [n_voxels], the most
    frequent label among each voxel's (current-sweep) points, ties to the smallest label, ``ignore_index`` where a
    voxel has no such point.  point_voxel_ids: int [N] (-1 = dropped point); point_labels: integer [N_cur] in 0..255;
    cur_point_indices: rows of the current sweep (multi-sweep configs), or None."""
    _need_gpu(point_voxel_ids, point_labels)
    ids = point_voxel_ids if cur_point_indices is None else point_voxel_ids[cur_point_indices]
    if ids.shape[0] != point_labels.shape[0]:
        raise _lib.Seg3dError("prepare_voxel_labels: one label per (current-sweep) point expected")
    seg = SegmentIndex(ids, n_voxels)
    lab = point_labels.to(torch.uint8).contiguous()
    out = torch.empty((int(n_voxels),), dtype=torch.uint8, device=lab.device)
    _lib.call("seg3d_voxel_majority_labels", _ptr(lab), _ptr(seg.order), _ptr(seg.offsets), int(n_voxels), int(ignore_index),
              _ptr(out), _stream())
    return out


def get_voxel_centers(voxel_coords, downsample_scale, voxel_size, point_cloud_range):
    """``seg3d.utils.pointops_utils.get_voxel_centers`` (pointops_utils.py:14-22): (z, y, x) cells -> xyz centres."""
    centers = voxel_coords.flip(1).float()  # (z, y, x) -> (x, y, z) without an index tensor
    vs, lo = _grid_constants(tuple(float(v) * float(downsample_scale) for v in voxel_size),
                             tuple(float(v) for v in point_cloud_range[0:3]), centers.device)
    return (centers + 0.5) * vs + lo


_GRID_CONSTANTS = {}


def _grid_constants(voxel_size, lo, device):
    """Device copies of (voxel size, range origin), made once: a host-to-device copy of a Python list in the middle of a
    step blocks the host until the GPU has drained everything queued before it."""
    key = (voxel_size, lo, str(device))
    if key not in _GRID_CONSTANTS:
        _GRID_CONSTANTS[key] = (torch.tensor(voxel_size, dtype=torch.float32, device=device),
                                torch.tensor(lo, dtype=torch.float32, device=device))
    return _GRID_CONSTANTS[key]


def _sample_offsets_nosync(batch_col, batch_size):
    """Cumulative row count per sample, int32 [batch_size], without the host sync torch.bincount hides."""
    ids = torch.arange(batch_size, device=batch_col.device, dtype=batch_col.dtype)
    return (batch_col.view(-1, 1) == ids.view(1, -1)).sum(dim=0).cumsum(0).to(torch.int32)


def aux_voxel_labels(voxel_coords, aux_voxel_coords, voxel_labels, batch_size, voxel_size, point_cloud_range,
                     aux_scale=8.0):
    """Ground truth of the auxiliary (stride-8) voxel head as tools/train.py:86-104 builds it: every coarse voxel takes
    the label of the nearest fine voxel centre of its sample (knn_query with k = 1).  coords: [M, 4] (b, z, y, x).
    The per-sample counts come from one compare-and-sum instead of a Python loop of host syncs."""
    centers = get_voxel_centers(voxel_coords[:, 1:], 1.0, voxel_size, point_cloud_range).contiguous()
    aux_centers = get_voxel_centers(aux_voxel_coords[:, 1:], aux_scale, voxel_size, point_cloud_range).contiguous()
    off = _sample_offsets_nosync(voxel_coords[:, 0], batch_size)  # nothing in this function waits for the GPU: it
    aux_off = _sample_offsets_nosync(aux_voxel_coords[:, 0], batch_size)  # runs between forward and backward
    # grid for this lookup: one fine voxel per finest cell, one coarse voxel per middle cell (every coarse site has a
    # fine voxel within ~2 coarse pitches: measured 0.96 ms vs 2.45 ms with the point-cloud default on the headline
    # scene); the third level only guards against a whole-segment scan on unusual inputs
    pitch = float(max(voxel_size))
    idx, _ = knn_query(1, centers, aux_centers, off, aux_off, levels=((pitch, 0), (8 * pitch, 3), (64 * pitch, 2)))
    return voxel_labels[idx.reshape(-1).long()]


# ------------------------------------------------------------------------------------------ loss
class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, keep_thresh):
        x = _f32c(logits)
        n, c = x.shape
        lab = labels.contiguous()
        lse = torch.empty((n,), dtype=torch.float32, device=x.device)
        stats = torch.empty((2,), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_cross_entropy_workspace_bytes", n)
        ws = _workspace(ws_bytes, x.device)
        _lib.call("seg3d_cross_entropy_fwd", _ptr(x), _ptr(lab), n, c, int(ignore_index), float(keep_thresh), _ptr(lse),
                  _ptr(stats), _ptr(ws), ws_bytes, _stream())
        ctx.save_for_backward(x, lab, lse, stats)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        x, lab, lse, stats = ctx.saved_tensors
        n, c = x.shape
        dx = torch.empty_like(x)
        gg = _f32c(g.reshape(1))
        _lib.call("seg3d_cross_entropy_bwd", _ptr(x), _ptr(lab), _ptr(lse), _ptr(stats), _ptr(gg), n, c, _ptr(dx), _stream())
        return dx, None, None, None


def _loss_args_ok(logits, labels, max_classes):
    return (logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and labels.dtype == torch.int64
            and labels.dim() == 1 and labels.shape[0] == logits.shape[0] and logits.shape[1] <= max_classes)


def cross_entropy(logits, labels, ignore_index=-100, keep_thresh=None):
    """``F.cross_entropy(logits, labels, ignore_index=...)`` (mean reduction) for float32 [n, C] logits and int64 labels
    on the GPU, in one pass each way.  ``keep_thresh``: OHEMCrossEntropyLoss(keep_thresh) of the reference
    (ohem_cross_entropy_loss.py:23-38) -- only rows whose softmax probability of the target is below it are averaged.
    Returns 0 (not NaN) when no row is counted."""
    if not _loss_args_ok(logits, labels, 4096):
        raise ValueError("cross_entropy: float32 [n, C <= 4096] logits and int64 [n] labels on the GPU")
    return _CrossEntropyFn.apply(logits, labels, ignore_index, keep_thresh if keep_thresh else 0.0)


class _LovaszSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, classes_mode, include, class_weight):
        x = _f32c(logits)
        n, c = x.shape
        lab = labels.contiguous()
        coef = torch.empty((n, c), dtype=torch.float32, device=x.device)
        stats = torch.empty((2 + c,), dtype=torch.float32, device=x.device)
        ws_bytes = _lib.query("seg3d_lovasz_workspace_bytes", n, c)
        if ws_bytes == 0:
            raise _lib.Seg3dError(f"seg3d_lovasz_workspace_bytes({n}, {c}) = 0: n * c must stay below 2^31, c <= 64")
        ws = _workspace(ws_bytes, x.device)
        _lib.call("seg3d_lovasz_softmax_fwd", _ptr(x), _ptr(lab), n, c, int(ignore_index), int(classes_mode),
                  _ptr(include) if include is not None else None, _ptr(class_weight) if class_weight is not None else None,
                  _ptr(coef), _ptr(stats), _ptr(ws), ws_bytes, _stream())
        ctx.save_for_backward(x, coef, stats)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        x, coef, stats = ctx.saved_tensors
        n, c = x.shape
        dx = torch.empty_like(x)
        gg = _f32c(g.reshape(1))
        _lib.call("seg3d_lovasz_softmax_bwd", _ptr(x), _ptr(coef), _ptr(stats), _ptr(gg), n, c, _ptr(dx), _stream())
        return dx, None, None, None, None, None


def lovasz_softmax(logits, labels, ignore_index=255, classes="present", class_weight=None):
    """Lovasz-softmax of the reference's ``LovaszLoss`` defaults (multi_class, per_image=False;
    seg3d/models/losses/lovasz_loss.py:118-158, 268-290) on float32 [n, C <= 64] logits and int64 labels: softmax, then
    for every class the Lovasz extension of the Jaccard loss over the descending errors, mean over the chosen classes
    ('present' | 'all' | list of class ids).  One device sort for all classes (seg3d_lovasz_softmax_fwd)."""
    if not _loss_args_ok(logits, labels, 64):
        raise ValueError("lovasz_softmax: float32 [n, C <= 64] logits and int64 [n] labels on the GPU")
    c = logits.shape[1]
    include = None
    if classes == "present":
        mode = 0
    elif classes == "all":
        mode = 1
    else:
        mode = 1
        include = torch.zeros((c,), dtype=torch.int32)
        include[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        include = include.to(logits.device)
    if class_weight is not None:
        class_weight = _f32c(torch.as_tensor(class_weight, dtype=torch.float32, device=logits.device))
    return _LovaszSoftmaxFn.apply(logits, labels, ignore_index, mode, include, class_weight)
