"""N > 1 path on CPU: world_size 2 over gloo (the GPU run uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openseg3d_amd import dist as D
    r, w, _ = D.init_job(backend="gloo")
    assert (r, w) == (rank, world) and dist.is_initialized()
    # scenes shard without overlap
    seeds = D.scene_seeds(rank, 3)
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    assert sorted(sum(gathered, [])) == list(range(3 * world))
    from openseg3d_amd import config, segformer  # noqa: F401  (the package imports on a CPU-only rank)
    from openseg3d_amd import ops as ops_mod
    # the wrapper bench.py trains through (tools/train.py:246-247, 276-279): DistributedDataParallel over this process group;
    # after backward every rank holds the mean of the ranks' gradients
    torch.manual_seed(7)
    small = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    twin = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    twin.load_state_dict(small.state_dict())
    os.environ["SEG3D_DDP"] = "torch"
    ddp = D.wrap_data_parallel(small, torch.device("cpu"))
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel) and not ddp.broadcast_buffers
    xs = [torch.randn(32 + 8 * r, 8, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    ddp(xs[rank]).square().mean().backward()
    want = [torch.zeros_like(p) for p in twin.parameters()]
    for r in range(world):  # every rank recomputes all ranks' local gradients on the unwrapped twin
        twin.zero_grad()
        twin(xs[r]).square().mean().backward()
        for acc, p in zip(want, twin.parameters()):
            acc += p.grad / world
    for p, ref_g in zip(small.parameters(), want):
        assert torch.allclose(p.grad, ref_g, rtol=1e-5, atol=1e-7)
    # the default wrapper (dist.SceneParallel: one exchange after the pass out of a flat arena): same surface, same result;
    # rank r starts from DIFFERENT weights -- construction must broadcast rank 0's; a dict result and a parameter that
    # takes no part in the pass; no_sync() accumulates locally
    del os.environ["SEG3D_DDP"]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
            self.unused = torch.nn.Parameter(torch.ones(5))

        def forward(self, x):
            y = self.body(x)
            return {"a": y, "b": y.detach(), "n": x.shape[0], "deep": {"list": [y * 2.0, "text"]}}

    torch.manual_seed(100 + rank)
    net = Net()
    sp = D.wrap_data_parallel(net, torch.device("cpu"))
    assert isinstance(sp, D.SceneParallel) and sp.module is net
    twin2 = Net()
    torch.manual_seed(100)
    ref0 = Net()  # what rank 0 drew
    for p, q in zip(net.parameters(), ref0.parameters()):
        assert torch.equal(p, q)
    twin2.load_state_dict(net.state_dict())
    sp._bucket = 40  # (elements per all-reduce slice: several slices over this small arena)
    res = sp(xs[rank])
    assert res["n"] == xs[rank].shape[0] and not res["b"].requires_grad
    res["a"].square().mean().backward()
    # the first synchronised pass exchanges after the pass and LEARNS the order in which gradients arrive (rank 0's)
    assert sp.exchanges == 1 and sp.early_slices == 0 and sp._learned
    assert sp._params[sp._order[-1]] is net.unused  # never fires: last in the arena, its slice closes in the final callback
    want = [torch.zeros_like(p) for p in twin2.body.parameters()]
    for r in range(world):
        twin2.zero_grad()
        twin2(xs[r])["a"].square().mean().backward()
        for acc, p in zip(want, twin2.body.parameters()):
            acc += p.grad / world
    for p, ref_g in zip(net.body.parameters(), want):
        assert torch.allclose(p.grad, ref_g, rtol=1e-5, atol=1e-7)
    assert net.unused.grad is not None and float(net.unused.grad.abs().sum()) == 0.0
    sp.zero_grad(set_to_none=True)
    with sp.no_sync():
        sp(xs[rank])["a"].square().mean().backward()
    assert sp.exchanges == 1
    twin2.zero_grad()
    twin2(xs[rank])["a"].square().mean().backward()
    for p, q in zip(net.body.parameters(), twin2.body.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-8)  # local gradient, nothing exchanged
    sp(xs[rank])["a"].square().mean().backward()  # accumulates onto the local gradient, then exchanges the sum
    assert sp.exchanges == 2
    # ... slice by slice DURING the pass this time: the arena follows the arrival order, and every slice whose gradients
    # (and those of the slices in front of it) had arrived was exchanged from its post-accumulate hook
    assert len(sp._slices) >= 4 and 0 < sp.early_slices < len(sp._slices)
    for p, ref_g in zip(net.body.parameters(), want):
        assert torch.allclose(p.grad, 2 * ref_g, rtol=1e-5, atol=1e-7)
    assert float(net.unused.grad.abs().sum()) == 0.0
    # a loss built from a tensor nested two containers deep still ends in an exchange (one level only before)
    sp.zero_grad(set_to_none=True)
    early = sp.early_slices
    nested = sp(xs[rank])
    assert nested["deep"]["list"][1] == "text"
    nested["deep"]["list"][0].square().mean().backward()
    assert sp.exchanges == 3 and sp.early_slices > early
    for p, ref_g in zip(net.body.parameters(), want):
        assert torch.allclose(p.grad, 4 * ref_g, rtol=1e-5, atol=1e-6)  # (2y)^2
    # the same exchange with the overlap switched off (SEG3D_DDP_OVERLAP=0): one exchange after the pass, same numbers
    sp.zero_grad(set_to_none=True)
    sp._overlap = False
    sp(xs[rank])["a"].square().mean().backward()
    assert sp.exchanges == 4
    for p, ref_g in zip(net.body.parameters(), want):
        assert torch.allclose(p.grad, ref_g, rtol=1e-5, atol=1e-7)
    sp._overlap = True

    class Blind(torch.nn.Module):  # a training forward whose result carries no differentiable tensor: refuse, do not diverge
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def forward(self, x):
            return {"n": 3, "t": (x * self.w).detach()}

    blind = D.SceneParallel(Blind())
    try:
        blind(torch.ones(3))
        raise AssertionError("expected a RuntimeError")
    except RuntimeError as e:
        assert "no differentiable tensor" in str(e)
    blind.close()
    # deferral is scoped to the wrapper's own parameters: a parameter of another module is not exempt
    assert net.body[0].weight in ops_mod.DEFER_OWNED and blind.module.w not in ops_mod.DEFER_OWNED
    sp.eval()
    with torch.no_grad():
        assert sp(xs[rank])["a"].shape == (xs[rank].shape[0], 4)
    # --sync_bn: the conversion the wrapper applies first (torch admits SyncBatchNorm under DDP on GPU modules only)
    conv = D.convert_sync_bn(torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.BatchNorm1d(8)))
    assert isinstance(conv[1], torch.nn.SyncBatchNorm)
    # SyncBatchNorm statistics: per-rank (mean, M2, count) gathered and merged == moments of the concatenated rows
    from openseg3d_amd import ops
    rows = [torch.randn(40 + 25 * r, 12, generator=torch.Generator().manual_seed(90 + r)) * (1 + r) + 3 * r for r in range(world)]
    mine = rows[rank].double()
    local = torch.cat([mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0), torch.tensor([float(mine.shape[0])], dtype=torch.float64)])
    g = ops._all_gather_rows(local, dist.group.WORLD)
    mean, m2, total = ops.combine_moments(g[:, :12], g[:, 12:24], g[:, 24:])
    allrows = torch.cat(rows).double()
    assert int(total) == allrows.shape[0]
    assert torch.allclose(mean, allrows.mean(0), rtol=1e-12, atol=1e-12)
    assert torch.allclose(m2 / total, allrows.var(0, unbiased=False), rtol=1e-12, atol=1e-12)
    # throughput aggregate: sum of units over max of time
    t, u = D.aggregate_throughput(1.0 + rank, 100.0 * (rank + 1), torch.device("cpu"))
    assert (t, u) == (float(world), 50.0 * world * (world + 1))  # slowest rank's time, all ranks' units
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_world_size_2_gloo(world):
    """(world 4: an arena in arrival order, slices closed in order and rank 0's order broadcast with more than one peer --
    what the 4- and 8-GPU runs of bench.py exercise over RCCL.)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.get(timeout=5) for _ in procs) == list(range(world))


_CHILD = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from openseg3d_amd import dist as D
rank, world, local = D.init_job(backend="gloo")
assert world == 2 and local == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == {fail_rank}:
    sys.exit(7)
D.job_barrier(torch.device("cpu"))
if rank == 0:
    print("SUM", int(t.item()), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.timeout(300)
def test_launcher_starts_and_reaps_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without an outer launcher goes through dist.launch_local_ranks: N fresh children with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exit code 0 only if every rank succeeded, and a failing rank does
    not leave the others waiting in a collective."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = tmp_path / "ok.py"
    ok.write_text(_CHILD.format(root=root, fail_rank=-1))
    # the launcher itself runs in a child so that its ranks' stdout can be captured here
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys\nsys.path.insert(0, {root!r})\nfrom openseg3d_amd import dist as D\n"
                   "sys.exit(D.launch_local_ranks([sys.executable, sys.argv[1]], 2))\n")
    out = subprocess.run([sys.executable, str(drv), str(ok)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert [l for l in out.stdout.splitlines() if l.startswith("SUM")] == ["SUM 3"]
    bad = tmp_path / "bad.py"
    bad.write_text(_CHILD.format(root=root, fail_rank=1))
    out = subprocess.run([sys.executable, str(drv), str(bad)], capture_output=True, text=True, timeout=240)
    assert out.returncode != 0 and "SUM" not in out.stdout
