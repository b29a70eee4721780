"""One rank of tests/test_gpu_training.py::test_sync_batchnorm_matches_full_batch (started by dist.launch_local_ranks;
all ranks share cuda:0 over gloo, which a one-GPU box allows and RCCL does not)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openseg3d_amd import dist as D, ops  # noqa: E402


def main():
    rank, world, _ = D.init_job(backend="gloo", share_device=True)
    dev = torch.device("cuda:0")
    c = 64
    rows = [300 + 170 * r for r in range(world)]
    gen = torch.Generator().manual_seed(3)
    xs = [(torch.randn(n, c, generator=gen) * (1.0 + 0.5 * r) + 0.3 * r) for r, n in enumerate(rows)]
    rs = [torch.randn(n, c, generator=gen) for n in rows]
    gs = [torch.randn(n, c, generator=gen) for n in rows]
    gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen)

    bn = torch.nn.SyncBatchNorm(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    x = xs[rank].to(dev).requires_grad_()
    res = rs[rank].to(dev).requires_grad_()
    y = ops.batch_norm_act(x, bn, relu=True, res=res)
    (y * gs[rank].to(dev)).sum().backward()

    # reference: one BatchNorm1d over the rows of all ranks, fp64
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double().train()
    with torch.no_grad():
        ref.weight.copy_(gamma)
        ref.bias.copy_(beta)
    xa = torch.cat(xs).double().requires_grad_()
    ra = torch.cat(rs).double().requires_grad_()
    ya = torch.relu(ref(xa) + ra)
    (ya * torch.cat(gs).double()).sum().backward()
    lo = sum(rows[:rank])
    sl = slice(lo, lo + rows[rank])

    def close(a, b, tol):
        err = float((a.detach().cpu().double() - b.detach()).abs().max())
        assert err <= tol * max(1.0, float(b.abs().max())), (err, tol)

    close(y, ya[sl], 1e-5)
    close(x.grad, xa.grad[sl], 1e-5)
    close(res.grad, ra.grad[sl], 1e-6)
    close(bn.running_mean, ref.running_mean, 1e-6)
    close(bn.running_var, ref.running_var, 1e-6)
    assert int(bn.num_batches_tracked) == 1
    # parameter gradients stay rank-local (DDP averages them): their sum over the ranks is the full-batch gradient
    gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
    dist.all_reduce(gw)
    dist.all_reduce(gb)
    close(gw, ref.weight.grad, 1e-5)
    close(gb, ref.bias.grad, 1e-5)
    # eval mode: running statistics, no exchange
    bn.eval()
    with torch.no_grad():
        ye = ops.batch_norm_act(x.detach(), bn, relu=False)
    close(ye, torch.nn.functional.batch_norm(xs[rank].double(), ref.running_mean, ref.running_var, ref.weight, ref.bias,
                                             False, 0.0, 1e-3), 1e-5)
    D.job_barrier(dev)
    if rank == 0:
        print("SYNCBN OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
