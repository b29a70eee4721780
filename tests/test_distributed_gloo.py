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
    from openseg3d_amd import config, dist as D, segformer
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # scenes shard without overlap
    seeds = D.scene_seeds(rank, 3)
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    assert sorted(sum(gathered, [])) == list(range(3 * world))
    assert D.shard_indices(5, rank, world) == ([0, 2, 4] if rank == 0 else [1, 3, 0])
    # every rank starts from rank 0's weights
    torch.manual_seed(100 + rank)
    cfg = config.default_cfg()
    model = segformer.build_segmentor(cfg, config.DatasetSpec(cfg))
    D.broadcast_parameters(model)
    probe = getattr(model.point_transformer.conv_down2, "0").weight.detach().clone()
    ref = probe.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(probe, ref)
    # gradient exchange: mean over ranks, bucket boundaries included
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (1 + i % 3))
    D.allreduce_gradients(model, bucket_bytes=1 << 20)
    for i, p in enumerate(model.parameters()):
        assert torch.allclose(p.grad, torch.full_like(p, 1.5 * (1 + i % 3)))
    # throughput aggregate: sum of units over max of time
    t, u = D.aggregate_throughput(1.0 + rank, 100.0 * (rank + 1), torch.device("cpu"))
    assert (t, u) == (2.0, 300.0)
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.get(timeout=5) for _ in procs) == [0, 1]
