"""CPU, world_size 2 over gloo: the bucketed gradient reducer averages gradients exactly like a single process on the
concatenated batch, keeps param.grad as views into its flat buckets, and survives several steps."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                               torch.nn.Linear(64, 8))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechain_amd.ddp import GradBucketReducer

    model = _model()
    if rank == 1:  # de-synchronise on purpose: the start-up broadcast must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradBucketReducer(model.parameters(), bucket_mb=0.005, adjacent=[[model[2].weight, model[0].weight]])  # ~1300 floats per bucket
    red.broadcast_parameters()
    assert len(red.buckets) >= 3
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    x_all, y_all = torch.randn(3, 8, 16, generator=g), torch.randn(3, 8, 8, generator=g)
    for step in range(3):
        red.zero_grad()
        x, y = x_all[step, rank * 4: rank * 4 + 4], y_all[step, rank * 4: rank * 4 + 4]
        ((model(x) - y) ** 2).mean().backward()
        red.wait()
        flat_ptrs = {b["flat"].data_ptr(): b["flat"].numel() for b in red.buckets}
        for p in model.parameters():  # gradients stayed views into the flat buckets
            assert any(base <= p.grad.data_ptr() < base + 4 * n for base, n in flat_ptrs.items())
        opt.step()
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch of 8
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    x_all, y_all = torch.randn(3, 8, 16, generator=g), torch.randn(3, 8, 8, generator=g)
    for step in range(3):
        opt.zero_grad()
        ((model(x_all[step]) - y_all[step]) ** 2).mean().backward()
        opt.step()
    for r in range(world):
        for a, b in zip(results[r], model.parameters()):
            a = torch.from_numpy(a)
            assert torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())
