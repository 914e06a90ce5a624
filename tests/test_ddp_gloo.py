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


# ---- gradient accumulation: no_sync() micro-steps (speechain/optim_sche/abs.py:180-213, `accum_grad`) ---------------------------------
def _accum_worker(rank, world, port, q, sinks):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechain_amd.ddp import GradBucketReducer

    model = _model()
    red = GradBucketReducer(model.parameters(), bucket_mb=0.005, sinks=sinks)
    red.broadcast_parameters()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(2)
    accum = 3
    x_all, y_all = torch.randn(2, accum, 8, 16, generator=g), torch.randn(2, accum, 8, 8, generator=g)
    launched = []
    orig = dist.all_reduce

    def counting(*a, **k):
        launched.append(1)
        return orig(*a, **k)

    dist.all_reduce = counting
    for step in range(2):
        red.zero_grad()
        for m in range(accum):
            x, y = x_all[step, m, rank * 4: rank * 4 + 4], y_all[step, m, rank * 4: rank * 4 + 4]
            loss = ((model(x) - y) ** 2).mean() / accum
            if m < accum - 1:
                before = len(launched)
                with red.no_sync():
                    loss.backward()
                assert len(launched) == before  # nothing left the rank
            else:
                loss.backward()
        red.wait()
        opt.step()
    assert len(launched) == 2 * len(red.buckets)  # one collective per bucket per UPDATE
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def _accum_case(sinks):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q, sinks)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one process, each update on the 3 x 8 samples of its micro-batches at once
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(2)
    x_all, y_all = torch.randn(2, 3, 8, 16, generator=g), torch.randn(2, 3, 8, 8, generator=g)
    for step in range(2):
        opt.zero_grad()
        ((model(x_all[step].reshape(24, 16)) - y_all[step].reshape(24, 8)) ** 2).mean().backward()
        opt.step()
    for r in range(world):
        for a, b in zip(results[r], model.parameters()):
            a = torch.from_numpy(a)
            assert torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())


def test_accumulated_micro_steps_match_one_large_batch_with_slot_overwriting_gradients():
    _accum_case(True)   # gradient sinks: slots are overwritten per micro-step, held steps go through the accumulators


def test_accumulated_micro_steps_match_one_large_batch_without_sinks():
    _accum_case(False)  # autograd adds into the slot views
