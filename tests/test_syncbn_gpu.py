"""Under DDP the reference turns every BatchNorm into ``torch.nn.SyncBatchNorm`` (speechain/runner.py:303-310 ``enable_syncbatchnorm`` default
True, :1856-1858 ``convert_sync_batchnorm`` before the DDP wrap): the Conv2d prenet and the Conv1d postnet then normalise with the statistics of
the batches of ALL ranks, forward and backward.  The HIP blocks do the same when their BatchNorm holder has been converted: C-sized float64
sums are all-reduced between the reduce and the finish kernels (include/speechain_amd.h, "Synchronised batch statistics").

Test: two processes (gloo over CUDA tensors, both on the one GPU of the box), each with half of a batch, against ONE process on the whole
batch with plain BatchNorm (that path is pinned to the reference by tests/golden/g9_prenet.npz, g10_tts.npz): outputs row for row, the SUM of
the ranks' parameter gradients (the loss is a sum over utterances), the running statistics.  f32 arithmetic: 2e-5 relative."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(kind):
    from speechain_amd.module.postnet.conv1d import Conv1dPostnet
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet

    torch.manual_seed(3)
    if kind == "prenet":
        m = Conv2dPrenet(input_size=80, conv_dims=[16, 24], conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=32)
    else:
        m = Conv1dPostnet(input_size=20, conv_dims=[32, -1, 0], conv_kernel=5, conv_batchnorm=True, conv_activation="Tanh", conv_dropout=None)
    with torch.no_grad():
        for name, p in m.named_parameters():  # BatchNorm weights / biases away from (1, 0) so that every term of the backward formulas counts
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape) * 0.3 + (1.0 if name.endswith("weight") else 0.0))
    return m


def _data(kind):
    g = torch.Generator().manual_seed(9)
    B = 6
    if kind == "prenet":
        x, lens = torch.randn(B, 71, 80, generator=g), torch.tensor([71, 64, 71, 50, 71, 33])
    else:
        x, lens = torch.randn(B, 37, 20, generator=g), torch.tensor([37, 30, 37, 21, 37, 12])
    return x, lens, g


def _run(m, x, lens, wsum):
    out = m(x.cuda(), lens.cuda())
    out = out[0] if isinstance(out, tuple) else out  # (the postnet returns the tensor alone)
    (out.float() * wsum.cuda()).sum().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()}
    bufs = {k: v.detach().float().cpu().clone() for k, v in m.named_buffers() if "running" in k}
    return out.detach().float().cpu(), grads, bufs


def _worker(rank, world, port, q, kind):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechain_amd import functional as Fn

    torch.cuda.set_device(0)
    Fn.set_compute_dtype(torch.float32)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(_build(kind)).cuda().train()
    assert any(isinstance(c, torch.nn.SyncBatchNorm) for c in m.modules())
    x, lens, g = _data(kind)
    wsum = torch.randn(*_shape(kind), generator=g)
    sel = slice(rank, None, world)  # (interleaved halves: both see the same padded length)
    out, grads, bufs = _run(m, x[sel], lens[sel], wsum[sel])
    q.put((rank, out.numpy(), {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in bufs.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _shape(kind):
    return (6, 17, 32) if kind == "prenet" else (6, 37, 20)


@pytest.mark.parametrize("kind", ["prenet", "postnet"])
def test_synchronised_batchnorm_equals_one_process_on_the_whole_batch(kind):
    from speechain_amd import functional as Fn

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, grads, bufs = q.get(timeout=300)
        res[r] = (torch.from_numpy(out), {k: torch.from_numpy(v) for k, v in grads.items()}, {k: torch.from_numpy(v) for k, v in bufs.items()})
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    Fn.set_compute_dtype(torch.float32)
    try:
        m = _build(kind).cuda().train()
        x, lens, g = _data(kind)
        wsum = torch.randn(*_shape(kind), generator=g)
        assert tuple(wsum.shape) == _shape(kind)
        out, grads, bufs = _run(m, x, lens, wsum)
    finally:
        Fn.set_compute_dtype(None)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    for r in range(world):
        assert rel(res[r][0], out[r::world]) < 2e-5, (kind, r, rel(res[r][0], out[r::world]))
        for k, v in bufs.items():  # running statistics: every rank holds the global ones (unbiased variance over all positions of all ranks)
            assert rel(res[r][2][k], v) < 2e-5, (k, rel(res[r][2][k], v))
    for k, v in grads.items():
        tot = res[0][1][k] + res[1][1][k]
        assert rel(tot, v) < 5e-5, (kind, k, rel(tot, v))
    # and the two ranks' contributions differ (each is its own half's sum, not a copy of the total)
    k0 = next(iter(grads))
    assert rel(res[0][1][k0], res[1][1][k0]) > 1e-3
