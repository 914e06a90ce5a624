"""CPU, world_size 2 over gloo: the collectives of FeatureNormalization's DDP branches (feat_norm.py:236-265 with the gathers
:341-424) give every rank the statistics a single process computes on the concatenated batch -- including ragged per-rank batch
sizes.  Only the (B, F)-sized bookkeeping runs here (torch + torch.distributed); the kernels have no CPU path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    gen = torch.Generator().manual_seed(21)
    feat = torch.randn(7, 12, 6, generator=gen) * 2 + 1
    lens = torch.tensor([12, 5, 9, 3, 12, 7, 2])
    ids = torch.tensor([4, 4, 8, 2, 8, 4, 2])
    return feat, lens, ids


def _stats(feat, lens, clamp=1e-10):
    means = torch.stack([feat[i][: lens[i]].mean(0) for i in range(feat.size(0))])
    stds = torch.clamp(torch.stack([feat[i][: lens[i]].std(0) for i in range(feat.size(0))]), min=clamp)
    return means, stds


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechain_amd.module.norm.feat_norm import FeatureNormalization, ddp_batch_statistics

    feat, lens, ids = _data()
    sl = slice(0, 4) if rank == 0 else slice(4, 7)  # ragged: 4 utterances on rank 0, 3 on rank 1
    means, stds = _stats(feat[sl], lens[sl])
    bm, bs = ddp_batch_statistics(means, stds, means.size(0))
    bm_only, none = ddp_batch_statistics(means, None, means.size(0))
    all_ids, all_means, all_stds = FeatureNormalization._gather_groups(ids[sl], means, stds, means.size(0))
    q.put((rank, bm.numpy(), bs.numpy(), bm_only.numpy(), none is None, np.array(all_ids), all_means.numpy(), all_stds.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_statistics_equal_the_single_process_result_on_the_concatenated_batch():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    feat, lens, ids = _data()
    means, stds = _stats(feat, lens)
    for rank, bm, bs, bm_only, none_ok, all_ids, all_means, all_stds in got:
        assert none_ok
        assert np.allclose(bm, means.mean(0).numpy(), atol=1e-6) and np.allclose(bs, stds.mean(0).numpy(), atol=1e-6)
        assert np.allclose(bm_only, bm)
        assert all_ids.tolist() == ids.tolist()
        assert np.allclose(all_means, means.numpy(), atol=0) and np.allclose(all_stds, stds.numpy(), atol=0)
    # and these batch statistics are what the oracle's global type registers after one training step on the whole batch
    m = R.FeatureNormalization()
    m(feat, lens, epoch=1)
    assert np.allclose(got[0][1], m.buffers["global_mean"].numpy(), atol=1e-6)
    assert np.allclose(got[0][2], m.buffers["global_std"].numpy(), atol=1e-6)
