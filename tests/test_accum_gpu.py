"""Gradient accumulation with gradient sinks on the GPU: the HIP weight-gradient kernels OVERWRITE their bucket slots, so the micro-steps of
``GradBucketReducer.no_sync()`` go through the reducer's accumulators (speechain/optim_sche/abs.py:180-213, ``accum_grad``; the
transformer-wide recipe uses accum_grad 3).  Three micro-batches of 2 utterances, each loss divided by 3, against one backward pass over the six
utterances at once -- every bucket, f32 arithmetic: 2e-5 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(seed, B, V):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 30, 64, generator=g)
    xlen = torch.full((B,), 30)
    text = torch.randint(1, V - 2, (B, 9), generator=g)
    text[:, 0] = V - 1
    text[:, -1] = V - 1
    return x.cuda(), xlen.cuda(), text.cuda(), torch.full((B,), 9).cuda()


def test_held_micro_steps_add_up_to_the_large_batch_gradient():
    from speechain_amd import functional as Fn
    from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
    from speechain_amd.harness import ASRTransformer

    Fn.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(1)
        V = 41
        model = ASRTransformer(d_model=64, num_heads=4, enc_layers=2, dec_layers=2, fdfwd_dim=128, vocab_size=V, dropout=0.0, posenc_maxlen=100).cuda().train()
        red = GradBucketReducer(list(model.parameters()), bucket_mb=0.25, adjacent=attention_fusion_groups(model))
        assert len(red.buckets) >= 2 and red.use_sinks
        x, xlen, text, tlen = _batch(5, 6, V)
        red.zero_grad()
        model(x, xlen, text, tlen)["loss"].backward()
        red.wait()
        torch.cuda.synchronize()
        want = [b["flat"].detach().clone() for b in red.buckets]
        for rnd in range(2):  # twice: the accumulators are reused, zero_grad() must forget the previous update
            red.zero_grad()
            for m in range(3):
                sl = slice(2 * m, 2 * m + 2)
                loss = model(x[sl], xlen[sl], text[sl], tlen[sl])["loss"] / 3
                if m < 2:
                    with red.no_sync():
                        loss.backward()
                else:
                    loss.backward()
            red.wait()
            torch.cuda.synchronize()
            for i, (b, w) in enumerate(zip(red.buckets, want)):
                e = float((b["flat"] - w).norm() / (w.norm() + 1e-30))
                assert e < 2e-5, (rnd, i, e)
            for b in red.buckets:  # param.grad is the slot view again
                for p, v in zip(b["params"], b["views"]):
                    assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()
        Fn.GradSinks.clear()
