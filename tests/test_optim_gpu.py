"""speechain_amd.optim.FusedAdam (csrc/optim.hip: global-norm clip + Adam over the flat gradient buckets, two launches) against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, the pair the reference's OptimScheduler.step runs (speechain/optim_sche/abs.py:232-253).
Tolerance: 2e-6 relative on parameters and moments after 6 steps (float32; the only differences are the order of the norm's
summation and one rounding in the bias-corrected denominator)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_norm", [0.5, None])
def test_fused_adam_matches_torch(max_norm):
    from speechain_amd.ddp import GradBucketReducer
    from speechain_amd.functional import GradSinks
    from speechain_amd.optim import FusedAdam

    gen = torch.Generator().manual_seed(3)
    shapes = [(257, 33), (64,), (5, 7, 3), (1,), (130001,), (96, 96)]  # odd sizes: scalar tails, several chunks, several buckets
    mine = [torch.nn.Parameter(torch.randn(s, generator=gen).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    try:
        red = GradBucketReducer(mine, bucket_mb=0.3)
        assert len(red.buckets) >= 2
        lr = torch.tensor([0.0], device="cuda")
        opt = FusedAdam(red, lr, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=max_norm)
        topt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
        for p, shape in zip(mine, shapes):
            assert p.shape == torch.Size(shape) and p.is_contiguous()
        for step in range(6):
            rate = 1e-3 * (1 + step)  # a scheduler changes the rate every step
            lr.fill_(rate)
            for g_ in topt.param_groups:
                g_["lr"] = rate
            scale = 10.0 if step % 2 else 0.01  # clipped and unclipped steps
            for p, q in zip(mine, ref):
                g = torch.randn(p.shape, generator=gen).cuda() * scale
                q.grad = g.clone()
                p.grad.copy_(g)  # the bucket slot
            if max_norm:
                tn = torch.nn.utils.clip_grad_norm_(ref, max_norm=max_norm)
            topt.step()
            opt.step()
            if max_norm:
                assert abs(float(opt.grad_norm) - float(tn)) < 1e-5 * float(tn)
                assert abs(float(opt.clip_coef) - min(1.0, max_norm / (float(tn) + 1e-6))) < 1e-6
            else:
                assert float(opt.clip_coef) == 1.0
        assert float(opt.state[2]) == 6.0
        for p, q in zip(mine, ref):
            assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=2e-7), float((p - q).abs().max())
            m, v = opt.state_of(p)
            st = topt.state[q]
            # torch forms m by lerp, m + (1 - b1)(g - m), the kernel by fma(b1, m, (1 - b1) g): equal up to an ulp of the LARGER of m and g,
            # so the moments are judged against each tensor's scale
            assert torch.allclose(m, st["exp_avg"], rtol=2e-6, atol=2e-6 * float(st["exp_avg"].abs().max()))
            assert torch.allclose(v, st["exp_avg_sq"], rtol=2e-6, atol=2e-6 * float(st["exp_avg_sq"].abs().max()))
    finally:
        GradSinks.clear()


def test_non_finite_gradient_norm_skips_the_step():
    """OptimScheduler.step (speechain/optim_sche/abs.py:239-247) does not update the parameters when the gradient norm is inf / nan.  The
    fused kernels do the same: parameters, both moments and the step counter stay bit for bit, `skipped` reads 1, and the next finite step
    continues as if the bad one had not happened (same result as torch.optim.Adam stepped only on the finite gradients)."""
    from speechain_amd.ddp import GradBucketReducer
    from speechain_amd.functional import GradSinks
    from speechain_amd.optim import FusedAdam

    gen = torch.Generator().manual_seed(5)
    shapes = [(40, 9), (7,), (70001,)]
    mine = [torch.nn.Parameter(torch.randn(s, generator=gen).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    try:
        red = GradBucketReducer(mine, bucket_mb=0.1)
        lr = torch.tensor([1e-3], device="cuda")
        opt = FusedAdam(red, lr, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=5.0)
        topt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
        for step, poison in enumerate([None, float("inf"), float("nan"), None]):
            for p, q in zip(mine, ref):
                g = torch.randn(p.shape, generator=gen).cuda()
                q.grad = g.clone()
                p.grad.copy_(g)
            if poison is not None:
                mine[2].grad.view(-1)[12345] = poison
                before = [(p.detach().clone(), *[t.clone() for t in opt.state_of(p)]) for p in mine]
                count = float(opt.state[2])
                opt.step()
                assert float(opt.skipped) == 1.0 and float(opt.clip_coef) == 0.0
                assert float(opt.state[2]) == count
                for p, (p0, m0, v0) in zip(mine, before):
                    m, v = opt.state_of(p)
                    assert torch.equal(p.detach(), p0) and torch.equal(m, m0) and torch.equal(v, v0)
            else:
                torch.nn.utils.clip_grad_norm_(ref, max_norm=5.0)
                topt.step()
                opt.step()
                assert float(opt.skipped) == 0.0
        assert float(opt.state[2]) == 2.0
        for p, q in zip(mine, ref):
            assert torch.isfinite(p).all()
            assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=2e-7)
    finally:
        GradSinks.clear()


def test_forward_sees_the_weights_fused_adam_wrote():
    """FusedAdam updates parameters with a raw kernel, behind torch's version counters.  The cached compute-dtype copies of the weights
    (bf16 matrices, their transposes, the concatenated q|k|v biases) must nevertheless be rebuilt: by `weights.refresh()` (one call per
    step: what a captured step does) and, without it, on the next use (`step()` marks the cache stale).  Checked on an attention block +
    feed-forward layer in bf16: the output after two optimizer steps equals that of a model whose caches were dropped, and differs
    from the output before the update."""
    from speechain_amd import functional as Fn
    from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
    from speechain_amd.functional import GradSinks
    from speechain_amd.module.transformer.encoder import TransformerEncoderLayer
    from speechain_amd.optim import FusedAdam

    for use_refresh in (True, False):
        torch.manual_seed(0)
        Fn.set_compute_dtype(torch.bfloat16)
        Fn.weights.invalidate()
        try:
            layer = TransformerEncoderLayer(d_model=64, num_heads=4, fdfwd_dim=128, att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0).cuda().train()
            for p in layer.parameters():  # biases away from zero, so a stale bias vector shows
                if p.dim() == 1:
                    torch.nn.init.normal_(p, std=0.5)
            red = GradBucketReducer(list(layer.parameters()), bucket_mb=1.0, adjacent=attention_fusion_groups(layer))
            lr = torch.tensor([5e-2], device="cuda")
            opt = FusedAdam(red, lr, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=5.0)
            x = torch.randn(2, 10, 64, device="cuda")
            mask = torch.ones(2, 1, 10, dtype=torch.bool, device="cuda")

            def fwd():
                return layer(x, mask)[0].float()

            y0 = fwd().detach().clone()
            for _ in range(2):
                if use_refresh:
                    Fn.weights.refresh()
                red.zero_grad()
                fwd().pow(2).sum().backward()
                red.wait()
                opt.step()
            if use_refresh:
                Fn.weights.refresh()
            y_cached = fwd().detach().clone()
            Fn.weights.invalidate()  # ground truth: every copy rebuilt from the master weights
            y_fresh = fwd().detach().clone()
            assert float((y_fresh - y0).abs().max()) > 1e-2  # the update was large enough to be seen
            assert torch.equal(y_cached, y_fresh), float((y_cached - y_fresh).abs().max())
        finally:
            Fn.set_compute_dtype(None)
            Fn.weights.invalidate()
            GradSinks.clear()


def test_slots_are_16_byte_aligned_and_sinks_work_behind_a_scalar_parameter():
    """A one-element parameter ahead of a Linear weight (the positional encoding's alpha in the TTS configurations): the weight's slot must
    still start on a 16-byte boundary, because the split-K reduction writes it with 16-byte stores.  The gradient written through the
    sink equals dz^T x."""
    from speechain_amd import functional as Fn
    from speechain_amd.ddp import GradBucketReducer
    from speechain_amd.functional import GradSinks

    torch.manual_seed(0)
    lin = torch.nn.Linear(96, 72).cuda()
    alpha = torch.nn.Parameter(torch.ones(1, device="cuda"))
    odd = torch.nn.Parameter(torch.ones(5, device="cuda"))
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        red = GradBucketReducer([lin.weight, odd, lin.bias, alpha])  # reverse order: alpha, bias, odd, weight
        b = red.buckets[0]
        for v in b["views"]:
            assert v.data_ptr() % 16 == 0
        assert b["flat"].numel() == 4 + 72 + 8 + 72 * 96
        flat_p = red.flatten_parameters()
        for p in (lin.weight, odd, lin.bias, alpha):
            assert p.data_ptr() % 16 == 0
        assert abs(float(flat_p[0].double().sum()) - float(sum(p.detach().double().sum() for p in (lin.weight, odd, lin.bias, alpha)))) < 1e-9  # padding is zero
        x = torch.randn(700, 96, device="cuda")
        red.zero_grad()
        y = Fn.linear(x.bfloat16(), [lin.weight], [lin.bias]) * alpha.bfloat16()
        y.float().sum().backward()
        red.wait()
        assert lin.weight.grad.data_ptr() == b["views"][3].data_ptr()
        ref = torch.ones(700, 72, device="cuda").t() @ x.bfloat16().float()
        assert torch.allclose(lin.weight.grad, ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()
        GradSinks.clear()
