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
