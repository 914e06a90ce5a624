"""Fused attention for head sizes other than 64 (csrc/attention_dh.hip: sc_attention_dh_fwd / _bwd) -- the reference's
MultiHeadedAttention is head-size agnostic (speechain/module/transformer/attention.py:53-65,88-106,131); FastSpeech2 uses 2 heads of 192.

Through the C ABI against an fp32 torch evaluation of matmul -> masked_fill(-inf) -> softmax -> matmul on the same bf16-rounded operands:
output, saved logsumexp, dQ, dK, dV (relative Frobenius 1e-2: the probabilities and dS are rounded to bf16 on their way into the second
products), for self- and cross-attention shapes, key-padding masks, the causal flag, ragged tile edges and head sizes 8 .. 256.  With
dropout the fused kernels must regenerate exactly the masks of the unfused path (same index scheme): module-level comparison under one
seed, outputs and every gradient."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, mask, causal, scale):
    """q (B, H, Tq, dh) ... fp32 autograd reference; mask (B, Tk) bool or None."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    if causal:
        Tq, Tk = s.shape[-2:]
        s = s.masked_fill(~torch.tril(torch.ones(Tq, Tk, dtype=torch.bool)), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v), lse


CASES = [  # B, H, dh, Tq, Tk, masked, causal
    (2, 2, 192, 640, 640, True, False),   # FastSpeech2 decoder
    (2, 2, 192, 128, 128, True, False),   # FastSpeech2 encoder
    (3, 2, 192, 70, 203, True, False),    # cross-attention shape, ragged tiles
    (2, 3, 128, 200, 200, False, True),   # causal
    (2, 4, 40, 130, 130, True, True),     # head size below one chunk, not a multiple of 16
    (1, 2, 256, 96, 300, True, False),    # the widest head
    (2, 2, 96, 64, 64, False, False),
    (2, 5, 8, 33, 47, True, False),       # the smallest head
]


@pytest.mark.parametrize("B,H,dh,Tq,Tk,masked,causal", CASES)
def test_fused_attention_any_head_size_vs_torch_fp32(B, H, dh, Tq, Tk, masked, causal):
    from speechain_amd import functional as Fn
    from speechain_amd._lib import lib, stream

    gen = torch.Generator().manual_seed(dh * 1000 + Tq)
    d = H * dh
    q = (torch.randn(B, Tq, d, generator=gen) * 0.8).bfloat16()
    k = (torch.randn(B, Tk, d, generator=gen) * 0.8).bfloat16()
    v = torch.randn(B, Tk, d, generator=gen).bfloat16()
    do = torch.randn(B, Tq, d, generator=gen).bfloat16()
    mask = None
    if masked:
        lens = torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=gen)
        lens[0] = Tk
        mask = torch.arange(Tk)[None, :] < lens[:, None]
    scale = 1.0 / math.sqrt(d)  # (the reference scales by 1 / sqrt(d_model) unless scale_dp_by_head)
    heads = lambda x, T: x.float().view(B, T, H, dh).transpose(1, 2).clone().requires_grad_()
    qr, kr, vr = heads(q, Tq), heads(k, Tk), heads(v, Tk)
    out_r, lse_r = _ref(qr, kr, vr, mask, causal, scale)
    (out_r * heads(do, Tq).detach()).sum().backward()
    back = lambda x, T: x.transpose(1, 2).reshape(B, T, d)

    qc, kc, vc, doc = q.cuda(), k.cuda(), v.cuda(), do.cuda()
    out = torch.empty_like(qc)
    lse = torch.empty(B * H, Tq, dtype=torch.float32, device="cuda")
    m8 = mask.to(torch.uint8).cuda().contiguous() if mask is not None else None
    lib.call("sc_attention_dh_fwd", Fn._p(qc), Fn._p(kc), Fn._p(vc), Fn._p(out), Fn._p(lse), B, H, Tq, Tk, dh, qc.stride(0), qc.stride(1),
             kc.stride(0), kc.stride(1), out.stride(0), out.stride(1), Fn._p(m8), Tk if m8 is not None else 0, 1 if causal else 0, scale, 0.0,
             None, 0, stream())
    dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
    ws_bytes = lib.value("sc_attention_dh_bwd_workspace", B, H, Tq)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device="cuda")
    lib.call("sc_attention_dh_bwd", Fn._p(qc), Fn._p(kc), Fn._p(vc), Fn._p(out), Fn._p(doc), Fn._p(lse), Fn._p(dq), Fn._p(dk), Fn._p(dv), B, H,
             Tq, Tk, dh, qc.stride(0), qc.stride(1), kc.stride(0), kc.stride(1), out.stride(0), out.stride(1), dq.stride(0), dq.stride(1),
             dk.stride(0), dk.stride(1), Fn._p(m8), Tk if m8 is not None else 0, 1 if causal else 0, scale, 0.0, None, 0, Fn._p(ws), ws_bytes,
             stream())
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(out.float().cpu(), back(out_r.detach(), Tq)) < 1e-2
    fin = torch.isfinite(lse_r)
    assert torch.equal(torch.isfinite(lse.cpu()).view(B, H, Tq), fin)
    assert float((lse.cpu().view(B, H, Tq)[fin] - lse_r[fin]).abs().max()) < 2e-2
    assert rel(dq.float().cpu(), back(qr.grad, Tq)) < 1.5e-2, rel(dq.float().cpu(), back(qr.grad, Tq))
    assert rel(dk.float().cpu(), back(kr.grad, Tk)) < 1.5e-2, rel(dk.float().cpu(), back(kr.grad, Tk))
    assert rel(dv.float().cpu(), back(vr.grad, Tk)) < 1.5e-2, rel(dv.float().cpu(), back(vr.grad, Tk))
    if mask is not None:  # padded keys receive no gradient at all
        dead = ~mask
        assert float(dk.float().cpu()[dead].abs().max() if dead.any() else 0.0) == 0.0
        assert float(dv.float().cpu()[dead].abs().max() if dead.any() else 0.0) == 0.0


@pytest.mark.parametrize("heads,d_model,T,drop", [(2, 384, 160, 0.2), (2, 384, 160, 0.0), (4, 144 * 2, 75, 0.1), (3, 120, 90, 0.3)])
def test_module_takes_the_fused_kernels_and_matches_the_unfused_path(heads, d_model, T, drop):
    """MultiHeadedAttention with head sizes 192 / 72 / 40, bf16, attention maps not requested: the fused kernels run (recorded at the C-ABI
    call level) and reproduce the unfused GEMM + softmax path -- same dropout masks under the same seed -- in the output and in the gradients of
    the input and of every projection."""
    from speechain_amd import functional as Fn
    from speechain_amd._lib import lib
    from speechain_amd.module.transformer.attention import MultiHeadedAttention

    Fn.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(11)
        m = MultiHeadedAttention(num_heads=heads, d_model=d_model, dropout=drop).cuda().train()
        gen = torch.Generator().manual_seed(12)
        x = torch.randn(3, T, d_model, generator=gen).cuda()
        lens = torch.tensor([T, T - 17, T // 2])
        mask = (torch.arange(T)[None, :] < lens[:, None])[:, None, :].cuda()
        w = torch.randn(3, T, d_model, generator=gen).cuda()
        results, called = [], []
        orig = lib.call

        def spy(name, *a):
            called.append(name)
            return orig(name, *a)

        lib.call = spy
        try:
            for probs in (True, False):
                Fn.set_attention_probs(probs)
                Fn.DropoutState.reset(123)
                Fn.weights.invalidate()
                called.clear()
                xin = x.clone().requires_grad_()
                y, p = m(xin, xin, xin, mask)
                (y.float() * w).sum().backward()
                torch.cuda.synchronize()
                results.append((y.detach().float().cpu(), xin.grad.float().cpu(), {k: v.grad.float().cpu().clone() for k, v in m.named_parameters()}))
                for prm in m.parameters():
                    prm.grad = None
                if probs:
                    assert p is not None and "sc_attention_dh_fwd" not in called
                else:
                    assert p is None and "sc_attention_dh_fwd" in called and "sc_attention_dh_bwd" in called and "sc_softmax_fwd" not in called
        finally:
            lib.call = orig
        rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        (y0, g0, p0), (y1, g1, p1) = results
        assert rel(y1, y0) < 1.5e-2, rel(y1, y0)
        assert rel(g1, g0) < 2e-2, rel(g1, g0)
        for k in p0:
            if "k_layer.bias" in k:  # (no gradient in exact arithmetic)
                continue
            assert rel(p1[k], p0[k]) < 2e-2, (k, rel(p1[k], p0[k]))
    finally:
        Fn.set_attention_probs(True)
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()
