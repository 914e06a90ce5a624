"""sc_decode_linear (csrc/decode_gemm.hip): the linear layers of one decoding step -- act(LayerNorm(x) W^T + b) + residual in one launch --
against fp32 torch on the same bf16-rounded operands: shapes of the ASR decoder's step (512 rows; q|k|v 1536 x 512, out 512 x 512, ffn 2048 x
512 with GELU, 512 x 2048 with a residual, the 5000-wide token head behind the final LayerNorm) and ragged ones (row / column tails, K of one
and of five k-tiles, ReLU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # M, N, K, ln, act, res, bias
    (512, 1536, 512, True, None, False, True),
    (512, 512, 512, False, None, True, True),
    (512, 2048, 512, True, "GELU", False, True),
    (512, 512, 2048, False, None, True, True),
    (512, 5000, 512, True, None, False, True),
    (37, 70, 160, True, "ReLU", True, True),
    (130, 200, 32, False, None, False, False),
    (64, 64, 1024, False, "GELU", True, True),
    (100, 96, 384, True, None, True, False),
]


@pytest.mark.parametrize("M,N,K,ln,act,res,bias", CASES)
def test_decode_linear_vs_torch_fp32(M, N, K, ln, act, res, bias):
    from speechain_amd import functional as Fn
    from speechain_amd._lib import lib, stream

    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g) if bias else None
    gam = 1 + 0.2 * torch.randn(K, generator=g) if ln else None
    bet = 0.1 * torch.randn(K, generator=g) if ln else None
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    a = x.float()
    if ln:
        a = torch.nn.functional.layer_norm(a, (K,), gam, bet, 1e-6).bfloat16().float()  # (the product reads the bf16-rounded normalised rows)
    ref = a @ w.float().t()
    if b is not None:
        ref = ref + b
    if act == "ReLU":
        ref = torch.relu(ref)
    elif act == "GELU":
        ref = torch.nn.functional.gelu(ref)
    if r is not None:
        ref = ref + r.float()
    xc, wc = x.cuda(), w.cuda()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    dev = lambda t: t.cuda() if t is not None else None
    bc, gc, bec, rc = dev(b), dev(gam), dev(bet), dev(r)
    lib.call("sc_decode_linear", Fn._p(xc), K, M, K, Fn._p(wc), K, N, Fn._p(bc), Fn._p(gc), Fn._p(bec), 1e-6, {None: 0, "ReLU": 1, "GELU": 2}[act],
             Fn._p(rc), N if rc is not None else 0, Fn._p(y), N, stream())
    torch.cuda.synchronize()
    e = float((y.float().cpu().double() - ref.double()).norm() / ref.double().norm())
    assert e < 6e-3, e  # (one bf16 rounding of the output, 2^-9 relative, plus the product's rounding)


@pytest.mark.parametrize("M,N,K,act,res,bias,offset", [
    (512, 1536, 512, None, False, True, 0.3), (512, 2048, 512, "GELU", False, True, 0.3), (512, 512, 512, None, False, True, 0.3),
    (37, 70, 160, "ReLU", True, True, 0.3), (100, 96, 384, None, True, False, 0.3), (512, 512, 512, None, False, True, 4.0)])
def test_decode_linear_layernorm_by_statistics(M, N, K, act, res, bias, offset):
    """Fn.decode_linear behind a LayerNorm takes sc_decode_linear_lnstats (raw rows x pre-scaled weight, statistics in the epilogue): against the
    fp32 LayerNorm + product of the same bf16 operands, also with a row mean of several standard deviations (the subtraction's cancellation).
    The pre-scaled weights follow a weight update (rebuilt in place)."""
    from speechain_amd import functional as Fn

    g = torch.Generator().manual_seed(M * 11 + N + K)
    x = (torch.randn(M, K, generator=g) * 1.5 + offset).bfloat16()
    lin = torch.nn.Linear(K, N, bias=bias)
    ln = torch.nn.LayerNorm(K, eps=1e-6)
    with torch.no_grad():
        lin.weight.copy_((torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().float())
        ln.weight.copy_(1 + 0.2 * torch.randn(K, generator=g))
        ln.bias.copy_(0.1 * torch.randn(K, generator=g))
    r = torch.randn(M, N, generator=g).bfloat16() if res else None

    def reference():
        ref = torch.nn.functional.layer_norm(x.float(), (K,), ln.weight.detach().cpu(), ln.bias.detach().cpu(), 1e-6) @ lin.weight.detach().cpu().float().t()
        if bias:
            ref = ref + lin.bias.detach().cpu()
        ref = torch.relu(ref) if act == "ReLU" else torch.nn.functional.gelu(ref) if act == "GELU" else ref
        return ref + r.float() if r is not None else ref

    lin, ln = lin.cuda(), ln.cuda()
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            seen = []
            orig = Fn.lib.call
            Fn.lib.call = lambda name, *a: (seen.append(name), orig(name, *a))[1]
            try:
                y = Fn.decode_linear(x.cuda(), [lin.weight], [lin.bias], ln=ln, activation=act, residual=r.cuda() if r is not None else None)
            finally:
                Fn.lib.call = orig
            assert "sc_decode_linear_lnstats" in seen, seen
            ref = reference()
            e = float((y.float().cpu().double() - ref.double()).norm() / ref.double().norm())
            # bf16 rounding of gamma . W and of the output; with the 4-sigma offset the rows' own bf16 rounding is 4x larger relative to their spread
            assert e < (8e-3 if offset < 1 else 2e-2), e
            lin.weight.mul_(0.5)  # a weight update: the version counter moves, the folded copy is rebuilt in place
            buf = Fn.weights.ln_folded([lin.weight], [lin.bias], ln)[0]
            y2 = Fn.decode_linear(x.cuda(), [lin.weight], [lin.bias], ln=ln, activation=act, residual=r.cuda() if r is not None else None)
            assert Fn.weights.ln_folded([lin.weight], [lin.bias], ln)[0].data_ptr() == buf.data_ptr()
            ref2 = reference()
            e2 = float((y2.float().cpu().double() - ref2.double()).norm() / ref2.double().norm())
            assert e2 < (8e-3 if offset < 1 else 2e-2), e2
    finally:
        Fn.set_compute_dtype(None)


def test_refusals():
    from speechain_amd import functional as Fn
    from speechain_amd._lib import lib, stream

    x = torch.zeros(8, 544, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(8, 544, dtype=torch.bfloat16, device="cuda")
    y = torch.zeros(8, 8, dtype=torch.bfloat16, device="cuda")
    g = torch.ones(544, device="cuda")
    with pytest.raises(RuntimeError, match="whole row on chip"):
        lib.call("sc_decode_linear", Fn._p(x), 544, 8, 544, Fn._p(w), 544, 8, None, Fn._p(g), Fn._p(g), 1e-6, 0, None, 0, Fn._p(y), 8, stream())
    with pytest.raises(RuntimeError, match="multiple of 32"):
        lib.call("sc_decode_linear", Fn._p(x), 544, 8, 40, Fn._p(w), 544, 8, None, None, None, 0.0, 0, None, 0, Fn._p(y), 8, stream())
