"""GPU: the batched GEMM entry point (sc_gemm) against torch fp32/fp64 matmul -- all operand layouts, both dtypes,
ragged shapes, batch strides and every epilogue mode.  fp32 mode must be exact-f32 accurate; bf16 within bf16 rounding."""
import math

import pytest
import torch

from speechain_amd import functional as Fn

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g)).to(dtype).cuda()


def _tol(dtype, K):
    return dict(atol=1e-4 * math.sqrt(K / 64), rtol=1e-5) if dtype == torch.float32 else dict(atol=0.06 * math.sqrt(K / 64), rtol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 75, 100), (33, 257, 31), (7968 // 8, 512, 512), (5, 5000, 64)])
def test_layouts_and_ragged_shapes(dtype, ta, tb, M, N, K):
    A = _mk((K, M) if ta else (M, K), dtype, 1)
    B = _mk((K, N) if tb else (N, K), dtype, 2)
    C = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    Fn.gemm(Fn.dcode(A), Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, A.stride(0), B.stride(0), N, ta=ta, tb=tb)
    opA = A.double().t() if ta else A.double()
    opB = B.double().t() if tb else B.double()
    ref = opA @ opB.t()
    assert torch.allclose(C.double(), ref, **_tol(dtype, K)), float((C.double() - ref).abs().max())


@pytest.mark.parametrize("M,N,K,res", [(2016, 512, 512, True), (2016, 512, 2048, True), (1000, 512, 512, False), (2016, 320, 1536, False)])
def test_few_tile_products_on_64x64_tiles(M, N, K, res):
    """The decoder-sized products (at most 80 tiles of 128 x 128) go to the two-wave 64 x 64 LDS-DMA tiles: plain and with the
    bias + residual epilogue, ragged in M and N."""
    A, B = _mk((M, K), torch.bfloat16, 11), _mk((N, K), torch.bfloat16, 12)
    bias = _mk((N,), torch.float32, 13)
    R = _mk((M, N), torch.bfloat16, 14)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    Fn.gemm(Fn.BF16, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, bias=bias, aux=Fn._p(R) if res else None, ldaux=N, aux_mode=1 if res else 0)
    from speechain_amd._lib import lib
    assert lib.load().sc_gemm_kernel_name(lib.load().sc_gemm_last_kernel()).decode() == "gemm_bf16_nt_dma_kernel<2, 1, 1, 1, false>"
    ref = A.double() @ B.double().t() + bias.double() + (R.double() if res else 0)
    assert torch.allclose(C.double(), ref, **_tol(torch.bfloat16, K)), float((C.double() - ref).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_head_strides(dtype):
    """Q K^T on head-interleaved (B, T, 3d) buffers, as the attention core issues it."""
    Bn, T, H, dh = 3, 37, 4, 16
    d = H * dh
    qkv = _mk((Bn, T, 3 * d), dtype, 3)
    ld = 40
    S = torch.zeros(Bn * H, T, ld, dtype=dtype, device="cuda")
    Fn.gemm(Fn.dcode(qkv), Fn._p(qkv, 0), Fn._p(qkv, d), Fn._p(S), T, T, dh, 3 * d, 3 * d, ld, nb=(Bn, H), sA=(T * 3 * d, dh),
            sB=(T * 3 * d, dh), sC=(H * T * ld, T * ld), alpha=0.25)
    q = qkv[..., :d].view(Bn, T, H, dh).transpose(1, 2).double()
    k = qkv[..., d:2 * d].view(Bn, T, H, dh).transpose(1, 2).double()
    ref = (q @ k.transpose(2, 3)) * 0.25
    got = S.view(Bn, H, T, ld)[..., :T].double()
    assert torch.allclose(got, ref, **_tol(dtype, dh))
    assert float(S.view(Bn, H, T, ld)[..., T:].abs().max()) == 0.0  # padding columns untouched


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_epilogues(dtype, act):
    M, N, K = 150, 200, 96
    A, B = _mk((M, K), dtype, 4), _mk((N, K), dtype, 5)
    bias = _mk((N,), torch.float32, 6)
    res = _mk((M, N), dtype, 7)
    C = torch.empty(M, N, dtype=dtype, device="cuda")
    pre = torch.empty(M, N, dtype=dtype, device="cuda")
    Fn.gemm(Fn.dcode(A), Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, bias=bias, act=act, pre_out=Fn._p(pre), aux=Fn._p(res),
            ldaux=N, aux_mode=1, alpha=0.5)
    z = 0.5 * (A.double() @ B.double().t()) + bias.double()
    f = [lambda t: t, torch.relu, lambda t: torch.nn.functional.gelu(t)][act]
    ref = f(z) + res.double()
    tol = _tol(dtype, K)
    assert torch.allclose(pre.double(), z, **tol)
    assert torch.allclose(C.double(), ref, **tol)
    # activation-gradient epilogue: out = (A B^T) * act'(aux)
    G = torch.empty(M, N, dtype=dtype, device="cuda")
    Fn.gemm(Fn.dcode(A), Fn._p(A), Fn._p(B), Fn._p(G), M, N, K, K, K, N, act=act, aux=Fn._p(pre), ldaux=N, aux_mode=2)
    zz = pre.double().requires_grad_()
    (gact,) = torch.autograd.grad(f(zz).sum(), zz)
    ref_g = (A.double() @ B.double().t()) * gact
    assert torch.allclose(G.double(), ref_g, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_f32_output_and_accumulate(dtype):
    M, N, K = 64, 96, 333
    A, B = _mk((K, M), dtype, 8), _mk((K, N), dtype, 9)
    C = torch.ones(M, N, dtype=torch.float32, device="cuda")
    Fn.gemm(Fn.dcode(A), Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, M, N, N, ta=1, tb=1, out_f32=True, accumulate=True)
    ref = 1.0 + A.double().t() @ B.double()
    assert torch.allclose(C.double(), ref, **_tol(dtype, K))


def test_dropout_epilogue_is_reproducible_and_unbiased():
    M, N, K = 256, 512, 64
    A, B = _mk((M, K), torch.float32, 10), _mk((N, K), torch.float32, 11)
    seed_ptr, _ = Fn.DropoutState.next(0, A.device)  # address of the device seed cell
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device="cuda")
        Fn.gemm(0, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, drop=(0.25, seed_ptr, 77))
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    ref = A @ B.t()
    kept = outs[0] != 0
    frac = 1.0 - float(kept.float().mean())
    assert abs(frac - 0.25) < 0.01
    assert torch.allclose(outs[0][kept], ref[kept] / 0.75, atol=1e-4, rtol=1e-5)
    # the stand-alone dropout kernel regenerates the same mask from (seed, offset)
    y = Fn._dropout_apply(ref.contiguous(), 0.25, seed_ptr, 77)
    assert torch.equal(y != 0, kept)
    # a new step (advance) re-seeds the cell: same offsets, different mask
    Fn.DropoutState.advance()
    C2 = torch.empty(M, N, device="cuda")
    Fn.gemm(0, Fn._p(A), Fn._p(B), Fn._p(C2), M, N, K, K, K, N, drop=(0.25, seed_ptr, 77))
    assert not torch.equal(C2 != 0, kept)
    assert abs(1.0 - float((C2 != 0).float().mean()) - 0.25) < 0.01


@pytest.mark.parametrize("M,N,K", [(7968, 2048, 512), (2016, 512, 512), (999, 130, 70), (300, 5000, 512)])
def test_wgrad_split_k(M, N, K):
    """dW = dz^T x through the split-K path (bf16) and the plain path (f32) against fp64."""
    for dtype in (torch.bfloat16, torch.float32):
        dz, x = _mk((M, N), dtype, 20), _mk((M, K), dtype, 21)
        dW = Fn.wgrad(Fn.dcode(dz), dz, x)
        ref = dz.double().t() @ x.double()
        assert dW.dtype == torch.float32 and dW.shape == (N, K)
        tol = dict(atol=2e-3 * math.sqrt(M / 64), rtol=1e-4) if dtype == torch.float32 else dict(atol=0.05 * math.sqrt(M / 64), rtol=2e-2)
        assert torch.allclose(dW.double(), ref, **tol), float((dW.double() - ref).abs().max())


@pytest.mark.parametrize("M,N,K", [(7968, 2048, 512), (7968, 512, 2048), (2016, 1536, 512), (999, 130, 70), (300, 5000, 512), (64, 96, 40)])
def test_wgrad_bias_gradient_from_the_same_launch(M, N, K):
    """db = column sums of dz, taken by the split-K wgrad kernel from the dz tiles it streams (bf16) or by the separate
    column-sum launch (f32 / no split): against the fp64 sum of the same values.  Only the order of an f32 summation
    differs, so the bound is M * 2^-23 of the column's absolute sum; dW must be unchanged by asking for db."""
    for dtype in (torch.bfloat16, torch.float32):
        dz, x = _mk((M, N), dtype, 30), _mk((M, K), dtype, 31)
        dW, db = Fn.wgrad(Fn.dcode(dz), dz, x, bias=True)
        assert db.dtype == torch.float32 and db.shape == (N,)
        ref = dz.double().sum(0)
        bound = dz.double().abs().sum(0) * M * 2.0 ** -23 + 1e-6
        err = (db.double() - ref).abs()
        assert bool((err <= bound).all()), float((err / bound).max())
        assert torch.equal(dW, Fn.wgrad(Fn.dcode(dz), dz, x))
        # deterministic: same launch twice, same bits
        assert torch.equal(db, Fn.wgrad(Fn.dcode(dz), dz, x, bias=True)[1])


@pytest.mark.parametrize("T,N,K,splitk", [(1024, 128, 128, 32), (1024, 256, 136, 16), (2048, 5000, 520, 4), (7968, 1536, 512, 8),
                                          (640, 8, 8, 2), (992, 264, 72, 31)])
def test_tn_split_k_dma_kernel_through_the_c_abi(monkeypatch, T, N, K, splitk):
    """dW (N, K) = dz^T x and db = column sums of dz straight through sc_gemm_splitk on the LDS-DMA TN kernel (tokens % 32 == 0,
    N % 8 == K % 8 == 0): one or two K tiles per slice, partial 128-column tiles, tiny outputs -- against fp64 of the same
    bf16 values, and bit-identical to the register-staged kernel's slabs order-independent part (same reduce kernel)."""
    from speechain_amd._lib import lib, stream

    monkeypatch.setenv("SC_GEMM_TN_DMA", "1")  # the kernel is opt-in (it is slower than the register-staged one: DESIGN.md section 6)
    dz, x = _mk((T, N), torch.bfloat16, 40), _mk((T, K), torch.bfloat16, 41)
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    ws = torch.empty(lib.value("sc_gemm_splitk_workspace", N, K, splitk), dtype=torch.uint8, device="cuda")
    lib.call("sc_gemm_splitk", 1, 1, Fn._p(dz), Fn._p(x), Fn._p(dW), N, K, T, N, K, 1.0, splitk, 0, Fn._p(db), Fn._p(ws), ws.numel(),
             stream())
    ref = dz.double().t() @ x.double()
    assert torch.allclose(dW.double(), ref, atol=0.05 * math.sqrt(T / 64), rtol=2e-2), float((dW.double() - ref).abs().max())
    rb = dz.double().sum(0)
    bound = dz.double().abs().sum(0) * T * 2.0 ** -23 + 1e-6
    assert bool(((db.double() - rb).abs() <= bound).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("act", [1, 2], ids=["relu", "gelu"])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_pre_gradient_output_and_multiply_mode(dtype, act, p):
    """act | 0x100: the second output is act'(pre) * dropout factor (with the SAME mask as the main output); aux_mode 3 multiplies a
    product by such a tensor -- the pair the feed-forward layer uses so that its data gradient needs no erf / exp / mask hash."""
    M, N, K = 300, 264, 96  # ragged: partial tiles on both sides
    A, B = (_mk((M, K), torch.float32, 21) * 0.1).to(dtype), _mk((N, K), dtype, 22)  # pre-activations of order one (a saturated GELU is 0 too)
    bias = _mk((N,), torch.float32, 23)
    h = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    g = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    seed, off = Fn.DropoutState.next(M * N, A.device) if p > 0 else (0, 0)
    Fn.gemm(Fn.dcode(A), Fn._p(A), Fn._p(B), Fn._p(h), M, N, K, K, K, N, bias=bias, act=act | 0x100, pre_out=Fn._p(g), drop=(p, seed, off))
    pre = A.double() @ B.double().t() + bias.double()
    fact = torch.nn.functional.relu(pre) if act == 1 else torch.nn.functional.gelu(pre)
    gprime = (pre > 0).double() if act == 1 else (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5)
    keep = 1.0 / (1.0 - p)
    if p > 0:  # the same elements are dropped in both outputs
        live = (fact.abs() > 1e-2).to(h.device)  # where the activation itself is clearly non-zero, a zero output means "dropped"
        dropped = (h == 0) & live
        assert torch.equal(dropped, (g == 0) & live)
        frac = float(dropped.float().sum() / live.float().sum())
        assert abs(frac - p) < 0.02, frac
        kept = ~(h == 0)
    else:
        kept = torch.ones_like(h, dtype=torch.bool)
    t = dict(atol=2e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=6e-2, rtol=2e-2)
    if act == 1:  # at ReLU's kink a rounding of `pre` decides the gate: compare away from it
        kept = kept & (pre.abs() > 1e-3).to(kept.device)
    assert torch.allclose(h.double()[kept], (fact.to(h.device) * keep)[kept], **t)
    assert torch.allclose(g.double()[kept], (gprime.to(h.device) * keep)[kept], **t)
    # aux_mode 3: C = (A2 B2^T) * g
    A2, B2 = _mk((M, K), dtype, 24), _mk((N, K), dtype, 25)
    C = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    Fn.gemm(Fn.dcode(A2), Fn._p(A2), Fn._p(B2), Fn._p(C), M, N, K, K, K, N, aux=Fn._p(g), ldaux=N, aux_mode=3)
    ref = (A2.double() @ B2.double().t()) * g.double()
    assert torch.allclose(C.double(), ref, **_tol(dtype, K)), float((C.double() - ref).abs().max())


def _grouped(problems, tile):
    """problems: list of (dz, x, want_db).  Returns [(dW, db or None)] computed by ONE sc_wgrad_grouped call."""
    import ctypes

    from speechain_amd._lib import lib, stream

    n = len(problems)
    outs = []
    for dz, x, want_db, *_ in problems:
        dW = torch.full((dz.size(1), x.size(1)), float("nan"), device="cuda")
        db = torch.full((dz.size(1),), float("nan"), device="cuda") if want_db else None
        outs.append((dW, db))
    arr = lambda ct, vals: (ct * n)(*vals)
    n_out, n_in = arr(ctypes.c_int, [p[0].size(1) for p in problems]), arr(ctypes.c_int, [p[1].size(1) for p in problems])
    ws = torch.empty(lib.value("sc_wgrad_grouped_workspace", n, n_out, n_in), dtype=torch.uint8, device="cuda")
    lib.call("sc_wgrad_grouped", n, arr(ctypes.c_void_p, [p[0].data_ptr() for p in problems]),
             arr(ctypes.c_void_p, [p[1].data_ptr() for p in problems]), arr(ctypes.c_void_p, [o[0].data_ptr() for o in outs]),
             arr(ctypes.c_void_p, [o[1].data_ptr() if o[1] is not None else 0 for o in outs]),
             n_out, n_in, arr(ctypes.c_int, [p[0].size(0) for p in problems]), arr(ctypes.c_longlong, [p[0].stride(0) for p in problems]),
             arr(ctypes.c_longlong, [p[1].stride(0) for p in problems]), arr(ctypes.c_int, [p[3] if len(p) > 3 else 1 for p in problems]), tile,
             ws.data_ptr(), ws.numel(), stream())
    return outs


@pytest.mark.parametrize("tile", [128, 256, 0])
def test_grouped_weight_gradients(tile):
    """sc_wgrad_grouped: a mixed list of layers in one call -- ragged output sizes (tiles hanging over both edges), different token
    counts (1 .. 40 K tiles: prologue-only, tail-only and steady-state loops), strided operands (q|k|v side by side), with and
    without the bias gradient -- against float64 dz^T x and column sums.  bf16 products accumulate in f32 in token order."""
    shapes = [(512, 512, 32 * 31), (1536, 512, 992), (40, 72, 32), (5000 // 8 * 8, 512, 64), (264, 136, 96), (2048, 512, 1280),
              (128, 128, 128), (8, 8, 160), (520, 2048, 320)]
    problems = []
    for i, (n_out, n_in, tok) in enumerate(shapes):
        wide = _mk((tok, n_out + 24), torch.bfloat16, 100 + i)
        dz = wide[:, 16:16 + n_out] if i % 3 == 1 else _mk((tok, n_out), torch.bfloat16, 100 + i)  # some with a row stride > n_out
        x = _mk((tok, n_in), torch.bfloat16, 200 + i)
        problems.append((dz, x, i % 2 == 0))
    # more units than one table-writer launch carries (56)
    for i in range(38):
        problems.append((_mk((64, 24), torch.bfloat16, 300 + i), _mk((64, 16), torch.bfloat16, 400 + i), i % 4 == 0))
    # two im2col products of 'same' Conv1d layers (3 and 5 taps): dW comes back in the nn.Conv1d weight layout (out, in, k)
    problems.append((_mk((160, 48), torch.bfloat16, 501), _mk((160, 3 * 40), torch.bfloat16, 502), True, 3))
    problems.append((_mk((96, 264), torch.bfloat16, 503), _mk((96, 5 * 16), torch.bfloat16, 504), False, 5))
    outs = _grouped(problems, tile)
    for (dz, x, want_db, *ck), (dW, db) in zip(problems, outs):
        ref = dz.double().t() @ x.double()
        if ck:  # GEMM column tap * C + c  ->  element (c, tap) of the Conv1d weight
            k = ck[0]
            ref = ref.view(ref.size(0), k, -1).permute(0, 2, 1).reshape(ref.size(0), -1)
        tok = dz.size(0)
        assert torch.allclose(dW.double(), ref, atol=2e-3 * math.sqrt(tok), rtol=1e-3), (dz.shape, x.shape, float((dW.double() - ref).abs().max()))
        if want_db:
            err = (db.double() - dz.double().sum(0)).abs()
            assert torch.allclose(db.double(), dz.double().sum(0), atol=1e-3 * math.sqrt(tok), rtol=1e-4), \
                (dz.shape, float(err.max()), (err > 1e-2 * math.sqrt(tok)).nonzero().flatten().tolist()[:16])


def test_grouped_weight_gradients_refuse_unsupported_shapes():
    from speechain_amd._lib import lib

    assert lib.value("sc_wgrad_grouped_supported", 512, 512, 7968, 512, 512) == 1
    assert lib.value("sc_wgrad_grouped_supported", 512, 512, 100, 512, 512) == 0   # tokens not a multiple of 32
    assert lib.value("sc_wgrad_grouped_supported", 31, 512, 64, 31, 512) == 0      # odd feature count
    assert lib.value("sc_wgrad_grouped_supported", 512, 512, 64, 500, 512) == 0    # row stride below the width
    with pytest.raises(RuntimeError, match="not supported"):
        _grouped([(_mk((100, 64), torch.bfloat16, 1), _mk((100, 64), torch.bfloat16, 2), False)], 0)


@pytest.mark.parametrize("M,N,K", [(300, 200, 72), (996, 512, 5000), (2016, 512, 200), (129, 264, 104), (7968, 520, 88)])
def test_partial_last_k_tile_on_the_dma_kernels(M, N, K):
    """K a multiple of 8 but not of 32 (the 5000-wide contraction of the vocabulary heads' data gradients) stays on the LDS-DMA kernels:
    out-of-range source chunks of the last K tile are redirected, their fragments zeroed.  Operands carry NaN / Inf right behind the
    valid columns of a wider buffer, so anything read from beyond K and not zeroed would poison the result."""
    from speechain_amd._lib import lib

    Aw = torch.full((M, K + 40), float("nan"), dtype=torch.bfloat16, device="cuda")
    Bw = torch.full((N, K + 40), float("inf"), dtype=torch.bfloat16, device="cuda")
    Aw[:, :K] = _mk((M, K), torch.bfloat16, 21)
    Bw[:, :K] = _mk((N, K), torch.bfloat16, 22)
    A, B = Aw[:, :K], Bw[:, :K]
    bias = _mk((N,), torch.float32, 23)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    Fn.gemm(Fn.BF16, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, A.stride(0), B.stride(0), N, bias=bias)
    assert "nt_dma" in lib.load().sc_gemm_kernel_name(lib.load().sc_gemm_last_kernel()).decode()
    ref = A.double() @ B.double().t() + bias.double()
    assert torch.isfinite(C.float()).all()
    assert torch.allclose(C.double(), ref, **_tol(torch.bfloat16, K)), float((C.double() - ref).abs().max())
