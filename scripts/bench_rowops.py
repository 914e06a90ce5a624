"""Back-to-back launch times of the row kernels at the train-step shapes (7968 x 512 LayerNorm rows, 7968 x 5000 CTC logits):
python scripts/bench_rowops.py  (env SC_LN_BWD_ROWS / SC_LN_BWD_CAP as in csrc/rowops.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd._lib import lib, ptr as _p, stream  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * n)


def main():
    dev = torch.device("cuda:0")
    R, D = 7968, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(R, D, device=dev, generator=g).bfloat16()
    dy = torch.randn(R, D, device=dev, generator=g).bfloat16()
    dres = torch.randn(R, D, device=dev, generator=g).bfloat16()
    gamma = torch.rand(D, device=dev) + 0.5
    beta = torch.randn(D, device=dev)
    y, dx, dz = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    ws = torch.empty(lib.value("sc_layernorm_bwd_workspace", R, D), dtype=torch.uint8, device=dev)
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    t = timed(lambda: lib.call("sc_layernorm_fwd", 1, _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), R, D, 1e-5, stream()))
    print(f"layernorm fwd  {R} x {D}: {t:6.2f} us  ({2 * R * D * 2 / t / 1e6:.2f} TB/s algorithmic)")
    t = timed(lambda: lib.call("sc_layernorm_bwd", 1, _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dg), _p(db), R, D, 0, _p(ws), ws.numel(), 1, stream()))
    print(f"layernorm bwd  plain       : {t:6.2f} us  ({3 * R * D * 2 / t / 1e6:.2f} TB/s)")
    t = timed(lambda: lib.call("sc_layernorm_bwd", 1, _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dg), _p(db), R, D, _p(dres), _p(ws), ws.numel(), 1, stream()))
    print(f"layernorm bwd  + residual  : {t:6.2f} us  ({4 * R * D * 2 / t / 1e6:.2f} TB/s)")
    t = timed(lambda: lib.call("sc_layernorm_bwd_dz", 1, _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dg), _p(db), R, D, _p(dres), _p(ws), ws.numel(), 1,
                               _p(dz), 0.1, _p(seed), 0, stream()))
    print(f"layernorm bwd  + res + dz  : {t:6.2f} us  ({5 * R * D * 2 / t / 1e6:.2f} TB/s)")
    # CTC head of the step: 32 utterances x 249 frames x 5000 classes, 40-token targets
    B, T, V, L = 32, 249, 5000, 40
    logits = torch.randn(B, T, V, device=dev, generator=g).bfloat16()
    lse = torch.empty(B * T, device=dev)
    t = timed(lambda: lib.call("sc_row_lse", 1, _p(logits), B * T, V, V, _p(lse), stream()))
    print(f"row_lse {B * T} x {V}: {t:6.2f} us  ({B * T * V * 2 / t / 1e6:.2f} TB/s)")
    in_len = torch.full((B,), T, dtype=torch.int64, device=dev)
    tgt = torch.randint(1, V, (B, L), device=dev, generator=g)
    tgt_len = torch.full((B,), L, dtype=torch.int64, device=dev)
    S = 2 * L + 1
    alpha, betaw = torch.empty(B, T, S, device=dev), torch.empty(B, T, S, device=dev)
    nll, gs = torch.empty(B, device=dev), torch.full((B,), 1.0 / B, device=dev)
    grad = torch.empty_like(logits)
    args = (1, _p(logits), _p(lse), _p(in_len), _p(tgt), tgt.stride(0), _p(tgt_len), B, T, V, S, 0)
    t = timed(lambda: lib.call("sc_ctc_fwd", *args, _p(alpha), _p(nll), stream()), 10)
    print(f"ctc alpha             : {t:6.2f} us")
    t = timed(lambda: lib.call("sc_ctc_fwd_train", *args, _p(alpha), _p(betaw), _p(nll), stream()), 10)
    print(f"ctc alpha + beta      : {t:6.2f} us")
    t = timed(lambda: lib.call("sc_ctc_grad", *args, _p(alpha), _p(betaw), _p(nll), _p(gs), _p(grad), stream()), 10)
    print(f"ctc grad              : {t:6.2f} us  ({2 * B * T * V * 2 / t / 1e6:.2f} TB/s)")


if __name__ == "__main__":
    main()
