"""Per-shape timing of the decoding step's linear layers: sc_decode_linear (with / without the folded LayerNorm; tile size forced through
SC_DECODE_TR in separate processes) against the LayerNorm launch + general GEMM kernel it replaces.  Launches captured back to back in a hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402


def timeit(fn, n=40):
    for _ in range(3):
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
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


Fn.set_compute_dtype(torch.bfloat16)
M = 512
print("SC_DECODE_TR =", os.environ.get("SC_DECODE_TR", "auto"))
with torch.no_grad():
    for (N, K, ln, act, res, name) in [(1536, 512, True, None, False, "q|k|v"), (512, 512, False, None, True, "out"), (512, 512, True, None, False, "q"),
                                       (2048, 512, True, "GELU", False, "ffn-in"), (512, 2048, False, None, True, "ffn-out"), (5000, 512, True, None, False, "head")]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        lin = torch.nn.Linear(K, N).cuda()
        norm = torch.nn.LayerNorm(K, eps=1e-6).cuda() if ln else None
        r = torch.randn(M, N, device="cuda").bfloat16() if res else None
        new = timeit(lambda: Fn.decode_linear(x, [lin.weight], [lin.bias], ln=norm, activation=act, residual=r))
        noln = timeit(lambda: Fn.decode_linear(x, [lin.weight], [lin.bias], ln=None, activation=act, residual=r))

        def old():
            xi = Fn.layer_norm(x, norm.weight, norm.bias, norm.eps) if ln else x
            if act:
                return Fn.linear_act(xi, lin.weight, lin.bias, act)
            return Fn.linear(xi, [lin.weight], [lin.bias], residual=r)

        print(f"{name:8s} N={N:5d} K={K:5d} ln={ln!s:5s}: decode_linear {new:6.2f} us | without the LayerNorm {noln:6.2f} us | LayerNorm launch + general GEMM {timeit(old):6.2f} us", flush=True)
