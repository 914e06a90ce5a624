"""NT GEMM launches on the shapes of the train step and of a decoding step, captured in a hipGraph and replayed back to back, with the kernel
each one dispatched to and its error against torch.  Written for the deep-ring experiment of round 2 (negative result
(e): the SC_GEMM_DEEP knob it compared lives in the history of csrc/gemm.hip, not in the tree); kept as the per-shape table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd._lib import lib  # noqa: E402


def timeit(fn, n=20):
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


def run(M, N, K, **kw):
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    extra = {}
    if kw.get("bias"):
        extra["bias"] = torch.randn(N, device="cuda")
    if kw.get("res"):
        res = torch.randn(M, N, device="cuda").bfloat16()
        extra.update(aux=Fn._p(res), ldaux=N, aux_mode=1)
    us = timeit(lambda: Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, **extra))
    name = lib.load().sc_gemm_kernel_name(lib.load().sc_gemm_last_kernel()).decode()
    ref = (A.float() @ B.float().t())
    if kw.get("bias"):
        ref = ref + extra["bias"]
    if kw.get("res"):
        ref = ref + res.float()
    err = float((C.float() - ref).abs().max() / ref.abs().max())
    print(f"M={M:5d} N={N:5d} K={K:5d} {str(kw):24s} {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s  {name}  relerr {err:.1e}", flush=True)


for shape in ((7968, 512, 512), (7968, 512, 1536), (7968, 512, 2048), (7968, 512, 1024), (7968, 1024, 512), (2016, 512, 512), (2016, 512, 2048),
              (2016, 1536, 512), (2016, 2048, 512), (512, 512, 512), (512, 1536, 512), (512, 2048, 512), (512, 512, 2048), (512, 5000, 512),
              (7968, 512, 96), (7968, 512, 160)):
    run(*shape)
run(7968, 512, 2048, bias=1, res=1)
run(512, 512, 512, bias=1, res=1)
