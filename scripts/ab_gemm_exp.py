"""A/B of the experiment switches of the 8-wave 128 x 128 NT GEMM kernel (csrc/gemm.hip: sc_gemm_experiment) inside ONE process, interleaved
rounds, launches captured back to back in a hipGraph: 0 base, 1 one workgroup per CU, 2 the older four waves issue every LDS-DMA,
4 static priority for the younger four waves, and combinations."""
import ctypes
import os
import statistics as st
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd._lib import lib  # noqa: E402

exp = lib.load().sc_gemm_experiment
exp.argtypes = [ctypes.c_int]
exp.restype = None
seed_ptr, _ = Fn.DropoutState.next(0, torch.device("cuda", 0))


def graph_of(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    return g, n


def time_graph(g, n, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


VARIANTS = [0, 2, 64]
for (M, N, K, kw) in [(7968, 512, 2048, {}), (7968, 512, 512, {}), (7968, 512, 1536, {}), (7968, 512, 2048, dict(res=1)), (7968, 512, 512, dict(res=1))]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    extra = {}
    if kw.get("res"):
        res = torch.randn(M, N, device="cuda").bfloat16()
        extra.update(bias=torch.randn(N, device="cuda"), aux=Fn._p(res), ldaux=N, aux_mode=1, drop=(0.1, seed_ptr, 0))
    graphs = {}
    ref = None
    for v in VARIANTS:
        exp(v)
        graphs[v] = graph_of(lambda: Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, **extra))
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        else:
            assert torch.equal(C, ref), f"variant {v} changes the result"
    exp(0)
    gt, nt = graph_of(lambda: torch.matmul(A, B.t()))
    res_t = {v: [] for v in VARIANTS}
    res_t["blaslt"] = []
    for rnd in range(6):
        for v in VARIANTS:
            res_t[v].append(time_graph(*graphs[v]))
        res_t["blaslt"].append(time_graph(gt, nt))
    print(f"M={M} N={N} K={K} {kw}: " + "  ".join(f"[{v}] {st.median(t):.2f} (min {min(t):.2f})" for v, t in res_t.items()), flush=True)
