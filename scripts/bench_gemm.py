import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechain_amd import functional as Fn

def timeit(fn, n=20):
    """GPU time per launch: n launches captured in one hipGraph (no host launch floor), replayed 5 times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3  # us

dev = "cuda"
seed_ptr, _ = Fn.DropoutState.next(0, torch.device("cuda", 0))
def run(M, N, K, ta=0, tb=0, **kw):
    A = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    B = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    extra = {}
    if kw.get("bias"): extra["bias"] = torch.randn(N, device=dev)
    if kw.get("act"): extra["act"] = kw["act"]
    if kw.get("pre"): pre = torch.empty_like(C); extra["pre_out"] = Fn._p(pre)
    if kw.get("res"): res = torch.randn(M, N, device=dev).bfloat16(); extra.update(aux=Fn._p(res), ldaux=N, aux_mode=1)
    if kw.get("drop"): extra["drop"] = (0.1, seed_ptr, 0)
    us = timeit(lambda: Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, A.stride(0), B.stride(0), N, ta=ta, tb=tb, **extra))
    tf = 2.0 * M * N * K / us / 1e6
    print(f"M={M:5d} N={N:5d} K={K:5d} ta={ta} tb={tb} {str(kw):60s} {us:8.1f} us  {tf:7.1f} TF/s")

for M in (7968,):
    run(M, 512, 512)
    run(M, 512, 512, bias=1, res=1, drop=1)
    run(M, 2048, 512)
    run(M, 2048, 512, bias=1, act=2, pre=1, drop=1)
    run(M, 512, 2048)
    run(M, 512, 2048, bias=1, res=1, drop=1)
    run(M, 1536, 512, bias=1)
    run(M, 5000, 512, bias=1)
    run(M, 512, 2048, tb=1)
    run(M, 2048, 512, tb=1)
    run(M, 512, 5000, tb=1)
run(2016, 512, 512)
run(2016, 2048, 512)
run(8192, 8192, 8192)
run(4096, 4096, 4096)
# torch (hipBLASLt) reference on the same shapes
for (M, N, K) in [(7968, 512, 512), (7968, 2048, 512), (7968, 512, 2048), (7968, 5000, 512), (4096, 4096, 4096), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    us = timeit(lambda: torch.matmul(A, B.t()))
    print(f"torch.matmul M={M} N={N} K={K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
