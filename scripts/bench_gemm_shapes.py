"""Back-to-back launch times of the NT bf16 GEMM on the train step's shapes (python scripts/bench_gemm_shapes.py): compare two builds by running it in two trees."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechain_amd import functional as Fn
from speechain_amd._lib import lib


def timed(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * n) * 1e3)
    return best

out = []
for (M, N, K) in [(7968, 1536, 512), (7968, 512, 512), (7968, 2048, 512), (7968, 512, 2048), (7968, 512, 1536), (7968, 1024, 512), (2016, 1536, 512),
                  (2016, 512, 512), (2016, 2048, 512), (2016, 512, 2048), (2016, 5000, 512), (7968, 5000, 512), (2016, 512, 5000), (7968, 512, 5000)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timed(lambda: Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N))
    name = lib.load().sc_gemm_kernel_name
    out.append(f"{M}x{N}x{K}: {t:6.2f}")
print("  ".join(out))
