"""How long the vendor library (torch.matmul -> hipBLASLt / rocBLAS) takes on the step's NT product shapes: a yardstick for csrc/gemm.hip,
not a code path of the package.   python scripts/blaslt_ref.py"""
import torch

shapes = [(7968, 512, 512), (7968, 1536, 512), (7968, 2048, 512), (7968, 512, 2048), (7968, 512, 1536), (2016, 512, 512), (2016, 512, 2048),
          (2016, 2048, 512), (7968, 5000, 512)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        torch.matmul(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"M={M:5d} N={N:5d} K={K:5d}: {us:6.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TF/s", flush=True)
