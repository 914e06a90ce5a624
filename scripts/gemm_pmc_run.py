"""A few eager launches of the step's heaviest NT GEMM shapes (for rocprofv3 --pmc passes: scripts/pmc_gemm.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402

for (M, N, K) in [(7968, 512, 2048), (7968, 512, 512), (7968, 2048, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(6):
        Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N)
        torch.matmul(A, B.t())
    torch.cuda.synchronize()
