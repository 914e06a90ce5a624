import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechain_amd import functional as Fn
torch.manual_seed(0)
bad = 0
for it in range(300):
    for (M, N, K) in [(20, 160, 64), (20, 64, 160), (7968, 512, 512), (333, 200, 100), (128, 128, 96)]:
        A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); pre = torch.empty_like(C)
        Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, bias=bias, act=2, pre_out=Fn._p(pre))
        z = A.float() @ B.float().t() + bias
        ref = torch.nn.functional.gelu(z)
        e1 = float((pre.float() - z).abs().max()); e2 = float((C.float() - ref).abs().max())
        if e1 > 0.1 * (K / 64) ** 0.5 + 0.02 * float(z.abs().max()) or e2 > 0.1 * (K / 64) ** 0.5 + 0.02 * float(z.abs().max()):
            bad += 1
            if bad < 10:
                d = (pre.float() - z).abs()
                idx = torch.nonzero(d > 0.5)
                print("BAD", it, (M, N, K), e1, e2, "n_bad", idx.shape[0], idx[:5].tolist())
print("bad", bad)
