"""Times sc_attention_fwd / sc_attention_bwd alone on the three shapes of the ASR step (B 32, 8 heads of 64, dropout 0.1, ragged key
masks): encoder self-attention (249 x 249), decoder self-attention (63 x 63, causal), decoder cross-attention (63 x 249).

    python scripts/bench_attn.py            # SC_ATTN_BWD_OLD=1 for the previous backward kernel
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd._lib import lib, stream  # noqa: E402


def run(name, Tq, Tk, causal, drop_p, B=32, H=8, dh=64, iters=50):
    gen = torch.Generator().manual_seed(1)
    q = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
    k = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    v = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    do = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
    lens = torch.randint(Tk * 2 // 3, Tk + 1, (B,), generator=gen).cuda()
    lens[0] = Tk
    m8 = (torch.arange(Tk, device="cuda")[None, :] < lens[:, None]).to(torch.uint8)[:, None, :].contiguous()
    out = torch.empty_like(q)
    lse = torch.empty(B * H, Tq, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    scale = 1.0 / math.sqrt(dh)

    def fwd():
        lib.call("sc_attention_fwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(lse), B, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0),
                 k.stride(1), out.stride(0), out.stride(1), Fn._p(m8), m8.stride(0), 0, causal, scale, drop_p, Fn._p(seed), 0, stream())

    def bwd():
        lib.call("sc_attention_bwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(do), Fn._p(lse), Fn._p(dq), Fn._p(dk), Fn._p(dv), B, H,
                 Tq, Tk, q.stride(0), q.stride(1), k.stride(0), k.stride(1), out.stride(0), out.stride(1), dq.stride(0), dq.stride(1),
                 dk.stride(0), dk.stride(1), Fn._p(m8), m8.stride(0), 0, causal, scale, drop_p, Fn._p(seed), 0, None, 0, stream())

    res = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    flops = 4.0 * B * H * Tq * Tk * dh
    print(f"{name:12s} Tq={Tq:4d} Tk={Tk:4d} causal={causal} p={drop_p}: fwd {res[0]:6.1f} us ({flops / res[0] * 1e-6:6.1f} TF/s)   "
          f"bwd {res[1]:6.1f} us ({2.5 * flops / res[1] * 1e-6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    for p in (0.1, 0.0):
        run("enc self", 249, 249, 0, p)
        run("dec self", 63, 63, 1, p)
        run("dec cross", 63, 249, 0, p)
