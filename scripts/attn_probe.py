"""Phase timing of attn_bwd_kl_kernel from an instrumented build (SC_EXTRA_HIPCC_FLAGS=-DSC_ATTN_TIMING python -m speechain_amd.build --force):
s_memtime at the phase boundaries of workgroup 0, per wave and 64-query round.  Rebuild without the flag afterwards."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd._lib import lib, stream  # noqa: E402

Tq, Tk, drop_p = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
B, H, dh = 32, 8, 64
gen = torch.Generator().manual_seed(1)
q = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
k = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
v = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
do = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
out = torch.empty_like(q)
lse = torch.empty(B * H, Tq, device="cuda")
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
scale = 1.0 / math.sqrt(dh)
lib.call("sc_attention_fwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(lse), B, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0),
         k.stride(1), out.stride(0), out.stride(1), None, 0, 0, 0, scale, drop_p, Fn._p(seed), 0, stream())
for _ in range(3):
    lib.call("sc_attention_bwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(do), Fn._p(lse), Fn._p(dq), Fn._p(dk), Fn._p(dv), B, H,
             Tq, Tk, q.stride(0), q.stride(1), k.stride(0), k.stride(1), out.stride(0), out.stride(1), dq.stride(0), dq.stride(1),
             dk.stride(0), dk.stride(1), None, 0, 0, 0, scale, drop_p, Fn._p(seed), 0, None, 0, stream())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 512)()
fn = lib.load().sc_attention_debug_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 512) == 0
NW = 8 if Tk > 128 else 4
t = [[[buf[(w * 8 + r) * 8 + p] for p in range(8)] for r in range(8)] for w in range(NW)]
t0 = min(t[w][7][0] for w in range(NW))
print(f"Tq={Tq} Tk={Tk} p={drop_p}; ticks relative to the first wave's start")
for w in range(NW):
    print(f"wave {w}: start {t[w][7][0] - t0}, prologue done {t[w][0][7] - t0}, end of rounds {t[w][7][1] - t0}, end {t[w][7][2] - t0}")
    for r in range((Tq + 63) // 64):
        a = t[w][r]
        print(f"   round {r}: top {a[0] - t0:7d} | fetch+barrier {a[1] - a[0]:6d} | phase A {a[2] - a[1]:6d} | barrier {a[3] - a[2]:6d} | dQ mfma {a[4] - a[3]:6d} | "
              f"stash {a[5] - a[4]:6d}")
