"""Where a launch of gemm_bf16_nt_dma_kernel spends its time, from an instrumented build
(SC_EXTRA_HIPCC_FLAGS=-DSC_GEMM_TIMING python -m speechain_amd.build --force; rebuild without the flag afterwards):
per workgroup the dispatch offset, prologue (entry -> first tile visible), K loop, epilogue (-> stores drained), and for workgroup 0 the
stamps of four steady-state K tiles per wave (top of the publish step, after the counted vmcnt wait, after the barrier, after the DMA issue)."""
import ctypes
import os
import statistics as st
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd._lib import lib  # noqa: E402

fn = lib.load().sc_gemm_debug_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
NP = 1024 * 8 + 8 * 4 * 4


def probe(M, N, K, **kw):
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    extra = {}
    if kw.get("bias"):
        extra["bias"] = torch.randn(N, device="cuda")
    if kw.get("res"):
        res = torch.randn(M, N, device="cuda").bfloat16()
        extra.update(aux=Fn._p(res), ldaux=N, aux_mode=1)
    run = lambda: Fn.gemm(1, Fn._p(A), Fn._p(B), Fn._p(C), M, N, K, K, K, N, **extra)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    # cold-ish: evict by touching a large buffer, then one launch
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    big.fill_(1)
    torch.cuda.synchronize()
    for mode in ("after a 256 MB fill", "back to back"):
        if mode == "back to back":
            run()
        run()
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * NP)()
        assert fn(buf, NP) == 0
        nwg = min(1024, ((M + 127) // 128) * ((N + 127) // 128))
        W = [[buf[i * 8 + k] for k in range(8)] for i in range(nwg)]
        W = [w for w in W if w[4] > 0]
        t0 = min(w[4] for w in W)
        start = sorted((w[4] - t0) * 10 for w in W)
        end = sorted((w[5] - t0) * 10 for w in W)
        pro = sorted(w[1] - w[0] for w in W)
        loop = sorted(w[2] - w[1] for w in W)
        epi = sorted(w[3] - w[2] for w in W)
        q = lambda v: f"min {v[0]} / med {v[len(v) // 2]} / max {v[-1]}"
        print(f"M={M} N={N} K={K} {kw} [{mode}] kernel {lib.load().sc_gemm_kernel_name(lib.load().sc_gemm_last_kernel()).decode()}: {len(W)} workgroups")
        print(f"   first stamp -> last stamp {end[-1]} ns;  starts (ns) {q(start)};  ends (ns) {q(end)}")
        print(f"   cycles: prologue {q(pro)} | K loop {q(loop)} ({K // 32} tiles: {loop[len(loop) // 2] / (K // 32):.0f} per tile) | epilogue {q(epi)}")
        # placement: workgroups per (XCC, SE, SH, CU) -- gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; and the K-loop time by how many share a CU
        import collections
        where = collections.defaultdict(list)
        for w in W:
            hw, xcc = w[6] & 0xFFFFFFFF, (w[6] >> 32) & 0xF
            where[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append(w[2] - w[1])
        by_n = collections.defaultdict(list)
        for k, v in where.items():
            by_n[len(v)] += v
        print(f"   placement: {len(where)} CUs used; " + "; ".join(f"{n} per CU: {len(v)} workgroups, K loop med {sorted(v)[len(v) // 2]} max {max(v)}" for n, v in sorted(by_n.items())))
        by_xcc = collections.defaultdict(list)
        for i, w in enumerate(W):
            by_xcc[(w[6] >> 32) & 0xF].append(w[2] - w[1])
        print("   K loop by XCC (median / max / count):", {k: (sorted(v)[len(v) // 2], max(v), len(v)) for k, v in sorted(by_xcc.items())})
        clk = collections.defaultdict(list)
        ph = collections.defaultdict(list)
        for w in W:
            x = (w[6] >> 32) & 0xF
            clk[x].append((w[3] - w[0]) / max(1, (w[5] - w[4])) * 0.1)  # shader cycles per ns = GHz (s_memtime ticks / 10 ns realtime ticks)
            ph[x].append((w[1] - w[0], w[3] - w[2]))
        print("   clock by XCC (GHz, median):", {k: round(sorted(v)[len(v) // 2], 3) for k, v in sorted(clk.items())})
        print("   prologue / epilogue cycles by XCC (median):", {k: (sorted(a for a, _ in v)[len(v) // 2], sorted(b for _, b in v)[len(v) // 2]) for k, v in sorted(ph.items())})
        slow = [i for i, w in enumerate(W) if (w[2] - w[1]) > 1.15 * loop[len(loop) // 2]]
        gx = (N + 127) // 128
        print(f"   slow workgroups (K loop > 1.15 x median): {len(slow)}")
        dec = sorted(w[2] - w[1] for w in W)
        print("   K loop deciles:", [dec[int(i * (len(dec) - 1) / 10)] for i in range(11)])
    base = 1024 * 8
    print("   workgroup 0, tiles 8..11 per wave: [wait for DMA | barrier | DMA issue | rest of the tile (reads + MFMA)] cycles")
    for w in range(8):
        rows = []
        for t in range(4):
            a = [buf[base + (w * 4 + t) * 4 + k] for k in range(4)]
            nxt = buf[base + (w * 4 + t + 1) * 4] if t < 3 else None
            rows.append(f"[{a[1] - a[0]} | {a[2] - a[1]} | {a[3] - a[2]} | {'' if nxt is None else nxt - a[3]}]")
        print(f"     wave {w}: " + " ".join(rows))


if len(sys.argv) > 1:  # experiment flags (csrc/gemm.hip: sc_gemm_experiment), e.g. 8 = XCD labels rotated by one
    e = lib.load().sc_gemm_experiment
    e.argtypes = [ctypes.c_int]
    e.restype = None
    e(int(sys.argv[1]))
    probe(7968, 512, 2048)
    probe(7968, 512, 512)
    sys.exit(0)
probe(7968, 512, 2048)
probe(7968, 512, 2048, bias=1, res=1)
probe(7968, 512, 512)
probe(7968, 1536, 512, bias=1)
