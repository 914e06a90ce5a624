"""The step's weight-gradient list (12 encoder + 6 decoder layers + heads, BASELINE configs[2]) through sc_wgrad_grouped, alone:
time per call (hipGraph-replayed, no host floor) for both tile sizes, and sub-lists that isolate one shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechain_amd._lib import lib, stream

dev = "cuda"
TE, TD = 7968, 2016


def problems(which):
    enc = [(1536, 512, TE), (512, 512, TE), (2048, 512, TE), (512, 2048, TE)]
    dec = [(1536, 512, TD), (512, 512, TD), (512, 512, TD), (1024, 512, TE), (512, 512, TD), (2048, 512, TD), (512, 2048, TD)]
    heads = [(5000, 512, TD), (5000, 512, TE)]
    if which == "step":
        return enc * 12 + dec * 6 + heads
    if which == "enc":
        return enc * 12
    if which == "enc4":
        return enc * 4
    if which == "ffn1":
        return [(2048, 512, TE)] * 16
    if which == "sq":
        return [(512, 512, TE)] * 64
    if which == "dec":
        return dec * 6 + heads
    if which == "dec6":
        return dec * 6
    if which == "ctc":
        return [(5000, 512, TE)]
    if which == "ctc8":
        return [(5000, 512, TE)] * 8
    if which == "ctcpad":
        return [(5120, 512, TE)] * 8
    if which == "vocab":
        return [(5000, 512, TD)]
    if which == "crosskv":
        return [(1024, 512, TE)] * 6
    if which == "short":
        return [(2048, 512, TD), (512, 2048, TD)] * 16
    raise KeyError(which)


def run(which, tile, reps=10):
    ps = problems(which)
    n = len(ps)
    dz = [torch.randn(t, no, device=dev).bfloat16() for no, ni, t in ps]
    x = [torch.randn(t, ni, device=dev).bfloat16() for no, ni, t in ps]
    dW = [torch.empty(no, ni, device=dev) for no, ni, t in ps]
    db = [torch.empty(no, device=dev) for no, ni, t in ps]
    arr = lambda ct, vals: (ct * n)(*vals)
    n_out, n_in = arr(ctypes.c_int, [p[0] for p in ps]), arr(ctypes.c_int, [p[1] for p in ps])
    ws = torch.empty(lib.value("sc_wgrad_grouped_workspace", n, n_out, n_in), dtype=torch.uint8, device=dev)
    args = (n, arr(ctypes.c_void_p, [t.data_ptr() for t in dz]), arr(ctypes.c_void_p, [t.data_ptr() for t in x]),
            arr(ctypes.c_void_p, [t.data_ptr() for t in dW]), arr(ctypes.c_void_p, [t.data_ptr() for t in db]),
            n_out, n_in, arr(ctypes.c_int, [p[2] for p in ps]),
            arr(ctypes.c_longlong, [p[0] for p in ps]), arr(ctypes.c_longlong, [p[1] for p in ps]), None, tile, ws.data_ptr(), ws.numel())
    fn = lambda: lib.call("sc_wgrad_grouped", *args, stream())
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = sum(2.0 * a * b * c for a, b, c in ps)
    by = sum(2.0 * c * (a + b) for a, b, c in ps)
    tiles = sum(-(-a // tile) * -(-b // tile) for a, b, c in ps)
    print(f"{which:6s} tile={tile} problems={n:3d} tiles={tiles:5d}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  operands {by / 1e6:7.1f} MB -> {by / us / 1e6:5.2f} TB/s if read once")


if __name__ == "__main__":
    which = sys.argv[1:] or ["step", "enc", "enc4", "ffn1", "sq", "dec"]
    for w in which:
        for tile in (256, 128):
            run(w, tile)
