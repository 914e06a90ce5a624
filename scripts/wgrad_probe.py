"""Per-XCC finishing times of the grouped weight-gradient launch (an instrumented build: SC_EXTRA_HIPCC_FLAGS=-DSC_WGRAD_TIMING python -m
speechain_amd.build --force): python scripts/wgrad_probe.py [step|enc|...] [tile].  Prints, per physical XCC, the number of workgroups that did
work, their K tiles, the time the last one ended and the mean time per K tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_wgrad_grouped as B
from speechain_amd._lib import lib, stream

which = sys.argv[1] if len(sys.argv) > 1 else "step"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ps = B.problems(which)
n = len(ps)
dev = "cuda"
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
for _ in range(3):
    lib.call("sc_wgrad_grouped", *args, stream())
torch.cuda.synchronize()
dll = lib.load()
buf = (ctypes.c_longlong * (4096 * 4))()
assert dll.sc_wgrad_debug_read(buf, 4096 * 4) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(4096) if buf[4 * i + 2] > buf[4 * i + 1] > 0]
t0 = min(r[1] for r in rows)
print(f"{which}, tile {tile}: {len(rows)} workgroups did work; kernel {max(r[2] for r in rows) - t0:d} ticks of 10 ns")
for xcc in range(8):
    rs = [r for r in rows if r[0] == xcc]
    if not rs:
        continue
    kt = sum(r[3] for r in rs)
    busy = sum(r[2] - r[1] for r in rs)
    print(f"  XCC {xcc}: {len(rs):4d} workgroups, {kt:7d} K tiles, last end {(max(r[2] for r in rs) - t0) / 100:8.1f} us, first start {(min(r[1] for r in rs) - t0) / 100:6.1f} us, "
          f"{busy / kt / 100 * 1000:7.1f} ns per K tile and workgroup")
