"""Host time of one frontend + SpecAugment step (bench.py --workload frontend) against its device time: python scripts/host_time_frontend.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.FrontendWorkload(dev, 0)
for _ in range(20):
    wl.step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    wl.step(False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6 * (t1 - t0) / 500:.1f} us per step; with the final drain {1e6 * (t2 - t0) / 500:.1f} us per step")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
wl.sa.host_draws = False  # (device draws can be captured; host draws read the CPU generator)
try:
    feat, flen = wl.fe(wl.wave, wl.lens)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        feat, flen = wl.fe(wl.wave, wl.lens)
    e0.record()
    for _ in range(100):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"device time of the frontend alone, replayed: {10 * e0.elapsed_time(e1):.1f} us per step")
except Exception as ex:  # noqa: BLE001
    print("graph capture of the frontend failed:", ex)
wl.sa.host_draws = True
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    wl.step(False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
