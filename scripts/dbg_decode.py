"""Where a beam-search batch spends its host time: launches, waiting for the device (fetch), the rest (numpy selection + bookkeeping)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from speechain_amd.infer_func import beam_search as BS

dev = torch.device("cuda", 0)
wl = bench.ASRDecodeWorkload(dev, 0)
for _ in range(4):
    wl.step(False)
torch.cuda.synchronize()
acc = {"launch": 0.0, "fetch": 0.0, "n": 0}
L, F = BS._Session.launch, BS._Session.fetch
def launch(self, pos):
    t = time.perf_counter(); L(self, pos); acc["launch"] += time.perf_counter() - t; acc["n"] += 1
def fetch(self, pos):
    t = time.perf_counter(); r = F(self, pos); acc["fetch"] += time.perf_counter() - t; return r
BS._Session.launch, BS._Session.fetch = launch, fetch
t0 = time.perf_counter()
for _ in range(10):
    wl.step(False)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
print(f"per batch: total {tot * 1e3:.2f} ms, launches {acc['launch'] / 10 * 1e3:.2f} ms ({acc['n'] / 10:.0f} of them), waiting in fetch {acc['fetch'] / 10 * 1e3:.2f} ms, "
      f"everything else on the host {(tot - (acc['launch'] + acc['fetch']) / 10) * 1e3:.2f} ms")
