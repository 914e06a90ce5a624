import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from speechain_amd._lib import lib
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
wl = bench.FrontendWorkload(dev, 0)
for _ in range(8):
    wl.step(False)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "events"
out = []
T = time.perf_counter
for i in range(400):
    t0 = T()
    if mode == "events":
        lib.timing = {"sc_frontend_logmel_fwd": wl.ev}
    feat, flen = wl.fe(wl.wave, wl.lens)
    lib.timing = None
    t1 = T()
    draws = wl.sa.draw(wl.flen_host, feat.size(0), feat.size(-1), torch.device("cpu"))
    t2 = T()
    o = wl.sa.apply(feat, draws)
    t3 = T()
    if t3 - t0 > 1e-3:
        out.append((i, round((t1 - t0) * 1e6), round((t2 - t1) * 1e6), round((t3 - t2) * 1e6)))
t0 = T(); torch.cuda.synchronize(); print(mode, "drain us", round((T() - t0) * 1e6), "outliers (step, frontend, draw, apply):", out)
import statistics
ph = [[], [], []]
for i in range(300):
    t0 = T(); feat, flen = wl.fe(wl.wave, wl.lens); t1 = T()
    draws = wl.sa.draw(wl.flen_host, feat.size(0), feat.size(-1), torch.device("cpu")); t2 = T()
    o = wl.sa.apply(feat, draws); t3 = T()
    ph[0].append(t1 - t0); ph[1].append(t2 - t1); ph[2].append(t3 - t2)
torch.cuda.synchronize()
print("median host us: frontend %.1f  draw %.1f  apply %.1f" % tuple(1e6 * statistics.median(p) for p in ph))
