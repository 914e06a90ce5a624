import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
wl = bench.ASRTrainStepWorkload(dev, 0, 1)
for _ in range(5): wl.step(False)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); wl.step(False); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue ms per step:", [round(t * 1e3, 1) for t in ts])
# break-down: forward only / backward only host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): wl.step(False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
