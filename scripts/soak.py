"""Soak run of a bench workload: N replayed steps on the same synthetic batch, the loss read every 50 steps -- it must stay finite and fall
(a race between the side-stream branches and the main path would show up as NaNs or a stuck loss).   python scripts/soak.py asr|tts [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "asr"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
wl = bench.ASRTrainStepWorkload(dev, 0, 1) if which == "asr" else bench.TTSTrainStepWorkload(dev, 0, 1)
if which == "tts":
    bench.TTSTrainStepWorkload.roofline = bench.ASRTrainStepWorkload.roofline
losses = []
for i in range(steps):
    loss = wl.step(False)
    if i % 50 == 0 or i == steps - 1:
        losses.append(float(loss))
        print(i, losses[-1], flush=True)
assert all(l == l and abs(l) < 1e6 for l in losses), losses
assert losses[-1] < losses[0], losses
print("soak ok", which, losses[0], "->", losses[-1])
