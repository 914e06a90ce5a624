"""Where the multi-rank control flow of bench.py spends its time on ONE rank (run under torch.distributed.run --nproc-per-node 1 with
BENCH_FORCE_DIST=1 BENCH_NO_PREFETCH=1 BENCH_NO_SEGMENTS=1, or without BENCH_FORCE_DIST for the single-rank flow): wall time of each
phase with a device synchronisation after it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

force = os.environ.get("BENCH_FORCE_DIST", "0") == "1"
device = torch.device("cuda:0")
torch.cuda.set_device(device)
if force:
    import torch.distributed as dist
    os.environ["SC_DDP_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group(backend="nccl", device_id=device)
wl = bench.ASRTrainStepWorkload(device, 0, 1, distributed=force)
for _ in range(8):
    wl.step(False)
torch.cuda.synchronize()
acc = {}


def phase(name, fn):
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


N = 30
for _ in range(N):
    wl.nstep += 1
    phase("load_batch(head)", wl._load_batch)
    if wl.segmented:
        for i, (g, ready) in enumerate(zip(wl.g_pieces, wl.piece_buckets)):
            phase(f"piece{i}", g.replay)
            phase(f"buckets{i}", lambda: [wl.reducer.launch_bucket(bi) for bi in ready])
        phase("wait", wl.reducer.wait)
    else:
        phase("body", wl.g_body.replay)
        if wl.reducer.multi:
            phase("reduce_all_now", wl.reducer.reduce_all_now)
    phase("tail", wl.g_tail.replay)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N * 1e3, 3), "ms", flush=True)
if force:
    dist.destroy_process_group()
