"""Conv2dPrenet alone, forward + backward at the bench shapes (32 x 1001 x 80 features), for per-kernel profiling:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pn -- python scripts/bench_prenet.py 64,64"""
import sys
import time

import torch

sys.path.insert(0, ".")
from speechain_amd import functional as Fn  # noqa: E402
from speechain_amd.module.prenet.conv2d import Conv2dPrenet  # noqa: E402

dims = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "64,64").split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Fn.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
m = Conv2dPrenet(input_size=80, conv_dims=dims, conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=512).cuda().train()
x = torch.randn(32, 1001, 80, device="cuda")
lens = torch.full((32,), 1001, device="cuda")
w = None
for it in range(iters + 3):
    if it == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    y, _ = m(x, lens)
    if w is None:
        w = torch.randn_like(y)
    for p in m.parameters():
        p.grad = None
    (y * w).sum().backward()
torch.cuda.synchronize()
print(f"prenet{dims}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per forward+backward (eager, {iters} iterations)")
