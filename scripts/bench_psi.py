"""Times sc_ctc_prefix_scores_fast alone at the decoding bench's shape (32 sentences x beam 16, 250 frames, 5000 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechain_amd import functional as Fn

B, beam, T, V, plen = 32, 16, 250, 5000, 40
g = torch.Generator().manual_seed(0)
x = torch.log_softmax(torch.randn(B, T, V, generator=g), -1).cuda()
enc_lens = torch.full((B,), T, dtype=torch.int32).cuda()
N = B * beam
r = (torch.randn(T, 2, N, generator=g) * 3 - 20).cuda()
eos_frame = torch.full((N,), T - 1, dtype=torch.int32).cuda()
last = torch.randint(1, V - 1, (N,), generator=g).cuda()
fac = Fn.ctc_prefix_factors(x)
psi = torch.empty(N, V, device="cuda")
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for rep in range(3):
    ts = []
    for i in range(20):
        junk.sum()  # push E out of the caches with clean lines, as the (read-mostly) rest of a position does
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r, last, plen, beam, 0, V - 1, out=psi, factors=fac)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
print("psi kernel (cold caches): median %.1f us, min %.1f us  [SC_CTC_PSI_MFMA=%s]" % (ts[10], ts[0], os.environ.get("SC_CTC_PSI_MFMA")))
