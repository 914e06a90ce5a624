import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cpu_ref as R
from speechain_amd.module.augment.specaug import SpecAugment
torch.manual_seed(0)
feat = torch.randn(2, 1001, 80)
mod = SpecAugment(input_size=80, freq_mask=False, time_mask=False)
for (c, p, tmin) in [(306, 310, 619), (448, 447, 619), (500, 503, 1001)]:
    d = dict(time_minlen=tmin, time_maxlen=1001, warp_center=c, warp_pos=p, freq_len=None, freq_pos=None, time_len=None, time_pos=None)
    out = mod.apply(feat.cuda(), d).cpu()
    ref = R.specaug_apply(feat, torch.tensor([1001, tmin]), d)
    err = (out - ref).abs().max(dim=2).values.max(dim=0).values
    t = int(err.argmax())
    print(c, p, tmin, "max err", float(err.max()), "at row", t, "rows>1e-5:", int((err > 1e-5).sum()))
    # expected source position
    if t < p:
        scale = torch.tensor((c - 1) / (p - 1), dtype=torch.float32); dst = t
    else:
        scale = torch.tensor((tmin - c - 1) / (tmin - p - 1), dtype=torch.float32); dst = t - p
    print("   scale", float(scale), "real", float(scale * dst), "rows with err:", torch.nonzero(err > 1e-5).flatten()[:10].tolist())

    # torch-explicit evaluated with the oracle formula (CPU) for comparison
    ref2 = R.specaug_apply(feat, torch.tensor([1001, tmin]), d, explicit_warp=True)
    print("   kernel-vs-explicit", float((out - ref2).abs().max()), " interp-vs-explicit", float((ref - ref2).abs().max()))
