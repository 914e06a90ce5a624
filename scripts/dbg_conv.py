import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechain_amd import functional as Fn
from speechain_amd.ddp import GradBucketReducer

Fn.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
B, T, C, N, k = 3, 50, 64, 256, 3
conv = torch.nn.Conv1d(C, N, k, padding=1).cuda()
conv2 = torch.nn.Conv1d(N, C, k, padding=1).cuda()
red = GradBucketReducer(list(conv.parameters()) + list(conv2.parameters()), bucket_mb=0.25)
x = torch.randn(B, T, C).cuda().bfloat16()
res = torch.randn(B, T, C).cuda().bfloat16()


def both(fn):
    outs = []
    for direct in (True, False):
        Fn._CONV_DIRECT = direct
        Fn.DropoutState.reset(77)
        Fn.DropoutState.advance()
        outs.append(fn().detach().float())
    return outs


for name, fn in (("conv1 relu drop", lambda: Fn.conv_linear(x, conv.weight, conv.bias, "ReLU", 0.1)),
                 ("conv1 none drop", lambda: Fn.conv_linear(x, conv.weight, conv.bias, None, 0.1)),
                 ("conv2 res nodrop", lambda: Fn.conv_linear(Fn.conv_linear(x, conv.weight, conv.bias, "ReLU", 0.0), conv2.weight, conv2.bias, None, 0.0, res)),
                 ("conv2 res drop", lambda: Fn.conv_linear(Fn.conv_linear(x, conv.weight, conv.bias, "ReLU", 0.0), conv2.weight, conv2.bias, None, 0.1, res)),
                 ("chain drop", lambda: Fn.conv_linear(Fn.conv_linear(x, conv.weight, conv.bias, "ReLU", 0.1), conv2.weight, conv2.bias, None, 0.1, res))):
    a, b = both(fn)
    print(name, "max diff", float((a - b).abs().max()), "frac differing", float(((a - b).abs() > 0).float().mean()), flush=True)
