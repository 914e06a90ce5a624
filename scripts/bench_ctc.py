"""CTC forward + backward at BASELINE sizes (32 x 249 x 5000 bf16 logits, 64 labels): time per call; run under rocprofv3
--kernel-trace --stats for the per-kernel split (row_lse, ctc_alpha, ctc_beta, ctc_grad)."""
import torch, sys
sys.path.insert(0,'.')
from speechain_amd.criterion.ctc import CTCLoss
from speechain_amd import functional as Fn
gen=torch.Generator().manual_seed(3)
B,T,V,L=32,249,5000,64
lg=torch.randn(B,T,V,generator=gen).bfloat16().cuda().requires_grad_()
text=torch.randint(1,V-2,(B,L),generator=gen); text[:,0]=V-1; text[:,-1]=V-1
tl=torch.full((B,),L); el=torch.full((B,),T)
c=CTCLoss()
for _ in range(3):
    loss=c(lg,el.cuda(),text.cuda(),tl.cuda()); g,=torch.autograd.grad(loss,lg)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    loss=c(lg,el.cuda(),text.cuda(),tl.cuda()); g,=torch.autograd.grad(loss,lg)
e1.record(); torch.cuda.synchronize()
print('ctc fwd+bwd per call: %.1f us'%(e0.elapsed_time(e1)/10*1e3), float(loss))
