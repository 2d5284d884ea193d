import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_svgp, make_inputs
dev=torch.device('cuda:0')
L,M,d,B=8,2000,8,4
syn=make_svgp(L,M,d,seed=1002,device='cuda:0',ls_bounds=(0.7,3.0)); model=syn.to_model(dev)
mu,S=make_inputs(B,d,seed=2000,scale=0.1,lo=0.3,hi=0.7)
pm64=model.packed(torch.float64,True,dev); pm32=model.packed(torch.float32,True,dev)
mu64=torch.tensor(mu,dtype=torch.float64,device=dev); S64=torch.tensor(S,dtype=torch.float64,device=dev)
# identical (f32-representable) inputs for both modes
mu32=mu64.float(); S32=S64.float(); mu64=mu32.double(); S64=S32.double()
f1,Sff,cr=ops.moment_match(pm64,mu64,S64)
g1,Gff,gc=ops.moment_match(pm32,mu32,S32)
h1,Hff,hc=ops.moment_match(pm32,mu32,S32,force_generic=True)
E=(Gff.double()-Sff).abs(); Eg=(Hff.double()-Sff).abs()
print('Sff[0]\n',Sff[0].cpu().numpy().round(5))
print('err mfma-bf16x3 [0]\n',E[0].cpu().numpy())
print('max err mfma',E.max().item(),' generic-f32',Eg.max().item(),' max|Sff|',Sff.abs().max().item())
ws=pm64.workspace(B, 3)
# weights
from gpflowpilco_amd import _lib
import ctypes
n=_lib.lib().mm_workspace_bytes(B,L,M,d,1,3)
print('f1 err',(g1.double()-f1).abs().max().item())
