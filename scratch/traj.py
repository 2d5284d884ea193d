import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_svgp, make_inputs
dev=torch.device('cuda:0')
L,M,d,B,H=8,2000,8,256,40
lo=float(sys.argv[1]); syn=make_svgp(L,M,d,seed=1002,device='cuda:0',ls_bounds=(lo,3.0)); model=syn.to_model(dev)
pm=model.packed(torch.float32,True,dev)
mu,S=make_inputs(B,d,seed=2000,scale=0.1,lo=0.3,hi=0.7)
mu_t=torch.tensor(mu,dtype=torch.float32,device=dev); S_t=torch.tensor(S,dtype=torch.float32,device=dev)
m,Sg,tm,tS=ops.rollout_closed(pm,mu_t,S_t,H,keep_trajectory=True)
pm.check_status(B)
print('ls lo',lo)
for h in range(0,H,6):
    dg=torch.diagonal(tS[h],dim1=-2,dim2=-1)
    print(h,'mean std',dg.mean().sqrt().item(),'max std',dg.max().sqrt().item(),'mu range',tm[h].min().item(),tm[h].max().item(), 'min eig', torch.linalg.eigvalsh(tS[h].double()).min().item())
