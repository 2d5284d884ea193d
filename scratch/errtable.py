import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from oracle import mm_oracle as mo
from tests.helpers import oracle_params, scale_err, to_dev
from tests.test_gpu_parity import CASES
dev=torch.device('cuda:0')
for case in CASES + [("m512",4,512,8,2,0.1)]:
  name,L,M,d,B,scale=case
  syn=make_svgp(L,M,d,seed=1000+L+M,mean_c=True)
  mu,Sigma=make_inputs(B,d,seed=7,scale=scale)
  f1o,Sffo,cro=mo.mm_gauss_svgp_mo(mu,Sigma,oracle_params(syn))
  model=syn.to_model(dev)
  for dt in (torch.float64,torch.float32):
    x=GaussianMoments((to_dev(mu,dev,dt),to_dev(Sigma,dev,dt)),centered=True)
    m=moment_matching(x,model)
    print(f"{name:12s} {str(dt):14s} f1 {scale_err(m.y.mean(),f1o):.2e} Sff {scale_err(m.y.covariance(),Sffo):.2e} cross {scale_err(m.cross[0],cro):.2e}  |Sff|max {np.abs(Sffo).max():.3g}")
