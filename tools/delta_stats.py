import numpy as np, sys
sys.path.insert(0,'/root/repo')
from gpflowpilco_amd.synthetic import make_svgp, make_inputs
L,M,d=8,2000,8
syn=make_svgp(L,M,d,seed=1002)
mu,Sig=make_inputs(4,d,seed=2000,scale=0.1,lo=0.3,hi=0.7)
Z=syn.Z; ls=syn.lengthscales
def delta(b,a,a2):
  La=ls[a]**2; Lb=ls[a2]**2; V=La*Lb/(La+Lb)
  S=Sig[b]+np.diag(V); Si=np.linalg.inv(S)
  T=np.diag(V)@Si@Sig[b]; T=0.5*(T+T.T)
  G=np.diag(1/La)@T@np.diag(1/Lb)
  Pa=np.linalg.inv(Sig[b]+np.diag(La)); Pb=np.linalg.inv(Sig[b]+np.diag(Lb))
  Dr=np.diag(1/La)-np.diag(1/La)@T@np.diag(1/La)-Pa
  Dc=np.diag(1/Lb)-np.diag(1/Lb)@T@np.diag(1/Lb)-Pb
  logk=-0.5*(np.linalg.slogdet(S)[1]-np.sum(np.log(V)))
  lna=0.5*np.sum(np.log(La))-0.5*np.linalg.slogdet(Sig[b]+np.diag(La))[1]
  lnb=0.5*np.sum(np.log(Lb))-0.5*np.linalg.slogdet(Sig[b]+np.diag(Lb))[1]
  zeta=Z-mu[b]
  rho=-0.5*np.einsum('id,de,ie->i',zeta,Dr,zeta)
  gam=-0.5*np.einsum('id,de,ie->i',zeta,Dc,zeta)
  return rho[:,None]+gam[None,:]+(logk-lna-lnb)+zeta@G@zeta.T
for (a,a2) in [(0,1),(2,5),(3,7),(0,0)]:
  D=delta(0,a,a2)
  Dp=np.abs(D[:1984,:1984]).reshape(31,64,62,32).max(axis=(1,3))
  print(a,a2,'ls',np.round(ls[a],2),np.round(ls[a2],2),'max|d|',np.abs(D).max(),'frac entries>1',(np.abs(D)>1).mean(),'frac tiles>1',(Dp>1).mean(),'>2',(Dp>2).mean(), '>0.5',(Dp>0.5).mean())
