import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import mm_oracle as mo
from oracle.pin_oracle import generate_covariance
from scipy.linalg import solve_triangular, cholesky
rng = np.random.default_rng(1003)
M,d,L,B = int(sys.argv[1]), 8, 1, 2
scale_x = float(sys.argv[2])
Z = rng.uniform(size=(M,d))
ls = np.exp(rng.uniform(np.log(0.3),np.log(3),size=(L,d)))
var = np.full(L,0.89**2); noise = 1e-2*var
a=0
K = mo.se_kernel(Z,None,ls[a],var[a]); Lk = cholesky(K+1e-6*np.eye(M),lower=True)
y = Lk@rng.standard_normal(M) + np.sqrt(noise[a])*rng.standard_normal(M)
Ky = K+noise[a]*np.eye(M)
beta = np.linalg.solve(Ky,y); C = -np.linalg.inv(Ky)
mu = rng.uniform(size=(B,d)); Sig = generate_covariance(rng,d,(B,),scale_x)
f32=np.float32
for b in range(B):
  La=ls[a]**2; V=La/2
  S=Sig[b]+np.diag(V); Si=np.linalg.inv(S)
  T=np.diag(V)-np.diag(V)@Si@np.diag(V); G=np.diag(1/La)@T@np.diag(1/La)
  Pa=np.linalg.inv(Sig[b]+np.diag(La))
  D=np.diag(1/La)-np.diag(1/La)@T@np.diag(1/La)-Pa
  lna=np.log(var[a])+np.sum(np.log(ls[a]))-0.5*np.linalg.slogdet(Sig[b]+np.diag(La))[1]
  logk=np.log(var[a]**2)-0.5*(np.linalg.slogdet(S)[1]-np.sum(np.log(V)))
  const=logk-2*lna
  zeta=Z-mu[b]
  lq=lna-0.5*np.einsum('id,de,ie->i',zeta,Pa,zeta); q=np.exp(lq); w=beta*q
  rho=-0.5*np.einsum('id,de,ie->i',zeta,D,zeta)
  g=zeta@G.T
  delta=rho[:,None]+rho[None,:]+const+zeta@g.T
  # exact
  E=np.expm1(delta)
  bt=(w[:,None]*E*w[None,:]).sum()
  ct0=(C*(q[:,None]*q[None,:])).sum(); ct1=(C*(q[:,None]*E*q[None,:])).sum()
  print(f'b={b} delta range [{delta.min():.3g},{delta.max():.3g}] beta-term {bt:.6g} Cterm0 {ct0:.6g} Cterm1 {ct1:.6g} ecov {var[a]+ct0+ct1:.6g}  sum|C q q| {np.abs(C*(q[:,None]*q[None,:])).sum():.4g} sum|w w| {np.abs(w).sum()**2:.4g} sqrt(sum (Cqq)^2) {np.sqrt(((C*(q[:,None]*q[None,:]))**2).sum()):.4g}')
  # fp32 emulation
  d32=(rho.astype(f32)[:,None]+(rho+const).astype(f32)[None,:]+zeta.astype(f32)@g.astype(f32).T)
  w32=w.astype(f32); q32=q.astype(f32); C32=C.astype(f32)
  En=np.exp(d32)-f32(1); Ea=np.expm1(d32)
  for nm,Ex in (('naive',En),('expm1',Ea)):
    btx=(w32[:,None]*Ex*w32[None,:]).astype(np.float64).sum()
    ct1x=((C32*q32[:,None])*Ex*q32[None,:]).astype(np.float64).sum()
    ctfull=((C32*q32[:,None])*(Ex+f32(1))*q32[None,:]).astype(np.float64).sum()
    print(f'   {nm}: beta-term err {abs(btx-bt):.3g}  Cterm1 err {abs(ct1x-ct1):.3g}  Cfull(fp32) err {abs(ctfull-ct0-ct1):.3g}')
  ct0_32=(C32.astype(np.float64)*(q[:,None]*q[None,:])).sum()
  print(f'   Cterm0 with fp32-stored C, fp64 math: err {abs(ct0_32-ct0):.3g}')
