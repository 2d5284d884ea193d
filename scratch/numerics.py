import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import mm_oracle as mo
from oracle.pin_oracle import generate_covariance
from scipy.linalg import solve_triangular, cholesky

rng = np.random.default_rng(1003)
M,d,L,B = int(sys.argv[1]) if len(sys.argv)>1 else 400, 8, 2, 2
scale_x = float(sys.argv[2]) if len(sys.argv)>2 else 0.1
Z = rng.uniform(size=(M,d))
ls = np.exp(rng.uniform(np.log(0.3),np.log(3),size=(L,d)))
var = np.full(L,0.89**2); noise = 1e-2*var
q_mu = np.empty((M,L)); q_sqrt = np.empty((L,M,M)); beta=np.empty((L,M)); C=np.empty((L,M,M))
for a in range(L):
  K = mo.se_kernel(Z,None,ls[a],var[a])
  Lk = cholesky(K+1e-6*np.eye(M),lower=True)
  y = Lk@rng.standard_normal(M) + np.sqrt(noise[a])*rng.standard_normal(M)
  Ky = K+noise[a]*np.eye(M)
  m = K@np.linalg.solve(Ky,y)
  S = K - K@np.linalg.solve(Ky,K)
  v = solve_triangular(Lk,m,lower=True)
  Sw = solve_triangular(Lk, solve_triangular(Lk,S,lower=True).T, lower=True)
  Sw = 0.5*(Sw+Sw.T)
  q_mu[:,a]=v; q_sqrt[a]=np.linalg.cholesky(Sw + 1e-12*np.eye(M))
  beta[a] = solve_triangular(Lk.T, v, lower=False)
  A = q_sqrt[a]@q_sqrt[a].T - np.eye(M)
  C[a] = solve_triangular(Lk.T, solve_triangular(Lk.T, A.T, lower=False).T, lower=False)  # L^-T A L^-1
  print('latent',a,'|beta| max',np.abs(beta[a]).max(),'cond K',np.linalg.cond(K+1e-6*np.eye(M)))
model = mo.SVGPParams(Z=np.broadcast_to(Z,(L,M,d)).copy(), lengthscales=ls, variance=var, q_mu=q_mu,q_sqrt=q_sqrt,whiten=True)
mu = rng.uniform(size=(B,d)); Sig = generate_covariance(rng,d,(B,),scale_x)
f1o,Sffo,cro = mo.mm_gauss_svgp_mo(mu,Sig,model)

def fused(dtype):
  f1=np.zeros((B,L)); Sff=np.zeros((B,L,L)); Sff_unc=np.zeros((B,L,L)); cs=np.zeros((B,L))
  for b in range(B):
    lq=[];w=[]
    for a in range(L):
      Lam=np.diag(ls[a]**2); P=np.linalg.inv(Sig[b]+Lam)
      zeta=Z-mu[b]
      lqa = np.log(var[a])+np.sum(np.log(ls[a]))-0.5*np.linalg.slogdet(Sig[b]+Lam)[1]-0.5*np.einsum('id,de,ie->i',zeta,P,zeta)
      lq.append(lqa); w.append((beta[a]*np.exp(lqa)).astype(dtype))
      f1[b,a]=w[-1].astype(np.float64).sum()
    for a in range(L):
      for a2 in range(a,L):
        La=ls[a]**2; Lb=ls[a2]**2; V=La*Lb/(La+Lb)
        S=Sig[b]+np.diag(V); Si=np.linalg.inv(S)
        T=np.diag(V)-np.diag(V)@Si@np.diag(V)
        G=np.diag(1/La)@T@np.diag(1/Lb)
        logk=np.log(var[a]*var[a2])-0.5*(np.linalg.slogdet(S)[1]-np.sum(np.log(V)))
        Pa=np.linalg.inv(Sig[b]+np.diag(La)); Pb=np.linalg.inv(Sig[b]+np.diag(Lb))
        Drow=np.diag(1/La)-np.diag(1/La)@T@np.diag(1/La)-Pa
        Dcol=np.diag(1/Lb)-np.diag(1/Lb)@T@np.diag(1/Lb)-Pb
        zeta=Z-mu[b]
        lna=np.log(var[a])+np.sum(np.log(ls[a]))-0.5*np.linalg.slogdet(Sig[b]+np.diag(La))[1]
        lnb=np.log(var[a2])+np.sum(np.log(ls[a2]))-0.5*np.linalg.slogdet(Sig[b]+np.diag(Lb))[1]
        const=logk-lna-lnb
        rho=(-0.5*np.einsum('id,de,ie->i',zeta,Drow,zeta)+const).astype(dtype)
        gam=(-0.5*np.einsum('id,de,ie->i',zeta,Dcol,zeta)).astype(dtype)
        g=(zeta@G.T).astype(dtype)   # g_j = G zeta_j
        zt=zeta.astype(dtype)
        delta = rho[:,None]+gam[None,:]+ (zt@g.T)
        e = np.expm1(delta) if dtype==np.float64 else (np.exp(delta)-dtype(1))
        val = (w[a][:,None]*e*w[a2][None,:]).astype(np.float64).sum()
        Sff[b,a,a2]=Sff[b,a2,a]=val
        # uncentred in dtype: log Q = delta + lq_i + lq_j
        lQ = (delta.astype(np.float64) + lq[a][:,None]+lq[a2][None,:]).astype(dtype) if dtype==np.float64 else (delta + lq[a].astype(dtype)[:,None]+lq[a2].astype(dtype)[None,:])
        Q = np.exp(lQ)
        f2 = (beta[a].astype(dtype)[:,None]*Q*beta[a2].astype(dtype)[None,:]).astype(np.float64).sum()
        Sff_unc[b,a,a2]=Sff_unc[b,a2,a]=f2-f1[b,a]*f1[b,a2]
        if a==a2:
          # centred C term: sum C_ij q_i q_j e^delta
          qq=np.exp(lq[a]).astype(dtype)
          cs_=(C[a].astype(dtype)*(qq[:,None]*np.exp(delta)*qq[None,:])).astype(np.float64).sum()
          Sff[b,a,a]+=var[a]+cs_; Sff_unc[b,a,a]+=var[a]+cs_
  return f1,Sff,Sff_unc
for dt in (np.float64,np.float32):
  f1,Sff,Sffu=fused(dt)
  print(dt.__name__,'f1 err',np.abs(f1-f1o).max(),'|f1|',np.abs(f1o).max())
  print('  Sff centred err',np.abs(Sff-Sffo).max(),' uncentred err',np.abs(Sffu-Sffo).max(),' |Sff| diag',np.diagonal(Sffo,axis1=1,axis2=2).ravel())
