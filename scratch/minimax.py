import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
def fit(R,deg):
    # near-minimax fit of f(x)=expm1(x)/x on [-R,R] via Chebyshev interpolation at many nodes + Remez-like reweight
    n=4000
    x=np.cos(np.pi*(np.arange(n)+0.5)/n)*R
    f=np.where(np.abs(x)>1e-8,np.expm1(x)/np.where(x==0,1,x),1+x/2)
    w=np.ones(n)
    for it in range(60):
        V=np.vander(x/R,deg+1,increasing=True)
        c,*_=np.linalg.lstsq(V*w[:,None],f*w,rcond=None)
        err=np.abs(V@c-f)/np.abs(f)
        w=w*(1+2*err/err.max()); w/=w.mean()
    coef=c/(R**np.arange(deg+1))   # monomial coefficients in x
    return coef
def test(coef,R):
    c32=coef.astype(np.float32)
    x=np.linspace(-R,R,2000001).astype(np.float32)
    p=np.full_like(x,c32[-1])
    for k in range(len(c32)-2,-1,-1):
        p=(p*x+c32[k]).astype(np.float32)   # not fused, pessimistic
    y=(p*x).astype(np.float32)
    ref=np.expm1(x.astype(np.float64))
    rel=np.abs(y-ref)/np.maximum(np.abs(ref),1e-30)
    return rel.max(), np.abs(y-ref).max()
for R,deg in [(0.75,5),(0.75,6),(0.75,7),(1.0,6),(1.0,7),(1.0,8),(0.5,5),(0.5,6)]:
    coef=fit(R,deg)
    r,a=test(coef,R)
    print(R,deg,'max rel err',r,'max abs',a)
    if (R,deg) in [(0.75,6),(1.0,7)]:
        print('   coefs:',', '.join(f'{v:.9e}f' for v in coef.astype(np.float32)))
# Taylor deg 10 for reference
co=np.array([1/np.math.factorial(k+1) for k in range(10)]) if hasattr(np,'math') else None
