import numpy as np
def remez_like(R,deg,iters=200):
    n=6000
    x=np.cos(np.pi*(np.arange(n)+0.5)/n)*R
    f=np.where(np.abs(x)>1e-9,np.expm1(x)/np.where(x==0,1,x),1+x/2)
    w=np.ones(n)
    best=None
    for it in range(iters):
        V=np.vander(x/R,deg+1,increasing=True)
        c,*_=np.linalg.lstsq(V*w[:,None],f*w,rcond=None)
        err=np.abs(V@c-f)/np.abs(f)
        if best is None or err.max()<best[0]: best=(err.max(),c.copy())
        w=w*(1+4*err/err.max()); w/=w.mean()
    return best[1]/(R**np.arange(deg+1)), best[0]
def test_fused(coef,R):
    c32=coef.astype(np.float32).astype(np.float64)
    x=np.linspace(-R,R,400001).astype(np.float32).astype(np.float64)
    p=np.full_like(x,c32[-1])
    for k in range(len(c32)-2,-1,-1):
        p=np.float32(p*x+c32[k]).astype(np.float64)   # fused: one rounding
    y=np.float32(p*x).astype(np.float64)
    ref=np.expm1(x)
    m=np.abs(ref)>0
    return (np.abs(y-ref)[m]/np.abs(ref)[m]).max()
for R,deg in [(1/64,1),(1/64,2),(1/32,2),(1/16,2),(1/16,3),(1/8,3),(0.25,3),(0.25,4),(0.5,4),(0.5,5),(0.75,6),(1.0,7)]:
    coef,e=remez_like(R,deg)
    print(f"R={R} deg={deg} fit-err={e:.2e} fused-f32 max rel err={test_fused(coef,R):.2e}")
    print('   ',', '.join(f'{v:.9e}f' for v in coef.astype(np.float32)))
