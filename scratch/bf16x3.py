import numpy as np
rng=np.random.default_rng(0)
def bf16(x):
    # round-to-nearest-even to bfloat16, returned as float32
    u=np.asarray(x,np.float32).view(np.uint32)
    r=((u>>16)&1)+0x7fff
    return ((u+r)&0xffff0000).view(np.float32)
def split3(x):
    x=np.asarray(x,np.float32)
    h=bf16(x); m=bf16(x-h); l=bf16(x-h-m)
    return h,m,l
M,d=512,8
A=rng.standard_normal((M,d)).astype(np.float32)*0.7
Z=(rng.uniform(size=(M,d))-0.5).astype(np.float32)
exact=A.astype(np.float64)@Z.astype(np.float64).T
f32=(A@Z.T)
Ah,Am,Al=split3(A); Zh,Zm,Zl=split3(Z)
terms=[(Ah,Zh),(Ah,Zm),(Am,Zh),(Ah,Zl),(Al,Zh),(Am,Zm)]
# emulate fp32 accumulation of exact bf16 products in MFMA order (k ascending within term, terms sequential)
acc=np.zeros((M,M),np.float32)
for a,z in terms[::-1]:   # small terms first or last? try both
    for k in range(d):
        acc=(acc+ (a[:,k:k+1].astype(np.float32)*z[:,k].astype(np.float32)[None,:])).astype(np.float32)
acc2=np.zeros((M,M),np.float32)
for a,z in terms:
    for k in range(d):
        acc2=(acc2+ (a[:,k:k+1]*z[:,k][None,:])).astype(np.float32)
scale=np.abs(A).astype(np.float64)@np.abs(Z).astype(np.float64).T
print('f32 fma-chain  max err',np.abs(f32-exact).max(),' rel to sum|a||z|',(np.abs(f32-exact)/scale).max())
print('bf16x3 small-first',np.abs(acc-exact).max(),(np.abs(acc-exact)/scale).max())
print('bf16x3 big-first  ',np.abs(acc2-exact).max(),(np.abs(acc2-exact)/scale).max())
print('residual of split',np.abs(A-(Ah+Am+Al)).max())
