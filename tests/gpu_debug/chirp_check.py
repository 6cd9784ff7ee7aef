import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import signals as S
from fnft_amd import capi
from oracle import load_oracle
orc=load_oracle()
rng=np.random.default_rng(5)
def truth(p,A,W,M):
    lW=np.log(np.clongdouble(W)); lA=np.log(np.clongdouble(A))
    m=np.arange(M,dtype=np.longdouble)
    z=np.exp(-(lA-m*lW))
    acc=np.zeros(M,np.clongdouble)
    for c in p: acc=acc*z+np.clongdouble(c)
    return acc
for deg,M,phiW,phiA in [(100,50,1e-3,0.1),(1000,200,1e-4,0.01),(2048,2048,6e-5,1e-2),(8000,300,3e-6,2e-3),(4096,4096,3.6e-5,1e-3),(16384,16384,1e-6,1e-3)]:
    p=(rng.standard_normal(deg+1)+1j*rng.standard_normal(deg+1))*np.exp(-np.linspace(-3,3,deg+1)**2)
    W=np.exp(1j*phiW); A=np.exp(1j*phiA)
    rc,out=capi.poly_chirpz(p,A,W,M)
    ref=orc.poly_chirpz(p,A,W,M)
    tr=truth(p,A,W,M) if deg*M<=70e6 else None
    e=lambda x,y: float(np.sum(np.abs(x-y))/np.sum(np.abs(y)))
    print(deg,M,rc,"gpu-vs-oracle %.2e"%e(out,ref), ("gpu-vs-truth %.2e oracle-vs-truth %.2e"%(e(out,tr),e(ref,tr))) if tr is not None else "")
