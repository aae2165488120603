"""Study behind csrc/ssm_wino5.hip: which interpolation points keep the two-dimensional Winograd form F(4x4,5x5) of the 5x5 layers
(conv2a / conv2b, scripts/models/layers.py:21-33 in the reference's UNet) inside the 5e-5 per-layer bar. Every step of the fp32 pipeline
(filter transform in float64 rounded once, B^T d B, the 64 channel sums, A^T m A) is emulated in numpy at the kernel's precision and
compared with a float64 direct convolution; the direct fp32 convolution and the one-dimensional F(4,5) form of csrc/ssm_wino1d.hip are
printed beside it.      python tests/emulate_winograd_5x5_2d.py       (CPU only, about a minute; not collected by pytest)
Result (64 channels, unit-scale output): {0, +-1, +-2, +-1/2, inf} 3e-6 rms / 3e-5 max - the set the kernel uses; the wino4 set scaled
by one more point pair is no better, and every set with a point beyond 2 is worse."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
f32=np.float32
from emulate_winograd_7x7_blocked import cook_toom, matapply
def run(x,w,mats,dt,m=4,r=5):
    AT,G,BT=mats; n=m+r-1
    B,C,H,W=x.shape; N=w.shape[0]; th,tw=H//m,W//m; pad=(r-1)//2
    xp=np.zeros((B,C,H+2*pad+m,W+2*pad+m),dt); xp[:,:,pad:pad+H,pad:pad+W]=x
    V=np.zeros((n,n,B,C,th,tw),dt)
    for ty in range(th):
        for tx in range(tw):
            d=xp[:,:,m*ty:m*ty+n,m*tx:m*tx+n]
            V[:,:,:,:,ty,tx]=matapply(BT,np.moveaxis(matapply(BT,np.moveaxis(d,3,0),dt),3,0),dt)
    M=np.zeros((n,n,B,N,th,tw),dt)
    for c in range(C):
        U=np.einsum('ik,nkl,jl->ijn',G,w[:,c].astype(np.float64),G).astype(dt)
        M=M+U[:,:,None,:,None,None]*V[:,:,:,c,None]
    y=matapply(AT,np.moveaxis(matapply(AT,M,dt),1,0),dt)
    out=np.zeros((B,N,H,W),dt)
    for i in range(m):
        for j in range(m): out[:,:,i::m,j::m]=y[j,i]
    return out
rng=np.random.default_rng(0)
B,C,H,W,N=1,64,32,48,32
x=rng.standard_normal((B,C,H,W)); w=rng.standard_normal((N,C,5,5))/np.sqrt(C*25)
ref=F.conv2d(torch.tensor(x),torch.tensor(w),padding=2).numpy()
d32=F.conv2d(torch.tensor(x).float(),torch.tensor(w).float(),padding=2).double().numpy()-ref
print("direct fp32 rms %.2e max %.2e"%(np.sqrt((d32**2).mean()),np.abs(d32).max()))
cands=[("0,+-1,+-2,+-.5,inf",[0,1,-1,2,-2,.5,-.5],True),("0,+-1,+-2,+-.5,4",[0,1,-1,2,-2,.5,-.5,4],False),("0,+-1,+-2,+-.5,.25?",[0,1,-1,2,-2,.5,-.5,.25],False),
("0,+-1,+-.5,+-1.5,inf",[0,1,-1,.5,-.5,1.5,-1.5],True),("0,+-1,+-2,1/2,-1/3?,inf",[0,1,-1,2,-2,.5,-1/3],True),("0,+-5/8,+-8/5,+-1,inf",[0,.625,-.625,1.6,-1.6,1,-1],True),
("0,+-.5,+-2,+-1,inf",[0,.5,-.5,2,-2,1,-1],True),("0,+-1,+-2,.5,-.25,inf",[0,1,-1,2,-2,.5,-.25],True),("0,+-.75,+-4/3,+-1? ,inf",[0,.75,-.75,4/3,-4/3,2,-.5],True)]
for name,pts,inf in cands:
    mats=cook_toom(4,5,pts,inf)
    o=run(x,w,mats,np.float64); assert np.abs(o-ref).max()<1e-7,(name,np.abs(o-ref).max())
    e=run(x.astype(f32),w.astype(f32),mats,np.float32)-ref
    print("%-26s rms %.2e max %.2e"%(name,np.sqrt((e**2).mean()),np.abs(e).max()),flush=True)
# 1-D F(4,5) reference (current kernel's form): along x only
mats=cook_toom(4,5,[0,1,-1,2,-2,.5,-.5],True)
AT,G,BT=mats
def run1d(x,w,dt):
    B,C,H,W=x.shape; N=w.shape[0]; tw=W//4
    xp=np.zeros((B,C,H+4,W+8),dt); xp[:,:,2:2+H,2:2+W]=x
    out=np.zeros((B,N,H,W),dt)
    M=np.zeros((8,B,N,H,tw),dt)
    for c in range(C):
        for ky in range(5):
            U=(G@w[:,c,ky,:].astype(np.float64).T).astype(dt)   # [8,N]
            d=np.stack([xp[:,c,ky:ky+H,j:j+4*tw:4] for j in range(8)],0)  # [8,B,H,tw]
            Vv=matapply(BT,d,dt)
            M=M+U[:,None,:,None,None]*Vv[:,:,None]
    y=matapply(AT,M,dt)
    for j in range(4): out[:,:,:,j::4]=y[j]
    return out
e=run1d(x.astype(f32),w.astype(f32),np.float32)-ref
print("1-D F(4,5) (current form)   rms %.2e max %.2e"%(np.sqrt((e**2).mean()),np.abs(e).max()))
