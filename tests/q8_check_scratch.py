import sys, os
ROOT="/root/repo"; PKG=os.path.join(ROOT,"superslomo-videointerpolation-pytorch_amd")
for p in (ROOT,PKG,os.path.join(PKG,"scripts")): sys.path.insert(0,p)
import torch
from ssm_amd import hipbind as hb
from oracle import ssm_oracle as O
dev=torch.device("cuda:0")
torch.manual_seed(0)
def run(k,cin,cout,B,H,W,q8,pool=False):
    w=torch.randn(cout,cin,k,k)/(cin*k*k)**0.5; b=torch.randn(cout)*0.1
    x=torch.randn(B,cin,H,W)
    want=O.conv2d_lrelu(x,w,b)
    pk=hb.PackedConv16(w.to(dev),b.to(dev),W,q8=q8)
    xp=hb.HPlanes(B,cin,H,W,dev,groups=pk.cin_p//8,q8=q8).load(x.to(dev))
    # round trip check of the layout conversion
    rt=float((xp.to_nchw().cpu()-x).abs().max())
    yp=hb.HPlanes(B,cout,H,W,dev,q8=q8)
    pp=hb.HPlanes(B,cout,H//2,W//2,dev,q8=q8) if pool else None
    hb.conv2d_hl8(xp.view(),pk.cin_p,None,0,pk,yp.view(),None,pp.view() if pool else None,B,H,W,lrelu=True)
    torch.cuda.synchronize()
    got=yp.to_nchw().cpu()
    err=float((got-want).abs().max())
    perr=float((pp.to_nchw().cpu()-O.avg_pool2(want)).abs().max()) if pool else 0.0
    return rt,err,perr,float(want.abs().max())
for (k,cin,cout,B,H,W,pool) in [(3,128,128,2,24,70,False),(3,64,64,1,40,64,True),(3,32,32,1,32,96,False),(3,256,256,1,12,40,False),(5,64,64,2,24,64,True),(7,32,32,1,24,64,True),(7,16,32,1,16,40,False),(3,64,32,2,16,64,False),(3,512,256,1,8,24,False)]:
    for q8 in (False,True):
        rt,err,perr,mx=run(k,cin,cout,B,H,W,q8,pool)
        print("k%d %3d->%3d B%d %dx%d %s  roundtrip %.1e  conv err %.2e  pool err %.2e (max %.1f)"%(k,cin,cout,B,H,W,"Q8 " if q8 else "x3 ",rt,err,perr,mx))

def run_ups(ca,cb,cout,B,h,w,q8):
    wt=torch.randn(cout,ca+cb,3,3)/((ca+cb)*9)**0.5; bs=torch.randn(cout)*0.1
    a=torch.randn(B,ca,h,w); b=torch.randn(B,cb,h,w) if cb else None
    u=O.upsample2x_bilinear(torch.cat([a,b],1) if cb else a)
    want=O.conv2d_lrelu(u,wt,bs)
    pk=hb.PackedConv16(wt.to(dev),bs.to(dev),2*w,q8=q8,ups=True)
    ap=hb.HPlanes(B,ca,h,w,dev,q8=q8).load(a.to(dev))
    bp=hb.HPlanes(B,cb,h,w,dev,q8=q8).load(b.to(dev)) if cb else None
    yp=hb.HPlanes(B,cout,2*h,2*w,dev,q8=q8)
    hb.conv2d_ups_hl8(ap.view(),ca,bp.view() if cb else None,cb,pk,yp.view(),None,B,2*h,2*w,lrelu=True)
    torch.cuda.synchronize()
    return float((yp.to_nchw().cpu()-want).abs().max()), float(want.abs().max())
for (ca,cb,cout,B,h,w) in [(64,64,32,1,16,40),(128,128,64,2,12,32),(256,256,128,1,8,24),(512,512,256,1,6,10),(512,0,512,1,4,6),(64,64,32,1,9,17)]:
    for q8 in (False,True):
        err,mx=run_ups(ca,cb,cout,B,h,w,q8)
        print("ups %d+%d->%d B%d %dx%d %s err %.2e (max %.1f)"%(ca,cb,cout,B,h,w,"Q8" if q8 else "x3",err,mx))
