"""Numerical study (CPU, not a test): how precise do the two error-compensation products of the split-fp16 convolution
(a_hi*b_lo + a_lo*b_hi) have to be?  The whole pair -> frame path is run on the oracle with its convolution replaced by an
emulation of: fp16 main product + correction products whose operands are quantised to block-scaled fp6 (e2m3, one E8M0 scale
per 32 channels - the operand format of v_mfma_scale_f32_32x32x64_f8f6f4) or block-scaled int8.

    python tests/emulate_correction_precision.py [size]     (default 128; result quoted in DESIGN.md 7)
"""
import sys, torch, math
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'superslomo-videointerpolation-pytorch_amd'))
import torch.nn.functional as F
from oracle import ssm_oracle as O
from ssm_amd.weights import synthetic_state_dict, synthetic_frames
torch.set_num_threads(8)
p1=synthetic_state_dict(1); p2=synthetic_state_dict(2)
H=W=int(sys.argv[1]) if len(sys.argv)>1 else 128
x=synthetic_frames(2,H,W,seed=42)
img6=torch.cat([x[:,0],x[:,1]],1)
ts=[0.125,0.5,0.875]
ref=torch.cat(O.interpolate_pair(p1,p2,img6,ts),0)
def split16(v):
    hi=v.to(torch.float16).to(torch.float32); lo=(v-hi).to(torch.float16).to(torch.float32); return hi,lo
def q_e2m3_block(v, dim, blk=32):
    """block-scaled (E8M0 scale per `blk` elements along `dim`) e2m3 quantisation, emulated in fp32"""
    sh=list(v.shape); C=sh[dim]
    pad=(-C)%blk
    vm=v.movedim(dim,-1)
    if pad: vm=F.pad(vm,(0,pad))
    vb=vm.reshape(*vm.shape[:-1],-1,blk)
    mx=vb.abs().amax(-1,keepdim=True).clamp_min(1e-30)
    sc=torch.exp2(torch.floor(torch.log2(mx))-2)          # max lands in [4,8)
    u=vb/sc
    a=u.abs()
    e=torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0,2)
    step=torch.where(a<1, torch.full_like(a,0.125), torch.exp2(e-3))
    q=(torch.round(a/step)*step).clamp(max=7.5)*torch.sign(u)
    out=(q*sc).reshape(*vm.shape)
    if pad: out=out[...,:C]
    return out.movedim(-1,dim)
def q_i8_block(v, dim, blk=32):
    vm=v.movedim(dim,-1); C=vm.shape[-1]; pad=(-C)%blk
    if pad: vm=F.pad(vm,(0,pad))
    vb=vm.reshape(*vm.shape[:-1],-1,blk)
    mx=vb.abs().amax(-1,keepdim=True).clamp_min(1e-30)
    q=torch.round(vb/mx*127)/127*mx
    out=q.reshape(*vm.shape)
    if pad: out=out[...,:C]
    return out.movedim(-1,dim)
MODE=None
def conv_emul(x,w,b):
    k=w.shape[-1]; pad=(k-1)//2
    wmax=float(w.abs().max()); sc=2.0**(3-math.ceil(math.log2(wmax)))
    ws=w*sc
    xh,xl=split16(x); wh,wl=split16(ws)
    y=F.conv2d(xh,wh,None,padding=pad)
    if MODE=='f16x3':
        y=y+F.conv2d(xl,wh,None,padding=pad)+F.conv2d(xh,wl,None,padding=pad)
    elif MODE=='e2m3':
        q=q_e2m3_block
        y=y+F.conv2d(q(xl,1),q(wh,1),None,padding=pad)+F.conv2d(q(xh,1),q(wl,1),None,padding=pad)
    elif MODE=='i8':
        q=q_i8_block
        y=y+F.conv2d(q(xl,1),q(wh,1),None,padding=pad)+F.conv2d(q(xh,1),q(wl,1),None,padding=pad)
    return y/sc+b.view(1,-1,1,1)
O.conv2d=conv_emul
for MODE in ('f16','f16x3','e2m3','i8'):
    out=torch.cat(O.interpolate_pair(p1,p2,img6,ts),0)
    d=(out-ref).abs()
    print(MODE, 'max-abs %.2e rms %.2e'%(float(d.max()), float(d.pow(2).mean().sqrt())))
