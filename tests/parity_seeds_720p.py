"""Study (GPU box, not collected by pytest): parity of the two split-precision modes against the CPU oracle at 736x1280 on
further synthetic pairs / t values than the bench uses.  r1l: f16f8 3.1e-4 ... 4.2e-4, f16x3 2.0e-4 ... 2.8e-4 (bar 1e-3);
r1q (two decoder levels in the sub-pixel form): f16f8 3.9e-4 ... 4.3e-4, f16x3 2.0e-4 ... 2.8e-4.

    python tests/parity_seeds_720p.py          (about 30 s of CPU oracle per pair)
"""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); PKG=os.path.join(ROOT,"superslomo-videointerpolation-pytorch_amd")
for p in (ROOT,PKG,os.path.join(PKG,"scripts")): sys.path.insert(0,p)
import torch
from ssm_amd.engine import PairEngine
from ssm_amd.weights import synthetic_frames, synthetic_state_dict
from oracle import ssm_oracle as O
dev=torch.device("cuda:0")
p1,p2=synthetic_state_dict(1),synthetic_state_dict(2)
sd1={k:v.to(dev) for k,v in p1.items()}; sd2={k:v.to(dev) for k,v in p2.items()}
H,W=736,1280
engs={m:PairEngine(sd1,sd2,1,2,H,W,dev,True,m) for m in ("f16f8","f16x3")}
for seed,ts in ((101,[0.125,0.875]),(202,[0.375,0.625]),(303,[0.25,0.75])):
    x=synthetic_frames(2,720,1280,seed=seed)
    img6=x.reshape(1,6,H,W)
    t0=time.time()
    want=torch.cat(O.interpolate_pair(p1,p2,img6,ts),0)
    dt=time.time()-t0
    t=torch.tensor(ts,device=dev)
    res={m:float((e.run(img6.to(dev),t,want_aux=False).cpu()-want).abs().max()) for m,e in engs.items()}
    print("seed %d t=%s: max-abs vs oracle %s  (oracle %.0f s)"%(seed,ts,{k:"%.2e"%v for k,v in res.items()},dt))
