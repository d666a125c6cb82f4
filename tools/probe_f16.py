import sys, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from diffusion_models_dsdiff_amd import ops
from util import rel_l2
g=torch.Generator().manual_seed(1)
for (N,H,W,Cin,Cout,ks) in [(2,16,16,64,64,3),(1,32,32,320,320,3),(2,8,8,960,480,1),(2,64,64,128,96,3)]:
    x=torch.randn(N,Cin,H,W,generator=g); w=torch.randn(Cout,Cin,ks,ks,generator=g)/(Cin*ks*ks)**0.5; b=torch.randn(Cout,generator=g)
    ref=F.conv2d(x.double(),w.double(),b.double(),padding=ks//2)
    out={}
    for prec in ("f32","bf16x6","f16x3","bf16x3"):
        for st in (("staged","adirect") if prec!="f32" else ("auto",)):
            y=ops.conv2d(ops.to_nhwc(x).cuda(),w.cuda(),b.cuda(),precision=prec,structure=st)
            out[prec+":"+st]=rel_l2(ops.to_nchw(y),ref)
    print((N,H,W,Cin,Cout,ks),{k:f"{v:.2e}" for k,v in out.items()})
# small / large magnitudes
x=torch.randn(1,64,8,8,generator=g); x[:,:16]*=1e-3; x[:,16:32]*=1e3; w=torch.randn(32,64,3,3,generator=g)/24
ref=F.conv2d(x.double(),w.double(),None,padding=1)
for prec in ("bf16x6","f16x3"):
    y=ops.conv2d(ops.to_nhwc(x).cuda(),w.cuda(),None,precision=prec); print("mixed magnitudes",prec,f"{rel_l2(ops.to_nchw(y),ref):.2e}")
