import os, sys, numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
ROOT = os.getcwd()
from neural_photo_editor_amd import IAN, synthetic as S
def rel(a,b):
    a,b=np.asarray(a,np.float64),np.asarray(b,np.float64); return float(np.abs(a-b).max()/(np.abs(b).max()+1e-30))
rgb=np.full((1,3,64,64),-1.0,np.float32); rgb[:,0]=1.0
for arch in ("IAN_simple","IAN"):
    fx=np.load(os.path.join(ROOT,"tests","golden","ref_%s.npz"%arch))
    z=fx["z_sample"][:1]
    for fwd,bwd,mm in ((0,0,1),(1,0,1),(0,1,1),(1,1,1),(1,1,256)):
        m=IAN(os.path.join(ROOT,"neural_photo_editor_amd","configs",arch+".py"),True,params=S.make_params(arch,1))
        h=m.handle
        h.set_option("tg_bf16x3",1); h.set_option("tg_bf16x3_min_m",mm); h.set_option("tg_bf16x3_fwd",fwd); h.set_option("tg_bf16x3_bwd",bwd)
        errs=[]
        for k,(c1,r1,c2,r2) in enumerate(fx["patches"].tolist()):
            errs.append((round(rel(m.imgradRGB(c1,r1,c2,r2,rgb,z),fx["grad_rgb_%d"%k]),8), round(rel(m.imgrad(c1,r1,c2,r2,z),fx["grad_light_%d"%k]),8)))
        xs=rel(m.sample_at(fx["z_sample"]),fx["x_sample"])
        print(arch,"fwd",fwd,"bwd",bwd,"min_m",mm,"x_sample",xs,"grads",errs,flush=True)
        m.close()
