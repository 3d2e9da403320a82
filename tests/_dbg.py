import sys; sys.path.insert(0,'tests')
import numpy as np, oracle_lib as O, rbt_lib
R=rbt_lib.module(); ctx=R.Context(device=0)
for seed in (1,2,3,4):
    w=[64,96,128,80][seed%4]; h=[64,80,48,128][(seed//4)%4]; bd=10 if seed%3 else 8
    fr=np.zeros((5,w*h*3//2),np.uint16)
    bs,rec=O.encode(fr,w,h,bd,qp=30,gop=2,stress_seed=seed,log2_ctb=0)
    dec,*_=ctx.decode(bs,verify_md5=False)
    for i in range(5):
        d=dec[i]!=rec[i]
        if d.any():
            Y=d[:w*h].reshape(h,w); ys,xs=np.nonzero(Y)
            print("seed",seed,"frame",i,"mismatch Y",int(Y.sum()),"C",int(d[w*h:].sum()), "first Y",(int(xs[0]),int(ys[0])) if len(xs) else None, "bbox", (int(xs.min()),int(ys.min()),int(xs.max()),int(ys.max())) if len(xs) else None)
            if len(xs): 
                x0,y0=int(xs[0])//4*4,int(ys[0])//4*4
                print(" gpu",dec[i][:w*h].reshape(h,w)[y0:y0+4,x0:x0+8].tolist()); print(" ref",rec[i][:w*h].reshape(h,w)[y0:y0+4,x0:x0+8].tolist())
        else: print("seed",seed,"frame",i,"ok")
