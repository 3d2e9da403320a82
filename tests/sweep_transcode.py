"""Randomised parity sweep (not collected by pytest): random sizes / CTB sizes / slice structures / QPs through rbt_transcode_gof vs the
oracle. Host emulation by default, SWEEP_GPU=1 on an MI355X: python tests/sweep_transcode.py"""
import sys, os; sys.path.insert(0,'tests')
import numpy as np, rbt_lib, oracle_lib as O, synth
R=rbt_lib.module()
ctx=R.Context(device=0) if os.environ.get("SWEEP_GPU") else R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
bad=0
r=np.random.default_rng(7)
for it in range(int(os.environ.get("SWEEP_N","36"))):
    w=int(r.choice([64,96,128,192,256])); h=int(r.choice([64,96,128,160]))
    n_pc=int(r.choice([1,2])); seed=int(r.integers(1,10000))
    geo,attr,occ=synth.make_gof(w,h,n_pc,seed) if (w%32==0 and h%32==0) else (None,None,None)
    if geo is None: continue
    lc=int(r.choice([4,5,6])); rows=int(r.choice([0,1,2,-1,-1])); qg=int(r.choice([20,24,28,32,40])); qa=int(r.choice([27,32,37,42]))
    lcin=int(r.choice([4,5,6])); rin=int(r.choice([0,1,3,-1]))
    sg,_=O.encode(geo,w,h,10,16,gop=2,log2_ctb=lcin,rows_per_slice=rin)
    sa,_=O.encode(attr,w,h,10,22,gop=2,log2_ctb=lcin,rows_per_slice=rin)
    so,_=O.encode(occ,w//2,h//2,8,8,gop=1,lossless=1,i_qp_offset=0,log2_ctb=lcin,rows_per_slice=rin)
    P=R.StreamParams
    outs=ctx.transcode_gof([so,sg,sa],[P(0,8,4,lc,rows,1,0),P(1,qg,4,lc,rows,1,0),P(19,qa,4,lc,rows,1,0)])
    exp=[O.transcode_substream(so,0,8,log2_ctb=lc,rows_per_slice=rows),O.transcode_substream(sg,1,qg,log2_ctb=lc,rows_per_slice=rows),O.transcode_substream(sa,19,qa,log2_ctb=lc,rows_per_slice=rows)]
    ok=all(a==b for a,b in zip(outs,exp))
    if not ok: bad+=1; print("MISMATCH",it,w,h,n_pc,seed,lc,rows,qg,qa,lcin,rin,[a==b for a,b in zip(outs,exp)])
print("done bad",bad)
