"""Randomised parity sweep (not collected by pytest): 320 seeded random-syntax streams decoded by the product and compared with the
oracle bit for bit. Host emulation by default, SWEEP_GPU=1 on an MI355X: python tests/sweep_decode.py"""
import sys; sys.path.insert(0,'tests')
import numpy as np, rbt_lib, oracle_lib as O
R=rbt_lib.module(); import os
ctx=R.Context(device=0) if os.environ.get("SWEEP_GPU") else R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
bad=0
for seed in range(100, 420):
    w=[64,96,128,80,144,160][seed%6]; h=[64,80,48,128,112,96][(seed//6)%6]
    bd=10 if seed%3 else 8; qp=[12,22,30,38,45][seed%5]
    fr=np.zeros((4,w*h*3//2),np.uint16)
    bs,rec=O.encode(fr,w,h,bd,qp=qp,gop=2,stress_seed=seed,log2_ctb=[0,4,5,6][seed%4])
    try:
        dec,dw,dh,dbd,chk,fail=ctx.decode(bs)
        ok=(dw,dh,dbd,fail)==(w,h,bd,0) and np.array_equal(dec,rec)
    except Exception as e:
        ok=False; print("seed",seed,"exc",e)
    if not ok: bad+=1; print("MISMATCH seed",seed,w,h,bd,qp)
print("done bad",bad)
