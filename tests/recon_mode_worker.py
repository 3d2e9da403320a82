"""Child process of the reconstruction-mode tests (the mode is read once per process: RBT_RECON_QUEUE / RBT_RECON_LEVEL / RBT_RECON_DIAG): decode and transcode cases
against the oracle in whatever mode the environment selects. argv[1]: "hostemu" or "gpu". Prints OK <cases>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import ctc_cases as CC
import oracle_lib as O
import rbt_lib
import synth

R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
gpu = sys.argv[1] == "gpu"
ctx = R.Context(device=0) if gpu else R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
n = 0
for seed in (3, 8, 14, 21):                                   # random-syntax streams: several slices per picture, uncovered CTBs never occur, P pictures two levels deep
    CC.check_decode_stress(ctx, seed); n += 1
for w, h, log2_ctb in ((64, 64, 6), (80, 48, 4), (256, 192, 5), (16, 128, 4), (200, 16, 4)) + (((1280, 1280, 6),) if gpu else ()):      # one CTB, one CTB column, one CTB row, many
    m = synth.make_maps(w - w % 16, h - h % 16, 7) if (w % 16 or h % 16) else synth.make_maps(w, h, 7)
    ww, hh = (w - w % 16, h - h % 16)
    bs, rec = O.encode(m["attr"], ww, hh, 10, 22, gop=2, log2_ctb=log2_ctb, rows_per_slice=0)
    dec, *_r = ctx.decode(bs)
    assert _r[-1] == 0 and np.array_equal(dec, rec), (w, h)
    n += 1
for depth in (1, 4, 16):                                      # merged launches (pipelines that share a stream) and jobs side by side
    ctx.set_depth(depth)
    streams, _ = CC.hm_gof(128, 128, 3, 40 + depth, 1)
    want = O.transcode_data(streams, [(0, 8, 4, 5, -1, 0), (1, 24, 4, 5, -1, 0), (19, 32, 4, 5, -1, 0)])
    jobs = [ctx.submit_gof(streams, gs.rate_params(R, 3)) for _ in range(min(depth, 6))]
    assert all(ctx.wait_gof(j) == want for j in jobs)
    n += 1
print("OK", n)
