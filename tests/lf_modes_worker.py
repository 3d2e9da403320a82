"""Worker of the loop-filter mode tests: the library reads RBT_FUSED_LF (decoder: deblocking + SAO in one launch through LDS tiles) and RBT_FUSED_ENC_LF (encoder: deblocking
inside the SAO kernel) once per process, so every mode needs a process of its own. Transcodes HM-like geometry / attribute streams (SAO, deblocking, NxN, transform trees) and a
wavefront stream with this library and compares every output with the CPU oracle's; prints "ok <n>" and exits 0 if all are identical. argv[1]: "hostemu" (CPU build of the
kernel bodies) or "gpu"."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
import rbt_lib
import synth

R = rbt_lib.module()
ctx = R.Context(lib_path=rbt_lib.HOSTEMU_LIB) if sys.argv[1] == "hostemu" else R.Context(device=0)
n = 0
for (w, h, seed) in ((192, 128, 9), (208, 144, 4)):          # the second size is no multiple of the 64x64 filter tile or of the CTB
    m = synth.make_maps(w - w % 16, h - h % 16, seed); w, h = w - w % 16, h - h % 16
    for key, vt, q0, q1 in (("geo", R.RBT_VIDEO_GEOMETRY, 16, 24), ("attr", R.RBT_VIDEO_ATTRIBUTE, 22, 32)):
        bs, rec = O.encode_hm(m[key], w, h, 10, q0)
        dec, dw, dh, dbd, chk, fail = ctx.decode(bs)
        assert (dw, dh, fail) == (w, h, 0) and np.array_equal(dec, rec), "decode"
        for ctb, rows in ((5, -1), (6, 0), (4, 1)):
            out = ctx.transcode_substream(bs, vt, q1, log2_ctb=ctb, rows_per_slice=rows, md5_sei=0)
            assert out == O.transcode_substream(bs, int(vt), q1, 4, ctb, rows, 0), ("transcode", key, ctb, rows)
            n += 1
fr = np.random.default_rng(5).integers(0, 1024, size=(4, 200 * 120 * 3 // 2), dtype=np.uint16)
a, ra = O.encode(fr, 200, 120, 10, 30, gop=2, log2_ctb=5, rows_per_slice=-1)
assert ctx.encode(fr, 200, 120, 10, 30, gop=2, log2_ctb=5, rows_per_slice=-1) == a and np.array_equal(ctx.decode(a)[0], ra), "noise"
ctx.close()
print("ok", n + 1)
