"""Picture sizes that are not multiples of the minimum CU size: the coded size is the display size rounded up (8 for
all-intra, 16 for I,P pairs), padded by repetition, and the padding is signalled as the conformance window (7.4.3.2.1),
which is what libx265 does behind PCCTranscoder.cpp:706. Real atlases hit this: a 1280x1296 atlas has a 640x648 occupancy
map whose 2x2 pool is 320x324. Host emulation here (CPU), the same cases on the GPU in test_gpu_transcode.py."""
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import synth


def cases(ctx, R):
    r = np.random.default_rng(11)
    # encoder: lossless all-intra and lossy I,P pairs at odd sizes, bit-exact streams; decoder output is the cropped picture
    for (w, h, gop, ll, bd, qp) in ((72, 44, 1, 1, 8, 8), (40, 44, 1, 1, 8, 8), (100, 60, 2, 0, 10, 30), (72, 84, 2, 0, 10, 24)):
        fr = r.integers(0, 2 if ll else 1 << bd, (4, w * h * 3 // 2)).astype(np.uint16)
        if not ll: fr[1] = np.clip(fr[0].astype(int) + r.integers(-2, 3, fr[0].shape), 0, (1 << bd) - 1); fr[3] = fr[2]
        a = ctx.encode(fr, w, h, bd, qp, gop=gop, lossless=ll, log2_ctb=5, rows_per_slice=1)
        b, rec = O.encode(fr, w, h, bd, qp, gop=gop, lossless=ll, i_qp_offset=0 if ll else -3, log2_ctb=5, rows_per_slice=1)
        assert a == b
        dec, dw, dh, dbd, chk, fail = ctx.decode(a)
        assert (dw, dh, dbd, chk, fail) == (w, h, bd, 4, 0) and np.array_equal(dec, rec)
        if ll: assert np.array_equal(dec, fr)
    # transcoder: 80x88 occupancy -> 40x44 pooled (coded 40x48); geometry whose input already carries a window
    geo, attr, occ = synth.make_gof(160, 176, 2, 5)
    so, _ = O.encode(occ, 80, 88, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=6, rows_per_slice=0)
    out = ctx.transcode_substream(so, R.RBT_VIDEO_OCCUPANCY, 8)
    assert out == O.transcode_substream(so, 0, 8)
    dec, dw, dh, dbd, chk, fail = ctx.decode(out)
    assert (dw, dh, fail) == (40, 44, 0)
    want = (occ[:, :80 * 88].reshape(2, 44, 2, 40, 2).max(axis=(2, 4)) > 0).astype(np.uint16)
    assert np.array_equal(dec[:, :40 * 44].reshape(2, 44, 40), want)
    fr = np.random.default_rng(3).integers(0, 1024, (4, 152 * 104 * 3 // 2)).astype(np.uint16); fr[1] = fr[0]; fr[3] = fr[2]
    sg, _ = O.encode(fr, 152, 104, 10, 16, gop=2, log2_ctb=6, rows_per_slice=0)          # coded 160x112, window 8 / 8
    assert ctx.transcode_substream(sg, R.RBT_VIDEO_GEOMETRY, 24) == O.transcode_substream(sg, 1, 24)
    # transcoding the transcoder's own windowed output again
    once = ctx.transcode_substream(sg, R.RBT_VIDEO_GEOMETRY, 24)
    assert ctx.transcode_substream(once, R.RBT_VIDEO_GEOMETRY, 32, verify_md5=1) == O.transcode_substream(once, 1, 32)


def test_conformance_window_hostemu():
    R = rbt_lib.module()
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    try: cases(c, R)
    finally: c.close()


@pytest.mark.gpu
def test_conformance_window_gpu():
    R = rbt_lib.module()
    c = R.Context(device=0)
    try: cases(c, R)
    finally: c.close()
