"""GPU decode / transcode of streams coded with the CTC toolset (the oracle's HM-like encoder: 35 intra modes + NxN, TU split, 4x4 transform
skip, AMP, merge / AMVP with quarter-sample vectors, TMVP, sign data hiding, per-CTB SAO), bit-exact vs the oracle, and the full-size benchmark
fixture (tests/golden/hm_r5_1280x1280_f32_*.annexb) through its decoded-picture-hash SEI."""
import json
import os
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    R = rbt_lib.module()
    c = R.Context(device=0)
    yield c
    c.close()


@pytest.mark.parametrize("w,h,seed", [(320, 256, 77), (256, 192, 41), (192, 320, 5)])
def test_hm_like_streams_decode_and_transcode_like_the_oracle(ctx, w, h, seed):
    R = rbt_lib.module()
    m = synth.make_maps(w, h, seed)
    for key, qp, po, vt, tq in (("geo", 16, -3, R.RBT_VIDEO_GEOMETRY, 24), ("attr", 22, 0, R.RBT_VIDEO_ATTRIBUTE, 32)):
        bs, rec = O.encode_hm(m[key], w, h, 10, qp, p_qp_offset=po)
        dec, dw, dh, bd, chk, fail = ctx.decode(bs)
        assert (dw, dh, bd, chk, fail) == (w, h, 10, 2, 0) and np.array_equal(dec, rec)
        assert ctx.transcode_substream(bs, vt, tq) == O.transcode_substream(bs, vt, tq)
    bs, rec = O.encode_hm(m["occ"], w // 2, h // 2, 8, 8, gop=1, i_qp_offset=0, lossless=1)
    assert np.array_equal(ctx.decode(bs)[0], rec)
    assert ctx.transcode_substream(bs, R.RBT_VIDEO_OCCUPANCY, 8) == O.transcode_substream(bs, 0, 8)


def _fixture(kind):
    man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32"]
    return open(os.path.join(GOLD, man["streams"][kind]["file"]), "rb").read()


@pytest.mark.parametrize("kind", ["geo", "attr"])
def test_fixture_full_size_hash_sei(ctx, kind):
    """64 pictures of 1280x1280 each: every decoded picture's MD5 equals the hash the (oracle) encoder put into the stream"""
    dec, w, h, bd, chk, fail = ctx.decode(_fixture(kind), verify_md5=True)
    assert (w, h, bd, dec.shape[0]) == (1280, 1280, 10, 64) and chk == 64 and fail == 0


def test_fixture_first_frame_transcode_vs_oracle(ctx):
    """the first point-cloud frame of the fixture (one closed GOP per sub-bitstream) through the whole path, R5 -> R3, vs the oracle"""
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    gof = [gs.split_pairs(_fixture(k))[0] for k in ("occ", "geo", "attr")]
    out = ctx.transcode_gof(gof, gs.rate_params(R, 3))
    assert out == O.transcode_data(gof, [(0, 8, 4, 5, gs.DEFAULT_ROWS, 0), (1, 24, 4, 5, gs.DEFAULT_ROWS, 0), (19, 32, 4, 5, gs.DEFAULT_ROWS, 0)])
