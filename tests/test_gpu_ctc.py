"""GPU parity on the stream structure of the real CTC inputs ("a real longdress_r5.bin ... accepted unchanged"): one IDR per sub-bitstream, then
TRAIL_N P pictures and TRAIL_R intra pictures with reference picture sets, POC 0..63 with a wrapping lsb, parameter sets at the IDR only
(cfg/hm/ctc-hm-geometry-ai.cfg:21-30; tests/ctc_cases.py). Through the C ABI of librbt.so against the oracle: random-syntax streams, small GOFs in both NAL type
variants, the full-size fixture tests/golden/hm_r5ctc_1280x1280_f32_*.annexb through the hash SEI of its 128 pictures, its transcode, and a V3C container of it."""
import json
import os
import numpy as np
import pytest
import ctc_cases as CC
import oracle_lib as O
import rbt_lib
import v3c_synth as V

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def R():
    return rbt_lib.module()


@pytest.fixture(scope="module")
def gs():
    return rbt_lib.module_file("gof_shard")


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(device=0)
    yield c
    c.close()


def _fixture(kind):
    man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
    return open(os.path.join(GOLD, man["streams"][kind]["file"]), "rb").read()


@pytest.mark.parametrize("seed", CC.STRESS_SEEDS)
def test_decode_random_syntax_streams_in_the_ctc_structure(ctx, seed):
    CC.check_decode_stress(ctx, seed)


@pytest.mark.parametrize("w,h,n_pc,ctc_gop", [(128, 128, 10, 1), (128, 128, 10, 2), (320, 256, 9, 1)])
def test_gof_in_the_ctc_structure_decodes_and_transcodes_like_the_oracle(ctx, R, gs, w, h, n_pc, ctc_gop):
    CC.check_gof(ctx, R, gs, w, h, n_pc, 31 + w, ctc_gop)


@pytest.mark.parametrize("kind,pics", [("geo", 64), ("attr", 64)])
def test_ctc_fixture_full_size_hash_sei(ctx, gs, kind, pics):
    """64 pictures of 1280x1280, ONE IDR: every decoded picture's MD5 equals the hash the (oracle) encoder put into the stream; POC runs 0..63 over 5 lsb bits"""
    s = _fixture(kind)
    hd = [x for x in O.slice_headers(s) if x["address"] == 0]
    assert [x["nal_type"] for x in hd] == [19] + [0, 1] * 31 + [0] and [x["poc"] for x in hd] == list(range(64)) and not gs.is_closed_pairs(s)
    dec, w, h, bd, chk, fail = ctx.decode(s, verify_md5=True)
    assert (w, h, bd, dec.shape[0]) == (1280, 1280, 10, pics) and chk == pics and fail == 0


def test_ctc_fixture_is_the_closed_fixture_in_another_structure(ctx):
    """same maps, same decisions: the pictures of the two fixtures are identical (the structures differ in headers only), the occupancy maps lossless"""
    man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32"]
    for kind in ("occ", "geo"):
        closed = open(os.path.join(GOLD, man["streams"][kind]["file"]), "rb").read()
        assert np.array_equal(ctx.decode(_fixture(kind))[0], ctx.decode(closed)[0])


def test_ctc_fixture_first_frames_transcode_vs_oracle(ctx, R, gs):
    """the first two point-cloud frames of the fixture (a prefix of each sub-bitstream: IDR, TRAIL_N, TRAIL_R, TRAIL_N) through the whole path, R5 -> R3, vs the oracle"""
    gof = [gs.first_pictures(_fixture(k), n) for k, n in (("occ", 2), ("geo", 4), ("attr", 4))]
    out = ctx.transcode_gof(gof, gs.rate_params(R, 3))
    assert out == O.transcode_data(gof, [(0, 8, 4, 5, gs.DEFAULT_ROWS, 0), (1, 24, 4, 5, gs.DEFAULT_ROWS, 0), (19, 32, 4, 5, gs.DEFAULT_ROWS, 0)])


def test_ctc_fixture_whole_gof_equals_its_frames_transcoded_alone(ctx, R, gs):
    """size-independent property at full size: the re-encoder's output is closed (IDR, P) pairs, and the pictures of a point-cloud frame reference nothing outside it, so the
    transcode of the whole 32-frame GOF is the concatenation of the transcodes of its frames handed over one by one (each with the parameter sets in front)"""
    gof = [_fixture(k) for k in ("occ", "geo", "attr")]
    params = gs.rate_params(R, 3)
    whole = ctx.transcode_gof(gof, params)
    pieces = [gs.frame_pieces(s, n) for s, n in zip(gof, (1, 2, 2))]
    ctx.set_depth(4)
    try:
        jobs = [ctx.submit_gof([p[k] for p in pieces], params) for k in (0, 1, 17, 31)]
        outs = [ctx.wait_gof(j) for j in jobs]
    finally:
        ctx.set_depth(1)
    for k, o in zip((0, 1, 17, 31), outs):
        for s in range(3):
            assert o[s] == gs.frame_pieces(whole[s], (1, 2, 2)[s])[k], (k, s)


def test_ctc_fixture_in_a_container(ctx, R, gs):
    """rbt_transcode_v3c on a container built from the fixture (first two frames: the oracle finishes them in seconds), and on a 2-GOF sequence (2 + 1 frames)"""
    gof = [gs.first_pictures(_fixture(k), n) for k, n in (("occ", 2), ("geo", 4), ("attr", 4))]
    seq = gs.make_sequence(gof, 3, 2)
    units = []
    for g, s in enumerate(seq):
        units += V.gof_units(s, 70 + g)
    data = V.sample_stream(units, 4)
    got = ctx.transcode_v3c(data, 24, 32)
    assert got == O.v3c_transcode(data, 24, 32, 4) and len(got) < len(data) // 2


@pytest.mark.parametrize("seed", CC.WP_SEEDS)
def test_decode_streams_with_weighted_prediction(ctx, seed):
    """explicit weighted sample prediction of P slices (libx265's output from preset "veryfast" up as an input: PPS weighted_pred_flag, pred_weight_table) on the GPU:
    the weights applied to the 14-bit intermediate of rc_mc_plane, integer and fractional vectors, luma and both chroma planes == oracle"""
    CC.check_decode_wp(ctx, seed)


def test_transcode_of_a_weighted_prediction_stream(ctx, R):
    bs, rec, w, h, bd, n = CC.wp_case(12)
    assert ctx.transcode_substream(bs, R.RBT_VIDEO_GEOMETRY, 30, log2_ctb=5, rows_per_slice=-1) == O.transcode_substream(bs, 1, 30, 4, 5, -1)
