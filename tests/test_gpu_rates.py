"""GPU parity (through the C ABI) at every CTC rate point and for the multi-GOF walks of BASELINE.json configs[3] / configs[4]:
R1..R5 target QPs (cfg/rate/ctc-r{1..5}.cfg:5-11) incl. R5's occupancy pass-through (PCCTranscoder.cpp:150), a 12-frame tail GOF
(300 = 9 x 32 + 12), the GOF-sharded sequence walk and the rate fan-out with one decode per input - all bit-exact vs the oracle."""
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import synth

pytestmark = pytest.mark.gpu
gs = rbt_lib.module_file("gof_shard")


@pytest.fixture(scope="module")
def ctx():
    R = rbt_lib.module()
    c = R.Context(device=0)
    yield c
    c.close()


def _gof(w, h, n_pc, seed):
    geo, attr, occ = synth.make_gof(w, h, n_pc, seed)
    sg, _ = O.encode(geo, w, h, 10, 16, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    sa, _ = O.encode(attr, w, h, 10, 22, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    so, _ = O.encode(occ, w // 2, h // 2, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    return [so, sg, sa]


def _oracle(gof, rate):
    gq, aq, pr = gs.RATE_POINTS[rate]
    return O.transcode_data(gof, [(0, 8, pr, 5, gs.DEFAULT_ROWS, 0), (1, gq, pr, 5, gs.DEFAULT_ROWS, 0), (19, aq, pr, 5, gs.DEFAULT_ROWS, 0)])


@pytest.mark.parametrize("rate", [1, 2, 3, 4, 5])
def test_every_rate_point_vs_oracle(ctx, rate):
    R = rbt_lib.module()
    gof = _gof(192, 128, 2, 31)
    out = ctx.transcode_gof(gof, gs.rate_params(R, rate))
    assert out == _oracle(gof, rate)
    if rate == 5:
        assert out[0] == gof[0]      # occupancyPrecision 2: transcodeData leaves the occupancy sub-bitstream alone


def test_twelve_frame_tail_gof_vs_oracle(ctx):
    R = rbt_lib.module()
    gof = _gof(128, 128, 12, 7)       # 12 point-cloud frames: 24 geometry + 24 attribute + 12 occupancy pictures
    assert [len(gs.split_pairs(s)) for s in gof] == [12, 12, 12]
    assert ctx.transcode_gof(gof, gs.rate_params(R, 3)) == _oracle(gof, 3)


def test_sequence_walk_equals_per_gof_calls_and_oracle(ctx):
    """configs[3] on one GPU: 44 frames in GOFs of 16 -> 16 + 16 + 12, several GOFs in flight"""
    R = rbt_lib.module()
    seq = gs.make_sequence(_gof(128, 128, 16, 3), 44, 16)
    assert [len(gs.split_pairs(g[2])) for g in seq] == [16, 16, 12]
    out = gs.transcode_sequence(ctx, seq, gs.rate_params(R, 3), depth=3)
    assert out == [ctx.transcode_gof(g, gs.rate_params(R, 3)) for g in seq]
    assert out[2] == _oracle(seq[2], 3)


def test_rate_fanout_decodes_once_and_matches_oracle(ctx):
    """configs[4] on one GPU: all five rate points from one input in one call (identical input buffers are decoded once)"""
    R = rbt_lib.module()
    gof = _gof(192, 128, 2, 13)
    fan = gs.transcode_fanout(ctx, R, [gof], depth=1)
    for rate in (1, 2, 3, 4, 5):
        assert fan[rate] == [_oracle(gof, rate)]
    # the same through separate calls
    assert fan[2][0] == ctx.transcode_gof(gof, gs.rate_params(R, 2))
