"""The stream structure of the real CTC inputs (one IDR, then TRAIL_N / TRAIL_R pictures with reference picture sets and POC running on; tests/ctc_cases.py) through
the product's host code and kernel bodies in the test-only host build (the GPU run of the same checks: tests/test_gpu_ctc.py), and through the helpers around the path
(access units, sequences, the V3C walk)."""
import os
import subprocess
import numpy as np
import pytest
import ctc_cases as CC
import oracle_lib as O
import rbt_lib
import v3c_synth as V


@pytest.fixture(scope="module")
def R():
    return rbt_lib.module()


@pytest.fixture(scope="module")
def gs():
    return rbt_lib.module_file("gof_shard")


@pytest.fixture(scope="module")
def ctx(R):
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    yield c
    c.close()


@pytest.mark.parametrize("seed", CC.STRESS_SEEDS)
def test_decode_random_syntax_streams_in_the_ctc_structure(ctx, seed):
    CC.check_decode_stress(ctx, seed)


@pytest.mark.parametrize("ctc_gop", [1, 2])
def test_gof_in_the_ctc_structure_decodes_and_transcodes_like_the_oracle(ctx, R, gs, ctc_gop):
    CC.check_gof(ctx, R, gs, 128, 128, 10, 31, ctc_gop)          # 20 pictures, 4 lsb bits: POC wraps


def test_poc_runs_on_through_trail_n_pictures_and_lsb_wraps(ctx):
    """40 pictures with 4 lsb bits: the product's host parser (slice_poc, host/rbt_hls.cpp: TRAIL_N pictures leave the POC anchor alone, 8.3.1) gives POC 0..39, and
    the NAL unit types are the CTC encoder's (IDR, then TRAIL_N P / TRAIL_R I). The values of the reference's own parser for such streams: test_slice_headers.py."""
    streams, _ = CC.hm_gof(64, 64, 20, 5, 1, 4)
    L = __import__("ctypes").CDLL(rbt_lib.HOSTEMU_LIB)
    L.rbt_hostemu_slice_headers.argtypes = O.lib().oracle_slice_headers.argtypes
    for s in streams[1:]:
        assert [x["poc"] for x in O.slice_headers(s, L.rbt_hostemu_slice_headers)] == list(range(40))
        assert [x["nal_type"] for x in O.slice_headers(s)][:5] == [19, 0, 1, 0, 1]


def test_access_unit_helpers(gs):
    (so, sg, sa), _ = CC.hm_gof(64, 64, 4, 9, 1, 4)
    closed = V.gof_streams(64, 64, 4, 9)
    assert [len(gs.access_units(s)) for s in (so, sg, sa)] == [4, 8, 8] and [len(gs.access_units(s)) for s in closed] == [4, 8, 8]
    assert all(b"".join(gs.access_units(s)) == s for s in (so, sg, sa) + tuple(closed))
    assert not gs.is_closed_pairs(sg) and gs.is_closed_pairs(closed[1])
    assert gs.parameter_sets(sg) and sg.startswith(gs.parameter_sets(sg)) and gs.access_units(sg)[0].startswith(gs.parameter_sets(sg))
    # a prefix of a stream is a stream; the pieces of a CTC stream decode on their own once they carry the parameter sets
    whole = O.decode(sg)[0]
    assert np.array_equal(O.decode(gs.first_pictures(sg, 4))[0], whole[:4])
    pieces = gs.frame_pieces(sg, 2)
    assert len(pieces) == 4 and all(np.array_equal(O.decode(p)[0], whole[2 * k:2 * k + 2]) for k, p in enumerate(pieces))
    assert gs.frame_pieces(closed[1], 2) == gs.split_pairs(closed[1])


def test_sequences_of_ctc_gofs_are_prefixes_and_closed_ones_still_permute(gs):
    streams, _ = CC.hm_gof(64, 64, 4, 9, 1, 4)
    seq = gs.make_sequence(streams, 10, 4)                           # 4 + 4 + 2 frames
    assert [[len(gs.access_units(s)) for s in g] for g in seq] == [[4, 8, 8], [4, 8, 8], [2, 4, 4]]
    assert seq[0] == list(streams) and all(s.startswith(t) for s, t in zip(streams, seq[2]))
    with pytest.raises(ValueError):
        gs.make_sequence(streams, 10, 3)                             # 8 pictures do not hold groups of 3 frames
    with pytest.raises(ValueError):
        gs.make_sequence([V.gof_streams(64, 64, 4, 9)[0]] + list(streams[1:]), 10, 4)      # mixed structures
    with pytest.raises(ValueError):
        gs.split_pairs(streams[1][5:])


def test_v3c_walk_over_ctc_gofs_equals_the_oracle(ctx, R, gs):
    """rbt_transcode_v3c on a container whose GOFs are in the CTC structure (3 GOFs: 3 + 3 + 1 frames), at two depths, sharded and merged"""
    streams, _ = CC.hm_gof(64, 64, 3, 21, 1, 4)
    seq = gs.make_sequence(streams, 7, 3)
    units = []
    for g, s in enumerate(seq):
        units += V.gof_units(s, 50 + g)
    data = V.sample_stream(units, 3)
    want = O.v3c_transcode(data, 24, 32, 4)
    for depth, per in ((1, 1), (4, 2)):
        ctx.set_depth(depth)
        assert ctx.transcode_v3c(data, 24, 32, gofs_per_job=per) == want
    parts = []
    for r in range(2):
        c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB, rank=r, world=2)
        parts.append(c.transcode_v3c(data, 24, 32)); c.close()
    assert gs.merge_v3c(R, parts, lib=ctx.L) == want


@pytest.mark.parametrize("seed", CC.WP_SEEDS)
def test_decode_streams_with_weighted_prediction(ctx, seed):
    """what libx265 writes from its preset "veryfast" up (the value the reference's scripts pass): P slices with explicit weighted sample prediction. Round 3 refused the
    PPS flag; now the table is parsed on the host (pinned through the reference's parser: tests/test_slice_headers.py wp_*) and applied in the motion compensation
    (rc_mc_plane); == the oracle's decoder == the generating encoder's reconstruction, hash SEI of every picture green"""
    CC.check_decode_wp(ctx, seed)


def test_transcode_of_a_weighted_prediction_stream(ctx, R):
    bs, rec, w, h, bd, n = CC.wp_case(12)      # 128 x 48, 10 bit: the coded size is the decoded size (multiples of 16)
    assert ctx.transcode_substream(bs, R.RBT_VIDEO_GEOMETRY, 30, log2_ctb=5, rows_per_slice=-1) == O.transcode_substream(bs, 1, 30, 4, 5, -1)
