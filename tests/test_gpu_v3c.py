"""SURVEY.md 8 row F3 on the GPU: rbt_transcode_v3c (the V3C sample stream walk of PccAppTranscoder.cpp:277-349 around transcodeData) through the C ABI
of librbt.so == the oracle's restatement, on synthetic containers (tests/v3c_synth.py) and on the full-size fixture GOF wrapped into one."""
import os
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import v3c_synth as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    return rbt_lib.module()


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(device=0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def container():
    gofs = [V.gof_streams(64, 64, 2, 11), V.gof_streams(128, 64, 1, 12), V.gof_streams(96, 96, 2, 13), V.gof_streams(192, 128, 3, 14), V.gof_streams(64, 128, 1, 15)]
    units = []
    for g, s in enumerate(gofs):
        units += V.gof_units(s, 100 + g, aux=(g == 1), extra_attr_partition=(g == 3))
    return V.sample_stream(units, 3)


@pytest.mark.parametrize("rate", [1, 3, 5])
def test_transcode_v3c_equals_oracle(R, ctx, container, rate):
    gs = rbt_lib.module_file("gof_shard")
    gq, aq, prec = gs.RATE_POINTS[rate]
    got = ctx.transcode_v3c(container, gq, aq, occupancy_precision=prec)
    assert got == O.v3c_transcode(container, gq, aq, prec)
    idx = R.v3c_index(got, ctx.L)
    assert [u["gof"] for u in idx][-1] == 4 and len(idx) == len(R.v3c_index(container, ctx.L))


@pytest.mark.parametrize("depth,per", [(1, 1), (4, 1), (16, 2), (16, 3)])
def test_transcode_v3c_depths(R, ctx, container, depth, per):
    ctx.set_depth(depth)
    try:
        assert ctx.transcode_v3c(container, 24, 32, gofs_per_job=per, forced_precision_bytes=4) == O.v3c_transcode(container, 24, 32, 4, 4)
    finally:
        ctx.set_depth(4)


@pytest.mark.parametrize("occupancy_rd,preset", [(1, 0), (0, 1), (1, 1)])
def test_transcode_v3c_options(R, ctx, container, occupancy_rd, preset):
    """rbt_v3c_params.occupancy_rd / .preset through the walk with several jobs in flight: every GOF's geometry / attribute units are coded with THAT GOF's occupancy map"""
    ctx.set_depth(4)
    got = ctx.transcode_v3c(container, 24, 32, gofs_per_job=2, occupancy_rd=occupancy_rd, preset=preset)
    assert got == O.v3c_transcode(container, 24, 32, occupancy_rd=occupancy_rd, preset=preset) and got != ctx.transcode_v3c(container, 24, 32)


def test_sharded_contexts_merge_to_the_unsharded_file(R, ctx, container):
    gs = rbt_lib.module_file("gof_shard")
    parts = []
    for r in range(3):
        c = R.Context(device=0, rank=r, world=3)
        parts.append(c.transcode_v3c(container, 24, 32)); c.close()
    assert gs.merge_v3c(R, parts, lib=ctx.L) == ctx.transcode_v3c(container, 24, 32)


def test_full_size_gof_in_a_container(R, ctx):
    """the HM-like 1280x1280 fixture (first 2 point-cloud frames of each video: the oracle finishes them in seconds) as a one-GOF container"""
    gs = rbt_lib.module_file("gof_shard")
    src = [open(os.path.join(ROOT, "tests/golden", f"hm_r5_1280x1280_f32_{k}.annexb"), "rb").read() for k in ("occ", "geo", "attr")]
    s = [b"".join(gs.split_pairs(x)[:2]) for x in src]
    data = V.sample_stream(V.gof_units(s, 7), 4)
    got = ctx.transcode_v3c(data, 24, 32)
    assert got == O.v3c_transcode(data, 24, 32, 4)
    assert len(got) < len(data) // 2


@pytest.mark.parametrize("depth,per", [(1, 1), (4, 1), (16, 0)])
def test_stream_walk_on_the_gpu(R, ctx, container, depth, per):
    """rbt_transcode_v3c_stream: GOFs handed to the sink in order while later jobs run; written as one sample stream they are the file rbt_transcode_v3c makes"""
    seen, got = [], []
    ctx.set_depth(depth)
    try:
        ctx.transcode_v3c_stream(container, lambda g, us: (seen.append(g), got.extend(us)) and None, 24, 32, gofs_per_job=per)
    finally:
        ctx.set_depth(4)
    assert seen == [0, 1, 2, 3, 4]
    assert R.v3c_write(got, 0, ctx.L) == O.v3c_transcode(container, 24, 32, 4)
