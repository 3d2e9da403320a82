"""the V3C sample stream either side of the hot path (rbt_v3c_index / rbt_v3c_write / rbt_transcode_v3c, include/rbt.h) against an
independent Python construction of the container (tests/v3c_synth.py) and against the oracle's restatement of the walk (oracle_v3c_transcode).
The index / write halves need no GPU; the transcode runs on the host build of the kernels here and on the GPU in test_gpu_v3c.py."""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import v3c_synth as V


@pytest.fixture(scope="module")
def R():
    return rbt_lib.module()


@pytest.fixture(scope="module")
def hlib(R):
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    return R.load(rbt_lib.HOSTEMU_LIB)


@pytest.fixture(scope="module")
def ctx(R, hlib):
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    yield c
    c.close()


@pytest.fixture(scope="module")
def container():
    """3 GOFs (2 / 1 / 2 point-cloud frames, different sizes), the second with auxiliary video units and a second attribute partition"""
    gofs = [V.gof_streams(64, 64, 2, 11), V.gof_streams(128, 64, 1, 12), V.gof_streams(96, 96, 2, 13)]
    units = []
    for g, s in enumerate(gofs):
        units += V.gof_units(s, 100 + g, aux=(g == 1), extra_attr_partition=(g == 1))
    return gofs, units


@pytest.mark.parametrize("precision", [2, 3, 4, 8])
def test_index_lists_every_unit(R, hlib, container, precision):
    gofs, units = container
    data = V.sample_stream(units, precision)
    idx = R.v3c_index(data, hlib)
    assert len(idx) == len(units)
    assert [u["type"] for u in idx] == [x[0] >> 3 for x in units]
    assert [data[u["offset"]:u["offset"] + u["size"]] for u in idx] == units
    assert [u["gof"] for u in idx] == [0] * 5 + [1] * 8 + [2] * 5
    # the three videos transcodeData names, per GOF; auxiliary video and the second attribute partition are not among them
    assert [u["video_type"] for u in idx if u["gof"] == 1] == [-1, -1, 0, 1, -1, 19, -1, -1]
    aux = [u for u in idx if u["auxiliary_video"]]
    assert [u["type"] for u in aux] == [V.GVD, V.AVD] and idx[12]["attribute_dimension_index"] == 1


def test_index_rejects_truncated_streams(R, hlib, container):
    data = V.sample_stream(container[1], 3)
    for cut in (len(data) - 1, len(data) - 100, 3, 2):
        with pytest.raises(R.RbtError) as e:
            R.v3c_index(data[:cut], hlib)
        assert e.value.code == -2
    assert R.v3c_index(data[:1], hlib) == []                                     # a header and no units


@pytest.mark.parametrize("largest,forced,want", [(5, 0, 1), (255, 0, 1), (257, 0, 2), (65535, 0, 2), (65537, 0, 3), (70000, 0, 3), (300, 4, 4), (70000, 2, 3)])
def test_write_precision_follows_the_largest_unit(R, hlib, largest, forced, want):
    """PCCBitstreamWriter::write (:57-91): min(max(ceil(ceilLog2(largest unit) / 8), 1), 8) bytes, at least the forced value"""
    units = [V.unit_header(V.VPS) + bytes(20), V.unit_header(V.AD) + bytes(largest - 4), V.unit_header(V.OVD) + b"\x01" * 9]
    data = R.v3c_write(units, forced, hlib)
    assert data == V.sample_stream(units, want)
    assert V.parse(data) == (want, units)


@pytest.mark.parametrize("largest,want", [(256, 2), (65536, 3), (1 << 24, 4)])
def test_write_at_powers_of_256_takes_one_byte_more_than_the_reference(R, hlib, largest, want):
    """ceilLog2(256) = 8 -> one byte by the reference's rule (PCCBitstreamWriter.cpp:71-72), which cannot hold 256: the size field would wrap to 0 and nothing could read the
    file back. The library deviates there (round-2 advisor finding): the precision that holds the largest unit."""
    units = [V.unit_header(V.AD) + bytes(largest - 4)]
    data = R.v3c_write(units, 0, hlib)
    assert V.parse(data) == (want, units)
    assert V.parse(R.v3c_write(units, want + 1, hlib)) == (want + 1, units)


def test_transcode_v3c_equals_oracle_and_manual_walk(R, ctx, container):
    gofs, units = container
    data = V.sample_stream(units, 3)
    got = ctx.transcode_v3c(data, 24, 32, occupancy_precision=4)
    assert got == O.v3c_transcode(data, 24, 32, 4)
    # the same through the sub-bitstream entry points, assembled by the Python container code
    P = R.StreamParams
    want_units = []
    for g, s in enumerate(gofs):
        o = [ctx.transcode_substream(s[0], 0, 8, 4, 5, -1, 0), ctx.transcode_substream(s[1], 1, 24, 4, 5, -1, 0), ctx.transcode_substream(s[2], 19, 32, 4, 5, -1, 0)]
        src = [u for u, i in zip(units, R.v3c_index(data, ctx.L)) if i["gof"] == g]
        k = 0
        for u, i in zip(src, [i for i in R.v3c_index(data, ctx.L) if i["gof"] == g]):
            if i["video_type"] >= 0:
                want_units.append(u[:4] + O.byte_to_sample_stream(o[k])); k += 1
            else:
                want_units.append(u)
    prec, out_units = V.parse(got)
    assert out_units == want_units and prec == 2
    # parameter sets and atlas data are carried over; the output is smaller and indexes like the input
    oi = R.v3c_index(got, ctx.L)
    assert [(u["type"], u["gof"], u["video_type"]) for u in oi] == [(u["type"], u["gof"], u["video_type"]) for u in R.v3c_index(data, ctx.L)]
    assert len(got) < len(data)


def test_transcode_v3c_leaves_occupancy_alone_unless_precision_4(R, ctx, container):
    gofs, units = container
    data = V.sample_stream(units[:5], 2)
    got = ctx.transcode_v3c(data, 28, 37, occupancy_precision=2, forced_precision_bytes=4)
    assert got == O.v3c_transcode(data, 28, 37, 2, 4)
    prec, out_units = V.parse(got)
    assert prec == 4 and out_units[2] == units[2] and out_units[:2] == units[:2] and out_units[3] != units[3]


def test_transcode_v3c_options_reach_the_encoder(R, ctx, container):
    """rbt_v3c_params.preset and .occupancy_rd: the walk hands them to every GOF's geometry / attribute units, as the oracle's walk does"""
    gofs, units = container
    data = V.sample_stream(units[:5], 4)
    plain = ctx.transcode_v3c(data, 24, 32)
    fast = ctx.transcode_v3c(data, 24, 32, preset=R.RBT_PRESET_FAST)
    assert fast == O.v3c_transcode(data, 24, 32, preset=1) and fast != plain
    both = ctx.transcode_v3c(data, 24, 32, preset=R.RBT_PRESET_FAST, occupancy_rd=1)
    assert both == O.v3c_transcode(data, 24, 32, preset=1, occupancy_rd=1) and len(both) < len(fast)


@pytest.mark.parametrize("depth,per", [(1, 1), (2, 1), (3, 2), (16, 2), (16, 0), (1, 0)])   # per 0: job shape by rbt_job_shape (3 GOFs: 2 jobs of 2 + 1)
def test_transcode_v3c_same_bytes_at_every_depth(R, ctx, container, depth, per):
    gofs, units = container
    data = V.sample_stream(units, 3)
    ctx.set_depth(depth)
    try:
        assert ctx.transcode_v3c(data, 24, 32, gofs_per_job=per) == O.v3c_transcode(data, 24, 32, 4)
        assert ctx.get_depth() == depth                                          # a short walk lowers the depth only for the call
    finally:
        ctx.set_depth(4)


def test_transcode_v3c_refuses_separate_map_streams(R, ctx, container):
    gofs, units = container
    u = list(units[:5])
    u.insert(4, V.unit_header(V.GVD, map_idx=1) + units[3][4:])                  # geometry D0 and D1 as separate streams
    with pytest.raises(R.RbtError) as e:
        ctx.transcode_v3c(V.sample_stream(u, 3), 24, 32)
    assert e.value.code == -3


def test_transcode_v3c_reports_damaged_video_units(R, ctx, container):
    gofs, units = container
    u = list(units[:5])
    u[3] = u[3][:4] + u[3][4:200]                                                # sample stream cut in the middle of a NAL unit
    with pytest.raises(R.RbtError) as e:
        ctx.transcode_v3c(V.sample_stream(u, 3), 24, 32)
    assert e.value.code == -2


def test_sharded_outputs_merge_into_the_unsharded_stream(R, hlib, container):
    """multi-GPU form: every rank's output holds the GOFs it owns; merged in GOF order by index + write == the single-rank output"""
    gofs, units = container
    data = V.sample_stream(units, 3)
    gs = rbt_lib.module_file("gof_shard")
    whole = O.v3c_transcode(data, 24, 32, 4)
    parts = []
    for r in range(2):
        c = R.Context(rank=r, world=2, lib_path=rbt_lib.HOSTEMU_LIB)
        parts.append(c.transcode_v3c(data, 24, 32)); c.close()
    assert [sorted({u["gof"] for u in R.v3c_index(p, hlib)}) for p in parts] == [[0, 1], [0]]   # rank 0: GOFs 0 and 2 (renumbered), rank 1: GOF 1
    assert gs.merge_v3c(R, parts, lib=hlib) == whole
    # more ranks than GOFs: the idle rank hands back a header and no units
    parts = []
    for r in range(4):
        c = R.Context(rank=r, world=4, lib_path=rbt_lib.HOSTEMU_LIB)
        parts.append(c.transcode_v3c(data, 24, 32, forced_precision_bytes=4)); c.close()
    assert len(parts[3]) == 1 and R.v3c_index(parts[3], hlib) == []
    assert gs.merge_v3c(R, parts, 4, lib=hlib) == O.v3c_transcode(data, 24, 32, 4, 4)


def test_wrap_and_unwrap_helpers(R, ctx, container):
    """gof_shard.wrap_v3c / unwrap_v3c (bench.py's container leg): the video units of a wrapped sequence hold the sub-bitstreams in sample stream form, and a
    transcode of the container carries exactly what the per-GOF walk produces"""
    gs = rbt_lib.module_file("gof_shard")
    gofs, _ = container
    data = gs.wrap_v3c(R, gofs, lib=ctx.L)
    assert gs.unwrap_v3c(R, data, lib=ctx.L, annexb=False) == [[O.byte_to_sample_stream(s) for s in g] for g in gofs]
    assert [[O.decode(s)[0].tobytes() for s in g] for g in gs.unwrap_v3c(R, data, lib=ctx.L)] == [[O.decode(s)[0].tobytes() for s in g] for g in gofs]
    walk = gs.transcode_sequence(ctx, gofs, gs.rate_params(R, 3), depth=2)
    out = gs.transcode_v3c(ctx, R, data, 24, 32, depth=2)
    assert gs.unwrap_v3c(R, out, lib=ctx.L, annexb=False) == [[O.byte_to_sample_stream(s) for s in g] for g in walk]


def test_stats_add_up_like_the_reference_report(R, hlib, container):
    """rbt_v3c_stats == PCCBitstreamStat::trace's figures (PCCBitstream.h:48-154) computed here from the unit list"""
    gofs, units = container
    data = V.sample_stream(units, 3)
    st = R.v3c_stats(data, hlib)
    by_type = [sum(len(u) for u in units if u[0] >> 3 == t) for t in range(5)]
    aux = lambda u: (u[3] & 1) if u[0] >> 3 == V.AVD else (u[2] >> 4) & 1
    geo = sum(len(u) - 4 for u in units if u[0] >> 3 == V.GVD); att = sum(len(u) - 4 for u in units if u[0] >> 3 == V.AVD)
    assert st["unit_size"] == by_type and (st["n_units"], st["n_gofs"], st["unit_size_precision_bytes"]) == (len(units), 3, 3)
    assert st["header"] == 1 + 3 * len(units) and st["total"] == len(data)
    assert st["geometry_video"] + st["geometry_aux_video"] == geo == st["total_geometry"] and st["attribute_video"] + st["attribute_aux_video"] == att == st["total_attribute"]
    assert st["geometry_aux_video"] == sum(len(u) - 4 for u in units if u[0] >> 3 == V.GVD and aux(u)) == 300
    assert st["occupancy_video"] == sum(len(u) - 4 for u in units if u[0] >> 3 == V.OVD)
    assert st["total_metadata"] == len(data) - geo - att


def test_transcode_v3c_empty_and_video_less_streams(R, ctx, container):
    """a sample stream with no units, and GOFs without video units (parameter set and atlas only), go through untouched apart from the size precision"""
    assert ctx.transcode_v3c(bytes([0x40]), 24, 32) == bytes([0x00]) == O.v3c_transcode(bytes([0x40]), 24, 32, 4)
    _, units = container
    meta = [u for u in units if u[0] >> 3 in (V.VPS, V.AD)]
    data = V.sample_stream(meta, 4)
    got = ctx.transcode_v3c(data, 24, 32, gofs_per_job=0)
    assert got == O.v3c_transcode(data, 24, 32, 4) and V.parse(got) == (2, meta)
    # a video unit with an empty payload (header only) is carried over as it is
    odd = units[:2] + [V.unit_header(V.OVD)] + units[3:5]
    data = V.sample_stream(odd, 3)
    got = ctx.transcode_v3c(data, 24, 32)
    assert got == O.v3c_transcode(data, 24, 32, 4) and V.parse(got)[1][2] == V.unit_header(V.OVD)


@pytest.mark.parametrize("depth,per", [(1, 1), (2, 1), (16, 0), (2, 2)])
def test_stream_walk_hands_gofs_over_in_order(R, ctx, container, depth, per):
    """rbt_transcode_v3c_stream: one sink call per GOF, in order, and the units written as one sample stream are the file rbt_transcode_v3c makes"""
    gofs, units = container
    extra = [V.unit_header(V.VPS) + b"tail-gof-without-video", V.unit_header(V.AD) + bytes(40)]       # a GOF with no video units at the end
    data = V.sample_stream(units + extra, 3)
    seen, got = [], []
    ctx.set_depth(depth)
    try:
        ctx.transcode_v3c_stream(data, lambda g, us: (seen.append(g), got.extend(us)) and None, 24, 32, gofs_per_job=per)
        assert seen == [0, 1, 2, 3]
        assert R.v3c_write(got, 0, ctx.L) == ctx.transcode_v3c(data, 24, 32) == O.v3c_transcode(data, 24, 32, 4)
        # a sink that gives up after the first GOF ends the walk with an error; the context stays usable
        calls = []
        with pytest.raises(R.RbtError) as e:
            ctx.transcode_v3c_stream(data, lambda g, us: calls.append(g) or True, 24, 32, gofs_per_job=per)
        assert e.value.code == -4 and calls == [0]
        assert ctx.transcode_v3c(data, 24, 32) == O.v3c_transcode(data, 24, 32, 4) and ctx.get_depth() == depth
    finally:
        ctx.set_depth(4)
