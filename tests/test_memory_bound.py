"""Device memory as a bound of the V3C walk (round-3 advisor finding: rbt_job_shape looks at the length of a walk only, so larger atlases or a fuller device ended in
RBT_ERR_NOMEM) and what is left of a context after a failure. Runs on the host build, whose stand-in device has a settable size (RBT_HOSTEMU_HBM_MB, tests/hostemu);
the same code paths on the GPU: tests/test_gpu_memory.py."""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import v3c_synth as V

MB = 1 << 20


@pytest.fixture(scope="module")
def R():
    return rbt_lib.module()


@pytest.fixture()
def ctx(R):
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    yield c
    os.environ.pop("RBT_HOSTEMU_HBM_MB", None); os.environ.pop("RBT_HBM_RESERVE_MB", None)
    c.close()


@pytest.fixture(scope="module")
def walk():
    """6 GOFs of 2 frames of 128x128 maps (4 MB of arenas each on the host build) and the same walk with a fourth GOF of 256x256 maps (15 MB)"""
    small = [V.gof_streams(128, 128, 2, 40 + g) for g in range(6)]
    mixed = small[:3] + [V.gof_streams(256, 256, 2, 77)] + small[3:]
    mk = lambda gofs: V.sample_stream([u for g, s in enumerate(gofs) for u in V.gof_units(s, 30 + g)], 3)
    return mk(small), mk(mixed)


@pytest.fixture(scope="module")
def sizes(R):
    """(small, big): device megabytes one GOF of the walks takes (128x128 and 256x256 maps, 2 frames) - measured, the tests below place the stand-in device's size around them"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    gs = rbt_lib.module_file("gof_shard")
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    out = []
    for w in (128, 256):
        j = c.submit_gof(V.gof_streams(w, w, 2, 5), gs.rate_params(R, 3)); out.append(c.job_memory(j) / MB); c.wait_gof(j)
    c.close()
    assert 1.5 < out[0] < 8 and out[1] > 3 * out[0], out
    return out


def test_memory_figures(ctx, R):
    gs = rbt_lib.module_file("gof_shard")
    os.environ["RBT_HOSTEMU_HBM_MB"] = "64"; os.environ["RBT_HBM_RESERVE_MB"] = "8"
    m0 = ctx.device_memory()
    assert (m0["total"], m0["free"], m0["in_use"], m0["reserve"]) == (64 * MB, 64 * MB, 0, 8 * MB)
    j = ctx.submit_gof(V.gof_streams(128, 128, 2, 5), gs.rate_params(R, 3))
    b = ctx.job_memory(j)
    m1 = ctx.device_memory()
    assert 1 * MB < b < 8 * MB and m1["in_use"] == b and m1["free"] == 64 * MB - b
    ctx.wait_gof(j)
    assert ctx.device_memory()["in_use"] == 0


@pytest.mark.parametrize("jobs_that_fit,depth,per", [(1.3, 4, 1), (2.5, 4, 0), (1.2, 16, 2), (300, 4, 0)])
def test_walk_is_bounded_by_device_memory_not_only_by_its_length(ctx, walk, sizes, jobs_that_fit, depth, per):
    """a device that holds one or two jobs of the walk's shape where the announced depth asks for 4 or 16: same bytes as the oracle, no error"""
    os.environ["RBT_HOSTEMU_HBM_MB"] = str(int(jobs_that_fit * sizes[0] * max(per, 1)) + 1)
    ctx.set_depth(depth)
    assert ctx.transcode_v3c(walk[0], 24, 32, gofs_per_job=per) == O.v3c_transcode(walk[0], 24, 32, 4)
    assert ctx.device_memory()["in_use"] == 0


def test_job_that_does_not_fit_next_to_others_runs_again_alone(ctx, walk, sizes):
    """the fourth GOF has larger maps than the first job's footprint promised: its job fails with RBT_ERR_NOMEM while others are in flight; the walk collects them, runs
    the GOFs of the failed job one at a time and ends with the oracle's bytes, GOFs in order"""
    os.environ["RBT_HOSTEMU_HBM_MB"] = str(int(sizes[1] + 0.5 * sizes[0]) + 1)      # the large GOF fits alone, not in one job with a small one (gofs_per_job 2), let alone next to another job
    ctx.set_depth(4)
    seen = []
    want = O.v3c_transcode(walk[1], 24, 32, 4)
    assert ctx.transcode_v3c(walk[1], 24, 32, gofs_per_job=2) == want
    ctx.transcode_v3c_stream(walk[1], lambda g, us: seen.append(g) and None, 24, 32, gofs_per_job=1)
    assert seen == list(range(7))


def test_a_gof_that_fits_nowhere_is_an_error_and_the_context_lives_on(ctx, R, walk, sizes):
    os.environ["RBT_HOSTEMU_HBM_MB"] = str(int(sizes[1]) - 1)                 # less than the 256x256 GOF needs
    ctx.set_depth(4)
    with pytest.raises(R.RbtError) as e:
        ctx.transcode_v3c(walk[1], 24, 32, gofs_per_job=1)
    assert e.value.code == -5 and "device allocation failed" in str(e.value)           # RBT_ERR_NOMEM, with the text of the call that failed
    assert ctx.device_memory()["in_use"] == 0
    ctx.set_depth(2); ctx.trim()
    assert ctx.transcode_v3c(walk[0], 24, 32) == O.v3c_transcode(walk[0], 24, 32, 4)


def test_error_text_and_context_state_survive_a_failed_walk(ctx, R, walk):
    """round-3 advisor finding: the drain of the remaining jobs and the depth restore cleared rbt_last_error; a failed walk left jobs in their slots. A container whose
    second GOF is damaged, walked with several jobs in flight: the error names the cause, and set_depth / trim / another walk work afterwards"""
    prec, units = V.parse(walk[0])
    bad = bytearray(units[8]); assert bad[0] >> 3 == V.GVD; bad[60:len(bad) - 20] = bytes(len(bad) - 80); units[8] = bytes(bad)     # geometry video of GOF 1: slice data zeroed
    data = V.sample_stream(units, prec)
    for depth, per in ((4, 1), (4, 0), (16, 2)):
        ctx.set_depth(depth)
        with pytest.raises(R.RbtError) as e:
            ctx.transcode_v3c(data, 24, 32, gofs_per_job=per)
        assert e.value.code in (-2, -3) and len(str(e.value).split(": ", 2)) == 3, str(e.value)      # code: generic text: the library's detail
        assert ctx.get_depth() == depth
        ctx.set_depth(3); ctx.trim()
    assert ctx.transcode_v3c(walk[0], 24, 32) == O.v3c_transcode(walk[0], 24, 32, 4)


def test_python_sequence_walk_is_bounded_by_memory_too(ctx, R, sizes):
    """gof_shard.transcode_sequence (bench.py's sequence_walk, the multi-rank Python host): its job loop waits for results before it submits what the device cannot hold"""
    gs = rbt_lib.module_file("gof_shard")
    seq = [V.gof_streams(128, 128, 2, 60 + g) for g in range(5)]
    P = gs.rate_params(R, 3)
    want = [[O.transcode_substream(s[0], 0, 8, rows_per_slice=-1, md5_sei=0), O.transcode_substream(s[1], 1, 24, rows_per_slice=-1, md5_sei=0), O.transcode_substream(s[2], 19, 32, rows_per_slice=-1, md5_sei=0)] for s in seq]
    os.environ["RBT_HOSTEMU_HBM_MB"] = str(int(1.4 * sizes[0]) + 1)                    # one job at a time fits
    assert gs.transcode_sequence(ctx, seq, P, depth=4, gofs_per_job=1) == want
    assert ctx.device_memory()["in_use"] == 0
