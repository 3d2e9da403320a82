"""Multi-rank path (BASELINE.json configs[3] and configs[4]) on CPU: world_size 2 over gloo, every rank running the REAL
flow - shard -> rbt_submit_gof / rbt_wait_gof -> gather -> stitch - on the test-only host build of the kernel bodies (tests/hostemu; no GPU
in this container). The stitched output must equal the unsharded run byte for byte, and the oracle's transcodeData per GOF. bench.py
--gpus N runs the same functions with backend nccl (RCCL)."""
import os
import subprocess
import torch.multiprocessing as mp
import pytest
import rbt_lib

W, H, GOF, FRAMES = 64, 64, 4, 14          # 14 frames in GOFs of 4 -> 4 + 4 + 4 + 2: a shorter tail GOF like 300 = 9 x 32 + 12


def _sequence():
    import oracle_lib as O
    import synth
    gs = rbt_lib.module_file("gof_shard")
    geo, attr, occ = synth.make_gof(W, H, GOF, 5)
    sg, _ = O.encode(geo, W, H, 10, 16, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    sa, _ = O.encode(attr, W, H, 10, 22, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    so, _ = O.encode(occ, W // 2, H // 2, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    return gs.make_sequence([so, sg, sa], FRAMES, GOF)


def _container(seq):
    """the same sequence as a V3C sample stream (tests/v3c_synth.py)"""
    import v3c_synth as V
    units = []
    for g, s in enumerate(seq):
        units += V.gof_units(s, 300 + g, aux=(g == 2))
    return V.sample_stream(units, 3)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    ctx = R.Context(lib_path=rbt_lib.HOSTEMU_LIB, rank=rank, world=world)
    seq = _sequence()
    out = gs.transcode_sequence(ctx, seq, gs.rate_params(R, 3), rank=rank, world=world, depth=2)
    fan = gs.transcode_fanout(ctx, R, seq[2:], rates=(1, 2, 3, 4, 5), rank=rank, world=world, depth=2)
    v3c = gs.transcode_v3c(ctx, R, _container(seq), 24, 32, rank=rank, world=world, depth=2)
    if rank == 0:
        q.put((out, fan, v3c))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.fixture(scope="module")
def hostemu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])


def test_sharding_rules():
    gs = rbt_lib.module_file("gof_shard")
    assert gs.gof_lengths(300) == [32] * 9 + [12] and gs.gof_lengths(32) == [32] and gs.gof_lengths(14, 4) == [4, 4, 4, 2]
    assert gs.gofs_of_rank(10, 0, 8) == [0, 8] and gs.gofs_of_rank(10, 1, 8) == [1, 9] and gs.gofs_of_rank(10, 7, 8) == [7]
    assert sorted(sum((gs.gofs_of_rank(10, r, 3) for r in range(3)), [])) == list(range(10))
    assert [gs.rates_of_rank((1, 2, 3, 4, 5), r, 8) for r in range(8)] == [[1], [2], [3], [4], [5], [], [], []]
    assert gs.rates_of_rank((1, 2, 3, 4, 5), 0, 2) == [1, 3, 5] and gs.rates_of_rank((1, 2, 3, 4, 5), 1, 2) == [2, 4]
    # cfg/rate/ctc-r{1..5}.cfg:5-11
    assert gs.RATE_POINTS == {1: (32, 42, 4), 2: (28, 37, 4), 3: (24, 32, 4), 4: (20, 27, 4), 5: (16, 22, 2)}


def test_job_shapes(hostemu):
    gs = rbt_lib.module_file("gof_shard")
    assert gs.spread(20, 3) == [3, 3, 3, 3, 3, 3, 2] and gs.spread(20, 2) == [2] * 10 and gs.spread(5, 3) == [3, 2] and gs.spread(0, 2) == [] and gs.spread(1, 4) == [1]
    assert all(sum(gs.spread(n, g)) == n and max(gs.spread(n, g)) <= g for n in range(1, 70) for g in range(1, 9))
    assert gs.job_shape(256) == (3, 16) and gs.job_shape(96) == (3, 16) and gs.job_shape(95) == (2, 16) and gs.job_shape(48) == (2, 16) and gs.job_shape(20) == (3, 7) and gs.job_shape(10) == (5, 2) and gs.job_shape(1) == (1, 1) and gs.job_shape(40) == (6, 7)
    R = rbt_lib.module(); L = R.load(rbt_lib.HOSTEMU_LIB)      # the library's copy of the rule (rbt_job_shape)
    assert all(R.job_shape(n, d, L) == gs.job_shape(n, d) for n in range(1, 130) for d in (1, 2, 4, 7, 16))
    assert gs.job_shape(20, 4) == (3, 4) and gs.job_shape(2) == (1, 2) and gs.job_shape(13) == (2, 7)


def test_context_owns_gofs_like_the_python_rule(hostemu):
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    for world in (1, 2, 3, 8):
        for rank in range(world):
            c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB, rank=rank, world=world)
            assert [g for g in range(10) if c.owns_gof(g)] == gs.gofs_of_rank(10, rank, world)
            c.close()
    for rank, world in ((-1, 2), (2, 2), (0, 0)):
        with pytest.raises(R.RbtError) as e:
            R.Context(lib_path=rbt_lib.HOSTEMU_LIB, rank=rank, world=world)
        assert e.value.code == -4      # RBT_ERR_PARAM


def test_sequence_and_fanout_world2_equal_unsharded_and_oracle(hostemu):
    import oracle_lib as O
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    seq = _sequence()
    assert [len(gs.split_pairs(g[1])) for g in seq] == [4, 4, 4, 2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29517
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, fan, v3c = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # unsharded: one context walks the whole sequence
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    assert out == gs.transcode_sequence(c, seq, gs.rate_params(R, 3), depth=3)
    assert out == gs.transcode_sequence(c, seq, gs.rate_params(R, 3), depth=2, gofs_per_job=3)      # 4 GOFs as jobs of 2 + 2
    assert out == gs.transcode_sequence(c, seq, gs.rate_params(R, 3), gofs_per_job=0)               # shape by job_shape
    assert fan == gs.transcode_fanout(c, R, seq[2:], depth=1)
    c.close()
    # the oracle's transcodeData, GOF by GOF (PCCTranscoder.cpp:145-168)
    assert out == [O.transcode_data(g, [(0, 8, 4, 5, gs.DEFAULT_ROWS, 0), (1, 24, 4, 5, gs.DEFAULT_ROWS, 0), (19, 32, 4, 5, gs.DEFAULT_ROWS, 0)]) for g in seq]
    for r, (gq, aq, pr) in gs.RATE_POINTS.items():
        assert fan[r] == [O.transcode_data(g, [(0, 8, pr, 5, gs.DEFAULT_ROWS, 0), (1, gq, pr, 5, gs.DEFAULT_ROWS, 0), (19, aq, pr, 5, gs.DEFAULT_ROWS, 0)]) for g in seq[2:]]
    # the container walk, sharded: the merged file is the one a single rank writes and the one the oracle's restatement of PccAppTranscoder's loop writes
    assert v3c == O.v3c_transcode(_container(seq), 24, 32, 4)
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB); assert v3c == c.transcode_v3c(_container(seq), 24, 32); c.close()
    assert fan[5][0][0] == seq[2][0]       # R5 keeps occupancy precision 2: the reference does not touch the occupancy stream (:150)


def test_bench_launcher_stops_all_ranks_when_one_fails():
    """python bench.py --gpus N (self-launched ranks): when a rank ends non-zero the parent stops the others and exits non-zero within seconds instead of waiting for ranks
    that sit in a barrier (round-3 review). The hook makes rank 1 fail and the other ranks never end; no GPU is touched."""
    import subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--backend", "gloo"], env=dict(os.environ, RBT_BENCH_SELFTEST="fail:1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "rank 1 ended with status 3" in r.stderr and time.time() - t0 < 60
