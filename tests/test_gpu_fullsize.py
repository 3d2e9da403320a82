"""Full-size parity on the GPU, byte for byte against the oracle (round-2 review: until now this comparison lived in bench.py's CPU-baseline leg only, where a failure
prints `false` instead of failing a test): 16 point-cloud frames of the committed 1280x1280 HM-like fixture
  - R5 -> R1 (BASELINE.json configs[2] at full size),
  - R5 -> R3 through rbt_submit_gof with two jobs in flight,
  - R5 -> R3 with occupancy-aware coding (rbt_stream_params.occupancy_rd).
The oracle runs per point-cloud frame in worker processes (closed GOPs: the per-frame streams concatenate to the whole)."""
import os
import subprocess
import sys
import tempfile
import numpy as np
import pytest
import rbt_lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N = 16


@pytest.fixture(scope="module")
def ctx():
    R = rbt_lib.module()
    c = R.Context(device=0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def frames():
    gs = rbt_lib.module_file("gof_shard")
    return {k: gs.split_pairs(open(os.path.join(HERE, "golden", f"hm_r5_1280x1280_f32_{k}.annexb"), "rb").read())[:N] for k in ("occ", "geo", "attr")}


def oracle_streams(frames, gq, aq, rows=-1, occ_rd=0):
    """[occupancy, geometry, attribute] streams of the N frames as the oracle transcodes them, one worker process per core"""
    ncore = max(1, min(16, os.cpu_count() or 1, N))
    per = {"n": np.array(N)}
    for q in range(N):
        per[f"o{q}"], per[f"g{q}"], per[f"a{q}"] = (np.frombuffer(frames[k][q], np.uint8) for k in ("occ", "geo", "attr"))
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "frames.npz"); np.savez(f, **per)
        worker = os.path.join(HERE, "oracle_frames_worker.py")
        procs = [subprocess.Popen([sys.executable, worker, f, os.path.join(td, f"out{i}.npz"), str(i), str(ncore), str(gq), str(aq), str(rows), str(occ_rd)]) for i in range(ncore)]
        assert all(p.wait(timeout=900) == 0 for p in procs)
        outs = {}
        for i in range(ncore):
            z = np.load(os.path.join(td, f"out{i}.npz"))
            outs.update({k: z[k].tobytes() for k in z.files})
    return [b"".join(outs[f"{c}{q}"] for q in range(N)) for c in "oga"]


def test_r1_sixteen_frames(ctx, frames):
    """configs[2] at full size: R5 -> R1 (QP 32 / 42, occupancy precision 4)"""
    R = rbt_lib.module()
    gs = rbt_lib.module_file("gof_shard")
    got = ctx.transcode_gof([b"".join(frames[k]) for k in ("occ", "geo", "attr")], gs.rate_params(R, 1))
    assert got == oracle_streams(frames, 32, 42)


def test_r3_two_jobs_in_flight(ctx, frames):
    """R5 -> R3, the 16 frames as two jobs of eight submitted before either is collected"""
    R = rbt_lib.module()
    gs = rbt_lib.module_file("gof_shard")
    ctx.set_depth(2)
    halves = [[b"".join(frames[k][h * 8:(h + 1) * 8]) for k in ("occ", "geo", "attr")] for h in (0, 1)]
    jobs = [ctx.submit_gof(s, gs.rate_params(R, 3)) for s in halves]
    outs = [ctx.wait_gof(j) for j in jobs]
    assert [outs[0][c] + outs[1][c] for c in range(3)] == oracle_streams(frames, 24, 32)


def test_r3_occupancy_aware(ctx, frames):
    """R5 -> R3 with occupancy_rd: every frame coded with the occupancy map its own occupancy picture comes out with"""
    R = rbt_lib.module()
    gs = rbt_lib.module_file("gof_shard")
    got = ctx.transcode_gof([b"".join(frames[k]) for k in ("occ", "geo", "attr")], gs.rate_params(R, 3, occupancy_rd=1))
    want = oracle_streams(frames, 24, 32, occ_rd=1)
    assert got == want
    plain = ctx.transcode_gof([b"".join(frames[k]) for k in ("occ", "geo", "attr")], gs.rate_params(R, 3))
    assert len(got[1]) < 0.6 * len(plain[1]) and len(got[2]) < 0.85 * len(plain[2]) and got[0] == plain[0]
