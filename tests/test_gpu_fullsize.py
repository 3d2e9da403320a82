"""Full-size parity on the GPU, byte for byte against the oracle (round-2 review: until now this comparison lived in bench.py's CPU-baseline leg only, where a failure
prints `false` instead of failing a test): 16 point-cloud frames of the committed 1280x1280 HM-like fixture
  - R5 -> R1 (BASELINE.json configs[2] at full size),
  - R5 -> R3 through rbt_submit_gof with two jobs in flight,
  - R5 -> R3 with occupancy-aware coding (rbt_stream_params.occupancy_rd).
The oracle runs per point-cloud frame in worker processes (closed GOPs: the per-frame streams concatenate to the whole).
Round 4: the same 16 frames in the CTC encoder's stream structure (one IDR per sub-bitstream; frames handed to the oracle with the parameter sets in front), and
configs[2] as it is worded - TWO sequences (the stand-ins of loot and redandblack: synthetic atlases of seeds 1000 and 1450, cfg/sequence start frames), coded with the
HM-like encoder on the box's cores, R5 -> R1."""
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
    N = len(frames["occ"])
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


def test_r3_sixteen_frames_in_the_ctc_stream_structure(ctx):
    """the CTC-structured fixture (one IDR, TRAIL_N / TRAIL_R pictures, POC running on): 16 frames R5 -> R3 in one call == the oracle's transcodes of the frames"""
    import json
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    man = json.load(open(os.path.join(HERE, "golden", "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
    streams = {k: open(os.path.join(HERE, "golden", man["streams"][k]["file"]), "rb").read() for k in ("occ", "geo", "attr")}
    pieces = {k: gs.frame_pieces(streams[k], 1 if k == "occ" else 2)[:N] for k in streams}
    got = ctx.transcode_gof([gs.first_pictures(streams[k], N * (1 if k == "occ" else 2)) for k in ("occ", "geo", "attr")], gs.rate_params(R, 3))
    assert got == oracle_streams(pieces, 24, 32)


def test_two_sequences_r1(ctx):
    """BASELINE.json configs[2]: two sequences, R5 -> R1. Two synthetic 1280x1280 sequences of 3 frames (seeds 1000 and 1450), coded here with the HM-like encoder in the
    CTC structure by worker processes, transcoded as two GOFs of one call and, again, as two jobs in flight; == the oracle frame by frame"""
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    nf, w, h = 3, 1280, 1280
    with tempfile.TemporaryDirectory() as td:
        tasks = [(kind, seed, f) for seed in (1000, 1450) for f in range(nf) for kind in ("attr", "geo", "occ")]
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "hm_encode_worker.py"), kind, str(seed), str(f), "1", str(w), str(h), os.path.join(td, f"{kind}_{seed}_{f}")]) for kind, seed, f in tasks]
        assert all(p.wait(timeout=900) == 0 for p in procs)
        seqs = [[b"".join(open(os.path.join(td, f"{kind}_{seed}_{f}"), "rb").read() for f in range(nf)) for kind in ("occ", "geo", "attr")] for seed in (1000, 1450)]
    P = gs.rate_params(R, 1)
    both = ctx.transcode_gof(seqs[0] + seqs[1], P * 2)
    ctx.set_depth(2)
    jobs = [ctx.submit_gof(s, P) for s in seqs]
    sep = [ctx.wait_gof(j) for j in jobs]
    assert both == sep[0] + sep[1]
    for s, got in zip(seqs, sep):
        pieces = {k: gs.frame_pieces(s[i], 1 if k == "occ" else 2) for i, k in enumerate(("occ", "geo", "attr"))}
        assert len(pieces["occ"]) == nf and got == oracle_streams(pieces, 32, 42)
    assert seqs[0][1] != seqs[1][1]
