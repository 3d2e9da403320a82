"""GPU parity: HIP decode path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    R = rbt_lib.module()
    c = R.Context(device=0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(1, 41))
def test_stress_streams(ctx, seed):
    """random-syntax streams: all intra modes, NxN, TU trees, TS, bypass, AMP, AMVP/merge, TMVP, SAO, dQP, SDH, slices"""
    w = [64, 96, 128, 80][seed % 4]; h = [64, 80, 48, 128][(seed // 4) % 4]
    bd = 10 if seed % 3 else 8
    fr = np.zeros((5, w * h * 3 // 2), np.uint16)
    bs, rec = O.encode(fr, w, h, bd, qp=30, gop=2, stress_seed=seed, log2_ctb=0)
    dec, dw, dh, dbd, chk, fail = ctx.decode(bs)
    assert (dw, dh, dbd) == (w, h, bd) and chk == 5 and fail == 0
    assert np.array_equal(dec, rec)


@pytest.mark.parametrize("log2_ctb,rows", [(5, 1), (6, 0), (4, 2), (5, -1), (6, -1), (5, -2), (4, -2)])   # -1: wavefront rows as dependent slice segments, -2: behind entry points (x265's form)
def test_product_streams(ctx, log2_ctb, rows):
    m = synth.make_maps(256, 192, 31)
    for key, qp in (("geo", 16), ("attr", 22)):
        bs, rec = O.encode(m[key], 256, 192, 10, qp, gop=2, log2_ctb=log2_ctb, rows_per_slice=rows)
        dec, w, h, bd, chk, fail = ctx.decode(bs)
        assert fail == 0 and chk == 2 and np.array_equal(dec, rec)
    bs, rec = O.encode(m["occ"], 128, 96, 8, 8, gop=1, lossless=1, log2_ctb=log2_ctb, rows_per_slice=rows)
    dec, *_ = ctx.decode(bs)
    assert np.array_equal(dec, m["occ"])


def test_full_size_frame_pair(ctx):
    """BASELINE.json size: 1280x1280 10-bit I/P pair, HM-like structure (CTB 64, one slice per picture); MD5 SEI self-check"""
    m = synth.make_maps(1280, 1280, 1051)
    bs, rec = O.encode(m["attr"], 1280, 1280, 10, 22, gop=2, log2_ctb=6, rows_per_slice=0)
    dec, w, h, bd, chk, fail = ctx.decode(bs)
    assert (w, h, bd, chk, fail) == (1280, 1280, 10, 2, 0)
    assert np.array_equal(dec, rec)


def test_corrupt_stream_is_rejected(ctx):
    R = rbt_lib.module()
    m = synth.make_maps(64, 64, 5)
    bs, _ = O.encode(m["geo"], 64, 64, 10, 24, gop=2)
    with pytest.raises(R.RbtError):
        ctx.decode(bs[:40])

@pytest.mark.parametrize("w", [1536, 1552, 4096, 4112])
def test_wide_pictures_use_the_larger_parser_variants(ctx, w):
    """the slice parser's LDS line buffers come in three sizes (pictures up to 1536 / 4096 / 8192 samples wide): widths on
    both sides of each boundary, several CTB rows so that every above-neighbour path reads the line buffers"""
    h = 48
    r = np.random.default_rng(w)
    fr = r.integers(0, 1024, (2, w * h * 3 // 2)).astype(np.uint16)
    fr[1] = np.clip(fr[0].astype(int) + r.integers(-2, 3, fr[0].shape), 0, 1023)
    for log2_ctb, seed in ((4, 0), (6, 7)):
        bs, rec = O.encode(fr, w, h, 10, qp=34, gop=2, stress_seed=seed, log2_ctb=log2_ctb)
        dec, dw, dh, dbd, chk, fail = ctx.decode(bs)
        assert (dw, dh, fail) == (w, h, 0) and np.array_equal(dec, rec)


def test_slice_segments_must_tile_the_picture(ctx):
    """missing / repeated / swapped slice segments on the GPU: refused by the first wave that sees the hole or the overlap (RbtSlice::end_addr), row tasks below do not
    wait out their bound, the context stays usable (tests/test_hostemu_parity.py slice_segment_damage)"""
    import time
    import test_hostemu_parity as T
    t0 = time.time()
    T.slice_segment_damage(ctx, rbt_lib.module())
    assert time.time() - t0 < 60


@pytest.mark.parametrize("mode", ["RBT_RECON_QUEUE", "RBT_RECON_LEVEL", "RBT_RECON_DIAG"])
def test_every_reconstruction_mode_on_the_gpu(mode):
    """one launch per anti-diagonal / per level with neighbour flags / per level with a device-side ready queue (k_recon_queue, round 4: persistent workgroups, arrival
    counts, no host involvement): same pictures, same streams, at depths 1, 4 and 16, incl. a full-size picture pair (tests/recon_mode_worker.py in a child process, the
    mode is read once per process). The default mixes the first two by depth; the ready queue measured slower under load (profiles/r04_recon_mode_sweep.txt) and stays opt-in."""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "recon_mode_worker.py"), "gpu"], env=dict(os.environ, **{mode: "1"}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("OK 13"), (r.stdout[-500:], r.stderr[-3000:])
