"""GPU build of the verification stage through the C ABI vs the oracle: point lists identical (order included),
occupancy / block-to-patch integer maps identical, D1 sums identical; random atlases with every orientation, plus the benchmark's full-size atlas
after a real R5 -> R3 transcode."""
import json
import os
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import pcc_cases
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    R = rbt_lib.module()
    c = R.Context(device=0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(8))
def test_reconstruction_matches_oracle_point_for_point(ctx, seed):
    R = rbt_lib.module()
    case = pcc_cases.random_atlas(R, seed, *((1280, 1280) if seed == 7 else (None, None)))
    got, want = ctx.reconstruct(*case), O.reconstruct(*case)
    assert got[0].shape == want[0].shape and got[0].shape[0] > 0
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("seed", range(4))
def test_d1_matches_oracle(ctx, seed):
    r = np.random.default_rng(seed)
    n = 200000 if seed == 0 else 3000
    a = r.integers(100, 400 if seed == 0 else 160, (n, 3)).astype(np.int16)
    b = np.clip(a[r.permutation(n)[: n * 5 // 6]] + r.integers(-2, 3, (n * 5 // 6, 3)), 0, 1023).astype(np.int16)
    if seed == 3: b = np.concatenate([b, np.array([[230, 40, 300]], np.int16)])
    got, want = ctx.d1(a, b), O.d1(a, b)
    for k in ("n_a", "n_b", "sse_ab", "sse_ba", "max_ab", "max_ba"):
        assert got[k] == want[k], k
    assert got["psnr"] == pytest.approx(want["psnr"], abs=1e-4)


def test_full_size_frame_after_transcode(ctx):
    """point-cloud frame 0 of the benchmark fixture: decode the R5 input and the R3 output of the whole path, rebuild both clouds from the synthetic
    atlas, compare with the oracle's reconstruction of the same maps, and D1 between them"""
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32"]
    gof = [gs.split_pairs(open(os.path.join(GOLD, man["streams"][k]["file"]), "rb").read())[0] for k in ("occ", "geo", "attr")]
    out = ctx.transcode_gof(gof, gs.rate_params(R, 3))
    w = h = 1280
    patches = synth.atlas_patches(R, w, h, 1051)
    clouds = []
    for streams, prec in ((gof, 2), (out, 4)):
        occ = ctx.decode(streams[0])[0][0][: (w // prec) * (h // prec)].reshape(h // prec, w // prec)
        geo = ctx.decode(streams[1])[0]; att = ctx.decode(streams[2])[0]
        d0, d1 = geo[0][: w * h].reshape(h, w), geo[1][: w * h].reshape(h, w)
        atlas = R.AtlasParams(w, h, 16, prec, 2, 1, 1, 0)
        got = ctx.reconstruct(atlas, patches, occ, d0, d1, 10, att[0], att[1], 10)
        want = O.reconstruct(atlas, patches, occ, d0, d1, 10, att[0], att[1], 10)
        for g, x in zip(got, want):
            assert np.array_equal(g, x)
        assert got[0].shape[0] > 100000
        clouds.append(got[0])
    got, want = ctx.d1(clouds[0], clouds[1]), O.d1(clouds[0], clouds[1])
    for k in ("n_a", "n_b", "sse_ab", "sse_ba", "max_ab", "max_ba"):
        assert got[k] == want[k], k
    assert 40 < got["psnr"] < 100


def test_d2_matches_oracle_and_brute_force(ctx):
    """rbt_d2 on the GPU (hash map voxel -> lowest index, ties walked in the bit volume, integer normal sums, double atomics) == oracle == brute force (tests/test_pcc_recon.py)"""
    import test_pcc_recon as T
    T.check_d2(ctx)


def test_d2_of_a_synthetic_frame(ctx):
    """source cloud of a 256x256 synthetic atlas with patch-axis normals against the cloud of perturbed depth maps: GPU == oracle, D2 PSNR >= D1 PSNR"""
    R = rbt_lib.module()
    w = h = 256
    src = synth.make_maps(w, h, 77)
    xyz, n = synth.source_normals(R, ctx.reconstruct, w, h, 77, src["occ_full"], src["geo"])
    x2, n2 = synth.source_normals(R, O.reconstruct, w, h, 77, src["occ_full"], src["geo"])
    assert np.array_equal(xyz, x2) and np.array_equal(n, n2) and xyz.shape[0] > 1000
    r = np.random.default_rng(1)
    g = src["geo"].copy(); g[:, : w * h] = np.clip(g[:, : w * h].astype(int) + 4 * r.integers(-1, 2, (2, w * h)), 0, 1023)
    pats = synth.atlas_patches(R, w, h, 77)
    occ4 = src["occ_full"].reshape(h // 4, 4, w // 4, 4).max(axis=(1, 3)).astype(np.uint16)
    dec = ctx.reconstruct(R.AtlasParams(w, h, 16, 4, 2, 1, 1, 0), pats, occ4, g[0][: w * h].reshape(h, w), g[1][: w * h].reshape(h, w), 10)[0]
    got, want = ctx.d2(xyz, n, dec), O.d2(xyz, n, dec)
    assert (got["n_a"], got["n_b"]) == (want["n_a"], want["n_b"]) and got["sse_ab"] == pytest.approx(want["sse_ab"], rel=1e-9) and got["sse_ba"] == pytest.approx(want["sse_ba"], rel=1e-9)
    assert got["psnr"] >= ctx.d1(xyz, dec)["psnr"] - 1e-3


@pytest.mark.parametrize("seed", range(10))
def test_geometry_smoothing_matches_oracle_on_the_gpu(ctx, seed):
    """rbt_atlas_params.geometry_smoothing on the GPU (k_sm_mark / k_sm_accum / k_sm_filter: one lane per point, cell centroids by atomics, the filter in the reference's
    float / double arithmetic without contraction): == oracle for every point of seam atlases (patches that meet in space; tests/pcc_cases.py), seed 9 at 1280 x 1280"""
    R = rbt_lib.module()
    case = pcc_cases.seam_atlas(R, seed, tiles=40 if seed == 9 else 3 + seed % 3, prec=[1, 2, 4][seed % 3], two_axes=seed % 2 == 1)
    if seed == 9:      # 40 x 40 tiles of 28 points would leave the 1024^3 volume: fold the grid of patches back into it
        for k, p in enumerate(case[1]): p.u1 = 40 + 28 * (k % 30); p.v1 = 40 + 28 * ((k // 30) % 30); p.d1 = 30 + 200 * (k // 900)
    got, want = ctx.reconstruct(*case), O.reconstruct(*case)
    assert got[0].shape == want[0].shape and got[0].shape[0] > 5000
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    assert ctx.n_smoothed == O.LAST_SMOOTHED and (ctx.n_smoothed > 100 or case[0].grid_size == 4)


def test_geometry_smoothing_random_atlases_and_off_switch(ctx):
    """random atlases (every orientation, sparse points: the smoothing has nothing to blend and must leave the cloud alone, as the oracle does); grid_size out of range is refused"""
    R = rbt_lib.module()
    for seed in range(8):
        case = pcc_cases.random_atlas(R, seed)
        a = R.ctc_smoothing(case[0])
        got, want = ctx.reconstruct(a, *case[1:]), O.reconstruct(a, *case[1:])
        assert all(np.array_equal(g, w_) for g, w_ in zip(got, want)) and ctx.n_smoothed == O.LAST_SMOOTHED
    bad = R.ctc_smoothing(case[0]); bad.grid_size = 1
    with pytest.raises(R.RbtError):
        ctx.reconstruct(bad, *case[1:])
