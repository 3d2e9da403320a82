"""Verification stage: reprojection of the geometry maps to points + colour fetch, and the D1 metric -
the kernel BODIES run as serial host code (tests/hostemu, no GPU here) against the oracle restatement of PCCCodec::generatePointCloud /
QualityMetrics::compute, on seeded random atlases that use every patch orientation, both projection modes and precisions 1 / 2 / 4.
The GPU build of the same is tests/test_gpu_pcc.py."""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
import rbt_lib
import pcc_cases


@pytest.fixture(scope="module")
def ctx():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    R = rbt_lib.module()
    c = R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(8))
def test_reconstruction_matches_oracle_point_for_point(ctx, seed):
    R = rbt_lib.module()
    case = pcc_cases.random_atlas(R, seed)
    got = ctx.reconstruct(*case)
    want = O.reconstruct(*case)
    assert got[0].shape == want[0].shape and got[0].shape[0] > 0
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_known_answer_single_patch(ctx):
    """one 16x16 patch, default orientation, projection mode 0: point (u + u1, v + v1, depth + d1) for every occupied pixel; D1 map equal -> no second point"""
    R = rbt_lib.module()
    atlas = R.AtlasParams(32, 32, 16, 1, 2, 1, 1, 0)
    occ = np.zeros((32, 32), np.uint16); occ[16:20, 0:3] = 1
    d0 = np.full((32, 32), 40, np.uint16); d1 = d0.copy(); d1[17, 1] = 44      # 10-bit samples: (40 + 2) >> 2 = 10, (44 + 2) >> 2 = 11
    p = R.Patch(0, 1, 1, 1, 100, 200, 7, 2, 0, 1, 0, 0, 1, 1)
    xyz, yuv, om, b2p = ctx.reconstruct(atlas, [p], occ, d0, d1, 10)
    assert b2p.tolist() == [[0, 0], [1, 0]] and om.sum() == 12
    want = []
    for v in range(4):
        for u in range(3):
            want.append([100 + u, 200 + v, 17])
            if (u, v) == (1, 1): want.append([101, 201, 18])
    assert xyz.tolist() == want and np.all(yuv == 512)


def test_lossy_occupancy_threshold_decides_block_ownership(ctx):
    """thresholdLossyOM = 2: generateOccupancyMap binarises the occupancy frame in place (PCCCodec.cpp:1599-1600) BEFORE
    generateBlockToPatchFromOccupancyMapVideo reads it (:1754), so a block whose samples are all in 1..2 is owned by no patch"""
    R = rbt_lib.module()
    atlas = R.AtlasParams(32, 32, 16, 1, 1, 1, 1, 2)
    occ = np.zeros((32, 32), np.uint16)
    occ[0:16, 0:16] = 2            # block (0, 0): only samples at the threshold
    occ[16:32, 0:16] = 1; occ[20, 5] = 3    # block (0, 1): one sample above it
    d0 = np.full((32, 32), 40, np.uint16)
    p = R.Patch(0, 0, 1, 2, 0, 0, 0, 2, 0, 1, 0, 0, 1, 1)
    for rec in (ctx.reconstruct, O.reconstruct):
        xyz, yuv, om, b2p = rec(atlas, [p], occ, d0, d0, 10)
        assert b2p.tolist() == [[0, 0], [1, 0]] and om.sum() == 1 and xyz.tolist() == [[5, 20, 10]]


@pytest.mark.parametrize("seed", range(4))
def test_d1_matches_oracle(ctx, seed):
    r = np.random.default_rng(seed)
    a = r.integers(100, 160, (3000, 3)).astype(np.int16)
    b = np.clip(a[r.permutation(3000)[:2500]] + r.integers(-2, 3, (2500, 3)), 0, 1023).astype(np.int16)
    if seed == 3: b = np.concatenate([b, np.array([[230, 40, 300]], np.int16)])   # an outlier: many shells / rings
    got, want = ctx.d1(a, b), O.d1(a, b)
    for k in ("n_a", "n_b", "sse_ab", "sse_ba", "max_ab", "max_ba"):
        assert got[k] == want[k], k
    assert got["psnr"] == pytest.approx(want["psnr"], abs=1e-4)
    # brute force on the unique points
    ua, ub = np.unique(a, axis=0).astype(np.int64), np.unique(b, axis=0).astype(np.int64)
    d = ((ua[:, None, :] - ub[None, :, :]) ** 2).sum(-1)
    assert got["sse_ab"] == int(d.min(1).sum()) and got["sse_ba"] == int(d.min(0).sum()) and got["n_a"] == len(ua) and got["n_b"] == len(ub)
    assert ctx.d1(a, a)["sse_ab"] == 0


def test_bad_arguments(ctx):
    R = rbt_lib.module()
    with pytest.raises(R.RbtError):
        ctx.d1(np.array([[0, 0, 2000]], np.int16), np.array([[0, 0, 0]], np.int16))
    atlas = R.AtlasParams(32, 32, 16, 1, 2, 1, 1, 0)
    z = np.zeros((32, 32), np.uint16)
    with pytest.raises(R.RbtError):
        ctx.reconstruct(atlas, [R.Patch(1, 1, 2, 1, 0, 0, 0, 2, 0, 1, 0, 0, 1, 1)], z, z, z)     # patch leaves the atlas


def d2_cases():
    """(a, normals_a, b) clouds for the point-to-plane metric, shared with tests/test_gpu_pcc.py: axis normals and slanted ones, duplicates on both sides, a sparse
    reconstruction (points of B that no source point is nearest to take their normal from the source) and an outlier"""
    out = []
    for seed in range(4):
        r = np.random.default_rng(40 + seed)
        a = r.integers(100, 140, (3000, 3)).astype(np.int16)
        if seed % 2:
            n = np.zeros((3000, 3), np.int16); n[np.arange(3000), r.integers(0, 3, 3000)] = 16384 * r.choice([-1, 1], 3000)
        else:
            v = r.normal(size=(3000, 3)); n = np.round(16384 * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.int16)
        b = np.clip(a[r.permutation(3000)[:2500 if seed < 3 else 300]] + r.integers(-2, 3, (2500 if seed < 3 else 300, 3)), 0, 1023).astype(np.int16)
        if seed == 2: b = np.concatenate([b, np.array([[230, 40, 300]], np.int16)])
        out.append((a, n, b))
    return out


def d2_brute_force(a, n, b):
    """the definition, by brute force on the merged points: (sum A -> B, sum B -> A, unique A, unique B)"""
    ua, ia = np.unique(a, axis=0, return_index=True); ub = np.unique(b, axis=0)
    na = n[ia].astype(np.float64)          # np.unique returns the FIRST occurrence: the lowest original index
    A, B = ua.astype(np.int64), ub.astype(np.int64)
    d = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1)
    tie_ab = d == d.min(1, keepdims=True); tie_ba = d == d.min(0, keepdims=True)
    acc = tie_ab.T.astype(np.float64) @ na; cnt = tie_ab.sum(0).astype(np.float64)
    for j in np.nonzero(cnt == 0)[0]:
        acc[j] = na[tie_ba[:, j]].sum(0); cnt[j] = tie_ba[:, j].sum()
    nb = acc / cnt[:, None]
    s_ab = s_ba = 0.0
    for i in range(len(A)):
        js = np.nonzero(tie_ab[i])[0]
        s_ab += np.mean(((A[i] - B[js]) * nb[js]).sum(1) ** 2)
    for j in range(len(B)):
        ii = np.nonzero(tie_ba[:, j])[0]
        s_ba += np.mean(((B[j] - A[ii]) * na[ii]).sum(1) ** 2)
    q = 16384.0 ** 2
    return s_ab / q, s_ba / q, len(A), len(B)


def check_d2(ctx):
    for a, n, b in d2_cases():
        got, want = ctx.d2(a, n, b), O.d2(a, n, b)
        assert (got["n_a"], got["n_b"]) == (want["n_a"], want["n_b"])
        for k in ("sse_ab", "sse_ba", "max_ab", "max_ba"):
            assert got[k] == pytest.approx(want[k], rel=1e-9), k          # sums of doubles in another order
        assert got["psnr"] == pytest.approx(want["psnr"], abs=1e-4) and got["psnr_ab"] == pytest.approx(want["psnr_ab"], abs=1e-4)
        s_ab, s_ba, n_a, n_b = d2_brute_force(a, n, b)
        assert (n_a, n_b) == (want["n_a"], want["n_b"]) and want["sse_ab"] == pytest.approx(s_ab, rel=1e-9) and want["sse_ba"] == pytest.approx(s_ba, rel=1e-9)
        # planes cannot be further away than points: D2 <= D1 per direction
        d1 = ctx.d1(a, b)
        assert got["sse_ab"] <= d1["sse_ab"] * (1 + 1e-9) and got["sse_ba"] <= d1["sse_ba"] * (1 + 1e-9)
    a, n, _ = d2_cases()[0]
    assert ctx.d2(a, n, a)["sse_ab"] == 0


def test_d2_matches_oracle_and_brute_force(ctx):
    check_d2(ctx)


def _no_smoothing(R, atlas):
    a = R.AtlasParams(*[getattr(atlas, n) for n, _ in R.AtlasParams._fields_]); a.geometry_smoothing = 0
    return a


@pytest.mark.parametrize("seed", range(10))
def test_geometry_smoothing_matches_oracle_point_for_point(ctx, seed):
    """rbt_atlas_params.geometry_smoothing (PCCCodec::smoothPointCloudPostprocess with gridSmoothing, what the CTC switches on): dense surfaces cut into patches that meet
    in space (pcc_cases.seam_atlas: grid sizes 4 / 6 / 8 / 16, thresholds 1 / 16 / 64, precisions 1 / 2 / 4, one or two projection axes). Product == oracle for every
    point, the count of moved points included; the smoothing moves points in most cases and only boundary points; everything else of the cloud is untouched."""
    R = rbt_lib.module()
    case = pcc_cases.seam_atlas(R, seed, tiles=3 + seed % 3, prec=[1, 2, 4][seed % 3], two_axes=seed % 2 == 1)
    got, want = ctx.reconstruct(*case), O.reconstruct(*case)
    assert got[0].shape == want[0].shape and got[0].shape[0] > 5000
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert ctx.n_smoothed == O.LAST_SMOOTHED
    plain = ctx.reconstruct(_no_smoothing(R, case[0]), *case[1:])
    assert ctx.n_smoothed == 0 and np.array_equal(plain[1], got[1]) and np.array_equal(plain[2], got[2])
    moved = (plain[0] != got[0]).any(axis=1)
    assert int(moved.sum()) <= O.LAST_SMOOTHED and (int(moved.sum()) > 100 or case[0].grid_size == 4)


@pytest.mark.parametrize("seed", [0, 3, 4, 8])
def test_geometry_smoothing_against_a_second_writing_of_the_reference_text(ctx, seed):
    """the same result from an independent numpy / Python transcription of the reference's loops (pcc_cases.smooth_reference / boundary_reference), fed with the
    boundary flags and patch indices recomputed from the maps: guards the oracle's restatement against slips of the pen (both are by the same reader: parity with the
    reference itself stays unpinned, PccLibCommon cannot be built here)"""
    R = rbt_lib.module()
    case = pcc_cases.seam_atlas(R, seed, tiles=3, prec=[1, 2, 4][seed % 3], two_axes=seed % 2 == 1)
    atlas, patches, occ = case[0], case[1], case[2]
    w = atlas.width
    idx = np.zeros((w, w), np.uint16)
    for k, p in enumerate(patches): idx[32 * p.v0 // 2:32 * p.v0 // 2 + 32, 32 * p.u0 // 2:32 * p.u0 // 2 + 32] = k
    plain = ctx.reconstruct(_no_smoothing(R, atlas), *case[1:6], np.concatenate([idx.ravel(), np.zeros(w * w // 2, np.uint16)]), np.concatenate([idx.ravel(), np.zeros(w * w // 2, np.uint16)]), 10)
    part = plain[1][:, 0].astype(np.int64)
    # the pixel of every point: reconstruct twice with pictures that carry x and y
    xs = ctx.reconstruct(_no_smoothing(R, atlas), *case[1:6], np.concatenate([np.tile(np.arange(w, dtype=np.uint16), w), np.zeros(w * w // 2, np.uint16)]), np.concatenate([np.tile(np.arange(w, dtype=np.uint16), w), np.zeros(w * w // 2, np.uint16)]), 10)[1][:, 0]
    ys = ctx.reconstruct(_no_smoothing(R, atlas), *case[1:6], np.concatenate([np.repeat(np.arange(w, dtype=np.uint16), w), np.zeros(w * w // 2, np.uint16)]), np.concatenate([np.repeat(np.arange(w, dtype=np.uint16), w), np.zeros(w * w // 2, np.uint16)]), 10)[1][:, 0]
    om = plain[2]
    boundary = np.array([pcc_cases.boundary_reference(om, int(x), int(y)) for x, y in zip(xs, ys)])
    want = pcc_cases.smooth_reference(plain[0], boundary, part, atlas.grid_size, atlas.threshold_smoothing)
    got = ctx.reconstruct(*case)[0]
    assert np.array_equal(got, want) and (got != plain[0]).any()
