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
