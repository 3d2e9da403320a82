"""Seeded random atlases for the verification-stage tests (shared by the CPU and the GPU test)."""
import numpy as np


def random_atlas(R, seed, w=None, h=None):
    r = np.random.default_rng(1000 + seed)
    res = 16
    prec = [1, 2, 4, 2][seed % 4]
    w = w or int(r.integers(4, 9)) * res
    h = h or int(r.integers(4, 9)) * res
    bw, bh = w // res, h // res
    used = np.zeros((bh, bw), bool)
    patches = []
    for _ in range(40):
        su, sv = int(r.integers(1, 4)), int(r.integers(1, 4))
        orient = int(r.integers(0, 8))
        sw, sh = (sv, su) if orient in (1, 5, 6, 7) else (su, sv)       # SWAP, ROT270, MROT90, ROT90 exchange the axes on the canvas
        if sw > bw or sh > bh: continue
        u0, v0 = int(r.integers(0, bw - sw + 1)), int(r.integers(0, bh - sh + 1))
        overlap = used[v0:v0 + sh, u0:u0 + sw].any()
        if overlap and r.random() < 0.8: continue                       # a few overlapping patches: the later one owns the shared blocks
        used[v0:v0 + sh, u0:u0 + sw] = True
        axes = [int(x) for x in r.permutation(3)]
        pm = int(r.integers(0, 2))
        patches.append(R.Patch(u0, v0, su, sv, int(r.integers(0, 300)), int(r.integers(0, 300)), int(r.integers(0, 200)) + (300 if pm else 0), axes[0], axes[1], axes[2], pm, orient,
                               1 if r.random() < 0.8 else 2, 1))
    occ_full = (r.random((h // prec, w // prec)) < 0.7)
    occ_full &= np.kron(used, np.ones((res // prec, res // prec), bool))
    thr = 0
    if seed % 3 == 0: occ = occ_full.astype(np.uint16) * int(r.integers(1, 255))
    else: occ = occ_full.astype(np.uint16)
    if seed % 8 == 7:
        # lossy occupancy (thresholdLossyOM = 2): samples in 1..2 are NOT occupied and must not make a patch own a block (PCCCodec.cpp:1599-1600 binarises the frame
        # before generateBlockToPatchFromOccupancyMapVideo :1754 reads it); some blocks hold nothing but such samples
        thr = 2
        low = np.kron(r.random((bh, bw)) < 0.4, np.ones((res // prec, res // prec), bool))
        occ = np.where(occ_full, np.where(low, r.integers(1, 3, occ_full.shape), r.integers(3, 200, occ_full.shape)), 0).astype(np.uint16)
    d0 = r.integers(0, 1024, (h, w)).astype(np.uint16)
    d1 = np.clip(d0.astype(int) + r.integers(0, 12, (h, w)) * (r.random((h, w)) < 0.5), 0, 1023).astype(np.uint16)
    t0 = r.integers(0, 1024, w * h * 3 // 2).astype(np.uint16)
    t1 = r.integers(0, 1024, w * h * 3 // 2).astype(np.uint16)
    atlas = R.AtlasParams(w, h, res, prec, 2, 1 if seed % 5 else 0, 1 if seed % 7 != 6 else 0, thr)
    return atlas, patches, occ, d0, d1, 10, t0, t1, 10
