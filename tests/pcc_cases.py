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


def seam_atlas(R, seed, tiles=4, prec=1, two_axes=False):
    """A dense surface cut into patches that MEET in 3-D, the situation geometry smoothing exists for (PCCCodec::smoothPointCloudPostprocess): tiles x tiles patches of
    2 x 2 blocks, each occupied but for a margin of 2 pixels (so its border points are boundary points, identifyBoundaryPoints), placed 28 apart in space so that the occupied
    areas of neighbours abut; depth = a smooth surface + an offset per patch (the seam a lossy geometry codec leaves) + noise; the far map a few levels behind.
    two_axes: every other patch projects along another axis (points of different patches then share cells away from the seams too)."""
    r = np.random.default_rng(7000 + seed)
    res, w = 16, tiles * 32
    patches, occ_full = [], np.zeros((w, w), bool)
    yy, xx = np.mgrid[0:w, 0:w]
    surf = 60 + 25 * np.sin(xx / 37.0 + seed) + 20 * np.cos(yy / 29.0)
    d0 = np.zeros((w, w))
    for j in range(tiles):
        for i in range(tiles):
            k = j * tiles + i
            flip = two_axes and (k % 2 == 1)
            axes = (1, 2, 0) if flip else (2, 0, 1)               # normal, tangent, bitangent
            patches.append(R.Patch(2 * i, 2 * j, 2, 2, 40 + 28 * i, 40 + 28 * j, 30, axes[0], axes[1], axes[2], 0, 0, 1, 1))
            sl = (slice(32 * j, 32 * j + 32), slice(32 * i, 32 * i + 32))
            occ_full[32 * j + 2:32 * j + 30, 32 * i + 2:32 * i + 30] = True
            d0[sl] = surf[sl] + int(r.integers(-6, 7)) + r.integers(-1, 2, (32, 32))
    holes = r.random((w, w)) < 0.01
    occ_full &= ~holes
    if prec > 1: occ_full = np.kron(occ_full[::prec, ::prec], np.ones((prec, prec), bool))
    d0 = np.clip(d0, 0, 255).astype(np.uint16) * 4                 # 10-bit video samples of 8-bit depths
    d1 = np.clip(d0 + 4 * r.integers(0, 4, (w, w)), 0, 1023).astype(np.uint16)
    occ = occ_full[::prec, ::prec].astype(np.uint16)
    t0 = r.integers(0, 1024, w * w * 3 // 2).astype(np.uint16)
    t1 = r.integers(0, 1024, w * w * 3 // 2).astype(np.uint16)
    atlas = R.AtlasParams(w, w, res, prec, 2, 1, 1, 0, 1, [8, 8, 4, 16, 6][seed % 5], [64, 64, 16, 64, 1][seed % 5])
    return atlas, patches, occ, d0, d1, 10, t0, t1, 10


def smooth_reference(xyz, boundary, partition, g, threshold):
    """Geometry smoothing once more, in numpy / Python straight from the reference's text (PCCCodec.cpp:52-145, :980-1104) with its types (float32 cell centres, float64
    filter): a second, independent writing of what oracle/pcc_recon.c smooth_grid restates, for the tests. xyz int16 [n, 3]; boundary bool [n]; partition [n] -> new xyz."""
    xyz = xyz.astype(np.int64); out = xyz.copy()
    if len(xyz) == 0: return out.astype(np.int16)
    w = (int(xyz.max()) + g - 1) // g
    disth, th = max(g // 2, 1), g * w
    inside = ~((xyz < disth).any(axis=1) | (th <= xyz + disth).any(axis=1))
    cells = {}
    lo = np.where(xyz % g < g // 2, -1, 0) + xyz // g
    for i in np.nonzero(boundary & inside)[0]:
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    cells.setdefault((lo[i, 0] + dx, lo[i, 1] + dy, lo[i, 2] + dz), [np.zeros(3, np.float32), 0, None, False])
    for j in np.nonzero(inside)[0]:
        c = cells.get(tuple(xyz[j] // g))
        if c is None: continue
        if c[1] == 0: c[2], c[3] = partition[j] + 1, False
        elif not c[3] and c[2] != partition[j] + 1: c[3] = True
        c[0] = (c[0] + xyz[j].astype(np.float32)).astype(np.float32); c[1] += 1
    for c in cells.values():
        if c[1]: c[0] = (c[0] / np.float32(c[1])).astype(np.float32)
    g2 = 2 * g
    for i in np.nonzero(boundary & inside)[0]:
        P = xyz[i]; S = lo[i]
        nb = [[[cells[(S[0] + dx, S[1] + dy, S[2] + dz)] for dx in (0, 1)] for dy in (0, 1)] for dz in (0, 1)]
        if not any(c[3] and c[1] for pl in nb for ro in pl for c in ro): continue
        W = (P - S * g - g // 2) * 2 + 1; Q = g2 - W
        c4 = np.zeros(3, np.float64); cnt = 0
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    c = nb[dz][dy][dx]; abc = int((W[0] if dx else Q[0]) * (W[1] if dy else Q[1]) * (W[2] if dz else Q[2]))
                    c4 = c4 + (c[0].astype(np.float64) if c[1] > 0 else P.astype(np.float64)) * np.float64(abc)
                    cnt += abc * c[1]
        c4 = c4 / np.float64(g2 ** 3); cnt //= g2 ** 3
        if cnt == 0: continue
        centroid = c4 * np.float64(cnt)
        d = P.astype(np.float64) * np.float64(cnt) - centroid
        dist2 = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / np.float64(cnt) + 0.5
        if dist2 >= max(int(threshold), cnt) * 2:
            out[i] = np.floor(centroid / np.float64(cnt) + 0.5).astype(np.int64)
    return out.astype(np.int16)


def boundary_reference(om, x, y):
    """PCCCodec::identifyBoundaryPoints (:266-325) for an occupied pixel, from the text, in Python"""
    H, W = om.shape; t = False
    if 0 < y < H - 1 and (om[y - 1, x] == 0 or om[y + 1, x] == 0): t = True
    if 0 < x < W - 1 and not t and (om[y, x + 1] == 0 or om[y, x - 1] == 0): t = True
    if 0 < y < H - 1 and x > 0 and not t and (om[y - 1, x - 1] == 0 or om[y + 1, x - 1] == 0): t = True
    if 0 < y < H - 1 and x < W - 1 and not t and (om[y - 1, x + 1] == 0 or om[y + 1, x + 1] == 0): t = True
    if y in (0, H - 1) or x in (0, W - 1): t = True
    if not t:
        for ix in range(-2, 3):
            for iy in range(-2, 3):
                if (abs(ix) > 1 or abs(iy) > 1) and 0 <= y + iy < H and 0 <= x + ix < W and om[y + iy, x + ix] == 0: t = True
        if y in (1, H - 2) or x in (1, W - 2): t = True
    return t
