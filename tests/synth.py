"""Synthetic V-PCC-like video maps: patch-packed geometry / attribute / occupancy frames.

There is no 8i data and no HM here or on the GPU box, so the transcoder input is generated: an atlas of axis-aligned
16-px aligned rectangular patches with smooth depth (geometry), textured colour (attribute) and a binary occupancy map.
Frame pairs (D0,D1)/(T0,T1) mimic the two V-PCC maps: the far map differs from the near map by a small offset.
"""
import numpy as np


def atlas_layout(w, h, seed, n_patches=None):
    r = np.random.default_rng(seed)
    occ = np.zeros((h, w), np.uint8)
    patches = []
    n = n_patches or max(3, (w * h) // (160 * 160))
    for _ in range(n * 3):
        pw = int(r.integers(2, max(3, w // 64))) * 16
        ph = int(r.integers(2, max(3, h // 64))) * 16
        if pw > w or ph > h:
            continue
        x = int(r.integers(0, (w - pw) // 16 + 1)) * 16
        y = int(r.integers(0, (h - ph) // 16 + 1)) * 16
        if occ[y:y + ph, x:x + pw].any():
            continue
        # blob-shaped occupancy inside the patch rectangle
        yy, xx = np.mgrid[0:ph, 0:pw]
        cx, cy = pw / 2, ph / 2
        rad = ((xx - cx) / (pw / 2)) ** 2 + ((yy - cy) / (ph / 2)) ** 2
        blob = rad < 0.95 + 0.1 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0)
        occ[y:y + ph, x:x + pw] = blob
        patches.append((x, y, pw, ph))
        if len(patches) >= n:
            break
    return occ, patches


def _smooth_fill(v, mask):
    """TMC2 pads unoccupied map pixels smoothly (push-pull); this is a light version: normalised box blurs, coarse to fine."""
    from scipy.ndimage import uniform_filter
    m = mask.astype(np.float64)
    out = np.full(v.shape, float((v * m).sum() / max(1.0, m.sum())))
    for k in (129, 33, 9, 3):
        num = uniform_filter(v * m, k, mode="nearest")
        den = uniform_filter(m, k, mode="nearest")
        out = np.where(den > 1e-3, num / np.maximum(den, 1e-3), out)
    return np.where(mask > 0, v, out)


def make_maps(w, h, seed, bit_depth=10, jitter=0):
    """-> dict with 'geo' [2, w*h*3/2], 'attr' [2, ...] uint16 (10-bit carried as 8-bit*4), 'occ' [1, (w/2*h/2)*3/2] (8-bit, precision 2)"""
    occ, patches = atlas_layout(w, h, seed)
    r = np.random.default_rng(seed + 7919 + jitter)
    yy, xx = np.mgrid[0:h, 0:w]
    d0 = np.zeros((h, w), np.float64)
    col = np.zeros((3, h, w), np.float64)
    for i, (x, y, pw, ph) in enumerate(patches):
        py, px = np.mgrid[0:ph, 0:pw]
        base = 40 + (i * 37) % 150
        depth = base + 10 * np.sin(px / (pw / 2.0) + i) * np.cos(py / (ph / 1.8)) + 0.04 * px
        d0[y:y + ph, x:x + pw] = depth
        for c in range(3):
            tex = 128 + 38 * np.sin(px / (5.0 + c) + i * 1.3) * np.cos(py / (6.0 + i % 3)) + 20 * np.sin((px + py) / (11.0 + c))
            col[c, y:y + ph, x:x + pw] = tex + r.normal(0, 1.6, (ph, pw))
    # integer depth surfaces; the far map D1 exceeds the near map D0 by 0..3 on a smooth sub-region (surface thickness)
    d0 = np.clip(np.round(d0), 0, 255) * occ
    thick = (np.sin(xx / 23.0 + seed) * np.cos(yy / 31.0) > 0.55) * (1 + (np.sin(xx / 5.0) > 0.3))
    d1 = np.clip(d0 + thick, 0, 255) * occ
    # unoccupied pixels: smooth padding (the TMC2 encoder dilates patches); a flat mid value is enough here
    pad_geo = 0
    scale = 1 << (bit_depth - 8)

    def pack(y_plane, u=None, v=None):
        Y = (y_plane.astype(np.int64) * scale).clip(0, (1 << bit_depth) - 1).astype(np.uint16)
        ch, cw = h // 2, w // 2
        U = np.full((ch, cw), 1 << (bit_depth - 1), np.uint16) if u is None else (u.astype(np.int64) * scale).clip(0, (1 << bit_depth) - 1).astype(np.uint16)
        V = np.full((ch, cw), 1 << (bit_depth - 1), np.uint16) if v is None else (v.astype(np.int64) * scale).clip(0, (1 << bit_depth) - 1).astype(np.uint16)
        return np.concatenate([Y.ravel(), U.ravel(), V.ravel()])

    del pad_geo
    f0 = _smooth_fill(d0, occ)
    geo = np.stack([pack(np.round(f0)), pack(np.round(np.where(occ, d1, f0)))])
    yv = 0.299 * col[0] + 0.587 * col[1] + 0.114 * col[2]
    uv = (128 + 0.5 * (col[2] - yv) / 0.886)[::2, ::2]
    vv = (128 + 0.5 * (col[0] - yv) / 0.701)[::2, ::2]
    m = np.where(occ, 1.0, 0.0)
    yf = _smooth_fill(yv, occ)
    y0 = yf
    y1 = np.where(occ, yv + r.normal(0, 0.7, (h, w)), yf)
    occ_c = occ[::2, ::2]
    uv = _smooth_fill(uv, occ_c); vv = _smooth_fill(vv, occ_c)
    attr = np.stack([pack(np.clip(y0, 0, 255), np.clip(uv, 0, 255), np.clip(vv, 0, 255)),
                     pack(np.clip(y1, 0, 255), np.clip(uv, 0, 255), np.clip(vv, 0, 255))])
    del m
    # occupancy video at precision 2: (w/2 x h/2), value 1 where any of the 2x2 block is occupied, 8-bit, chroma 128
    o2 = occ.reshape(h // 2, 2, w // 2, 2).max(axis=(1, 3)).astype(np.uint16)
    oc = np.full(((h // 4) * (w // 4),), 128, np.uint16)
    occ_frame = np.concatenate([o2.ravel(), oc, oc])[None, :]
    return {"geo": geo, "attr": attr, "occ": occ_frame, "occ_full": occ}


def make_gof(w, h, n_pc_frames, seed0, bit_depth=10):
    """One group of frames: geometry/attribute streams of 2*n frames, occupancy stream of n frames (precision 2)."""
    geo, attr, occ = [], [], []
    for i in range(n_pc_frames):
        m = make_maps(w, h, seed0, bit_depth, jitter=i)
        geo.append(m["geo"]); attr.append(m["attr"]); occ.append(m["occ"])
    return np.concatenate(geo), np.concatenate(attr), np.concatenate(occ)


def make_gof_maps(w, h, n_pc, seed):
    """n_pc point-cloud frames: 4 base atlases jittered by a few pixels per frame."""
    bases = [make_maps(w, h, seed + k) for k in range(min(4, n_pc))]
    geo, attr, occ = [], [], []
    ys, cs = w * h, (w // 2) * (h // 2)

    def roll(frame, ww, hh, d):
        y = np.roll(frame[: ww * hh].reshape(hh, ww), d, axis=1).ravel()
        c = (ww // 2) * (hh // 2)
        u = np.roll(frame[ww * hh: ww * hh + c].reshape(hh // 2, ww // 2), d // 2, axis=1).ravel()
        v = np.roll(frame[ww * hh + c:].reshape(hh // 2, ww // 2), d // 2, axis=1).ravel()
        return np.concatenate([y, u, v])
    for i in range(n_pc):
        b = bases[i % len(bases)]
        d = 2 * (i // len(bases))
        geo += [roll(b["geo"][0], w, h, d), roll(b["geo"][1], w, h, d)]
        attr += [roll(b["attr"][0], w, h, d), roll(b["attr"][1], w, h, d)]
        occ += [roll(b["occ"][0], w // 2, h // 2, d // 2)]
    del ys, cs
    return np.stack(geo), np.stack(attr), np.stack(occ)


def atlas_patches(R, w, h, seed):
    """rbt_patch list of the synthetic atlas make_maps(w, h, seed) draws (atlas_layout): every rectangle is a patch in default orientation,
    projected along one axis, placed in a 1024^3 volume (coordinates stay below 1024: peak 1023 of the D1 metric)."""
    _, rects = atlas_layout(w, h, seed)
    out = []
    for i, (x, y, pw, ph) in enumerate(rects):
        out.append(R.Patch(x // 16, y // 16, pw // 16, ph // 16, (i * 97) % (1023 - pw), (i * 53) % (1023 - ph), (i * 31) % 700, i % 3, (i + 1) % 3, (i + 2) % 3, 0, 0, 1, 1))
    return out


def atlas_index_picture(w, h, seed):
    """a 4:2:0 picture whose luma carries, at every pixel of a patch rectangle of make_maps(w, h, seed), that patch's index: handed to the reconstruction as
    "attribute" it comes back as the patch index of every point (the colour fetch reads the point's pixel)"""
    _, rects = atlas_layout(w, h, seed)
    y = np.zeros((h, w), np.uint16)
    for i, (x, yy, pw, ph) in enumerate(rects):
        y[yy:yy + ph, x:x + pw] = i
    return np.concatenate([y.ravel(), np.zeros(w * h // 2, np.uint16)])


def source_normals(R, reconstruct, w, h, seed, occ_full, geo):
    """the source cloud of make_maps(w, h, seed) with one normal per point: the projection axis of the point's patch (Q14, 16384 = 1.0), pointing the way the
    depth grows. reconstruct: ctx.reconstruct of the library or of the oracle (same signature)."""
    pats = atlas_patches(R, w, h, seed)
    idx = atlas_index_picture(w, h, seed)
    xyz, yuv, _, _ = reconstruct(R.AtlasParams(w, h, 16, 1, 2, 1, 1, 0), pats, occ_full.astype(np.uint16), geo[0][: w * h].reshape(h, w), geo[1][: w * h].reshape(h, w), 10, idx, idx, 10)
    axes = np.array([p.normal_axis for p in pats]); sign = np.array([1 if p.projection_mode == 0 else -1 for p in pats])
    pi = yuv[:, 0].astype(np.int64)
    n = np.zeros((xyz.shape[0], 3), np.int16)
    n[np.arange(xyz.shape[0]), axes[pi]] = 16384 * sign[pi]
    return xyz, n
