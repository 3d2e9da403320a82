"""Same-data anchor for the quality half of the metric (round-2 review item 2): the SOURCE maps of the benchmark GOF (tests/synth.py make_gof_maps, seed 1051, the maps the
committed R5 fixture was coded from) coded DIRECTLY at the R3 and R1 QPs with the oracle's HM-like encoder mode - what BASELINE.md's R3 / R1 rows (test/R15.0_32Frames.xlsm
rows 27, 25: a direct encode at each rate point) are for 8i data. Records, per rate: bytes of the geometry and attribute streams (+ the occupancy map at precision 4, lossless),
luma PSNR of the reconstruction against the source, D1 / D2 of point-cloud frame 0 against the source cloud. bench.py puts these figures next to the transcode's
(quality.anchor_direct_encode): the transcoder should land at or slightly below them in quality (second-generation loss) at comparable bytes.

    python tests/golden/make_anchor.py [--frames 32] [--jobs 7]          (about 3 minutes per rate point on 7 cores)
writes tests/golden/anchor_direct_encode.json. CPU only (oracle); never run on the GPU box."""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "rabbit-transcoding_amd"))
RATE_POINTS = {1: (32, 42, 4), 2: (28, 37, 4), 3: (24, 32, 4), 4: (20, 27, 4)}   # cfg/rate/ctc-rN.cfg:5-11
W = H = 1280
SEED = 1051


def _encode(args):
    kind, frames, qp = args
    import oracle_lib as O
    if kind == "geo":
        bs, rec = O.encode_hm(frames, W, H, 10, qp, gop=2, i_qp_offset=-3, p_qp_offset=-3, md5_sei=0, want_recon=True)
    elif kind == "attr":
        bs, rec = O.encode_hm(frames, W, H, 10, qp, gop=2, i_qp_offset=-3, p_qp_offset=0, md5_sei=0, want_recon=True)
    else:
        bs, rec = O.encode_hm(frames, W // 4, H // 4, 8, 8, gop=1, i_qp_offset=0, lossless=1, md5_sei=0, want_recon=False)
    return len(bs), (rec[:, : W * H] if rec is not None else None)


def psnr(a, b, peak=1023.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10 * np.log10(peak * peak / mse) if mse > 0 else float("inf")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--rates", type=int, nargs="*", default=[3, 1])
    a = ap.parse_args()
    import oracle_lib as O
    import rbt_lib
    import synth
    O.build()
    R = rbt_lib.module()
    n = a.frames
    geo, attr, occ2 = synth.make_gof_maps(W, H, n, SEED)
    # occupancy video of a direct encode at precision 4: the 4x4 OR of the atlas occupancy
    occ4 = []
    for k in range(n):
        full = np.roll(synth.make_maps(W, H, SEED + k % 4)["occ_full"], 2 * (k // 4), axis=1)
        y = full.reshape(H // 4, 4, W // 4, 4).max(axis=(1, 3)).astype(np.uint16)
        occ4.append(np.concatenate([y.ravel(), np.full((W // 4) * (H // 4) // 2, 128, np.uint16)]))
    occ4 = np.stack(occ4)
    src0 = synth.make_maps(W, H, SEED)
    c_src, n_src = synth.source_normals(R, O.reconstruct, W, H, SEED, src0["occ_full"], src0["geo"])
    pats = synth.atlas_patches(R, W, H, SEED)
    out = {"generator": "tests/golden/make_anchor.py", "frames": n, "width": W, "height": H, "seed": SEED,
           "encoder": "oracle HM-like mode (oracle/hevc_enc.c hm_like): CTU 64, TU 4..32 with trees, 35 intra modes + NxN, motion search, AMP, transform skip, sign hiding, SAO; decisions by SAD + lambda * bits",
           "rates": {}}
    for rate in a.rates:
        gq, aq, _ = RATE_POINTS[rate]
        tasks = []
        for i in range(n):
            tasks += [("attr", attr[2 * i:2 * i + 2], aq), ("geo", geo[2 * i:2 * i + 2], gq), ("occ", occ4[i:i + 1], 8)]
        with ProcessPoolExecutor(a.jobs) as ex:
            res = list(ex.map(_encode, tasks))
        ab, gb, ob = sum(r[0] for r in res[0::3]), sum(r[0] for r in res[1::3]), sum(r[0] for r in res[2::3])
        arec, grec = np.concatenate([r[1] for r in res[0::3]]), np.concatenate([r[1] for r in res[1::3]])
        g0 = res[1][1]
        cloud = O.reconstruct(R.AtlasParams(W, H, 16, 4, 2, 1, 1, 0), pats, occ4[0][: (W // 4) * (H // 4)].reshape(H // 4, W // 4), g0[0].reshape(H, W), g0[1].reshape(H, W), 10)[0]
        d1, d2 = O.d1(c_src, cloud), O.d2(c_src, n_src, cloud)
        out["rates"][f"R{rate}"] = {"geometry_qp": gq, "attribute_qp": aq, "occupancy_precision": 4,
                                    "bytes": {"occupancy": ob, "geometry": gb, "attribute": ab, "total": ob + gb + ab},
                                    "geometry_psnr_y_vs_source_db": round(psnr(geo[:, : W * H], grec), 3), "attribute_psnr_y_vs_source_db": round(psnr(attr[:, : W * H], arec), 3),
                                    "d1_psnr_frame0_vs_source_db": round(d1["psnr"], 3), "d2_psnr_frame0_vs_source_db": round(d2["psnr"], 3), "points_frame0": int(cloud.shape[0])}
        print(f"R{rate}", out["rates"][f"R{rate}"], flush=True)
    json.dump(out, open(os.path.join(HERE, "anchor_direct_encode.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
