"""Encoder laboratory (development tool, CPU oracle only): what a change of RBT-E1 does to bytes and quality.

Transcodes the first K point-cloud frames of the committed R5 fixture to a rate point with the ORACLE (the GPU encoder mirrors it bit for
bit, so the figures carry over) and prints, per sub-bitstream, bytes and luma PSNR of the output pictures against (a) the decoded R5 input
and (b) the uncoded synthetic source, plus D1 of point-cloud frame 0 against the source cloud. With --anchor it also codes the SOURCE maps
directly at the target QPs with the oracle's HM-like mode (the same-data anchor next to BASELINE.md's R3 / R1 rows).

    python tests/enc_lab.py [--frames 4] [--rate 3] [--anchor] [--env RBT_ENC_RDOQ=0 ...]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rabbit-transcoding_amd"))

RATE_POINTS = {1: (32, 42, 4), 2: (28, 37, 4), 3: (24, 32, 4), 4: (20, 27, 4), 5: (16, 22, 2)}
W = H = 1280
SEED = 1051


def split_pairs(stream: bytes):
    import gof_shard as gs
    return gs.split_pairs(stream)


def fixture(kind):
    return open(os.path.join(HERE, "golden", f"hm_r5_{W}x{H}_f32_{kind}.annexb"), "rb").read()


def _job(args):
    kind, idx, piece, qp, rows, env = args
    os.environ.update(env)
    import oracle_lib as O
    t = time.time()
    if kind == "gof":      # one point-cloud frame as transcodeData sees it: (occupancy, geometry, attribute), occupancy-aware coding on
        gq, aq = qp
        out = O.transcode_data(list(piece), [(0, 8, 4, 5, rows, 0, 0), (1, gq, 4, 5, rows, 0, 1), (19, aq, 4, 5, rows, 0, 1)])
        return kind, idx, out, time.time() - t
    vt = {"occ": 0, "geo": 1, "attr": 19}[kind]
    out = O.transcode_substream(piece, vt, qp, 4, 5, rows, 0)
    return kind, idx, out, time.time() - t


def psnr(a, b, peak=1023.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10 * np.log10(peak * peak / mse) if mse > 0 else float("inf")


def occ_mask_full(k):
    """full-resolution occupancy of point-cloud frame k of the benchmark GOF (synth.make_gof_maps: base k % 4 rolled by 2 * (k // 4))"""
    import synth
    return np.roll(synth.make_maps(W, H, SEED + k % 4)["occ_full"], 2 * (k // 4), axis=1)


def d1_frame(O, occ_stream, geo_stream, prec, k=0):
    """D1 of point-cloud frame k (< 4: the base atlases, not rolled) of the output against the cloud of the uncoded source maps"""
    import synth
    import rbt_lib
    R = rbt_lib.module()
    src = synth.make_maps(W, H, SEED + k)
    pats = synth.atlas_patches(R, W, H, SEED + k)

    def cloud(occ_plane, p, g2):
        return O.reconstruct(R.AtlasParams(W, H, 16, p, 2, 1, 1, 0), pats, occ_plane, g2[0][: W * H].reshape(H, W), g2[1][: W * H].reshape(H, W), 10)[0]
    c_src = cloud(src["occ_full"].astype(np.uint16), 1, src["geo"])
    od = O.decode(occ_stream)[0][0]
    ow = W // prec
    c_out = cloud(od[: ow * ow].reshape(ow, ow), prec, O.decode(geo_stream)[0])
    return O.d1(c_src, c_out)["psnr"], int(c_out.shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--rate", type=int, default=3)
    ap.add_argument("--rows", type=int, default=-1)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--anchor", action="store_true")
    ap.add_argument("--f4", action="store_true", help="occupancy-aware coding (occupancy_rd) through oracle_transcode_data")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-d1", action="store_true")
    ap.add_argument("--d1-frames", type=int, default=4, help="D1 of the first n point-cloud frames (at most 4: the base atlases)")
    ap.add_argument("--gqp", type=int, default=0, help="geometry QP instead of the rate point's")
    ap.add_argument("--aqp", type=int, default=0, help="attribute QP instead of the rate point's")
    a = ap.parse_args()
    env = dict(e.split("=", 1) for e in a.env)
    os.environ.update(env)
    import oracle_lib as O
    import synth
    O.build()
    gq, aq, prec = RATE_POINTS[a.rate]
    gq, aq = a.gqp or gq, a.aqp or aq
    k = a.frames
    ins = {kind: split_pairs(fixture(kind))[:k] for kind in ("occ", "geo", "attr")}
    tasks = []
    nd1 = 0 if a.no_d1 else min(k, a.d1_frames, 4)
    if a.f4:
        for i in range(k):
            tasks.append(("gof", i, (ins["occ"][i], ins["geo"][i], ins["attr"][i]), (gq, aq), a.rows, env))
    else:
        for i in range(k):
            tasks += [("attr", i, ins["attr"][i], aq, a.rows, env), ("geo", i, ins["geo"][i], gq, a.rows, env)]
        for i in range(max(nd1, 1)):
            tasks.append(("occ", i, ins["occ"][i], 8, a.rows, env))
    t0 = time.time()
    with ProcessPoolExecutor(min(a.jobs, len(tasks))) as ex:
        res = list(ex.map(_job, tasks))
    outs = {"geo": [None] * k, "attr": [None] * k, "occ": [None] * k}
    cpu_s = 0.0
    for kind, idx, out, dt in res:
        if kind == "gof":
            outs["occ"][idx], outs["geo"][idx], outs["attr"][idx] = out
        else:
            outs[kind][idx] = out
        cpu_s += dt
    geo_src, attr_src, _ = synth.make_gof_maps(W, H, k, SEED)
    rep = {"frames": k, "rate": a.rate, "f4": bool(a.f4), "env": env, "wall_s": round(time.time() - t0, 1), "cpu_s": round(cpu_s, 1)}
    for kind, src in (("geo", geo_src), ("attr", attr_src)):
        b_in = sum(len(x) for x in ins[kind])
        b_out = sum(len(x) for x in outs[kind])
        d_in = np.concatenate([O.decode(x)[0] for x in ins[kind]])[:, : W * H]
        d_out = np.concatenate([O.decode(x)[0] for x in outs[kind]])[:, : W * H]
        s = src[:, : W * H]
        # occupied samples only: what the decoder's reconstruction reads
        m = np.stack([occ_mask_full(i // 2).ravel() for i in range(2 * k)]) > 0
        rep[kind] = {"bytes_in": b_in, "bytes_out": b_out, "ratio": round(b_out / b_in, 4),
                     "psnr_vs_input": round(psnr(d_in, d_out), 3), "psnr_vs_source": round(psnr(s, d_out), 3),
                     "psnr_occ_vs_input": round(psnr(d_in[m], d_out[m]), 3), "psnr_occ_vs_source": round(psnr(s[m], d_out[m]), 3),
                     "input_psnr_vs_source": round(psnr(s, d_in), 3)}
    tot_in = rep["geo"]["bytes_in"] + rep["attr"]["bytes_in"]
    tot_out = rep["geo"]["bytes_out"] + rep["attr"]["bytes_out"]
    rep["ratio_geo_attr"] = round(tot_out / tot_in, 4)
    if nd1:
        ds = [d1_frame(O, outs["occ"][i], outs["geo"][i], prec, i) for i in range(nd1)]
        rep["d1_frame0"] = round(ds[0][0], 3)
        rep["points_frame0"] = ds[0][1]
        rep["d1_frames"] = [round(d[0], 3) for d in ds]
        rep["d1_mean"] = round(float(np.mean([d[0] for d in ds])), 3)
    if a.anchor:
        # direct encode of the SOURCE maps at the target QPs with the oracle's HM-like mode (CTC toolset): BASELINE.md's Rn rows on this data
        def enc(frames, qp, pq):
            return O.encode_hm(frames, W, H, 10, qp, gop=2, i_qp_offset=-3, p_qp_offset=pq, md5_sei=0, want_recon=True)
        an = {}
        for kind, src, qp, pq in (("geo", geo_src, gq, -3), ("attr", attr_src, aq, 0)):
            nb = 0
            recs = []
            for i in range(k):
                bs, rec = enc(src[2 * i: 2 * i + 2], qp, pq)
                nb += len(bs)
                recs.append(rec)
            rec = np.concatenate(recs)[:, : W * H]
            an[kind] = {"bytes": nb, "psnr_vs_source": round(psnr(src[:, : W * H], rec), 3)}
        rep["anchor"] = an
    print(json.dumps(rep, indent=1))
    if a.json:
        json.dump(rep, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
