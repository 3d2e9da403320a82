"""Generates the benchmark's R5 input: one 32-frame GOF of 1280x1280 synthetic V-PCC maps (tests/synth.py make_gof_maps, seed 1051) coded by
the ORACLE's HM-like encoder (oracle/hevc_enc.c, hm_like) with the toolset of the CTC input streams - cfg/hm/ctc-hm-geometry-ai.cfg /
ctc-hm-attribute-ai.cfg: CTU 64, TU 4..32, motion search, transform skip, SAO, AMP, decoded-picture-hash SEI; geometry QP 16 (P offset -3),
attribute QP 22 (I offset -3), cfg/rate/ctc-r5.cfg:5-6; occupancy precision 2, lossless all-intra (ctc-hm-occupancy-map-ai-main10.cfg).
There is no 8i data, no HM and no libx265 in the build container or on the GPU box, so this is the closest stand-in for an HM-coded longdress
R5 GOF that can be made here. Every point-cloud frame is a closed GOP (IDR + P, parameter sets repeated), so the frames are encoded in parallel
processes and concatenated.

    python tests/golden/make_hm_gof.py [--frames 32] [--width 1280] [--height 1280] [--jobs 7]
writes tests/golden/hm_r5_<w>x<h>_f<frames>_{occ,geo,attr}.annexb and prints their sizes and MD5s (recorded in hm_r5_manifest.json).
"""
import argparse
import hashlib
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _encode(args):
    kind, frames, w, h, ctc, first = args
    import oracle_lib as O
    if ctc:
        # the CTC encoder's stream structure (oracle_enc_params.ctc_gop = 1): one IDR with the parameter sets, then TRAIL_N P pictures / TRAIL_R intra pictures with the
        # GOP table's reference picture sets, POC running on (5 lsb bits: it wraps once in 64 pictures; HM's default of 8 would not); the occupancy stream
        # (IntraPeriod 1, DecodingRefreshType 0) is one IDR and TRAIL_R intra pictures. A piece only depends on its position (first_idx): same bytes as one serial run.
        kw = dict(ctc_gop=1, log2_max_poc_lsb=5, first_idx=first, want_recon=False)
        if kind == "geo":
            return O.encode_ctc(frames, w, h, 10, 16, i_qp_offset=-3, p_qp_offset=-3, md5_sei=1, **kw)[0]
        if kind == "attr":
            return O.encode_ctc(frames, w, h, 10, 22, i_qp_offset=-3, p_qp_offset=0, md5_sei=1, **kw)[0]
        return O.encode_ctc(frames, w, h, 8, 8, gop=1, i_qp_offset=0, lossless=1, md5_sei=0, **kw)[0]
    if kind == "geo":
        return O.encode_hm(frames, w, h, 10, 16, gop=2, i_qp_offset=-3, p_qp_offset=-3, md5_sei=1, want_recon=False)[0]
    if kind == "attr":
        return O.encode_hm(frames, w, h, 10, 22, gop=2, i_qp_offset=-3, p_qp_offset=0, md5_sei=1, want_recon=False)[0]
    return O.encode_hm(frames, w, h, 8, 8, gop=1, i_qp_offset=0, lossless=1, md5_sei=0, want_recon=False)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=1280)
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--seed", type=int, default=1051)
    ap.add_argument("--ctc", type=int, default=0, help="1: the stream structure of the CTC's HM encoder (cfg/hm/ctc-hm-geometry-ai.cfg:21-30) instead of closed (IDR, P) pairs; files hm_r5ctc_*")
    a = ap.parse_args()
    import oracle_lib as O
    import synth
    O.build()
    w, h, n = a.width, a.height, a.frames
    geo, attr, occ = synth.make_gof_maps(w, h, n, a.seed)
    tasks = []
    for i in range(n):
        tasks += [("attr", attr[2 * i:2 * i + 2], w, h, a.ctc, 2 * i), ("geo", geo[2 * i:2 * i + 2], w, h, a.ctc, 2 * i), ("occ", occ[i:i + 1], w // 2, h // 2, a.ctc, i)]
    with ProcessPoolExecutor(a.jobs) as ex:
        res = list(ex.map(_encode, tasks))
    streams = {"attr": b"".join(res[0::3]), "geo": b"".join(res[1::3]), "occ": b"".join(res[2::3])}
    man = {"generator": "tests/golden/make_hm_gof.py" + (" --ctc 1" if a.ctc else ""), "frames": n, "width": w, "height": h, "seed": a.seed, "streams": {}}
    for k, v in streams.items():
        name = f"hm_r5{'ctc' if a.ctc else ''}_{w}x{h}_f{n}_{k}.annexb"
        open(os.path.join(HERE, name), "wb").write(v)
        man["streams"][k] = {"file": name, "bytes": len(v), "md5": hashlib.md5(v).hexdigest()}
        print(name, len(v), man["streams"][k]["md5"])
    mpath = os.path.join(HERE, "hm_r5_manifest.json")
    allm = json.load(open(mpath)) if os.path.exists(mpath) else {}
    allm[f"{w}x{h}_f{n}" + ("_ctc" if a.ctc else "")] = man
    json.dump(allm, open(mpath, "w"), indent=1)


if __name__ == "__main__":
    main()
