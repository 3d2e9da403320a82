#!/usr/bin/env python3
"""Generates the committed golden fixtures from the REFERENCE run in this container (never on the GPU box).

  hevc_rom_tables.json   output of oracle/_ref/hevc_hls_ref tables  (reference ROM: PccHevcTComRom.cpp, PccHevcContextTables.h)
  hls_<name>.annexb      parameter sets emitted by the oracle encoder for the product configurations
  hls_<name>.json        the same NAL units parsed by the reference's TDecCavlc (oracle/_ref/hevc_hls_ref hls)

Run:  oracle/ref_build.sh && python3 tests/golden/make_golden.py
"""
import json, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

REF = os.path.join(ROOT, "oracle", "_ref", "hevc_hls_ref")


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/hevc_hls_ref missing: run oracle/ref_build.sh (needs /root/reference)")
    tables = subprocess.check_output([REF, "tables"]).decode()
    json.loads(tables)
    open(os.path.join(HERE, "hevc_rom_tables.json"), "w").write(tables)
    cfgs = {"geo10_gop2": dict(w=64, h=64, bd=10, qp=24, gop=2, lossless=0, log2_ctb=5, rows=1),
            "occ8_lossless": dict(w=64, h=32, bd=8, qp=8, gop=1, lossless=1, log2_ctb=5, rows=1),
            "attr10_ctb64_oneslice": dict(w=128, h=64, bd=10, qp=22, gop=2, lossless=0, log2_ctb=6, rows=0),
            "occ8_window_40x44": dict(w=40, h=44, bd=8, qp=8, gop=1, lossless=1, log2_ctb=5, rows=1),   # coded 40x48, conformance window
            "geo10_wave": dict(w=64, h=64, bd=10, qp=24, gop=2, lossless=0, log2_ctb=5, rows=-1)}      # wavefront mode: entropy_coding_sync + dependent slice segments
    for name, c in cfgs.items():
        fr = np.full((2, c["w"] * c["h"] * 3 // 2), 100, np.uint16)
        bs, _ = O.encode(fr, c["w"], c["h"], c["bd"], c["qp"], gop=c["gop"], lossless=c["lossless"], log2_ctb=c["log2_ctb"], rows_per_slice=c["rows"])
        p = os.path.join(HERE, f"hls_{name}.annexb")
        open(p, "wb").write(bs)
        out = subprocess.check_output([REF, "hls", p]).decode()
        out = "".join(l + "\n" for l in out.splitlines() if l[:1] in "[],{")   # the reference parser also printf()s progress lines
        json.loads(out)
        open(os.path.join(HERE, f"hls_{name}.json"), "w").write(out)
        print(name, len(bs), "bytes")
    slice_goldens()


CTC_SEEDS = (2, 5, 6, 9, 11, 16)   # random-syntax streams in the CTC structure: 2, 6: I P P groups; 5, 9, 11: every trailing picture TRAIL_R; explicit / predicted sets
WP_SEEDS = (4, 7, 9, 10, 15, 22)   # random-syntax streams with weighted prediction (odd: in the CTC structure)
WAVE_SEEDS = (13, 17)    # random-syntax streams with entropy_coding_sync (13: with dependent slice segments as well)


def slice_goldens():
    """slices_<name>.annexb / .json: every slice segment header of oracle-coded streams through the reference's parseSliceHeader (hevc_hls_ref slices).
    The SPS fields that syntax depends on (the reference's parseSPS does not store them) are passed on the command line from the oracle's own settings."""
    import synth
    m = synth.make_maps(128, 128, 3)
    zeros = np.zeros((5, 96 * 64 * 3 // 2), np.uint16)
    cases = {"e1_geo_rows": (O.encode(m["geo"], 128, 128, 10, 24, gop=2, log2_ctb=5, rows_per_slice=1)[0], [8, 5, 1, 0, 1]),
             "e1_occ_lossless": (O.encode(m["occ"], 64, 64, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=5, rows_per_slice=1)[0], [8, 5, 0, 0, 1]),
             "hm_attr": (O.encode_hm(m["attr"], 128, 128, 10, 22)[0], [8, 6, 1, 1, 1]),
             "e1_geo_wave": (O.encode(m["geo"], 128, 128, 10, 24, gop=2, log2_ctb=5, rows_per_slice=-1)[0], [8, 5, 1, 0, 1])}   # wavefront mode: dependent slice segments, entry point syntax
    for seed in WAVE_SEEDS + (3, 7, 12, 21):       # random-syntax streams: several slices per picture, two references, TMVP, cabac_init, chroma offsets, deblocking overrides
        bs = O.encode(zeros, 96, 64, 10, qp=30, gop=2, stress_seed=seed, log2_ctb=0)[0]
        sps = O.sps_fields(bs)
        cases[f"stress{seed}"] = (bs, [sps["log2_max_poc_lsb"], sps["log2_ctb"], sps["sao"], sps["tmvp"], sps["num_st_rps"]])
    # The structure of the CTC's HM encoder (cfg/hm/ctc-hm-geometry-ai.cfg:21-30: one IDR, then TRAIL_N P pictures and TRAIL_R intra pictures with the GOP
    # table's reference picture sets, POC running on and wrapping its lsb, parameter sets only at the IDR): the HM-like mode in both NAL type variants, and
    # random-syntax streams (I P I P / I P P, two references, sets coded in the slice header with and without inter-set prediction). Last argument 1: the
    # harness builds the SPS's sets {-1}, {-2}, {-1,-2} (the reference's parseSPS does not keep them).
    geo20 = np.concatenate([m["geo"]] * 10)
    attr20 = np.concatenate([m["attr"]] * 10)
    cases["ctc_hm_geo"] = (O.encode_ctc(geo20, 128, 128, 10, 16, ctc_gop=1, log2_max_poc_lsb=4, p_qp_offset=-3)[0], [4, 6, 1, 1, 2, 1])
    cases["ctc_hm_attr_trail_r"] = (O.encode_ctc(attr20, 128, 128, 10, 22, ctc_gop=2, log2_max_poc_lsb=5)[0], [5, 6, 1, 1, 2, 1])
    zeros23 = np.zeros((23, 96 * 64 * 3 // 2), np.uint16)
    for seed in CTC_SEEDS:
        bs = O.encode_ctc(zeros23, 96, 64, 10, 30, ctc_gop=1 + seed % 2, log2_max_poc_lsb=4 + seed % 3, hm=0, stress_seed=seed)[0]
        sps = O.sps_fields(bs)
        cases[f"ctc_stress{seed}"] = (bs, [sps["log2_max_poc_lsb"], sps["log2_ctb"], sps["sao"], sps["tmvp"], sps["num_st_rps"], 1])
    # weighted prediction (PPS weighted_pred_flag, pred_weight_table in every P slice: what libx265 writes from preset "veryfast" up): random-syntax streams with random
    # tables - luma / chroma flags per reference, weights around 2^denominator, offsets, the chroma offset prediction of 7.4.7.3 - in closed groups and in the CTC structure
    noise = np.random.default_rng(77).integers(0, 1024, (9, 96 * 64 * 3 // 2)).astype(np.uint16)
    for seed in WP_SEEDS:
        ctc = seed % 2
        bs = O.encode_ex(noise, False, width=96, height=64, bit_depth=10, qp=30, gop=2, stress_seed=seed, weighted_pred=1, ctc_gop=ctc, log2_max_poc_lsb=4 if ctc else 0, md5_sei=1)[0]
        sps = O.sps_fields(bs)
        cases[f"wp_stress{seed}"] = (bs, [sps["log2_max_poc_lsb"], sps["log2_ctb"], sps["sao"], sps["tmvp"], sps["num_st_rps"], ctc])
    for name, (bs, a) in cases.items():
        p = os.path.join(HERE, f"slices_{name}.annexb")
        open(p, "wb").write(bs)
        out = subprocess.check_output([REF, "slices", p] + [str(x) for x in a]).decode()
        out = "".join(l + "\n" for l in out.splitlines() if l[:1] in "[],{")
        json.loads(out)
        open(os.path.join(HERE, f"slices_{name}.json"), "w").write(out)
        print("slices", name, len(bs), "bytes", len(json.loads(out)), "slice headers")


if __name__ == "__main__":
    main()
