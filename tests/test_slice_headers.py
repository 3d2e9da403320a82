"""Slice segment headers pinned against the REFERENCE's own parser: tests/golden/slices_*.json holds what TDecCavlc::parseSliceHeader
(dependencies/PccLibHevcParser/source/PccHevcTDecCAVLC.cpp:1138, compiled in place into oracle/_ref by oracle/ref_build.sh) reads from
streams the oracle encoder wrote - RBT-E1 (one slice per CTB row, SAO flags), lossless occupancy, the HM-like mode (TMVP, five merge candidates) and
random-syntax streams (several slices per picture, two references, cabac_init, chroma QP offsets, deblocking overrides; every intra picture of these
is an IDR picture), wavefront streams (entropy_coding_sync: entry point syntax; dependent slice segments, whose fields are those of their slice's
first segment), and - round 4, `ctc_*` - streams in the structure of the CTC's HM encoder (cfg/hm/ctc-hm-geometry-ai.cfg:21-30): ONE IDR, then
trailing pictures with POC running on and its lsb wrapping (4..6 bits), intra pictures as TRAIL_R with slice_type I and a reference picture set,
non-referenced P pictures as TRAIL_N (which do not move the POC anchor, 8.3.1), sets taken from the SPS by index or coded in the slice header
with and without inter-set prediction; for these the reference picture set itself (`rps`: delta POC, used flag) is compared as well.
Checked here without the reference: the oracle's slice header parser and the PRODUCT's host-side parser (host/rbt_hls.cpp, through the test build)
read the same streams to the same values, field by field. What this pins is the header syntax; the reference's parseSPS does not store the SPS
fields that syntax depends on (PccHevcTDecCAVLC.cpp:732), so the generator hands them over from the oracle's own settings."""
import ctypes as C
import glob
import json
import os
import subprocess
import pytest
import oracle_lib as O
import rbt_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[7:-5] for p in glob.glob(os.path.join(GOLD, "slices_*.json")))


@pytest.fixture(scope="module")
def product_parser():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    L = C.CDLL(rbt_lib.HOSTEMU_LIB)
    L.rbt_hostemu_slice_headers.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int]
    return L.rbt_hostemu_slice_headers


def test_there_are_goldens():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_slice_headers_match_reference_parser(product_parser, name):
    bs = open(os.path.join(GOLD, f"slices_{name}.annexb"), "rb").read()
    ref = json.load(open(os.path.join(GOLD, f"slices_{name}.json")))
    O.lib().oracle_slice_headers.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int]
    ours = O.slice_headers(bs)
    prod = O.slice_headers(bs, product_parser)
    assert len(ref) == len(ours) == len(prod) and len(ref) > 0
    for k, (r, o, p) in enumerate(zip(ref, ours, prod)):
        for f in O.SLICE_FIELDS:
            if r["slice_type"] == 2 and f in ("tmvp", "cabac_init"): continue       # not coded in I slices; parsers keep different defaults
            if r["deblocking_disabled"] and f in ("beta_offset_div2", "tc_offset_div2"): continue   # not coded (and not used) when the slice switches the filter off
            assert r[f] == o[f] == p[f], (name, k, f, r[f], o[f], p[f])
        if "rps" in r: assert r["rps"] == o["rps"] == p["rps"], (name, k, r["rps"], o["rps"], p["rps"])
        if "wp" in r: assert r["wp"] == o["wp"] == p["wp"], (name, k, r["wp"], o["wp"], p["wp"])


def test_ctc_goldens_hold_the_structure_of_the_ctc_encoder():
    """what the ctc_* goldens contain, by the REFERENCE parser's reading: one IDR, intra slices in TRAIL_R pictures carrying a reference picture set,
    TRAIL_N P pictures, POC beyond the lsb range (wrapped at least once), two-entry sets"""
    seen = set()
    for name in CASES:
        if not name.startswith("ctc_"): continue
        ref = json.load(open(os.path.join(GOLD, f"slices_{name}.json")))
        first = [r for r in ref if r["address"] == 0 and not r["dependent"]]
        assert [r["nal_type"] for r in first].count(19) == 1 and first[0]["nal_type"] == 19
        assert [r["poc"] for r in first] == list(range(len(first))) and len(first) >= 20       # POC runs on: lsb of 4..6 bits wrapped
        for r in ref:
            if r["nal_type"] == 1 and r["slice_type"] == 2 and r["rps"]: seen.add("intra TRAIL_R with a set")
            if r["nal_type"] == 0 and r["slice_type"] == 1: seen.add("TRAIL_N P")
            if len(r["rps"]) == 2: seen.add("two entries")
            if r["rps"] == [[-2, 1]]: seen.add("the GOP table's {-2}")
    assert seen == {"intra TRAIL_R with a set", "TRAIL_N P", "two entries", "the GOP table's {-2}"}, seen


def test_wp_goldens_hold_weight_tables():
    """the wp_* goldens, by the REFERENCE parser's reading (xParsePredWeightTable): P slices with tables whose flags, weights and offsets vary - luma and chroma flags both
    set and unset, weights off 2^denominator, non-zero luma offsets, chroma offsets moved by the prediction of 7.4.7.3, one and two references"""
    seen = set()
    for name in CASES:
        if not name.startswith("wp_"): continue
        for r in json.load(open(os.path.join(GOLD, f"slices_{name}.json"))):
            if r["slice_type"] != 1: assert r["wp"] == []; continue
            ld, cd = r["wp"][0], r["wp"][1]
            assert len(r["wp"]) == 2 + r["num_ref_idx"]
            if r["num_ref_idx"] == 2: seen.add("two references")
            for lf, cf, wy, oy, wcb, ocb, wcr, ocr in r["wp"][2:]:
                seen.add("luma on" if lf else "luma off"); seen.add("chroma on" if cf else "chroma off")
                if not lf: assert (wy, oy) == (1 << ld, 0)
                if not cf: assert (wcb, ocb, wcr, ocr) == (1 << cd, 0, 1 << cd, 0)
                if lf and wy != 1 << ld: seen.add("luma weight")
                if lf and oy: seen.add("luma offset")
                if cf and (ocb or ocr): seen.add("chroma offset")
    assert seen == {"two references", "luma on", "luma off", "chroma on", "chroma off", "luma weight", "luma offset", "chroma offset"}, seen
