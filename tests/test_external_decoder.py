"""Row N1 of the round-3 review: everything green in this repository is green against the builder's own reading of H.265 (product == oracle). The only route to an
independent pin is a third-party decoder - ffmpeg / libavcodec, HM's TAppDecoder, libde265 - and none exists in the build container or on the GPU boxes. These tests
look for one (tests/external_tools.py) and SKIP with that reason when there is none; where one exists they decode what RBT-E1 writes in every mode, plus both
committed fixtures, and compare the pictures with the oracle's decoder sample by sample (the -m gpu half does the same for rbt_decode and compares D1 / D2 with the
reference's libx265 path)."""
import json
import os
import numpy as np
import pytest
import external_tools as X
import oracle_lib as O
import rbt_lib
import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
need_decoder = pytest.mark.skipif(X.find_decoder() is None, reason="no third-party HEVC decoder on PATH (ffmpeg, TAppDecoder(Static), dec265): parity with one stays unpinned on this box")
need_x265 = pytest.mark.skipif(not X.have_libx265(), reason="no ffmpeg with libx265 on PATH: the +-0.05 dB D1 / D2 comparison with the reference's encoder cannot be made on this box")


def test_probe_reports_what_it_found():
    d = X.describe()
    assert d["external_decoder"] in ("absent", "ffmpeg", "TAppDecoderStatic", "TAppDecoder", "dec265") and d["libx265"] in ("absent", "present")


def _e1_streams():
    """(name, stream, display w, display h, bit depth) of RBT-E1 output in every slice structure, lossless occupancy, a conformance window, occupancy-aware coding"""
    m = synth.make_maps(256, 192, 31)
    out = []
    for rows in (-1, -2, 1, 0):                  # wavefront rows as dependent slice segments, behind entry points, independent row slices, one slice per picture
        for key, qp in (("geo", 24), ("attr", 32)):
            out.append((f"e1_{key}_rows{rows}", O.encode(m[key], 256, 192, 10, qp, gop=2, log2_ctb=5, rows_per_slice=rows)[0], 256, 192, 10))
    out.append(("e1_occ_lossless", O.encode(m["occ"], 128, 96, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=5, rows_per_slice=-1)[0], 128, 96, 8))
    r = np.random.default_rng(5)
    occ = (r.integers(0, 2, (3, 40 * 44 * 3 // 2)) * 1).astype(np.uint16); occ[:, 40 * 44:] = 128
    out.append(("e1_occ_window_40x44", O.encode(occ, 40, 44, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=5, rows_per_slice=1)[0], 40, 44, 8))
    geo, attr, occ2 = synth.make_gof(128, 128, 2, 21)
    src = [O.encode(occ2, 64, 64, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=6, rows_per_slice=0)[0], O.encode(geo, 128, 128, 10, 16, gop=2, log2_ctb=6, rows_per_slice=0)[0],
           O.encode(attr, 128, 128, 10, 22, gop=2, log2_ctb=6, rows_per_slice=0)[0]]
    f4 = O.transcode_data(src, [(0, 8, 4, 5, -1, 0, 0), (1, 24, 4, 5, -1, 0, 1), (19, 32, 4, 5, -1, 0, 1)])          # occupancy_rd
    out += [("f4_occ", f4[0], 32, 32, 8), ("f4_geo", f4[1], 128, 128, 10), ("f4_attr", f4[2], 128, 128, 10)]
    return out


def test_stream_list_of_the_hook_is_sound():
    """the streams the hook would hand to a third-party decoder, through the oracle's (runs everywhere: keeps the hook's set-up from rotting while no decoder exists)"""
    names = []
    for name, bs, w, h, bd in _e1_streams():
        dec, dw, dh, dbd, chk, fail = O.decode(bs)
        assert (dw, dh, dbd, fail) == (w, h, bd, 0) and dec.shape[1] == w * h * 3 // 2, name
        names.append(name)
    assert len(names) == 13 and "e1_attr_rows-2" in names and "f4_geo" in names


@need_decoder
def test_external_decoder_reads_rbt_e1_output_like_the_oracle():
    for name, bs, w, h, bd in _e1_streams():
        want = O.decode(bs)[0]
        got = X.decode(bs, w, h, bd)
        assert got.shape == want.shape and np.array_equal(got, want), name


@need_decoder
@pytest.mark.parametrize("structure", ["1280x1280_f32", "1280x1280_f32_ctc"])
def test_external_decoder_reads_the_fixtures_like_the_oracle(structure):
    """the HM-like streams (the decoder's input side: 35 modes, NxN, TU trees, AMP, merge / AMVP, TMVP, transform skip, sign hiding, SAO) in both stream structures;
    the first 3 point-cloud frames of each (the oracle's decoder takes about a second per full-size picture)"""
    gs = rbt_lib.module_file("gof_shard")
    man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))[structure]
    for kind, pics, (w, h, bd) in (("occ", 3, (640, 640, 8)), ("geo", 6, (1280, 1280, 10)), ("attr", 6, (1280, 1280, 10))):
        bs = gs.first_pictures(open(os.path.join(GOLD, man["streams"][kind]["file"]), "rb").read(), pics)
        assert np.array_equal(X.decode(bs, w, h, bd), O.decode(bs)[0]), (structure, kind)


@need_decoder
@pytest.mark.gpu
def test_external_decoder_agrees_with_rbt_decode_on_whole_fixtures():
    R = rbt_lib.module()
    ctx = R.Context(device=0)
    try:
        for structure in ("1280x1280_f32", "1280x1280_f32_ctc"):
            man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))[structure]
            for kind, (w, h, bd) in (("occ", (640, 640, 8)), ("geo", (1280, 1280, 10)), ("attr", (1280, 1280, 10))):
                bs = open(os.path.join(GOLD, man["streams"][kind]["file"]), "rb").read()
                assert np.array_equal(X.decode(bs, w, h, bd), ctx.decode(bs)[0]), (structure, kind)
                out = ctx.transcode_substream(bs, {"occ": 0, "geo": 1, "attr": 19}[kind], {"occ": 8, "geo": 24, "attr": 32}[kind], rows_per_slice=-1)
                ow, oh = (w // 2, h // 2) if kind == "occ" else (w, h)
                assert np.array_equal(X.decode(out, ow, oh, bd), ctx.decode(out)[0]), (structure, kind, "transcoded")
    finally:
        ctx.close()


@need_x265
@pytest.mark.gpu
def test_d1_d2_against_the_reference_libx265_path():
    """north star: decoded point clouds must match the reference libx265 path's D1 / D2 within +-0.05 dB at the same QP. The first 4 point-cloud frames of the CTC-structured
    fixture, R5 -> R3, through ffmpeg + libx265 with the reference's options (tests/external_tools.py x265_transcode) and through librbt; D1 / D2 means of the four frames."""
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    ctx = R.Context(device=0)
    try:
        man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
        src = [gs.first_pictures(open(os.path.join(GOLD, man["streams"][k]["file"]), "rb").read(), n) for k, n in (("occ", 4), ("geo", 8), ("attr", 8))]
        ours = ctx.transcode_gof(src, gs.rate_params(R, 3))
        occ_in = ctx.decode(src[0])[0][:, : 640 * 640].reshape(-1, 640, 640)
        pooled = (occ_in.reshape(-1, 320, 2, 320, 2).max(axis=(2, 4)) > 0).astype(np.uint16)      # the reference pools, then codes lossless: the output occupancy is this either way
        theirs_geo = X.decode(X.x265_transcode(src[1], 1280, 1280, 10, 24), 1280, 1280, 10)
        w = h = 1280
        res = {"ours": [], "x265": []}
        for k in range(4):
            sk = synth.make_maps(w, h, 1051 + k); pk = synth.atlas_patches(R, w, h, 1051 + k)
            cs, ns = synth.source_normals(R, ctx.reconstruct, w, h, 1051 + k, sk["occ_full"], sk["geo"])
            for key, geo in (("ours", ctx.decode(ours[1])[0]), ("x265", theirs_geo)):
                c = ctx.reconstruct(R.AtlasParams(w, h, 16, 4, 2, 1, 1, 0), pk, pooled[k], geo[2 * k][: w * h].reshape(h, w), geo[2 * k + 1][: w * h].reshape(h, w), 10)[0]
                res[key].append((ctx.d1(cs, c)["psnr"], ctx.d2(cs, ns, c)["psnr"]))
        d1 = [sum(a for a, _ in res[k]) / 4 for k in ("ours", "x265")]; d2 = [sum(b for _, b in res[k]) / 4 for k in ("ours", "x265")]
        print("D1 ours / x265:", d1, "D2:", d2)
        assert abs(d1[0] - d1[1]) <= 0.05 and abs(d2[0] - d2[1]) <= 0.05, (d1, d2)
    finally:
        ctx.close()
