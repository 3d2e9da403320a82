"""Streams in the structure of the CTC's HM encoder (cfg/hm/ctc-hm-geometry-ai.cfg:21-30, ctc-hm-occupancy-map-ai-main10.cfg:22-29), made by the oracle's
encoder (oracle_enc_params.ctc_gop): ONE IDR_W_RADL with the parameter sets, then trailing pictures with POC running on - intra pictures as TRAIL_R with
slice_type I and the GOP table's reference picture set {-2}, P pictures referencing POC - 1 (TRAIL_N where nothing references them, as HM marks them;
variant 2: TRAIL_R throughout), pic_order_cnt_lsb narrow enough to wrap. Shared by the host-build tests (test_ctc_structure.py) and the GPU tests
(test_gpu_ctc.py), which run the same checks on the product's two builds."""
import numpy as np
import oracle_lib as O
import synth

STRESS_SEEDS = list(range(1, 25))


def stress_case(seed):
    """random-syntax stream in the CTC structure: I P I P or I P P groups, one or two references, reference picture sets from the SPS or coded in the slice
    header (plain and predicted from another set), 4..6 POC lsb bits over 23 pictures -> (stream, reconstruction, w, h, bit depth, pictures)"""
    w = [64, 96, 128, 80][seed % 4]; h = [64, 80, 48, 128][(seed // 4) % 4]
    bd = 10 if seed % 3 else 8
    n = 23
    fr = np.zeros((n, w * h * 3 // 2), np.uint16)
    bs, rec = O.encode_ctc(fr, w, h, bd, 30, ctc_gop=1 + seed % 2, log2_max_poc_lsb=4 + seed % 3, hm=0, stress_seed=seed)
    return bs, rec, w, h, bd, n


def hm_gof(w, h, n_pc, seed, ctc_gop=1, lsb=4):
    """[occupancy, geometry, attribute] sub-bitstreams of one GOF of n_pc point-cloud frames, HM-like toolset, CTC structure, with their reconstructions"""
    geo, attr, occ = synth.make_gof(w, h, n_pc, seed)
    so, ro = O.encode_ctc(occ, w // 2, h // 2, 8, 8, ctc_gop=ctc_gop, log2_max_poc_lsb=lsb, gop=1, i_qp_offset=0, lossless=1, md5_sei=0)
    sg, rg = O.encode_ctc(geo, w, h, 10, 16, ctc_gop=ctc_gop, log2_max_poc_lsb=lsb, p_qp_offset=-3)
    sa, ra = O.encode_ctc(attr, w, h, 10, 22, ctc_gop=ctc_gop, log2_max_poc_lsb=lsb)
    return [so, sg, sa], [ro, rg, ra]


def check_decode_stress(ctx, seed):
    bs, rec, w, h, bd, n = stress_case(seed)
    dec, dw, dh, dbd, chk, fail = ctx.decode(bs)
    assert (dw, dh, dbd, chk, fail) == (w, h, bd, n, 0) and np.array_equal(dec, rec)
    ref, *_ = O.decode(bs)
    assert np.array_equal(ref, rec)


def check_gof(ctx, R, gs, w, h, n_pc, seed, ctc_gop, rate=3):
    """decode == the encoder's reconstruction (hash SEI checked on every picture), transcodeData == oracle, and the structure really is the CTC's"""
    streams, recs = hm_gof(w, h, n_pc, seed, ctc_gop)
    for s, r, pics in zip(streams, recs, (n_pc, 2 * n_pc, 2 * n_pc)):
        dec, dw, dh, bd, chk, fail = ctx.decode(s, verify_md5=True)
        assert dec.shape[0] == pics and fail == 0 and np.array_equal(dec, r)
        hd = [x for x in O.slice_headers(s) if x["address"] == 0]
        assert [x["nal_type"] for x in hd].count(19) == 1 and [x["poc"] for x in hd] == list(range(pics)) and not gs.is_closed_pairs(s)
    assert chk == 2 * n_pc                                        # geometry / attribute carry the hash SEI
    gq, aq, prec = gs.RATE_POINTS[rate]
    out = ctx.transcode_gof(streams, gs.rate_params(R, rate))
    assert out == O.transcode_data(streams, [(0, 8, prec, 5, gs.DEFAULT_ROWS, 0), (1, gq, prec, 5, gs.DEFAULT_ROWS, 0), (19, aq, prec, 5, gs.DEFAULT_ROWS, 0)])
    return streams, out


WP_SEEDS = list(range(1, 21))


def wp_case(seed):
    """random-syntax stream with weighted prediction (PPS weighted_pred_flag, a pred_weight_table of random weights and offsets per P slice - what libx265 writes from its
    preset "veryfast" up, i.e. the reference's own output handed back as an input), on noise so that the weights matter; odd seeds in the CTC stream structure"""
    w = [64, 96, 128, 80][seed % 4]; h = [64, 80, 48, 128][(seed // 4) % 4]
    bd = 10 if seed % 3 else 8
    fr = np.random.default_rng(900 + seed).integers(0, 1 << bd, (7, w * h * 3 // 2)).astype(np.uint16)
    ctc = seed % 2
    bs, rec = O.encode_ex(fr, True, width=w, height=h, bit_depth=bd, qp=28, gop=2, stress_seed=seed, weighted_pred=1, ctc_gop=ctc, log2_max_poc_lsb=4 if ctc else 0, md5_sei=1)
    return bs, rec, w, h, bd, 7


def check_decode_wp(ctx, seed):
    bs, rec, w, h, bd, n = wp_case(seed)
    dec, dw, dh, dbd, chk, fail = ctx.decode(bs)
    assert (dw, dh, dbd, chk, fail) == (w, h, bd, n, 0) and np.array_equal(dec, rec)
    plain = O.encode_ex(np.zeros((7, w * h * 3 // 2), np.uint16), False, width=w, height=h, bit_depth=bd, qp=28, gop=2, stress_seed=seed, weighted_pred=0, md5_sei=1)[0]
    assert any(x["wp"] for x in O.slice_headers(bs)) and not any(x["wp"] for x in O.slice_headers(plain))
