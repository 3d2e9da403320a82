"""Pins the oracle's normative HEVC tables against the REFERENCE's ROM (tests/golden/hevc_rom_tables.json, produced by
oracle/_ref/hevc_hls_ref from /root/reference/dependencies/PccLibHevcParser — see tests/golden/make_golden.py)."""
import json
import os
import oracle_lib as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hevc_rom_tables.json")))
L = O.lib()


def test_dct_matrices():
    for n in (4, 8, 16, 32):
        ref = G[f"T{n}"]
        for k in range(n):
            for x in range(n):
                assert L.oracle_dct_coef(n, k, x) == ref[k * n + x], (n, k, x)


def test_dst_and_quant():
    for k in range(4):
        for x in range(4):
            assert L.oracle_dst_coef(k, x) == G["DST4"][k * 4 + x]
    for i in range(6):
        assert L.oracle_quant_scale(i) == G["quantScales"][i]
        assert L.oracle_dequant_scale(i) == G["invQuantScales"][i]
    for q in range(58):
        assert L.oracle_chroma_qp(q) == G["chromaScale420"][q]
    for i in range(16):
        assert L.oracle_sig_ctx_4x4(i) == G["ctxIndMap4x4"][i]


def test_scan_orders():
    for l in (2, 3, 4, 5):
        ref = G[f"scan_diag_{l}"]
        assert [L.oracle_scan_raster(0, l, i) for i in range(1 << (2 * l))] == ref
    for l in (2, 3):   # horizontal / vertical scans are only used for 4x4 and 8x8 TBs
        assert [L.oracle_scan_raster(1, l, i) for i in range(1 << (2 * l))] == G[f"scan_hor_{l}"]
        assert [L.oracle_scan_raster(2, l, i) for i in range(1 << (2 * l))] == G[f"scan_ver_{l}"]


def test_cabac_init_values():
    """Reference rows are [B, P, I]; the oracle rows are initType [0 (I), 1 (P default), 2 (B default)]."""
    # (oracle first ctx, count, reference table, offset in reference row)
    layout = [("INIT_SAO_MERGE_FLAG", 0, 1, 0), ("INIT_SAO_TYPE_IDX", 1, 1, 0), ("INIT_SPLIT_FLAG", 2, 3, 0),
              ("INIT_CU_TRANSQUANT_BYPASS_FLAG", 5, 1, 0), ("INIT_SKIP_FLAG", 6, 3, 0), ("INIT_PRED_MODE", 9, 1, 0),
              ("INIT_PART_SIZE", 10, 4, 0), ("INIT_INTRA_PRED_MODE", 14, 1, 0), ("INIT_CHROMA_PRED_MODE", 15, 1, 0),
              ("INIT_QT_ROOT_CBF", 16, 1, 0), ("INIT_MERGE_FLAG_EXT", 17, 1, 0), ("INIT_MERGE_IDX_EXT", 18, 1, 0),
              ("INIT_INTER_DIR", 19, 5, 0), ("INIT_REF_PIC", 24, 2, 0), ("INIT_MVP_IDX", 26, 1, 0),
              ("INIT_TRANS_SUBDIV_FLAG", 27, 3, 0), ("INIT_QT_CBF", 30, 2, 0), ("INIT_QT_CBF", 32, 5, 5), ("INIT_MVD", 37, 2, 0),
              ("INIT_DQP", 39, 2, 0), ("INIT_TRANSFORMSKIP_FLAG", 41, 2, 0), ("INIT_LAST", 43, 15, 0), ("INIT_LAST", 58, 3, 15),
              ("INIT_LAST", 61, 15, 0), ("INIT_LAST", 76, 3, 15), ("INIT_SIG_CG_FLAG", 79, 4, 0), ("INIT_SIG_FLAG", 83, 27, 0),
              ("INIT_SIG_FLAG", 110, 15, 28), ("INIT_SIG_FLAG", 125, 1, 27), ("INIT_SIG_FLAG", 126, 1, 43),
              ("INIT_ONE_FLAG", 127, 24, 0), ("INIT_ABS_FLAG", 151, 6, 0)]
    assert L.oracle_ctx_count() == 157
    covered = set()
    for name, first, cnt, off in layout:
        arr = G[name]
        width = len(arr) // 3
        for init_type, ref_row in ((0, 2), (1, 1), (2, 0)):
            for i in range(cnt):
                ref = arr[ref_row * width + off + i]
                got = L.oracle_ctx_init(init_type, first + i)
                # contexts the reference marks CNU (154) are unused for that slice type
                assert got == ref, (name, init_type, i, got, ref)
        covered.update(range(first, first + cnt))
    assert covered == set(range(157))
