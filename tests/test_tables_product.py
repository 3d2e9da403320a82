"""Pins the PRODUCT's normative tables (rabbit-transcoding_amd/csrc/rbt_tables.h, read through the test-only host build of the
kernel headers, tests/hostemu) against the oracle's and against the reference ROM dump (tests/golden/hevc_rom_tables.json,
produced by oracle/_ref/hevc_hls_ref from /root/reference/dependencies/PccLibHevcParser, see tests/golden/make_golden.py)."""
import ctypes as C
import json
import os
import subprocess
import pytest
import oracle_lib as O
import rbt_lib

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hevc_rom_tables.json")))


@pytest.fixture(scope="module")
def T():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(__file__), "hostemu")])
    L = C.CDLL(rbt_lib.HOSTEMU_LIB)
    L.rbt_hostemu_table.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    return lambda name, i=0, j=0, k=0: L.rbt_hostemu_table(name.encode(), i, j, k)


def test_transform_and_quant_tables_vs_reference_rom(T):
    for n in (4, 8, 16, 32):                      # T_N[k][x] = T32[k * 32/N][x]
        for k in range(n):
            for x in range(n):
                assert T("dct32", k * (32 // n), x) == G[f"T{n}"][k * n + x] == O.lib().oracle_dct_coef(n, k, x)
    for k in range(4):
        for x in range(4):
            assert T("dst4", k, x) == G["DST4"][k * 4 + x]
    for i in range(6):
        assert T("quant_scale", i) == G["quantScales"][i] and T("dequant_scale", i) == G["invQuantScales"][i]
    for q in range(58):
        assert T("chroma_qp", q) == G["chromaScale420"][q]
    for i in range(16):
        assert T("sig_ctx_4x4", i) == G["ctxIndMap4x4"][i]


def test_scans_vs_reference_rom(T):
    def raster(scan_idx, log2, i):               # product layout: k_scan[scan_idx][log2 of the side in units][pos] = x | y << 4
        sb, p = i >> 4, i & 15
        s = T("scan", scan_idx, log2 - 2, sb); q = T("scan", scan_idx, 2, p)
        return ((((s >> 4) << 2) + (q >> 4)) << log2) + ((s & 15) << 2) + (q & 15)
    for l in (2, 3, 4, 5):
        assert [raster(0, l, i) for i in range(1 << (2 * l))] == G[f"scan_diag_{l}"]
    for l in (2, 3):
        assert [raster(1, l, i) for i in range(1 << (2 * l))] == G[f"scan_hor_{l}"]
        assert [raster(2, l, i) for i in range(1 << (2 * l))] == G[f"scan_ver_{l}"]


def test_cabac_init_values_vs_oracle(T):
    """tests/test_oracle_tables.py pins the oracle's init values against the reference's ContextTables; same layout here."""
    L = O.lib()
    assert T("ctx_count") == L.oracle_ctx_count() == 157
    for t in range(3):
        for i in range(157):
            assert T("ctx_init", t, i) == L.oracle_ctx_init(t, i), (t, i)


def test_tables_restated_from_the_standard_match_the_oracle(T):
    """rangeTabLPS, state transitions, intra angles, interpolation filters, deblocking tables: absent from the reference, so
    this only shows that product and oracle carry the SAME restatement (parity unpinned for these)."""
    L = O.lib(); L.oracle_table.argtypes = [C.c_char_p, C.c_int, C.c_int]
    o = lambda name, i=0, j=0: L.oracle_table(name.encode(), i, j)
    for s in range(64):
        assert T("next_lps", s) == o("next_lps", s)
        for q in range(4):
            assert T("range_lps", s, q) == o("range_lps", s, q)
    for m in range(35): assert T("intra_angle", m) == o("intra_angle", m)
    for m in range(15): assert T("intra_inv_angle", m) == o("intra_inv_angle", m)
    for f in range(4):
        for t in range(8): assert T("luma_filter", f, t) == o("luma_filter", f, t)
    for f in range(8):
        for t in range(4): assert T("chroma_filter", f, t) == o("chroma_filter", f, t)
    for q in range(52): assert T("beta", q) == o("beta", q)
    for q in range(54): assert T("tc", q) == o("tc", q)
    # invariants of the standard's tables that do not need a second copy: filters sum to 64, angles are antisymmetric around 18 / 26 ... 10
    for f in range(4): assert sum(T("luma_filter", f, t) for t in range(8)) == 64
    for f in range(8): assert sum(T("chroma_filter", f, t) for t in range(4)) == 64
    for m in range(2, 18): assert T("intra_angle", m) == T("intra_angle", 36 - m)
    for m in range(11, 18): assert round(8192 / T("intra_angle", m)) == T("intra_inv_angle", m - 11)
