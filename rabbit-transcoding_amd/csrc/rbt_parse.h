// Slice-segment entropy decoding (H.265 7.3.8 / 9.3) as a WAVE-UNIFORM routine: one 64-lane wave per slice segment.
// CABAC is a serial dependency chain, so every lane executes the same scalar sequence; what the 64 lanes buy is the
// cooperative fill of the per-4x4 maps (RBT_PAR_FOR) and free redundancy instead of divergence. Parallelism comes
// from the ~160 independent slice segments of a GOF (64 geometry + 64 attribute + 32 occupancy pictures).
//
// Replaces the entropy-decoding half of libavcodec's hevc decoder as driven by PCCTranscoder.cpp:428-448.
// Output (HBM): per-CTB command lists (PU / TU records), coefficient levels in the dense coef planes, the per-4x4
// maps (pm, edges, dm, qp, mv, ref, refpoc) and per-CTB SAO parameters. Reconstruction runs in later kernels.
#pragma once
#include "rbt_cabac.h"
#include "rbt_types.h"

#define RBT_NO_REFPOC ((int32_t)0x80000000)
#ifdef RBT_PROFILE
#define PZ_STAMP(s, id) do { unsigned long long n_ = __builtin_readcyclecounter(); if (RBT_LANE0) { (s)->L->prof[id] += n_ - (s)->t_last; (s)->L->profn[id]++; } (s)->t_last = __builtin_readcyclecounter(); } while (0)
#else
#define PZ_STAMP(s, label) do { } while (0)
#endif
#if defined(RBT_TRACE) && !defined(RBT_HOSTEMU)
#define RBT_TR(...) do { if (RBT_LANE0) printf(__VA_ARGS__); } while (0)
#else
#define RBT_TR(...) do { } while (0)
#endif

// LDS of one slice parser = this header + the line buffers of the CTB row above behind it (pz_above_*). The line buffers are
// sized by the kernel variant (picture width, RBT_PARSE_CAP4_*): a parser is one lone wave, and its LDS footprint decides how
// many parsers and how many reconstruction workgroups of other pictures fit on a CU next to each other.
struct alignas(16) RbtParseLds {
  uint8_t cur_pm[256], cur_dm[256], cur_edges[256]; int8_t cur_qp[256], cur_ref[256]; int32_t cur_mv[256];   // current CTB, row stride 16 units
  uint8_t left_pm[16], left_dm[16]; int8_t left_ref[16]; int32_t left_mv[16];                                 // right column of the left CTB
  RbtSao sao_left;
  uint8_t scan[3][4][64];                                                                                      // k_scan copied once per slice
#ifdef RBT_PROFILE
  unsigned long long prof[32]; unsigned int profn[32];
#endif
  int32_t ref_poc[RBT_MAX_REFS], ref_frame[RBT_MAX_REFS];   // RefPicList0 of the slice (indexed at run time: kept out of the register-resident parser state)
  uint8_t wpp_ctx[4 * 64];                                   // context variables after the second CTB of the CTB row above (9.3.2.2 storage process, TableStateIdxWpp)
  int32_t cap4, pad_[3];                                     // capacity of the line buffers in 4-sample units (multiple of 8)
};
// bottom row of the CTB row above: mv int32[cap4], pm / dm / ref bytes[cap4]; per CTB column (>= 16 samples wide, cap4 / 4 of
// them): the slice that decoded the CTB above (0xFFFF none) and its SAO parameters
enum { RBT_PARSE_CAP4_S = 384, RBT_PARSE_CAP4_M = 1024, RBT_PARSE_CAP4_L = 2048 };   // pictures up to 1536 / 4096 / 8192 samples wide
#define RBT_PARSE_LDS_BYTES(cap4) (sizeof(RbtParseLds) + (size_t)(cap4) * 7 + (size_t)((cap4) / 4) * (2 + sizeof(RbtSao)))
RBT_DEV RBT_LDS_AS int32_t* pz_above_mv(RBT_LDS_AS RbtParseLds* L) { return (RBT_LDS_AS int32_t*)(L + 1); }
RBT_DEV RBT_LDS_AS uint8_t* pz_above_pm(RBT_LDS_AS RbtParseLds* L, int cap4) { return (RBT_LDS_AS uint8_t*)(L + 1) + 4 * cap4; }
RBT_DEV RBT_LDS_AS uint8_t* pz_above_dm(RBT_LDS_AS RbtParseLds* L, int cap4) { return (RBT_LDS_AS uint8_t*)(L + 1) + 5 * cap4; }
RBT_DEV RBT_LDS_AS int8_t* pz_above_ref(RBT_LDS_AS RbtParseLds* L, int cap4) { return (RBT_LDS_AS int8_t*)(L + 1) + 6 * cap4; }
RBT_DEV RBT_LDS_AS uint16_t* pz_above_slice(RBT_LDS_AS RbtParseLds* L, int cap4) { return (RBT_LDS_AS uint16_t*)((RBT_LDS_AS uint8_t*)(L + 1) + 7 * cap4); }
RBT_DEV RBT_LDS_AS RbtSao* pz_sao_above(RBT_LDS_AS RbtParseLds* L, int cap4) { return (RBT_LDS_AS RbtSao*)((RBT_LDS_AS uint8_t*)(L + 1) + 7 * cap4 + (cap4 / 4) * 2); }
struct RbtParse {
  RBT_LDS_AS RbtParseLds* L; int ctb_x, ctb_y, left_ok, corner_ok, corner_pm, corner_dm, corner_ref, corner_mv;
  RbtFrame* f; const RbtFrame* frames; int slice_idx;
  // picture / slice parameters packed into a few wave-uniform words (read with pzc_* / pzs_*: one s_bfe each)
  uint32_t c_dim, c_logs, c_flags, s_bits, s_qp; int s_poc;
  // wave-uniform copies of the picture's map pointers (SGPR pairs instead of a pointer load per access)
  RbtCmd* m_cmds;                       // the per-CTB map pointers are fetched from the RbtFrame where they are used (once per CTB)
  int16_t *m_coef0, *m_coef1, *m_coef2; int m_cap;
  RbtCabacDec c;
  int qp_y, qp_pred, qp_y_prev, is_cu_qp_delta_coded, cu_qp_delta_val;
  int ctb_addr; uint32_t n_cmds;
  int cu_x, cu_y, cu_log2, cu_pred_mode, cu_part_mode, cu_tq_bypass;
  int qp_key, qp_packed;             // qP of Y | Cb << 8 | Cr << 16 (incl. QpBdOffset) cached for QpY == qp_key (chroma mapping costs ~300 cycles)
  int il_packed, intra_chroma, max_trafo_depth, last_pu_merge;   // no arrays / index-selected fields here: they would pin the whole struct in scratch
  int error;
  // Register-resident neighbour context (one value per lane). Current CTB, four horizontally adjacent 4x4 units per lane:
  // unit (ux,uy) lives in lane uy * 4 + (ux >> 2), byte ux & 3 (mv: register r_mv<ux & 3>).
  RBT_VEC(uint32_t, r_pm); RBT_VEC(uint32_t, r_dm); RBT_VEC(uint32_t, r_ed); RBT_VEC(uint32_t, r_qp); RBT_VEC(uint32_t, r_ref);
  RBT_VEC(uint32_t, r_mv0); RBT_VEC(uint32_t, r_mv1); RBT_VEC(uint32_t, r_mv2); RBT_VEC(uint32_t, r_mv3);
  // Units around the CTB: lane 0 above-left corner, lanes 1..17 the row above (16 units + the first above-right one),
  // lanes 32..47 the column to the left. Availability is folded in: an unavailable unit reads pm = RBT_MODE_NONE.
  RBT_VEC(uint32_t, n_pm); RBT_VEC(uint32_t, n_dm); RBT_VEC(uint32_t, n_ref); RBT_VEC(uint32_t, n_mv);
#ifdef RBT_PROFILE
  unsigned long long t_res, t_ctb, t_cu, t_a, t_b, t_c, t_d, t_tu, t_hdr, t_fill, t_mpm, t_last; uint32_t n_res, n_cu;
#endif
};
struct RbtMv { int x, y, ref; };
// ---- packed parameter words ----
// c_dim: w | h << 16.  c_logs: bit_depth 0..3, log2_ctb 4..6, log2_min_cb 7..9, log2_min_tb 10..12, log2_max_tb 13..15, th_depth_inter 16..18,
// th_depth_intra 19..21, diff_cu_qp_delta_depth 22..23.  c_flags: amp 0, sao 1, strong_intra 2, tmvp 3, sign_hiding 4, cabac_init_present 5, cip 6,
// transform_skip 7, cu_qp_delta 8, tq_bypass_enabled 9, cb_qp_offset 16..23 (signed), cr_qp_offset 24..31 (signed).
// s_bits: slice_type 0..1, sao_luma 2, sao_chroma 3, temporal_mvp 4, cabac_init_flag 5, max_merge_cand 6..8, num_ref_idx 9..13, collocated_ref_idx 14..17.
// s_qp: qp 0..7, cb_qp_offset 8..15, cr_qp_offset 16..23 (all signed).
RBT_DEV int pzc_w(const RbtParse* s) { return (int)rbt_bfe<0, 16>(s->c_dim); }
RBT_DEV int pzc_h(const RbtParse* s) { return (int)rbt_bfe<16, 16>(s->c_dim); }
RBT_DEV int pzc_cw(const RbtParse* s) { return pzc_w(s) >> 1; }
RBT_DEV int pzc_ch(const RbtParse* s) { return pzc_h(s) >> 1; }
RBT_DEV int pzc_w4(const RbtParse* s) { return pzc_w(s) >> 2; }
RBT_DEV int pzc_h4(const RbtParse* s) { return pzc_h(s) >> 2; }
RBT_DEV int pzc_bit_depth(const RbtParse* s) { return (int)rbt_bfe<0, 4>(s->c_logs); }
RBT_DEV int pzc_log2_ctb(const RbtParse* s) { return (int)rbt_bfe<4, 3>(s->c_logs); }
RBT_DEV int pzc_log2_min_cb(const RbtParse* s) { return (int)rbt_bfe<7, 3>(s->c_logs); }
RBT_DEV int pzc_log2_min_tb(const RbtParse* s) { return (int)rbt_bfe<10, 3>(s->c_logs); }
RBT_DEV int pzc_log2_max_tb(const RbtParse* s) { return (int)rbt_bfe<13, 3>(s->c_logs); }
RBT_DEV int pzc_th_depth_inter(const RbtParse* s) { return (int)rbt_bfe<16, 3>(s->c_logs); }
RBT_DEV int pzc_th_depth_intra(const RbtParse* s) { return (int)rbt_bfe<19, 3>(s->c_logs); }
RBT_DEV int pzc_diff_cu_qp_delta_depth(const RbtParse* s) { return (int)rbt_bfe<22, 2>(s->c_logs); }
RBT_DEV int pzc_w_ctb(const RbtParse* s) { int L = pzc_log2_ctb(s); return (pzc_w(s) + (1 << L) - 1) >> L; }
RBT_DEV int pzc_h_ctb(const RbtParse* s) { int L = pzc_log2_ctb(s); return (pzc_h(s) + (1 << L) - 1) >> L; }
RBT_DEV int pzc_amp(const RbtParse* s) { return (int)rbt_bfe<0, 1>(s->c_flags); }
RBT_DEV int pzc_sign_hiding(const RbtParse* s) { return (int)rbt_bfe<4, 1>(s->c_flags); }
RBT_DEV int pzc_transform_skip(const RbtParse* s) { return (int)rbt_bfe<7, 1>(s->c_flags); }
RBT_DEV int pzc_cu_qp_delta(const RbtParse* s) { return (int)rbt_bfe<8, 1>(s->c_flags); }
RBT_DEV int pzc_tq_bypass_enabled(const RbtParse* s) { return (int)rbt_bfe<9, 1>(s->c_flags); }
RBT_DEV int pzc_cb_qp_offset(const RbtParse* s) { return rbt_bfe_i<16, 8>(s->c_flags); }
RBT_DEV int pzc_cr_qp_offset(const RbtParse* s) { return rbt_bfe_i<24, 8>(s->c_flags); }
RBT_DEV int pzs_slice_type(const RbtParse* s) { return (int)rbt_bfe<0, 2>(s->s_bits); }
RBT_DEV int pzs_sao_luma(const RbtParse* s) { return (int)rbt_bfe<2, 1>(s->s_bits); }
RBT_DEV int pzs_sao_chroma(const RbtParse* s) { return (int)rbt_bfe<3, 1>(s->s_bits); }
RBT_DEV int pzs_temporal_mvp(const RbtParse* s) { return (int)rbt_bfe<4, 1>(s->s_bits); }
RBT_DEV int pzs_cabac_init_flag(const RbtParse* s) { return (int)rbt_bfe<5, 1>(s->s_bits); }
RBT_DEV int pzs_max_merge_cand(const RbtParse* s) { return (int)rbt_bfe<6, 3>(s->s_bits); }
RBT_DEV int pzs_num_ref_idx(const RbtParse* s) { return (int)rbt_bfe<9, 5>(s->s_bits); }
RBT_DEV int pzs_collocated_ref_idx(const RbtParse* s) { return (int)rbt_bfe<14, 4>(s->s_bits); }
RBT_DEV int pzs_qp(const RbtParse* s) { return rbt_bfe_i<0, 8>(s->s_qp); }
RBT_DEV int pzs_cb_qp_offset(const RbtParse* s) { return rbt_bfe_i<8, 8>(s->s_qp); }
RBT_DEV int pzs_cr_qp_offset(const RbtParse* s) { return rbt_bfe_i<16, 8>(s->s_qp); }
RBT_DEV int pzs_poc(const RbtParse* s) { return s->s_poc; }
RBT_DEV int pz_il(const RbtParse* s, int i) { return (s->il_packed >> (8 * i)) & 255; }   // luma intra modes of the (up to four) PUs, 8 bits each
RBT_DEV void pz_set_il(RbtParse* s, int i, int v) { s->il_packed = (s->il_packed & ~(255 << (8 * i))) | (v << (8 * i)); }
#ifdef RBT_NO_WALKER_UNI
#define PZ_WU(x) (x)
#else
#define PZ_WU(x) RBT_UNI(x)
#endif

// ---- neighbour context -------------------------------------------------------------------------------------------
// The parser never reads the HBM maps of its own picture back. What later syntax depends on (prediction mode, skip,
// depth, intra mode, QP, motion) is kept per 4x4 unit in REGISTERS for the current CTB and the units around it, read with
// v_readlane and updated with a handful of lane-parallel VALU instructions per block (no LDS round trip, no sync).
// LDS only carries the context from one CTB to the next (right column, bottom row over the picture width, corner); the
// HBM maps are written once per CTB (pz_end_ctb) with plain stores that nobody waits for.
RBT_DEV int pz_idx(const RbtParse* s, int x, int y) { return (y >> 2) * pzc_w4(s) + (x >> 2); }
#define PZ_RD8(reg, k) ((int)((RBT_VGET(reg, (k) >> 2) >> (((k) & 3) * 8)) & 255u))
#define PZ_NB_BASE 256
// handle of luma position (xn,yn): -1 = not available (outside the picture, other slice, not decoded yet);
// else 0..255 = unit of the current CTB, PZ_NB_BASE + lane = surrounding unit
RBT_DEV int pz_ld_pm(const RbtParse* s, int loc) { return loc < PZ_NB_BASE ? PZ_RD8(s->r_pm, loc) : (int)RBT_VGET(s->n_pm, loc - PZ_NB_BASE); }
RBT_DEV int pz_ld_dm(const RbtParse* s, int loc) { return loc < PZ_NB_BASE ? PZ_RD8(s->r_dm, loc) : (int)RBT_VGET(s->n_dm, loc - PZ_NB_BASE); }
RBT_DEV int pz_ld_ref(const RbtParse* s, int loc) { return (int8_t)(loc < PZ_NB_BASE ? PZ_RD8(s->r_ref, loc) : (int)RBT_VGET(s->n_ref, loc - PZ_NB_BASE)); }
RBT_DEV int pz_ld_mv(const RbtParse* s, int loc) {
  if (loc >= PZ_NB_BASE) return (int)RBT_VGET(s->n_mv, loc - PZ_NB_BASE);
  int l = loc >> 2, j = loc & 3;
  uint32_t a = RBT_VGET(s->r_mv0, l), b = RBT_VGET(s->r_mv1, l), c = RBT_VGET(s->r_mv2, l), d = RBT_VGET(s->r_mv3, l);
  return (int)(j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d)));
}
// raw handle of luma position (xn,yn) from the geometry alone (no availability read): -1 = outside what is tracked
RBT_DEV int pz_loc_raw(const RbtParse* s, int xn, int yn) {
  const int dx = xn - s->ctb_x, dy = yn - s->ctb_y, ctb = 1 << pzc_log2_ctb(s);
  if (dx < -1 || dy < -1 || dy >= ctb) return -1;
  if (dy < 0) { const int i = (dx + 4) >> 2; return i > (ctb >> 2) + 1 ? -1 : PZ_NB_BASE + i; }
  if (dx < 0) return PZ_NB_BASE + 32 + (dy >> 2);
  if (dx >= ctb) return -1;
  return (dy >> 2) * 16 + (dx >> 2);
}
RBT_DEV int pz_loc(const RbtParse* s, int xn, int yn) {
  const int loc = pz_loc_raw(s, xn, yn);
  if (loc < 0) return -1;
  return (pz_ld_pm(s, loc) & RBT_PM_MODE_MASK) == RBT_MODE_NONE ? -1 : loc;
}
RBT_DEV int pz_avail(const RbtParse* s, int xn, int yn) { return pz_loc(s, xn, yn) >= 0; }
RBT_DEV int pz_mode(const RbtParse* s, int x, int y) { return pz_ld_pm(s, pz_loc(s, x, y)) & RBT_PM_MODE_MASK; }   // caller checked pz_avail
RBT_DEV int pz_dm(const RbtParse* s, int x, int y) { return pz_ld_dm(s, pz_loc(s, x, y)); }
// neighbour summary in one lookup: -1 = unavailable, else pm | dm << 8
RBT_DEV int pz_nb(const RbtParse* s, int x, int y) {
  const int loc = pz_loc_raw(s, x, y);
  if (loc < 0) return -1;
  const int pm = pz_ld_pm(s, loc), dm = pz_ld_dm(s, loc);            // both lane reads issued together: one VALU->SALU hand-over
  return (pm & RBT_PM_MODE_MASK) == RBT_MODE_NONE ? -1 : (pm | (dm << 8));
}
RBT_DEV int pz_cur(const RbtParse* s, int x, int y) { return ((y - s->ctb_y) >> 2) * 16 + ((x - s->ctb_x) >> 2); }
// ---- lane-parallel block updates: byte mask of the units of lane p that lie inside a block (unit coordinates in the CTB)
RBT_DEV uint32_t pz_sq_mask(int p, int bx, int by, int n4) {            // aligned square, n4 = 1, 2, 4, 8 or 16 units
  const int uy = p >> 2, ux0 = (p & 3) << 2;
  const uint32_t m = n4 >= 4 ? ((unsigned)(ux0 - bx) < (unsigned)n4 ? 0xFFFFFFFFu : 0u) : (ux0 == (bx & ~3) ? (((1u << (8 * n4)) - 1u) << (8 * (bx & 3))) : 0u);
  return (unsigned)(uy - by) < (unsigned)n4 ? m : 0u;
}
RBT_DEV uint32_t pz_rect_mask(int p, int bx, int by, int w4, int h4) {  // any rectangle (AMP prediction units)
  const int uy = p >> 2, ux0 = (p & 3) << 2; uint32_t m = 0;
  for (int j = 0; j < 4; j++) if ((unsigned)(ux0 + j - bx) < (unsigned)w4) m |= 0xFFu << (8 * j);
  return (unsigned)(uy - by) < (unsigned)h4 ? m : 0u;
}
RBT_DEV uint32_t pz_left_mask(int p, int bx, uint32_t m) { return ((p & 3) << 2) == (bx & ~3) ? (m & (0xFFu << (8 * (bx & 3)))) : 0u; }   // units in column bx
RBT_DEV uint32_t pz_top_mask(int p, int by, uint32_t m) { return (p >> 2) == by ? m : 0u; }                                                // units in row by
#define PZ_REP4(v) ((uint32_t)((v) & 255) * 0x01010101u)
#ifdef RBT_PROFILE
#define PZ_TF0() unsigned long long tf_ = __builtin_readcyclecounter()
#define PZ_TF1() (s->t_fill += __builtin_readcyclecounter() - tf_)
#else
#define PZ_TF0() ((void)0)
#define PZ_TF1() ((void)0)
#endif
// coding unit: every unit gets its mode / depth|mode / QP, and the CU boundary becomes a TU+PU edge. Each unit belongs to
// exactly one CU, so edges are assigned, not OR-ed.
RBT_DEV void pz_fill_cu(RbtParse* s, int x, int y, int N, int pm, int dm, int qp) {
  PZ_TF0();
  const int n4 = N >> 2, bx = (x - s->ctb_x) >> 2, by = (y - s->ctb_y) >> 2;
  RBT_VFOR(p, 64) {
    const uint32_t m = pz_sq_mask(p, bx, by, n4), ml = pz_left_mask(p, bx, m), mt = pz_top_mask(p, by, m);
    RBT_V(s->r_pm, p) = (RBT_V(s->r_pm, p) & ~m) | (PZ_REP4(pm) & m);
    RBT_V(s->r_dm, p) = (RBT_V(s->r_dm, p) & ~m) | (PZ_REP4(dm) & m);
    RBT_V(s->r_qp, p) = (RBT_V(s->r_qp, p) & ~m) | (PZ_REP4(qp) & m);
    RBT_V(s->r_ed, p) = (RBT_V(s->r_ed, p) & ~m) | (PZ_REP4(RBT_EV_TU | RBT_EV_PU) & ml) | (PZ_REP4(RBT_EH_TU | RBT_EH_PU) & mt);
  }
  PZ_TF1();
}
RBT_DEV void pz_fill_dm(RbtParse* s, int x, int y, int N, int v) {
  PZ_TF0();
  const int n4 = N >> 2, bx = (x - s->ctb_x) >> 2, by = (y - s->ctb_y) >> 2;
  RBT_VFOR(p, 64) { const uint32_t m = pz_sq_mask(p, bx, by, n4); RBT_V(s->r_dm, p) = (RBT_V(s->r_dm, p) & ~m) | (PZ_REP4(v) & m); }
  PZ_TF1();
}
RBT_DEV void pz_fill_qp(RbtParse* s, int x, int y, int N, int v) {
  PZ_TF0();
  const int n4 = N >> 2, bx = (x - s->ctb_x) >> 2, by = (y - s->ctb_y) >> 2;
  RBT_VFOR(p, 64) { const uint32_t m = pz_sq_mask(p, bx, by, n4); RBT_V(s->r_qp, p) = (RBT_V(s->r_qp, p) & ~m) | (PZ_REP4(v) & m); }
  PZ_TF1();
}
// transform unit: non-zero flag of the luma TB and its TU edges in one pass
RBT_DEV void pz_fill_tu(RbtParse* s, int x, int y, int N, int nz) {
  PZ_TF0();
  const int n4 = N >> 2, bx = (x - s->ctb_x) >> 2, by = (y - s->ctb_y) >> 2;
  RBT_VFOR(p, 64) {
    const uint32_t m = pz_sq_mask(p, bx, by, n4), ml = pz_left_mask(p, bx, m), mt = pz_top_mask(p, by, m);
    if (nz) RBT_V(s->r_pm, p) |= PZ_REP4(RBT_PM_NZ) & m;
    RBT_V(s->r_ed, p) |= (PZ_REP4(RBT_EV_TU) & ml) | (PZ_REP4(RBT_EH_TU) & mt);
  }
  PZ_TF1();
}
// prediction unit: motion, mode and PU edges of an arbitrary rectangle
RBT_DEV void pz_fill_pu(RbtParse* s, int x, int y, int w, int h, int mode, int ref, uint32_t packed_mv) {
  PZ_TF0();
  const int bx = (x - s->ctb_x) >> 2, by = (y - s->ctb_y) >> 2, w4 = w >> 2, h4 = h >> 2;
  RBT_VFOR(p, 64) {
    const uint32_t m = pz_rect_mask(p, bx, by, w4, h4), ml = pz_left_mask(p, bx, m), mt = pz_top_mask(p, by, m);
    RBT_V(s->r_pm, p) = (RBT_V(s->r_pm, p) & ~(PZ_REP4(RBT_PM_MODE_MASK) & m)) | (PZ_REP4(mode) & m);
    RBT_V(s->r_ref, p) = (RBT_V(s->r_ref, p) & ~m) | (PZ_REP4(ref) & m);
    if (m & 0x000000FFu) RBT_V(s->r_mv0, p) = packed_mv;
    if (m & 0x0000FF00u) RBT_V(s->r_mv1, p) = packed_mv;
    if (m & 0x00FF0000u) RBT_V(s->r_mv2, p) = packed_mv;
    if (m & 0xFF000000u) RBT_V(s->r_mv3, p) = packed_mv;
    RBT_V(s->r_ed, p) |= (PZ_REP4(RBT_EV_PU) & ml) | (PZ_REP4(RBT_EH_PU) & mt);
  }
  PZ_TF1();
}
// start of a CTB: nothing of it is decoded yet; fetch the surrounding units from the LDS line buffers
// Hand-over between the waves of two CTB rows of a wavefront stream. The record of row r (RbtFrame::prow_line, 256-byte aligned and only ever written by
// the wave of row r: the reading wave's own stores never touch its cache lines) holds the row's bottom line of units, its slice per CTB and its SAO parameters:
//   mv int32[w4] | RbtSao[w_ctb] | slice u16[w_ctb] | pm bytes[w4] | dm bytes[w4] | ref bytes[w4]
struct PzRowRec { int32_t* mv; RbtSao* sao; uint16_t* slice; uint8_t *pm, *dm; int8_t* ref; };
RBT_DEV PzRowRec pz_row_rec(const RbtParse* s, int row) {
  uint8_t* p = s->f->prow_line + (size_t)row * s->f->prow_line_bytes; const int w4 = pzc_w4(s), wc = pzc_w_ctb(s);
  PzRowRec r; r.mv = (int32_t*)p; r.sao = (RbtSao*)(p + (size_t)4 * w4); r.slice = (uint16_t*)(p + (size_t)4 * w4 + sizeof(RbtSao) * wc);
  r.pm = p + (size_t)4 * w4 + (sizeof(RbtSao) + 2) * wc; r.dm = r.pm + w4; r.ref = (int8_t*)(r.dm + w4);
  return r;
}
// producer: after pz_end_ctb of CTB (rx, ry) the line buffers hold this row's bottom line for the CTB's columns
RBT_DEV void pz_export_row(RbtParse* s, int rx, int ry) {
  RBT_LDS_AS RbtParseLds* L = s->L; const int cap4 = RBT_UNI(L->cap4), l2 = pzc_log2_ctb(s), n4 = 1 << (l2 - 2), w4 = pzc_w4(s), x40 = rx << (l2 - 2), cnt = rbt_min(n4, w4 - x40);
  const PzRowRec r = pz_row_rec(s, ry);
  RBT_LDS_AS uint8_t *a_pm = pz_above_pm(L, cap4), *a_dm = pz_above_dm(L, cap4); RBT_LDS_AS int8_t* a_ref = pz_above_ref(L, cap4); RBT_LDS_AS int32_t* a_mv = pz_above_mv(L);
  RBT_PAR_FOR(i, cnt) { r.pm[x40 + i] = a_pm[x40 + i]; r.dm[x40 + i] = a_dm[x40 + i]; r.ref[x40 + i] = a_ref[x40 + i]; r.mv[x40 + i] = a_mv[x40 + i]; }
  if (RBT_LANE0) {
    r.slice[rx] = pz_above_slice(L, cap4)[rx];
    uint32_t* g = (uint32_t*)&r.sao[rx]; const RBT_LDS_AS uint32_t* a = (const RBT_LDS_AS uint32_t*)&pz_sao_above(L, cap4)[rx];
    for (int w = 0; w < 6; w++) g[w] = a[w];
  }
}
// consumer (row task, first CTB row of this wave): CTB column `col` of the row above into the line buffers this wave would have filled itself
RBT_DEV void pz_import_above(RbtParse* s, int col, int ry) {
  RBT_LDS_AS RbtParseLds* L = s->L;
  const int cap4 = RBT_UNI(L->cap4), l2 = pzc_log2_ctb(s), n4 = 1 << (l2 - 2), w4 = pzc_w4(s), wc = pzc_w_ctb(s), x40 = col << (l2 - 2);
  if (col >= wc) return;
  const int cnt = rbt_min(n4, w4 - x40); const PzRowRec r = pz_row_rec(s, ry - 1);
  RBT_LDS_AS uint8_t *a_pm = pz_above_pm(L, cap4), *a_dm = pz_above_dm(L, cap4); RBT_LDS_AS int8_t* a_ref = pz_above_ref(L, cap4); RBT_LDS_AS int32_t* a_mv = pz_above_mv(L);
  RBT_PAR_FOR(i, cnt) { a_pm[x40 + i] = r.pm[x40 + i]; a_dm[x40 + i] = r.dm[x40 + i]; a_ref[x40 + i] = r.ref[x40 + i]; a_mv[x40 + i] = r.mv[x40 + i]; }
  if (RBT_LANE0) {
    pz_above_slice(L, cap4)[col] = r.slice[col];
    const uint32_t* g = (const uint32_t*)&r.sao[col]; RBT_LDS_AS uint32_t* a = (RBT_LDS_AS uint32_t*)&pz_sao_above(L, cap4)[col];
    for (int w = 0; w < 6; w++) a[w] = g[w];
  }
  RBT_SYNC_LDS();
}
RBT_DEV void pz_begin_ctb(RbtParse* s, int rx, int ry) {
  RBT_LDS_AS RbtParseLds* L = s->L;
  s->ctb_x = rx << pzc_log2_ctb(s); s->ctb_y = ry << pzc_log2_ctb(s);
  const int cx4 = s->ctb_x >> 2, cap4 = RBT_UNI(L->cap4);
  RBT_LDS_AS uint8_t *a_pm = pz_above_pm(L, cap4), *a_dm = pz_above_dm(L, cap4); RBT_LDS_AS int8_t* a_ref = pz_above_ref(L, cap4); RBT_LDS_AS int32_t* a_mv = pz_above_mv(L);
  RBT_LDS_AS uint16_t* a_slice = pz_above_slice(L, cap4);
  RBT_VFOR(p, 64) {
    RBT_V(s->r_pm, p) = PZ_REP4(RBT_MODE_NONE); RBT_V(s->r_dm, p) = PZ_REP4(1); RBT_V(s->r_ed, p) = 0; RBT_V(s->r_qp, p) = 0; RBT_V(s->r_ref, p) = 0xFFFFFFFFu;
    RBT_V(s->r_mv0, p) = 0; RBT_V(s->r_mv1, p) = 0; RBT_V(s->r_mv2, p) = 0; RBT_V(s->r_mv3, p) = 0;
    uint32_t pm = RBT_MODE_NONE, dm = 1, ref = 0xFF, mv = 0;
    if (p == 0) { if (s->corner_ok) { pm = (uint32_t)s->corner_pm; dm = (uint32_t)s->corner_dm; ref = (uint32_t)s->corner_ref & 255u; mv = (uint32_t)s->corner_mv; } }
    else if (p <= 17) {
      const int xa4 = cx4 - 1 + p;
      if (ry > 0 && xa4 < pzc_w4(s) && a_slice[(xa4 << 2) >> pzc_log2_ctb(s)] == s->slice_idx) { pm = a_pm[xa4]; dm = a_dm[xa4]; ref = (uint8_t)a_ref[xa4]; mv = (uint32_t)a_mv[xa4]; }
    } else if (p >= 32 && p < 48) {
      if (s->left_ok) { pm = L->left_pm[p - 32]; dm = L->left_dm[p - 32]; ref = (uint8_t)L->left_ref[p - 32]; mv = (uint32_t)L->left_mv[p - 32]; }
    }
    RBT_V(s->n_pm, p) = pm; RBT_V(s->n_dm, p) = dm; RBT_V(s->n_ref, p) = ref; RBT_V(s->n_mv, p) = mv;
  }
}
// end of a CTB: spill its units to LDS, write them to the HBM maps, then roll the line buffers
RBT_DEV void pz_end_ctb(RbtParse* s, int rx, int ry) {
  RBT_LDS_AS RbtParseLds* L = s->L;
  int cx = s->ctb_x, cy = s->ctb_y, ctb = 1 << pzc_log2_ctb(s), n4 = ctb >> 2;
  RBT_VFOR(p, 64) {
    const uint32_t pm = RBT_V(s->r_pm, p), dm = RBT_V(s->r_dm, p), ed = RBT_V(s->r_ed, p), qp = RBT_V(s->r_qp, p), rf = RBT_V(s->r_ref, p);
    for (int j = 0; j < 4; j++) {
      const int k = 4 * p + j;
      L->cur_pm[k] = (uint8_t)(pm >> (8 * j)); L->cur_dm[k] = (uint8_t)(dm >> (8 * j)); L->cur_edges[k] = (uint8_t)(ed >> (8 * j));
      L->cur_qp[k] = (int8_t)(qp >> (8 * j)); L->cur_ref[k] = (int8_t)(rf >> (8 * j));
    }
    L->cur_mv[4 * p] = (int32_t)RBT_V(s->r_mv0, p); L->cur_mv[4 * p + 1] = (int32_t)RBT_V(s->r_mv1, p); L->cur_mv[4 * p + 2] = (int32_t)RBT_V(s->r_mv2, p); L->cur_mv[4 * p + 3] = (int32_t)RBT_V(s->r_mv3, p);
  }
  RBT_SYNC_LDS();
  // map pointers are read from the picture record here, once per CTB, instead of living in registers for the whole slice
  uint8_t *m_pm = s->f->pm, *m_dm = s->f->dm, *m_edges = s->f->edges; int8_t *m_qp = s->f->qp, *m_ref = s->f->ref; int16_t* m_mv = s->f->mv; int32_t* m_refpoc = s->f->refpoc;
  RBT_PAR_FOR(u, n4 * n4) {
    int ux = u % n4, uy = u / n4, x = cx + ux * 4, y = cy + uy * 4;
    if (x < pzc_w(s) && y < pzc_h(s)) {
      int k = uy * 16 + ux, gk = (y >> 2) * pzc_w4(s) + (x >> 2), pm = L->cur_pm[k], ref = L->cur_ref[k], mv = L->cur_mv[k];
      m_pm[gk] = (uint8_t)pm; m_dm[gk] = L->cur_dm[k]; m_edges[gk] = L->cur_edges[k]; m_qp[gk] = L->cur_qp[k]; m_ref[gk] = (int8_t)ref;
      m_mv[2 * gk] = (int16_t)(mv & 0xFFFF); m_mv[2 * gk + 1] = (int16_t)(mv >> 16);
      int mode = pm & RBT_PM_MODE_MASK;
      m_refpoc[gk] = (mode == RBT_MODE_INTER || mode == RBT_MODE_SKIP) ? s->L->ref_poc[ref < 0 ? 0 : ref] : RBT_NO_REFPOC;
    }
  }
  // above-left corner of the NEXT CTB = last unit of the old above row under this CTB
  const int cap4 = RBT_UNI(L->cap4);
  RBT_LDS_AS uint8_t *a_pm = pz_above_pm(L, cap4), *a_dm = pz_above_dm(L, cap4); RBT_LDS_AS int8_t* a_ref = pz_above_ref(L, cap4); RBT_LDS_AS int32_t* a_mv = pz_above_mv(L);
  RBT_LDS_AS uint16_t* a_slice = pz_above_slice(L, cap4);
  int last = rbt_min(cx + ctb, pzc_w(s)) / 4 - 1;
  int c_ok = ry > 0 && a_slice[rx] == s->slice_idx;
  int c_pm = a_pm[last], c_dm = a_dm[last], c_ref = a_ref[last], c_mv = a_mv[last];
  RBT_SYNC_LDS();
  s->corner_ok = c_ok; s->corner_pm = c_pm; s->corner_dm = c_dm; s->corner_ref = c_ref; s->corner_mv = c_mv;
  int rows = rbt_min(ctb, pzc_h(s) - cy) >> 2, cols = rbt_min(ctb, pzc_w(s) - cx) >> 2;
  RBT_PAR_FOR(i, cols) { int k = (rows - 1) * 16 + i, a = (cx >> 2) + i; a_pm[a] = L->cur_pm[k]; a_dm[a] = L->cur_dm[k]; a_ref[a] = L->cur_ref[k]; a_mv[a] = L->cur_mv[k]; }
  RBT_PAR_FOR(i, 16) { int k = i * 16 + cols - 1; L->left_pm[i] = i < rows ? L->cur_pm[k] : RBT_MODE_NONE; L->left_dm[i] = L->cur_dm[k]; L->left_ref[i] = L->cur_ref[k]; L->left_mv[i] = L->cur_mv[k]; }
  if (RBT_LANE0) a_slice[rx] = (uint16_t)s->slice_idx;
  s->left_ok = rx + 1 < pzc_w_ctb(s);
  if (rx + 1 >= pzc_w_ctb(s)) s->corner_ok = 0;
  RBT_SYNC_LDS();
}
RBT_DEV void pz_emit(RbtParse* s, const RbtCmd& cmd) {
  if ((int)s->n_cmds >= s->m_cap) { s->error = 3; return; }
  if (RBT_LANE0) s->m_cmds[(size_t)s->ctb_addr * s->m_cap + s->n_cmds] = cmd;
  s->n_cmds++;
}
// the same record assembled as four words (one 16-byte store): type | x4 << 8 | y4 << 16 | log2 << 24, a | b << 8 | c << 16 | d << 24,
// mvx | mvy << 16, qp[0] | qp[1] << 8 | qp[2] << 16
struct alignas(16) RbtCmdWords { uint32_t w[4]; };
RBT_DEV void pz_emit_words(RbtParse* s, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  if ((int)s->n_cmds >= s->m_cap) { s->error = 3; return; }
  if (RBT_LANE0) { RbtCmdWords v; v.w[0] = w0; v.w[1] = w1; v.w[2] = w2; v.w[3] = w3; *(RbtCmdWords*)&s->m_cmds[(size_t)s->ctb_addr * s->m_cap + s->n_cmds] = v; }
  s->n_cmds++;
}

// SAO parameters of the current CTB while they are parsed: plain ints reached through compile-time indices only, so they stay in registers (the 24-byte
// RbtSao with its byte arrays, indexed by a run-time component, lived in scratch memory: a round trip to it per access on the slice's latency chain).
struct PzSao { int type[3], band[3], eo[3], off[3][4]; };
// RbtSao is 24 plain bytes = six words: type[3] band_pos[3] eo_class[3] offset[3][4] pad[3]
RBT_DEV uint32_t pz_sao_word(const PzSao* p, int w) {
  const uint32_t by[24] = {(uint32_t)p->type[0], (uint32_t)p->type[1], (uint32_t)p->type[2], (uint32_t)p->band[0], (uint32_t)p->band[1], (uint32_t)p->band[2], (uint32_t)p->eo[0], (uint32_t)p->eo[1],
                           (uint32_t)p->eo[2], (uint32_t)p->off[0][0], (uint32_t)p->off[0][1], (uint32_t)p->off[0][2], (uint32_t)p->off[0][3], (uint32_t)p->off[1][0], (uint32_t)p->off[1][1], (uint32_t)p->off[1][2],
                           (uint32_t)p->off[1][3], (uint32_t)p->off[2][0], (uint32_t)p->off[2][1], (uint32_t)p->off[2][2], (uint32_t)p->off[2][3], 0u, 0u, 0u};
  return (by[4 * w] & 255u) | ((by[4 * w + 1] & 255u) << 8) | ((by[4 * w + 2] & 255u) << 16) | ((by[4 * w + 3] & 255u) << 24);
}
RBT_DEV void pz_sao_unpack(PzSao* p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5) {
#define PZ_B(w, k) (int)(((w) >> (8 * (k))) & 255u)
#define PZ_S(w, k) (int)(int8_t)(((w) >> (8 * (k))) & 255u)
  p->type[0] = PZ_B(w0, 0); p->type[1] = PZ_B(w0, 1); p->type[2] = PZ_B(w0, 2); p->band[0] = PZ_B(w0, 3); p->band[1] = PZ_B(w1, 0); p->band[2] = PZ_B(w1, 1);
  p->eo[0] = PZ_B(w1, 2); p->eo[1] = PZ_B(w1, 3); p->eo[2] = PZ_B(w2, 0);
  p->off[0][0] = PZ_S(w2, 1); p->off[0][1] = PZ_S(w2, 2); p->off[0][2] = PZ_S(w2, 3); p->off[0][3] = PZ_S(w3, 0);
  p->off[1][0] = PZ_S(w3, 1); p->off[1][1] = PZ_S(w3, 2); p->off[1][2] = PZ_S(w3, 3); p->off[1][3] = PZ_S(w4, 0);
  p->off[2][0] = PZ_S(w4, 1); p->off[2][1] = PZ_S(w4, 2); p->off[2][2] = PZ_S(w4, 3); p->off[2][3] = PZ_S(w5, 0);
#undef PZ_B
#undef PZ_S
}
RBT_DEV void pz_sao_from_lds(PzSao* d, const RBT_LDS_AS RbtSao* l) { const RBT_LDS_AS uint32_t* w = (const RBT_LDS_AS uint32_t*)l; pz_sao_unpack(d, w[0], w[1], w[2], w[3], w[4], w[5]); }
// sao_offset_abs / sign / band_position / eo_class of component CI (7.3.8.3)
template <int CI> RBT_DEV void pz_sao_comp(RbtParse* s, RbtCabacDec* c, PzSao* p, int bd, int cmax) {
  if ((CI == 0 && !pzs_sao_luma(s)) || (CI > 0 && !pzs_sao_chroma(s))) return;
  if (CI == 2) p->type[2] = p->type[1];
  else { int t = 0; if (rbt_cd_bin(c, CTX_SAO_TYPE)) t = rbt_cd_bypass(c) ? 2 : 1; p->type[CI] = t; }
  const int type = p->type[CI];
  if (!type) return;
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  while (a0 < cmax && rbt_cd_bypass(c)) a0++;
  while (a1 < cmax && rbt_cd_bypass(c)) a1++;
  while (a2 < cmax && rbt_cd_bypass(c)) a2++;
  while (a3 < cmax && rbt_cd_bypass(c)) a3++;
  if (type == 1) {
    if (a0 && rbt_cd_bypass(c)) a0 = -a0;
    if (a1 && rbt_cd_bypass(c)) a1 = -a1;
    if (a2 && rbt_cd_bypass(c)) a2 = -a2;
    if (a3 && rbt_cd_bypass(c)) a3 = -a3;
    p->band[CI] = (int)rbt_cd_bypass_n(c, 5);
  } else {
    a2 = -a2; a3 = -a3;
    if (CI == 0) p->eo[0] = (int)rbt_cd_bypass_n(c, 2);
    else if (CI == 1) p->eo[1] = (int)rbt_cd_bypass_n(c, 2);
    else p->eo[2] = p->eo[1];
  }
  const int sc = 1 << (bd - rbt_min(bd, 10));
  p->off[CI][0] = a0 * sc; p->off[CI][1] = a1 * sc; p->off[CI][2] = a2 * sc; p->off[CI][3] = a3 * sc;
}
// ------------------------------------------------------------------------------------------------ SAO (7.3.8.3)
RBT_DEV void pz_sao(RbtParse* s, int rx, int ry) {
  RbtCabacDec* c = &s->c;
  rx = PZ_WU(rx); ry = PZ_WU(ry);
  PzSao p; pz_sao_unpack(&p, 0u, 0u, 0u, 0u, 0u, 0u);
  int wc = pzc_w_ctb(s);
  if (pzs_sao_luma(s) || pzs_sao_chroma(s)) {
    int merge_left = 0, merge_up = 0;
    if (rx > 0 && s->left_ok) merge_left = rbt_cd_bin(c, CTX_SAO_MERGE);
    if (ry > 0 && !merge_left && pz_above_slice(s->L, RBT_UNI(s->L->cap4))[rx] == s->slice_idx) merge_up = rbt_cd_bin(c, CTX_SAO_MERGE);
    if (merge_left) pz_sao_from_lds(&p, &s->L->sao_left);
    else if (merge_up) pz_sao_from_lds(&p, &pz_sao_above(s->L, RBT_UNI(s->L->cap4))[rx]);
    else {
      int bd = pzc_bit_depth(s), cmax = (1 << (rbt_min(bd, 10) - 5)) - 1;
      pz_sao_comp<0>(s, c, &p, bd, cmax); pz_sao_comp<1>(s, c, &p, bd, cmax); pz_sao_comp<2>(s, c, &p, bd, cmax);
    }
  }
  if (RBT_LANE0) {
    uint32_t* g = (uint32_t*)&s->f->sao[ry * wc + rx]; RBT_LDS_AS uint32_t* l = (RBT_LDS_AS uint32_t*)&s->L->sao_left; RBT_LDS_AS uint32_t* a = (RBT_LDS_AS uint32_t*)&pz_sao_above(s->L, s->L->cap4)[rx];
#pragma unroll
    for (int w = 0; w < 6; w++) { const uint32_t v = pz_sao_word(&p, w); g[w] = v; l[w] = v; a[w] = v; }
  }
  RBT_SYNC_LDS();
}

// 4x4 scan positions packed 4 bits per entry (x | y << 2): diagonal, horizontal, vertical; and ctxIdxMap of 4x4 TBs
RBT_DEV uint64_t pz_scan4_const(int scan_idx) { return scan_idx == 0 ? 0xFBE7AD369C258140ull : (scan_idx == 1 ? 0xFEDCBA9876543210ull : 0xFB73EA62D951C840ull); }
#define PZ_SIGCTX4 0x8877886654325410ull
// ------------------------------------------------------------------------------------------------ residual_coding (7.3.8.11)
// Levels go straight to the dense coefficient plane of component c_idx at TB origin (x0,y0) (component samples).
RBT_DEV int pz_residual(RbtParse* s, int c_idx, int x0, int y0, int log2, int scan_idx) {
#ifdef RBT_PROFILE
  unsigned long long t0_ = __builtin_readcyclecounter(); s->n_res++;
#endif
  PZ_STAMP(s, 0);
  RbtCabacDec cl; rbt_cd_localise(&cl, &s->c); RbtCabacDec* c = &cl;
  x0 = RBT_UNI(x0); y0 = RBT_UNI(y0); log2 = RBT_UNI(log2); scan_idx = RBT_UNI(scan_idx);
  const uint64_t ps = pz_scan4_const(scan_idx);
#define PZ_POS(n) ((int)((ps >> (4 * (n))) & 15))
  int16_t* plane = rbt_uni_ptr(c_idx == 0 ? s->m_coef0 : (c_idx == 1 ? s->m_coef1 : s->m_coef2)); int pst = RBT_UNI(c_idx ? pzc_cw(s) : pzc_w(s));
  const int tq_bypass = RBT_UNI(s->cu_tq_bypass), sdh_on = RBT_UNI(pzc_sign_hiding(s)), ts_on = RBT_UNI(pzc_transform_skip(s));
  int ts_flag = 0;
  if (ts_on && !tq_bypass && log2 <= 2) ts_flag = rbt_cd_bin(c, CTX_TRANSFORM_SKIP + (c_idx ? 1 : 0));
  const int chroma = c_idx != 0;
  int ctx_off, ctx_shift;
  if (c_idx == 0) { ctx_off = 3 * (log2 - 2) + ((log2 - 1) >> 2); ctx_shift = (log2 + 1) >> 2; }
  else { ctx_off = 15; ctx_shift = log2 - 2; }
  PZ_STAMP(s, 1);
  int maxp = (log2 << 1) - 1, px = 0, py = 0;
  while (px < maxp && rbt_cd_bin_last(c, ctx_off + (px >> ctx_shift))) px++;
  while (py < maxp && rbt_cd_bin_last(c, 18 + ctx_off + (py >> ctx_shift))) py++;
  int lx = px, ly = py;
  if (px > 3) { int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + (int)rbt_cd_bypass_n<false>(c, nb); }
  if (py > 3) { int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + (int)rbt_cd_bypass_n<false>(c, nb); }
  if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
#ifdef RBT_PROFILE
  unsigned long long ta_ = __builtin_readcyclecounter(); s->t_a += ta_ - t0_;
#endif
  // Everything that is not the serial arithmetic decode runs on the lanes: lane i holds sub-block scan entry i, lane p
  // (p < 16) the position of scan index p inside a 4x4 sub-block, its sig_coeff_flag context and, after the bins of a
  // sub-block are known, the level / sign / address of the coefficient at that scan index.
  PZ_STAMP(s, 2);
  // (round 3: per-slice lane tables with shifts and lane reads instead of this LDS read and the two ballots were measured SLOWER - 321 against 188 cycles per block)
  const RBT_LDS_AS uint8_t* sb_scan = s->L->scan[scan_idx][log2 - 2];
  const int n_sb = 1 << (2 * (log2 - 2));
  RBT_VEC(int, v_sbscan); RBT_VEC(int, v_pos);
  int last_sb, last_pos;
  if (log2 == 2) {     // (round 4) a 4x4 block - most blocks of an intra stream - is its only sub-block: no scan table to fetch from LDS, no search for the last sub-block
    RBT_VFOR(p, 64) { RBT_V(v_sbscan, p) = p == 0 ? 0 : 0xFFFF; RBT_V(v_pos, p) = PZ_POS(p & 15); }
    last_sb = 0;
  } else {
    RBT_VFOR(p, 64) { RBT_V(v_sbscan, p) = p < n_sb ? (int)sb_scan[p] : 0xFFFF; RBT_V(v_pos, p) = PZ_POS(p & 15); }
    uint64_t mb; const int key = (lx >> 2) | ((ly >> 2) << 4);
    RBT_VBALLOT(mb, p, 64, RBT_V(v_sbscan, p) == key); last_sb = mb ? __builtin_ctzll(mb) : 0;
  }
  { uint64_t mb; const int ikey = (lx & 3) | ((ly & 3) << 2);
    RBT_VBALLOT(mb, p, 16, RBT_V(v_pos, p) == ikey); last_pos = mb ? __builtin_ctzll(mb) : 0; }
  PZ_STAMP(s, 3);
  uint64_t csbf = 0;   // bit (ys*8+xs)
  const int sbw = 1 << (log2 - 2);
  int greater1_ctx = 1, first_sb_done = 0, err = 0;
  const int sign_hiding = sdh_on && !tq_bypass;
  const int sig_c0 = chroma ? 27 : 0, g1_c0 = chroma ? 16 : 0, g2_c0 = chroma ? 4 : 0;
  for (int i = last_sb; i >= 0; i--) {
    const int sbv = RBT_VGET(v_sbscan, i), xs = sbv & 15, ys = sbv >> 4;
    const int right = xs + 1 < sbw ? (int)((csbf >> (ys * 8 + xs + 1)) & 1) : 0, below = ys + 1 < sbw ? (int)((csbf >> ((ys + 1) * 8 + xs)) & 1) : 0;
#ifdef RBT_PROFILE
    unsigned long long tb0_ = __builtin_readcyclecounter();
#endif
    int infer_dc = 0;
    if (i < last_sb && i > 0) { if (!rbt_cd_bin_csbf(c, rbt_min(right + below, 1) + (chroma ? 2 : 0))) continue; infer_dc = 1; }
    csbf |= 1ull << (ys * 8 + xs);
    // sigCtx (9.3.4.2.5) = per-CG base + a 2-bit pattern value looked up by the position inside the CG
    const int prev_csbf = right | (below << 1);
    const uint32_t pat = prev_csbf == 0 ? 0x00010516u : (prev_csbf == 1 ? 0x000055AAu : (prev_csbf == 2 ? 0x06060606u : 0xAAAAAAAAu));
    const int cg_base = !chroma ? ((xs | ys) ? 3 : 0) + (log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21) : 27 + (log2 == 3 ? 9 : 12);
    const int dc_cg = (xs | ys) == 0;
    RBT_VEC(int, v_sc);
    RBT_VFOR(p, 16) {
      const int p4 = RBT_V(v_pos, p);                  // x | y << 2 = raster index inside the sub-block
      RBT_V(v_sc, p) = log2 == 2 ? (int)((PZ_SIGCTX4 >> (4 * p4)) & 15) + sig_c0 : ((dc_cg && p4 == 0) ? sig_c0 : cg_base + (int)((pat >> (2 * p4)) & 3));
    }
    uint32_t sig_mask = i == last_sb ? 1u << last_pos : 0u;   // bit n = coefficient at scan index n significant
    const int start = i == last_sb ? last_pos - 1 : 15;
    for (int n = start; n >= 1; n--) sig_mask |= (uint32_t)rbt_cd_bin_sig(c, RBT_VGET(v_sc, n)) << n;
    if (start >= 0) {
      if (infer_dc && sig_mask == 0) sig_mask = 1u;
      else sig_mask |= (uint32_t)rbt_cd_bin_sig(c, RBT_VGET(v_sc, 0));
    }
#ifdef RBT_PROFILE
    unsigned long long tb1_ = __builtin_readcyclecounter(); s->t_b += tb1_ - tb0_;
#endif
    if (!sig_mask) continue;
    const int nsig = __builtin_popcount(sig_mask);
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    ctx_set += first_sb_done & (int)((uint32_t)(greater1_ctx - 1) >> 31);   // previous sub-block ended with greater1Ctx == 0
    first_sb_done = 1; greater1_ctx = 1;
    // the k-th significant coefficient in decode order is the k-th set bit of sig_mask from the top
    uint32_t g1_mask = 0; const int n8 = rbt_min(nsig, 8), g1_base = (ctx_set << 2) + g1_c0;
    for (int k = 0; k < n8; k++) {
      const int g1 = rbt_cd_bin_gt1(c, g1_base + greater1_ctx);
      g1_mask |= (uint32_t)g1 << k;
      // 0 stays 0, a 1 bin resets to 0, otherwise count up to 3 (integer arithmetic, see rbt_cd_core)
      greater1_ctx = (int)((0x3320u >> (4 * greater1_ctx)) & 15u) & (g1 - 1);
    }
    const int first_g1 = g1_mask ? __builtin_ctz(g1_mask) : -1;
    int g2 = 0;
    if (g1_mask) g2 = rbt_cd_bin_gt2(c, ctx_set + g2_c0);
#ifdef RBT_PROFILE
    unsigned long long tc1_ = __builtin_readcyclecounter(); s->t_c += tc1_ - tb1_;
#endif
    const int hi = 31 - __builtin_clz(sig_mask), lo = __builtin_ctz(sig_mask);
    const int hidden = sign_hiding && (hi - lo > 3);
    const int nsign = nsig - (hidden ? 1 : 0);
    // lanes: decode-order index, base level and "needs coeff_abs_level_remaining" of every scan index
    RBT_VEC(int, v_k); RBT_VEC(int, v_a); RBT_VEC(int, v_rem); uint64_t need64;
    RBT_VFOR(p, 16) {
      const int k = __builtin_popcount(sig_mask >> (p + 1)), isf = k == first_g1;
      const int a = 1 + (k < 8 ? (int)((g1_mask >> k) & 1) : 0) + (isf ? g2 : 0);
      RBT_V(v_k, p) = k; RBT_V(v_a, p) = a; RBT_V(v_rem, p) = 0;
    }
    RBT_VBALLOT(need64, p, 16, ((sig_mask >> p) & 1) && RBT_V(v_a, p) == (RBT_V(v_k, p) < 8 ? (RBT_V(v_k, p) == first_g1 ? 3 : 2) : 1));
    const uint32_t signs = rbt_cd_bypass_n<false>(c, nsign);     // sign of decode-order index k = bit nsign-1-k
    { uint32_t m = (uint32_t)need64; int rice = 0;
      while (m) {
        const int n = 31 - __builtin_clz(m); m &= ~(1u << n);
        int pre = 0; while (pre < 32 && rbt_cd_bypass<false>(c)) pre++;
        int v;
        if (pre <= 3) v = (pre << rice) + (int)rbt_cd_bypass_n<false>(c, rice);
        else { int sl = pre - 3 + rice; if (sl > 30) { err = 4; sl = 30; } v = (int)((((1u << (pre - 3)) + 3u - 1u) << rice) + rbt_cd_bypass_n<false>(c, sl)); }   // no early exit: keeps the loop single-exit
        const int k = __builtin_popcount(sig_mask >> (n + 1)), a = (k < 8 ? (k == first_g1 ? 3 : 2) : 1) + v;
        if (a > 3 * (1 << rice)) rice = rbt_min(rice + 1, 4);
        RBT_VSET(v_rem, n, v);
      } }
    uint64_t odd64 = 0;
    if (hidden) RBT_VBALLOT(odd64, p, 16, ((sig_mask >> p) & 1) && ((RBT_V(v_a, p) + RBT_V(v_rem, p)) & 1));
    const int parity = __builtin_popcountll(odd64) & 1;
    RBT_VFOR(p, 16) {
      if ((sig_mask >> p) & 1) {
        const int k = RBT_V(v_k, p), a = RBT_V(v_a, p) + RBT_V(v_rem, p), p4 = RBT_V(v_pos, p);
        const int neg = k < nsign ? (int)((signs >> (nsign - 1 - k)) & 1) : parity;
        const int xc = (xs << 2) + (p4 & 3), yc = (ys << 2) + (p4 >> 2);
        plane[(size_t)(y0 + yc) * pst + x0 + xc] = (int16_t)rbt_clip3(-32768, 32767, neg ? -a : a);
      }
    }
#ifdef RBT_PROFILE
    s->t_d += __builtin_readcyclecounter() - tc1_;
#endif
  }
  PZ_STAMP(s, 4);
  s->c = cl;
  if (err) s->error = err;
  PZ_STAMP(s, 5);
#ifdef RBT_PROFILE
  s->t_res += __builtin_readcyclecounter() - t0_;
#endif
  return ts_flag;
#undef PZ_POS
}

// ------------------------------------------------------------------------------------------------ QP
RBT_DEV int pz_wrap_qp(const RbtParse* s, int v) { int bdo = 6 * (pzc_bit_depth(s) - 8); return ((v + 52 + 2 * bdo) % (52 + bdo)) - bdo; }
RBT_DEV void pz_start_qg(RbtParse* s, int xqg, int yqg) {
  int ctb_mask = ~((1 << pzc_log2_ctb(s)) - 1);
  s->qp_y_prev = s->qp_y; s->is_cu_qp_delta_coded = 0; s->cu_qp_delta_val = 0;
  int qa = s->qp_y_prev, qb = s->qp_y_prev;
  if (xqg > 0 && ((xqg - 1) & ctb_mask) == (xqg & ctb_mask) && pz_avail(s, xqg - 1, yqg)) qa = (int8_t)PZ_RD8(s->r_qp, pz_cur(s, xqg - 1, yqg));
  if (yqg > 0 && ((yqg - 1) & ctb_mask) == (yqg & ctb_mask) && pz_avail(s, xqg, yqg - 1)) qb = (int8_t)PZ_RD8(s->r_qp, pz_cur(s, xqg, yqg - 1));
  s->qp_pred = (qa + qb + 1) >> 1;
}
RBT_DEV int pz_chroma_qp(const RbtParse* s, int c_idx) {
  int off = c_idx == 1 ? pzc_cb_qp_offset(s) + pzs_cb_qp_offset(s) : pzc_cr_qp_offset(s) + pzs_cr_qp_offset(s);
  int bdo = 6 * (pzc_bit_depth(s) - 8);
  int qpi = rbt_clip3(-bdo, 57, s->qp_y + off);
  return (qpi < 0 ? qpi : rbt_chroma_qp(qpi)) + bdo;
}
RBT_DEV int pz_scan_idx(int pred_mode, int log2, int c_idx, int intra_mode) {
  if (pred_mode == RBT_MODE_INTRA && (log2 == 2 || (log2 == 3 && c_idx == 0))) {
    if (intra_mode >= 6 && intra_mode <= 14) return 2;
    if (intra_mode >= 22 && intra_mode <= 30) return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ transform tree (7.3.8.8-10)
RBT_DEV void pz_transform_unit(RbtParse* s, int x0, int y0, int xb, int yb, int log2, int blk, int cbf_luma, int cbf_cb, int cbf_cr) {
  RbtCabacDec* c = &s->c;
#ifdef RBT_PROFILE
  unsigned long long ttu_ = __builtin_readcyclecounter();
#endif
  x0 = PZ_WU(x0); y0 = PZ_WU(y0); xb = PZ_WU(xb); yb = PZ_WU(yb); log2 = PZ_WU(log2); blk = PZ_WU(blk); cbf_luma = PZ_WU(cbf_luma); cbf_cb = PZ_WU(cbf_cb); cbf_cr = PZ_WU(cbf_cr);
  int N = 1 << log2;
  if ((cbf_luma || cbf_cb || cbf_cr) && pzc_cu_qp_delta(s) && !s->is_cu_qp_delta_coded) {
    int v = 0; while (v < 5 && rbt_cd_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0))) v++;
    if (v == 5) { int k = 0; while (k < 16 && rbt_cd_bypass(c)) { v += 1 << k; k++; } v += (int)rbt_cd_bypass_n(c, k); }
    if (v && rbt_cd_bypass(c)) v = -v;
    s->is_cu_qp_delta_coded = 1; s->cu_qp_delta_val = v;
    s->qp_y = pz_wrap_qp(s, s->qp_pred + v);
    pz_fill_qp(s, s->cu_x, s->cu_y, 1 << s->cu_log2, s->qp_y);
  }
  PZ_STAMP(s, 6);
  int intra = s->cu_pred_mode == RBT_MODE_INTRA;
  int part = 0;
  if (s->cu_part_mode == RBT_PART_NxN && intra) part = ((y0 - s->cu_y) >= (1 << (s->cu_log2 - 1)) ? 2 : 0) + ((x0 - s->cu_x) >= (1 << (s->cu_log2 - 1)) ? 1 : 0);
  pz_fill_tu(s, x0, y0, N, cbf_luma);
  PZ_STAMP(s, 7);
  int chroma_here = log2 > 2 || blk == 3;
  const int ctb_mask = (1 << pzc_log2_ctb(s)) - 1;
  const uint32_t w0 = (uint32_t)RBT_CMD_TU | ((uint32_t)((x0 & ctb_mask) >> 2) << 8) | ((uint32_t)((y0 & ctb_mask) >> 2) << 16) | ((uint32_t)log2 << 24);
  int flags = (cbf_luma ? RBT_TU_CBF_Y : 0) | (intra ? RBT_TU_INTRA : 0) | (chroma_here ? RBT_TU_CHROMA : 0);
  if (chroma_here) flags |= (cbf_cb ? RBT_TU_CBF_CB : 0) | (cbf_cr ? RBT_TU_CBF_CR : 0);
  PZ_STAMP(s, 8);
  if (cbf_luma && pz_residual(s, 0, x0, y0, log2, pz_scan_idx(s->cu_pred_mode, log2, 0, pz_il(s, part)))) flags |= RBT_TU_TS_Y;
  PZ_STAMP(s, 9);
  if (chroma_here && !s->error) {
    int xc = (log2 > 2 ? x0 : xb) >> 1, yc = (log2 > 2 ? y0 : yb) >> 1, l2c = log2 > 2 ? log2 - 1 : 2;
    int sc = pz_scan_idx(s->cu_pred_mode, l2c, 1, s->intra_chroma);
    if (cbf_cb && pz_residual(s, 1, xc, yc, l2c, sc)) flags |= RBT_TU_TS_CB;
    if (cbf_cr && !s->error && pz_residual(s, 2, xc, yc, l2c, sc)) flags |= RBT_TU_TS_CR;
  }
  PZ_STAMP(s, 10);
  if (s->qp_key != s->qp_y) {
    s->qp_key = s->qp_y;
    s->qp_packed = ((s->qp_y + 6 * (pzc_bit_depth(s) - 8)) & 255) | ((pz_chroma_qp(s, 1) & 255) << 8) | ((pz_chroma_qp(s, 2) & 255) << 16);
  }
  pz_emit_words(s, w0, (uint32_t)(flags & 255) | ((uint32_t)(pz_il(s, part) & 255) << 8) | ((uint32_t)(s->intra_chroma & 255) << 16) | ((uint32_t)(s->cu_tq_bypass & 255) << 24), 0u, (uint32_t)s->qp_packed);
  PZ_STAMP(s, 11);
#ifdef RBT_PROFILE
  s->t_tu += __builtin_readcyclecounter() - ttu_;
#endif
}
RBT_DEV void pz_transform_tree(RbtParse* s, int x0, int y0, int xb0, int yb0, int log2, int depth0, int blk0, int pcb, int pcr) {
  // Depth-first walk without recursion and without a stack in memory (a private array would live in scratch, and every
  // scratch access is an HBM-latency round trip for this lone wave): the per-level child counter (4 bits) and cbf_cb /
  // cbf_cr flags (2 bits) are packed into two registers, node coordinates are updated incrementally.
  RbtCabacDec* c = &s->c;
  (void)xb0; (void)yb0; (void)depth0; (void)blk0;
  int lvl = 0, x = x0, y = y0, lg = log2;
  uint32_t states = 15u;                               // nibble lvl: 15 = not parsed yet, 0..3 = next child, 4 = done
  uint32_t flags = (uint32_t)((pcb ? 1 : 0) | (pcr ? 2 : 0));   // 2 bits per level: flags of the PARENT of the nodes at that level
  const int intra = s->cu_pred_mode == RBT_MODE_INTRA;
  const int intra_split = intra && s->cu_part_mode == RBT_PART_NxN;
  while (!s->error) {
    lvl = PZ_WU(lvl); x = PZ_WU(x); y = PZ_WU(y); lg = PZ_WU(lg); states = (uint32_t)PZ_WU(states); flags = (uint32_t)PZ_WU(flags);
    int st = (int)((states >> (4 * lvl)) & 15u);
    if (st == 15) {
      int inter_split = pzc_th_depth_inter(s) == 0 && !intra && s->cu_part_mode != RBT_PART_2Nx2N && lvl == 0;
      int split;
      if (lg <= pzc_log2_max_tb(s) && lg > pzc_log2_min_tb(s) && lvl < s->max_trafo_depth && !(intra_split && lvl == 0))
        split = rbt_cd_bin(c, CTX_SPLIT_TRANSFORM + 5 - lg);
      else split = (lg > pzc_log2_max_tb(s) || (intra_split && lvl == 0) || inter_split) ? 1 : 0;
      PZ_STAMP(s, 12);
      int ppcb = (int)((flags >> (2 * lvl)) & 1u), ppcr = (int)((flags >> (2 * lvl + 1)) & 1u);
      int cbf_cb = 0, cbf_cr = 0;
      if (lg > 2) {
        if (lvl == 0 || ppcb) cbf_cb = rbt_cd_bin(c, CTX_CBF_CHROMA + lvl);
        if (lvl == 0 || ppcr) cbf_cr = rbt_cd_bin(c, CTX_CBF_CHROMA + lvl);
      } else { cbf_cb = ppcb; cbf_cr = ppcr; }
      if (!split) {
        int cbf_luma = 1;
        if (intra || lvl != 0 || cbf_cb || cbf_cr) cbf_luma = rbt_cd_bin(c, CTX_CBF_LUMA + (lvl == 0 ? 1 : 0));
        PZ_STAMP(s, 13);
        int k = lvl ? (int)((states >> (4 * (lvl - 1))) & 15u) - 1 : 0, h = 1 << lg;
        int xb = lvl ? x - (k & 1) * h : x, yb = lvl ? y - (k >> 1) * h : y;
        pz_transform_unit(s, x, y, xb, yb, lg, k, cbf_luma, cbf_cb, cbf_cr);
        PZ_STAMP(s, 14);
        st = 4;
      } else {
        flags = (flags & ~(3u << (2 * (lvl + 1)))) | ((uint32_t)(cbf_cb | (cbf_cr << 1)) << (2 * (lvl + 1)));
        st = 0;
      }
    }
    if (st < 4) {                                      // descend into child st
      states = (states & ~(15u << (4 * lvl))) | ((uint32_t)(st + 1) << (4 * lvl));
      int h = 1 << (lg - 1);
      x += (st & 1) * h; y += (st >> 1) * h; lg--; lvl++;
      states = (states & ~(15u << (4 * lvl))) | (15u << (4 * lvl));
    } else {                                           // node complete: ascend
      if (lvl == 0) break;
      lvl--;
      int k = (int)((states >> (4 * lvl)) & 15u) - 1, h = 1 << lg;
      x -= (k & 1) * h; y -= (k >> 1) * h; lg++;
    }
  }
}

// ------------------------------------------------------------------------------------------------ motion (8.5.3.2)
RBT_DEV int pz_pu_avail(const RbtParse* s, int xn, int yn) { int loc = pz_loc(s, xn, yn); return loc >= 0 && (pz_ld_pm(s, loc) & RBT_PM_MODE_MASK) != RBT_MODE_INTRA; }
RBT_DEV RbtMv pz_mv_at(const RbtParse* s, int x, int y) { int loc = pz_loc(s, x, y), v = pz_ld_mv(s, loc); RbtMv m; m.x = (int16_t)(v & 0xFFFF); m.y = (int16_t)(v >> 16); m.ref = pz_ld_ref(s, loc); return m; }
RBT_DEV int pz_mv_same(RbtMv a, RbtMv b) { return a.x == b.x && a.y == b.y && a.ref == b.ref; }
RBT_DEV int pz_scale_mv(int mv, int tb, int td) {
  td = rbt_clip3(-128, 127, td); tb = rbt_clip3(-128, 127, tb);
  int tx = (16384 + (rbt_abs(td) >> 1)) / td;
  int dsf = rbt_clip3(-4096, 4095, (tb * tx + 32) >> 6);
  int p = dsf * mv;
  return rbt_clip3(-32768, 32767, (p < 0 ? -1 : 1) * ((rbt_abs(p) + 127) >> 8));
}
RBT_DEV int pz_temporal(const RbtParse* s, int xpb, int ypb, int w, int h, int ref_idx, RbtMv* out) {
  if (!pzs_temporal_mvp(s)) return 0;
  const RbtFrame* col = &s->frames[s->L->ref_frame[pzs_collocated_ref_idx(s)]];
  for (int k = 0; k < 2; k++) {
    int x = k ? xpb + (w >> 1) : xpb + w, y = k ? ypb + (h >> 1) : ypb + h;
    if (k == 0 && ((ypb >> pzc_log2_ctb(s)) != (y >> pzc_log2_ctb(s)) || x >= pzc_w(s) || y >= pzc_h(s))) continue;
    x = (x >> 4) << 4; y = (y >> 4) << 4;
    int i = (y >> 2) * pzc_w4(s) + (x >> 2);
    int cm = col->pm[i] & RBT_PM_MODE_MASK;
    if (cm == RBT_MODE_INTRA || cm == RBT_MODE_NONE) continue;
    int td = col->poc - col->refpoc[i], tb = pzs_poc(s) - s->L->ref_poc[ref_idx];
    int mx = col->mv[2 * i], my = col->mv[2 * i + 1];
    if (td != tb && td != 0) { mx = pz_scale_mv(mx, tb, td); my = pz_scale_mv(my, tb, td); }
    out->x = mx; out->y = my; out->ref = ref_idx;
    return 1;
  }
  return 0;
}
RBT_DEV RbtMv pz_merge(const RbtParse* s, int xpb, int ypb, int w, int h, int part_idx, int merge_idx) {
  int pm = s->cu_part_mode, maxc = pzs_max_merge_cand(s);
  // up to five candidates in named variables (a private array filled through a run-time index would live in scratch memory)
  RbtMv l0 = {0, 0, 0}, l1 = l0, l2 = l0, l3 = l0, l4 = l0; int n = 0;
#define PZ_PUSH(q) do { const RbtMv q_ = (q); if (n == 0) l0 = q_; else if (n == 1) l1 = q_; else if (n == 2) l2 = q_; else if (n == 3) l3 = q_; else if (n == 4) l4 = q_; n++; } while (0)
  RbtMv ca1 = {0, 0, 0}, cb1 = {0, 0, 0};
  int a1 = pz_pu_avail(s, xpb - 1, ypb + h - 1) && !((pm == RBT_PART_Nx2N || pm == RBT_PART_nLx2N || pm == RBT_PART_nRx2N) && part_idx == 1);
  if (a1) { ca1 = pz_mv_at(s, xpb - 1, ypb + h - 1); PZ_PUSH(ca1); }
  int b1 = pz_pu_avail(s, xpb + w - 1, ypb - 1) && !((pm == RBT_PART_2NxN || pm == RBT_PART_2NxnU || pm == RBT_PART_2NxnD) && part_idx == 1);
  int b1_in = 0, b0_in = 0, a0_in = 0;
  if (b1) { cb1 = pz_mv_at(s, xpb + w - 1, ypb - 1); if (!(a1 && pz_mv_same(ca1, cb1))) { PZ_PUSH(cb1); b1_in = 1; } }
  if (pz_pu_avail(s, xpb + w, ypb - 1)) { RbtMv q = pz_mv_at(s, xpb + w, ypb - 1); if (!(b1 && pz_mv_same(cb1, q))) { PZ_PUSH(q); b0_in = 1; } }
  if (pz_pu_avail(s, xpb - 1, ypb + h)) { RbtMv q = pz_mv_at(s, xpb - 1, ypb + h); if (!(a1 && pz_mv_same(ca1, q))) { PZ_PUSH(q); a0_in = 1; } }
  if (a1 + b1_in + b0_in + a0_in != 4 && pz_pu_avail(s, xpb - 1, ypb - 1)) {
    RbtMv q = pz_mv_at(s, xpb - 1, ypb - 1);
    if (!(a1 && pz_mv_same(ca1, q)) && !(b1 && pz_mv_same(cb1, q))) PZ_PUSH(q);
  }
  if (n > maxc) n = maxc;
  if (n < maxc) { RbtMv t; if (pz_temporal(s, xpb, ypb, w, h, 0, &t)) PZ_PUSH(t); }
  int zero_idx = 0;
  while (n < maxc) { RbtMv z = {0, 0, zero_idx < pzs_num_ref_idx(s) ? zero_idx : 0}; PZ_PUSH(z); zero_idx++; }
#undef PZ_PUSH
  RbtMv r = merge_idx == 0 ? l0 : (merge_idx == 1 ? l1 : (merge_idx == 2 ? l2 : (merge_idx == 3 ? l3 : l4)));
  return r;
}
RBT_DEV RbtMv pz_amvp(const RbtParse* s, int xpb, int ypb, int w, int h, int ref_idx, int mvp_flag) {
  int tgt = s->L->ref_poc[ref_idx], cur = pzs_poc(s);
  // spatial candidates A0, A1 (left: below-left, left) and B0, B1, B2 (above: above-right, above, above-left); positions by arithmetic and availability
  // as bit masks - small private arrays indexed by a run-time k would live in scratch memory
#define PZ_XA(k) (xpb - 1)
#define PZ_YA(k) (ypb + h - (k))
#define PZ_XB(k) ((k) == 0 ? xpb + w : ((k) == 1 ? xpb + w - 1 : xpb - 1))
#define PZ_YB(k) (ypb - 1)
  int ava = 0, avb = 0;
  for (int k = 0; k < 2; k++) ava |= (pz_pu_avail(s, PZ_XA(k), PZ_YA(k)) ? 1 : 0) << k;
  for (int k = 0; k < 3; k++) avb |= (pz_pu_avail(s, PZ_XB(k), PZ_YB(k)) ? 1 : 0) << k;
  int fa = 0, fb = 0; RbtMv ma = {0, 0, 0}, mb = {0, 0, 0};
  for (int k = 0; k < 2 && !fa; k++) if ((ava >> k) & 1) { RbtMv q = pz_mv_at(s, PZ_XA(k), PZ_YA(k)); if (s->L->ref_poc[q.ref] == tgt) { ma = q; fa = 1; } }
  for (int k = 0; k < 2 && !fa; k++) if ((ava >> k) & 1) {
    RbtMv q = pz_mv_at(s, PZ_XA(k), PZ_YA(k)); int td = cur - s->L->ref_poc[q.ref], tb = cur - tgt;
    ma = q; fa = 1; if (td != tb && td != 0) { ma.x = pz_scale_mv(q.x, tb, td); ma.y = pz_scale_mv(q.y, tb, td); }
  }
  int is_scaled = ava != 0;
  for (int k = 0; k < 3 && !fb; k++) if ((avb >> k) & 1) { RbtMv q = pz_mv_at(s, PZ_XB(k), PZ_YB(k)); if (s->L->ref_poc[q.ref] == tgt) { mb = q; fb = 1; } }
  if (!is_scaled && fb) { ma = mb; fa = 1; }
  if (!is_scaled) {
    fb = 0;
    for (int k = 0; k < 3 && !fb; k++) if ((avb >> k) & 1) {
      RbtMv q = pz_mv_at(s, PZ_XB(k), PZ_YB(k)); int td = cur - s->L->ref_poc[q.ref], tb = cur - tgt;
      mb = q; fb = 1; if (td != tb && td != 0) { mb.x = pz_scale_mv(q.x, tb, td); mb.y = pz_scale_mv(q.y, tb, td); }
    }
  }
#undef PZ_XA
#undef PZ_YA
#undef PZ_XB
#undef PZ_YB
  RbtMv l0 = {0, 0, ref_idx}, l1 = {0, 0, ref_idx}; int n = 0;
  if (fa) { l0 = ma; n = 1; }
  if (fb && !(fa && ma.x == mb.x && ma.y == mb.y)) { if (n == 0) l0 = mb; else l1 = mb; n++; }
  if (n < 2) { RbtMv t; if (pz_temporal(s, xpb, ypb, w, h, ref_idx, &t)) { if (n == 0) l0 = t; else l1 = t; n++; } }
  RbtMv r = mvp_flag ? l1 : l0; r.ref = ref_idx;
  return r;
}
RBT_DEV int pz_mvd_comp(RbtCabacDec* c, int gt0, int gt1) {
  if (!gt0) return 0;
  int v = 1;
  if (gt1) { int k = 1; v = 2; while (k < 24 && rbt_cd_bypass(c)) { v += 1 << k; k++; } v += (int)rbt_cd_bypass_n(c, k); }
  return rbt_cd_bypass(c) ? -v : v;
}
RBT_DEV void pz_prediction_unit(RbtParse* s, int x0, int y0, int w, int h, int part_idx, int skip) {
  RbtCabacDec* c = &s->c;
  x0 = PZ_WU(x0); y0 = PZ_WU(y0); w = PZ_WU(w); h = PZ_WU(h); part_idx = PZ_WU(part_idx); skip = PZ_WU(skip);
  RbtMv mv;
  int merge = skip ? 1 : rbt_cd_bin(c, CTX_MERGE_FLAG);
  s->last_pu_merge = merge;
  if (merge) {
    int idx = 0;
    if (pzs_max_merge_cand(s) > 1) { idx = rbt_cd_bin(c, CTX_MERGE_IDX); if (idx) while (idx < pzs_max_merge_cand(s) - 1 && rbt_cd_bypass(c)) idx++; }
    mv = pz_merge(s, x0, y0, w, h, part_idx, idx);
  } else {
    int ref_idx = 0;
    if (pzs_num_ref_idx(s) > 1) {
      int mx = pzs_num_ref_idx(s) - 1;
      while (ref_idx < mx) { int b = ref_idx < 2 ? rbt_cd_bin(c, CTX_REF_IDX + ref_idx) : rbt_cd_bypass(c); if (!b) break; ref_idx++; }
    }
    int gx0 = rbt_cd_bin(c, CTX_MVD_GT0), gy0 = rbt_cd_bin(c, CTX_MVD_GT0);
    int gx1 = gx0 ? rbt_cd_bin(c, CTX_MVD_GT1) : 0, gy1 = gy0 ? rbt_cd_bin(c, CTX_MVD_GT1) : 0;
    int dx = pz_mvd_comp(c, gx0, gx1), dy = pz_mvd_comp(c, gy0, gy1);
    int mvp = rbt_cd_bin(c, CTX_MVP_FLAG);
    mv = pz_amvp(s, x0, y0, w, h, ref_idx, mvp);
    mv.x = (int16_t)(mv.x + dx); mv.y = (int16_t)(mv.y + dy);
  }
  if (mv.ref < 0 || mv.ref >= pzs_num_ref_idx(s)) { s->error = 5; return; }
  pz_fill_pu(s, x0, y0, w, h, skip ? RBT_MODE_SKIP : RBT_MODE_INTER, mv.ref, ((uint32_t)(uint16_t)mv.y << 16) | (uint16_t)mv.x);
  RbtCmd cmd; cmd.type = RBT_CMD_PU; cmd.x4 = (uint8_t)((x0 & ((1 << pzc_log2_ctb(s)) - 1)) >> 2); cmd.y4 = (uint8_t)((y0 & ((1 << pzc_log2_ctb(s)) - 1)) >> 2);
  cmd.log2 = 0; cmd.a = (uint8_t)(w >> 2); cmd.b = (uint8_t)(h >> 2); cmd.c = (uint8_t)mv.ref; cmd.d = 0; cmd.mvx = (int16_t)mv.x; cmd.mvy = (int16_t)mv.y;
  cmd.qp[0] = cmd.qp[1] = cmd.qp[2] = 0; cmd.pad = 0;
  pz_emit(s, cmd);
}

// ------------------------------------------------------------------------------------------------ coding unit (7.3.8.5)
// known_left / known_above >= 0: the neighbour is a prediction unit of the SAME coding unit (NxN: units 1 and 3 have unit 0 / 2 to their left, units 2 and 3 have
// unit 0 / 1 above) whose mode was derived a moment ago - no neighbour lookup (two lane reads and their VALU -> SALU hand-over each)
RBT_DEV void pz_intra_mpm(const RbtParse* s, int xp, int yp, int cand[3], int known_left = -1, int known_above = -1) {
  int ca = 1, cb = 1;
  if (known_left >= 0) ca = known_left;
  else { int nl = pz_nb(s, xp - 1, yp); if (nl >= 0 && (nl & RBT_PM_MODE_MASK) == RBT_MODE_INTRA) ca = (nl >> 8) & 63; }
  if (known_above >= 0) cb = known_above;
  else if (((yp - 1) >> pzc_log2_ctb(s)) == (yp >> pzc_log2_ctb(s))) { int na = pz_nb(s, xp, yp - 1); if (na >= 0 && (na & RBT_PM_MODE_MASK) == RBT_MODE_INTRA) cb = (na >> 8) & 63; }
  if (ca == cb) {
    if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
    else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
  } else { cand[0] = ca; cand[1] = cb; cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26); }
}
RBT_DEV void pz_coding_unit(RbtParse* s, int x0, int y0, int log2, int depth) {
  RbtCabacDec* c = &s->c;
#ifdef RBT_PROFILE
  unsigned long long tcu_ = __builtin_readcyclecounter(); s->n_cu++;
#endif
  x0 = PZ_WU(x0); y0 = PZ_WU(y0); log2 = PZ_WU(log2); depth = PZ_WU(depth);
#ifdef RBT_PROFILE
  s->t_last = __builtin_readcyclecounter();
#endif
  PZ_STAMP(s, 15);
  int N = 1 << log2;
  s->cu_x = x0; s->cu_y = y0; s->cu_log2 = log2; s->cu_tq_bypass = 0; s->cu_part_mode = RBT_PART_2Nx2N; s->cu_pred_mode = RBT_MODE_INTRA;
  if (pzc_cu_qp_delta(s)) s->qp_y = pz_wrap_qp(s, s->qp_pred + s->cu_qp_delta_val);
  if (pzc_tq_bypass_enabled(s)) s->cu_tq_bypass = rbt_cd_bin(c, CTX_CU_TQ_BYPASS);
  int skip = 0;
  if (pzs_slice_type(s) != RBT_SLICE_I) {
    int nl = pz_nb(s, x0 - 1, y0), na = pz_nb(s, x0, y0 - 1);
    int cl = nl >= 0 && (nl & RBT_PM_MODE_MASK) == RBT_MODE_SKIP, ca = na >= 0 && (na & RBT_PM_MODE_MASK) == RBT_MODE_SKIP;
    skip = rbt_cd_bin(c, CTX_CU_SKIP + cl + ca);
  }
  if (skip) {
    s->cu_pred_mode = RBT_MODE_SKIP;
    pz_fill_cu(s, x0, y0, N, RBT_MODE_NONE | (s->cu_tq_bypass ? RBT_PM_TQ_BYPASS : 0), (depth << 6) | 1, s->qp_y);
    RBT_SYNC_LDS();
    pz_prediction_unit(s, x0, y0, N, N, 0, 1);
    RBT_SYNC_LDS();
    return;
  }
  if (pzs_slice_type(s) != RBT_SLICE_I) s->cu_pred_mode = rbt_cd_bin(c, CTX_PRED_MODE) ? RBT_MODE_INTRA : RBT_MODE_INTER;
  if (s->cu_pred_mode == RBT_MODE_INTRA) {
    if (log2 == pzc_log2_min_cb(s)) s->cu_part_mode = rbt_cd_bin(c, CTX_PART_MODE) ? RBT_PART_2Nx2N : RBT_PART_NxN;
    if (s->cu_part_mode == RBT_PART_NxN && log2 == 3 && pzc_log2_min_tb(s) > 2) { s->error = 6; return; }
  } else {
    if (rbt_cd_bin(c, CTX_PART_MODE)) s->cu_part_mode = RBT_PART_2Nx2N;
    else if (log2 == pzc_log2_min_cb(s)) {
      if (log2 == 3) s->cu_part_mode = rbt_cd_bin(c, CTX_PART_MODE + 1) ? RBT_PART_2NxN : RBT_PART_Nx2N;
      else if (rbt_cd_bin(c, CTX_PART_MODE + 1)) s->cu_part_mode = RBT_PART_2NxN;
      else s->cu_part_mode = rbt_cd_bin(c, CTX_PART_MODE + 2) ? RBT_PART_Nx2N : RBT_PART_NxN;
    } else if (!pzc_amp(s)) s->cu_part_mode = rbt_cd_bin(c, CTX_PART_MODE + 1) ? RBT_PART_2NxN : RBT_PART_Nx2N;
    else {
      int hor = rbt_cd_bin(c, CTX_PART_MODE + 1);
      if (rbt_cd_bin(c, CTX_PART_MODE + 3)) s->cu_part_mode = hor ? RBT_PART_2NxN : RBT_PART_Nx2N;
      else { int b = rbt_cd_bypass(c); s->cu_part_mode = hor ? (b ? RBT_PART_2NxnD : RBT_PART_2NxnU) : (b ? RBT_PART_nRx2N : RBT_PART_nLx2N); }
    }
  }
  PZ_STAMP(s, 16);
  if (s->cu_pred_mode == RBT_MODE_INTRA) {
    pz_fill_cu(s, x0, y0, N, RBT_MODE_INTRA | (s->cu_tq_bypass ? RBT_PM_TQ_BYPASS : 0), (depth << 6) | 1, s->qp_y);
    RBT_SYNC_LDS();
    PZ_STAMP(s, 17);
    int np = s->cu_part_mode == RBT_PART_NxN ? 4 : 1, pb = N >> (np == 4);
    // per prediction unit: prev_intra_luma_pred_flag, then mpm_idx or rem_intra_luma_pred_mode - packed into two words (private arrays indexed by a
    // run-time i live in scratch memory: two round trips to it per CU on the latency chain of the slice)
    uint32_t prev_bits = 0, val_bytes = 0;
    for (int i = 0; i < np; i++) prev_bits |= (uint32_t)rbt_cd_bin(c, CTX_PREV_INTRA_LUMA) << i;
    for (int i = 0; i < np; i++) {
      int v;
      if ((prev_bits >> i) & 1) { v = rbt_cd_bypass(c); if (v) v += rbt_cd_bypass(c); }
      else v = (int)rbt_cd_bypass_n(c, 5);
      val_bytes |= (uint32_t)v << (8 * i);
    }
    PZ_STAMP(s, 18);
    for (int i = 0; i < np; i++) {
      int xp = x0 + (i & 1) * pb, yp = y0 + (i >> 1) * pb;
#ifdef RBT_PROFILE
      unsigned long long tm_ = __builtin_readcyclecounter();
#endif
      int cand[3]; pz_intra_mpm(s, xp, yp, cand, (np == 4 && (i & 1)) ? pz_il(s, i - 1) : -1, (np == 4 && (i & 2)) ? pz_il(s, i - 2) : -1);
#ifdef RBT_PROFILE
      s->t_mpm += __builtin_readcyclecounter() - tm_;
#endif
      int mode; const int pv = (int)((prev_bits >> i) & 1u), val = (int)((val_bytes >> (8 * i)) & 255u);
      if (pv) mode = val == 0 ? cand[0] : (val == 1 ? cand[1] : cand[2]);
      else {
        if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
        if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
        if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
        mode = val;
        if (mode >= cand[0]) mode++;
        if (mode >= cand[1]) mode++;
        if (mode >= cand[2]) mode++;
      }
      pz_set_il(s, i, mode);
      pz_fill_dm(s, xp, yp, pb, (depth << 6) | mode);
      RBT_SYNC_LDS();
    }
    PZ_STAMP(s, 19);
    int icp = 4;
    if (rbt_cd_bin(c, CTX_INTRA_CHROMA)) icp = (int)rbt_cd_bypass_n(c, 2);
    int cmode = icp == 0 ? 0 : (icp == 1 ? 26 : (icp == 2 ? 10 : 1));
    if (icp == 4) s->intra_chroma = pz_il(s, 0);
    else s->intra_chroma = cmode == pz_il(s, 0) ? 34 : cmode;
  } else {
    pz_fill_cu(s, x0, y0, N, RBT_MODE_NONE | (s->cu_tq_bypass ? RBT_PM_TQ_BYPASS : 0), (depth << 6) | 1, s->qp_y);
    RBT_SYNC_LDS();
    int h2 = N >> 1, q = N >> 2, pmode = s->cu_part_mode;
    int np = pmode == RBT_PART_2Nx2N ? 1 : (pmode == RBT_PART_NxN ? 4 : 2);
    for (int i = 0; i < np && !s->error; i++) {
      int px = 0, py = 0, pw = N, ph = N;
      switch (pmode) {
        case RBT_PART_2NxN: ph = h2; py = i * h2; break;
        case RBT_PART_Nx2N: pw = h2; px = i * h2; break;
        case RBT_PART_NxN: pw = ph = h2; px = (i & 1) * h2; py = (i >> 1) * h2; break;
        case RBT_PART_2NxnU: ph = i ? N - q : q; py = i ? q : 0; break;
        case RBT_PART_2NxnD: ph = i ? q : N - q; py = i ? N - q : 0; break;
        case RBT_PART_nLx2N: pw = i ? N - q : q; px = i ? q : 0; break;
        case RBT_PART_nRx2N: pw = i ? q : N - q; px = i ? N - q : 0; break;
        default: break;
      }
      pz_prediction_unit(s, x0 + px, y0 + py, pw, ph, i, 0);
      RBT_SYNC_LDS();
    }
    if (s->error) return;
  }
#ifdef RBT_PROFILE
  s->t_hdr += __builtin_readcyclecounter() - tcu_;
#endif
#ifdef RBT_PROFILE
#define PZ_CU_END() (s->t_cu += __builtin_readcyclecounter() - tcu_)
#else
#define PZ_CU_END() ((void)0)
#endif
  PZ_STAMP(s, 20);
  int rqt_root_cbf = 1;
  if (s->cu_pred_mode != RBT_MODE_INTRA && !(s->cu_part_mode == RBT_PART_2Nx2N && s->last_pu_merge)) rqt_root_cbf = rbt_cd_bin(c, CTX_RQT_ROOT_CBF);
  if (rqt_root_cbf) {
    s->max_trafo_depth = s->cu_pred_mode == RBT_MODE_INTRA ? pzc_th_depth_intra(s) + (s->cu_part_mode == RBT_PART_NxN) : pzc_th_depth_inter(s);
    pz_transform_tree(s, x0, y0, x0, y0, log2, 0, 0, 0, 0);
  }
  RBT_SYNC_LDS();
  PZ_STAMP(s, 21);
  PZ_CU_END();
}

// ------------------------------------------------------------------------------------------------ coding quadtree + slice data
RBT_DEV void pz_coding_quadtree(RbtParse* s, int x0, int y0, int log2) {
  // same stack-free walk as pz_transform_tree; children outside the picture are skipped (7.3.8.4)
  int lvl = 0, x = x0, y = y0, lg = log2;
  uint32_t states = 15u;
  while (!s->error) {
    lvl = PZ_WU(lvl); x = PZ_WU(x); y = PZ_WU(y); lg = PZ_WU(lg); states = (uint32_t)PZ_WU(states);
    int st = (int)((states >> (4 * lvl)) & 15u);
    int N = 1 << lg;
    PZ_STAMP(s, 22);
    if (st == 15) {
      int split;
      if (x + N <= pzc_w(s) && y + N <= pzc_h(s) && lg > pzc_log2_min_cb(s)) {
        int nl = pz_nb(s, x - 1, y), na = pz_nb(s, x, y - 1);
        int cl = nl >= 0 && (nl >> 14) > lvl, ca = na >= 0 && (na >> 14) > lvl;
        split = rbt_cd_bin(&s->c, CTX_SPLIT_CU + cl + ca);
      } else split = lg > pzc_log2_min_cb(s);
      PZ_STAMP(s, 23);
      if (pzc_cu_qp_delta(s) && lg >= pzc_log2_ctb(s) - pzc_diff_cu_qp_delta_depth(s)) pz_start_qg(s, x, y);
      if (!split) { pz_coding_unit(s, x, y, lg, lvl); st = 4;
#ifdef RBT_PROFILE
        s->t_last = __builtin_readcyclecounter();
#endif
      } else st = 0;
    }
    // next child that lies inside the picture
    int h = N >> 1;
    while (st < 4 && (x + (st & 1) * h >= pzc_w(s) || y + (st >> 1) * h >= pzc_h(s))) st++;
    if (st < 4) {
      states = (states & ~(15u << (4 * lvl))) | ((uint32_t)(st + 1) << (4 * lvl));
      x += (st & 1) * h; y += (st >> 1) * h; lg--; lvl++;
      states = (states & ~(15u << (4 * lvl))) | (15u << (4 * lvl));
    } else {
      if (lvl == 0) break;
      lvl--;
      int k = (int)((states >> (4 * lvl)) & 15u) - 1, hh = 1 << lg;
      x -= (k & 1) * hh; y -= (k >> 1) * hh; lg++;
    }
    PZ_STAMP(s, 24);
  }
}

// HBM image of a suspended slice parser. The parser can stop in front of any CTB row (`row_limit`) and continue in a later
// launch: the reconstruction of the rows that are complete then runs underneath the parsing of the rest of the picture.
struct RbtParseSave {
  uint32_t phase;                 // 0 not started, 1 suspended, 2 finished
  int32_t sc[24];                 // scalar parser / engine state
  uint32_t buf_lo, buf_hi;
  uint32_t ctx[4][64];            // context variables (one word per lane and register)
  uint32_t lds[(RBT_PARSE_LDS_BYTES(RBT_PARSE_CAP4_L) + 3) / 4];
};
// Entry: parses one slice segment (save == nullptr: in one go; else up to CTB row `row_limit`, resuming where it stopped).
// `lds` holds RBT_PARSE_LDS_BYTES(cap4) bytes; the launcher picks cap4 >= the width of every picture of the launch / 4.
RBT_DEV void rbt_parse_slice(RbtFrame* frames, RbtSlice* slices, int slice_idx, const uint8_t* rbsp, RBT_LDS_AS RbtParseLds* lds, int cap4, RbtParseSave* save, int row_limit) {
  RbtParse s;
  RbtParseSave* sv = save ? save + slice_idx : nullptr;   // (own index: slice_idx is re-pointed at the slice's head further down)
  const int phase = sv ? RBT_UNI((int)sv->phase) : 0;
  if (phase == 2) return;
  const RbtSlice* gs = &slices[slice_idx];
  if (RBT_UNI((int)gs->dependent) && !RBT_UNI((int)gs->row_task)) return;   // parsed by the wave of the segment before it (next_seg chain)
  const int wpp = RBT_UNI((int)gs->wpp);
  int seg = slice_idx;                                       // the slice segment being read
  const int own_idx = slice_idx;                             // where this wave reports (n_ctbs_decoded, resume state)
  slice_idx = RBT_UNI(gs->head);                             // the SLICE this wave's CTBs belong to: availability, ctb_slice
  // row task: this wave starts at a CTB row of a wavefront stream whose upper neighbour another wave parses
  int import_row = -1;                                       // set before the CTB loop
  int ctb_limit = RBT_UNI(gs->ctb_limit); uint32_t seg_first = 0;   // of the segment being read: CTBs it may hold (0: to its end), value of `count` at its start
  int seg_end = RBT_UNI(gs->end_addr);                       // ... and the CTB address it has to end at (RbtSlice::end_addr)
  uint32_t seen_above = 0;
  s.L = lds; s.left_ok = 0; s.corner_ok = 0; s.corner_pm = s.corner_dm = s.corner_ref = s.corner_mv = 0; s.ctb_x = s.ctb_y = 0;
  if (phase == 0) { RBT_LDS_AS uint16_t* a_slice = pz_above_slice(lds, cap4); RBT_PAR_FOR(i, cap4 / 4) a_slice[i] = 0xFFFF; }
  else { RBT_LDS_AS uint32_t* lw = (RBT_LDS_AS uint32_t*)lds; RBT_PAR_FOR(i, (int)(RBT_PARSE_LDS_BYTES(cap4) / 4)) lw[i] = sv->lds[i]; }
  RBT_SYNC();
  if (RBT_LANE0) lds->cap4 = cap4;
  s.frames = frames; s.f = &frames[RBT_UNI(gs->frame)]; s.slice_idx = slice_idx; s.error = 0;
  s.s_bits = (uint32_t)RBT_UNI((gs->slice_type & 3) | ((gs->sao_luma & 1) << 2) | ((gs->sao_chroma & 1) << 3) | ((gs->temporal_mvp & 1) << 4) | ((gs->cabac_init_flag & 1) << 5) |
                               ((gs->max_merge_cand & 7) << 6) | ((gs->num_ref_idx & 31) << 9) | ((gs->collocated_ref_idx & 15) << 14));
  s.s_qp = (uint32_t)RBT_UNI((uint32_t)(uint8_t)gs->qp | ((uint32_t)(uint8_t)gs->cb_qp_offset << 8) | ((uint32_t)(uint8_t)gs->cr_qp_offset << 16));
  s.s_poc = RBT_UNI(gs->poc);
  RBT_PAR_FOR(i, RBT_MAX_REFS) { lds->ref_poc[i] = gs->ref_poc[i]; lds->ref_frame[i] = gs->ref_frame[i]; }
  RBT_PAR_FOR(i, 3 * 4 * 64) lds->scan[i / 256][(i / 64) & 3][i & 63] = k_scan[i / 256][(i / 64) & 3][i & 63];
  { const RbtFrame* f = s.f; const RbtStreamCfg* g = &f->cfg;
    s.m_cmds = rbt_uni_ptr(f->cmds); s.m_coef0 = rbt_uni_ptr(f->coef[0]); s.m_coef1 = rbt_uni_ptr(f->coef[1]); s.m_coef2 = rbt_uni_ptr(f->coef[2]); s.m_cap = RBT_UNI(f->cmd_cap);
    s.c_dim = (uint32_t)RBT_UNI((uint32_t)g->w | ((uint32_t)g->h << 16));
    s.c_logs = (uint32_t)RBT_UNI((g->bit_depth & 15) | ((g->log2_ctb & 7) << 4) | ((g->log2_min_cb & 7) << 7) | ((g->log2_min_tb & 7) << 10) | ((g->log2_max_tb & 7) << 13) |
                                 ((g->th_depth_inter & 7) << 16) | ((g->th_depth_intra & 7) << 19) | ((g->diff_cu_qp_delta_depth & 3) << 22));
    s.c_flags = (uint32_t)RBT_UNI((uint32_t)(g->amp & 1) | ((uint32_t)(g->sao & 1) << 1) | ((uint32_t)(g->strong_intra & 1) << 2) | ((uint32_t)(g->tmvp & 1) << 3) | ((uint32_t)(g->sign_hiding & 1) << 4) |
                                  ((uint32_t)(g->cabac_init_present & 1) << 5) | ((uint32_t)(g->cip & 1) << 6) | ((uint32_t)(g->transform_skip & 1) << 7) | ((uint32_t)(g->cu_qp_delta & 1) << 8) |
                                  ((uint32_t)(g->tq_bypass_enabled & 1) << 9) | ((uint32_t)(uint8_t)g->cb_qp_offset << 16) | ((uint32_t)(uint8_t)g->cr_qp_offset << 24)); }
  int init_type = pzs_slice_type(&s) == RBT_SLICE_I ? 0 : (pzs_cabac_init_flag(&s) ? 2 : 1);
#ifdef RBT_PROFILE
  unsigned long long t_all_ = __builtin_readcyclecounter(); s.t_res = s.t_ctb = s.t_cu = s.t_a = s.t_b = s.t_c = s.t_d = s.t_tu = s.t_hdr = s.t_fill = s.t_mpm = 0; s.n_res = s.n_cu = 0; s.c.n_bins = s.c.n_byp = 0;
#endif
#ifdef RBT_PROFILE
  RBT_PAR_FOR(i, 32) { lds->prof[i] = 0; lds->profn[i] = 0; }
  s.t_last = __builtin_readcyclecounter();
#endif
  int n_ctb = pzc_w_ctb(&s) * pzc_h_ctb(&s), end = 0, addr = RBT_UNI(gs->ctb_addr);
  if (RBT_UNI((int)gs->row_task)) import_row = addr / pzc_w_ctb(&s);
  // a wave that starts in the middle of a CTB row (a slice that begins there) shares the row's progress counter with the wave of the row's first part:
  // it adds to the counter only once that part is complete, so the counter always means "the row is parsed up to here"
  int prefix_row = wpp && addr % pzc_w_ctb(&s) ? addr / pzc_w_ctb(&s) : -1; const int prefix_need = addr % pzc_w_ctb(&s);
  uint32_t count = 0;
  s.qp_key = 0x7FFFFFFF; s.qp_packed = 0;
  if (phase != 0) { seg = RBT_UNI(sv->sc[22]); seg_end = RBT_UNI(slices[seg].end_addr); }
  rbt_cd_start(&s.c, rbsp + (uint32_t)RBT_UNI(slices[seg].data_off), (uint32_t)RBT_UNI(slices[seg].data_size));
  if (phase == 0) {
    rbt_ctx_init(&s.c.cs, init_type, pzs_qp(&s));
    s.qp_y = pzs_qp(&s); s.qp_pred = pzs_qp(&s); s.qp_y_prev = pzs_qp(&s); s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0;
    s.last_pu_merge = 0; s.max_trafo_depth = 0; s.intra_chroma = 1; s.il_packed = 0x01010101;
  } else {
    // resume: scalar state, bit reservoir (the word cursor is re-based on the same aligned pointer), context registers
    const int32_t* q = sv->sc;
    s.left_ok = RBT_UNI(q[0]); s.corner_ok = RBT_UNI(q[1]); s.corner_pm = RBT_UNI(q[2]); s.corner_dm = RBT_UNI(q[3]); s.corner_ref = RBT_UNI(q[4]); s.corner_mv = RBT_UNI(q[5]);
    s.qp_y = RBT_UNI(q[6]); s.qp_pred = RBT_UNI(q[7]); s.qp_y_prev = RBT_UNI(q[8]); s.is_cu_qp_delta_coded = RBT_UNI(q[9]); s.cu_qp_delta_val = RBT_UNI(q[10]);
    s.il_packed = RBT_UNI(q[11]); s.intra_chroma = RBT_UNI(q[12]); s.max_trafo_depth = RBT_UNI(q[13]); s.last_pu_merge = RBT_UNI(q[14]);
    addr = RBT_UNI(q[15]); count = (uint32_t)RBT_UNI(q[16]);
    s.c.widx = (uint32_t)RBT_UNI(q[17]); s.c.nbuf = RBT_UNI(q[18]); s.c.range = (uint32_t)RBT_UNI(q[19]); s.c.value = (uint32_t)RBT_UNI(q[20]); s.c.avail = RBT_UNI(q[21]);
    s.c.buf = ((uint64_t)(uint32_t)RBT_UNI(sv->buf_hi) << 32) | (uint32_t)RBT_UNI(sv->buf_lo);
    s.c.next_raw = s.c.widx < s.c.n_words ? s.c.w[s.c.widx] : 0;
#ifdef RBT_HOSTEMU
    for (int i = 0; i < RBT_CTX_COUNT; i++) s.c.cs.st[i] = (uint8_t)sv->ctx[i >> 6][i & 63];
#else
    { const int ln = (int)threadIdx.x & 63; s.c.cs.st0 = (int)sv->ctx[0][ln]; s.c.cs.st1 = (int)sv->ctx[1][ln]; s.c.cs.st2 = (int)sv->ctx[2][ln]; s.c.cs.st3 = (int)sv->ctx[3][ln]; }
    rbt_ctx_tables(&s.c.cs);
#endif
  }
  while (!end) {
    if (addr >= n_ctb) { s.error = 1; break; }
    addr = RBT_UNI(addr);
    int rx = RBT_UNI(addr % pzc_w_ctb(&s)), ry = RBT_UNI(addr / pzc_w_ctb(&s));
    if (sv && ry >= row_limit) {
      // suspend in front of this CTB
      RBT_SYNC_LDS();
      { RBT_LDS_AS uint32_t* lw = (RBT_LDS_AS uint32_t*)lds; RBT_PAR_FOR(i, (int)(RBT_PARSE_LDS_BYTES(cap4) / 4)) sv->lds[i] = lw[i]; }
#ifdef RBT_HOSTEMU
      for (int i = 0; i < RBT_CTX_COUNT; i++) sv->ctx[i >> 6][i & 63] = s.c.cs.st[i];
#else
      { const int ln = (int)threadIdx.x & 63; sv->ctx[0][ln] = (uint32_t)s.c.cs.st0; sv->ctx[1][ln] = (uint32_t)s.c.cs.st1; sv->ctx[2][ln] = (uint32_t)s.c.cs.st2; sv->ctx[3][ln] = (uint32_t)s.c.cs.st3; }
#endif
      if (RBT_LANE0) {
        int32_t* q = sv->sc;
        q[0] = s.left_ok; q[1] = s.corner_ok; q[2] = s.corner_pm; q[3] = s.corner_dm; q[4] = s.corner_ref; q[5] = s.corner_mv;
        q[6] = s.qp_y; q[7] = s.qp_pred; q[8] = s.qp_y_prev; q[9] = s.is_cu_qp_delta_coded; q[10] = s.cu_qp_delta_val;
        q[11] = s.il_packed; q[12] = s.intra_chroma; q[13] = s.max_trafo_depth; q[14] = s.last_pu_merge;
        q[15] = addr; q[16] = (int32_t)count; q[22] = seg;
        q[17] = (int32_t)s.c.widx; q[18] = s.c.nbuf; q[19] = (int32_t)s.c.range; q[20] = (int32_t)s.c.value; q[21] = s.c.avail;
        sv->buf_lo = (uint32_t)s.c.buf; sv->buf_hi = (uint32_t)(s.c.buf >> 32);
        sv->phase = 1;
      }
      return;
    }
    if (RBT_LANE0) s.f->ctb_slice[addr] = (uint16_t)slice_idx;
    s.ctb_addr = addr; s.n_cmds = 0;
    if (rx == 0) { s.left_ok = 0; s.corner_ok = 0; }
    if (import_row >= 0 && ry != import_row) import_row = -1;           // further rows of this wave have their upper neighbour in its own line buffers
    if (import_row == ry) {
      // wait until the wave of the row above has parsed the CTB above-right (the last one of the row for the last column), then take over what this CTB
      // needs of that row: its own column at the start of the row, the next column always
      const int wc = pzc_w_ctb(&s);
      seen_above = rbt_flag_wait_seen(&s.f->prow_done[ry - 1], (uint32_t)(rx + 2 < wc ? rx + 2 : wc), seen_above, &s.f->error);
      // the wait ran out, or the picture went bad meanwhile (the wave of the row above gave up and released its rows): what it would hand over is not there
      { const int32_t pe = rbt_err_peek(&s.f->error); if (pe) { s.error = pe; break; } }
      if (rx == 0) pz_import_above(&s, 0, ry);
      pz_import_above(&s, rx + 1, ry);
    }
    if (wpp && rx == 0) {
      // first CTB of a row of a wavefront stream (9.3.1): the context variables of the CTB above-right after it was parsed when that CTB is
      // available (inside the picture, same slice), the initial ones otherwise; QpY prediction restarts from SliceQpY (8.6.1)
      RBT_SYNC_LDS();
      const int tr_ok = ry > 0 && pzc_w_ctb(&s) > 1 && RBT_UNI((int)pz_above_slice(lds, cap4)[1]) == slice_idx;
      if (!tr_ok) rbt_ctx_init(&s.c.cs, init_type, pzs_qp(&s));
      else if (import_row == ry) rbt_ctx_load_g(&s.c.cs, s.f->prow_ctx + (size_t)(ry - 1) * 256);
      else rbt_ctx_load(&s.c.cs, lds->wpp_ctx);
      s.qp_y = pzs_qp(&s);
    }
#ifdef RBT_PROFILE
    unsigned long long tb_ = __builtin_readcyclecounter();
#endif
    pz_begin_ctb(&s, rx, ry);
#ifdef RBT_PROFILE
    s.t_ctb += __builtin_readcyclecounter() - tb_;
#endif
    pz_sao(&s, rx, ry);
    pz_coding_quadtree(&s, rx << pzc_log2_ctb(&s), ry << pzc_log2_ctb(&s), pzc_log2_ctb(&s));
    if (RBT_LANE0) s.f->cmd_count[addr] = s.n_cmds;
    if (s.error) break;
#ifdef RBT_PROFILE
    unsigned long long te_ = __builtin_readcyclecounter();
#endif
    pz_end_ctb(&s, rx, ry);
#ifdef RBT_PROFILE
    s.t_ctb += __builtin_readcyclecounter() - te_;
#endif
    if (wpp && rx == 1) { rbt_ctx_store(&s.c.cs, lds->wpp_ctx); rbt_ctx_store_g(&s.c.cs, s.f->prow_ctx + (size_t)ry * 256); }   // storage process after the second CTB of a row
    if (wpp) {
      pz_export_row(&s, rx, ry);
      if (prefix_row == ry) { rbt_flag_wait(&s.f->prow_done[ry], (uint32_t)prefix_need, &s.f->error); prefix_row = -1; }
      RBT_FLAG_PUBLISH(&s.f->prow_done[ry], rx + 1);
    }   // bottom line of this CTB and (rx == 1) the context variables are out: a row task below may go on
    end = rbt_cd_terminate(&s.c);
    addr++; count++;
    if (rbt_cd_overrun(&s.c)) { s.error = 2; break; }
    if (!end && wpp && RBT_UNI(addr % pzc_w_ctb(&s)) == 0) {
      // end_of_subset_one_bit, byte_alignment(): the bit that ended the arithmetic codeword is the alignment bit; the next CTB row is its own
      // codeword from the next byte on
      if (!rbt_cd_terminate(&s.c)) { s.error = 1; break; }
      if (ctb_limit > 0 && (int)(count - seg_first) == ctb_limit) break;   // this wave had one substream of the segment: the next row is another wave's
      rbt_cd_restart_aligned(&s.c);
    }
    if (!end && ctb_limit > 0 && (int)(count - seg_first) >= ctb_limit) { s.error = 1; break; }   // the substream should have ended here
    // every entry covers exactly the CTBs up to where the next one of the picture starts: ending early leaves a hole (and a row task below waiting for a row nobody
    // parses), going on runs into CTBs another wave writes
    if ((end || (ctb_limit > 0 && (int)(count - seg_first) == ctb_limit)) ? addr != seg_end : addr >= seg_end) { s.error = 3; break; }
    if (end) {
      const int nxt = RBT_UNI(slices[seg].next_seg);
      if (nxt >= 0) {   // dependent slice segment: same slice, the context variables and the QpY predictor go on (9.3.1, 8.6.1); its own arithmetic codeword
        if (RBT_UNI(slices[nxt].ctb_addr) != addr) { s.error = 1; break; }
        seg = nxt; end = 0; ctb_limit = RBT_UNI(slices[seg].ctb_limit); seg_first = count; seg_end = RBT_UNI(slices[seg].end_addr);
        rbt_cd_start(&s.c, rbsp + (uint32_t)RBT_UNI(slices[seg].data_off), (uint32_t)RBT_UNI(slices[seg].data_size));
      }
    }
    RBT_SYNC_LDS();
  }
#ifdef RBT_PROFILE
  if (RBT_LANE0) for (int i = 0; i < 22; i++) printf("stamp %d: %llu cycles, %u hits\n", i, lds->prof[i], lds->profn[i]);
  if (RBT_LANE0) for (int i = 22; i < 26; i++) printf("stamp %d: %llu cycles, %u hits\n", i, lds->prof[i], lds->profn[i]);
  if (RBT_LANE0) printf("slice %d: total %llu cyc, residual %llu (%u TBs) [setup+last %llu, csbf+sig %llu, gt1/2 %llu, levels %llu], TU total (incl. residual) %llu, CU header %llu (%u CUs), ctb begin/end %llu, CU total %llu, fills %llu, mpm %llu, bins ctx %u bypass %u, bits %u\n", slice_idx, __builtin_readcyclecounter() - t_all_, s.t_res, s.n_res, s.t_a, s.t_b, s.t_c, s.t_d, s.t_tu, s.t_hdr, s.n_cu, s.t_ctb, s.t_cu, s.t_fill, s.t_mpm, s.c.n_bins, s.c.n_byp, s.c.widx * 32u - (uint32_t)s.c.nbuf);
#endif
  if (RBT_LANE0) { slices[own_idx].n_ctbs_decoded = count; if (s.error) s.f->error = s.error; if (sv) sv->phase = 2; }
  // a wave that gave up lets the row below go on at once (the picture is marked bad; nobody waits out the bound for rows that will not come)
  if (s.error && wpp && addr < n_ctb) { const int wc = pzc_w_ctb(&s); RBT_FLAG_PUBLISH(&s.f->prow_done[addr / wc], wc); }
}
