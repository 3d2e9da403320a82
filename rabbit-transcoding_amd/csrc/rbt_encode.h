// RBT-E1 encoder kernels: the MI355X replacement of the libx265 encode the reference drives through libavcodec
// (PCCTranscoder.cpp:548-592 encodeVideo, :825-904 setEncoderOptions) and of resize_frame2 (:594-646).
// Bit-exact counterpart of oracle/hevc_enc.c (product mode). Stages, all one 64-lane wave per work item:
//   en_analyse_ctb     open-loop intra analysis of one CTB, blocks of 32 -> 16 -> 8: a two-step search over the 35 modes (or, in a transcode, planar,
//                      DC and the input stream's own modes), blocks inside a well-predicted block skipped, bottom-up quadtree (parallel over every CTB
//                      of every I picture)
//   en_intra_ctb       closed-loop intra coding of one CTB: predict, forward DST/DCT, dead-zone quantiser, reconstruction; every CU as one transform unit
//                      or four, decided on the reconstruction, 4x4 luma blocks with the DST or transform skip (serial along a CTB row; rows are independent slices or, in wavefront mode, follow the
//                      row above at a distance of two CTBs)
//   en_inter_ctb       P pictures: zero-motion merge from the reconstructed IDR, 16x16 CUs, skip merging (fully parallel)
//   en_sao_ctb         SAO parameters of one CTB from source-vs-reconstruction statistics, applied by the same wave
//   en_entropy_slice   wave-uniform CABAC encoding of one slice segment (independent slices in parallel; wavefront mode: a row starts from the context
//                      variables the row above had after its second CTB)
//   en_pool_sample     2x2 OR-pool of the occupancy map
#pragma once
#include "rbt_cabac.h"
#include "rbt_recon.h"
#include "rbt_filter.h"

RBT_CONST uint16_t k_lambda16[76] = {3,    3,    4,    4,    5,    5,    6,    7,    8,    9,    10,   11,   12,   14,   15,   17,   19,   22,   24,
                                     27,   30,   34,   38,   43,   48,   54,   61,   68,   77,   86,   97,   108,  122,  137,  153,  172,  193,  217,
                                     244,  273,  307,  344,  387,  434,  487,  547,  614,  689,  773,  868,  974,  1093, 1227, 1378, 1546, 1736, 1948,
                                     2187, 2454, 2755, 3092, 3471, 3896, 4373, 4909, 5510, 6185, 6942, 7792, 8747, 9818, 11020, 12370, 13884, 15585, 17493};
RBT_CONST uint8_t k_intra_cand[11] = {0, 1, 26, 10, 2, 6, 14, 18, 22, 30, 34};
#define RBT_SPLIT_BITS 48     // oracle SPLIT_BITS
#define RBT_PARTIAL_COST 0x0FFFFFFF
#define RBT_AN_GOOD 2               // average absolute prediction error per sample at which a block is not subdivided further (oracle AN_GOOD; 0 for lossless streams)
#define RBT_AN_SKIPPED 0x0FFFFFFE    // cost of a block not evaluated because the block around it is good enough: never chosen by the split decision

// Every kernel declares only the LDS it uses: the footprint per workgroup decides how many waves are resident per CU.
struct RbtEncLds {             // inter coding (k_enc_inter)
  RbtReconLds rc;
  int16_t lvl[32 * 32];      // quantised levels of the current TB
};
struct RbtEntropyLds {         // entropy coder (k_entropy): every slice of a 64-picture batch resident at once
  uint8_t scan[3][4][64];    // k_scan staged once per slice
  uint8_t cu_l2[81], cu_md[81], cu_fl[81];   // cu_log2 / cu_mode / cu_flags of the CTB's 8x8 units and of the column / row before it:
                                             // (uy + 1) * 9 + ux + 1, ux,uy = -1..7; cu_l2 = 0xFF where the unit is not available (6.4.1)
  // levels of the current CTB (n = CTB size, row stride n / n/2), fetched once per CTB: Y at 0, Cb at n*n, Cr at n*n*5/4. LAST member:
  // the kernel variant for CTBs up to 32x32 declares only RBT_ENTROPY_LDS_BYTES(5) of it (4.3 instead of 13 KB per slice)
  alignas(16) int16_t lv[64 * 64 * 3 / 2];
};
#define RBT_ENTROPY_LDS_BYTES(tl2) (sizeof(RbtEntropyLds) - sizeof(int16_t) * (64 * 64 * 3 / 2 - (3 << (2 * (tl2) - 1))))
RBT_DEV RBT_LDS_AS int16_t* en_lv(RBT_LDS_AS RbtEntropyLds* l, int c, int log2_ctb) { const int nn = 1 << (2 * log2_ctb); return l->lv + (c == 0 ? 0 : c == 1 ? nn : nn + (nn >> 2)); }

// sum of v over the lanes of the wave (rbt_recon.h rbt_wave_sum; the second argument is a leftover of the LDS form)
RBT_DEV int en_wave_sum(int v, RBT_LDS_AS RbtEncLds* l) { (void)l; return rbt_wave_sum(v); }
// SATD building block (oracle/hevc_enc.c satd_block): absolute 8x8 Hadamard coefficients of one 8x8 tile of residuals, lane p = sample (p & 7, p >> 3).
// Six butterfly stages over the lane index bits without touching LDS: the mirror inside 8 lanes (pairs i and 7 - i instead of i and i + 4: applied to the
// samples, a pairing that is linear in the index bits only permutes the coefficients, and only the SUM of magnitudes is used; it has to come first), quad
// swaps, the rotation by 8 inside a row of 16, a swizzle across 16 and a permute across 32. EN_HAD8X8_ACC adds the tile's magnitudes to an accumulator that en_wave_sum reduces at the end.
#ifdef RBT_HOSTEMU
static inline int en_had8x8_sum_host(const int* r) {
  int t[64], sum = 0;
  for (int y = 0; y < 8; y++) for (int k = 0; k < 8; k++) { int s = 0; for (int x = 0; x < 8; x++) s += (__builtin_popcount(k & x) & 1) ? -r[y * 8 + x] : r[y * 8 + x]; t[y * 8 + k] = s; }
  for (int k = 0; k < 8; k++) for (int x = 0; x < 8; x++) { int s = 0; for (int y = 0; y < 8; y++) s += (__builtin_popcount(k & y) & 1) ? -t[y * 8 + x] : t[y * 8 + x]; sum += s < 0 ? -s : s; }
  return sum;
}
#define EN_HAD8X8_ACC(vr, acc) (acc) += en_had8x8_sum_host(vr)
#else
RBT_DEV int en_had_stage(int v, int partner, int upper) { return upper ? partner - v : partner + v; }
RBT_DEV int en_had8x8_abs(int v) {
  const int lane = (int)threadIdx.x & 63;
  v = en_had_stage(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false), lane & 4);    // row_half_mirror: FIRST, while the lane bits still index samples
  v = en_had_stage(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false), lane & 1);     // quad_perm:[1,0,3,2]
  v = en_had_stage(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false), lane & 2);     // quad_perm:[2,3,0,1]
  v = en_had_stage(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false), lane & 8);    // row_ror:8
  v = en_had_stage(v, __builtin_amdgcn_ds_swizzle(v, 0x401F), lane & 16);                      // lane ^ 16 (bit mode: and 0x1F, or 0, xor 0x10)
  v = en_had_stage(v, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v), lane & 32);
  return v < 0 ? -v : v;
}
#define EN_HAD8X8_ACC(vr, acc) (acc) += en_had8x8_abs(vr)
#endif
// occupancy-aware coding: does the 4x4 luma unit (ux,uy) of the picture make a point (or sit next to one that does)?
RBT_DEV int en_occ_unit(const RbtFrame* f, int ux, int uy) { return ux < f->occ4_w && uy < f->occ4_h && f->occ4[(size_t)uy * f->occ4_w + ux] != 0; }
// mean of `sum` over `cnt` samples, rounded half away from zero (oracle/hevc_enc.c recon_tb: what the unoccupied samples of a partly occupied block ask for)
RBT_DEV int en_round_mean(int sum, int cnt) { return sum >= 0 ? (sum + cnt / 2) / cnt : -((-sum + cnt / 2) / cnt); }
// RbtFrame::occ4 of unit (i,j) of a W x H picture from the occupancy map `occ` (ow x oh samples, the atlas scaled down by s = W / ow): does any occupancy sample that covers
// part of the unit or of one of its eight neighbours say "occupied"? (oracle/vpcc_path.c occ_units: the units' own samples, then one unit of margin - the sample ranges of
// neighbouring units follow each other, so the union is one rectangle)
RBT_DEV int en_occ_unit_value(const uint16_t* occ, int ow, int oh, int s, int w4, int h4, int i, int j) {
  const int x_lo = (4 * rbt_max(i - 1, 0)) / s, x_hi = rbt_min(ow - 1, (4 * rbt_min(i + 1, w4 - 1) + 3) / s), y_lo = (4 * rbt_max(j - 1, 0)) / s, y_hi = rbt_min(oh - 1, (4 * rbt_min(j + 1, h4 - 1) + 3) / s);
  int any = 0;
  for (int y = y_lo; y <= y_hi; y++) for (int x = x_lo; x <= x_hi; x++) any |= occ[(size_t)y * ow + x] != 0;
  return any;
}
// rounding offset of the intra quantiser in 1/512 of a level (oracle/hevc_enc.c e1_quant_intra): by the level below and, for a first level, the position
RBT_DEV int en_rq_offset(int lf_is0, int lf_is1, int xy_sum) { return lf_is0 ? (xy_sum <= 2 ? 180 : 145) : (lf_is1 ? 195 : 230); }
RBT_DEV int en_chroma_qp(const RbtFrame* f, const RbtSlice* sl, int c_idx, int qp_y) {
  int off = c_idx == 1 ? f->cfg.cb_qp_offset + sl->cb_qp_offset : f->cfg.cr_qp_offset + sl->cr_qp_offset;
  int bdo = 6 * (f->cfg.bit_depth - 8);
  int qpi = rbt_clip3(-bdo, 57, qp_y + off);
  return (qpi < 0 ? qpi : rbt_chroma_qp(qpi)) + bdo;
}

// ------------------------------------------------------------------------------------------------ analysis (I pictures)
// availability (6.4.1) of luma position (xn,yn) as intra reference of the block at (xc,yc), both inside or next to CTB
// (cx,cy): same result as rc_avail, from flags of the four neighbouring CTBs fetched once instead of map reads per sample
struct EnCtbNb { int cx, cy, ctb, w, h, left, above_left, above, above_right; };
RBT_DEV int en_avail(const EnCtbNb* q, int xc, int yc, int xn, int yn) {
  if (xn < 0 || yn < 0 || xn >= q->w || yn >= q->h) return 0;
  const int dx = xn - q->cx, dy = yn - q->cy;
  if (dy >= q->ctb) return 0;
  if (dy < 0) return dx < 0 ? q->above_left : (dx < q->ctb ? q->above : q->above_right);
  if (dx < 0) return q->left;
  if (dx >= q->ctb) return 0;
  return rc_z_before(dx >> 2, dy >> 2, (xc - q->cx) >> 2, (yc - q->cy) >> 2);
}
// The analysis kernel is throughput work (one wave per CTB, 400 CTBs per picture): what it keeps in LDS decides how many
// waves share a SIMD and hide each other's LDS latency. 10 KB instead of the 30 KB of RbtEncLds: 14 workgroups per CU.
struct RbtAnalyseLds {
  int32_t nb[132], nbf[132], ref[100];   // reference samples of the current block (plain / smoothed), angular reference array
  uint16_t src[65 * 66];                 // source samples of a 32x32 quadrant and what its blocks reference around it: (yy + 1) * 66 + xx + 1
  int32_t cost[3][16]; uint8_t mode[3][16], split[3][16];
  uint8_t hint[64];                      // transcoder: the input stream's luma intra mode of the quadrant's 4x4 units (uy * 8 + ux; 255 = not intra / outside)
};
RBT_DEV int en_nb_unit_av(const EnCtbNb* q, int p, int x0, int y0, int S) { int xn, yn; rc_nb_unit_xy(p, x0, y0, S, 0, &xn, &yn); return en_avail(q, x0, y0, xn, yn); }   // unit p of the block at (x0,y0) of the picture (rc_nb_unit_xy)
RBT_DEV void en_analyse_ctb(RbtFrame* f, const RbtSlice* slices, int ctb_addr, RBT_LDS_AS RbtAnalyseLds* l) {
  const RbtStreamCfg gcopy = rc_cfg_uni(&f->cfg); const RbtStreamCfg* g = &gcopy;
  int ctb = 1 << g->log2_ctb, rx = ctb_addr % g->w_ctb, ry = ctb_addr / g->w_ctb, cx = rx << g->log2_ctb, cy = ry << g->log2_ctb;
  const int my_slice = f->ctb_slice[ctb_addr];
  const RbtSlice* sl = &slices[my_slice];
  RBT_LDS_AS RbtAnalyseLds* rl = l;
  EnCtbNb nbq; nbq.cx = cx; nbq.cy = cy; nbq.ctb = ctb; nbq.w = g->w; nbq.h = g->h;
  nbq.left = rx > 0 && f->ctb_slice[ctb_addr - 1] == my_slice;
  nbq.above = ry > 0 && f->ctb_slice[ctb_addr - g->w_ctb] == my_slice;
  nbq.above_left = rx > 0 && ry > 0 && f->ctb_slice[ctb_addr - g->w_ctb - 1] == my_slice;
  nbq.above_right = ry > 0 && rx + 1 < g->w_ctb && f->ctb_slice[ctb_addr - g->w_ctb + 1] == my_slice;
  const uint16_t* srcp = f->src[0];
  const int hints = f->hint_dm != nullptr, hint_w4 = f->hint_w4, hint_h4 = f->hint_h4, satd_on = f->enc_tools & RBT_ET_SATD;
  // CTBs larger than 32 are analysed as independent 32x32 quadrants (a 64x64 intra CU is always split)
  int nq = ctb > 32 ? 2 : 1, qs = ctb > 32 ? 32 : ctb;
  for (int q = 0; q < nq * nq; q++) {
    int qx = cx + (q % nq) * 32, qy = cy + (q / nq) * 32;
    if (nq > 1 && (qx >= g->w || qy >= g->h)) continue;
    // source samples of the quadrant and of everything its blocks can reference (one row / column before it, 2 * 32 beyond)
    // rows yy = -1..63, columns 0..63 as 8-byte groups of four samples (qx is a multiple of 32, the width a multiple of 8), several loads in flight
    if (!(RBT_ABLATE & 0x100))
#pragma unroll 4
    RBT_PAR_FOR(i, 65 * 16) {
      const int x4 = i & 15, yy = (i >> 4) - 1, x = qx + 4 * x4, y = qy + yy;
      RbtU2 v; v.x = 0; v.y = 0;
      if (y >= 0 && y < g->h && x < g->w) v = *(const RbtU2*)(srcp + (size_t)y * g->w + x);
      RBT_LDS_AS uint16_t* d = &l->src[(yy + 1) * 66 + 4 * x4 + 1];
      d[0] = (uint16_t)v.x; d[1] = (uint16_t)(v.x >> 16); d[2] = (uint16_t)v.y; d[3] = (uint16_t)(v.y >> 16);
    }
    RBT_PAR_FOR(i, 65) { const int x = qx - 1, y = qy + i - 1; l->src[i * 66] = (x >= 0 && y >= 0 && y < g->h) ? srcp[(size_t)y * g->w + x] : 0; }
    if (hints && !(RBT_ABLATE & 0x400)) RBT_PAR_FOR(u, 64) {                // once per quadrant instead of two dependent loads per quarter of every block
      const int hx = (qx >> 2) + (u & 7), hy = (qy >> 2) + (u >> 3); int v = 255;
      if (hx < hint_w4 && hy < hint_h4) { const size_t hk = (size_t)hy * hint_w4 + hx; if ((f->hint_pm[hk] & RBT_PM_MODE_MASK) == RBT_MODE_INTRA) v = f->hint_dm[hk] & 63; }
      l->hint[u] = (uint8_t)v;
    }
    RBT_SYNC_LDS();
    for (int si = 2; si >= 0; si--) {                      // largest blocks first: the 8x8 blocks take their candidates from the 16x16 block around them
      int S = 8 << si; if (S > qs) continue;
      int nb = qs / S, lg = 3 + si;
      for (int b = 0; b < nb * nb; b++) {
        int x0 = qx + (b % nb) * S, y0 = qy + (b / nb) * S, best = 0x7FFFFFFF, bmode = 0;
        if (x0 >= g->w || y0 >= g->h) best = 0;
        else if (x0 + S > g->w || y0 + S > g->h) best = RBT_PARTIAL_COST;
        else if (si < 2 && 2 * S <= qs && (RBT_UNI(l->cost[si + 1][((b / nb) >> 1) * (nb >> 1) + ((b % nb) >> 1)]) <= (f->lossless ? 0 : RBT_AN_GOOD) * 4 * S * S ||
                                            RBT_UNI(l->cost[si + 1][((b / nb) >> 1) * (nb >> 1) + ((b % nb) >> 1)]) == RBT_AN_SKIPPED)) best = RBT_AN_SKIPPED;   // the block around it is good enough: it will not be split
        else {
          // reference samples once per block: availability masks, substitution (8.4.4.2.2) while gathering, and the smoothed copy
          const int tot = 4 * S + 1, bx0 = x0 - qx, by0 = y0 - qy;
          if (!(RBT_ABLATE & 0x200)) {
          uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(S, 0), en_nb_unit_av(&nbq, p, x0, y0, S));
          RcNbMap nm; rc_nb_map(&nm, m, S, 0);
          RBT_PAR_FOR(i, tot) {
            int v = 1 << (g->bit_depth - 1);
            if (m) { const int j = rc_nb_source(&nm, i); v = j < 2 * S ? l->src[(by0 + 2 * S - j) * 66 + bx0] : l->src[by0 * 66 + bx0 + j - 2 * S]; }
            rl->nb[i] = v;
          }
          RBT_SYNC_LDS();
          rc_intra_filter_apply(g, lg, rl->nb, rl->nbf);
          }
          // 16x16 / 32x32: 11 coarse candidates, then the angular modes within two of the best coarse one; 8x8 inside a complete 16x16 block: planar, DC,
          // vertical, horizontal and the angular modes within two of the 16x16 block's mode (2, 18, 34 when that is not angular) (oracle/hevc_enc.c analyse_ctb_intra)
          int parent = -1;
          if (si == 0 && qs >= 16) { const int pb = ((b / nb) >> 1) * (nb >> 1) + ((b % nb) >> 1); if (RBT_UNI(l->cost[1][pb]) < RBT_PARTIAL_COST) parent = RBT_UNI(l->mode[1][pb]); }
          // transcoder: planar, DC and the modes the input stream coded at the block's four quarters (distinct ones, in that order; vertical and horizontal
          // instead where the input has no intra mode) - oracle/hevc_enc.c analyse_ctb_intra, hint_modes. Candidates packed 6 bits each.
          uint64_t hc = 0; int nh = 0;
          if (hints) {
            int any = 0;
            hc = 1ull << 6; nh = 2;
            if (!(RBT_ABLATE & 0x400)) for (int q = 0; q < 4; q++) {
              const int v = RBT_UNI(l->hint[((by0 + (q >> 1) * (S >> 1)) >> 2) * 8 + ((bx0 + (q & 1) * (S >> 1)) >> 2)]);
              if (v < 35) {
                any = 1;
                int dup = 0; for (int t = 0; t < nh; t++) dup |= (int)((hc >> (6 * t)) & 63) == v;
                if (!dup) { hc |= (uint64_t)v << (6 * nh); nh++; }
              }
            }
            if (!any) { hc |= (26ull << 12) | (10ull << 18); nh = 4; }
          }
          int coarse = 0;
          RBT_VEC(int, v_last); RBT_VEC(int, v_keep);           // 8x8 blocks (one sample per lane): the residual of the candidate just tried / of the best so far - the SATD below needs no second prediction
          RBT_VFOR(p, 64) { RBT_V(v_last, p) = 0; RBT_V(v_keep, p) = 0; }
          for (int k = 0; k < 15; k++) {
            int mode;
            if (best == 0) break;                               // cannot get better
            if (hints) { if (k >= nh) break; mode = (int)((hc >> (6 * k)) & 63); }
            else if (parent >= 0) {
              if (k >= 9) break;
              if (k < 4) mode = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 26 : 10));
              else if (parent >= 2) { mode = parent + (k - 6); if (mode < 2 || mode > 34 || mode == 10 || mode == 26) continue; }
              else { if (k >= 7) break; mode = k == 4 ? 2 : (k == 5 ? 18 : 34); }
            } else if (k < 11) mode = k_intra_cand[k];
            else { if (k == 11) coarse = bmode; if (coarse < 2) break; mode = coarse + (k == 11 ? -2 : k == 12 ? -1 : k == 13 ? 1 : 2); if (mode < 2 || mode > 34) continue; }
            RBT_LDS_AS int32_t* fin = rc_intra_filter_needed(0, lg, mode) ? rl->nbf : rl->nb;
            RcIntraCtx qc; if (!(RBT_ABLATE & 0x800)) rc_intra_setup(g, 0, lg, mode, fin, rl->ref, &qc); else { qc.N = S; qc.log2 = lg; qc.mode = mode; qc.c_idx = 0; qc.maxv = 1023; qc.ang = 1; qc.ver = mode >= 18; qc.dc = 1; qc.edge = 0; }
            int part = (RBT_ABLATE & 0x1000) ? 1 + k : 0;
#define EN_BODY(PV) RBT_PAR_FOR(i, S * S) { const int x = i & (S - 1), y = i >> lg, r = (int)l->src[(by0 + y + 1) * 66 + bx0 + x + 1] - (PV); part += rbt_abs(r); RBT_V(v_last, i & 63) = r; }
            if (!(RBT_ABLATE & 0x1000)) RC_INTRA_KINDS(&qc, fin, rl->ref, EN_BODY);
#undef EN_BODY
            int sad = en_wave_sum(part, (RBT_LDS_AS RbtEncLds*)0);
            if (sad < best) { best = sad; bmode = mode; if (S == 8) { RBT_VFOR(p, 64) RBT_V(v_keep, p) = RBT_V(v_last, p); } }
            RBT_SYNC_LDS();                                     // rl->ref is rebuilt by the next mode
          }
          if (satd_on && best > 0 && S == 8 && !(RBT_ABLATE & (0x2000 | 0x1000))) {
            int acc = 0; EN_HAD8X8_ACC(v_keep, acc);             // lane p = sample (p & 7, p >> 3) in the SAD pass and in the transform alike
            best = (en_wave_sum(acc, (RBT_LDS_AS RbtEncLds*)0) + 4) >> 3;
          } else if (satd_on && best > 0 && !(RBT_ABLATE & 0x2000)) {
            // the mode by SAD, the block's cost (what the split decisions compare) by the SATD of that mode (oracle/hevc_enc.c analyse_ctb_intra, satd_block)
            RBT_LDS_AS int32_t* fin = rc_intra_filter_needed(0, lg, bmode) ? rl->nbf : rl->nb;
            RcIntraCtx qc; rc_intra_setup(g, 0, lg, bmode, fin, rl->ref, &qc);
            int acc = 0; const int tw = S >> 3;
#define EN_BODY(PV) for (int tix = 0; tix < tw * tw; tix++) { \
              const int tx = (tix % tw) * 8, ty = (tix / tw) * 8; \
              RBT_VEC(int, v_r); \
              RBT_VFOR(p, 64) { const int x = tx + (p & 7), y = ty + (p >> 3); RBT_V(v_r, p) = (int)l->src[(by0 + y + 1) * 66 + bx0 + x + 1] - (PV); } \
              EN_HAD8X8_ACC(v_r, acc); }
            RC_INTRA_KINDS(&qc, fin, rl->ref, EN_BODY);
#undef EN_BODY
            best = (en_wave_sum(acc, (RBT_LDS_AS RbtEncLds*)0) + 4) >> 3;
            RBT_SYNC_LDS();
          }
        }
    if (RBT_LANE0) { l->cost[si][b] = best; l->mode[si][b] = (uint8_t)bmode; }
      }
    }
    RBT_SYNC_LDS();
    int lam = k_lambda16[rbt_clip3(0, 75, sl->qp + 6 * (g->bit_depth - 8))], pen = (lam * RBT_SPLIT_BITS) >> 4;
    for (int si = 1; si < 3; si++) {
      int S = 8 << si; if (S > qs) break;
      int nb = qs / S, nbc = nb * 2;
      RBT_PAR_FOR(b, nb * nb) {
        int bx = b % nb, by = b / nb;
        int child = l->cost[si - 1][(2 * by) * nbc + 2 * bx] + l->cost[si - 1][(2 * by) * nbc + 2 * bx + 1] + l->cost[si - 1][(2 * by + 1) * nbc + 2 * bx] +
                    l->cost[si - 1][(2 * by + 1) * nbc + 2 * bx + 1] + pen;
        int split = child < l->cost[si][b];
        l->split[si][b] = (uint8_t)split;
        if (split) l->cost[si][b] = child;
      }
      RBT_SYNC_LDS();
    }
    // leaf CU size / mode per 8x8 unit of the quadrant
    int n8 = qs / 8;
    RBT_PAR_FOR(u, n8 * n8) {
      int ux = u % n8, uy = u / n8, x = qx + ux * 8, y = qy + uy * 8;
      if (x < g->w && y < g->h) {
        int lg = 3, mode = l->mode[0][uy * n8 + ux];
        int inside32 = qx + 32 <= g->w && qy + 32 <= g->h, inside16 = (x & ~15) + 16 <= g->w && (y & ~15) + 16 <= g->h;
        int b16 = (uy / 2) * (n8 / 2) + ux / 2;
        int split32 = qs >= 32 ? (inside32 ? l->split[2][0] : 1) : 1;
        int split16 = qs >= 16 ? (inside16 ? l->split[1][b16] : 1) : 1;
        if (qs >= 32 && !split32) { lg = 5; mode = l->mode[2][0]; }
        else if (qs >= 16 && !split16) { lg = 4; mode = l->mode[1][b16]; }
        int k = (y >> 3) * f->w8 + (x >> 3);
        f->cu_log2[k] = (uint8_t)lg; f->cu_mode[k] = (uint8_t)mode;
      }
    }
    RBT_SYNC_LDS();
  }
}

// ------------------------------------------------------------------------------------------------ transform / quantiser
// forward transform of l->rc.res (residual, N x N) into l->rc.res (coefficients); HM shift convention
template <int LOG2> RBT_DEV void en_fwd_transform_n(int is_dst, int bd, RBT_LDS_AS RbtReconLdsCore* r) {   // fully unrolled per size (see rc_inv_transform_n)
  constexpr int N = 1 << LOG2; const int s1 = LOG2 + bd - 9, s2 = LOG2 + 6;
  RBT_PAR_FOR(i, N * N) {
    int k = i & (N - 1), y = i >> LOG2, s = 0;
#pragma unroll
    for (int x = 0; x < N; x++) s += rc_tcoef(r, N, is_dst, k, x) * r->res[y * N + x];
    r->tmp[i] = (int16_t)(s1 > 0 ? (s + (1 << (s1 - 1))) >> s1 : s);
  }
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, N * N) {
    int kh = i & (N - 1), kv = i >> LOG2, s = 0;
#pragma unroll
    for (int y = 0; y < N; y++) s += rc_tcoef(r, N, is_dst, kv, y) * r->tmp[y * N + kh];
    r->res[i] = (int16_t)rbt_clip3(-32768, 32767, (s + (1 << (s2 - 1))) >> s2);
  }
  RBT_SYNC_LDS();
}
template <int LOG2> RBT_DEV void en_fwd_transform_pair_n(int bd, RBT_LDS_AS RbtReconLdsCore* r) {   // two blocks at res / tmp offsets 0 and 256
  constexpr int N = 1 << LOG2, NN = N * N; const int s1 = LOG2 + bd - 9, s2 = LOG2 + 6;
  RBT_PAR_FOR(i, 2 * NN) {
    const int b = i >> (2 * LOG2), j = i & (NN - 1), k = j & (N - 1), y = j >> LOG2; int s = 0;
#pragma unroll
    for (int x = 0; x < N; x++) s += rc_tcoef(r, N, 0, k, x) * r->res[b * 256 + y * N + x];
    r->tmp[b * 256 + j] = (int16_t)(s1 > 0 ? (s + (1 << (s1 - 1))) >> s1 : s);
  }
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, 2 * NN) {
    const int b = i >> (2 * LOG2), j = i & (NN - 1), kh = j & (N - 1), kv = j >> LOG2; int s = 0;
#pragma unroll
    for (int y = 0; y < N; y++) s += rc_tcoef(r, N, 0, kv, y) * r->tmp[b * 256 + y * N + kh];
    r->res[b * 256 + j] = (int16_t)rbt_clip3(-32768, 32767, (s + (1 << (s2 - 1))) >> s2);
  }
  RBT_SYNC_LDS();
}
// 32 x 32 on the matrix cores: tmp[y][k] = (sum_x res[y][x] T[k][x] + round) >> s1, then res[kv][kh] = clip16((sum_y T[kv][y] tmp[y][kh] + round) >> s2)
RBT_DEV void en_fwd_transform_32(int bd, RBT_LDS_AS RbtReconLdsCore* r) {
#ifdef RBT_HOSTEMU
  en_fwd_transform_n<5>(0, bd, r);
#else
  mf_mm32<false>(r->dct, 32, 1, r->res, r->tmp, 5 + bd - 9, 0);
  mf_mm32<true>(r->dct, 32, 1, r->tmp, r->res, 5 + 6, 1);
#endif
}
RBT_DEV void en_fwd_transform(int log2, int is_dst, int bd, RBT_LDS_AS RbtReconLdsCore* r) {
  if (log2 == 2) en_fwd_transform_n<2>(is_dst, bd, r);
  else if (log2 == 3) en_fwd_transform_n<3>(0, bd, r);
  else if (log2 == 4) en_fwd_transform_n<4>(0, bd, r);
  else en_fwd_transform_32(bd, r);
}
// dead-zone quantiser of l->rc.res into l->lvl; returns the number of non-zero levels
RBT_DEV int en_quant(int log2, int qp, int bd, int is_intra, RBT_LDS_AS RbtEncLds* l) {
  int N = 1 << log2, qbits = 14 + qp / 6 + (15 - bd - log2), sc = k_quant_scale[qp % 6], part = 0;
  long long add = (long long)(is_intra ? 171 : 85) << (qbits - 9);
  RBT_PAR_FOR(i, N * N) {
    int c = l->rc.res[i], a = rbt_abs(c);
    long long q = ((long long)a * sc + add) >> qbits;
    if (q > 32767) q = 32767;
    l->lvl[i] = (int16_t)(c < 0 ? -q : q);
    part += q != 0;
  }
  int nz = en_wave_sum(part, l);
  return nz;
}
// codes one TB: residual = src - prediction (prediction in l->rc.pred), levels -> coef plane, reconstruction -> pix. Returns cbf.
RBT_DEV int en_code_tb(RbtFrame* f, int c_idx, int x0, int y0, int log2, int qp, int is_intra, RBT_LDS_AS RbtEncLds* l) {
  const RbtStreamCfg* g = &f->cfg;
  int N = 1 << log2, pw = c_idx ? g->cw : g->w, bd = g->bit_depth, maxv = (1 << bd) - 1;
  const uint16_t* sp = f->src[c_idx]; uint16_t* rp = f->pix[c_idx]; int16_t* cp = f->coef[c_idx];
  RBT_PAR_FOR(i, N * N) { int x = i & (N - 1), y = i >> log2; l->rc.res[i] = (int16_t)((int)sp[(size_t)(y0 + y) * pw + x0 + x] - (int)l->rc.pred[i]); }
  RBT_SYNC_LDS();
  // occupancy-aware coding (see en_tile_intra_tb; x0, y0 are picture coordinates of component c_idx)
  int occ_none = 0;
  if (f->occ4 != nullptr) {
    const int sh = c_idx ? 1 : 0;
#define EN_TBP_OCC(i) en_occ_unit(f, ((x0 + ((i) & (N - 1))) << sh) >> 2, ((y0 + ((i) >> log2)) << sh) >> 2)
    int pc = 0, ps = 0;
    RBT_PAR_FOR(i, N * N) { if (EN_TBP_OCC(i)) { pc++; ps += l->rc.res[i]; } }
    const int cnt = en_wave_sum(pc, l);
    occ_none = cnt == 0;
    if (cnt > 0 && cnt < N * N) {
      const int mean = en_round_mean(en_wave_sum(ps, l), cnt);
      RBT_PAR_FOR(i, N * N) { if (!EN_TBP_OCC(i)) l->rc.res[i] = (int16_t)mean; }
      RBT_SYNC_LDS();
    }
#undef EN_TBP_OCC
  }
  int nz;
  if (occ_none) { RBT_PAR_FOR(i, N * N) l->lvl[i] = 0; nz = 0; }
  else if (f->lossless) {
    int part = 0;
    RBT_PAR_FOR(i, N * N) { l->lvl[i] = l->rc.res[i]; part += l->rc.res[i] != 0; }
    nz = en_wave_sum(part, l);
  } else {
    en_fwd_transform(log2, c_idx == 0 && log2 == 2 && is_intra, bd, &l->rc);
    nz = en_quant(log2, qp, bd, is_intra, l);
  }
  RBT_SYNC_LDS();
  // the levels of a block without any are not stored: the entropy coder reads a block's levels only when its cbf is set (P pictures are mostly such blocks: 0.3 GB per GOF)
  if (nz) { RBT_PAR_FOR(i, N * N) { int x = i & (N - 1), y = i >> log2; cp[(size_t)(y0 + y) * pw + x0 + x] = l->lvl[i]; } }
  if (nz && !f->lossless) {
    int bd_shift = bd + log2 - 5, scale = (16 * k_dequant_scale[qp % 6]) << (qp / 6);
    long long add = 1ll << (bd_shift - 1);
    RBT_PAR_FOR(i, N * N) { long long v = ((long long)l->lvl[i] * scale + add) >> bd_shift; l->rc.res[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
    RBT_SYNC_LDS();
    rc_inv_transform(log2, c_idx == 0 && log2 == 2 && is_intra, 0, bd, &l->rc);
  } else if (nz) {
    RBT_PAR_FOR(i, N * N) l->rc.res[i] = l->lvl[i];
    RBT_SYNC_LDS();
  }
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2;
    rp[(size_t)(y0 + y) * pw + x0 + x] = (uint16_t)(nz ? rbt_clip3(0, maxv, (int)l->rc.pred[i] + l->rc.res[i]) : l->rc.pred[i]);
  }
  RBT_SYNC();
  return nz != 0;
}
// (a CU of 16 or 32 coded as four transform units calls this once per unit: the unit's left and top borders are transform edges for the deblocking filter)
RBT_DEV void en_fill_cu_maps(RbtFrame* f, int x0, int y0, int N, int pm_val, int qp_y, int cbf_bits_or_flags, int set_flags) {
  const RbtStreamCfg* g = &f->cfg;
  int n4 = N >> 2, n8 = N >> 3;
  RBT_PAR_FOR(i, n4 * n4) {
    int k = ((y0 >> 2) + i / n4) * g->w4 + (x0 >> 2) + i % n4;
    f->pm[k] = (uint8_t)pm_val; f->qp[k] = (int8_t)qp_y;
    int e = 0;
    if (i % n4 == 0) e |= RBT_EV_TU | RBT_EV_PU;
    if (i / n4 == 0) e |= RBT_EH_TU | RBT_EH_PU;
    f->edges[k] = (uint8_t)e;
  }
  if (set_flags) RBT_PAR_FOR(i, n8 * n8) { int k = ((y0 >> 3) + i / n8) * f->w8 + (x0 >> 3) + i % n8; f->cu_flags[k] = (uint8_t)cbf_bits_or_flags; }
}

// ------------------------------------------------------------------------------------------------ I pictures: one CTB
// ---- closed-loop intra coding of one CTB inside LDS --------------------------------------------------------------
// Same staging as the decoder's rbt_recon_ctb: the reconstruction of the CTB, its border and the availability flags of the
// 4x4 units around it live in LDS, every TB predicts / transforms / reconstructs there, and the CTB is written back with
// coalesced row stores. Levels go straight to the coefficient plane (nobody waits for those stores).
// The body and the row above it are separate arrays (the row above spans 2n+1 samples, the body n+1 per row): 17 KB per
// workgroup instead of 29, which together with the rest keeps five workgroups (every CTB row of a 32-picture batch) on a CU.
// TL2 = log2 of the largest CTB the kernel variant handles (5 or 6). The transcoder codes with 32x32 CTBs by default, and a
// tile sized for 64x64 costs 28 KB of LDS per workgroup instead of 18 (5 instead of 8 slices in flight per CU).
template <int TL2> struct RbtEncTileT {
  static constexpr int TS_Y = (1 << TL2) + 1, TS_C = (1 << (TL2 - 1)) + 1;   // body row strides: column -1 (left border) .. n-1
  uint16_t y[(1 << TL2) * TS_Y], top_y[2 * (1 << TL2) + 2];   // body: yy * stride + xx + 1 (xx = -1..n-1); top: xx + 1 (xx = -1..2n-1)
  uint16_t c[2][(1 << (TL2 - 1)) * TS_C], top_c[2][(1 << TL2) + 2];
  uint8_t uav[((1 << (TL2 - 2)) + 1) * RC_US];
  uint16_t sb[32 * 32 + 2 * 16 * 16];                        // source samples of the current CU: Y, Cb, Cr
  uint8_t cu_l2[64], cu_md[64];                              // cu_log2 / cu_mode of the CTB's 8x8 units (analysis result; cu_md: the closed-loop choice once a CU is coded)
  uint8_t occ_u[(1 << (TL2 - 2)) * (1 << (TL2 - 2))];        // occupancy-aware coding: RbtFrame::occ4 of the CTB's 4x4 luma units (uy * n4 + ux)
  uint8_t left_md[16];                                       // luma modes of the 8x8 units in the last column of the CTB to the left ([8] = that CTB is available, 6.4.1)
  // one TU or four (en_intra_cu_luma): levels and reconstruction of the CU's luma coded as ONE transform block, kept while it is coded as four;
  // levels of the current quarter; source samples of the current quarter (luma, or Cb at 0 and Cr at 256)
  int16_t lv0[32 * 32]; uint16_t rec0[32 * 32]; int16_t lv1[16 * 16]; uint16_t ss[512];
};
// TB scratch of the intra-coding kernel: the core + the prediction. The smoothed / angular reference arrays alias `tmp` (dead until
// the forward transform) and the quantised levels alias the luma part of `sb` (the source samples are consumed when the residual is
// formed): 14.7 instead of 18 KB per slice, 11 instead of 8 slices in flight per CU.
struct RbtEncIntraScratch : RbtReconLdsCore { uint16_t pred[32 * 32]; };
template <int TL2> struct RbtEncTileLdsT { RbtEncIntraScratch rc; RbtEncTileT<TL2> t; };
RBT_DEV int en_quant_scale(int r) { const uint64_t lo = 26214ull | (23302ull << 16) | (20560ull << 32) | (18396ull << 48), hi = 16384ull | (14564ull << 16); return (int)(((r < 4 ? lo : hi) >> (16 * (r & 3))) & 0xFFFF); }
// What the intra chain asks of the picture over and over, read once per CTB into scalar registers. Through the RbtFrame pointer every use was a load from HBM with a wait
// behind it - nothing lets the compiler keep a value across the stores in between - in the middle of a serial chain: six per transform block.
struct EnCtbCtx : RbtStreamCfg { int lossless, enc_tools, f4, w8; int16_t* coef[3]; uint8_t *pm, *edges, *cu_flags, *cu_mode, *cu_ts; int8_t* qp; const uint16_t* src[3]; };
RBT_DEV void en_ctb_ctx(EnCtbCtx* e, const RbtFrame* f) {
  *(RbtStreamCfg*)e = rc_cfg_uni(&f->cfg);
  e->lossless = RBT_UNI(f->lossless); e->enc_tools = RBT_UNI(f->enc_tools); e->f4 = RBT_UNI(f->occ4 != nullptr);
  for (int c = 0; c < 3; c++) e->coef[c] = rbt_uni_ptr(f->coef[c]);
  e->w8 = RBT_UNI(f->w8); e->pm = rbt_uni_ptr(f->pm); e->edges = rbt_uni_ptr(f->edges); e->cu_flags = rbt_uni_ptr(f->cu_flags); e->cu_mode = rbt_uni_ptr(f->cu_mode); e->qp = rbt_uni_ptr(f->qp);
  e->cu_ts = rbt_uni_ptr(f->cu_ts); for (int c = 0; c < 3; c++) e->src[c] = rbt_uni_ptr(f->src[c]);
}
// en_fill_cu_maps for the intra chain: the maps' addresses from the CTB's context
RBT_DEV void en_fill_cu_maps_ctx(const EnCtbCtx* e, int x0, int y0, int N, int pm_val, int qp_y, int flags) {
  const int n4 = N >> 2, n8 = N >> 3;
  RBT_PAR_FOR(i, n4 * n4) {
    const int k = ((y0 >> 2) + i / n4) * e->w4 + (x0 >> 2) + i % n4;
    e->pm[k] = (uint8_t)pm_val; e->qp[k] = (int8_t)qp_y;
    int ev = 0;
    if (i % n4 == 0) ev |= RBT_EV_TU | RBT_EV_PU;
    if (i / n4 == 0) ev |= RBT_EH_TU | RBT_EH_PU;
    e->edges[k] = (uint8_t)ev;
  }
  RBT_PAR_FOR(i, n8 * n8) { const int k = ((y0 >> 3) + i / n8) * e->w8 + (x0 >> 3) + i % n8; e->cu_flags[k] = (uint8_t)flags; }
}
// one intra TB: (x0,y0) relative to the CTB and (gx,gy) in the picture, both in samples of component c_idx; returns cbf
template <int TL2> RBT_DEV int en_tile_intra_tb(const RbtStreamCfg* g, RbtFrame* f, RBT_LDS_AS RbtEncTileLdsT<TL2>* L, int c_idx, int x0, int y0, int gx, int gy, int log2, int mode, int qp,
                             const RBT_LDS_AS uint16_t* src, int mark_l4, int mux, int muy, RBT_LDS_AS int16_t* lvl_buf = nullptr, long long* cost = nullptr, int lam2 = 0, int* ssd_out = nullptr, int* ts_out = nullptr, int reuse_nb = 0) {
  const EnCtbCtx* e = static_cast<const EnCtbCtx*>(g);                 // every caller hands an EnCtbCtx in (en_intra_ctb)
  RBT_LDS_AS RbtEncIntraScratch* r = &L->rc; RBT_LDS_AS RbtEncTileT<TL2>* t = &L->t;
  // lvl_buf: where the levels go (default: over the luma part of `sb`, i.e. over the source once the residual is formed). cost (needs a lvl_buf that leaves
  // `src` alone): distortion * 256 + lam2 * rate of the block as the oracle's recon_tb / hm_decide_tu_split count them - squared error of the residual
  // against its reconstruction (before clipping), 3 + 2 * floor(log2 |level|) bits per non-zero level plus 3, or 1 bit for an empty block
  RBT_LDS_AS int32_t* const r_nbf = (RBT_LDS_AS int32_t*)r->tmp; RBT_LDS_AS int32_t* const r_ref = r_nbf + 132; RBT_LDS_AS int16_t* const lvl = lvl_buf ? lvl_buf : (RBT_LDS_AS int16_t*)t->sb;
  const int N = 1 << log2, sh = c_idx ? 1 : 0, bd = g->bit_depth, maxv = (1 << bd) - 1, n4 = (1 << g->log2_ctb) >> 2, pw = c_idx ? g->cw : g->w;
  RBT_LDS_AS uint16_t* tile = c_idx == 0 ? t->y : t->c[c_idx - 1]; const int S = c_idx == 0 ? RbtEncTileT<TL2>::TS_Y : RbtEncTileT<TL2>::TS_C;
  const RBT_LDS_AS uint16_t* top = c_idx == 0 ? t->top_y : t->top_c[c_idx - 1];
  // reference samples: availability masks, substitution while gathering, smoothing, mode set-up
  const int tot = 4 * N + 1;
  if (!reuse_nb) {          // reuse_nb: r->nb still holds this block's references (en_refine_mode gathered them for the same position and size, nothing has gathered since)
    uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(N, sh), rc_nb_unit_av(t->uav, p, x0, y0, N, sh, n4));
    RcNbMap nm; rc_nb_map(&nm, m, N, sh);
    const RBT_LDS_AS uint16_t* trow = y0 ? tile + (y0 - 1) * S : top;        // the row above the TB, from its corner on: trow[x0 + k]
    RBT_PAR_FOR(i, tot) {
      int v = 1 << (bd - 1);
      if (m) { const int j = rc_nb_source(&nm, i); v = j < 2 * N ? tile[(y0 + 2 * N - 1 - j) * S + x0] : trow[x0 + j - 2 * N]; }
      r->nb[i] = v;
    }
    RBT_SYNC_LDS();
  }
  RBT_LDS_AS int32_t* fin = rc_intra_filter(g, c_idx, log2, mode, r->nb, r_nbf);
  RcIntraCtx q; rc_intra_setup(g, c_idx, log2, mode, fin, r_ref, &q);
  // prediction and residual
#define EN_BODY(PV) RBT_PAR_FOR(i, N * N) { const int x = i & (N - 1), y = i >> log2, pv = (PV); r->pred[i] = (uint16_t)pv; r->res[i] = (int16_t)((int)src[i] - pv); }
  RC_INTRA_KINDS(&q, fin, r_ref, EN_BODY);
#undef EN_BODY
  RBT_SYNC_LDS();
  // occupancy-aware coding (oracle/hevc_enc.c recon_tb): a block without an occupied sample carries no residual; in a partly occupied one the other samples ask for
  // the mean residual of the occupied ones; unoccupied samples stay out of every distortion sum below
  const int f4 = e->f4; int occ_none = 0;
#define EN_TB_OCC(i) (t->occ_u[((((y0) + ((i) >> log2)) << sh) >> 2) * n4 + ((((x0) + ((i) & (N - 1))) << sh) >> 2)])
  if (f4) {
    int pc = 0, ps = 0;
    RBT_PAR_FOR(i, N * N) { if (EN_TB_OCC(i)) { pc++; ps += r->res[i]; } }
    const int cnt = en_wave_sum(pc, (RBT_LDS_AS RbtEncLds*)0);
    occ_none = cnt == 0;
    if (cnt > 0 && cnt < N * N) {
      const int mean = en_round_mean(en_wave_sum(ps, (RBT_LDS_AS RbtEncLds*)0), cnt);
      RBT_PAR_FOR(i, N * N) { if (!EN_TB_OCC(i)) r->res[i] = (int16_t)mean; }
      RBT_SYNC_LDS();
    }
  }
  // ts_out (4x4 luma blocks of a stream with transform_skip_enabled_flag, lvl_buf with 64 entries): the block is also coded without the transform and the
  // cheaper way kept (oracle/hevc_enc.c hm_tb_finish): residual kept at lvl_buf + 32, transform-skip levels at + 16, their reconstruction at + 48
  if (ts_out) { *ts_out = 0; RBT_PAR_FOR(i, 16) lvl[32 + i] = r->res[i]; }
  int nz;
  if (occ_none) { RBT_PAR_FOR(i, N * N) lvl[i] = 0; nz = 0; }
  else if (e->lossless) {
    int part = 0;
    RBT_PAR_FOR(i, N * N) { lvl[i] = r->res[i]; part += r->res[i] != 0; }
    nz = en_wave_sum(part, (RBT_LDS_AS RbtEncLds*)0);
  } else {
    en_fwd_transform(log2, c_idx == 0 && log2 == 2, bd, r);
    const int qbits = 14 + qp / 6 + (15 - bd - log2), sc = en_quant_scale(qp % 6); int part = 0;
    const int rq = e->enc_tools & RBT_ET_RQ;
    RBT_PAR_FOR(i, N * N) {
      const int cv = r->res[i], a = rbt_abs(cv);
      const long long tq = (long long)a * sc; const int lf = (int)(tq >> qbits);
      const int off = rq ? en_rq_offset(lf == 0, lf == 1, (i & (N - 1)) + (i >> log2)) : 171;      // oracle/hevc_enc.c e1_quant_intra
      long long qv = (tq + ((long long)off << (qbits - 9))) >> qbits;
      if (qv > 32767) qv = 32767;
      lvl[i] = (int16_t)(cv < 0 ? -qv : qv);
      part += qv != 0;
    }
    nz = en_wave_sum(part, (RBT_LDS_AS RbtEncLds*)0);
  }
  RBT_SYNC_LDS();
  if (nz && !e->lossless) {
    const int bd_shift = bd + log2 - 5, scale = (16 * rc_level_scale(qp % 6)) << (qp / 6);
    const long long add = 1ll << (bd_shift - 1);
    RBT_PAR_FOR(i, N * N) { long long v = ((long long)lvl[i] * scale + add) >> bd_shift; r->res[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
    RBT_SYNC_LDS();
    rc_inv_transform(log2, c_idx == 0 && log2 == 2, 0, bd, r);
  } else if (nz) {
    RBT_PAR_FOR(i, N * N) r->res[i] = lvl[i];
    RBT_SYNC_LDS();
  }
  if (ts_out && !e->lossless && !occ_none) {
    // transform skip (7.3.8.11 transform_skip_flag, 8.6.4.2): the residual scaled by 2^(15 - bitDepth - 2) is quantised like coefficients; both ways are
    // priced as distortion * 256 + lambda^2 * rate, the flag costs one bit more, and transform skip needs a non-zero level
    const int tsh = 15 - bd - 2, qbits = 14 + qp / 6 + (15 - bd - log2), sc = en_quant_scale(qp % 6), bd_shift = bd + log2 - 5, scale = (16 * rc_level_scale(qp % 6)) << (qp / 6), ish = 20 - bd;
    const long long qadd = (long long)171 << (qbits - 9), dadd = 1ll << (bd_shift - 1);
    int p_nz = 0, p_d0 = 0, p_d1 = 0, p_b0 = 0, p_b1 = 0;
    RBT_PAR_FOR(i, 16) {
      const int rs = lvl[32 + i], cv = rbt_clip3(-32768, 32767, rs << tsh), a = rbt_abs(cv);
      long long qv = ((long long)a * sc + qadd) >> qbits;
      if (qv > 32767) qv = 32767;
      const int lt = (int)(cv < 0 ? -qv : qv);
      long long dq = ((long long)lt * scale + dadd) >> bd_shift; dq = dq < -32768 ? -32768 : (dq > 32767 ? 32767 : dq);
      const int r1 = (int)((((int)dq << 7) + (1 << (ish - 1))) >> ish), r0 = nz ? (int)r->res[i] : 0, a0 = rbt_abs((int)lvl[i]);
      lvl[16 + i] = (int16_t)lt; lvl[48 + i] = (int16_t)r1;
      p_nz += lt != 0; if (!f4 || EN_TB_OCC(i)) { p_d0 += (rs - r0) * (rs - r0); p_d1 += (rs - r1) * (rs - r1); }
      if (a0) p_b0 += 3 + 2 * (31 - __builtin_clz((unsigned)a0));
      if (qv) p_b1 += 3 + 2 * (31 - __builtin_clz((unsigned)qv));
    }
    const int nz1 = en_wave_sum(p_nz, (RBT_LDS_AS RbtEncLds*)0), d0 = en_wave_sum(p_d0, (RBT_LDS_AS RbtEncLds*)0), d1 = en_wave_sum(p_d1, (RBT_LDS_AS RbtEncLds*)0);
    const int b0 = en_wave_sum(p_b0, (RBT_LDS_AS RbtEncLds*)0), b1 = en_wave_sum(p_b1, (RBT_LDS_AS RbtEncLds*)0);
    const long long c0 = (long long)d0 * 256 + (long long)lam2 * (b0 ? b0 + 3 : 1), c1 = (long long)d1 * 256 + (long long)lam2 * ((b1 ? b1 + 3 : 1) + 1);
    RBT_SYNC_LDS();
    if (nz1 && c1 < c0) {
      RBT_PAR_FOR(i, 16) { lvl[i] = lvl[16 + i]; r->res[i] = lvl[48 + i]; }
      nz = nz1; *ts_out = 1;
      RBT_SYNC_LDS();
    }
  }
  if (mark_l4 != -2) {                                  // -2: a trial whose block is coded again, or replaced, before anything reads it (the coded mode trial's runner-up): only its cost
    { int16_t* cp = e->coef[c_idx] + (size_t)gy * pw + gx; RBT_PAR_FOR(i, N * N) cp[(size_t)(i >> log2) * pw + (i & (N - 1))] = lvl[i]; }
    RBT_PAR_FOR(i, N * N) {
      const int x = i & (N - 1), y = i >> log2;
      tile[(y0 + y) * S + x0 + x + 1] = (uint16_t)(nz ? rbt_clip3(0, maxv, (int)r->pred[i] + r->res[i]) : r->pred[i]);
    }
  }
  if (mark_l4 >= 0) { RBT_PAR_FOR(i, 1 << (2 * mark_l4)) t->uav[(muy + (i >> mark_l4) + 1) * RC_US + mux + (i & ((1 << mark_l4) - 1)) + 1] = 1; }
  if (cost) {
    int ps = 0, pb = 0;
    RBT_PAR_FOR(i, N * N) {
      const int d = (int)src[i] - (int)r->pred[i] - (nz ? (int)r->res[i] : 0), a = rbt_abs((int)lvl[i]);
      if (!f4 || EN_TB_OCC(i)) ps += d * d;                              // at most 16 samples per lane and 1024 per block: fits 32 bits
      if (a) pb += 3 + 2 * (31 - __builtin_clz((unsigned)a));
    }
    const int ssd = en_wave_sum(ps, (RBT_LDS_AS RbtEncLds*)0), bits = en_wave_sum(pb, (RBT_LDS_AS RbtEncLds*)0);
    *cost = (long long)ssd * 256 + (long long)lam2 * (bits ? bits + 3 : 1);
    if (ssd_out) *ssd_out = ssd;
  }
#undef EN_TB_OCC
  RBT_SYNC_LDS();
  return nz != 0;
}
// Closed-loop choice of a CU's luma intra mode (oracle/hevc_enc.c e1_refine_mode): the analysis' mode (chosen open loop, from source neighbours), the three
// most probable modes (8.4.2: ca / cb = candIntraPredModeA / B, the modes of the coded CUs to the left and above, DC where there is none in reach), planar
// and DC - distinct ones, in that order - predicted from the RECONSTRUCTED neighbours in the tile; cost = 16 * SATD + lambda * bits (2 for the first most
// probable mode, 3 for the other two, 6 otherwise), ties keep the earlier candidate. src: the CU's source samples, row stride N.
// *second: the runner-up (the first of the cheapest among the other candidates; -1 if there is only one), *bits_best / *bits_second: their mode bits - for the coded trial
template <int TL2> RBT_DEV int en_refine_mode(const RbtStreamCfg* g, RBT_LDS_AS RbtEncTileLdsT<TL2>* L, int x0, int y0, int lg, int an_mode, int ca, int cb, int lam16, const RBT_LDS_AS uint16_t* src, int* second, int* bits_best, int* bits_second) {
  RBT_LDS_AS RbtEncIntraScratch* r = &L->rc; RBT_LDS_AS RbtEncTileT<TL2>* t = &L->t;
  RBT_LDS_AS int32_t* const r_nbf = (RBT_LDS_AS int32_t*)r->tmp; RBT_LDS_AS int32_t* const r_ref = r_nbf + 132;
  const int N = 1 << lg, bd = g->bit_depth, n4 = (1 << g->log2_ctb) >> 2, S = RbtEncTileT<TL2>::TS_Y;
  int m0, m1, m2;
  if (ca == cb) { if (ca < 2) { m0 = 0; m1 = 1; m2 = 26; } else { m0 = ca; m1 = 2 + ((ca + 29) & 31); m2 = 2 + ((ca - 1) & 31); } }
  else { m0 = ca; m1 = cb; m2 = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26); }
  uint64_t cand = 0; int nc = 0;
  { const int pre[6] = {an_mode, m0, m1, m2, 0, 1};
#pragma unroll
    for (int i = 0; i < 6; i++) { int dup = 0; for (int k = 0; k < nc; k++) dup |= (int)((cand >> (6 * k)) & 63) == pre[i]; if (!dup) { cand |= (uint64_t)pre[i] << (6 * nc); nc++; } } }
  // reference samples from the reconstruction (as en_tile_intra_tb gathers them) and their smoothed copy, once for all candidates
  const int tot = 4 * N + 1;
  uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(N, 0), rc_nb_unit_av(t->uav, p, x0, y0, N, 0, n4));
  RcNbMap nm; rc_nb_map(&nm, m, N, 0);
  const RBT_LDS_AS uint16_t* trow = y0 ? t->y + (y0 - 1) * S : t->top_y;
  RBT_PAR_FOR(i, tot) {
    int v = 1 << (bd - 1);
    if (m) { const int j = rc_nb_source(&nm, i); v = j < 2 * N ? t->y[(y0 + 2 * N - 1 - j) * S + x0] : trow[x0 + j - 2 * N]; }
    r->nb[i] = v;
  }
  RBT_SYNC_LDS();
  rc_intra_filter_apply(g, lg, r->nb, r_nbf);
  int best = 0x7FFFFFFF, bm = an_mode, sec = 0x7FFFFFFF, sm = -1, bb = 6, sb = 6; const int tw = N >> 3;
  for (int k = 0; k < nc; k++) {
    const int mode = (int)((cand >> (6 * k)) & 63);
    RBT_LDS_AS int32_t* fin = rc_intra_filter_needed(0, lg, mode) ? r_nbf : r->nb;
    RcIntraCtx q; rc_intra_setup(g, 0, lg, mode, fin, r_ref, &q);
    int acc = 0;
#define EN_BODY(PV) for (int tix = 0; tix < tw * tw; tix++) { \
      const int tx = (tix % tw) * 8, ty = (tix / tw) * 8; \
      RBT_VEC(int, v_r); \
      RBT_VFOR(p, 64) { const int x = tx + (p & 7), y = ty + (p >> 3); RBT_V(v_r, p) = (int)src[y * N + x] - (PV); } \
      EN_HAD8X8_ACC(v_r, acc); }
    RC_INTRA_KINDS(&q, fin, r_ref, EN_BODY);
#undef EN_BODY
    const int bits = mode == m0 ? 2 : (mode == m1 || mode == m2) ? 3 : 6;
    const int c = ((en_wave_sum(acc, (RBT_LDS_AS RbtEncLds*)0) + 4) >> 3) * 16 + lam16 * bits;
    if (c < best) { sec = best; sm = bm; sb = bb; best = c; bm = mode; bb = bits; }
    else if (c < sec) { sec = c; sm = mode; sb = bits; }
    RBT_SYNC_LDS();                                     // r_ref is rebuilt by the next candidate
  }
  *second = nc >= 2 ? sm : -1; *bits_best = bb; *bits_second = sb;
  return bm;
}
// Luma of one intra CU of a stream with max_transform_hierarchy_depth_intra = 1 (RBT-E1, not lossless): coded as one transform block, then as four
// (each quarter predicted from the reconstruction so far, quarters before it included) unless one block already codes it to within lambda^2 / 4 per
// sample, and the cheaper way is kept (oracle/hevc_enc.c
// hm_decide_tu_split: luma only, distortion * 256 + lambda^2 * rate, 3 lambda^2 for the split). Returns the luma cbf of the CU, or with *split = 1 the
// cbf of quarter i in bit i. On return the tile holds the chosen reconstruction, the coefficient plane the chosen levels and every 4x4 unit of the CU
// is marked available.
// have_whole: the coded mode trial has just coded the CU as one block with this mode (levels in lv0, reconstruction in the tile, cost / distortion / cbf handed in)
template <int TL2> RBT_DEV int en_intra_cu_luma(const RbtStreamCfg* g, RbtFrame* f, RBT_LDS_AS RbtEncTileLdsT<TL2>* L, int x0, int y0, int gx, int gy, int lg, int mode, int qp, int lam2, int* split, int* ts_bits,
                                              int have_whole = 0, long long c_whole_in = 0, int ssd0_in = 0, int cbf0_in = 0, int reuse_nb = 0) {
  const EnCtbCtx* e = static_cast<const EnCtbCtx*>(g);                 // every caller hands an EnCtbCtx in (en_intra_ctb)
  RBT_LDS_AS RbtEncTileT<TL2>* t = &L->t;
  const int N = 1 << lg, h = N >> 1, S = RbtEncTileT<TL2>::TS_Y;
  long long c_whole = c_whole_in, c_split = 3ll * lam2, cq = 0;
  int ssd0 = ssd0_in, cbf0 = cbf0_in;
  if (!have_whole) cbf0 = en_tile_intra_tb(g, f, L, 0, x0, y0, gx, gy, lg, mode, qp, t->sb, -1, 0, 0, t->lv0, &c_whole, lam2, &ssd0, nullptr, reuse_nb);
  *split = 0; *ts_bits = 0;
  if ((RBT_ABLATE & 0x10000) || (!e->lossless && (long long)ssd0 * 256 < (long long)(lam2 >> 2) * N * N)) {     // coded to within lambda^2 / 4 per sample by one transform: not tried as four (lossless: the bits alone decide)
    RBT_PAR_FOR(i, 1 << (2 * (lg - 2))) t->uav[((y0 >> 2) + (i >> (lg - 2)) + 1) * RC_US + (x0 >> 2) + (i & ((1 << (lg - 2)) - 1)) + 1] = 1;
    RBT_SYNC_LDS();
    return cbf0;
  }
  RBT_PAR_FOR(i, N * N) t->rec0[i] = t->y[(y0 + (i >> lg)) * S + x0 + (i & (N - 1)) + 1];
  RBT_SYNC_LDS();
  int cbf1 = 0, tsm = 0;
  for (int b = 0; b < 4; b++) {
    const int ox = (b & 1) * h, oy = (b >> 1) * h;
    RBT_PAR_FOR(i, h * h) t->ss[i] = t->sb[(oy + (i >> (lg - 1))) * N + ox + (i & (h - 1))];
    RBT_SYNC_LDS();
    int ts = 0;
    if (en_tile_intra_tb(g, f, L, 0, x0 + ox, y0 + oy, gx + ox, gy + oy, lg - 1, mode, qp, t->ss, lg - 3, (x0 + ox) >> 2, (y0 + oy) >> 2, t->lv1, &cq, lam2, nullptr, (lg == 3 && g->transform_skip) ? &ts : nullptr)) cbf1 |= 1 << b;
    tsm |= ts << b;
    c_split += cq;
  }
  *split = c_split < c_whole;
  if (*split) { *ts_bits = tsm; return cbf1; }
  RBT_PAR_FOR(i, N * N) t->y[(y0 + (i >> lg)) * S + x0 + (i & (N - 1)) + 1] = t->rec0[i];
  { int16_t* cp = e->coef[0] + (size_t)gy * g->w + gx; RBT_PAR_FOR(i, N * N) cp[(size_t)(i >> lg) * g->w + (i & (N - 1))] = t->lv0[i]; }
  RBT_SYNC_LDS();
  return cbf0;
}
// Cb and Cr TB of one CU in the same passes (see rc_tile_tb_cpair); returns cbf_cb | cbf_cr << 1. src: Cb block, then Cr at +256.
template <int TL2> RBT_DEV int en_tile_intra_tb_cpair(const RbtStreamCfg* g, RbtFrame* f, RBT_LDS_AS RbtEncTileLdsT<TL2>* L, int x0, int y0, int gx, int gy, int log2, int mode, int qp_cb, int qp_cr,
                                   const RBT_LDS_AS uint16_t* src) {
  const EnCtbCtx* e = static_cast<const EnCtbCtx*>(g);                 // every caller hands an EnCtbCtx in (en_intra_ctb)
  RBT_LDS_AS RbtEncIntraScratch* r = &L->rc; RBT_LDS_AS RbtEncTileT<TL2>* t = &L->t;
  RBT_LDS_AS int32_t* const r_ref = (RBT_LDS_AS int32_t*)r->tmp + 132; RBT_LDS_AS int32_t* const r_ref2 = r_ref + 100; RBT_LDS_AS int16_t* const lvl = (RBT_LDS_AS int16_t*)t->sb;
  const int N = 1 << log2, NN = N * N, bd = g->bit_depth, maxv = (1 << bd) - 1, n4 = (1 << g->log2_ctb) >> 2, pw = g->cw, S = RbtEncTileT<TL2>::TS_C;
  const int tot = 4 * N + 1;
  uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(N, 1), rc_nb_unit_av(t->uav, p, x0, y0, N, 1, n4));
  RcNbMap nm; rc_nb_map(&nm, m, N, 1);
  RBT_PAR_FOR(i, 2 * tot) {
    const int b = i >= tot, idx = i - b * tot;
    int v = 1 << (bd - 1);
    if (m) {
      const int j = rc_nb_source(&nm, idx);
      const RBT_LDS_AS uint16_t* tile = t->c[b]; const RBT_LDS_AS uint16_t* trow = y0 ? tile + (y0 - 1) * S : t->top_c[b];
      v = j < 2 * N ? tile[(y0 + 2 * N - 1 - j) * S + x0] : trow[x0 + j - 2 * N];
    }
    r->nb[b * 66 + idx] = v;
  }
  RBT_SYNC_LDS();
  RcIntraCtx q0, q1;
  q0.N = N; q0.log2 = log2; q0.mode = mode; q0.c_idx = 1; q0.maxv = maxv; q0.ang = 0; q0.ver = mode >= 18; q0.dc = 0; q0.edge = 0; q1 = q0; q1.c_idx = 2;
  if (mode == 1) {
    int s0, s1; rc_dc_pair(r->nb, N, bd, &s0, &s1);
    q0.dc = s0 >> (log2 + 1); q1.dc = s1 >> (log2 + 1);
  } else if (mode >= 2) {
    const int ang = rc_intra_angle(mode), ver = mode >= 18, last = (N * ang) >> 5, inv = (mode >= 11 && mode <= 25) ? rc_intra_inv_angle(mode) : 0;
    q0.ang = q1.ang = ang;
    RBT_PAR_FOR(i, 2 * (3 * N + 1)) {
      const int b = i >= 3 * N + 1, x = i - b * (3 * N + 1) - N; int v = 0;
      const RBT_LDS_AS int32_t* nb = r->nb + b * 66;
#define RC_LEFT(y) nb[2 * N - 1 - (y)]
#define RC_TOP(x) nb[2 * N + 1 + (x)]
      if (x >= 0 && x <= N) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
      else if (x < 0) { if (ang < 0 && last < -1 && x >= last) { int k = -1 + ((x * inv + 128) >> 8); v = ver ? RC_LEFT(k) : RC_TOP(k); } }
      else if (ang >= 0) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
#undef RC_LEFT
#undef RC_TOP
      (b ? r_ref2 : r_ref)[x + 32] = v;
    }
    RBT_SYNC_LDS();
  }
  // prediction and residual of both planes (lane i: sample j of plane b; plane 1's references 66, its angular array 100 entries behind plane 0's)
#define EN_BODY(PV) RBT_PAR_FOR(i, 2 * NN) { \
    const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2; \
    const RcIntraCtx* qp = b ? &q1 : &q0; const RBT_LDS_AS int32_t* nbp = r->nb + b * 66; const RBT_LDS_AS int32_t* refp = r_ref + b * 100; \
    const int pv = (PV); r->pred[b * 256 + j] = (uint16_t)pv; r->res[b * 256 + j] = (int16_t)((int)src[b * 256 + j] - pv); }
  RC_INTRA_KINDS_PAIR(&q0, qp, nbp, refp, EN_BODY);
#undef EN_BODY
  RBT_SYNC_LDS();
  // occupancy-aware coding (see en_tile_intra_tb): chroma sample (x,y) stands for the luma samples (2x..2x+1, 2y..2y+1), all in one 4x4 unit
  int occ_none = 0;
  if (e->f4) {
#define EN_TBC_OCC(j) (t->occ_u[((y0 + ((j) >> log2)) >> 1) * n4 + ((x0 + ((j) & (N - 1))) >> 1)])
    int pc = 0, ps0 = 0, ps1 = 0;
    RBT_PAR_FOR(j, NN) { if (EN_TBC_OCC(j)) { pc++; ps0 += r->res[j]; ps1 += r->res[256 + j]; } }
    const int cnt = en_wave_sum(pc, (RBT_LDS_AS RbtEncLds*)0);
    occ_none = cnt == 0;
    if (cnt > 0 && cnt < NN) {
      const int m0v = en_round_mean(en_wave_sum(ps0, (RBT_LDS_AS RbtEncLds*)0), cnt), m1v = en_round_mean(en_wave_sum(ps1, (RBT_LDS_AS RbtEncLds*)0), cnt);
      RBT_PAR_FOR(j, NN) { if (!EN_TBC_OCC(j)) { r->res[j] = (int16_t)m0v; r->res[256 + j] = (int16_t)m1v; } }
      RBT_SYNC_LDS();
    }
#undef EN_TBC_OCC
  }
  int part = 0;                                                          // non-zero counts: Cb in the low half, Cr in the high half
  if (occ_none) { RBT_PAR_FOR(i, 2 * NN) { const int b = i >= NN, j = i - b * NN; lvl[b * 256 + j] = 0; } }
  else if (e->lossless) {
    RBT_PAR_FOR(i, 2 * NN) { const int b = i >= NN, j = i - b * NN; lvl[b * 256 + j] = r->res[b * 256 + j]; part += (r->res[b * 256 + j] != 0) << (16 * b); }
  } else {
    if (log2 == 2) en_fwd_transform_pair_n<2>(bd, r); else if (log2 == 3) en_fwd_transform_pair_n<3>(bd, r); else en_fwd_transform_pair_n<4>(bd, r);
    const int qb_cb = 14 + qp_cb / 6 + (15 - bd - log2), qb_cr = 14 + qp_cr / 6 + (15 - bd - log2), sc_cb = en_quant_scale(qp_cb % 6), sc_cr = en_quant_scale(qp_cr % 6);
    const int rq = e->enc_tools & RBT_ET_RQ;
    RBT_PAR_FOR(i, 2 * NN) {
      const int b = i >= NN, j = i - b * NN, qbits = b ? qb_cr : qb_cb;
      const int cv = r->res[b * 256 + j], a = rbt_abs(cv);
      const long long tq = (long long)a * (b ? sc_cr : sc_cb); const int lf = (int)(tq >> qbits);
      const int off = rq ? en_rq_offset(lf == 0, lf == 1, (j & (N - 1)) + (j >> log2)) : 171;
      long long qv = (tq + ((long long)off << (qbits - 9))) >> qbits;
      if (qv > 32767) qv = 32767;
      lvl[b * 256 + j] = (int16_t)(cv < 0 ? -qv : qv);
      part += (qv != 0) << (16 * b);
    }
  }
  const int nzp = en_wave_sum(part, (RBT_LDS_AS RbtEncLds*)0), nz0 = nzp & 0xFFFF, nz1 = nzp >> 16;
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, 2 * NN) { const int b = i >= NN, j = i - b * NN; (b ? e->coef[2] : e->coef[1])[(size_t)(gy + (j >> log2)) * pw + gx + (j & (N - 1))] = lvl[b * 256 + j]; }
  if ((nz0 | nz1) && !e->lossless) {
    const int bd_shift = bd + log2 - 5, sc_cb = (16 * rc_level_scale(qp_cb % 6)) << (qp_cb / 6), sc_cr = (16 * rc_level_scale(qp_cr % 6)) << (qp_cr / 6);
    const long long add = 1ll << (bd_shift - 1);
    RBT_PAR_FOR(i, 2 * NN) {
      const int b = i >= NN, j = i - b * NN;
      if (b ? nz1 : nz0) { long long v = ((long long)lvl[b * 256 + j] * (b ? sc_cr : sc_cb) + add) >> bd_shift; r->res[b * 256 + j] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
    }
    RBT_SYNC_LDS();
    const int sh = 20 - bd;
    if (log2 == 2) rc_inv_transform_pair_n<2>(sh, nz0, nz1, r); else if (log2 == 3) rc_inv_transform_pair_n<3>(sh, nz0, nz1, r); else rc_inv_transform_pair_n<4>(sh, nz0, nz1, r);
  } else if (nz0 | nz1) {
    RBT_PAR_FOR(i, 2 * NN) { const int b = i >= NN, j = i - b * NN; r->res[b * 256 + j] = lvl[b * 256 + j]; }
    RBT_SYNC_LDS();
  }
  RBT_PAR_FOR(i, 2 * NN) {
    const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2, nz = b ? nz1 : nz0;
    t->c[b][(y0 + y) * S + x0 + x + 1] = (uint16_t)(nz ? rbt_clip3(0, maxv, (int)r->pred[b * 256 + j] + r->res[b * 256 + j]) : r->pred[b * 256 + j]);
  }
  RBT_SYNC_LDS();
  return (nz0 != 0) | ((nz1 != 0) << 1);
}
// carry_left: the CTB to the left was coded by this wave just before (its reconstruction is still in the tile): take the
// left border from LDS instead of reading back stores that may still be in flight
template <int TL2> RBT_DEV void en_intra_ctb(RbtFrame* f, const RbtSlice* slices, int ctb_addr, RBT_LDS_AS RbtEncTileLdsT<TL2>* L, int carry_left) {
  EnCtbCtx gcopy; en_ctb_ctx(&gcopy, f); const RbtStreamCfg* g = &gcopy; const EnCtbCtx* e = &gcopy;
  RBT_LDS_AS RbtEncTileT<TL2>* t = &L->t;
  const int ctb = 1 << g->log2_ctb, n4 = ctb >> 2, n8 = ctb >> 3, rx = ctb_addr % g->w_ctb, ry = ctb_addr / g->w_ctb, cx = rx << g->log2_ctb, cy = ry << g->log2_ctb;
  const RbtSlice* sl = &slices[f->ctb_slice[ctb_addr]];
  const int qp_y = sl->qp, bd = g->bit_depth;
  const int qp_l = qp_y + 6 * (bd - 8), qp_cb = en_chroma_qp(f, sl, 1, qp_y), qp_cr = en_chroma_qp(f, sl, 2, qp_y);
  const int tu_rd = g->th_depth_intra > 0, lam16 = k_lambda16[rbt_clip3(0, 75, qp_l)], lam2 = lam16 * lam16;
  // ---- borders, unit availability, analysis results ----
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, nn = ctb >> sh, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, ox = cx >> sh, oy = cy >> sh, S = c ? RbtEncTileT<TL2>::TS_C : RbtEncTileT<TL2>::TS_Y;
    const uint16_t* p = f->pix[c]; RBT_LDS_AS uint16_t* tile = c == 0 ? t->y : t->c[c - 1]; RBT_LDS_AS uint16_t* top = c == 0 ? t->top_y : t->top_c[c - 1];
    if (carry_left) { RBT_PAR_FOR(i, nn) tile[i * S] = tile[i * S + nn]; RBT_SYNC_LDS(); }
    else { RBT_PAR_FOR(i, nn) { int x = ox - 1, y = oy + i; tile[i * S] = (x >= 0 && y < ph) ? p[(size_t)y * pw + x] : 0; } }
    RBT_PAR_FOR(i, 2 * nn + 1) { int x = ox + i - 1, y = oy - 1; top[i] = (x >= 0 && y >= 0 && x < pw) ? p[(size_t)y * pw + x] : 0; }
  }
  RBT_PAR_FOR(i, ((1 << (TL2 - 2)) + 1) * RC_US) {
    int ux = i % RC_US - 1, uy = i / RC_US - 1, a = 0;
    if ((uy < 0 && ux < 2 * n4) || (ux < 0 && uy < n4)) a = rc_unit_avail(f, ctb_addr, (cx >> 2) + ux, (cy >> 2) + uy);
    t->uav[i] = (uint8_t)a;
  }
  if (e->f4) { RBT_PAR_FOR(i, n4 * n4) t->occ_u[i] = (uint8_t)en_occ_unit(f, (cx >> 2) + i % n4, (cy >> 2) + i / n4); }
  const int refine = (e->enc_tools & RBT_ET_REFINE) && !(RBT_ABLATE & 0x4000), rdm = refine && tu_rd && !e->lossless && (e->enc_tools & RBT_ET_RDM) && !(RBT_ABLATE & 0x8000);
  if (refine) {
    // modes of the CUs along the left border (candIntraPredModeA of this CTB's first column): carried in LDS when this wave has just coded that CTB
    const int left_ok = rx > 0 && f->ctb_slice[ctb_addr - 1] == f->ctb_slice[ctb_addr];
    if (carry_left) { RBT_PAR_FOR(i, n8) t->left_md[i] = t->cu_md[i * 8 + n8 - 1]; }
    else { RBT_PAR_FOR(i, n8) { const int y = cy + i * 8; t->left_md[i] = (uint8_t)((left_ok && y < g->h) ? f->cu_mode[(y >> 3) * f->w8 + ((cx - 8) >> 3)] : 1); } }
    if (RBT_LANE0) t->left_md[8] = (uint8_t)left_ok;
    RBT_SYNC_LDS();
  }
  RBT_PAR_FOR(i, n8 * n8) {
    const int ux = i & (n8 - 1), uy = i / n8, x = cx + ux * 8, y = cy + uy * 8;
    int l2 = 3, md = 1;
    if (x < g->w && y < g->h) { const int k = (y >> 3) * f->w8 + (x >> 3); l2 = f->cu_log2[k]; md = f->cu_mode[k]; }
    t->cu_l2[uy * 8 + ux] = (uint8_t)l2; t->cu_md[uy * 8 + ux] = (uint8_t)md;
  }
  RBT_SYNC();
  // leaf CUs in z-order: walk 8x8 units in Morton order, a CU is coded when its first unit is reached
  for (int z = 0; z < n8 * n8; z++) {
    int ux = 0, uy = 0;
    for (int b = 0; b < 3; b++) { ux |= ((z >> (2 * b)) & 1) << b; uy |= ((z >> (2 * b + 1)) & 1) << b; }
    const int x0 = ux * 8, y0 = uy * 8;                              // relative to the CTB
    if (cx + x0 >= g->w || cy + y0 >= g->h) continue;
    const int lg = RBT_UNI(t->cu_l2[uy * 8 + ux]), N = 1 << lg, Nc = N >> 1;   // wave-uniform by construction; said so, or sizes, positions and every trip count below live in vector registers
    if ((x0 & (N - 1)) || (y0 & (N - 1))) continue;
    int mode = RBT_UNI(t->cu_md[uy * 8 + ux]);
    // source samples of the CU's three TBs: one HBM round trip
    { const uint16_t* sp = e->src[0] + (size_t)(cy + y0) * g->w + cx + x0; RBT_PAR_FOR(i, N * N) t->sb[i] = sp[(size_t)(i >> lg) * g->w + (i & (N - 1))]; }
    for (int q = 0; q < 2; q++) { const uint16_t* sp = (q ? e->src[2] : e->src[1]) + (size_t)((cy + y0) >> 1) * g->cw + ((cx + x0) >> 1); RBT_PAR_FOR(i, Nc * Nc) t->sb[1024 + 256 * q + i] = sp[(size_t)(i >> (lg - 1)) * g->cw + (i & (Nc - 1))]; }
    RBT_SYNC();
    int have_w = 0, ssd_w = 0, cbf_w = 0; long long c_w = 0;
    if (refine) {
      const int ca = x0 > 0 ? (int)t->cu_md[uy * 8 + ux - 1] : (t->left_md[8] ? (int)t->left_md[uy] : 1), cb = y0 > 0 ? (int)t->cu_md[(uy - 1) * 8 + ux] : 1;   // above: inside this CTB only (8.4.2)
      int second, b1, b2;
      mode = en_refine_mode<TL2>(g, L, x0, y0, lg, mode, RBT_UNI(ca), RBT_UNI(cb), lam16, t->sb, &second, &b1, &b2);
      if (rdm && second >= 0 && lg >= 4) {
        // the SATD says which two modes to look at, the coded block which of them to take (oracle/hevc_enc.c e1_mode_trial; ties: the SATD's choice; 16x16 and 32x32 CUs
        // only: on 8x8 CUs, more than half of all, the trial moved nothing). The runner-up first:
        // when the SATD's choice stands - most of the time - its block is already coded and en_intra_cu_luma goes straight on to the four-way form
        long long c2 = 0;
        en_tile_intra_tb(g, f, L, 0, x0, y0, cx + x0, cy + y0, lg, second, qp_l, t->sb, -2, 0, 0, t->lv0, &c2, lam2, nullptr, nullptr, 1);
        cbf_w = en_tile_intra_tb(g, f, L, 0, x0, y0, cx + x0, cy + y0, lg, mode, qp_l, t->sb, -1, 0, 0, t->lv0, &c_w, lam2, &ssd_w, nullptr, 1);
        if (c2 + (long long)lam2 * b2 < c_w + (long long)lam2 * b1) mode = second; else have_w = 1;
      }
      const int nu = N >> 3;
      RBT_PAR_FOR(i, nu * nu) { const int vx = ux + i % nu, vy = uy + i / nu; t->cu_md[vy * 8 + vx] = (uint8_t)mode; e->cu_mode[(((cy + y0) >> 3) + i / nu) * e->w8 + ((cx + x0) >> 3) + i % nu] = (uint8_t)mode; }
      RBT_SYNC_LDS();
    }
    int split = 0, cbf = 0, cy4 = 0, ts_bits = 0;
    if (tu_rd) cy4 = en_intra_cu_luma(g, f, L, x0, y0, cx + x0, cy + y0, lg, mode, qp_l, lam2, &split, &ts_bits, have_w, c_w, ssd_w, cbf_w, refine != 0);      // after the mode choice r->nb holds the CU's references
    else cy4 = en_tile_intra_tb(g, f, L, 0, x0, y0, cx + x0, cy + y0, lg, mode, qp_l, t->sb, lg - 2, x0 >> 2, y0 >> 2);
    if (split && lg >= 4) {
      // four transform units, each with its own Cb / Cr blocks: chroma block b is predicted when the units 0..b of the CU are reconstructed, not more
      const int hh = N >> 1, hc = Nc >> 1, n4u = hh >> 2;
      RBT_PAR_FOR(i, 4 * n4u * n4u) { const int ux = i % (2 * n4u), uy = i / (2 * n4u); t->uav[((y0 >> 2) + uy + 1) * RC_US + (x0 >> 2) + ux + 1] = 0; }
      RBT_SYNC_LDS();
      for (int b = 0; b < 4; b++) {
        const int ox = (b & 1) * hh, oy = (b >> 1) * hh;
        RBT_PAR_FOR(i, n4u * n4u) t->uav[(((y0 + oy) >> 2) + i / n4u + 1) * RC_US + ((x0 + ox) >> 2) + i % n4u + 1] = 1;
        RBT_PAR_FOR(i, 2 * hc * hc) { const int q = i >= hc * hc, j = i - q * hc * hc; t->ss[q * 256 + j] = t->sb[1024 + 256 * q + ((oy >> 1) + j / hc) * Nc + (ox >> 1) + j % hc]; }
        RBT_SYNC_LDS();
        const int cc = en_tile_intra_tb_cpair(g, f, L, (x0 + ox) >> 1, (y0 + oy) >> 1, (cx + x0 + ox) >> 1, (cy + y0 + oy) >> 1, lg - 2, mode, qp_cb, qp_cr, t->ss);
        const int fl = RBT_CU_TU_SPLIT | ((cy4 >> b) & 1 ? RBT_CU_CBF_Y : 0) | ((cc & 1) ? RBT_CU_CBF_CB : 0) | ((cc & 2) ? RBT_CU_CBF_CR : 0);
        en_fill_cu_maps_ctx(e, cx + x0 + ox, cy + y0 + oy, hh, RBT_MODE_INTRA | ((fl & RBT_CU_CBF_Y) ? RBT_PM_NZ : 0), qp_y, fl);
      }
      continue;
    }
    { const int cc = en_tile_intra_tb_cpair(g, f, L, x0 >> 1, y0 >> 1, (cx + x0) >> 1, (cy + y0) >> 1, lg - 1, mode, qp_cb, qp_cr, t->sb + 1024);
      if (cc & 1) cbf |= RBT_CU_CBF_CB;
      if (cc & 2) cbf |= RBT_CU_CBF_CR; }
    if (split) cbf |= RBT_CU_TU_SPLIT | ((cy4 & 1) ? RBT_CU_CBF_Y : 0) | ((cy4 >> 1) * RBT_CU_CBF_Y1);   // 8x8 CU as four 4x4 luma blocks: their cbf bits
    else if (cy4) cbf |= RBT_CU_CBF_Y;
    en_fill_cu_maps_ctx(e, cx + x0, cy + y0, N, RBT_MODE_INTRA | (e->lossless ? RBT_PM_TQ_BYPASS : 0) | (cy4 ? RBT_PM_NZ : 0), qp_y, cbf);
    if (lg == 3 && g->transform_skip && RBT_LANE0) e->cu_ts[((cy + y0) >> 3) * e->w8 + ((cx + x0) >> 3)] = (uint8_t)ts_bits;   // transform_skip_flag of the four 4x4 luma blocks
  }
  // ---- write the CTB back (clipped to the picture) ----
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, nn = ctb >> sh, lnn = g->log2_ctb - sh, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, ox = cx >> sh, oy = cy >> sh, S = c ? RbtEncTileT<TL2>::TS_C : RbtEncTileT<TL2>::TS_Y;
    uint16_t* p = f->pix[c]; RBT_LDS_AS uint16_t* tile = c == 0 ? t->y : t->c[c - 1];
    RBT_PAR_FOR(i, nn * nn) { int x = i & (nn - 1), y = i >> lnn; if (ox + x < pw && oy + y < ph) p[(size_t)(oy + y) * pw + ox + x] = tile[y * S + x + 1]; }
  }
}

// ------------------------------------------------------------------------------------------------ P pictures: one CTB
RBT_DEV void en_inter_ctb(RbtFrame* frames, RbtFrame* f, const RbtSlice* slices, int ctb_addr, RBT_LDS_AS RbtEncLds* l) {
  const RbtStreamCfg* g = &f->cfg;
  const RbtFrame* ref = &frames[f->ref_frame];
  int ctb = 1 << g->log2_ctb, cx = (ctb_addr % g->w_ctb) << g->log2_ctb, cy = (ctb_addr / g->w_ctb) << g->log2_ctb;
  const RbtSlice* sl = &slices[f->ctb_slice[ctb_addr]];
  int qp_y = sl->qp, bd = g->bit_depth;
  int qp[3] = {qp_y + 6 * (bd - 8), en_chroma_qp(f, sl, 1, qp_y), en_chroma_qp(f, sl, 2, qp_y)};
  int n16 = ctb / 16;
  for (int b = 0; b < n16 * n16; b++) {
    int x0 = cx + (b % n16) * 16, y0 = cy + (b / n16) * 16;
    if (x0 >= g->w || y0 >= g->h) continue;
    int cbf = 0;
    for (int c = 0; c < 3; c++) {
      int sh = c ? 1 : 0, S = 16 >> sh, pw = c ? g->cw : g->w;
      const uint16_t* rp = ref->out[c];
      RBT_PAR_FOR(i, S * S) { int x = i & (S - 1), y = i / S; l->rc.pred[i] = rp[(size_t)((y0 >> sh) + y) * pw + (x0 >> sh) + x]; }
      RBT_SYNC_LDS();
      if (en_code_tb(f, c, x0 >> sh, y0 >> sh, 4 - sh, qp[c], 0, l)) cbf |= 1 << c;
    }
    int flags = cbf | (cbf ? 0 : RBT_CU_SKIP);
    // 16x16 CU maps; merged skips are rewritten below
    int n4 = 4;
    RBT_PAR_FOR(i, n4 * n4) {
      int k = ((y0 >> 2) + i / n4) * g->w4 + (x0 >> 2) + i % n4;
      f->pm[k] = (uint8_t)((cbf ? RBT_MODE_INTER : RBT_MODE_SKIP) | ((cbf & 1) ? RBT_PM_NZ : 0));
      f->qp[k] = (int8_t)qp_y; f->mv[2 * k] = 0; f->mv[2 * k + 1] = 0; f->ref[k] = 0; f->refpoc[k] = f->ref_poc;
      int e = 0; if (i % n4 == 0) e |= RBT_EV_TU | RBT_EV_PU; if (i / n4 == 0) e |= RBT_EH_TU | RBT_EH_PU;
      f->edges[k] = (uint8_t)e;
    }
    RBT_PAR_FOR(i, 4) { int k = ((y0 >> 3) + i / 2) * f->w8 + (x0 >> 3) + i % 2; f->cu_flags[k] = (uint8_t)flags; f->cu_log2[k] = 4; f->cu_mode[k] = 1; }
    RBT_SYNC();
  }
  // merge skips up the tree: a node whose four children are skip CUs becomes one skip CU
  for (int lg = 5; lg <= g->log2_ctb; lg++) {
    int S = 1 << lg, nb = ctb / S;
    for (int b = 0; b < nb * nb; b++) {
      int x0 = cx + (b % nb) * S, y0 = cy + (b / nb) * S, h = S >> 1;
      if (x0 + S > g->w || y0 + S > g->h) continue;
      int all = 1;
      for (int c = 0; c < 4; c++) {
        int k = ((y0 + (c >> 1) * h) >> 3) * f->w8 + ((x0 + (c & 1) * h) >> 3);
        if (!(f->cu_flags[k] & RBT_CU_SKIP) || f->cu_log2[k] != lg - 1) all = 0;
      }
      if (!all) continue;
      int n8 = S >> 3, n4 = S >> 2;
      RBT_PAR_FOR(i, n8 * n8) { int k = ((y0 >> 3) + i / n8) * f->w8 + (x0 >> 3) + i % n8; f->cu_log2[k] = (uint8_t)lg; }
      RBT_PAR_FOR(i, n4 * n4) {
        int k = ((y0 >> 2) + i / n4) * g->w4 + (x0 >> 2) + i % n4;
        int e = 0; if (i % n4 == 0) e |= RBT_EV_TU | RBT_EV_PU; if (i / n4 == 0) e |= RBT_EH_TU | RBT_EH_PU;
        f->edges[k] = (uint8_t)e;
      }
      RBT_SYNC();
    }
  }
}

// ------------------------------------------------------------------------------------------------ entropy coding
struct RbtEnt { RbtFrame* f; const RbtSlice* sl; int slice_idx; RbtCabacEnc c; RBT_LDS_AS RbtEntropyLds* l;
                int w, h, log2_ctb, log2_min_cb, tq_bypass_enabled, is_p, cx, cy, th_intra;
#ifdef RBT_PROFILE
                unsigned long long t_stage, t_res, t_cu, t_resA, t_resB; unsigned n_cu, n_tb, n_bins;
#endif
};
// encoder scan tables use x | y << 4 in k_scan; the lane code wants sub-block entries unchanged and 4x4 positions as packed immediates
RBT_DEV uint8_t k_scan_packed(int a, int b, int c) { return k_scan[a][b][c]; }

// 4x4 scan positions packed 4 bits per entry (x | y << 2): diagonal, horizontal, vertical; ctxIdxMap of 4x4 TBs (same
// packed immediates as the parser uses)
RBT_DEV uint64_t en_scan4_const(int scan_idx) { return scan_idx == 0 ? 0xFBE7AD369C258140ull : (scan_idx == 1 ? 0xFEDCBA9876543210ull : 0xFB73EA62D951C840ull); }
#define EN_SIGCTX4 0x8877886654325410ull
RBT_DEV int en_group_idx(int v) { if (v < 4) return v; const int lg = 31 - __builtin_clz((unsigned)v); return 2 * lg + ((v >> (lg - 1)) & 1); }   // last_sig_coeff prefix of position v
RBT_DEV int en_min_in_group(int g) { return g < 4 ? g : (2 + (g & 1)) << ((g >> 1) - 1); }
// residual_coding (7.3.8.11) of one TB. Its levels are already in LDS (`lv`, row stride N): en_load_cu_levels fetched the
// three TBs of the CU with one HBM round trip. Same split as the parser: what is not a bin runs on the lanes (lane i =
// sub-block i for the significance masks, lane p = scan position p of the current sub-block for contexts and levels), the
// serial part is bins only.
RBT_DEV void en_write_residual(RbtEnt* s, int c_idx, const RBT_LDS_AS int16_t* lv, int lst, int log2, int scan_idx, int ts = 0) {
  RbtCabacEnc* c = &s->c; RBT_LDS_AS RbtEntropyLds* l = s->l;
#ifdef RBT_PROFILE
  unsigned long long ta_ = __builtin_readcyclecounter(); s->n_tb++;
#endif
  log2 = RBT_UNI(log2); scan_idx = RBT_UNI(scan_idx); lst = RBT_UNI(lst);
  const int chroma = c_idx != 0;
  if (log2 == 2 && s->f->cfg.transform_skip && !s->f->lossless) rbt_ce_bin0(c, CTX_TRANSFORM_SKIP + chroma, RBT_UNI(ts));   // transform_skip_flag (7.3.8.11): RBT-E1 only ever sets it for luma
  const uint64_t ps = en_scan4_const(scan_idx);
  const RBT_LDS_AS uint8_t* sb_scan = l->scan[scan_idx][log2 - 2];
  const int n_sb = 1 << (2 * (log2 - 2));
  RBT_VEC(int, v_sbscan); RBT_VEC(int, v_pos); RBT_VEC(int, v_cgmask);
  RBT_VFOR(p, 64) { RBT_V(v_sbscan, p) = p < n_sb ? (int)sb_scan[p] : 0; RBT_V(v_pos, p) = (int)((ps >> (4 * (p & 15))) & 15); RBT_V(v_cgmask, p) = 0; }
  // significance masks of the sub-blocks: 64 (sub-block, scan position) pairs per pass, one level read per lane, a ballot
  // gives the masks of four sub-blocks at once
  for (int pass = 0; pass * 4 < n_sb; pass++) {
    uint64_t nzm;
    RBT_VBALLOT(nzm, p, 64, (pass * 4 + (p >> 4)) < n_sb && lv[(((int)sb_scan[pass * 4 + (p >> 4)] >> 4 << 2) + (RBT_V(v_pos, p) >> 2)) * lst + (((int)sb_scan[pass * 4 + (p >> 4)] & 15) << 2) + (RBT_V(v_pos, p) & 3)] != 0);
    for (int k = 0; k < 4 && pass * 4 + k < n_sb; k++) RBT_VSET(v_cgmask, pass * 4 + k, (int)((nzm >> (16 * k)) & 0xFFFF));
  }
  uint64_t nz64; RBT_VBALLOT(nz64, p, 64, RBT_V(v_cgmask, p) != 0);
  const int last_sb = nz64 ? 63 - __builtin_clzll(nz64) : 0;
  const int last_mask = RBT_VGET(v_cgmask, last_sb), last_pos = last_mask ? 31 - __builtin_clz((unsigned)last_mask) : 0;
  { const int e = RBT_VGET(v_sbscan, last_sb), q = (int)((ps >> (4 * last_pos)) & 15);
    int cx = ((e & 15) << 2) + (q & 3), cy = ((e >> 4) << 2) + (q >> 2);
    if (scan_idx == 2) { const int t = cx; cx = cy; cy = t; }
    int ctx_off, ctx_shift;
    if (c_idx == 0) { ctx_off = 3 * (log2 - 2) + ((log2 - 1) >> 2); ctx_shift = (log2 + 1) >> 2; }
    else { ctx_off = 15; ctx_shift = log2 - 2; }
    const int maxp = (log2 << 1) - 1, px = en_group_idx(cx), py = en_group_idx(cy);
    for (int i = 0; i < px; i++) rbt_ce_bin_last(c, ctx_off + (i >> ctx_shift), 1);
    if (px < maxp) rbt_ce_bin_last(c, ctx_off + (px >> ctx_shift), 0);
    for (int i = 0; i < py; i++) rbt_ce_bin_last(c, 18 + ctx_off + (i >> ctx_shift), 1);
    if (py < maxp) rbt_ce_bin_last(c, 18 + ctx_off + (py >> ctx_shift), 0);
    if (px > 3) rbt_ce_bypass_n(c, (uint32_t)(cx - en_min_in_group(px)), (px >> 1) - 1);
    if (py > 3) rbt_ce_bypass_n(c, (uint32_t)(cy - en_min_in_group(py)), (py >> 1) - 1); }
#ifdef RBT_PROFILE
  unsigned long long tb_ = __builtin_readcyclecounter(); s->t_resA += tb_ - ta_;
#endif
  uint64_t csbf = 0;
  const int sbw = 1 << (log2 - 2), sig_c0 = chroma ? 27 : 0;
  int greater1_ctx = 1, first_sb_done = 0;
  for (int i = last_sb; i >= 0; i--) {
    const int sbv = RBT_VGET(v_sbscan, i), xs = sbv & 15, ys = sbv >> 4;
    const int right = xs + 1 < sbw ? (int)((csbf >> (ys * 8 + xs + 1)) & 1) : 0, below = ys + 1 < sbw ? (int)((csbf >> ((ys + 1) * 8 + xs)) & 1) : 0;
    const uint32_t mask = (uint32_t)RBT_VGET(v_cgmask, i);
    int infer_dc = 0;
    if (i < last_sb && i > 0) { rbt_ce_bin_res2(c, rbt_min(right + below, 1) + (chroma ? 2 : 0), mask != 0); if (!mask) continue; infer_dc = 1; }
    csbf |= 1ull << (ys * 8 + xs);
    const int prev_csbf = right | (below << 1);
    const uint32_t pat = prev_csbf == 0 ? 0x00010516u : (prev_csbf == 1 ? 0x000055AAu : (prev_csbf == 2 ? 0x06060606u : 0xAAAAAAAAu));
    const int cg_base = !chroma ? ((xs | ys) ? 3 : 0) + (log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21) : 27 + (log2 == 3 ? 9 : 12);
    const int dc_cg = (xs | ys) == 0;
    // lane p: sig_coeff_flag context and level of scan position p of this sub-block
    RBT_VEC(int, v_sc); RBT_VEC(int, v_abs); uint64_t neg64;
    RBT_VFOR(p, 16) {
      const int p4 = RBT_V(v_pos, p);
      RBT_V(v_sc, p) = log2 == 2 ? (int)((EN_SIGCTX4 >> (4 * p4)) & 15) + sig_c0 : ((dc_cg && p4 == 0) ? sig_c0 : cg_base + (int)((pat >> (2 * p4)) & 3));
      const int v = lv[((ys << 2) + (p4 >> 2)) * lst + (xs << 2) + (p4 & 3)];
      RBT_V(v_abs, p) = v < 0 ? -v : v;
    }
    RBT_VBALLOT(neg64, p, 16, lv[((ys << 2) + (RBT_V(v_pos, p) >> 2)) * lst + (xs << 2) + (RBT_V(v_pos, p) & 3)] < 0);
    const int start = i == last_sb ? last_pos - 1 : 15;
    for (int n = start; n >= 1; n--) rbt_ce_bin_sig(c, RBT_VGET(v_sc, n), (int)((mask >> n) & 1));
    if (start >= 0 && !(infer_dc && (mask >> 1) == 0)) rbt_ce_bin_sig(c, RBT_VGET(v_sc, 0), (int)(mask & 1));
    if (!mask) continue;
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    ctx_set += first_sb_done & (int)((uint32_t)(greater1_ctx - 1) >> 31);
    first_sb_done = 1; greater1_ctx = 1;
    const int g1_base = (ctx_set << 2) + (chroma ? 16 : 0);
    int first_g1 = -1, first_g1_abs = 0, k = 0, nsig = 0;
    uint32_t m = mask, signs = 0;
    while (m && k < 8) {
      const int n = 31 - __builtin_clz(m); m &= ~(1u << n);
      const int a = RBT_VGET(v_abs, n), g1 = a > 1;
      rbt_ce_bin_res2(c, 4 + g1_base + greater1_ctx, g1);
      if (g1) { greater1_ctx = 0; if (first_g1 < 0) { first_g1 = k; first_g1_abs = a; } }
      else if (greater1_ctx > 0 && greater1_ctx < 3) greater1_ctx++;
      k++;
    }
    if (first_g1 >= 0) rbt_ce_bin_res2(c, 28 + ctx_set + (chroma ? 4 : 0), first_g1_abs > 2);
    m = mask;
    while (m) { const int n = 31 - __builtin_clz(m); m &= ~(1u << n); signs = (signs << 1) | (uint32_t)((neg64 >> n) & 1); nsig++; }
    rbt_ce_bypass_n(c, signs, nsig);      // sign_data_hiding is off in RBT-E1 streams
    int rice = 0; k = 0; m = mask;
    while (m) {
      const int n = 31 - __builtin_clz(m); m &= ~(1u << n);
      const int a = RBT_VGET(v_abs, n), base = k < 8 ? (k == first_g1 ? 3 : 2) : 1;
      if (a >= base) {
        const int v = a - base;
        if (v < (4 << rice)) { const int pre = v >> rice; rbt_ce_bypass_n(c, ((1u << pre) - 1u) << 1, pre + 1); rbt_ce_bypass_n(c, (uint32_t)(v & ((1 << rice) - 1)), rice); }
        else {
          int p = 4; while (v >= (((1 << (p - 2)) + 2) << rice)) p++;
          rbt_ce_bypass_n(c, ((1u << p) - 1u) << 1, p + 1);
          rbt_ce_bypass_n(c, (uint32_t)(v - (((1 << (p - 3)) + 2) << rice)), p - 3 + rice);
        }
        if (a > 3 * (1 << rice)) rice = rbt_min(rice + 1, 4);
      }
      k++;
    }
  }
}
RBT_DEV int en_scan_idx(int is_intra, int log2, int c_idx, int mode) {
  if (is_intra && (log2 == 2 || (log2 == 3 && c_idx == 0))) { if (mode >= 6 && mode <= 14) return 2; if (mode >= 22 && mode <= 30) return 1; }
  return 0;
}
// index of the 8x8 unit that holds luma position (x,y) in the staged CTB maps; x,y may lie one unit left of / above the CTB
RBT_DEV int en_u(const RbtEnt* s, int x, int y) { return (((y - s->cy) >> 3) + 1) * 9 + ((x - s->cx) >> 3) + 1; }
// stages cu_log2 / cu_mode / cu_flags of CTB (cx,cy) and of its left column / above row (with availability) into LDS
RBT_DEV void en_stage_ctb(RbtEnt* s, int cx, int cy) {
#ifdef RBT_PROFILE
  unsigned long long ts_ = __builtin_readcyclecounter();
#endif
  const RbtFrame* f = s->f; RBT_LDS_AS RbtEntropyLds* l = s->l;
  s->cx = cx; s->cy = cy;
  const int L = s->log2_ctb, wc = (s->w + (1 << L) - 1) >> L, ac = (cy >> L) * wc + (cx >> L), my = f->ctb_slice[ac];
  int coded = 0;                                                     // some block of THIS CTB has levels (its cbf bits)
  RBT_PAR_FOR(i, 81) {
    const int ux = i % 9 - 1, uy = i / 9 - 1, x = cx + ux * 8, y = cy + uy * 8;
    int l2 = 0xFF, md = 1, fl = 0;
    if (x >= 0 && y >= 0 && x < s->w && y < s->h && ux < (1 << (L - 3)) && uy < (1 << (L - 3))) {
      const int an = (y >> L) * wc + (x >> L);
      if (an == ac || f->ctb_slice[an] == my) { const int k = (y >> 3) * f->w8 + (x >> 3); l2 = f->cu_log2[k]; md = f->cu_mode[k]; fl = f->cu_flags[k]; }
      if (an == ac) coded |= (fl & (RBT_CU_CBF_Y | RBT_CU_CBF_CB | RBT_CU_CBF_CR | 7 * RBT_CU_CBF_Y1)) != 0;
    }
    l->cu_l2[i] = (uint8_t)l2; l->cu_md[i] = (uint8_t)md; l->cu_fl[i] = (uint8_t)fl;
  }
  // a CTB without a coded block (P pictures: most of them are skipped CUs throughout) has no levels to fetch
  if (s->is_p && en_wave_sum(coded, (RBT_LDS_AS RbtEncLds*)0) == 0) {
    RBT_SYNC();
#ifdef RBT_PROFILE
    s->t_stage += __builtin_readcyclecounter() - ts_;
#endif
    return;
  }
  // the CTB's levels: 8-byte groups of four, several loads in flight (one HBM round trip per CTB instead of one per CU)
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, nn = (1 << L) >> sh, pw = s->w >> sh, ph = s->h >> sh, ox = cx >> sh, oy = cy >> sh;
    const RbtU2* cp = (const RbtU2*)(f->coef[c] + (size_t)oy * pw + ox); RBT_LDS_AS RbtU2* cd = (RBT_LDS_AS RbtU2*)en_lv(l, c, L);
    const int q4 = nn >> 2, lq4 = L - sh - 2, rows = rbt_min(nn, ph - oy), cols4 = rbt_min(nn, pw - ox) >> 2;
#pragma unroll 4
    RBT_PAR_FOR(i, nn * q4) {
      const int x4 = i & (q4 - 1), y = i >> lq4;
      RbtU2 v; v.x = 0; v.y = 0;
      if (x4 < cols4 && y < rows) v = cp[((size_t)y * pw >> 2) + x4];
      cd[i].x = v.x; cd[i].y = v.y;
    }
  }
  RBT_SYNC();
#ifdef RBT_PROFILE
  s->t_stage += __builtin_readcyclecounter() - ts_;
#endif
}
RBT_DEV void en_write_cu(RbtEnt* s, int x0, int y0, int log2, int depth) {
  RbtCabacEnc* c = &s->c; const RbtFrame* f = s->f; RBT_LDS_AS RbtEntropyLds* l = s->l;
#ifdef RBT_PROFILE
  unsigned long long tc_ = __builtin_readcyclecounter(); s->n_cu++;
#endif
  x0 = RBT_UNI(x0); y0 = RBT_UNI(y0); log2 = RBT_UNI(log2);
  const int u = en_u(s, x0, y0), flags = RBT_UNI(l->cu_fl[u]), mode = RBT_UNI(l->cu_md[u]);
  const int cbf_cb = (flags & RBT_CU_CBF_CB) != 0, cbf_cr = (flags & RBT_CU_CBF_CR) != 0, cbf_y = (flags & RBT_CU_CBF_Y) != 0;
  const int is_p = s->is_p;
  if (s->tq_bypass_enabled) rbt_ce_bin0(c, CTX_CU_TQ_BYPASS, f->lossless ? 1 : 0);
  const int ul = u - 1, ua = u - 9;                                  // left / above 8x8 unit
  const int av_l = RBT_UNI(l->cu_l2[ul]) != 0xFF, av_a = RBT_UNI(l->cu_l2[ua]) != 0xFF;
  if (is_p) {
    const int cl = av_l && (RBT_UNI(l->cu_fl[ul]) & RBT_CU_SKIP), ca = av_a && (RBT_UNI(l->cu_fl[ua]) & RBT_CU_SKIP);
    rbt_ce_bin0(c, CTX_CU_SKIP + (cl ? 1 : 0) + (ca ? 1 : 0), (flags & RBT_CU_SKIP) ? 1 : 0);
    if (flags & RBT_CU_SKIP) return;                 // merge_idx absent: MaxNumMergeCand == 1
    rbt_ce_bin0(c, CTX_PRED_MODE, 0);
    rbt_ce_bin0(c, CTX_PART_MODE, 1);
    rbt_ce_bin0(c, CTX_MERGE_FLAG, 1);
  } else {
    if (log2 == s->log2_min_cb) rbt_ce_bin0(c, CTX_PART_MODE, 1);
    int ca = 1, cb = 1;
    if (av_l) ca = RBT_UNI(l->cu_md[ul]);
    if (av_a && y0 > s->cy) cb = RBT_UNI(l->cu_md[ua]);              // the above candidate only counts inside the same CTB
    int c0, c1, c2;
    if (ca == cb) { if (ca < 2) { c0 = 0; c1 = 1; c2 = 26; } else { c0 = ca; c1 = 2 + ((ca + 29) % 32); c2 = 2 + ((ca - 2 + 1) % 32); } }
    else { c0 = ca; c1 = cb; c2 = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26); }
    int idx = mode == c0 ? 0 : (mode == c1 ? 1 : (mode == c2 ? 2 : -1));
    // the oracle keeps the LAST matching candidate; candidates are distinct, so first == last
    rbt_ce_bin0(c, CTX_PREV_INTRA_LUMA, idx >= 0);
    if (idx >= 0) { rbt_ce_bypass(c, idx > 0); if (idx > 0) rbt_ce_bypass(c, idx > 1); }
    else {
      int t;
      if (c0 > c1) { t = c0; c0 = c1; c1 = t; }
      if (c0 > c2) { t = c0; c0 = c2; c2 = t; }
      if (c1 > c2) { t = c1; c1 = c2; c2 = t; }
      int rem = mode;
      if (rem > c2) rem--;
      if (rem > c1) rem--;
      if (rem > c0) rem--;
      rbt_ce_bypass_n(c, (uint32_t)rem, 5);
    }
    rbt_ce_bin0(c, CTX_INTRA_CHROMA, 0);             // intra_chroma_pred_mode = 4 (DM)
  }
  const int intra = !is_p;
  const int ctb = 1 << s->log2_ctb, rx0 = x0 - s->cx, ry0 = y0 - s->cy;
  // transform tree (7.3.8.8): inter CUs are one TU (max_transform_hierarchy_depth_inter = 0, no flag); intra CUs of a stream with
  // max_transform_hierarchy_depth_intra = 1 are one TU or four (split_transform_flag at depth 0)
  if (intra && s->th_intra > 0) {
    const int split = (flags & RBT_CU_TU_SPLIT) != 0;
    rbt_ce_bin0(c, CTX_SPLIT_TRANSFORM + 5 - log2, split);
    if (split) {
      const int h = 1 << (log2 - 1);
      const int cu_ts = (log2 == 3 && s->f->cfg.transform_skip) ? RBT_UNI(s->f->cu_ts[(y0 >> 3) * s->f->w8 + (x0 >> 3)]) : 0;
      int fl[4], pcb = cbf_cb, pcr = cbf_cr;
      if (log2 > 3) {   // the 8x8 units of a quarter carry that quarter's cbf bits
        pcb = pcr = 0;
        for (int b = 0; b < 4; b++) { fl[b] = RBT_UNI(l->cu_fl[en_u(s, x0 + (b & 1) * h, y0 + (b >> 1) * h)]); pcb |= (fl[b] & RBT_CU_CBF_CB) != 0; pcr |= (fl[b] & RBT_CU_CBF_CR) != 0; }
      } else for (int b = 0; b < 4; b++) fl[b] = (flags & (b ? RBT_CU_CBF_Y1 << (b - 1) : RBT_CU_CBF_Y)) ? RBT_CU_CBF_Y : 0;
      rbt_ce_bin0(c, CTX_CBF_CHROMA + 0, pcb);
      rbt_ce_bin0(c, CTX_CBF_CHROMA + 0, pcr);
#ifdef RBT_PROFILE
      unsigned long long tr_ = __builtin_readcyclecounter(); s->t_cu += tr_ - tc_;
#endif
      for (int b = 0; b < 4; b++) {
        const int qx = rx0 + (b & 1) * h, qy = ry0 + (b >> 1) * h;
        int qcb = 0, qcr = 0;
        if (log2 > 3) {
          if (pcb) { qcb = (fl[b] & RBT_CU_CBF_CB) != 0; rbt_ce_bin0(c, CTX_CBF_CHROMA + 1, qcb); }
          if (pcr) { qcr = (fl[b] & RBT_CU_CBF_CR) != 0; rbt_ce_bin0(c, CTX_CBF_CHROMA + 1, qcr); }
        }
        const int qy_cbf = (fl[b] & RBT_CU_CBF_Y) != 0;
        rbt_ce_bin0(c, CTX_CBF_LUMA + 0, qy_cbf);
        if (qy_cbf) en_write_residual(s, 0, en_lv(l, 0, s->log2_ctb) + qy * ctb + qx, ctb, log2 - 1, en_scan_idx(1, log2 - 1, 0, mode), log2 == 3 ? (cu_ts >> b) & 1 : 0);
        if (log2 > 3) {
          if (qcb) en_write_residual(s, 1, en_lv(l, 1, s->log2_ctb) + (qy >> 1) * (ctb >> 1) + (qx >> 1), ctb >> 1, log2 - 2, en_scan_idx(1, log2 - 2, 1, mode));
          if (qcr) en_write_residual(s, 2, en_lv(l, 2, s->log2_ctb) + (qy >> 1) * (ctb >> 1) + (qx >> 1), ctb >> 1, log2 - 2, en_scan_idx(1, log2 - 2, 2, mode));
        } else if (b == 3) {   // 4x4 luma blocks: the chroma blocks of the 8x8 unit follow the fourth
          if (pcb) en_write_residual(s, 1, en_lv(l, 1, s->log2_ctb) + (ry0 >> 1) * (ctb >> 1) + (rx0 >> 1), ctb >> 1, 2, en_scan_idx(1, 2, 1, mode));
          if (pcr) en_write_residual(s, 2, en_lv(l, 2, s->log2_ctb) + (ry0 >> 1) * (ctb >> 1) + (rx0 >> 1), ctb >> 1, 2, en_scan_idx(1, 2, 2, mode));
        }
      }
#ifdef RBT_PROFILE
      s->t_res += __builtin_readcyclecounter() - tr_;
#endif
      return;
    }
  }
  rbt_ce_bin0(c, CTX_CBF_CHROMA + 0, cbf_cb);
  rbt_ce_bin0(c, CTX_CBF_CHROMA + 0, cbf_cr);
  if (!is_p || cbf_cb || cbf_cr) rbt_ce_bin0(c, CTX_CBF_LUMA + 1, cbf_y);
#ifdef RBT_PROFILE
  unsigned long long tr_ = __builtin_readcyclecounter(); s->t_cu += tr_ - tc_;
#endif
  if (cbf_y) en_write_residual(s, 0, en_lv(l, 0, s->log2_ctb) + ry0 * ctb + rx0, ctb, log2, en_scan_idx(intra, log2, 0, mode));
  if (cbf_cb) en_write_residual(s, 1, en_lv(l, 1, s->log2_ctb) + (ry0 >> 1) * (ctb >> 1) + (rx0 >> 1), ctb >> 1, log2 - 1, en_scan_idx(intra, log2 - 1, 1, mode));
  if (cbf_cr) en_write_residual(s, 2, en_lv(l, 2, s->log2_ctb) + (ry0 >> 1) * (ctb >> 1) + (rx0 >> 1), ctb >> 1, log2 - 1, en_scan_idx(intra, log2 - 1, 2, mode));
#ifdef RBT_PROFILE
  s->t_res += __builtin_readcyclecounter() - tr_;
#endif
  (void)depth;
}
RBT_DEV void en_write_quadtree(RbtEnt* s, int x0, int y0, int log2) {
  // stack-free depth-first walk (per-level child counters packed in one register, see the parser's pz_coding_quadtree)
  RBT_LDS_AS RbtEntropyLds* l = s->l;
  int lvl = 0, x = x0, y = y0, lg = log2;
  uint32_t states = 15u;
  for (;;) {
    lvl = RBT_UNI(lvl); x = RBT_UNI(x); y = RBT_UNI(y); lg = RBT_UNI(lg); states = (uint32_t)RBT_UNI(states);
    int st = (int)((states >> (4 * lvl)) & 15u);
    const int N = 1 << lg, h = N >> 1;
    if (st == 15) {
      const int can_flag = x + N <= s->w && y + N <= s->h && lg > s->log2_min_cb;
      const int u = en_u(s, x, y);
      int split = RBT_UNI(l->cu_l2[u]) < lg;
      if (!can_flag) split = lg > s->log2_min_cb;
      if (can_flag) {
        const int l2l = RBT_UNI(l->cu_l2[u - 1]), l2a = RBT_UNI(l->cu_l2[u - 9]);
        const int cl = l2l != 0xFF && (s->log2_ctb - l2l) > lvl, ca = l2a != 0xFF && (s->log2_ctb - l2a) > lvl;
        rbt_ce_bin0(&s->c, CTX_SPLIT_CU + (cl ? 1 : 0) + (ca ? 1 : 0), split);
      }
      if (!split) { en_write_cu(s, x, y, lg, lvl); st = 4; } else st = 0;
    }
    while (st < 4 && (x + (st & 1) * h >= s->w || y + (st >> 1) * h >= s->h)) st++;   // children outside the picture are skipped
    if (st < 4) {
      states = (states & ~(15u << (4 * lvl))) | ((uint32_t)(st + 1) << (4 * lvl));
      x += (st & 1) * h; y += (st >> 1) * h; lg--; lvl++;
      states = (states & ~(15u << (4 * lvl))) | (15u << (4 * lvl));
    } else {
      if (lvl == 0) break;
      lvl--;
      const int k = (int)((states >> (4 * lvl)) & 15u) - 1, hh = 1 << lg;
      x -= (k & 1) * hh; y -= (k >> 1) * hh; lg++;
    }
  }
}
// ------------------------------------------------------------------------------------------------ SAO parameter decision
// One wave per CTB, after deblocking: per component the statistics of source minus deblocked reconstruction - count and sum per band (32) and per edge
// class x category (4 x 4) - accumulated with LDS adds, then offsets = rounded means clipped to +-7 (edge categories keep their sign), the type with the
// largest distortion reduction minus lambda * rate; Cb and Cr share type and edge class (7.3.8.3). Mirrors hm_sao_decide in oracle/hevc_enc.c exactly.
// The same wave then applies the offsets to its CTB (needs rbt_filter.h).
// Cost matters here (1600 CTBs x 128 pictures per GOF): the edge statistics live in registers (16 (class, category) pairs, predicated adds, one wave
// reduction each at the end - no LDS traffic per sample), the band statistics go to eight LDS copies selected by the lane (neighbouring samples share a
// band, a single table would serialise the adds of the whole wave), the offsets are computed with the lanes over the table entries (one division per lane).
// Measured on MI355X, encoder stages per job with 16 GOFs in flight: no SAO 33 ms; one LDS table + decision on every lane 62 ms; this version 48 ms;
// staging the CTB and its halo in LDS with packed 8 / 16-bit register statistics 57 ms (11 KB of LDS per wave cost more residency than the loads it saved).
struct RbtSaoLds { int32_t bcnt[8][32], bsum[8][32]; int32_t ecnt[16], esum[16]; int32_t off[3][48], gain[48]; long long tgain[3][5]; int32_t bpos[3]; };
// round 3: the CTB and 4 luma (2 chroma) samples around it, deblocked in LDS (rbt_filter.h: the scheme of rbt_loopfilter_tile with the CTB as the tile) - the reconstruction is
// read once, the statistics, the decision and the offsets work on LDS, the output is written once; no deblocking pass over the picture. The region is sized by the CTB (TL2):
// with the 64x64 size for every stream a wave held 18.6 KB and two waves fitted a SIMD, which cost more than the fusion saved (k_enc_sao 3.8 -> 35 ms per launch under load)
template <int TL2> struct RbtSaoRegionT { uint16_t ry[((1 << TL2) + 8) * ((1 << TL2) + 8)]; uint16_t rc[2][((1 << TL2) / 2 + 4) * ((1 << TL2) / 2 + 4)]; };
RBT_DEV int en_sao_round_div(int sum, int cnt) { return cnt ? (sum >= 0 ? sum + cnt / 2 : sum - cnt / 2) / cnt : 0; }
// REGION (RBT_FUSED_ENC_LF=1): the CTB is deblocked here, in LDS - ry / rc0 / rc1: the region of RbtSaoRegionT (rows of ctb + 8 luma, ctb / 2 + 4 chroma samples). Otherwise (the
// default) the picture was deblocked in place by k_deblock and the samples are read where they lie (neighbours of a sample come from L2): a third less instructions than
// staging the region, and no LDS beyond the statistics
// Wave totals of sixteen per-lane values at once, left in dst[0..15] (LDS). Sixteen separate reductions were 27 % of the SAO kernel (4 DPP adds, 4 lane reads and 3 scalar
// adds each); here the lanes of a pair keep half of the values each and hand the other half over (twice: 16 -> 8 -> 4 values per lane, summed over a quad), the quads of a
// row are added by two rotations, the four rows through the LDS crossbar: ~60 instructions for all sixteen. Lane l < 4 ends with the totals of values (l & 1) * 8 + (l & 2) * 2 + 0..3.
RBT_DEV void en_wave_sum16(const int (&v)[16], RBT_LDS_AS int32_t* dst) {
#ifdef RBT_HOSTEMU
  for (int q = 0; q < 16; q++) dst[q] = v[q];                        // the serial PAR_FOR already accumulated everything
#else
  const int lane = (int)threadIdx.x & 63, b0 = lane & 1, b1 = lane & 2;
  int a[8], b[4];
#pragma unroll
  for (int k = 0; k < 8; k++) { const int keep = b0 ? v[k + 8] : v[k], send = b0 ? v[k] : v[k + 8]; a[k] = keep + __builtin_amdgcn_update_dpp(0, send, 0xB1, 0xF, 0xF, false); }   // quad_perm:[1,0,3,2]
#pragma unroll
  for (int k = 0; k < 4; k++) { const int keep = b1 ? a[k + 4] : a[k], send = b1 ? a[k] : a[k + 4]; b[k] = keep + __builtin_amdgcn_update_dpp(0, send, 0x4E, 0xF, 0xF, false); }   // quad_perm:[2,3,0,1]
#pragma unroll
  for (int k = 0; k < 4; k++) {
    b[k] += __builtin_amdgcn_update_dpp(0, b[k], 0x124, 0xF, 0xF, false);      // row_ror:4
    b[k] += __builtin_amdgcn_update_dpp(0, b[k], 0x128, 0xF, 0xF, false);      // row_ror:8: every lane of a row holds the row's sum of its four values
    b[k] += __builtin_amdgcn_ds_swizzle(b[k], 0x401F);                         // lane ^ 16
    b[k] += __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, b[k]);
  }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 4; k++) dst[(lane & 1) * 8 + (lane & 2) * 2 + k] = b[k];
  }
#endif
}
template <bool REGION> RBT_DEV void en_sao_ctb(RbtFrame* f, const RbtSlice* slices, int ctb_addr, RBT_LDS_AS RbtSaoLds* L, RBT_LDS_AS uint16_t* ry, RBT_LDS_AS uint16_t* rc0, RBT_LDS_AS uint16_t* rc1) {
  const RbtStreamCfg gcopy = rc_cfg_uni(&f->cfg); const RbtStreamCfg* g = &gcopy;
  const int ctb = 1 << g->log2_ctb, cxi = ctb_addr % g->w_ctb, cyi = ctb_addr / g->w_ctb, bd = g->bit_depth, has_occ = RBT_UNI(f->occ4 != nullptr);
  const RbtSlice* sl = &slices[f->ctb_slice[ctb_addr]];
  const long long lam16 = k_lambda16[rbt_clip3(0, 75, sl->qp + 6 * (bd - 8))], lam = lam16 * lam16;
  // a CTB made of skipped CUs only is a copy of the (already filtered) reference: no statistics, no offsets (P pictures are mostly that)
  int all_skip = 0;
  if (f->ref_frame >= 0) {
    const int n8 = ctb >> 3; uint64_t m0 = 0, m1 = 0;
    RBT_VBALLOT(m0, p, rbt_min(64, n8 * n8), (cxi * n8 + p % n8 >= f->w8) || (cyi * n8 + p / n8 >= f->h8) || (f->cu_flags[(cyi * n8 + p / n8) * f->w8 + cxi * n8 + p % n8] & RBT_CU_SKIP));
    m1 = n8 * n8 >= 64 ? ~0ull : (1ull << (n8 * n8)) - 1;
    all_skip = m0 == m1;
  }
  // ---- the CTB's reconstruction + halo into LDS, deblocked there: vertical edges, then horizontal ones (8.7.2; see rbt_loopfilter_tile for why a halo of 4 is exact)
  const int lx0 = cxi * ctb, ly0 = cyi * ctb, ox = lx0 - 4, oy = ly0 - 4, cox = lx0 / 2 - 2, coy = ly0 / 2 - 2, RS = ctb + 8, RSC = ctb / 2 + 4;
  if constexpr (REGION) {
    RBT_PAR_FOR(i, RS * RS) { const int x = ox + i % RS, y = oy + i / RS; ry[i] = (x >= 0 && y >= 0 && x < g->w && y < g->h) ? f->pix[0][(size_t)y * g->w + x] : 0; }
    RBT_PAR_FOR(i, 2 * RSC * RSC) {
      const int c = i / (RSC * RSC), j = i % (RSC * RSC), x = cox + j % RSC, y = coy + j / RSC;
      (c ? rc1 : rc0)[j] = (x >= 0 && y >= 0 && x < g->cw && y < g->ch) ? f->pix[1 + c][(size_t)y * g->cw + x] : 0;
    }
    RBT_SYNC_LDS();
    for (int dir = 0; dir < 2 && REGION; dir++) {
      const int ne = ctb / 8 + 1, ns = RS / 4;
      RBT_PAR_FOR(i, ne * ns) {
        const int e = i / ns, sg = i % ns;
        const int x = dir == 0 ? lx0 + 8 * e : ox + 4 * sg, y = dir == 0 ? oy + 4 * sg : ly0 + 8 * e;
        if (x >= 0 && y >= 0 && x < g->w && y < g->h) {
          const int bs = fl_bs(f, slices, x, y, dir);
          if (bs) {
            fl_luma_core(f, slices, x, y, dir, bs, &ry[(y - oy) * RS + (x - ox)], dir == 0 ? 1 : RS, dir == 0 ? RS : 1);
            if (bs == 2 && !(dir == 0 ? (x & 15) : (y & 15)))
              for (int c = 0; c < 2; c++) fl_chroma_core(f, slices, 1 + c, x, y, dir, &(c ? rc1 : rc0)[((y >> 1) - coy) * RSC + ((x >> 1) - cox)], dir == 0 ? 1 : RSC, dir == 0 ? RSC : 1);
          }
        }
      }
      RBT_SYNC_LDS();
    }
  }
  // deblocked sample of plane c at picture position (x,y) (inside the CTB or one sample around it)
#define EN_SAO_RP(c, x, y) (REGION ? ((c) == 0 ? (int)ry[((y) - oy) * RS + ((x) - ox)] : (int)((c) == 1 ? rc0 : rc1)[((y) - coy) * RSC + ((x) - cox)]) : (int)f->pix[c][(size_t)(y) * pw + (x)])
  for (int c = 0; c < 3 && !all_skip; c++) {
    const int sh = c ? 1 : 0, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, n = ctb >> sh, lgn = g->log2_ctb - sh;
    const int x0 = (cxi * ctb) >> sh, y0 = (cyi * ctb) >> sh;
    const uint16_t* sp = f->src[c];
    RBT_PAR_FOR(i, 8 * 32) { L->bcnt[i >> 5][i & 31] = 0; L->bsum[i >> 5][i & 31] = 0; }
    RBT_SYNC_LDS();
    int ecnt[16], esum[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { ecnt[q] = 0; esum[q] = 0; }
    // (round 4) a CTB whose region of the plane does not touch the picture border - nearly all of them - has every neighbour of every sample inside the picture: no clamps,
    // no inside tests, neighbours at fixed distances from the sample, and a lane's (count, sum) of a category in ONE register (sum * 128 + count: a lane sees at most 64
    // samples of a CTB, |difference| < 2^12; the serial host build of the tests, where one "lane" sees them all, keeps 64 bits) - the statistics are the same numbers as
    // the general loop's below
    const int interior = !REGION && x0 > 0 && y0 > 0 && x0 + n < pw && y0 + n < ph;
    if (interior) {
#ifdef RBT_HOSTEMU
      typedef long long EnSaoAcc; const int ash = 13;
#else
      typedef int EnSaoAcc; const int ash = 7;
#endif
      EnSaoAcc acc[16];
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] = 0;
      const uint16_t* rp = f->pix[c];
      RBT_PAR_FOR(i, n * n) {
        const int x = x0 + (i & (n - 1)), y = y0 + (i >> lgn);
        if (!has_occ || en_occ_unit(f, (x << sh) >> 2, (y << sh) >> 2)) {
          const size_t o = (size_t)y * pw + x;
          const int v = rp[o], d = (int)sp[o] - v, b = rbt_min(31, v >> (bd - 5)), copy = i & 7; const EnSaoAcc dp = (EnSaoAcc)d * (1 << ash) + 1;
          RBT_LDS_ADD(&L->bcnt[copy][b], 1); RBT_LDS_ADD(&L->bsum[copy][b], d);
#pragma unroll
          for (int cls = 0; cls < 4; cls++) {
            const int step = cls == 0 ? -1 : (cls == 1 ? -pw : (cls == 2 ? -pw - 1 : -pw + 1));
            const int va = rp[o + step], vb = rp[o - step];
            const int k = 2 + (v > va) - (v < va) + (v > vb) - (v < vb), cat = k < 2 ? k : k - 1;       // k == 2: no category
#pragma unroll
            for (int q = 0; q < 4; q++) acc[cls * 4 + q] += (k != 2 && cat == q) ? dp : (EnSaoAcc)0;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q++) { ecnt[q] = (int)(acc[q] & ((1 << ash) - 1)); esum[q] = (int)(acc[q] >> ash); }
    } else
    RBT_PAR_FOR(i, n * n) {
      const int x = x0 + (i & (n - 1)), y = y0 + (i >> lgn);
      if (x < pw && y < ph && (!has_occ || en_occ_unit(f, (x << sh) >> 2, (y << sh) >> 2))) {      // occupancy-aware coding: samples no point is made of have no say in the offsets
        const int v = EN_SAO_RP(c, x, y), d = (int)sp[(size_t)y * pw + x] - v, b = rbt_min(31, v >> (bd - 5)), copy = i & 7;
        RBT_LDS_ADD(&L->bcnt[copy][b], 1); RBT_LDS_ADD(&L->bsum[copy][b], d);
#pragma unroll
        for (int cls = 0; cls < 4; cls++) {
          const int dxa = cls == 1 ? 0 : (cls == 3 ? 1 : -1), dya = cls == 0 ? 0 : -1, xa = x + dxa, ya = y + dya, xb = x - dxa, yb = y - dya;
          const int inside = !(xa < 0 || ya < 0 || xb < 0 || yb < 0 || xa >= pw || xb >= pw || ya >= ph || yb >= ph);
          const int va = EN_SAO_RP(c, rbt_clip3(0, pw - 1, xa), rbt_clip3(0, ph - 1, ya)), vb = EN_SAO_RP(c, rbt_clip3(0, pw - 1, xb), rbt_clip3(0, ph - 1, yb));
          const int k = 2 + (v > va) - (v < va) + (v > vb) - (v < vb), cat = k < 2 ? k : k - 1;       // k == 2: no category
#pragma unroll
          for (int q = 0; q < 4; q++) { const int hit = inside && k != 2 && cat == q; ecnt[cls * 4 + q] += hit; esum[cls * 4 + q] += hit ? d : 0; }
        }
      }
    }
    en_wave_sum16(ecnt, L->ecnt); en_wave_sum16(esum, L->esum);
    RBT_SYNC_LDS();
    // offsets and distortion reductions, lanes over the 32 bands and the 16 (class, category) pairs
    RBT_PAR_FOR(j, 48) {
      int cnt, sum, o;
      if (j < 32) { cnt = 0; sum = 0; for (int q = 0; q < 8; q++) { cnt += L->bcnt[q][j]; sum += L->bsum[q][j]; } o = rbt_clip3(-7, 7, en_sao_round_div(sum, cnt)); }
      else { cnt = L->ecnt[j - 32]; sum = L->esum[j - 32]; o = en_sao_round_div(sum, cnt); o = ((j - 32) & 3) < 2 ? rbt_clip3(0, 7, o) : rbt_clip3(-7, 0, o); }
      L->off[c][j] = o; L->gain[j] = 2 * o * sum - cnt * o * o;
    }
    RBT_SYNC_LDS();
    { long long best = -1; int bp = 0;
      for (int b = 0; b <= 28; b++) { const long long gsum = (long long)L->gain[b] + L->gain[b + 1] + L->gain[b + 2] + L->gain[b + 3]; if (gsum > best) { best = gsum; bp = b; } }
      if (RBT_LANE0) { L->bpos[c] = bp; L->tgain[c][0] = best * 256 - lam * 18; } }
    for (int cls = 0; cls < 4; cls++) {
      long long gs = 0;
      for (int k = 0; k < 4; k++) gs += L->gain[32 + cls * 4 + k];
      if (RBT_LANE0) L->tgain[c][1 + cls] = gs * 256 - lam * 12;
    }
    RBT_SYNC_LDS();
  }
  RbtSao out; for (int i = 0; i < (int)sizeof(out); i++) ((uint8_t*)&out)[i] = 0;
  int bt = -1; long long bgn = 0;
  if (!all_skip) for (int t = 0; t < 5; t++) if (L->tgain[0][t] > bgn) { bgn = L->tgain[0][t]; bt = t; }
  if (bt >= 0) { out.type[0] = bt == 0 ? 1 : 2; out.band_pos[0] = (uint8_t)(bt == 0 ? L->bpos[0] : 0); out.eo_class[0] = (uint8_t)(bt ? bt - 1 : 0);
    for (int k = 0; k < 4; k++) out.offset[0][k] = (int8_t)(bt == 0 ? L->off[0][L->bpos[0] + k] : L->off[0][32 + (bt - 1) * 4 + k]); }
  bt = -1; bgn = 0;
  if (!all_skip) for (int t = 0; t < 5; t++) if (L->tgain[1][t] + L->tgain[2][t] > bgn) { bgn = L->tgain[1][t] + L->tgain[2][t]; bt = t; }
  if (bt >= 0) for (int c = 1; c < 3; c++) { out.type[c] = bt == 0 ? 1 : 2; out.band_pos[c] = (uint8_t)(bt == 0 ? L->bpos[c] : 0); out.eo_class[c] = (uint8_t)(bt ? bt - 1 : 0);
    for (int k = 0; k < 4; k++) out.offset[c][k] = (int8_t)(bt == 0 ? L->off[c][L->bpos[c] + k] : L->off[c][32 + (bt - 1) * 4 + k]); }
  // a type whose offsets are all zero costs bits for nothing
  { const int any0 = out.offset[0][0] | out.offset[0][1] | out.offset[0][2] | out.offset[0][3];
    const int any1 = out.offset[1][0] | out.offset[1][1] | out.offset[1][2] | out.offset[1][3], any2 = out.offset[2][0] | out.offset[2][1] | out.offset[2][2] | out.offset[2][3];
    if (!any0) out.type[0] = 0;
    if (!any1 && !any2) out.type[1] = out.type[2] = 0; }
  for (int c = 0; c < 3; c++) if (!out.type[c]) { out.band_pos[c] = 0; out.eo_class[c] = 0; for (int k = 0; k < 4; k++) out.offset[c][k] = 0; }
  if (out.type[1] != 2) out.eo_class[1] = out.eo_class[2] = 0;
  if (RBT_LANE0) f->sao[ctb_addr] = out;
  // ... and applied on the spot (8.7.3): the CTB's samples go from the deblocked picture to the output picture with the parameters still in registers
  // (round 4: the picture-resident form through rbt_sao_ctb_p - inner samples of the CTB without the per-sample slice and border look-ups; the LDS-region form as before)
  if constexpr (!REGION) rbt_sao_ctb_p(f, slices, ctb_addr, &out);
  else for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, n = ctb >> sh, lgn = g->log2_ctb - sh, x0 = (cxi * ctb) >> sh, y0 = (cyi * ctb) >> sh;
    RBT_PAR_FOR(i, n * n) {
      const int x = x0 + (i & (n - 1)), y = y0 + (i >> lgn);
      if (x < pw && y < ph) f->out[c][(size_t)y * pw + x] = (uint16_t)rbt_sao_value(f, slices, c, x, y, &out, [&](int xx, int yy) { return EN_SAO_RP(c, xx, yy); });
    }
  }
#undef EN_SAO_RP
  RBT_SYNC_LDS();
}
// sao() of one CTB (7.3.8.3) from the decided parameters: merged with the left / upper CTB of the same slice when they carry the same parameters
// A CTB's SAO record as three wave-uniform 64-bit words, fields picked out with shifts: a private copy of the struct indexed by run-time (component, offset) lives in scratch
// memory on this hardware - eight round trips to it per CTB on the entropy coder's serial chain (round 4).
struct EnSaoWords { uint64_t q[3]; };
RBT_DEV EnSaoWords en_sao_words(const RbtSao* p) {
  static_assert(sizeof(RbtSao) == 24, "three 64-bit words");
  const uint32_t* w = (const uint32_t*)p; EnSaoWords r;
  r.q[0] = (uint64_t)(uint32_t)RBT_UNI(w[0]) | ((uint64_t)(uint32_t)RBT_UNI(w[1]) << 32);       // written out: compiled for size the loop stays a loop, and an array indexed by its counter goes to scratch
  r.q[1] = (uint64_t)(uint32_t)RBT_UNI(w[2]) | ((uint64_t)(uint32_t)RBT_UNI(w[3]) << 32);
  r.q[2] = (uint64_t)(uint32_t)RBT_UNI(w[4]) | ((uint64_t)(uint32_t)RBT_UNI(w[5]) << 32);
  return r;
}
RBT_DEV int en_sao_byte(const EnSaoWords& r, int k) { const uint64_t q = k < 8 ? r.q[0] : (k < 16 ? r.q[1] : r.q[2]); return (int)((q >> (8 * (k & 7))) & 255u); }   // byte k of the record: type 0..2, band_pos 3..5, eo_class 6..8, offset 9..20
RBT_DEV int en_sao_same(const EnSaoWords& a, const EnSaoWords& b) { return a.q[0] == b.q[0] && a.q[1] == b.q[1] && a.q[2] == b.q[2]; }
RBT_DEV void en_write_sao(RbtCabacEnc* c, const RbtFrame* f, int addr, int rx, int ry, int wc, int bd) {
  const EnSaoWords p = en_sao_words(&f->sao[addr]);
  const int my = f->ctb_slice[addr], can_left = rx > 0 && f->ctb_slice[addr - 1] == my, can_up = ry > 0 && f->ctb_slice[addr - wc] == my;
  int merge_left = 0, merge_up = 0;
  if (can_left) merge_left = en_sao_same(p, en_sao_words(&f->sao[addr - 1]));
  if (can_up && !merge_left) merge_up = en_sao_same(p, en_sao_words(&f->sao[addr - wc]));
  merge_left = RBT_UNI(merge_left); merge_up = RBT_UNI(merge_up);
  if (can_left) rbt_ce_bin0(c, CTX_SAO_MERGE, merge_left);
  if (can_up && !merge_left) rbt_ce_bin0(c, CTX_SAO_MERGE, merge_up);
  if (merge_left || merge_up) return;
  const int cmax = (1 << (rbt_min(bd, 10) - 5)) - 1;
  for (int ci = 0; ci < 3; ci++) {
    const int t = en_sao_byte(p, ci);
    if (ci < 2) { rbt_ce_bin0(c, CTX_SAO_TYPE, t != 0); if (t) rbt_ce_bypass(c, t == 2); }
    if (!t) continue;
    for (int i = 0; i < 4; i++) { const int a = rbt_abs((int)(int8_t)en_sao_byte(p, 9 + 4 * ci + i)); if (a) rbt_ce_bypass_n(c, (1u << a) - 1u, a); if (a < cmax) rbt_ce_bypass(c, 0); }
    if (t == 1) {
      for (int i = 0; i < 4; i++) { const int o = (int)(int8_t)en_sao_byte(p, 9 + 4 * ci + i); if (o) rbt_ce_bypass(c, o < 0); }
      rbt_ce_bypass_n(c, (uint32_t)en_sao_byte(p, 3 + ci), 5);
    } else if (ci < 2) rbt_ce_bypass_n(c, (uint32_t)en_sao_byte(p, 6 + ci), 2);
  }
}
RBT_DEV void en_entropy_slice(RbtFrame* frames, RbtSlice* slices, int slice_idx, uint8_t* out, RBT_LDS_AS RbtEntropyLds* l) {
  RbtEnt s; s.sl = &slices[slice_idx]; s.f = &frames[RBT_UNI(s.sl->frame)]; s.slice_idx = slice_idx; s.l = l;
  const RbtSlice* sl = s.sl; const RbtStreamCfg* g = &s.f->cfg;
  s.w = RBT_UNI(g->w); s.h = RBT_UNI(g->h); s.log2_ctb = RBT_UNI(g->log2_ctb); s.log2_min_cb = RBT_UNI(g->log2_min_cb); s.tq_bypass_enabled = RBT_UNI(g->tq_bypass_enabled);
  s.th_intra = RBT_UNI(g->th_depth_intra);
  s.is_p = RBT_UNI(sl->slice_type) == RBT_SLICE_P; s.cx = s.cy = 0;
  RBT_PAR_FOR(i, 3 * 4 * 64) l->scan[i / 256][(i / 64) & 3][i & 63] = k_scan_packed(i / 256, (i / 64) & 3, i & 63);
  rbt_ctx_init(&s.c.cs, s.is_p ? 1 : 0, RBT_UNI(sl->qp));
  const int wave_mode = RBT_UNI((int)sl->wpp), wc0 = (s.w + (1 << s.log2_ctb) - 1) >> s.log2_ctb, row0 = RBT_UNI(sl->ctb_addr) / wc0, hc0 = RBT_UNI(g->h_ctb);
  if (wave_mode && row0 > 0 && wc0 > 1) {
    // wavefront mode (every segment is one CTB row of a one-slice picture): the context variables of the row above after its second CTB (9.3.1)
    rbt_flag_wait(&s.f->row_done[hc0 + row0 - 1], 2u, &s.f->error);
    rbt_ctx_load_g(&s.c.cs, s.f->row_ctx + (size_t)(row0 - 1) * 256);
  }
  s.c.out = rbt_uni_ptr(out + (uint32_t)RBT_UNI(sl->out_off)); s.c.cap = (uint32_t)RBT_UNI(sl->out_cap); s.c.n = 0; s.c.overflow = 0;
  rbt_ce_start(&s.c);
#ifdef RBT_PROFILE
  unsigned long long tall_ = __builtin_readcyclecounter(); s.t_stage = s.t_res = s.t_cu = s.t_resA = s.t_resB = 0; s.n_cu = s.n_tb = s.n_bins = 0;
#endif
  const int n_ctbs = RBT_UNI(sl->n_ctbs), first = RBT_UNI(sl->ctb_addr), wc = (s.w + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
  for (int a = 0; a < n_ctbs; a++) {
    const int addr = first + a, rx = addr % wc, ry = addr / wc;
    if (RBT_UNI(sl->sao_luma | sl->sao_chroma)) en_write_sao(&s.c, s.f, addr, rx, ry, wc, RBT_UNI(g->bit_depth));
    en_stage_ctb(&s, rx << s.log2_ctb, ry << s.log2_ctb);
    en_write_quadtree(&s, rx << s.log2_ctb, ry << s.log2_ctb, s.log2_ctb);
    if (wave_mode && rx == 1) { rbt_ctx_store_g(&s.c.cs, s.f->row_ctx + (size_t)ry * 256); RBT_FLAG_PUBLISH(&s.f->row_done[hc0 + ry], 2u); }
    rbt_ce_terminate(&s.c, a == n_ctbs - 1);
  }
  rbt_ce_align_zero(&s.c);
#ifdef RBT_PROFILE
  if (RBT_LANE0 && slice_idx == 10) printf("entropy slice %d: total %llu cyc, stage %llu, cu header %llu (%u CUs), residual %llu (%u TBs; setup+last %llu), bytes %u\n", slice_idx, __builtin_readcyclecounter() - tall_, s.t_stage, s.t_cu, s.n_cu, s.t_res, s.n_tb, s.t_resA, s.c.n);
#endif
  if (RBT_LANE0) slices[slice_idx].out_size = s.c.overflow ? 0xFFFFFFFFu : s.c.n;
}

// ------------------------------------------------------------------------------------------------ occupancy OR-pool
// resize_frame2 (PCCTranscoder.cpp:615-637): out[v][u] = any(in[v*f+v1][u*f+u1] > 0) ? 1 : 0
RBT_DEV void en_pool_sample(const uint16_t* in, int w, int factor, uint16_t* out, int ow, int u, int v) {
  int any = 0;
  for (int v1 = 0; v1 < factor; v1++) for (int u1 = 0; u1 < factor; u1++) any |= in[(size_t)(v * factor + v1) * w + u * factor + u1] > 0;
  out[(size_t)v * ow + u] = (uint16_t)(any ? 1 : 0);
}
