// In-loop filters: deblocking (H.265 8.7.2) and sample adaptive offset (8.7.3). HBM-bound streaming kernels:
// one lane per 4-sample edge segment (deblocking, one launch per edge direction) / per sample (SAO).
// Replaces the loop-filter stage of libavcodec's hevc decoder (PCCTranscoder.cpp:428-448); shared with the encoder.
#pragma once
#include "rbt_tables.h"
#include "rbt_types.h"

// slice index of the CTB that holds luma sample (x,y). A CTB no decoded slice covers (0xFFFF: truncated or damaged stream, reported as an error by the host)
// answers with the picture's first slice, so that every table read stays in bounds whatever the stream did.
RBT_DEV int fl_slice_of_ctb(const RbtFrame* f, int ctb) { const int v = f->ctb_slice[ctb]; return v == 0xFFFF ? f->first_slice : v; }
RBT_DEV int fl_slice_at(const RbtFrame* f, int x, int y) { return fl_slice_of_ctb(f, (y >> f->cfg.log2_ctb) * f->cfg.w_ctb + (x >> f->cfg.log2_ctb)); }

// bS of the edge on the left (dir 0) / top (dir 1) boundary of the 4x4 unit at luma (x,y); 0 = not filtered
RBT_DEV int fl_bs(const RbtFrame* f, const RbtSlice* slices, int x, int y, int dir) {
  const RbtStreamCfg* g = &f->cfg;
  int iq = (y >> 2) * g->w4 + (x >> 2);
  int e = dir == 0 ? (f->edges[iq] & 3) : ((f->edges[iq] >> 2) & 3);
  if (!e) return 0;
  int xp = dir == 0 ? x - 1 : x, yp = dir == 0 ? y : y - 1;
  if (xp < 0 || yp < 0) return 0;
  int ip = (yp >> 2) * g->w4 + (xp >> 2);
  int sq = fl_slice_at(f, x, y), sp = fl_slice_at(f, xp, yp);
  if (slices[sq].deblocking_disabled) return 0;
  if (sp != sq && !slices[sq].lf_across) return 0;
  int mp = f->pm[ip], mq = f->pm[iq];
  if ((mp & RBT_PM_MODE_MASK) == RBT_MODE_INTRA || (mq & RBT_PM_MODE_MASK) == RBT_MODE_INTRA) return 2;
  if ((e & 1) && ((mp | mq) & RBT_PM_NZ)) return 1;
  if (f->refpoc[ip] != f->refpoc[iq]) return 1;
  if (rbt_abs(f->mv[2 * ip] - f->mv[2 * iq]) >= 4 || rbt_abs(f->mv[2 * ip + 1] - f->mv[2 * iq + 1]) >= 4) return 1;
  return 0;
}

// one 4-sample luma edge segment: q = the first q0 sample, sa = step across the edge, sl = step along it (a plane in HBM or a tile in LDS)
template <class PTR> RBT_DEV void fl_luma_core(const RbtFrame* f, const RbtSlice* slices, int x, int y, int dir, int bs, PTR q, int sa, int sl) {
  const RbtStreamCfg* g = &f->cfg;
  int bd = g->bit_depth, maxv = (1 << bd) - 1;
  int iq = (y >> 2) * g->w4 + (x >> 2), ip = dir == 0 ? iq - 1 : iq - g->w4;
  const RbtSlice* s = &slices[fl_slice_at(f, x, y)];
  int qpl = (f->qp[iq] + f->qp[ip] + 1) >> 1;
  int beta = k_beta_table[rbt_clip3(0, 51, qpl + (s->beta_offset_div2 << 1))] * (1 << (bd - 8));
  int tc = k_tc_table[rbt_clip3(0, 53, qpl + 2 * (bs - 1) + (s->tc_offset_div2 << 1))] * (1 << (bd - 8));
#define FP(i, k) ((int)q[-((i) + 1) * sa + (k) * sl])
#define FQ(i, k) ((int)q[(i) * sa + (k) * sl])
  int dp0 = rbt_abs(FP(2, 0) - 2 * FP(1, 0) + FP(0, 0)), dp3 = rbt_abs(FP(2, 3) - 2 * FP(1, 3) + FP(0, 3));
  int dq0 = rbt_abs(FQ(2, 0) - 2 * FQ(1, 0) + FQ(0, 0)), dq3 = rbt_abs(FQ(2, 3) - 2 * FQ(1, 3) + FQ(0, 3));
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
  if (d >= beta) return;
  int ds0 = 2 * dpq0 < (beta >> 2) && rbt_abs(FP(3, 0) - FP(0, 0)) + rbt_abs(FQ(0, 0) - FQ(3, 0)) < (beta >> 3) && rbt_abs(FP(0, 0) - FQ(0, 0)) < ((5 * tc + 1) >> 1);
  int ds3 = 2 * dpq3 < (beta >> 2) && rbt_abs(FP(3, 3) - FP(0, 3)) + rbt_abs(FQ(0, 3) - FQ(3, 3)) < (beta >> 3) && rbt_abs(FP(0, 3) - FQ(0, 3)) < ((5 * tc + 1) >> 1);
  int strong = ds0 && ds3;
  int dEp = dp < ((beta + (beta >> 1)) >> 3), dEq = dq < ((beta + (beta >> 1)) >> 3);
  int no_p = f->pm[ip] & RBT_PM_TQ_BYPASS, no_q = f->pm[iq] & RBT_PM_TQ_BYPASS;
  for (int k = 0; k < 4; k++) {
    int p0 = FP(0, k), p1 = FP(1, k), p2 = FP(2, k), p3 = FP(3, k), q0 = FQ(0, k), q1 = FQ(1, k), q2 = FQ(2, k), q3 = FQ(3, k);
    PTR c = q + k * sl;
    if (strong) {
      if (!no_p) {
        c[-1 * sa] = (uint16_t)rbt_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        c[-2 * sa] = (uint16_t)rbt_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
        c[-3 * sa] = (uint16_t)rbt_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (!no_q) {
        c[0] = (uint16_t)rbt_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        c[sa] = (uint16_t)rbt_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
        c[2 * sa] = (uint16_t)rbt_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (rbt_abs(delta) < tc * 10) {
        delta = rbt_clip3(-tc, tc, delta);
        if (!no_p) {
          c[-sa] = (uint16_t)rbt_clip3(0, maxv, p0 + delta);
          if (dEp) c[-2 * sa] = (uint16_t)rbt_clip3(0, maxv, p1 + rbt_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
        }
        if (!no_q) {
          c[0] = (uint16_t)rbt_clip3(0, maxv, q0 - delta);
          if (dEq) c[sa] = (uint16_t)rbt_clip3(0, maxv, q1 + rbt_clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
        }
      }
    }
  }
#undef FP
#undef FQ
}
RBT_DEV void fl_luma_segment(RbtFrame* f, const RbtSlice* slices, int x, int y, int dir, int bs) {
  const int st = f->cfg.w;
  fl_luma_core(f, slices, x, y, dir, bs, f->pix[0] + (size_t)y * st + x, dir == 0 ? 1 : st, dir == 0 ? st : 1);
}
// one chroma edge segment (two samples along the edge) of plane c_idx at luma position (xl,yl): q = the first q0 sample
template <class PTR> RBT_DEV void fl_chroma_core(const RbtFrame* f, const RbtSlice* slices, int c_idx, int xl, int yl, int dir, PTR q, int sa, int sl) {
  const RbtStreamCfg* g = &f->cfg;
  int bd = g->bit_depth, maxv = (1 << bd) - 1;
  int iq = (yl >> 2) * g->w4 + (xl >> 2), ip = dir == 0 ? iq - 1 : iq - g->w4;
  const RbtSlice* s = &slices[fl_slice_at(f, xl, yl)];
  int off = c_idx == 1 ? g->cb_qp_offset : g->cr_qp_offset;
  int qpc = rbt_chroma_qp(((f->qp[iq] + f->qp[ip] + 1) >> 1) + off);
  int tc = k_tc_table[rbt_clip3(0, 53, qpc + 2 + (s->tc_offset_div2 << 1))] * (1 << (bd - 8));
  int no_p = f->pm[ip] & RBT_PM_TQ_BYPASS, no_q = f->pm[iq] & RBT_PM_TQ_BYPASS;
  for (int k = 0; k < 2; k++) {
    PTR c = q + k * sl;
    int p0 = c[-sa], p1 = c[-2 * sa], q0 = c[0], q1 = c[sa];
    int delta = rbt_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!no_p) c[-sa] = (uint16_t)rbt_clip3(0, maxv, p0 + delta);
    if (!no_q) c[0] = (uint16_t)rbt_clip3(0, maxv, q0 - delta);
  }
}
RBT_DEV void fl_chroma_segment(RbtFrame* f, const RbtSlice* slices, int c_idx, int xl, int yl, int dir) {
  const int st = f->cfg.cw;
  fl_chroma_core(f, slices, c_idx, xl, yl, dir, f->pix[c_idx] + (size_t)(yl >> 1) * st + (xl >> 1), dir == 0 ? 1 : st, dir == 0 ? st : 1);
}
// Edges lie on the 8x8 grid: only every second column (dir 0) / row (dir 1) of 4x4 units can carry one. rbt_deblock_edge_count / rbt_deblock_edge_unit enumerate exactly
// those units, so that every lane of the deblocking launch has an edge segment to look at (round 4: the launch over all units left every second lane idle and paid for it in
// whole-wave instructions).
RBT_DEV int rbt_deblock_edge_count(const RbtStreamCfg* g, int dir) { return dir == 0 ? ((g->w4 + 1) >> 1) * g->h4 : g->w4 * ((g->h4 + 1) >> 1); }
RBT_DEV int rbt_deblock_edge_unit(const RbtStreamCfg* g, int dir, int e) {
  if (dir == 0) { const int hw = (g->w4 + 1) >> 1; return (e / hw) * g->w4 + 2 * (e % hw); }
  return 2 * (e / g->w4) * g->w4 + e % g->w4;
}
// one 4x4 unit of one picture for edge direction `dir`
RBT_DEV void rbt_deblock_unit(RbtFrame* f, const RbtSlice* slices, int unit, int dir) {
  const RbtStreamCfg* g = &f->cfg;
  int x = (unit % g->w4) << 2, y = (unit / g->w4) << 2;
  if (dir == 0 ? (x & 7) : (y & 7)) return;
  int bs = fl_bs(f, slices, x, y, dir);
  if (!bs) return;
  fl_luma_segment(f, slices, x, y, dir, bs);
  if (bs == 2 && !(dir == 0 ? (x & 15) : (y & 15))) { fl_chroma_segment(f, slices, 1, x, y, dir); fl_chroma_segment(f, slices, 2, x, y, dir); }
}

// SAO of one sample of component c: reads f->pix (deblocked), writes f->out
// SAO of one sample with the parameters *s of its CTB (the encoder applies them right after deciding them: en_sao_ctb)
// (FETCH: the deblocked sample of plane c at (x,y) - from the plane in HBM, or from a tile in LDS)
template <class FETCH> RBT_DEV int rbt_sao_value(const RbtFrame* f, const RbtSlice* slices, int c, int x, int y, const RbtSao* s, FETCH fetch) {
  const RbtStreamCfg* g = &f->cfg;
  int sh = c ? 1 : 0, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, bd = g->bit_depth, maxv = (1 << bd) - 1;
  int xl = x << sh, yl = y << sh;
  int ctb = (yl >> g->log2_ctb) * g->w_ctb + (xl >> g->log2_ctb);
  const RbtSlice* sl = &slices[fl_slice_of_ctb(f, ctb)];
  int v = fetch(x, y), outv = v;
  int type = s->type[c];
  if ((c ? sl->sao_chroma : sl->sao_luma) && type && !(f->pm[(yl >> 2) * g->w4 + (xl >> 2)] & RBT_PM_TQ_BYPASS)) {
    if (type == 1) {
      int band = v >> (bd - 5), k = (band - s->band_pos[c]) & 31;
      if (k < 4) outv = rbt_clip3(0, maxv, v + s->offset[c][k]);
    } else {
      int cls = s->eo_class[c];
      int dxa = cls == 1 ? 0 : (cls == 3 ? 1 : -1), dya = cls == 0 ? 0 : -1;
      int xa = x + dxa, ya = y + dya, xb = x - dxa, yb = y - dya;
      if (xa >= 0 && ya >= 0 && xb >= 0 && yb >= 0 && xa < pw && xb < pw && ya < ph && yb < ph) {
        int sa_ = fl_slice_at(f, xa << sh, ya << sh), sb_ = fl_slice_at(f, xb << sh, yb << sh), sc_ = fl_slice_of_ctb(f, ctb);
        int ok = !((sa_ != sc_ && !slices[sa_ > sc_ ? sa_ : sc_].lf_across) || (sb_ != sc_ && !slices[sb_ > sc_ ? sb_ : sc_].lf_across));
        if (ok) {
          int va = fetch(xa, ya), vb = fetch(xb, yb);
          int e = 2 + (v > va) - (v < va) + (v > vb) - (v < vb);
          if (e == 0 || e == 1 || e == 2) e = (e == 2) ? 0 : e + 1;
          if (e) outv = rbt_clip3(0, maxv, v + s->offset[c][e - 1]);
        }
      }
    }
  }
  return outv;
}
RBT_DEV void rbt_sao_sample_p(RbtFrame* f, const RbtSlice* slices, int c, int x, int y, const RbtSao* s) {
  const int pw = c ? f->cfg.cw : f->cfg.w; const uint16_t* sp = f->pix[c];
  f->out[c][(size_t)y * pw + x] = (uint16_t)rbt_sao_value(f, slices, c, x, y, s, [&](int xx, int yy) { return (int)sp[(size_t)yy * pw + xx]; });
}
// SAO of one CTB by one workgroup (round 4; k_sao_ctb): what rbt_sao_sample does per sample, with everything that is the same for a CTB - its parameters, its slice's
// flags, whether the PPS allows lossless CUs at all - read once, and the neighbourhood tests done only where they can fail: a sample that does not lie on the border of
// the CTB's region of a plane has both edge-class neighbours inside the same CTB, hence inside the picture and inside the same slice, so the inner (n - 2)^2 samples take
// a short path (three loads, the class, one of four offsets held in registers) and the ring of 4n - 4 border samples goes through rbt_sao_value as before. The per-sample
// form cost 3.3 instructions per sample in all 64 lanes - 1.05 G of the path's 18.6 G per GOF (profiles/r04_pmc_sq.json) - almost all of it CTB / slice look-ups.
RBT_DEV void rbt_sao_ctb_p(RbtFrame* f, const RbtSlice* slices, int ctb, const RbtSao* s) {   // (s: the CTB's parameters - the encoder applies them while it still holds them)
  const RbtStreamCfg* g = &f->cfg;
  const int cx = (ctb % g->w_ctb) << g->log2_ctb, cy = (ctb / g->w_ctb) << g->log2_ctb;
  const RbtSlice* sl = &slices[fl_slice_of_ctb(f, ctb)];
  const int bd = g->bit_depth, maxv = (1 << bd) - 1, bypass_possible = g->tq_bypass_enabled;
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, x0 = cx >> sh, y0 = cy >> sh;
    const int nw = rbt_min((1 << g->log2_ctb) >> sh, pw - x0), nh = rbt_min((1 << g->log2_ctb) >> sh, ph - y0);
    const uint16_t* sp = f->pix[c]; uint16_t* dp = f->out[c];
    const int type = (c ? sl->sao_chroma : sl->sao_luma) ? s->type[c] : 0;
    if (!type) { RBT_BLK_FOR(i, nw * nh) { const size_t o = (size_t)(y0 + i / nw) * pw + x0 + i % nw; dp[o] = sp[o]; } continue; }
    const int o0 = s->offset[c][0], o1 = s->offset[c][1], o2 = s->offset[c][2], o3 = s->offset[c][3];
    if (type == 1) {                                                      // band offset: no neighbours
      const int bp = s->band_pos[c];
      RBT_BLK_FOR(i, nw * nh) {
        const int x = x0 + i % nw, y = y0 + i / nw; const size_t o = (size_t)y * pw + x;
        const int v = sp[o], k = ((v >> (bd - 5)) - bp) & 31;
        int r = v;
        if (k < 4 && !(bypass_possible && (f->pm[((y << sh) >> 2) * g->w4 + ((x << sh) >> 2)] & RBT_PM_TQ_BYPASS))) r = rbt_clip3(0, maxv, v + (k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3))));
        dp[o] = (uint16_t)r;
      }
      continue;
    }
    const int cls = s->eo_class[c], dxa = cls == 1 ? 0 : (cls == 3 ? 1 : -1), dya = cls == 0 ? 0 : -1, step = dya * pw + dxa;
    const int iw = nw - 2, ih = nh - 2;
    if (iw > 0 && ih > 0) {
      RBT_BLK_FOR(i, iw * ih) {                                           // inner samples: neighbours in the same CTB
        const int x = x0 + 1 + i % iw, y = y0 + 1 + i / iw; const size_t o = (size_t)y * pw + x;
        const int v = sp[o], va = sp[o + step], vb = sp[o - step];
        const int e = (v > va) - (v < va) + (v > vb) - (v < vb);         // -2 valley, -1, 0 none, 1, 2 peak: offsets 0 1 - 2 3 (edgeIdx 1 2 0 3 4 of 8.7.3.2)
        int r = v;
        if (e != 0 && !(bypass_possible && (f->pm[((y << sh) >> 2) * g->w4 + ((x << sh) >> 2)] & RBT_PM_TQ_BYPASS))) r = rbt_clip3(0, maxv, v + (e == -2 ? o0 : (e == -1 ? o1 : (e == 1 ? o2 : o3))));
        dp[o] = (uint16_t)r;
      }
    }
    const int ring = iw > 0 && ih > 0 ? 2 * nw + 2 * ih : nw * nh;         // the border of the region (all of it when it is one or two samples wide): the general routine
    RBT_BLK_FOR(i, ring) {
      int lx, ly;
      if (!(iw > 0 && ih > 0)) { lx = i % nw; ly = i / nw; }
      else if (i < nw) { lx = i; ly = 0; }
      else if (i < 2 * nw) { lx = i - nw; ly = nh - 1; }
      else if (i < 2 * nw + ih) { lx = 0; ly = 1 + i - 2 * nw; }
      else { lx = nw - 1; ly = 1 + i - 2 * nw - ih; }
      const int x = x0 + lx, y = y0 + ly;
      dp[(size_t)y * pw + x] = (uint16_t)rbt_sao_value(f, slices, c, x, y, s, [&](int xx, int yy) { return (int)sp[(size_t)yy * pw + xx]; });
    }
  }
}
RBT_DEV void rbt_sao_ctb(RbtFrame* f, const RbtSlice* slices, int ctb) { rbt_sao_ctb_p(f, slices, ctb, &f->sao[ctb]); }
RBT_DEV void rbt_sao_sample(RbtFrame* f, const RbtSlice* slices, int c, int x, int y) {
  const RbtStreamCfg* g = &f->cfg; const int sh = c ? 1 : 0;
  rbt_sao_sample_p(f, slices, c, x, y, &f->sao[(((y << sh) >> g->log2_ctb)) * g->w_ctb + ((x << sh) >> g->log2_ctb)]);
}

// ---- deblocking + SAO of one 64x64 luma tile in LDS (round 3: one read of the reconstruction and one write of the output per sample instead of four full-picture passes) ----
// The tile's outputs need the deblocked samples one sample beyond it (SAO's neighbours); a deblocked sample needs the horizontal-edge filter of its row group, which reads the
// vertically-filtered samples up to 4 rows away; those need the reconstruction up to 4 columns away. So a workgroup loads the tile with a halo of 4 luma (2 chroma) samples,
// filters every vertical edge x0 + 8k (k = 0..8) over the 72 rows, then every horizontal edge y0 + 8k over the 72 columns - each edge moves at most 3 samples either side and
// reads 4, so edges 8 apart never touch each other's samples and everything the tile's outputs depend on is exact - applies SAO from LDS and writes ITS 64x64 samples of `out`.
// The halo is computed again by the neighbouring tiles ((72 / 64)^2 = 1.27x the filter arithmetic) instead of waited for. Needs out != pix (pictures with SAO).
#define RBT_LF_TILE 64
#define RBT_LF_R (RBT_LF_TILE + 8)
#define RBT_LF_RC (RBT_LF_TILE / 2 + 4)
struct RbtLoopLds { uint16_t y[RBT_LF_R * RBT_LF_R]; uint16_t c[2][RBT_LF_RC * RBT_LF_RC]; };
RBT_DEV void rbt_loopfilter_tile(RbtFrame* f, const RbtSlice* slices, int tile, RBT_LDS_AS RbtLoopLds* L) {
  const RbtStreamCfg* g = &f->cfg;
  const int tw = (g->w + RBT_LF_TILE - 1) / RBT_LF_TILE, x0 = (tile % tw) * RBT_LF_TILE, y0 = (tile / tw) * RBT_LF_TILE, ox = x0 - 4, oy = y0 - 4, cox = x0 / 2 - 2, coy = y0 / 2 - 2;
  RBT_BLK_FOR(i, RBT_LF_R * RBT_LF_R) { const int x = ox + i % RBT_LF_R, y = oy + i / RBT_LF_R; L->y[i] = (x >= 0 && y >= 0 && x < g->w && y < g->h) ? f->pix[0][(size_t)y * g->w + x] : 0; }
  RBT_BLK_FOR(i, 2 * RBT_LF_RC * RBT_LF_RC) {
    const int c = i / (RBT_LF_RC * RBT_LF_RC), j = i % (RBT_LF_RC * RBT_LF_RC), x = cox + j % RBT_LF_RC, y = coy + j / RBT_LF_RC;
    L->c[c][j] = (x >= 0 && y >= 0 && x < g->cw && y < g->ch) ? f->pix[1 + c][(size_t)y * g->cw + x] : 0;
  }
  RBT_SYNC();
  for (int dir = 0; dir < 2; dir++) {
    RBT_BLK_FOR(i, 9 * 18) {
      const int e = i / 18, sg = i % 18;                                              // edge e of the tile, 4-sample segment sg along it
      const int x = dir == 0 ? x0 + 8 * e : ox + 4 * sg, y = dir == 0 ? oy + 4 * sg : y0 + 8 * e;
      if (x >= 0 && y >= 0 && x < g->w && y < g->h) {
        const int bs = fl_bs(f, slices, x, y, dir);
        if (bs) {
          fl_luma_core(f, slices, x, y, dir, bs, &L->y[(y - oy) * RBT_LF_R + (x - ox)], dir == 0 ? 1 : RBT_LF_R, dir == 0 ? RBT_LF_R : 1);
          if (bs == 2 && !(dir == 0 ? (x & 15) : (y & 15)))
            for (int c = 0; c < 2; c++) fl_chroma_core(f, slices, 1 + c, x, y, dir, &L->c[c][((y >> 1) - coy) * RBT_LF_RC + ((x >> 1) - cox)], dir == 0 ? 1 : RBT_LF_RC, dir == 0 ? RBT_LF_RC : 1);
        }
      }
    }
    RBT_SYNC();
  }
  RBT_BLK_FOR(i, RBT_LF_TILE * RBT_LF_TILE) {
    const int x = x0 + i % RBT_LF_TILE, y = y0 + i / RBT_LF_TILE;
    if (x < g->w && y < g->h) {
      const RbtSao* sp = &f->sao[(y >> g->log2_ctb) * g->w_ctb + (x >> g->log2_ctb)];
      f->out[0][(size_t)y * g->w + x] = (uint16_t)rbt_sao_value(f, slices, 0, x, y, sp, [&](int xx, int yy) { return (int)L->y[(yy - oy) * RBT_LF_R + (xx - ox)]; });
    }
  }
  RBT_BLK_FOR(i, 2 * (RBT_LF_TILE / 2) * (RBT_LF_TILE / 2)) {
    const int c = i / ((RBT_LF_TILE / 2) * (RBT_LF_TILE / 2)), j = i % ((RBT_LF_TILE / 2) * (RBT_LF_TILE / 2)), x = x0 / 2 + j % (RBT_LF_TILE / 2), y = y0 / 2 + j / (RBT_LF_TILE / 2);
    if (x < g->cw && y < g->ch) {
      const RbtSao* sp = &f->sao[((y << 1) >> g->log2_ctb) * g->w_ctb + ((x << 1) >> g->log2_ctb)];
      f->out[1 + c][(size_t)y * g->cw + x] = (uint16_t)rbt_sao_value(f, slices, 1 + c, x, y, sp, [&](int xx, int yy) { return (int)L->c[c][(yy - coy) * RBT_LF_RC + (xx - cox)]; });
    }
  }
}
