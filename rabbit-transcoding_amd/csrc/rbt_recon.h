// Picture reconstruction device routines (H.265 8.4.4.2 intra, 8.5.3.3 inter, 8.6 scaling + inverse transforms),
// executed by ONE 64-lane wave per CTB (wave-synchronous: a workgroup barrier of a single-wave workgroup is free),
// lanes spread over the samples of the current block, tiles staged in LDS.
// Replaces the reconstruction half of libavcodec's hevc decoder as driven by PCCTranscoder.cpp:428-448, and is shared
// with the encoder's reconstruction loop (rbt_encode.h).
#pragma once
#include "rbt_tables.h"
#include "rbt_types.h"

struct RbtReconLds {
  int32_t nb[132];        // neighbour samples: [0] = p[-1][2N-1] .. [2N-1] = p[-1][0], [2N] = corner, [2N+1+x] = p[x][-1]
  int32_t nbf[132];       // filtered neighbours
  int32_t ref[100];       // angular reference array, index offset 32
  uint8_t av[132];
  int16_t res[32 * 32];   // dequantised coefficients, then residual
  int32_t tmp[32 * 32];   // first transform stage
  uint16_t pred[32 * 32];
};

RBT_DEV int rc_morton(int x4, int y4) {
  int z = 0;
  for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1));
  return z;
}
// z-scan availability (6.4.1) of luma sample (xn,yn) for the block whose first 4x4 unit is at (xc,yc)
RBT_DEV int rc_avail(const RbtFrame* f, int xc, int yc, int xn, int yn) {
  const RbtStreamCfg* g = &f->cfg;
  if (xn < 0 || yn < 0 || xn >= g->w || yn >= g->h) return 0;
  int L = g->log2_ctb, an = (yn >> L) * g->w_ctb + (xn >> L), ac = (yc >> L) * g->w_ctb + (xc >> L);
  if (an > ac || f->ctb_slice[an] != f->ctb_slice[ac]) return 0;
  if (an == ac) {
    int m = (1 << L) - 1;
    if (rc_morton((xn & m) >> 2, (yn & m) >> 2) >= rc_morton((xc & m) >> 2, (yc & m) >> 2)) return 0;
  }
  if (g->cip && (f->pm[(yn >> 2) * g->w4 + (xn >> 2)] & RBT_PM_MODE_MASK) != RBT_MODE_INTRA) return 0;
  return 1;
}
RBT_DEV int rc_tcoef(int N, int is_dst, int k, int n) { return is_dst ? k_dst4[k][n] : k_dct32[k * (32 / N)][n]; }

// ---- intra prediction of one TB into lds->pred (8.4.4.2). `src` is the plane neighbours are read from. ----
RBT_DEV void rc_intra_pred(const RbtFrame* f, const uint16_t* src, int c_idx, int x0, int y0, int log2, int mode, RBT_LDS_AS RbtReconLds* l) {
  const RbtStreamCfg* g = &f->cfg;
  int N = 1 << log2, sh = c_idx ? 1 : 0, pw = c_idx ? g->cw : g->w, bd = g->bit_depth, maxv = (1 << bd) - 1;
  int xcL = x0 << sh, ycL = y0 << sh, tot = 4 * N + 1;
  RBT_PAR_FOR(i, tot) {
    int xn, yn;
    if (i < 2 * N) { xn = x0 - 1; yn = y0 + (2 * N - 1 - i); }
    else if (i == 2 * N) { xn = x0 - 1; yn = y0 - 1; }
    else { xn = x0 + (i - 2 * N - 1); yn = y0 - 1; }
    int a = rc_avail(f, xcL, ycL, xn << sh, yn << sh);
    l->av[i] = (uint8_t)a;
    l->nb[i] = a ? src[(size_t)yn * pw + xn] : 0;
  }
  RBT_SYNC_LDS();
  // substitution (8.4.4.2.2): uniform serial scan
  {
    int first = -1;
    for (int i = 0; i < tot; i++) if (l->av[i]) { first = i; break; }
    if (first < 0) { RBT_PAR_FOR(i, tot) l->nb[i] = 1 << (bd - 1); }
    else {
      RBT_PAR_FOR(i, tot) {
        if (!l->av[i]) {
          int j = i; while (j >= 0 && !l->av[j]) j--;
          // unavailable samples before the first available one take its value; later ones copy the nearest below
          l->nbf[i] = j >= 0 ? l->nb[j] : l->nb[first];
        } else l->nbf[i] = l->nb[i];
      }
      RBT_SYNC_LDS();
      RBT_PAR_FOR(i, tot) l->nb[i] = l->nbf[i];
    }
  }
  RBT_SYNC_LDS();
  int filt = 0;
  if (c_idx == 0 && mode != 1 && N != 4) {
    int md = rbt_min(rbt_abs(mode - 26), rbt_abs(mode - 10));
    int thr = N == 8 ? 7 : (N == 16 ? 1 : 0);
    filt = md > thr;
  }
  if (filt) {
    int corner = l->nb[2 * N], bl = l->nb[0], tr = l->nb[4 * N];
    int strong = g->strong_intra && N == 32 && rbt_abs(corner + tr - 2 * l->nb[2 * N + 32]) < (1 << (bd - 5)) &&
                 rbt_abs(corner + bl - 2 * l->nb[2 * N - 32]) < (1 << (bd - 5));
    RBT_PAR_FOR(i, tot) {
      int v;
      if (i == 0 || i == 4 * N) v = l->nb[i];
      else if (strong) {
        if (i == 2 * N) v = corner;
        else if (i < 2 * N) { int k = 2 * N - 1 - i; v = ((63 - k) * corner + (k + 1) * bl + 32) >> 6; }
        else { int k = i - 2 * N - 1; v = ((63 - k) * corner + (k + 1) * tr + 32) >> 6; }
      } else v = (l->nb[i - 1] + 2 * l->nb[i] + l->nb[i + 1] + 2) >> 2;
      l->nbf[i] = v;
    }
    RBT_SYNC_LDS();
    RBT_PAR_FOR(i, tot) l->nb[i] = l->nbf[i];
    RBT_SYNC_LDS();
  }
#define RC_LEFT(y) l->nb[2 * N - 1 - (y)]
#define RC_TOP(x) l->nb[2 * N + 1 + (x)]
  if (mode == 0) {
    RBT_PAR_FOR(i, N * N) {
      int x = i & (N - 1), y = i >> log2;
      l->pred[i] = (uint16_t)(((N - 1 - x) * RC_LEFT(y) + (x + 1) * RC_TOP(N) + (N - 1 - y) * RC_TOP(x) + (y + 1) * RC_LEFT(N) + N) >> (log2 + 1));
    }
  } else if (mode == 1) {
    int sum = N;
    for (int i = 0; i < N; i++) sum += RC_TOP(i) + RC_LEFT(i);
    int dc = sum >> (log2 + 1);
    int edge = c_idx == 0 && N < 32;
    RBT_PAR_FOR(i, N * N) {
      int x = i & (N - 1), y = i >> log2, v = dc;
      if (edge) {
        if (x == 0 && y == 0) v = (RC_LEFT(0) + 2 * dc + RC_TOP(0) + 2) >> 2;
        else if (y == 0) v = (RC_TOP(x) + 3 * dc + 2) >> 2;
        else if (x == 0) v = (RC_LEFT(y) + 3 * dc + 2) >> 2;
      }
      l->pred[i] = (uint16_t)v;
    }
  } else {
    int ang = k_intra_angle[mode], ver = mode >= 18;
    int last = (N * ang) >> 5;
    int inv = (mode >= 11 && mode <= 25) ? k_intra_inv_angle[mode - 11] : 0;
    // ref[x], x = -N .. 2N  (stored at index x + 32)
    RBT_PAR_FOR(i, 3 * N + 1) {
      int x = i - N, v = 0;
      if (x >= 0 && x <= N) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
      else if (x < 0) { if (ang < 0 && last < -1 && x >= last) { int k = -1 + ((x * inv + 128) >> 8); v = ver ? RC_LEFT(k) : RC_TOP(k); } }
      else if (ang >= 0) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
      l->ref[x + 32] = v;
    }
    RBT_SYNC_LDS();
    int edge = c_idx == 0 && N < 32 && (mode == 26 || mode == 10);
    RBT_PAR_FOR(i, N * N) {
      int x = i & (N - 1), y = i >> log2;
      int a = ver ? y : x, b = ver ? x : y;
      int idx = ((a + 1) * ang) >> 5, fr = ((a + 1) * ang) & 31;
      int v = fr ? ((32 - fr) * l->ref[32 + b + idx + 1] + fr * l->ref[32 + b + idx + 2] + 16) >> 5 : l->ref[32 + b + idx + 1];
      if (edge) {
        if (mode == 26 && x == 0) v = rbt_clip3(0, maxv, RC_TOP(0) + ((RC_LEFT(y) - RC_LEFT(-1)) >> 1));
        if (mode == 10 && y == 0) v = rbt_clip3(0, maxv, RC_LEFT(0) + ((RC_TOP(x) - RC_TOP(-1)) >> 1));
      }
      l->pred[i] = (uint16_t)v;
    }
  }
#undef RC_LEFT
#undef RC_TOP
  RBT_SYNC_LDS();
}

// ---- scaling (8.6.3, flat lists) of the TB's levels from the coefficient plane into lds->res ----
RBT_DEV void rc_dequant(const int16_t* plane, int pst, int x0, int y0, int log2, int qp, int bd, RBT_LDS_AS RbtReconLds* l) {
  int N = 1 << log2, bd_shift = bd + log2 - 5;
  int scale = (16 * k_dequant_scale[qp % 6]) << (qp / 6);
  long long add = 1ll << (bd_shift - 1);
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2;
    long long v = ((long long)plane[(size_t)(y0 + y) * pst + x0 + x] * scale + add) >> bd_shift;
    l->res[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
  RBT_SYNC_LDS();
}
// ---- inverse transform of lds->res in place (8.6.4.2) ----
RBT_DEV void rc_inv_transform(int log2, int is_dst, int ts, int bd, RBT_LDS_AS RbtReconLds* l) {
  int N = 1 << log2, sh = 20 - bd;
  if (ts) {
    RBT_PAR_FOR(i, N * N) l->res[i] = (int16_t)((((int)l->res[i] << 7) + (1 << (sh - 1))) >> sh);
    RBT_SYNC_LDS();
    return;
  }
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2, s = 0;
    for (int k = 0; k < N; k++) s += rc_tcoef(N, is_dst, k, y) * l->res[k * N + x];
    l->tmp[i] = rbt_clip3(-32768, 32767, (s + 64) >> 7);
  }
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2, s = 0;
    for (int k = 0; k < N; k++) s += rc_tcoef(N, is_dst, k, x) * l->tmp[y * N + k];
    l->res[i] = (int16_t)((s + (1 << (sh - 1))) >> sh);
  }
  RBT_SYNC_LDS();
}

// ---- one TB of the decoder: prediction (intra) + residual, written to f->pix ----
RBT_DEV void rc_decode_tb(RbtFrame* f, int c_idx, int x0, int y0, int log2, int intra, int mode, int cbf, int ts, int tq_bypass, int qp, RBT_LDS_AS RbtReconLds* l) {
  const RbtStreamCfg* g = &f->cfg;
  int N = 1 << log2, pw = c_idx ? g->cw : g->w, bd = g->bit_depth, maxv = (1 << bd) - 1;
  uint16_t* p = f->pix[c_idx];
  if (intra) rc_intra_pred(f, p, c_idx, x0, y0, log2, mode, l);
  if (cbf) {
    if (tq_bypass) {
      RBT_PAR_FOR(i, N * N) { int x = i & (N - 1), y = i >> log2; l->res[i] = f->coef[c_idx][(size_t)(y0 + y) * pw + x0 + x]; }
      RBT_SYNC_LDS();
    } else {
      rc_dequant(f->coef[c_idx], pw, x0, y0, log2, qp, bd, l);
      rc_inv_transform(log2, c_idx == 0 && log2 == 2 && intra, ts, bd, l);
    }
  }
  if (!intra && !cbf) return;
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2;
    size_t o = (size_t)(y0 + y) * pw + x0 + x;
    int base = intra ? l->pred[i] : p[o];
    p[o] = (uint16_t)(cbf ? rbt_clip3(0, maxv, base + l->res[i]) : base);
  }
  RBT_SYNC();   // the next TB reads these samples from HBM/L2 as neighbours
}

// ---- uni-directional motion compensation of one PU (8.5.3.3) from ref->out into f->pix ----
RBT_DEV int rc_refpix(const uint16_t* p, int w, int h, int x, int y) { return p[(size_t)rbt_clip3(0, h - 1, y) * w + rbt_clip3(0, w - 1, x)]; }
RBT_DEV void rc_mc_plane(uint16_t* dst, const uint16_t* ref, int pw, int ph, int x0, int y0, int bw, int bh, int xint, int yint, int xf, int yf, int taps,
                         const int8_t* fx, const int8_t* fy, int bd) {
  int sh1 = rbt_min(4, bd - 8), sh3 = 14 - bd, half = taps / 2 - 1, maxv = (1 << bd) - 1;
  int fsh = 14 - bd, fadd = fsh ? 1 << (fsh - 1) : 0;
  RBT_PAR_FOR(i, bw * bh) {
    int x = i % bw, y = i / bw, xi = x0 + xint + x, yi = y0 + yint + y, v;
    if (!xf && !yf) v = rc_refpix(ref, pw, ph, xi, yi) << sh3;
    else if (!yf) { int s = 0; for (int k = 0; k < taps; k++) s += fx[k] * rc_refpix(ref, pw, ph, xi + k - half, yi); v = s >> sh1; }
    else if (!xf) { int s = 0; for (int k = 0; k < taps; k++) s += fy[k] * rc_refpix(ref, pw, ph, xi, yi + k - half); v = s >> sh1; }
    else {
      int s = 0;
      for (int j = 0; j < taps; j++) {
        int t = 0;
        for (int k = 0; k < taps; k++) t += fx[k] * rc_refpix(ref, pw, ph, xi + k - half, yi + j - half);
        s += fy[j] * (t >> sh1);
      }
      v = s >> 6;
    }
    dst[(size_t)(y0 + y) * pw + x0 + x] = (uint16_t)rbt_clip3(0, maxv, (v + fadd) >> fsh);
  }
}
RBT_DEV void rc_inter_pu(RbtFrame* f, const RbtFrame* ref, int x0, int y0, int w, int h, int mvx, int mvy) {
  const RbtStreamCfg* g = &f->cfg;
  rc_mc_plane(f->pix[0], ref->out[0], g->w, g->h, x0, y0, w, h, mvx >> 2, mvy >> 2, mvx & 3, mvy & 3, 8, k_luma_filter[mvx & 3], k_luma_filter[mvy & 3], g->bit_depth);
  for (int c = 1; c < 3; c++)
    rc_mc_plane(f->pix[c], ref->out[c], g->cw, g->ch, x0 >> 1, y0 >> 1, w >> 1, h >> 1, mvx >> 3, mvy >> 3, mvx & 7, mvy & 7, 4, k_chroma_filter[mvx & 7], k_chroma_filter[mvy & 7], g->bit_depth);
  RBT_SYNC();
}

// ---- reconstruct one CTB of the decoder from its command list ----
RBT_DEV void rbt_recon_ctb(RbtFrame* frames, const RbtSlice* slices, int frame_idx, int ctb_addr, RBT_LDS_AS RbtReconLds* l) {
  RbtFrame* f = &frames[frame_idx];
  const RbtStreamCfg* g = &f->cfg;
  int cx = (ctb_addr % g->w_ctb) << g->log2_ctb, cy = (ctb_addr / g->w_ctb) << g->log2_ctb;
  uint32_t n = f->cmd_count[ctb_addr];
  if ((int)n > f->cmd_cap) n = (uint32_t)f->cmd_cap;
  const RbtCmd* cmds = f->cmds + (size_t)ctb_addr * f->cmd_cap;
  const RbtSlice* sl = &slices[f->ctb_slice[ctb_addr]];
  for (uint32_t k = 0; k < n; k++) {
    RbtCmd c = cmds[k];
    int x0 = cx + c.x4 * 4, y0 = cy + c.y4 * 4;
    if (c.type == RBT_CMD_PU) {
      rc_inter_pu(f, &frames[sl->ref_frame[c.c]], x0, y0, c.a * 4, c.b * 4, c.mvx, c.mvy);
    } else if (c.type == RBT_CMD_TU) {
      int fl = c.a, log2 = c.log2, intra = (fl & RBT_TU_INTRA) != 0;
      rc_decode_tb(f, 0, x0, y0, log2, intra, c.b, fl & RBT_TU_CBF_Y, (fl & RBT_TU_TS_Y) != 0, c.d, c.qp[0], l);
      if (fl & RBT_TU_CHROMA) {
        int xc = (log2 > 2 ? x0 : x0 - 4) >> 1, yc = (log2 > 2 ? y0 : y0 - 4) >> 1, l2c = log2 > 2 ? log2 - 1 : 2;
        rc_decode_tb(f, 1, xc, yc, l2c, intra, c.c, fl & RBT_TU_CBF_CB, (fl & RBT_TU_TS_CB) != 0, c.d, c.qp[1], l);
        rc_decode_tb(f, 2, xc, yc, l2c, intra, c.c, fl & RBT_TU_CBF_CR, (fl & RBT_TU_TS_CR) != 0, c.d, c.qp[2], l);
      }
    }
  }
}
