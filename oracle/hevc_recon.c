/* ORACLE — test infrastructure only (see oracle/README.md). Restates H.265 clause 8 reconstruction processes. */
#include "hevc_recon.h"

/* ------------------------------------------------------------------------------------------------ meta */
hevc_meta* hevc_meta_alloc(int w, int h, int log2_ctb) {
  hevc_meta* m = (hevc_meta*)calloc(1, sizeof(*m));
  m->w = w; m->h = h; m->w4 = (w + 3) / 4; m->h4 = (h + 3) / 4; m->log2_ctb = log2_ctb;
  m->w_ctb = (w + (1 << log2_ctb) - 1) >> log2_ctb; m->h_ctb = (h + (1 << log2_ctb) - 1) >> log2_ctb;
  size_t n = (size_t)m->w4 * m->h4;
  m->pred_mode = (uint8_t*)malloc(n); m->done = (uint8_t*)malloc(n); m->intra_mode = (uint8_t*)malloc(n);
  m->cu_depth = (uint8_t*)malloc(n); m->qp = (int8_t*)malloc(n); m->tq_bypass = (uint8_t*)malloc(n);
  m->nz = (uint8_t*)malloc(n); m->edge_v = (uint8_t*)malloc(n); m->edge_h = (uint8_t*)malloc(n);
  m->mv = (int16_t*)malloc(n * 4); m->ref_idx = (int8_t*)malloc(n);
  m->ctb_slice = (uint16_t*)malloc(sizeof(uint16_t) * m->w_ctb * m->h_ctb);
  m->sao = (hevc_sao*)malloc(sizeof(hevc_sao) * m->w_ctb * m->h_ctb);
  hevc_meta_reset(m);
  return m;
}
void hevc_meta_reset(hevc_meta* m) {
  size_t n = (size_t)m->w4 * m->h4;
  memset(m->pred_mode, META_UNDECODED, n); memset(m->done, 0, n); memset(m->intra_mode, 1, n);
  memset(m->cu_depth, 0, n); memset(m->qp, 0, n); memset(m->tq_bypass, 0, n); memset(m->nz, 0, n);
  memset(m->edge_v, 0, n); memset(m->edge_h, 0, n); memset(m->mv, 0, n * 4); memset(m->ref_idx, -1, n);
  memset(m->ctb_slice, 0, sizeof(uint16_t) * m->w_ctb * m->h_ctb);
  memset(m->sao, 0, sizeof(hevc_sao) * m->w_ctb * m->h_ctb);
  m->n_slices = 0;
}
void hevc_meta_free(hevc_meta* m) {
  if (!m) return;
  free(m->pred_mode); free(m->done); free(m->intra_mode); free(m->cu_depth); free(m->qp); free(m->tq_bypass);
  free(m->nz); free(m->edge_v); free(m->edge_h); free(m->mv); free(m->ref_idx); free(m->ctb_slice); free(m->sao); free(m);
}
int hevc_avail_intra(const hevc_meta* m, int xc, int yc, int xn, int yn) {
  if (xn < 0 || yn < 0 || xn >= m->w || yn >= m->h) return 0;
  int i = meta_idx(m, xn, yn);
  if (!m->done[i]) return 0;
  if (meta_slice_at(m, xn, yn) != meta_slice_at(m, xc, yc)) return 0;
  if (m->constrained_intra_pred && m->pred_mode[i] != MODE_INTRA) return 0;
  return 1;
}
int hevc_avail_cu(const hevc_meta* m, int xc, int yc, int xn, int yn) {
  if (xn < 0 || yn < 0 || xn >= m->w || yn >= m->h) return 0;
  if (m->pred_mode[meta_idx(m, xn, yn)] == META_UNDECODED) return 0;
  return meta_slice_at(m, xn, yn) == meta_slice_at(m, xc, yc);
}

/* ------------------------------------------------------------------------------------------------ transforms */
/* 8.6.3 with flat scaling lists (m = 16) */
void hevc_dequant(const int16_t* lvl, int16_t* d, int log2, int qp, int bit_depth) {
  int n = 1 << (2 * log2);
  int bd_shift = bit_depth + log2 - 5;
  int scale = (16 * k_dequant_scale[qp % 6]) << (qp / 6);
  int64_t add = 1LL << (bd_shift - 1);
  for (int i = 0; i < n; i++) {
    int64_t v = ((int64_t)lvl[i] * scale + add) >> bd_shift;
    d[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
}
static inline int tcoef(int N, int is_dst, int k, int n) { return is_dst ? k_dst4[k][n] : hevc_dct_coef(N, k, n); }

/* 8.6.4.2: columns first (shift 7, clip to 16 bit), then rows (shift 20 - bitDepth) */
void hevc_inv_transform(const int16_t* d, int16_t* res, int log2, int is_dst, int bit_depth) {
  int N = 1 << log2;
  int tmp[32 * 32];
  for (int x = 0; x < N; x++)
    for (int y = 0; y < N; y++) {
      int s = 0;
      for (int k = 0; k < N; k++) s += tcoef(N, is_dst, k, y) * d[k * N + x];
      tmp[y * N + x] = clip3(-32768, 32767, (s + 64) >> 7);
    }
  int sh = 20 - bit_depth;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      int s = 0;
      for (int k = 0; k < N; k++) s += tcoef(N, is_dst, k, x) * tmp[y * N + k];
      res[y * N + x] = (int16_t)((s + (1 << (sh - 1))) >> sh);
    }
}
/* 8.6.4.2 transform-skip branch (v1): r = d << 7, then bdShift = 20 - bitDepth */
void hevc_inv_transform_skip(const int16_t* d, int16_t* res, int log2, int bit_depth) {
  int n = 1 << (2 * log2), sh = 20 - bit_depth;
  for (int i = 0; i < n; i++) res[i] = (int16_t)((((int)d[i] << 7) + (1 << (sh - 1))) >> sh);
}
/* encoder-side forward transform: rows then columns, HM shift convention */
void hevc_fwd_transform(const int16_t* res, int16_t* coef, int log2, int is_dst, int bit_depth) {
  int N = 1 << log2;
  int tmp[32 * 32];
  int s1 = log2 + bit_depth - 9, s2 = log2 + 6;
  for (int y = 0; y < N; y++)
    for (int k = 0; k < N; k++) {
      int s = 0;
      for (int x = 0; x < N; x++) s += tcoef(N, is_dst, k, x) * res[y * N + x];
      tmp[y * N + k] = s1 > 0 ? (s + (1 << (s1 - 1))) >> s1 : s;
    }
  for (int kv = 0; kv < N; kv++)
    for (int kh = 0; kh < N; kh++) {
      int s = 0;
      for (int y = 0; y < N; y++) s += tcoef(N, is_dst, kv, y) * tmp[y * N + kh];
      coef[kv * N + kh] = (int16_t)clip3(-32768, 32767, (s + (1 << (s2 - 1))) >> s2);
    }
}
/* dead-zone scalar quantiser (no RDOQ, no sign hiding): intra 171/512, inter 85/512 */
int hevc_quant(const int16_t* coef, int16_t* lvl, int log2, int qp, int bit_depth, int is_intra) {
  int n = 1 << (2 * log2), nz = 0;
  int qbits = 14 + qp / 6 + (15 - bit_depth - log2);
  int64_t add = (int64_t)(is_intra ? 171 : 85) << (qbits - 9);
  int sc = k_quant_scale[qp % 6];
  for (int i = 0; i < n; i++) {
    int a = iabs(coef[i]);
    int64_t l = ((int64_t)a * sc + add) >> qbits;
    if (l > 32767) l = 32767;
    lvl[i] = (int16_t)(coef[i] < 0 ? -l : l);
    nz += l != 0;
  }
  return nz;
}

/* ------------------------------------------------------------------------------------------------ intra (8.4.4.2) */
void hevc_intra_pred_buf(const hevc_frame* f, const hevc_meta* m, int c_idx, int x0, int y0, int log2, int mode, uint16_t* pred) {
  int N = 1 << log2;
  int sh = c_idx ? 1 : 0;                 /* 4:2:0 */
  int pw = c_idx ? f->cw : f->w;
  const uint16_t* pl = f->p[c_idx];
  int bd = f->bit_depth;
  int maxv = (1 << bd) - 1;
  /* linear neighbour array: a[0] = p[-1][2N-1] ... a[2N-1] = p[-1][0], a[2N] = p[-1][-1], a[2N+1+x] = p[x][-1] */
  int a[4 * 32 + 1], av[4 * 32 + 1];
  int xcL = x0 << sh, ycL = y0 << sh;
  int any = 0;
  for (int i = 0; i <= 4 * N; i++) {
    int xn, yn;
    if (i < 2 * N) { xn = x0 - 1; yn = y0 + (2 * N - 1 - i); }
    else if (i == 2 * N) { xn = x0 - 1; yn = y0 - 1; }
    else { xn = x0 + (i - 2 * N - 1); yn = y0 - 1; }
    av[i] = hevc_avail_intra(m, xcL, ycL, xn << sh, yn << sh);
    if (av[i]) { a[i] = pl[(size_t)yn * pw + xn]; any = 1; } else a[i] = 0;
  }
  if (!any) { for (int i = 0; i <= 4 * N; i++) a[i] = 1 << (bd - 1); }
  else {
    if (!av[0]) { int j = 1; while (!av[j]) j++; a[0] = a[j]; av[0] = 1; }
    for (int i = 1; i <= 4 * N; i++) if (!av[i]) a[i] = a[i - 1];
  }
  /* filtering (8.4.4.2.3) */
  int filt = 0;
  if (c_idx == 0 && mode != 1 && N != 4) {
    int md = imin(iabs(mode - 26), iabs(mode - 10));
    int thr = N == 8 ? 7 : (N == 16 ? 1 : 0);
    filt = md > thr;
  }
  if (filt) {
    int fa[4 * 32 + 1];
    int corner = a[2 * N], bl = a[0], tr = a[4 * N];
    if (m->strong_intra_smoothing && N == 32 && iabs(corner + tr - 2 * a[2 * N + 32]) < (1 << (bd - 5)) &&
        iabs(corner + bl - 2 * a[2 * N - 32]) < (1 << (bd - 5))) {
      fa[2 * N] = corner; fa[0] = bl; fa[4 * N] = tr;
      for (int i = 0; i < 63; i++) {
        fa[2 * N - 1 - i] = ((63 - i) * corner + (i + 1) * bl + 32) >> 6;   /* p[-1][i] */
        fa[2 * N + 1 + i] = ((63 - i) * corner + (i + 1) * tr + 32) >> 6;   /* p[i][-1] */
      }
    } else {
      fa[0] = a[0]; fa[4 * N] = a[4 * N];
      for (int i = 1; i < 4 * N; i++) fa[i] = (a[i - 1] + 2 * a[i] + a[i + 1] + 2) >> 2;
    }
    memcpy(a, fa, sizeof(int) * (4 * N + 1));
  }
#define LEFT(y) a[2 * N - 1 - (y)]   /* p[-1][y], y = -1..2N-1 */
#define TOP(x) a[2 * N + 1 + (x)]    /* p[x][-1], x = -1..2N-1 */
  if (mode == 0) {
    for (int y = 0; y < N; y++)
      for (int x = 0; x < N; x++)
        pred[y * N + x] = (uint16_t)(((N - 1 - x) * LEFT(y) + (x + 1) * TOP(N) + (N - 1 - y) * TOP(x) + (y + 1) * LEFT(N) + N) >> (log2 + 1));
  } else if (mode == 1) {
    int s = N;
    for (int i = 0; i < N; i++) s += TOP(i) + LEFT(i);
    int dc = s >> (log2 + 1);
    for (int i = 0; i < N * N; i++) pred[i] = (uint16_t)dc;
    if (c_idx == 0 && N < 32) {
      pred[0] = (uint16_t)((LEFT(0) + 2 * dc + TOP(0) + 2) >> 2);
      for (int x = 1; x < N; x++) pred[x] = (uint16_t)((TOP(x) + 3 * dc + 2) >> 2);
      for (int y = 1; y < N; y++) pred[y * N] = (uint16_t)((LEFT(y) + 3 * dc + 2) >> 2);
    }
  } else {
    int ang = k_intra_angle[mode];
    int refb[3 * 32 + 2]; int* ref = refb + 32;     /* ref[-N .. 2N] */
    if (mode >= 18) {
      for (int x = 0; x <= N; x++) ref[x] = TOP(x - 1);
      if (ang < 0) {
        int last = (N * ang) >> 5;
        if (last < -1) { int inv = k_intra_inv_angle[mode - 11]; for (int x = last; x <= -1; x++) ref[x] = LEFT(-1 + ((x * inv + 128) >> 8)); }
      } else for (int x = N + 1; x <= 2 * N; x++) ref[x] = TOP(x - 1);
      for (int y = 0; y < N; y++) {
        int idx = ((y + 1) * ang) >> 5, fr = ((y + 1) * ang) & 31;
        for (int x = 0; x < N; x++)
          pred[y * N + x] = (uint16_t)(fr ? ((32 - fr) * ref[x + idx + 1] + fr * ref[x + idx + 2] + 16) >> 5 : ref[x + idx + 1]);
      }
      if (mode == 26 && c_idx == 0 && N < 32)
        for (int y = 0; y < N; y++) pred[y * N] = (uint16_t)clip3(0, maxv, TOP(0) + ((LEFT(y) - LEFT(-1)) >> 1));
    } else {
      for (int x = 0; x <= N; x++) ref[x] = LEFT(x - 1);
      if (ang < 0) {
        int last = (N * ang) >> 5;
        if (last < -1) { int inv = k_intra_inv_angle[mode - 11]; for (int x = last; x <= -1; x++) ref[x] = TOP(-1 + ((x * inv + 128) >> 8)); }
      } else for (int x = N + 1; x <= 2 * N; x++) ref[x] = LEFT(x - 1);
      for (int x = 0; x < N; x++) {
        int idx = ((x + 1) * ang) >> 5, fr = ((x + 1) * ang) & 31;
        for (int y = 0; y < N; y++)
          pred[y * N + x] = (uint16_t)(fr ? ((32 - fr) * ref[y + idx + 1] + fr * ref[y + idx + 2] + 16) >> 5 : ref[y + idx + 1]);
      }
      if (mode == 10 && c_idx == 0 && N < 32)
        for (int x = 0; x < N; x++) pred[x] = (uint16_t)clip3(0, maxv, LEFT(0) + ((TOP(x) - TOP(-1)) >> 1));
    }
  }
#undef LEFT
#undef TOP
}
void hevc_intra_pred(hevc_frame* f, const hevc_meta* m, int c_idx, int x0, int y0, int log2, int mode) {
  uint16_t pred[32 * 32];
  int N = 1 << log2, pw = c_idx ? f->cw : f->w;
  hevc_intra_pred_buf(f, m, c_idx, x0, y0, log2, mode, pred);
  for (int y = 0; y < N; y++) memcpy(f->p[c_idx] + (size_t)(y0 + y) * pw + x0, pred + y * N, 2 * N);
}

void hevc_intra_mpm(const hevc_meta* m, int xp, int yp, int cand[3]) {
  int ca = 1, cb = 1;
  if (hevc_avail_cu(m, xp, yp, xp - 1, yp) && m->pred_mode[meta_idx(m, xp - 1, yp)] == MODE_INTRA) ca = m->intra_mode[meta_idx(m, xp - 1, yp)];
  if (hevc_avail_cu(m, xp, yp, xp, yp - 1) && m->pred_mode[meta_idx(m, xp, yp - 1)] == MODE_INTRA && ((yp - 1) >> m->log2_ctb) == (yp >> m->log2_ctb)) cb = m->intra_mode[meta_idx(m, xp, yp - 1)];
  if (ca == cb) {
    if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
    else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
  } else {
    cand[0] = ca; cand[1] = cb;
    cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26);
  }
}

/* ------------------------------------------------------------------------------------------------ inter (8.5.3.3) */
static inline int refpix(const uint16_t* p, int w, int h, int x, int y) { return p[(size_t)clip3(0, h - 1, y) * w + clip3(0, w - 1, x)]; }

static void mc_block(uint16_t* dst, int dw, const uint16_t* ref, int rw, int rh, int x0, int y0, int bw, int bh,
                     int xint, int yint, int xf, int yf, int taps, const int8_t* fx, const int8_t* fy, int bd, const int* wp) {   /* wp: {w0, o0, log2WD} or NULL */
  int sh1 = imin(4, bd - 8), sh2 = 6, sh3 = 14 - bd;
  int half = taps / 2 - 1;   /* taps before the integer position */
  int maxv = (1 << bd) - 1;
  int fsh = 14 - bd, fadd = fsh ? 1 << (fsh - 1) : 0;
  static int tmp[(64 + 8) * 64];
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      int v;
      int xi = x0 + xint + x, yi = y0 + yint + y;
      if (!xf && !yf) v = refpix(ref, rw, rh, xi, yi) << sh3;
      else if (!yf) { int s = 0; for (int k = 0; k < taps; k++) s += fx[k] * refpix(ref, rw, rh, xi + k - half, yi); v = s >> sh1; }
      else if (!xf) { int s = 0; for (int k = 0; k < taps; k++) s += fy[k] * refpix(ref, rw, rh, xi, yi + k - half); v = s >> sh1; }
      else {
        int s = 0;
        for (int j = 0; j < taps; j++) {
          int t = 0;
          for (int k = 0; k < taps; k++) t += fx[k] * refpix(ref, rw, rh, xi + k - half, yi + j - half);
          s += fy[j] * (t >> sh1);
        }
        v = s >> sh2;
      }
      (void)tmp;
      if (wp) {   /* 8.5.3.3.4.3 (8-265): explicit weighting of the 14-bit intermediate */
        const int r = wp[2] >= 1 ? ((v * wp[0] + (1 << (wp[2] - 1))) >> wp[2]) + wp[1] : v * wp[0] + wp[1];
        dst[(size_t)(y0 + y) * dw + x0 + x] = (uint16_t)clip3(0, maxv, r);
      } else
      dst[(size_t)(y0 + y) * dw + x0 + x] = (uint16_t)clip3(0, maxv, (v + fadd) >> fsh);
    }
}
void hevc_inter_pred_wp(hevc_frame* f, const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy, const hevc_wp* wp) {
  int t[3][3];
  if (wp) for (int c = 0; c < 3; c++) { t[c][0] = wp->w[c]; t[c][1] = wp->o[c]; t[c][2] = wp->shift[c != 0]; }
  mc_block(f->p[0], f->w, ref->p[0], ref->w, ref->h, x0, y0, w, h, mvx >> 2, mvy >> 2, mvx & 3, mvy & 3, 8,
           k_luma_filter[mvx & 3], k_luma_filter[mvy & 3], f->bit_depth, wp ? t[0] : NULL);
  for (int c = 1; c < 3; c++)
    mc_block(f->p[c], f->cw, ref->p[c], ref->cw, ref->ch, x0 / 2, y0 / 2, w / 2, h / 2, mvx >> 3, mvy >> 3, mvx & 7, mvy & 7, 4,
             k_chroma_filter[mvx & 7], k_chroma_filter[mvy & 7], f->bit_depth, wp ? t[c] : NULL);
}
void hevc_inter_pred(hevc_frame* f, const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy) { hevc_inter_pred_wp(f, ref, x0, y0, w, h, mvx, mvy, NULL); }

/* luma prediction block (8.5.3.3.3 + the uni-directional weighting of 8.5.3.3.4.2) into out[w*h]; separable evaluation of the same
 * arithmetic as mc_block (row filter >> shift1, column filter >> 6), used by the HM-like encoder's motion search */
void hevc_mc_luma_buf(const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy, uint16_t* out) {
  int bd = ref->bit_depth, sh1 = imin(4, bd - 8), sh3 = 14 - bd, maxv = (1 << bd) - 1, fsh = 14 - bd, fadd = fsh ? 1 << (fsh - 1) : 0;
  int xf = mvx & 3, yf = mvy & 3, xi0 = x0 + (mvx >> 2), yi0 = y0 + (mvy >> 2);
  const int8_t* fx = k_luma_filter[xf]; const int8_t* fy = k_luma_filter[yf];
  static int row[(64 + 8) * 64];
  for (int y = -3; y < h + 4; y++) {
    if (!yf && (y < 0 || y >= h)) continue;
    for (int x = 0; x < w; x++) {
      int v;
      if (!xf) v = refpix(ref->p[0], ref->w, ref->h, xi0 + x, yi0 + y);
      else { int s = 0; for (int k = 0; k < 8; k++) s += fx[k] * refpix(ref->p[0], ref->w, ref->h, xi0 + x + k - 3, yi0 + y); v = s >> sh1; }
      row[(y + 3) * 64 + x] = v;
    }
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v;
      if (!yf) v = xf ? row[(y + 3) * 64 + x] : row[(y + 3) * 64 + x] << sh3;
      else { int s = 0; for (int k = 0; k < 8; k++) s += fy[k] * row[(y + k) * 64 + x]; v = xf ? s >> 6 : s >> sh1; }
      out[y * w + x] = (uint16_t)clip3(0, maxv, (v + fadd) >> fsh);
    }
}

/* ------------------------------------------------------------------------------------------------ deblocking (8.7.2) */
/* bS of the edge between 4x4 units ip (P side) and iq (Q side); tu = transform edge */
static int edge_bs(const hevc_meta* m, int ip, int iq, int tu, const hevc_slice_meta* sp, const hevc_slice_meta* sq) {
  if (m->pred_mode[ip] == MODE_INTRA || m->pred_mode[iq] == MODE_INTRA) return 2;
  if (tu && (m->nz[ip] || m->nz[iq])) return 1;
  int rp = sp->ref_poc[m->ref_idx[ip] < 0 ? 0 : m->ref_idx[ip]], rq = sq->ref_poc[m->ref_idx[iq] < 0 ? 0 : m->ref_idx[iq]];
  if (rp != rq) return 1;
  if (iabs(m->mv[2 * ip] - m->mv[2 * iq]) >= 4 || iabs(m->mv[2 * ip + 1] - m->mv[2 * iq + 1]) >= 4) return 1;
  return 0;
}
/* returns bS for the edge on the left (dir 0) or top (dir 1) side of the 4x4 unit at luma (x,y); 0 = no filtering */
static int unit_bs(const hevc_meta* m, int x, int y, int dir) {
  int iq = meta_idx(m, x, y);
  int e = dir == 0 ? m->edge_v[iq] : m->edge_h[iq];
  if (!e) return 0;
  int xp = dir == 0 ? x - 1 : x, yp = dir == 0 ? y : y - 1;
  if (xp < 0 || yp < 0) return 0;
  int ip = meta_idx(m, xp, yp);
  const hevc_slice_meta* sq = &m->slices[meta_slice_at(m, x, y)];
  const hevc_slice_meta* sp = &m->slices[meta_slice_at(m, xp, yp)];
  if (sq->deblocking_disabled) return 0;
  if (sp != sq && !sq->loop_filter_across) return 0;
  return edge_bs(m, ip, iq, e & 1, sp, sq);
}

static void deblock_luma_edge(hevc_frame* f, const hevc_meta* m, int x, int y, int dir, int bs) {
  /* one 4-sample segment; (x,y) is the first Q sample; dir 0 = vertical edge (filter across x) */
  int bd = f->bit_depth, maxv = (1 << bd) - 1;
  uint16_t* pl = f->p[0]; int st = f->w;
  int sa = dir == 0 ? 1 : st;          /* step across the edge */
  int sl = dir == 0 ? st : 1;          /* step along the edge */
  int iq = meta_idx(m, x, y), ip = dir == 0 ? meta_idx(m, x - 1, y) : meta_idx(m, x, y - 1);
  const hevc_slice_meta* sq = &m->slices[meta_slice_at(m, x, y)];
  int qpl = (m->qp[iq] + m->qp[ip] + 1) >> 1;
  int beta = k_beta_table[clip3(0, 51, qpl + (sq->beta_offset_div2 << 1))] * (1 << (bd - 8));
  int tc = k_tc_table[clip3(0, 53, qpl + 2 * (bs - 1) + (sq->tc_offset_div2 << 1))] * (1 << (bd - 8));
  uint16_t* q = pl + (size_t)y * st + x;
#define P(i, k) ((int)q[-((i) + 1) * sa + (k) * sl])
#define Q(i, k) ((int)q[(i) * sa + (k) * sl])
  int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
  int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
  if (d >= beta) return;
  int ds0 = 2 * dpq0 < (beta >> 2) && iabs(P(3, 0) - P(0, 0)) + iabs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && iabs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
  int ds3 = 2 * dpq3 < (beta >> 2) && iabs(P(3, 3) - P(0, 3)) + iabs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && iabs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
  int strong = ds0 && ds3;
  int dEp = dp < ((beta + (beta >> 1)) >> 3), dEq = dq < ((beta + (beta >> 1)) >> 3);
  int no_p = m->tq_bypass[ip], no_q = m->tq_bypass[iq];
  for (int k = 0; k < 4; k++) {
    int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0 = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
    uint16_t* c = q + k * sl;
    if (strong) {
      if (!no_p) {
        c[-1 * sa] = (uint16_t)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        c[-2 * sa] = (uint16_t)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
        c[-3 * sa] = (uint16_t)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (!no_q) {
        c[0] = (uint16_t)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        c[sa] = (uint16_t)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
        c[2 * sa] = (uint16_t)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (iabs(delta) < tc * 10) {
        delta = clip3(-tc, tc, delta);
        if (!no_p) {
          c[-sa] = (uint16_t)clip3(0, maxv, p0 + delta);
          if (dEp) c[-2 * sa] = (uint16_t)clip3(0, maxv, p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
        }
        if (!no_q) {
          c[0] = (uint16_t)clip3(0, maxv, q0 - delta);
          if (dEq) c[sa] = (uint16_t)clip3(0, maxv, q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
        }
      }
    }
  }
#undef P
#undef Q
}
static void deblock_chroma_edge(hevc_frame* f, const hevc_meta* m, int c_idx, int xl, int yl, int dir) {
  /* (xl,yl): luma coords of the first Q sample of a 4-luma-sample segment (2 chroma samples) with bS == 2 */
  int bd = f->bit_depth, maxv = (1 << bd) - 1;
  uint16_t* pl = f->p[c_idx]; int st = f->cw;
  int sa = dir == 0 ? 1 : st, sl = dir == 0 ? st : 1;
  int iq = meta_idx(m, xl, yl), ip = dir == 0 ? meta_idx(m, xl - 1, yl) : meta_idx(m, xl, yl - 1);
  const hevc_slice_meta* sq = &m->slices[meta_slice_at(m, xl, yl)];
  int off = c_idx == 1 ? m->cb_qp_offset : m->cr_qp_offset;
  int qpc = hevc_chroma_qp(((m->qp[iq] + m->qp[ip] + 1) >> 1) + off);
  int tc = k_tc_table[clip3(0, 53, qpc + 2 + (sq->tc_offset_div2 << 1))] * (1 << (bd - 8));
  uint16_t* q = pl + (size_t)(yl / 2) * st + xl / 2;
  for (int k = 0; k < 2; k++) {
    uint16_t* c = q + k * sl;
    int p0 = c[-sa], p1 = c[-2 * sa], q0 = c[0], q1 = c[sa];
    int delta = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!m->tq_bypass[ip]) c[-sa] = (uint16_t)clip3(0, maxv, p0 + delta);
    if (!m->tq_bypass[iq]) c[0] = (uint16_t)clip3(0, maxv, q0 - delta);
  }
}
void hevc_deblock(hevc_frame* f, const hevc_meta* m) {
  uint8_t* bsmap = (uint8_t*)malloc((size_t)m->w4 * m->h4);
  for (int dir = 0; dir < 2; dir++) {
    /* bS derivation uses pre-deblock metadata only, so deriving per direction up front is equivalent */
    for (int y = 0; y < m->h; y += 4)
      for (int x = 0; x < m->w; x += 4) {
        int on_grid = dir == 0 ? (x & 7) == 0 : (y & 7) == 0;
        bsmap[meta_idx(m, x, y)] = on_grid ? (uint8_t)unit_bs(m, x, y, dir) : 0;
      }
    for (int y = 0; y < m->h; y += 4)
      for (int x = 0; x < m->w; x += 4) { int bs = bsmap[meta_idx(m, x, y)]; if (bs) deblock_luma_edge(f, m, x, y, dir, bs); }
    for (int y = 0; y < m->h; y += 4)
      for (int x = 0; x < m->w; x += 4) {
        int on_grid = dir == 0 ? (x & 15) == 0 : (y & 15) == 0;
        if (on_grid && bsmap[meta_idx(m, x, y)] == 2) { deblock_chroma_edge(f, m, 1, x, y, dir); deblock_chroma_edge(f, m, 2, x, y, dir); }
      }
  }
  free(bsmap);
}

/* ------------------------------------------------------------------------------------------------ SAO (8.7.3) */
void hevc_sao_apply(hevc_frame* dst, const hevc_frame* src, const hevc_meta* m) {
  hevc_frame_copy(dst, src);
  int bd = src->bit_depth, maxv = (1 << bd) - 1;
  int ctb = 1 << m->log2_ctb;
  static const int8_t eo_dx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}};
  static const int8_t eo_dy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
  for (int cy = 0; cy < m->h_ctb; cy++)
    for (int cx = 0; cx < m->w_ctb; cx++) {
      const hevc_sao* s = &m->sao[cy * m->w_ctb + cx];
      const hevc_slice_meta* sl = &m->slices[m->ctb_slice[cy * m->w_ctb + cx]];
      for (int c = 0; c < 3; c++) {
        if (!(c ? sl->sao_chroma : sl->sao_luma) || !s->type[c]) continue;
        int sh = c ? 1 : 0;
        int pw = c ? src->cw : src->w, ph = c ? src->ch : src->h;
        int x0 = (cx * ctb) >> sh, y0 = (cy * ctb) >> sh;
        int x1 = imin(pw, x0 + (ctb >> sh)), y1 = imin(ph, y0 + (ctb >> sh));
        const uint16_t* sp = src->p[c]; uint16_t* dp = dst->p[c];
        int band_tab[32]; memset(band_tab, 0, sizeof(band_tab));
        if (s->type[c] == 1) for (int k = 0; k < 4; k++) band_tab[(k + s->band_pos[c]) & 31] = k + 1;
        for (int y = y0; y < y1; y++)
          for (int x = x0; x < x1; x++) {
            int iu = meta_idx(m, x << sh, y << sh);
            if (m->tq_bypass[iu]) continue;      /* pcm_loop_filter_disabled handled by the caller setting tq_bypass */
            int v = sp[(size_t)y * pw + x], off = 0;
            if (s->type[c] == 1) { int b = band_tab[v >> (bd - 5)]; off = b ? s->offset[c][b - 1] : 0; }
            else {
              int cls = s->eo_class[c];
              int xa = x + eo_dx[cls][0], ya = y + eo_dy[cls][0], xb = x + eo_dx[cls][1], yb = y + eo_dy[cls][1];
              if (xa < 0 || ya < 0 || xb < 0 || yb < 0 || xa >= pw || xb >= pw || ya >= ph || yb >= ph) continue;
              /* neighbours in another slice whose filtering across slice boundaries is off are treated as unavailable */
              int sa_ = meta_slice_at(m, xa << sh, ya << sh), sb_ = meta_slice_at(m, xb << sh, yb << sh), sc_ = meta_slice_at(m, x << sh, y << sh);
              if ((sa_ != sc_ && !(m->slices[sa_ > sc_ ? sa_ : sc_].loop_filter_across)) || (sb_ != sc_ && !(m->slices[sb_ > sc_ ? sb_ : sc_].loop_filter_across))) continue;
              int va = sp[(size_t)ya * pw + xa], vb = sp[(size_t)yb * pw + xb];
              int e = 2 + (v > va) - (v < va) + (v > vb) - (v < vb);
              if (e == 0 || e == 1 || e == 2) e = (e == 2) ? 0 : e + 1;
              off = e ? s->offset[c][e - 1] : 0;
            }
            dp[(size_t)y * pw + x] = (uint16_t)clip3(0, maxv, v + off);
          }
      }
    }
}

/* ------------------------------------------------------------------------------------------------ MV prediction */
#include <limits.h>
typedef hevc_mvcand mvcand;
static int pu_avail(const hevc_mvpred* s, int xc, int yc, int xn, int yn) {
  const hevc_meta* m = s->m;
  if (!hevc_avail_cu(m, xc, yc, xn, yn)) return 0;
  return m->pred_mode[meta_idx(m, xn, yn)] != MODE_INTRA;
}
static inline mvcand mv_at(const hevc_meta* m, int x, int y) { int i = meta_idx(m, x, y); mvcand c = {m->mv[2 * i], m->mv[2 * i + 1], m->ref_idx[i]}; return c; }
static inline int mv_same(mvcand a, mvcand b) { return a.x == b.x && a.y == b.y && a.ref == b.ref; }
static int scale_mv(int mv, int tb, int td) {
  td = clip3(-128, 127, td); tb = clip3(-128, 127, tb);
  int tx = (16384 + (iabs(td) >> 1)) / td;
  int dsf = clip3(-4096, 4095, (tb * tx + 32) >> 6);
  int p = dsf * mv;
  return clip3(-32768, 32767, (p < 0 ? -1 : 1) * ((iabs(p) + 127) >> 8));
}
/* temporal candidate (8.5.3.2.8) for list 0 with target ref_idx */
static int temporal_cand(const hevc_mvpred* s, int xpb, int ypb, int w, int h, int ref_idx, mvcand* out) {
  const hevc_colinfo* col = s->col;
  if (!col) return 0;
  int cand_xy[2][2] = {{xpb + w, ypb + h}, {xpb + (w >> 1), ypb + (h >> 1)}};
  for (int k = 0; k < 2; k++) {
    int x = cand_xy[k][0], y = cand_xy[k][1];
    if (k == 0 && ((ypb >> s->log2_ctb) != (y >> s->log2_ctb) || x >= s->pic_w || y >= s->pic_h)) continue;
    x = (x >> 4) << 4; y = (y >> 4) << 4;
    int i = (y >> 2) * col->w4 + (x >> 2);
    if (col->refpoc[i] == INT_MIN) continue;
    int td = col->poc - col->refpoc[i], tb = s->cur_poc - s->ref_poc[ref_idx];
    int mx = col->mv[2 * i], my = col->mv[2 * i + 1];
    if (td != tb && td != 0) { mx = scale_mv(mx, tb, td); my = scale_mv(my, tb, td); }
    out->x = (int16_t)mx; out->y = (int16_t)my; out->ref = ref_idx;
    return 1;
  }
  return 0;
}
hevc_mvcand hevc_merge_candidate(const hevc_mvpred* s, int xpb, int ypb, int w, int h, int part_idx, int merge_idx) {
  const hevc_meta* m = s->m; int pm = s->part_mode;
  mvcand list[6]; int n = 0;
  int xa1 = xpb - 1, ya1 = ypb + h - 1, xb1 = xpb + w - 1, yb1 = ypb - 1;
  int a1 = pu_avail(s, xpb, ypb, xa1, ya1) && !((pm == PART_Nx2N || pm == PART_nLx2N || pm == PART_nRx2N) && part_idx == 1);
  mvcand ca1 = {0, 0, 0}, cb1 = {0, 0, 0};
  if (a1) { ca1 = mv_at(m, xa1, ya1); list[n++] = ca1; }
  int b1 = pu_avail(s, xpb, ypb, xb1, yb1) && !((pm == PART_2NxN || pm == PART_2NxnU || pm == PART_2NxnD) && part_idx == 1);
  int b1_in = 0, b0_in = 0, a0_in = 0;
  if (b1) { cb1 = mv_at(m, xb1, yb1); if (!(a1 && mv_same(ca1, cb1))) { list[n++] = cb1; b1_in = 1; } }
  if (pu_avail(s, xpb, ypb, xpb + w, ypb - 1)) { mvcand c = mv_at(m, xpb + w, ypb - 1); if (!(b1 && mv_same(cb1, c))) { list[n++] = c; b0_in = 1; } }
  if (pu_avail(s, xpb, ypb, xpb - 1, ypb + h)) { mvcand c = mv_at(m, xpb - 1, ypb + h); if (!(a1 && mv_same(ca1, c))) { list[n++] = c; a0_in = 1; } }
  if (a1 + b1_in + b0_in + a0_in != 4 && pu_avail(s, xpb, ypb, xpb - 1, ypb - 1)) {
    mvcand c = mv_at(m, xpb - 1, ypb - 1);
    if (!(a1 && mv_same(ca1, c)) && !(b1 && mv_same(cb1, c))) list[n++] = c;
  }
  if (n > s->max_merge_cand) n = s->max_merge_cand;
  if (n < s->max_merge_cand) { mvcand t; if (temporal_cand(s, xpb, ypb, w, h, 0, &t)) list[n++] = t; }
  int zero_idx = 0;
  while (n < s->max_merge_cand) { mvcand z = {0, 0, zero_idx < s->num_ref_idx ? zero_idx : 0}; list[n++] = z; zero_idx++; }
  return list[merge_idx];
}
hevc_mvcand hevc_amvp_candidate(const hevc_mvpred* s, int xpb, int ypb, int w, int h, int ref_idx, int mvp_flag) {
  const hevc_meta* m = s->m;
  int tgt_poc = s->ref_poc[ref_idx], cur = s->cur_poc;
  int xa[2] = {xpb - 1, xpb - 1}, ya[2] = {ypb + h, ypb + h - 1};
  int xb[3] = {xpb + w, xpb + w - 1, xpb - 1}, yb[3] = {ypb - 1, ypb - 1, ypb - 1};
  int ava[2], avb[3];
  for (int k = 0; k < 2; k++) ava[k] = pu_avail(s, xpb, ypb, xa[k], ya[k]);
  for (int k = 0; k < 3; k++) avb[k] = pu_avail(s, xpb, ypb, xb[k], yb[k]);
  int fa = 0, fb = 0; mvcand ma = {0, 0, 0}, mb = {0, 0, 0};
  for (int k = 0; k < 2 && !fa; k++) if (ava[k]) { mvcand c = mv_at(m, xa[k], ya[k]); if (s->ref_poc[c.ref] == tgt_poc) { ma = c; fa = 1; } }
  for (int k = 0; k < 2 && !fa; k++) if (ava[k]) {
    mvcand c = mv_at(m, xa[k], ya[k]); int td = cur - s->ref_poc[c.ref], tb = cur - tgt_poc;
    ma = c; fa = 1; if (td != tb && td != 0) { ma.x = (int16_t)scale_mv(c.x, tb, td); ma.y = (int16_t)scale_mv(c.y, tb, td); }
  }
  int is_scaled = ava[0] || ava[1];
  for (int k = 0; k < 3 && !fb; k++) if (avb[k]) { mvcand c = mv_at(m, xb[k], yb[k]); if (s->ref_poc[c.ref] == tgt_poc) { mb = c; fb = 1; } }
  if (!is_scaled && fb) { ma = mb; fa = 1; }
  if (!is_scaled) {
    fb = 0;
    for (int k = 0; k < 3 && !fb; k++) if (avb[k]) {
      mvcand c = mv_at(m, xb[k], yb[k]); int td = cur - s->ref_poc[c.ref], tb = cur - tgt_poc;
      mb = c; fb = 1; if (td != tb && td != 0) { mb.x = (int16_t)scale_mv(c.x, tb, td); mb.y = (int16_t)scale_mv(c.y, tb, td); }
    }
  }
  mvcand list[3]; int n = 0;
  if (fa) list[n++] = ma;
  if (fb && !(fa && ma.x == mb.x && ma.y == mb.y)) list[n++] = mb;
  if (n < 2) { mvcand t; if (temporal_cand(s, xpb, ypb, w, h, ref_idx, &t)) list[n++] = t; }
  while (n < 2) { mvcand z = {0, 0, ref_idx}; list[n++] = z; }
  mvcand r = list[mvp_flag]; r.ref = ref_idx;
  return r;
}
