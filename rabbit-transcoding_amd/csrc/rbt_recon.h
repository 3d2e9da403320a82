// Picture reconstruction device routines (H.265 8.4.4.2 intra, 8.5.3.3 inter, 8.6 scaling + inverse transforms),
// executed by ONE 64-lane wave per CTB (wave-synchronous: a workgroup barrier of a single-wave workgroup is free),
// lanes spread over the samples of the current block, tiles staged in LDS.
// Replaces the reconstruction half of libavcodec's hevc decoder as driven by PCCTranscoder.cpp:428-448, and is shared
// with the encoder's reconstruction loop (rbt_encode.h).
#pragma once
#include "rbt_tables.h"
#include "rbt_types.h"
#include "rbt_mfma.h"

// Scratch of one TB. The decoder's CTB kernel declares only the core (its smoothed / angular reference arrays alias `tmp`,
// which is dead once the residual is in `res`): LDS per workgroup decides how many CTBs are in flight on a CU.
struct RbtReconLdsCore {
  int32_t nb[132];        // neighbour samples: [0] = p[-1][2N-1] .. [2N-1] = p[-1][0], [2N] = corner, [2N+1+x] = p[x][-1]
  alignas(16) int16_t res[32 * 32];   // dequantised coefficients, then residual
  alignas(16) int16_t tmp[32 * 32];   // first transform stage (16-bit by construction: 8.6.4.2 clips it, the forward stage's shift keeps it below 2^15)
  int8_t dct[32 * 32]; int8_t dst[16];   // transform matrices, staged once per workgroup (rc_stage_tables)
};
struct RbtReconLds : RbtReconLdsCore {   // encoder kernels: prediction kept next to the residual
  int32_t nbf[132];       // filtered neighbours
  int32_t ref[100];       // angular reference array, index offset 32
  int32_t ref2[100];      // second angular reference array (Cb / Cr processed together)
  uint16_t pred[32 * 32];
};
RBT_DEV void rc_stage_tables(RBT_LDS_AS RbtReconLdsCore* l) {
  RBT_PAR_FOR(i, 1024) l->dct[i] = k_dct32[i >> 5][i & 31];
  RBT_PAR_FOR(i, 16) l->dst[i] = k_dst4[i >> 2][i & 3];
  RBT_SYNC_LDS();
}

RBT_DEV int rc_morton(int x4, int y4) {
  int z = 0;
  for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1));
  return z;
}
// rc_morton(ax, ay) < rc_morton(bx, by) without interleaving anything: the coordinate whose difference has the higher top bit decides (y on a tie: its bits are the upper ones)
RBT_DEV int rc_z_before(int ax, int ay, int bx, int by) {
  const int dx = ax ^ bx, dy = ay ^ by, x_decides = dy < dx && dy < (dx ^ dy);
  return x_decides ? ax < bx : ay < by;
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
RBT_DEV int rc_tcoef(const RBT_LDS_AS RbtReconLdsCore* l, int N, int is_dst, int k, int n) { return is_dst ? l->dst[k * 4 + n] : l->dct[k * (32 / N) * 32 + n]; }

// Small normative tables as packed immediates (no memory access on the dependent path of a TB):
// intraPredAngle (Table 8-5), invAngle (Table 8-6) and levelScale (8.6.3)
RBT_DEV int rc_intra_angle(int mode) {           // modes 2..34
  const int k = mode <= 18 ? mode - 2 : 34 - mode, d = k < 8 ? 8 - k : k - 8;                 // |angle| = M[d], M = {0,2,5,9,13,17,21,26,32}
  const uint64_t M = 0ull | (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) | (26ull << 42) | (32ull << 48);
  const int a = (int)((M >> (6 * d)) & 63);
  return k <= 8 ? a : -a;
}
RBT_DEV int rc_intra_inv_angle(int mode) {       // modes 11..25
  const int d = mode < 18 ? 18 - mode : mode - 18;                                             // invAngle = -IA[d]
  const uint64_t lo = 256ull | (315ull << 16) | (390ull << 32) | (482ull << 48), hi = 630ull | (910ull << 16) | (1638ull << 32) | (4096ull << 48);
  return -(int)(((d < 4 ? lo : hi) >> (16 * (d & 3))) & 0xFFFF);
}
RBT_DEV int rc_level_scale(int r) { return (int)((0x484039332D28ull >> (8 * r)) & 255); }      // {40,45,51,57,64,72}[r]
// sum of v over the lanes of the wave (host emulation: the PAR_FOR around the call already accumulated everything)
RBT_DEV int rbt_wave_sum(int v) {
#ifdef RBT_HOSTEMU
  return v;
#else
  // DPP butterflies inside each row of 16 lanes (quad swaps, half-row mirror, row mirror), then the four row sums
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);    // quad_perm:[1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);    // quad_perm:[2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);   // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);   // row_mirror
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
#endif
}
// ---- the 4N+1 reference samples of a TB: which one each is taken from (8.4.4.2.2) ----
// Availability is a property of 4x4 luma units, so the neighbours come in UNITS of us = 4 >> sh samples (sh = 1 for chroma) that share one flag: nu = 2N / us units up the
// left column (unit 0 at the bottom), the corner (unit nu), nu units along the row above (up to unit 2 nu <= 32). One ballot over the units gives the whole picture; the
// substitution rule - a sample that is not available takes the nearest available one below it in index order, the ones before the first available one take that one -
// is then one median per sample whenever the available units form ONE run (always, but for slices that begin inside a CTB row and constrained intra prediction), and a
// find-last-set on the unit mask otherwise. (Rounds 1-3 tested every sample's unit and searched three 64-bit sample masks per lane: 0.6 G of the reconstruction's
// 3.1 G instructions per GOF, tools/ablate_run.sh.)
struct RcNbMap { uint64_t m; int two_n, usl, nu, lo, hi, run; };   // m: the units' availability; lo / hi: first / last available sample index; run: one run of units
RBT_DEV int rc_nb_unit_of(const RcNbMap* q, int i) { return i < q->two_n ? i >> q->usl : (i == q->two_n ? q->nu : q->nu + 1 + ((i - q->two_n - 1) >> q->usl)); }
RBT_DEV int rc_nb_unit_lo(const RcNbMap* q, int u) { return u < q->nu ? u << q->usl : (u == q->nu ? q->two_n : q->two_n + 1 + ((u - q->nu - 1) << q->usl)); }
RBT_DEV int rc_nb_unit_hi(const RcNbMap* q, int u) { return u < q->nu ? (u << q->usl) + (1 << q->usl) - 1 : (u == q->nu ? q->two_n : q->two_n + ((u - q->nu) << q->usl)); }
// position (relative to the TB's plane origin, like x0 / y0) of one sample of unit p
RBT_DEV void rc_nb_unit_xy(int p, int x0, int y0, int N, int sh, int* xn, int* yn) {
  const int usl = 2 - sh, nu = (2 * N) >> usl;
  if (p < nu) { *xn = x0 - 1; *yn = y0 + 2 * N - 1 - (p << usl); }
  else if (p == nu) { *xn = x0 - 1; *yn = y0 - 1; }
  else { *xn = x0 + ((p - nu - 1) << usl); *yn = y0 - 1; }
}
RBT_DEV int rc_nb_units(int N, int sh) { return ((4 * N) >> (2 - sh)) + 1; }
RBT_DEV void rc_nb_map(RcNbMap* q, uint64_t m, int N, int sh) {
  q->m = m; q->two_n = 2 * N; q->usl = 2 - sh; q->nu = (2 * N) >> q->usl; q->lo = q->hi = -1; q->run = 0;
  if (m) {
    const int ua = __builtin_ctzll(m), ub = 63 - __builtin_clzll(m); const uint64_t t = m >> ua;
    q->lo = rc_nb_unit_lo(q, ua); q->hi = rc_nb_unit_hi(q, ub); q->run = (t & (t + 1)) == 0;
  }
}
// index of the sample that reference sample i is taken from (itself when available); the caller has checked q->m != 0
RBT_DEV int rc_nb_source(const RcNbMap* q, int i) {
  if (q->run) return rbt_clip3(q->lo, q->hi, i);
  const int u = rc_nb_unit_of(q, i); const uint64_t t = q->m & ((2ull << u) - 1);
  if (!t) return q->lo;
  const int uj = 63 - __builtin_clzll(t);
  return uj == u ? i : rc_nb_unit_hi(q, uj);
}
// Substitution (8.4.4.2.2) + smoothing (8.4.4.2.3) of the gathered neighbours; `have_nb` = 0 when the caller wants the
// substituted samples fetched through `fetch(j)` semantics instead (tile variant fills l->nb itself). Returns the array that
// holds the final reference samples (l->nb or l->nbf: the two are swapped, never copied).
RBT_DEV int rc_intra_filter_needed(int c_idx, int log2, int mode) {       // filterFlag of 8.4.4.2.3
  const int N = 1 << log2;
  if (c_idx != 0 || mode == 1 || N == 4) return 0;
  const int md = rbt_min(rbt_abs(mode - 26), rbt_abs(mode - 10)), thr = N == 8 ? 7 : (N == 16 ? 1 : 0);
  return md > thr;
}
RBT_DEV void rc_intra_filter_apply(const RbtStreamCfg* g, int log2, RBT_LDS_AS int32_t* nb, RBT_LDS_AS int32_t* alt) {   // nb -> alt
  const int N = 1 << log2, bd = g->bit_depth, tot = 4 * N + 1;
  int corner = nb[2 * N], bl = nb[0], tr = nb[4 * N];
  int strong = g->strong_intra && N == 32 && rbt_abs(corner + tr - 2 * nb[2 * N + 32]) < (1 << (bd - 5)) &&
               rbt_abs(corner + bl - 2 * nb[2 * N - 32]) < (1 << (bd - 5));
  RBT_PAR_FOR(i, tot) {
    int v;
    if (i == 0 || i == 4 * N) v = nb[i];
    else if (strong) {
      if (i == 2 * N) v = corner;
      else if (i < 2 * N) { int k = 2 * N - 1 - i; v = ((63 - k) * corner + (k + 1) * bl + 32) >> 6; }
      else { int k = i - 2 * N - 1; v = ((63 - k) * corner + (k + 1) * tr + 32) >> 6; }
    } else v = (nb[i - 1] + 2 * nb[i] + nb[i + 1] + 2) >> 2;
    alt[i] = v;
  }
  RBT_SYNC_LDS();
}
RBT_DEV RBT_LDS_AS int32_t* rc_intra_filter(const RbtStreamCfg* g, int c_idx, int log2, int mode, RBT_LDS_AS int32_t* nb, RBT_LDS_AS int32_t* alt) {
  if (!rc_intra_filter_needed(c_idx, log2, mode)) return nb;
  if (!(RBT_ABLATE & 4)) rc_intra_filter_apply(g, log2, nb, alt);
  return alt;
}
// Prediction value of sample (x,y) of the TB from the final reference samples `nb` (planar / DC / angular incl. edge
// filters). Angular modes read l->ref, DC reads `dc`; both are prepared by rc_intra_setup.
struct RcIntraCtx { int N, log2, mode, c_idx, maxv, ang, ver, dc, edge; };
RBT_DEV void rc_intra_setup(const RbtStreamCfg* g, int c_idx, int log2, int mode, RBT_LDS_AS int32_t* nb, RBT_LDS_AS int32_t* ref, RcIntraCtx* q) {
  const int N = 1 << log2, bd = g->bit_depth;
  q->N = N; q->log2 = log2; q->mode = mode; q->c_idx = c_idx; q->maxv = (1 << bd) - 1; q->ang = 0; q->ver = mode >= 18; q->dc = 0; q->edge = 0;
#define RC_LEFT(y) nb[2 * N - 1 - (y)]
#define RC_TOP(x) nb[2 * N + 1 + (x)]
  if (mode == 1) {
    // sum of the 2N neighbours (lanes 0..N-1 hold the top row, N..2N-1 the left column)
    int part = 0;
    if (!(RBT_ABLATE & 8)) RBT_PAR_FOR(p, 2 * N) part += p < N ? RC_TOP(p) : RC_LEFT(p - N);
    q->dc = (N + rbt_wave_sum(part)) >> (log2 + 1); q->edge = c_idx == 0 && N < 32;
  } else if (mode >= 2) {
    const int ang = rc_intra_angle(mode), ver = mode >= 18, last = (N * ang) >> 5;
    const int inv = (mode >= 11 && mode <= 25) ? rc_intra_inv_angle(mode) : 0;
    q->ang = ang; q->edge = c_idx == 0 && N < 32 && (mode == 26 || mode == 10);
    // ref[x], x = -N .. 2N  (stored at index x + 32)
    if (!(RBT_ABLATE & 8)) RBT_PAR_FOR(i, 3 * N + 1) {
      int x = i - N, v = 0;
      if (x >= 0 && x <= N) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
      else if (x < 0) { if (ang < 0 && last < -1 && x >= last) { int k = -1 + ((x * inv + 128) >> 8); v = ver ? RC_LEFT(k) : RC_TOP(k); } }
      else if (ang >= 0) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
      ref[x + 32] = v;
    }
    RBT_SYNC_LDS();
  }
}
// The prediction of sample (x,y), one routine per KIND of mode so that a loop over the samples of a block holds nothing but its own kind's arithmetic (the one routine
// that switched on the mode per sample cost every iteration ~30 scalar instructions of branching around ~20 of work: tools/ablate_run.sh, round 4). `nb`: the final
// reference samples (smoothed or not), `ref`: the angular reference array of rc_intra_setup.
RBT_DEV int rc_pred_planar(const RcIntraCtx* q, const RBT_LDS_AS int32_t* nb, int x, int y) {
  const int N = q->N;
  return ((N - 1 - x) * RC_LEFT(y) + (x + 1) * RC_TOP(N) + (N - 1 - y) * RC_TOP(x) + (y + 1) * RC_LEFT(N) + N) >> (q->log2 + 1);
}
RBT_DEV int rc_pred_dc_edge(const RcIntraCtx* q, const RBT_LDS_AS int32_t* nb, int x, int y) {      // DC of a luma block below 32x32: first row and column smoothed (8.4.4.2.5)
  const int N = q->N; int v = q->dc;
  if (x == 0 && y == 0) v = (RC_LEFT(0) + 2 * q->dc + RC_TOP(0) + 2) >> 2;
  else if (y == 0) v = (RC_TOP(x) + 3 * q->dc + 2) >> 2;
  else if (x == 0) v = (RC_LEFT(y) + 3 * q->dc + 2) >> 2;
  return v;
}
// angular (8.4.4.2.6): VER = mode >= 18. A zero fraction needs no case of its own: (32 * a + 0 * b + 16) >> 5 == a, and the second sample read then lies inside the array
template <int VER> RBT_DEV int rc_pred_angular(const RcIntraCtx* q, const RBT_LDS_AS int32_t* ref, int x, int y) {
  const int a = VER ? y : x, b = VER ? x : y, t = (a + 1) * q->ang, idx = t >> 5, fr = t & 31;
  return ((32 - fr) * ref[32 + b + idx + 1] + fr * ref[32 + b + idx + 2] + 16) >> 5;
}
RBT_DEV int rc_pred_angular_edge(const RcIntraCtx* q, const RBT_LDS_AS int32_t* nb, const RBT_LDS_AS int32_t* ref, int x, int y) {   // pure vertical / horizontal of a luma block below 32x32 (angle 0): first column / row corrected
  const int N = q->N; int v = ref[32 + (q->ver ? x : y) + 1];
  if (q->mode == 26 && x == 0) v = rbt_clip3(0, q->maxv, RC_TOP(0) + ((RC_LEFT(y) - RC_LEFT(-1)) >> 1));
  if (q->mode == 10 && y == 0) v = rbt_clip3(0, q->maxv, RC_LEFT(0) + ((RC_TOP(x) - RC_TOP(-1)) >> 1));
  return v;
}
// RC_INTRA_KINDS(q, nb, ref, BODY): BODY(PV) once per kind, PV = the expression of the prediction of (x,y) - BODY declares x and y (and may declare nb / ref per lane,
// two planes at a time: the names are only expanded inside it). q decides the kind and must be wave-uniform in mode / edge / ver.
#define RC_INTRA_KINDS(q, nb, ref, BODY) do { \
    if ((q)->mode == 0) { BODY(rc_pred_planar((q), (nb), x, y)) } \
    else if ((q)->mode == 1) { if ((q)->edge) { BODY(rc_pred_dc_edge((q), (nb), x, y)) } else { BODY((q)->dc) } } \
    else if ((q)->edge) { BODY(rc_pred_angular_edge((q), (nb), (ref), x, y)) } \
    else if ((q)->ver) { BODY(rc_pred_angular<1>((q), (ref), x, y)) } \
    else { BODY(rc_pred_angular<0>((q), (ref), x, y)) } } while (0)
// two planes at a time (Cb / Cr): `qu` (wave-uniform) decides the kind, `ql` / nb / ref are the lane's own plane's, declared by BODY
#define RC_INTRA_KINDS_PAIR(qu, ql, nb, ref, BODY) do { \
    if ((qu)->mode == 0) { BODY(rc_pred_planar((ql), (nb), x, y)) } \
    else if ((qu)->mode == 1) { BODY((ql)->dc) } \
    else if ((qu)->ver) { BODY(rc_pred_angular<1>((ql), (ref), x, y)) } \
    else { BODY(rc_pred_angular<0>((ql), (ref), x, y)) } } while (0)
#undef RC_LEFT
#undef RC_TOP
// ---- scaling (8.6.3, flat lists) of the TB's levels from the coefficient plane into lds->res ----
template <class CP> RBT_DEV void rc_dequant(CP plane, int pst, int x0, int y0, int log2, int qp, int bd, RBT_LDS_AS RbtReconLdsCore* l) {
  int N = 1 << log2, bd_shift = bd + log2 - 5;
  int scale = (16 * rc_level_scale(qp % 6)) << (qp / 6);
  long long add = 1ll << (bd_shift - 1);
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> log2;
    long long v = ((long long)plane[(size_t)(y0 + y) * pst + x0 + x] * scale + add) >> bd_shift;
    l->res[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
  RBT_SYNC_LDS();
}
// ---- inverse transform of lds->res in place (8.6.4.2) ----
// Specialised per size so that the dot products unroll fully: with a run-time trip count every iteration would wait for
// its own LDS reads (~130 cycles each) instead of having all of them in flight.
template <int LOG2> RBT_DEV void rc_inv_transform_n(int is_dst, int sh, RBT_LDS_AS RbtReconLdsCore* l) {
  constexpr int N = 1 << LOG2;
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> LOG2, s = 0;
#pragma unroll
    for (int k = 0; k < N; k++) s += rc_tcoef(l, N, is_dst, k, y) * l->res[k * N + x];
    l->tmp[i] = (int16_t)rbt_clip3(-32768, 32767, (s + 64) >> 7);
  }
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, N * N) {
    int x = i & (N - 1), y = i >> LOG2, s = 0;
#pragma unroll
    for (int k = 0; k < N; k++) s += rc_tcoef(l, N, is_dst, k, x) * l->tmp[y * N + k];
    l->res[i] = (int16_t)((s + (1 << (sh - 1))) >> sh);
  }
  RBT_SYNC_LDS();
}
// 32 x 32: both stages on the matrix cores (rbt_mfma.h). tmp[y][x] = clip16((sum_k T[k][y] res[k][x] + 64) >> 7), then res[y][x] = (sum_k tmp[y][k] T[k][x] + round) >> sh
RBT_DEV void rc_inv_transform_32(int sh, RBT_LDS_AS RbtReconLdsCore* l) {
#ifdef RBT_HOSTEMU
  rc_inv_transform_n<5>(0, sh, l);
#else
  mf_mm32<true>(l->dct, 1, 32, l->res, l->tmp, 7, 1);
  mf_mm32<false>(l->dct, 1, 32, l->tmp, l->res, sh, 0);
#endif
}
RBT_DEV void rc_inv_transform(int log2, int is_dst, int ts, int bd, RBT_LDS_AS RbtReconLdsCore* l) {
  int N = 1 << log2, sh = 20 - bd;
  if (ts) {
    RBT_PAR_FOR(i, N * N) l->res[i] = (int16_t)((((int)l->res[i] << 7) + (1 << (sh - 1))) >> sh);
    RBT_SYNC_LDS();
    return;
  }
  if (log2 == 2) rc_inv_transform_n<2>(is_dst, sh, l);
  else if (log2 == 3) rc_inv_transform_n<3>(0, sh, l);
  else if (log2 == 4) rc_inv_transform_n<4>(0, sh, l);
  else rc_inv_transform_32(sh, l);
}

// two blocks of the same size at once (Cb and Cr of a TU): block b lives at res / tmp offset b * 256; m0 / m1 = block present
template <int LOG2> RBT_DEV void rc_inv_transform_pair_n(int sh, int m0, int m1, RBT_LDS_AS RbtReconLdsCore* l) {
  constexpr int N = 1 << LOG2, NN = N * N;
  RBT_PAR_FOR(i, 2 * NN) {
    const int b = i >> (2 * LOG2), j = i & (NN - 1), x = j & (N - 1), y = j >> LOG2;
    if (b ? m1 : m0) {
      int s = 0;
#pragma unroll
      for (int k = 0; k < N; k++) s += rc_tcoef(l, N, 0, k, y) * l->res[b * 256 + k * N + x];
      l->tmp[b * 256 + j] = (int16_t)rbt_clip3(-32768, 32767, (s + 64) >> 7);
    }
  }
  RBT_SYNC_LDS();
  RBT_PAR_FOR(i, 2 * NN) {
    const int b = i >> (2 * LOG2), j = i & (NN - 1), x = j & (N - 1), y = j >> LOG2;
    if (b ? m1 : m0) {
      int s = 0;
#pragma unroll
      for (int k = 0; k < N; k++) s += rc_tcoef(l, N, 0, k, x) * l->tmp[b * 256 + y * N + k];
      l->res[b * 256 + j] = (int16_t)((s + (1 << (sh - 1))) >> sh);
    }
  }
  RBT_SYNC_LDS();
}

// ---- uni-directional motion compensation of one PU (8.5.3.3) from ref->out into the CTB tile ----
// Separable, staged through LDS: the PU is walked in sub-blocks of up to 32 x 16 samples; for each one the wave loads the reference
// window (sub-block + filter halo, clamped to the picture: 8.5.3.3.3 reference padding) from HBM once with row-major coalesced loads,
// filters its rows into a 16-bit intermediate (H.265's shift1 keeps it in 16 bits), then its columns straight into the destination,
// completing prediction + residual in the same pass. Per luma sample of a 2-D fractional vector: 8 + 8 * (16 + 7) / 16 multiply-adds
// on LDS operands instead of 72 on clamped HBM loads; every reference sample is fetched once per sub-block instead of up to 64 times.
// The window and the intermediate live in the TB scratch of the calling wave (res / tmp: idle while prediction units are processed).
RBT_DEV int rc_refpix(const uint16_t* p, int w, int h, int x, int y) { return p[(size_t)rbt_clip3(0, h - 1, y) * w + rbt_clip3(0, w - 1, x)]; }
// dst(x,y) = dst[(dy0 + y) * dstride + dx0 + x]: a picture plane in HBM or the CTB tile in LDS
// add_res: dst holds the residual of the block (int16 bit patterns, 0 where nothing is coded); the sample is completed in place
template <int TAPS, class DP> RBT_DEV void rc_mc_plane(DP dst, int dstride, int dx0, int dy0, const uint16_t* ref, int pw, int ph, int x0, int y0, int bw, int bh, int xint, int yint, int xf, int yf,
                         const int8_t* fx, const int8_t* fy, int bd, int add_res, RBT_LDS_AS RbtReconLdsCore* l, int wp_w = 0, int wp_o = 0, int wp_shift = -1) {
  // wp_shift >= 0: explicit weighted sample prediction (8.5.3.3.4.3, (8-265)) of the 14-bit intermediate v: ((v * w + 2^(log2WD - 1)) >> log2WD) + o, log2WD = wp_shift
  constexpr int HALF = TAPS / 2 - 1, SBW = 32, SBH = 16;
  static_assert((SBW + 7) * (SBH + 7) <= 32 * 32 && SBW * (SBH + 7) <= 32 * 32, "window and intermediate fit the TB scratch");
  const int sh1 = rbt_min(4, bd - 8), maxv = (1 << bd) - 1, fsh = 14 - bd, fadd = fsh ? 1 << (fsh - 1) : 0;
  if (!xf && !yf) {                                                         // integer vector: (ref << shift3 + round) >> shift3 == ref
    RBT_PAR_FOR(i, bw * bh) {
      const int y = i / bw, x = i - y * bw, o = (dy0 + y) * dstride + dx0 + x; int pr = rc_refpix(ref, pw, ph, x0 + xint + x, y0 + yint + y);
      if (wp_shift >= 0) { const int v = pr << fsh; pr = rbt_clip3(0, maxv, (wp_shift >= 1 ? ((v * wp_w + (1 << (wp_shift - 1))) >> wp_shift) : v * wp_w) + wp_o); }
      dst[o] = (uint16_t)(add_res ? rbt_clip3(0, maxv, pr + (int16_t)dst[o]) : pr);
    }
    return;
  }
  int cx[TAPS], cy[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; k++) { cx[k] = fx[k]; cy[k] = fy[k]; }
  RBT_LDS_AS uint16_t* const win = (RBT_LDS_AS uint16_t*)l->res; RBT_LDS_AS int16_t* const mid = l->tmp;
  const int hx = xf ? HALF : 0, hy = yf ? HALF : 0, ex = xf ? TAPS - 1 : 0, ey = yf ? TAPS - 1 : 0;
  for (int sy = 0; sy < bh; sy += SBH) for (int sx = 0; sx < bw; sx += SBW) {
    const int sw = rbt_min(SBW, bw - sx), shh = rbt_min(SBH, bh - sy), ww = sw + ex, wh = shh + ey;
    const int gx = x0 + xint + sx - hx, gy = y0 + yint + sy - hy;
    RBT_PAR_FOR(i, ww * wh) { const int r = i / ww, c = i - r * ww; win[i] = (uint16_t)rc_refpix(ref, pw, ph, gx + c, gy + r); }
    RBT_SYNC_LDS();
    RBT_PAR_FOR(i, sw * wh) {                                               // rows
      const int r = i / sw, x = i - r * sw; int v;
      if (xf) { int s = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) s += cx[k] * win[r * ww + x + k];
        v = s >> sh1; }
      else v = win[r * ww + x];
      mid[i] = (int16_t)v;
    }
    RBT_SYNC_LDS();
    RBT_PAR_FOR(i, sw * shh) {                                              // columns, rounding, residual
      const int y = i / sw, x = i - y * sw; int v;
      if (yf) { int s = 0;
#pragma unroll
        for (int j = 0; j < TAPS; j++) s += cy[j] * mid[(y + j) * sw + x];
        v = xf ? s >> 6 : s >> sh1; }
      else v = mid[i];
      const int o = (dy0 + sy + y) * dstride + dx0 + sx + x;
      const int pr = wp_shift < 0 ? rbt_clip3(0, maxv, (v + fadd) >> fsh) : rbt_clip3(0, maxv, (wp_shift >= 1 ? ((v * wp_w + (1 << (wp_shift - 1))) >> wp_shift) : v * wp_w) + wp_o);
      dst[o] = (uint16_t)(add_res ? rbt_clip3(0, maxv, pr + (int16_t)dst[o]) : pr);
    }
    RBT_SYNC_LDS();                                                         // the next sub-block reuses window and intermediate
  }
}
// ---- decoder: one CTB reconstructed inside LDS ----------------------------------------------------------------------
// The TBs of a CTB form a serial chain (intra prediction reads the samples the previous TB wrote). Through HBM every link
// of that chain costs a store round trip plus a load round trip (~3-5 us per TB); here the CTB's samples, its border, the
// availability of every 4x4 unit around it and its coefficient levels are fetched into LDS once, every TB works in LDS,
// and the finished CTB is written back with coalesced row stores.
// The coefficient levels are staged IN the tile: a sample position holds the level of that position until its TB is
// reconstructed (a TB reads the levels of its own area, then overwrites them with samples). Inter blocks take two steps:
// first every inter TB of the CTB turns its levels into the residual in place (inter residuals depend on nothing), then
// the prediction units, in decoding order, complete pred + residual in place. A separate staging area for the levels cost
// 12 KB of LDS per workgroup, i.e. 4 instead of 7 CTBs in flight per CU.
#define RC_TS_Y 65       // body row stride: column -1 (left border) .. n-1; the row above (-1 .. 2n-1, incl. the above-right CTB) is a separate array
#define RC_TS_C 33
#define RC_US 34         // unit availability stride: ux = -1 .. 32
struct alignas(8) RbtU2 { uint32_t x, y; };
struct RbtCtbTile {
  uint16_t y[64 * RC_TS_Y], top_y[130]; uint16_t c[2][32 * RC_TS_C], top_c[2][66];   // sample (xx,yy) relative to the CTB: body yy * stride + xx + 1, row above: top[xx + 1]
};
// The luma TBs of a CTB form one serial chain and its Cb/Cr TBs another; nothing connects the two (prediction reads samples of
// the own component, the intra-reference availability of a 4x4 unit follows the decoding order for both). The workgroup has two
// waves: wave 0 walks the CTB's commands for luma, wave 1 for Cb/Cr, each with its own TB scratch and its own copy of the unit
// availability (both apply the same marks in the same order); they share the tile and never synchronise. The CTB takes the time
// of the longer chain instead of the sum (Cb/Cr were 46 % of a CTB).
enum { RC_ROLE_ALL = 0, RC_ROLE_LUMA = 1, RC_ROLE_CHROMA = 2 };
struct RbtReconRole { RbtReconLdsCore rc; uint8_t uav[17 * RC_US]; };   // uav: 4x4 luma unit (ux,uy) usable as intra reference: (uy + 1) * RC_US + ux + 1
struct RbtReconCtbLds { RbtCtbTile t; RbtReconRole role[2]; };

// availability of unit p of the TB at (x0,y0) (plane samples relative to the CTB) from the CTB's unit flags
RBT_DEV int rc_nb_unit_av(const RBT_LDS_AS uint8_t* uav, int p, int x0, int y0, int N, int sh, int n4) {
  int xn, yn; rc_nb_unit_xy(p, x0, y0, N, sh, &xn, &yn);
  const int ux = (xn << sh) >> 2, uy = (yn << sh) >> 2;                  // -1 for the border column / row
  return uy < n4 && uav[(uy + 1) * RC_US + ux + 1];
}
// mark_l4 >= 0: also flags the TB's (1 << mark_l4)^2 luma units at (mux,muy) as decoded in the same pass
RBT_DEV void rc_tile_tb(const RbtStreamCfg* g, RBT_LDS_AS RbtCtbTile* t, RBT_LDS_AS RbtReconRole* R, int c_idx, int x0, int y0, int log2, int intra, int mode, int cbf, int ts, int tq_bypass, int qp,
                        int mark_l4, int mux, int muy, int mark_flag) {
  // (x0,y0): TB origin relative to the CTB, in samples of component c_idx
  RBT_LDS_AS RbtReconLdsCore* l = &R->rc; RBT_LDS_AS uint8_t* uav = R->uav;
  RBT_LDS_AS int32_t* const l_nbf = (RBT_LDS_AS int32_t*)l->tmp; RBT_LDS_AS int32_t* const l_ref = l_nbf + 132;   // alias tmp (dead after the transform)
  const int N = 1 << log2, sh = c_idx ? 1 : 0, bd = g->bit_depth, maxv = (1 << bd) - 1, n4 = (1 << g->log2_ctb) >> 2;
  RBT_LDS_AS uint16_t* tile = c_idx == 0 ? t->y : t->c[c_idx - 1]; const int S = c_idx == 0 ? RC_TS_Y : RC_TS_C;
  const RBT_LDS_AS uint16_t* top = c_idx == 0 ? t->top_y : t->top_c[c_idx - 1];
  const RBT_LDS_AS int16_t* coef = (const RBT_LDS_AS int16_t*)tile + 1;      // levels of the TB: where its samples will be (row stride S)
  // residual first: it does not depend on the prediction, and the prediction pass can then add it on the fly
  if (cbf && !(RBT_ABLATE & 1)) {
    if (tq_bypass) {
      if (!intra) return;                                                    // inter + bypass: the levels are the residual already
      RBT_PAR_FOR(i, N * N) { int x = i & (N - 1), y = i >> log2; l->res[i] = coef[(y0 + y) * S + x0 + x]; }
      RBT_SYNC_LDS();
    } else {
      rc_dequant(coef, S, x0, y0, log2, qp, bd, l);
      rc_inv_transform(log2, c_idx == 0 && log2 == 2 && intra, ts, bd, l);
    }
  }
  RBT_LDS_AS int32_t* fin = l->nb; RcIntraCtx q;
  if (intra) {
    // availability of the 4N+1 neighbours straight from the unit flags (no LDS round trip), then every lane fetches the
    // sample its index is substituted from (8.4.4.2.2) - gather and substitution in one pass
    const int tot = 4 * N + 1;
    if (!(RBT_ABLATE & 2)) {
      uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(N, sh), rc_nb_unit_av(uav, p, x0, y0, N, sh, n4));
      RcNbMap nm; rc_nb_map(&nm, m, N, sh);
      const RBT_LDS_AS uint16_t* trow = y0 ? tile + (y0 - 1) * S : top;      // the row above the TB, from its corner on: trow[x0 + k]
      RBT_PAR_FOR(i, tot) {
        int v = 1 << (bd - 1);
        if (m) { const int j = rc_nb_source(&nm, i); v = j < 2 * N ? tile[(y0 + 2 * N - 1 - j) * S + x0] : trow[x0 + j - 2 * N]; }
        l->nb[i] = v;
      }
    }
    RBT_SYNC_LDS();
    fin = rc_intra_filter(g, c_idx, log2, mode, l->nb, l_nbf);
    rc_intra_setup(g, c_idx, log2, mode, fin, l_ref, &q);
  }
  if (intra) {
#define RC_BODY(PV) RBT_PAR_FOR(i, N * N) { const int x = i & (N - 1), y = i >> log2, o = (y0 + y) * S + x0 + x + 1, base = (PV); tile[o] = (uint16_t)(cbf ? rbt_clip3(0, maxv, base + l->res[i]) : base); }
    if (!(RBT_ABLATE & 16)) RC_INTRA_KINDS(&q, fin, l_ref, RC_BODY);
#undef RC_BODY
  } else if (cbf) {                                                          // inter: leave the residual where the levels were
    RBT_PAR_FOR(i, N * N) { int x = i & (N - 1), y = i >> log2; tile[(y0 + y) * S + x0 + x + 1] = (uint16_t)l->res[i]; }
  }
  if (mark_l4 >= 0) { RBT_PAR_FOR(i, 1 << (2 * mark_l4)) uav[(muy + (i >> mark_l4) + 1) * RC_US + mux + (i & ((1 << mark_l4) - 1)) + 1] = (uint8_t)mark_flag; }
  RBT_SYNC_LDS();
}
// DC sums (incl. the rounding term N) of two planes whose references sit at nb[0..] and nb[66..]: lanes 0..2N-1 take plane 0 (top row, then left column), 2N..4N-1 plane 1;
// one reduction for both - plane 1 in the high half of the word - while 2N samples fit 16 bits (N <= 16: up to 11-bit video), two otherwise
RBT_DEV void rc_dc_pair(const RBT_LDS_AS int32_t* nb, int N, int bd, int* s0, int* s1) {
  int p0 = 0, p1 = 0;
  if (!(RBT_ABLATE & 8)) RBT_PAR_FOR(p, 4 * N) { const int b = p >= 2 * N, k = p - b * 2 * N, v = nb[b * 66 + (k < N ? 2 * N + 1 + k : 2 * N - 1 - (k - N))]; if (b) p1 += v; else p0 += v; }
  if (bd <= 11) { const uint32_t both = (uint32_t)rbt_wave_sum((int)((uint32_t)p0 + ((uint32_t)p1 << 16))); *s0 = N + (int)(both & 0xFFFF); *s1 = N + (int)(both >> 16); }
  else { *s0 = N + rbt_wave_sum(p0); *s1 = N + rbt_wave_sum(p1); }
}
// Cb and Cr TB of one TU in the same passes. The two blocks share position, size, prediction mode and availability and
// differ only in data, so every phase (and every wait for LDS) is paid once for both; chroma is never smoothed (8.4.4.2.3).
RBT_DEV void rc_tile_tb_cpair(const RbtStreamCfg* g, RBT_LDS_AS RbtCtbTile* t, RBT_LDS_AS RbtReconRole* R, int x0, int y0, int log2, int intra, int mode, int cbf_cb, int cbf_cr, int tq_bypass, int qp_cb, int qp_cr) {
  RBT_LDS_AS RbtReconLdsCore* l = &R->rc; RBT_LDS_AS uint8_t* uav = R->uav;
  RBT_LDS_AS int32_t* const l_ref = (RBT_LDS_AS int32_t*)l->tmp + 132; RBT_LDS_AS int32_t* const l_ref2 = l_ref + 100;   // alias tmp (dead after the transform)
  const int N = 1 << log2, NN = N * N, bd = g->bit_depth, maxv = (1 << bd) - 1, n4 = (1 << g->log2_ctb) >> 2, S = RC_TS_C;
  if ((cbf_cb | cbf_cr) && !(RBT_ABLATE & 1)) {
    if (tq_bypass) {
      if (!intra) return;                                                    // inter + bypass: the levels are the residual already
      RBT_PAR_FOR(i, 2 * NN) { const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2; if (b ? cbf_cr : cbf_cb) l->res[b * 256 + j] = ((const RBT_LDS_AS int16_t*)t->c[b])[(y0 + y) * S + x0 + x + 1]; }
      RBT_SYNC_LDS();
    } else {
      const int bd_shift = bd + log2 - 5, sc_cb = (16 * rc_level_scale(qp_cb % 6)) << (qp_cb / 6), sc_cr = (16 * rc_level_scale(qp_cr % 6)) << (qp_cr / 6);
      const long long add = 1ll << (bd_shift - 1);
      RBT_PAR_FOR(i, 2 * NN) {
        const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2;
        if (b ? cbf_cr : cbf_cb) {
          long long v = ((long long)((const RBT_LDS_AS int16_t*)t->c[b])[(y0 + y) * S + x0 + x + 1] * (b ? sc_cr : sc_cb) + add) >> bd_shift;
          l->res[b * 256 + j] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
      }
      RBT_SYNC_LDS();
      const int sh = 20 - bd;
      if (log2 == 2) rc_inv_transform_pair_n<2>(sh, cbf_cb, cbf_cr, l);
      else if (log2 == 3) rc_inv_transform_pair_n<3>(sh, cbf_cb, cbf_cr, l);
      else rc_inv_transform_pair_n<4>(sh, cbf_cb, cbf_cr, l);
    }
  }
  RcIntraCtx q0, q1;
  if (intra) {
    const int tot = 4 * N + 1;                                           // <= 65: plane b keeps its references at nb[b * 66 ..]
    if (!(RBT_ABLATE & 2)) {
      uint64_t m; RBT_VBALLOT(m, p, rc_nb_units(N, 1), rc_nb_unit_av(uav, p, x0, y0, N, 1, n4));
      RcNbMap nm; rc_nb_map(&nm, m, N, 1);
      RBT_PAR_FOR(i, 2 * tot) {
        const int b = i >= tot, idx = i - b * tot;
        int v = 1 << (bd - 1);
        if (m) {
          const int j = rc_nb_source(&nm, idx);
          const RBT_LDS_AS uint16_t* tile = t->c[b]; const RBT_LDS_AS uint16_t* trow = y0 ? tile + (y0 - 1) * S : t->top_c[b];
          v = j < 2 * N ? tile[(y0 + 2 * N - 1 - j) * S + x0] : trow[x0 + j - 2 * N];
        }
        l->nb[b * 66 + idx] = v;
      }
    }
    RBT_SYNC_LDS();
    // mode set-up of both planes (rc_intra_setup, two at a time)
    q0.N = N; q0.log2 = log2; q0.mode = mode; q0.c_idx = 1; q0.maxv = maxv; q0.ang = 0; q0.ver = mode >= 18; q0.dc = 0; q0.edge = 0; q1 = q0; q1.c_idx = 2;
    if (mode == 1) {
      int s0, s1; rc_dc_pair(l->nb, N, bd, &s0, &s1);
      q0.dc = s0 >> (log2 + 1); q1.dc = s1 >> (log2 + 1);
    } else if (mode >= 2) {
      const int ang = rc_intra_angle(mode), ver = mode >= 18, last = (N * ang) >> 5, inv = (mode >= 11 && mode <= 25) ? rc_intra_inv_angle(mode) : 0;
      q0.ang = q1.ang = ang;
      if (!(RBT_ABLATE & 8)) RBT_PAR_FOR(i, 2 * (3 * N + 1)) {
        const int b = i >= 3 * N + 1, x = i - b * (3 * N + 1) - N; int v = 0;
        const RBT_LDS_AS int32_t* nb = l->nb + b * 66;
#define RC_LEFT(y) nb[2 * N - 1 - (y)]
#define RC_TOP(x) nb[2 * N + 1 + (x)]
        if (x >= 0 && x <= N) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
        else if (x < 0) { if (ang < 0 && last < -1 && x >= last) { int k = -1 + ((x * inv + 128) >> 8); v = ver ? RC_LEFT(k) : RC_TOP(k); } }
        else if (ang >= 0) v = ver ? RC_TOP(x - 1) : RC_LEFT(x - 1);
#undef RC_LEFT
#undef RC_TOP
        (b ? l_ref2 : l_ref)[x + 32] = v;
      }
      RBT_SYNC_LDS();
    }
  }
  if (intra && !(RBT_ABLATE & 16)) {
    // lane i: sample j of plane b; plane 1 keeps its references 66 and its angular array 100 entries behind plane 0's; the kind is the same for both
#define RC_BODY(PV) RBT_PAR_FOR(i, 2 * NN) { \
      const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2, o = (y0 + y) * S + x0 + x + 1, cbf = b ? cbf_cr : cbf_cb; \
      const RcIntraCtx* qp = b ? &q1 : &q0; const RBT_LDS_AS int32_t* nbp = l->nb + b * 66; const RBT_LDS_AS int32_t* refp = l_ref + b * 100; \
      const int base = (PV); t->c[b][o] = (uint16_t)(cbf ? rbt_clip3(0, maxv, base + l->res[b * 256 + j]) : base); }
    RC_INTRA_KINDS_PAIR(&q0, qp, nbp, refp, RC_BODY);
#undef RC_BODY
  } else if (!intra && (cbf_cb || cbf_cr)) {
    RBT_PAR_FOR(i, 2 * NN) {
      const int b = i >= NN, j = i - b * NN, x = j & (N - 1), y = j >> log2, o = (y0 + y) * S + x0 + x + 1;
      if (b ? cbf_cr : cbf_cb) t->c[b][o] = (uint16_t)l->res[b * 256 + j];   // inter: leave the residual where the levels were
    }
  }
  RBT_SYNC_LDS();
}
RBT_DEV void rc_tile_mark(RBT_LDS_AS uint8_t* uav, int ux, int uy, int w4, int h4, int flag) {   // any rectangle (prediction units)
  RBT_PAR_FOR(i, w4 * h4) uav[(uy + i / w4 + 1) * RC_US + ux + i % w4 + 1] = (uint8_t)flag;
  RBT_SYNC_LDS();
}
RBT_DEV void rc_tile_mark_sq(RBT_LDS_AS uint8_t* uav, int ux, int uy, int l4, int flag) {           // square of 1 << l4 units (transform units)
  RBT_PAR_FOR(i, 1 << (2 * l4)) uav[(uy + (i >> l4) + 1) * RC_US + ux + (i & ((1 << l4) - 1)) + 1] = (uint8_t)flag;
  RBT_SYNC_LDS();
}
// usable-as-intra-reference flag of the 4x4 unit (gxu,gyu) of the picture for CTB ac (a unit outside the current CTB)
RBT_DEV int rc_unit_avail(const RbtFrame* f, int ac, int gxu, int gyu) {
  const RbtStreamCfg* g = &f->cfg;
  if (gxu < 0 || gyu < 0 || gxu >= g->w4 || gyu >= g->h4) return 0;
  int L = g->log2_ctb, an = ((gyu << 2) >> L) * g->w_ctb + ((gxu << 2) >> L);
  if (an >= ac || f->ctb_slice[an] != f->ctb_slice[ac]) return 0;
  if (g->cip && (f->pm[gyu * g->w4 + gxu] & RBT_PM_MODE_MASK) != RBT_MODE_INTRA) return 0;
  return 1;
}
// The picture's stream parameters as wave-uniform words (same reason as rc_cmd_uni below: what a vector load delivered would keep every size, shift and bound derived
// from it in vector registers)
RBT_DEV RbtStreamCfg rc_cfg_uni(const RbtStreamCfg* p) {
  static_assert(sizeof(RbtStreamCfg) % 4 == 0, "whole words");
  int32_t w[sizeof(RbtStreamCfg) / 4]; __builtin_memcpy(w, p, sizeof w);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(RbtStreamCfg) / 4); i++) w[i] = RBT_UNI(w[i]);
  RbtStreamCfg c; __builtin_memcpy(&c, w, sizeof c);
  return c;
}
// A command as wave-uniform words: every lane of the wave reads the same one, but what arrives through a vector load sits in vector registers and everything derived
// from it - block size, position, flags, the loops' trip counts, every branch - would be computed per lane and branched on as if lanes could disagree.
// (Applied where the command is USED: the load of the next command stays in flight while the current one is processed.)
struct alignas(16) RbtCmdRaw { int32_t w[4]; };
RBT_DEV RbtCmdRaw rc_cmd_load(const RbtCmd* p) { return *(const RbtCmdRaw*)p; }
RBT_DEV RbtCmd rc_cmd_uni(const RbtCmdRaw& raw) {
  union { RbtCmdRaw r; RbtCmd c; } u;
  u.r.w[0] = RBT_UNI(raw.w[0]); u.r.w[1] = RBT_UNI(raw.w[1]); u.r.w[2] = RBT_UNI(raw.w[2]); u.r.w[3] = RBT_UNI(raw.w[3]);
  return u.c;
}
// ROLE: RC_ROLE_LUMA / RC_ROLE_CHROMA = the calling wave's half of the CTB (see RbtReconRole); RC_ROLE_ALL = everything on one wave.
template <int ROLE>
RBT_DEV void rbt_recon_ctb(RbtFrame* frames, const RbtSlice* slices, int frame_idx, int ctb_addr, RBT_LDS_AS RbtCtbTile* t, RBT_LDS_AS RbtReconRole* R) {
  constexpr bool DO_Y = ROLE != RC_ROLE_CHROMA, DO_C = ROLE != RC_ROLE_LUMA;
  frame_idx = RBT_UNI(frame_idx); ctb_addr = RBT_UNI(ctb_addr);
  RbtFrame* f = &frames[frame_idx];
  const RbtStreamCfg gcopy = rc_cfg_uni(&f->cfg);                        // private copy: not reloaded after every store, and in scalar registers
  const RbtStreamCfg* g = &gcopy;
  const int ctb = 1 << g->log2_ctb, n4 = ctb >> 2, cx = (ctb_addr % g->w_ctb) << g->log2_ctb, cy = (ctb_addr / g->w_ctb) << g->log2_ctb;
  uint32_t n = (uint32_t)RBT_UNI(f->cmd_count[ctb_addr]);
  if ((int)n > f->cmd_cap) n = (uint32_t)f->cmd_cap;
  const RbtCmd* cmds = f->cmds + (size_t)ctb_addr * f->cmd_cap;
  const RbtSlice* sl = &slices[f->ctb_slice[ctb_addr]];
  rc_stage_tables(&R->rc);
  // ---- fetch: borders, unit availability, coefficient levels (one HBM round trip for everything) ----
  if (!(RBT_ABLATE & 32)) for (int c = DO_Y ? 0 : 1; c < (DO_C ? 3 : 1); c++) {
    const int sh = c ? 1 : 0, nn = ctb >> sh, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, ox = cx >> sh, oy = cy >> sh, S = c ? RC_TS_C : RC_TS_Y;
    const uint16_t* p = f->pix[c]; RBT_LDS_AS uint16_t* tile = c == 0 ? t->y : t->c[c - 1]; RBT_LDS_AS uint16_t* top = c == 0 ? t->top_y : t->top_c[c - 1];
    RBT_PAR_FOR(i, 2 * nn + 1) { int x = ox + i - 1, y = oy - 1; top[i] = (x >= 0 && y >= 0 && x < pw) ? p[(size_t)y * pw + x] : 0; }
    RBT_PAR_FOR(i, nn) { int x = ox - 1, y = oy + i; tile[i * S] = (x >= 0 && y < ph) ? p[(size_t)y * pw + x] : 0; }
    // coefficient levels: 4 per lane and load (8-byte aligned: widths are multiples of 8, chroma of 4), several loads in flight
    const RbtU2* cp = (const RbtU2*)(f->coef[c] + (size_t)oy * pw + ox);
    const int q4 = nn >> 2, lq4 = g->log2_ctb - sh - 2, rows = rbt_min(nn, ph - oy), cols4 = rbt_min(nn, pw - ox) >> 2;
#pragma unroll 4
    RBT_PAR_FOR(i, nn * q4) {
      const int x4 = i & (q4 - 1), y = i >> lq4;
      RbtU2 v; v.x = 0; v.y = 0;
      if (x4 < cols4 && y < rows) v = cp[((size_t)y * pw >> 2) + x4];
      RBT_LDS_AS uint16_t* d = tile + y * S + 4 * x4 + 1;                   // into the body of the tile (rows are not 8-byte aligned there)
      d[0] = (uint16_t)v.x; d[1] = (uint16_t)(v.x >> 16); d[2] = (uint16_t)v.y; d[3] = (uint16_t)(v.y >> 16);
    }
  }
  RBT_PAR_FOR(i, 17 * RC_US) {
    int ux = i % RC_US - 1, uy = i / RC_US - 1, a = 0;
    if ((uy < 0 && ux < 2 * n4) || (ux < 0 && uy < n4)) a = rc_unit_avail(f, ctb_addr, (cx >> 2) + ux, (cy >> 2) + uy);
    R->uav[i] = (uint8_t)a;
  }
  RBT_SYNC_LDS();                                                         // everything a wave reads below it fetched itself
  // ---- inter TBs first (P slices): levels -> residual in place ----
  const int has_inter = sl->slice_type != RBT_SLICE_I;
  if (has_inter && !(RBT_ABLATE & 128)) {
    RbtCmdRaw nx0; if (n) nx0 = rc_cmd_load(&cmds[0]);
    for (uint32_t k = 0; k < n; k++) {
      const RbtCmd c = rc_cmd_uni(nx0);
      if (k + 1 < n) nx0 = rc_cmd_load(&cmds[k + 1]);
      if (c.type != RBT_CMD_TU || (c.a & RBT_TU_INTRA)) continue;
      const int x0 = c.x4 * 4, y0 = c.y4 * 4, fl = c.a, log2 = c.log2;
      if (DO_Y && (fl & RBT_TU_CBF_Y)) rc_tile_tb(g, t, R, 0, x0, y0, log2, 0, c.b, 1, (fl & RBT_TU_TS_Y) != 0, c.d, c.qp[0], -1, 0, 0, 0);
      if (DO_C && (fl & RBT_TU_CHROMA) && (fl & (RBT_TU_CBF_CB | RBT_TU_CBF_CR))) {
        const int xc = (log2 > 2 ? x0 : x0 - 4) >> 1, yc = (log2 > 2 ? y0 : y0 - 4) >> 1, l2c = log2 > 2 ? log2 - 1 : 2;
        if (fl & (RBT_TU_TS_CB | RBT_TU_TS_CR)) {
          if (fl & RBT_TU_CBF_CB) rc_tile_tb(g, t, R, 1, xc, yc, l2c, 0, c.c, 1, (fl & RBT_TU_TS_CB) != 0, c.d, c.qp[1], -1, 0, 0, 0);
          if (fl & RBT_TU_CBF_CR) rc_tile_tb(g, t, R, 2, xc, yc, l2c, 0, c.c, 1, (fl & RBT_TU_TS_CR) != 0, c.d, c.qp[2], -1, 0, 0, 0);
        } else rc_tile_tb_cpair(g, t, R, xc, yc, l2c, 0, c.c, (fl & RBT_TU_CBF_CB) != 0, (fl & RBT_TU_CBF_CR) != 0, c.d, c.qp[1], c.qp[2]);
      }
    }
    RBT_SYNC_LDS();
  }
  // ---- the CTB's commands, in decoding order: prediction units complete pred + residual, intra TBs predict and add ----
  RbtCmdRaw nxt; if (n) nxt = rc_cmd_load(&cmds[0]);
  for (uint32_t k = 0; k < n; k++) {
    const RbtCmd c = rc_cmd_uni(nxt);
    if (k + 1 < n) nxt = rc_cmd_load(&cmds[k + 1]);                       // fetched while command k is processed
    const int x0 = c.x4 * 4, y0 = c.y4 * 4;                               // relative to the CTB
    if (c.type == RBT_CMD_PU) {
      if (RBT_ABLATE & 64) continue;
      const RbtFrame* ref = &frames[sl->ref_frame[c.c]];
      const int w = c.a * 4, h = c.b * 4, mvx = c.mvx, mvy = c.mvy;
      const int wpn = sl->wp_on, ri = c.c;                                 // explicit weights of this reference index (slices of a PPS with weighted_pred_flag)
      if (DO_Y) rc_mc_plane<8>(t->y, RC_TS_Y, x0 + 1, y0, ref->out[0], g->w, g->h, cx + x0, cy + y0, w, h, mvx >> 2, mvy >> 2, mvx & 3, mvy & 3, k_luma_filter[mvx & 3], k_luma_filter[mvy & 3], g->bit_depth, 1, &R->rc,
                               wpn ? sl->wp_w[ri][0] : 0, wpn ? sl->wp_o[ri][0] : 0, wpn ? sl->wp_shift[0] : -1);
      if (DO_C) for (int cc = 1; cc < 3; cc++)
        rc_mc_plane<4>(t->c[cc - 1], RC_TS_C, (x0 >> 1) + 1, y0 >> 1, ref->out[cc], g->cw, g->ch, (cx + x0) >> 1, (cy + y0) >> 1, w >> 1, h >> 1, mvx >> 3, mvy >> 3, mvx & 7, mvy & 7,
                       k_chroma_filter[mvx & 7], k_chroma_filter[mvy & 7], g->bit_depth, 1, &R->rc, wpn ? sl->wp_w[ri][cc] : 0, wpn ? sl->wp_o[ri][cc] : 0, wpn ? sl->wp_shift[1] : -1);
      rc_tile_mark(R->uav, c.x4, c.y4, c.a, c.b, !g->cip);
    } else if (c.type == RBT_CMD_TU && (c.a & RBT_TU_INTRA)) {
      const int fl = c.a, log2 = c.log2;
      if (DO_Y) rc_tile_tb(g, t, R, 0, x0, y0, log2, 1, c.b, fl & RBT_TU_CBF_Y, (fl & RBT_TU_TS_Y) != 0, c.d, c.qp[0], log2 - 2, c.x4, c.y4, 1);
      else rc_tile_mark_sq(R->uav, c.x4, c.y4, log2 - 2, 1);             // the mark the luma TB leaves behind
      if (DO_C && (fl & RBT_TU_CHROMA)) {
        const int xc = (log2 > 2 ? x0 : x0 - 4) >> 1, yc = (log2 > 2 ? y0 : y0 - 4) >> 1, l2c = log2 > 2 ? log2 - 1 : 2;
        if (fl & (RBT_TU_TS_CB | RBT_TU_TS_CR)) {                      // transform skip (rare): one plane at a time
          rc_tile_tb(g, t, R, 1, xc, yc, l2c, 1, c.c, fl & RBT_TU_CBF_CB, (fl & RBT_TU_TS_CB) != 0, c.d, c.qp[1], -1, 0, 0, 0);
          rc_tile_tb(g, t, R, 2, xc, yc, l2c, 1, c.c, fl & RBT_TU_CBF_CR, (fl & RBT_TU_TS_CR) != 0, c.d, c.qp[2], -1, 0, 0, 0);
        } else rc_tile_tb_cpair(g, t, R, xc, yc, l2c, 1, c.c, (fl & RBT_TU_CBF_CB) != 0, (fl & RBT_TU_CBF_CR) != 0, c.d, c.qp[1], c.qp[2]);
      }
    }
  }
  // ---- write the CTB back (clipped to the picture) ----
  for (int c = DO_Y ? 0 : 1; c < (DO_C ? 3 : 1); c++) {
    const int sh = c ? 1 : 0, nn = ctb >> sh, pw = c ? g->cw : g->w, ph = c ? g->ch : g->h, ox = cx >> sh, oy = cy >> sh, S = c ? RC_TS_C : RC_TS_Y;
    uint16_t* p = f->pix[c]; RBT_LDS_AS uint16_t* tile = c == 0 ? t->y : t->c[c - 1];
    const int lnn = g->log2_ctb - sh;
    RBT_PAR_FOR(i, nn * nn) { int x = i & (nn - 1), y = i >> lnn; if (ox + x < pw && oy + y < ph) p[(size_t)(oy + y) * pw + ox + x] = RBT_ABLATE ? (uint16_t)(tile[y * S + x + 1] & ((1 << g->bit_depth) - 1)) : tile[y * S + x + 1]; }
  }
}

// ---- the dependency graph of a picture's CTBs inside one level (k_recon_queue; the same rule k_recon_level waits by): CTB (x, y) needs its left neighbour and its
// above-right neighbour - which needed the one above, which needed the one above-left - or, in the last column, the one above. rc_ctb_need: how many of the two exist;
// rc_ctb_successors: the CTBs that count (x, y) among theirs (at most three: right; below-left; below when x is the last column), returns how many.
RBT_DEV int rc_ctb_need(int x, int y) { return (x > 0) + (y > 0); }
RBT_DEV int rc_ctb_successors(int w, int h, int x, int y, int* succ) {
  int n = 0;
  if (x + 1 < w) succ[n++] = y * w + x + 1;
  if (y + 1 < h) {
    if (x >= 1) succ[n++] = (y + 1) * w + x - 1;
    if (x == w - 1) succ[n++] = (y + 1) * w + x;
  }
  return n;
}
