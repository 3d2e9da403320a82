// CABAC engines (H.265 9.3.4.3 decode, 9.3.4.x encode) for wave-uniform execution: one wave runs one slice segment,
// every lane executes the same scalar sequence, lane 0 owns the global stores.
//
// The engine state lives in registers, not memory: the 157 context variables are spread over the 64 lanes of three
// VGPRs (context i = lane i & 63 of register i >> 6) and are read / updated with v_readlane / a lane-select; rangeTabLPS
// and the LPS state transition table sit in two more VGPRs (lane = pStateIdx). Everything derived from them is
// wave-uniform, so range / offset arithmetic runs on the scalar unit and a bin costs no LDS or global-memory access.
// The bitstream is consumed through aligned 32-bit words (scalar loads).
#pragma once
#include "rbt_tables.h"

enum {
  CTX_SAO_MERGE = 0, CTX_SAO_TYPE = 1, CTX_SPLIT_CU = 2, CTX_CU_TQ_BYPASS = 5, CTX_CU_SKIP = 6, CTX_PRED_MODE = 9,
  CTX_PART_MODE = 10, CTX_PREV_INTRA_LUMA = 14, CTX_INTRA_CHROMA = 15, CTX_RQT_ROOT_CBF = 16, CTX_MERGE_FLAG = 17,
  CTX_MERGE_IDX = 18, CTX_INTER_PRED_IDC = 19, CTX_REF_IDX = 24, CTX_MVP_FLAG = 26, CTX_SPLIT_TRANSFORM = 27,
  CTX_CBF_LUMA = 30, CTX_CBF_CHROMA = 32, CTX_MVD_GT0 = 37, CTX_MVD_GT1 = 38, CTX_CU_QP_DELTA = 39,
  CTX_TRANSFORM_SKIP = 41, CTX_LAST_X = 43, CTX_LAST_Y = 61, CTX_CSBF = 79, CTX_SIG = 83, CTX_GT1 = 127, CTX_GT2 = 151
};

// ------------------------------------------------------------------------------------------------ context store
struct RbtCtxStore {
#ifdef RBT_HOSTEMU
  uint8_t st[RBT_CTX_COUNT + 3];
#else
  int st0, st1, st2;     // lane-distributed context variables: pStateIdx << 1 | valMps
  int lps_tab;           // lane s: rangeTabLPS[s][0..3] packed little-endian
  int nxt_tab;           // lane s: transIdxLps[s]
#endif
};
RBT_DEV int rbt_ctx_initval(int init_type, int qp, int i) {
  int iv = k_ctx_init[init_type][i];
  int m = (iv >> 4) * 5 - 45, n = ((iv & 15) << 3) - 16;
  int pre = rbt_clip3(1, 126, ((m * qp) >> 4) + n);
  int mps = pre <= 63 ? 0 : 1;
  return ((mps ? pre - 64 : 63 - pre) << 1) | mps;
}
RBT_DEV void rbt_ctx_init(RbtCtxStore* s, int init_type, int qp) {
  qp = rbt_clip3(0, 51, qp);
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) s->st[i] = (uint8_t)rbt_ctx_initval(init_type, qp, i);
#else
  int lane = (int)threadIdx.x & 63;
  s->st0 = rbt_ctx_initval(init_type, qp, lane);
  s->st1 = rbt_ctx_initval(init_type, qp, lane + 64);
  s->st2 = lane + 128 < RBT_CTX_COUNT ? rbt_ctx_initval(init_type, qp, lane + 128) : 0;
  s->lps_tab = (int)(k_range_lps[lane][0] | (k_range_lps[lane][1] << 8) | (k_range_lps[lane][2] << 16) | ((uint32_t)k_range_lps[lane][3] << 24));
  s->nxt_tab = k_next_lps[lane];
#endif
}
RBT_DEV int rbt_ctx_get(const RbtCtxStore* s, int ctx) {
#ifdef RBT_HOSTEMU
  return s->st[ctx];
#else
  int lane = ctx & 63, r = ctx >> 6;
  if (r == 0) return __builtin_amdgcn_readlane(s->st0, lane);
  if (r == 1) return __builtin_amdgcn_readlane(s->st1, lane);
  return __builtin_amdgcn_readlane(s->st2, lane);
#endif
}
RBT_DEV void rbt_ctx_set(RbtCtxStore* s, int ctx, int v) {
#ifdef RBT_HOSTEMU
  s->st[ctx] = (uint8_t)v;
#else
  // v_writelane has no builtin in this toolchain: a per-lane select (v_cmp + v_cndmask) does the same job
  int me = (int)threadIdx.x & 63, lane = ctx & 63, r = ctx >> 6;
  if (r == 0) s->st0 = me == lane ? v : s->st0;
  else if (r == 1) s->st1 = me == lane ? v : s->st1;
  else s->st2 = me == lane ? v : s->st2;
#endif
}
RBT_DEV int rbt_lps(const RbtCtxStore* s, int state, int q) {
#ifdef RBT_HOSTEMU
  (void)s; return k_range_lps[state][q];
#else
  return (int)(((uint32_t)__builtin_amdgcn_readlane(s->lps_tab, state) >> (8 * q)) & 255u);
#endif
}
RBT_DEV int rbt_next_lps(const RbtCtxStore* s, int state) {
#ifdef RBT_HOSTEMU
  (void)s; return k_next_lps[state];
#else
  return __builtin_amdgcn_readlane(s->nxt_tab, state);
#endif
}

// ------------------------------------------------------------------------------------------------ decoder
struct RbtCabacDec {
  const uint32_t* w; uint32_t n_words, widx;   // aligned word cursor over the slice data
  uint32_t next_raw;                           // word widx, loaded one refill ahead so its latency is hidden (raw, per-lane copy)
  uint64_t buf; int nbuf;                      // bit reservoir (MSB first)
  uint32_t range, offset;
  uint32_t bits_total, bits_read;
#ifdef RBT_PROFILE
  uint32_t n_bins, n_byp;
#endif
  RbtCtxStore cs;
};
RBT_DEV uint32_t rbt_cd_bits(RbtCabacDec* c, int n) {
  if (n == 0) return 0;
  if (c->nbuf < n) {
    uint32_t v = (uint32_t)RBT_UNI(__builtin_bswap32(c->next_raw));
    c->widx++;
    c->next_raw = c->widx < c->n_words ? c->w[c->widx] : 0;
    c->buf = (c->buf << 32) | v; c->nbuf += 32;
  }
  uint32_t r = (uint32_t)(c->buf >> (c->nbuf - n)) & ((1u << n) - 1u);
  c->nbuf -= n; c->bits_read += (uint32_t)n;
  return r;
}
// p .. p+size is the slice data; the allocation is padded so that the aligned words covering it can be read
RBT_DEV void rbt_cd_start(RbtCabacDec* c, const uint8_t* p, uint32_t size) {
  uintptr_t a = (uintptr_t)p; int mis = (int)(a & 3);
  c->w = (const uint32_t*)(a - (uintptr_t)mis); c->n_words = (size + (uint32_t)mis + 3) >> 2; c->widx = 0;
  c->buf = 0; c->nbuf = 0; c->bits_total = size * 8; c->bits_read = 0;
  c->next_raw = c->n_words ? c->w[0] : 0;
  if (mis) { (void)rbt_cd_bits(c, 8 * mis); c->bits_read = 0; }
  c->range = 510; c->offset = rbt_cd_bits(c, 9);
}
RBT_DEV int rbt_cd_bin(RbtCabacDec* c, int ctx) {
#ifdef RBT_PROFILE
  c->n_bins++;
#endif
  int st = rbt_ctx_get(&c->cs, ctx);
  int s = st >> 1, mps = st & 1, bin;
  uint32_t lps = (uint32_t)rbt_lps(&c->cs, s, (int)((c->range >> 6) & 3));
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = !mps; c->offset -= c->range; c->range = lps;
    if (s == 0) mps = 1 - mps;
    s = rbt_next_lps(&c->cs, s);
  } else { bin = mps; s = s >= 62 ? s : s + 1; }
  rbt_ctx_set(&c->cs, ctx, (s << 1) | mps);
  if (c->range < 256) {
    int sh = __builtin_clz(c->range) - 23;
    c->range <<= sh; c->offset = (c->offset << sh) | rbt_cd_bits(c, sh);
  }
  return bin;
}
RBT_DEV int rbt_cd_bypass(RbtCabacDec* c) {
#ifdef RBT_PROFILE
  c->n_byp++;
#endif
  c->offset = (c->offset << 1) | rbt_cd_bits(c, 1);
  if (c->offset >= c->range) { c->offset -= c->range; return 1; }
  return 0;
}
RBT_DEV uint32_t rbt_cd_bypass_n(RbtCabacDec* c, int n) { uint32_t v = 0; while (n--) v = (v << 1) | (uint32_t)rbt_cd_bypass(c); return v; }
RBT_DEV int rbt_cd_terminate(RbtCabacDec* c) {
  c->range -= 2;
  if (c->offset >= c->range) return 1;
  if (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | rbt_cd_bits(c, 1); }
  return 0;
}
// Copies the engine into a function-local object whose scalar fields are marked wave-uniform: the local lives in SGPRs
// (scalar ALU, scalar branches) for the duration of a hot loop, independent of where the enclosing parser state sits.
RBT_DEV void rbt_cd_localise(RbtCabacDec* d, const RbtCabacDec* c) {
  d->next_raw = c->next_raw;
  d->w = rbt_uni_ptr(c->w); d->n_words = (uint32_t)RBT_UNI(c->n_words); d->widx = (uint32_t)RBT_UNI(c->widx);
  d->buf = ((uint64_t)(uint32_t)RBT_UNI((uint32_t)(c->buf >> 32)) << 32) | (uint32_t)RBT_UNI((uint32_t)c->buf);
  d->nbuf = RBT_UNI(c->nbuf); d->range = (uint32_t)RBT_UNI(c->range); d->offset = (uint32_t)RBT_UNI(c->offset);
  d->bits_total = (uint32_t)RBT_UNI(c->bits_total); d->bits_read = (uint32_t)RBT_UNI(c->bits_read);
  d->cs = c->cs;
#ifdef RBT_PROFILE
  d->n_bins = c->n_bins; d->n_byp = c->n_byp;
#endif
}
RBT_DEV int rbt_cd_overrun(const RbtCabacDec* c) { return c->bits_read > c->bits_total + 64; }

// ------------------------------------------------------------------------------------------------ encoder
struct RbtCabacEnc {
  uint8_t* out; uint32_t cap, n;          // byte output (lane 0 stores)
  uint32_t acc; int nacc;                 // bit accumulator
  uint32_t low, range; int outstanding, first;
  RbtCtxStore cs;
  int overflow;
};
RBT_DEV void rbt_ce_write_bit(RbtCabacEnc* c, int b) {
  c->acc = (c->acc << 1) | (uint32_t)(b & 1);
  if (++c->nacc == 8) {
    if (c->n < c->cap) { if (RBT_LANE0) c->out[c->n] = (uint8_t)c->acc; } else c->overflow = 1;
    c->n++; c->acc = 0; c->nacc = 0;
  }
}
RBT_DEV void rbt_ce_write_bits(RbtCabacEnc* c, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) rbt_ce_write_bit(c, (int)((v >> i) & 1)); }
RBT_DEV void rbt_ce_put(RbtCabacEnc* c, int b) {
  if (c->first) c->first = 0; else rbt_ce_write_bit(c, b);
  while (c->outstanding > 0) { rbt_ce_write_bit(c, 1 - b); c->outstanding--; }
}
RBT_DEV void rbt_ce_renorm(RbtCabacEnc* c) {
  while (c->range < 256) {
    if (c->low < 256) rbt_ce_put(c, 0);
    else if (c->low >= 512) { c->low -= 512; rbt_ce_put(c, 1); }
    else { c->low -= 256; c->outstanding++; }
    c->range <<= 1; c->low <<= 1;
  }
}
RBT_DEV void rbt_ce_start(RbtCabacEnc* c) { c->low = 0; c->range = 510; c->first = 1; c->outstanding = 0; }
RBT_DEV void rbt_ce_bin(RbtCabacEnc* c, int ctx, int bin) {
  int st = rbt_ctx_get(&c->cs, ctx);
  int s = st >> 1, mps = st & 1;
  uint32_t lps = (uint32_t)rbt_lps(&c->cs, s, (int)((c->range >> 6) & 3));
  c->range -= lps;
  if (bin != mps) { c->low += c->range; c->range = lps; if (s == 0) mps = 1 - mps; s = rbt_next_lps(&c->cs, s); }
  else s = s >= 62 ? s : s + 1;
  rbt_ctx_set(&c->cs, ctx, (s << 1) | mps);
  rbt_ce_renorm(c);
}
RBT_DEV void rbt_ce_bypass(RbtCabacEnc* c, int bin) {
  c->low <<= 1;
  if (bin) c->low += c->range;
  if (c->low >= 1024) { rbt_ce_put(c, 1); c->low -= 1024; }
  else if (c->low < 512) rbt_ce_put(c, 0);
  else { c->low -= 512; c->outstanding++; }
}
RBT_DEV void rbt_ce_bypass_n(RbtCabacEnc* c, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) rbt_ce_bypass(c, (int)((v >> i) & 1)); }
RBT_DEV void rbt_ce_terminate(RbtCabacEnc* c, int bin) {
  c->range -= 2;
  if (bin) {
    c->low += c->range; c->range = 2; rbt_ce_renorm(c);
    rbt_ce_put(c, (int)((c->low >> 9) & 1)); rbt_ce_write_bits(c, ((c->low >> 7) & 3) | 1, 2);
  } else rbt_ce_renorm(c);
}
RBT_DEV void rbt_ce_align_zero(RbtCabacEnc* c) { while (c->nacc) rbt_ce_write_bit(c, 0); }
