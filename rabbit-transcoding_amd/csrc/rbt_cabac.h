// CABAC engines (H.265 9.3.4.3 decode, 9.3.4.x encode) for wave-uniform execution: one wave runs one slice segment,
// every lane executes the same scalar sequence, lane 0 owns the global stores.
#pragma once
#include "rbt_tables.h"

enum {
  CTX_SAO_MERGE = 0, CTX_SAO_TYPE = 1, CTX_SPLIT_CU = 2, CTX_CU_TQ_BYPASS = 5, CTX_CU_SKIP = 6, CTX_PRED_MODE = 9,
  CTX_PART_MODE = 10, CTX_PREV_INTRA_LUMA = 14, CTX_INTRA_CHROMA = 15, CTX_RQT_ROOT_CBF = 16, CTX_MERGE_FLAG = 17,
  CTX_MERGE_IDX = 18, CTX_INTER_PRED_IDC = 19, CTX_REF_IDX = 24, CTX_MVP_FLAG = 26, CTX_SPLIT_TRANSFORM = 27,
  CTX_CBF_LUMA = 30, CTX_CBF_CHROMA = 32, CTX_MVD_GT0 = 37, CTX_MVD_GT1 = 38, CTX_CU_QP_DELTA = 39,
  CTX_TRANSFORM_SKIP = 41, CTX_LAST_X = 43, CTX_LAST_Y = 61, CTX_CSBF = 79, CTX_SIG = 83, CTX_GT1 = 127, CTX_GT2 = 151
};

RBT_DEV void rbt_ctx_init(RBT_LDS_AS uint8_t* st, int init_type, int qp) {
  qp = rbt_clip3(0, 51, qp);
  RBT_PAR_FOR(i, RBT_CTX_COUNT) {
    int iv = k_ctx_init[init_type][i];
    int m = (iv >> 4) * 5 - 45, n = ((iv & 15) << 3) - 16;
    int pre = rbt_clip3(1, 126, ((m * qp) >> 4) + n);
    int mps = pre <= 63 ? 0 : 1;
    st[i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
  }
}

// ------------------------------------------------------------------------------------------------ decoder
struct RbtCabacDec {
  const uint8_t* p; uint32_t size, pos;   // byte cursor of the next refill
  uint64_t buf; int nbuf;                 // bit reservoir (MSB first)
  uint32_t range, offset;
  RBT_LDS_AS uint8_t* st;
};
RBT_DEV uint32_t rbt_cd_bits(RbtCabacDec* c, int n) {
  if (n == 0) return 0;
  if (c->nbuf < n) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) { uint32_t b = c->pos < c->size ? c->p[c->pos] : 0; c->pos++; v = (v << 8) | b; }
    c->buf = (c->buf << 32) | v; c->nbuf += 32;
  }
  uint32_t r = (uint32_t)(c->buf >> (c->nbuf - n)) & ((1u << n) - 1u);
  c->nbuf -= n;
  return r;
}
RBT_DEV void rbt_cd_start(RbtCabacDec* c, const uint8_t* p, uint32_t size, RBT_LDS_AS uint8_t* st) {
  c->p = p; c->size = size; c->pos = 0; c->buf = 0; c->nbuf = 0; c->st = st; c->range = 510; c->offset = rbt_cd_bits(c, 9);
}
RBT_DEV int rbt_cd_bin(RbtCabacDec* c, int ctx) {
  int s = c->st[ctx] >> 1, mps = c->st[ctx] & 1, bin;
  uint32_t lps = k_range_lps[s][(c->range >> 6) & 3];
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = !mps; c->offset -= c->range; c->range = lps;
    if (s == 0) mps = 1 - mps;
    s = k_next_lps[s];
  } else { bin = mps; s = s >= 62 ? s : s + 1; }
  c->st[ctx] = (uint8_t)((s << 1) | mps);
  if (c->range < 256) {
    int sh = 0; uint32_t r = c->range; while (r < 256) { r <<= 1; sh++; }
    c->range = r; c->offset = (c->offset << sh) | rbt_cd_bits(c, sh);
  }
  return bin;
}
RBT_DEV int rbt_cd_bypass(RbtCabacDec* c) {
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
RBT_DEV uint32_t rbt_cd_bytes_consumed(const RbtCabacDec* c) { return c->pos - (uint32_t)(c->nbuf >> 3); }

// ------------------------------------------------------------------------------------------------ encoder
struct RbtCabacEnc {
  uint8_t* out; uint32_t cap, n;          // byte output (lane 0 stores)
  uint32_t acc; int nacc;                 // bit accumulator
  uint32_t low, range; int outstanding, first;
  RBT_LDS_AS uint8_t* st;
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
RBT_DEV void rbt_ce_start(RbtCabacEnc* c, RBT_LDS_AS uint8_t* st) { c->low = 0; c->range = 510; c->first = 1; c->outstanding = 0; c->st = st; }
RBT_DEV void rbt_ce_bin(RbtCabacEnc* c, int ctx, int bin) {
  int s = c->st[ctx] >> 1, mps = c->st[ctx] & 1;
  uint32_t lps = k_range_lps[s][(c->range >> 6) & 3];
  c->range -= lps;
  if (bin != mps) { c->low += c->range; c->range = lps; if (s == 0) mps = 1 - mps; s = k_next_lps[s]; }
  else s = s >= 62 ? s : s + 1;
  c->st[ctx] = (uint8_t)((s << 1) | mps);
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
