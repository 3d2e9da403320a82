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
// Register file layout of the 157 context variables (value = pStateIdx << 1 | valMps), one VGPR per syntax class so that
// a call site addresses its register statically:
//   reg 0  lanes 0..42   everything outside residual_coding (ctx 0..42)
//   reg 1  lanes 0..43   sig_coeff_flag                      (CTX_SIG + i)
//   reg 2  lanes 0..3    coded_sub_block_flag, 4..27 greater1, 28..33 greater2
//   reg 3  lanes 0..17   last_sig_coeff_x_prefix, 18..35 last_sig_coeff_y_prefix
struct RbtCtxStore {
#ifdef RBT_HOSTEMU
  uint8_t st[RBT_CTX_COUNT + 3];
#else
  int st0, st1, st2, st3;
  int lps_tab;           // lane s: rangeTabLPS[s][0..3] packed little-endian
  int nxt_tab;           // lane s: transIdxLps[s]
  int trans_tab;         // lane s: rbt_trans_word(s)
#endif
};
RBT_DEV int rbt_ctx_reg(int ctx) { return ctx < CTX_LAST_X ? 0 : (ctx < CTX_CSBF ? 3 : (ctx < CTX_SIG ? 2 : (ctx < CTX_GT1 ? 1 : 2))); }
RBT_DEV int rbt_ctx_lane(int ctx) { return ctx < CTX_LAST_X ? ctx : (ctx < CTX_CSBF ? ctx - CTX_LAST_X : (ctx < CTX_SIG ? ctx - CTX_CSBF : (ctx < CTX_GT1 ? ctx - CTX_SIG : ctx - CTX_GT1 + 4))); }
RBT_DEV int rbt_ctx_initval(int init_type, int qp, int i) {
  int iv = k_ctx_init[init_type][i];
  int m = (iv >> 4) * 5 - 45, n = ((iv & 15) << 3) - 16;
  int pre = rbt_clip3(1, 126, ((m * qp) >> 4) + n);
  int mps = pre <= 63 ? 0 : 1;
  return ((mps ? pre - 64 : 63 - pre) << 1) | mps;
}
// transition word of pStateIdx s: bits 0..7 = (transIdxLps << 1 | (s == 0)) ^ 1, bits 8..15 = (transIdxMps << 1) ^ 1; the
// selected byte XOR !valMps is the updated context variable
RBT_DEV int rbt_trans_word(int s) { s &= 63; int nm = s < 62 ? s + 1 : s; return (((k_next_lps[s] << 1) | (s == 0)) ^ 1) | (((nm << 1) ^ 1) << 8); }
// the constant lookup registers (rangeTabLPS, transitions); separate so that a resumed parser can rebuild them
RBT_DEV void rbt_ctx_tables(RbtCtxStore* s) {
#ifdef RBT_HOSTEMU
  (void)s;
#else
  int l = (int)threadIdx.x & 63;
  s->lps_tab = (int)(k_range_lps[l][0] | (k_range_lps[l][1] << 8) | (k_range_lps[l][2] << 16) | ((uint32_t)k_range_lps[l][3] << 24));
  s->nxt_tab = k_next_lps[l];
  s->trans_tab = rbt_trans_word(l);
#endif
}
RBT_DEV void rbt_ctx_init(RbtCtxStore* s, int init_type, int qp) {
  qp = rbt_clip3(0, 51, qp);
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) s->st[i] = (uint8_t)rbt_ctx_initval(init_type, qp, i);
#else
  int l = (int)threadIdx.x & 63;
  s->st0 = l < CTX_LAST_X ? rbt_ctx_initval(init_type, qp, l) : 0;
  s->st1 = l < 44 ? rbt_ctx_initval(init_type, qp, CTX_SIG + l) : 0;
  s->st2 = l < 4 ? rbt_ctx_initval(init_type, qp, CTX_CSBF + l) : (l < 34 ? rbt_ctx_initval(init_type, qp, CTX_GT1 + l - 4) : 0);
  s->st3 = l < 36 ? rbt_ctx_initval(init_type, qp, CTX_LAST_X + l) : 0;
  rbt_ctx_tables(s);
#endif
}
// generic access by context index (any syntax class)
RBT_DEV int rbt_ctx_get(const RbtCtxStore* s, int ctx) {
#ifdef RBT_HOSTEMU
  return s->st[ctx];
#else
  int r = rbt_ctx_reg(ctx), lane = rbt_ctx_lane(ctx);
  int v0 = __builtin_amdgcn_readlane(s->st0, lane), v1 = __builtin_amdgcn_readlane(s->st1, lane), v2 = __builtin_amdgcn_readlane(s->st2, lane), v3 = __builtin_amdgcn_readlane(s->st3, lane);
  return r == 0 ? v0 : (r == 1 ? v1 : (r == 2 ? v2 : v3));
#endif
}
RBT_DEV void rbt_ctx_set(RbtCtxStore* s, int ctx, int v) {
#ifdef RBT_HOSTEMU
  s->st[ctx] = (uint8_t)v;
#else
  // v_writelane has no builtin in this toolchain: a per-lane select (v_cmp + v_cndmask) does the same job
  int me = (int)threadIdx.x & 63, r = rbt_ctx_reg(ctx), lane = rbt_ctx_lane(ctx);
  s->st0 = (r == 0 && me == lane) ? v : s->st0; s->st1 = (r == 1 && me == lane) ? v : s->st1;
  s->st2 = (r == 2 && me == lane) ? v : s->st2; s->st3 = (r == 3 && me == lane) ? v : s->st3;
#endif
}
// copies of all context variables in LDS (wavefront storage / synchronisation processes, 9.3.2.2 / 9.3.2.4): 4 x 64 bytes, register r lane l at r * 64 + l
RBT_DEV void rbt_ctx_store(const RbtCtxStore* s, RBT_LDS_AS uint8_t* dst) {
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) dst[i] = s->st[i];
#else
  const int l = (int)threadIdx.x & 63;
  dst[l] = (uint8_t)s->st0; dst[64 + l] = (uint8_t)s->st1; dst[128 + l] = (uint8_t)s->st2; dst[192 + l] = (uint8_t)s->st3;
#endif
}
RBT_DEV void rbt_ctx_load(RbtCtxStore* s, const RBT_LDS_AS uint8_t* src) {
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) s->st[i] = src[i];
#else
  const int l = (int)threadIdx.x & 63;
  s->st0 = src[l]; s->st1 = src[64 + l]; s->st2 = src[128 + l]; s->st3 = src[192 + l];
#endif
}
// the same copies in global memory (the entropy coder hands them from the wave of one CTB row to the wave of the next)
RBT_DEV void rbt_ctx_store_g(const RbtCtxStore* s, uint8_t* dst) {
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) dst[i] = s->st[i];
#else
  const int l = (int)threadIdx.x & 63;
  dst[l] = (uint8_t)s->st0; dst[64 + l] = (uint8_t)s->st1; dst[128 + l] = (uint8_t)s->st2; dst[192 + l] = (uint8_t)s->st3;
#endif
}
RBT_DEV void rbt_ctx_load_g(RbtCtxStore* s, const uint8_t* src) {
#ifdef RBT_HOSTEMU
  for (int i = 0; i < RBT_CTX_COUNT; i++) s->st[i] = src[i];
#else
  const int l = (int)threadIdx.x & 63;
  s->st0 = src[l]; s->st1 = src[64 + l]; s->st2 = src[128 + l]; s->st3 = src[192 + l];
#endif
}
RBT_DEV int rbt_lps(const RbtCtxStore* s, int state, int q) {
#ifdef RBT_HOSTEMU
  (void)s; return k_range_lps[state][q];
#else
  return (int)(((uint32_t)__builtin_amdgcn_readlane(s->lps_tab, state) >> (8 * q)) & 255u);
#endif
}
RBT_DEV int rbt_trans(const RbtCtxStore* s, int state) {
#ifdef RBT_HOSTEMU
  (void)s; return rbt_trans_word(state);
#else
  return __builtin_amdgcn_readlane(s->trans_tab, state);
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
// The arithmetic decoder keeps ivlOffset pre-scaled: `value` = ivlOffset << 22 with the next `avail` (>= 7 between bins)
// bits of the slice data already sitting below it, so renormalisation is a plain left shift and the bitstream is touched
// once per 16 bits instead of once per bin. Bit 31 is head-room for the doubling in a bypass bin.
struct RbtCabacDec {
  const RBT_CONST_AS uint32_t* w; uint32_t n_words, widx;   // aligned word cursor over the slice data (scalar loads)
  uint32_t next_raw;                           // word widx, loaded one refill ahead so its latency is hidden
  uint64_t buf; int nbuf;                      // bit reservoir (MSB first)
  uint32_t range, value; int avail;
  uint32_t bits_total;                         // bits of the aligned words that belong to the slice data
#ifdef RBT_PROFILE
  uint32_t n_bins, n_byp;
#endif
  RbtCtxStore cs;
};
enum { RBT_CD_SCALE = 22 };
// The engine state is wave-uniform by construction. Outside the hot loops the compiler cannot always prove it (the state
// passes through joins it considers divergent), and one unproven value turns the whole syntax parser into exec-masked
// vector code; re-asserting uniformity here is free where it is already known (the readfirstlane folds away).
RBT_DEV void rbt_cd_assert_uniform(RbtCabacDec* c) {
  c->range = (uint32_t)RBT_UNI(c->range); c->value = (uint32_t)RBT_UNI(c->value); c->avail = RBT_UNI(c->avail); c->nbuf = RBT_UNI(c->nbuf);
}
RBT_DEV uint32_t rbt_cd_bits(RbtCabacDec* c, int n) {
  if (c->nbuf < n) {                                   // once per 32 bits
    uint32_t v = (uint32_t)RBT_UNI(__builtin_bswap32(c->next_raw));
    c->widx = (uint32_t)RBT_UNI(c->widx) + 1;
    c->buf = ((uint64_t)(uint32_t)RBT_UNI((uint32_t)(c->buf >> 32)) << 32) | (uint32_t)RBT_UNI((uint32_t)c->buf);
    { const RBT_CONST_AS uint32_t* w = (const RBT_CONST_AS uint32_t*)rbt_uni_ptr((const uint32_t*)(uintptr_t)c->w);   // uniform base: scalar load
      c->next_raw = c->widx < (uint32_t)RBT_UNI(c->n_words) ? w[c->widx] : 0; }
    c->buf = (c->buf << 32) | v; c->nbuf += 32;
  }
  c->nbuf -= n;
  return (uint32_t)(c->buf >> c->nbuf) & ((1u << n) - 1u);   // n == 0 yields 0
}
RBT_DEV void rbt_cd_refill(RbtCabacDec* c) {          // avail in 0..6 -> 16..22
  c->value |= rbt_cd_bits(c, 16) << (6 - c->avail);
  c->avail += 16;
}
// p .. p+size is the slice data; the allocation is padded so that the aligned words covering it can be read
RBT_DEV void rbt_cd_start(RbtCabacDec* c, const uint8_t* p, uint32_t size) {
  uintptr_t a = (uintptr_t)p; int mis = (int)(a & 3);
  c->w = (const RBT_CONST_AS uint32_t*)(a - (uintptr_t)mis); c->n_words = (size + (uint32_t)mis + 3) >> 2; c->widx = 0;
  c->buf = 0; c->nbuf = 0; c->bits_total = (size + (uint32_t)mis) * 8;
  c->next_raw = c->n_words ? c->w[0] : 0;
  if (mis) (void)rbt_cd_bits(c, 8 * mis);
  c->range = 510;
  c->value = rbt_cd_bits(c, 9) << RBT_CD_SCALE; c->avail = 0;
  rbt_cd_refill(c);
}
// After a terminating bin equal to 1 that does not end the slice segment (end_of_subset_one_bit): the engine has consumed exactly the bits up to and
// including the one that ended the codeword (9 at start, one per renormalisation shift = everything pulled from the stream minus `avail`); the next
// codeword starts at the next byte boundary.
RBT_DEV void rbt_cd_restart_aligned(RbtCabacDec* c) {
  const uint32_t used = (uint32_t)RBT_UNI(c->widx) * 32u - (uint32_t)RBT_UNI(c->nbuf) - (uint32_t)RBT_UNI(c->avail);   // bits from the aligned base
  const uint32_t byte = (used + 7u) >> 3, total = (uint32_t)RBT_UNI(c->bits_total) >> 3;
  const uint8_t* base = (const uint8_t*)rbt_uni_ptr((const uint32_t*)(uintptr_t)c->w);
  rbt_cd_start(c, base + byte, total > byte ? total - byte : 0u);
}
// Decodes one bin given the context variable value `st` (pStateIdx << 1 | valMps); stores the updated variable to *nst.
template <bool AU> RBT_DEV int rbt_cd_core(RbtCabacDec* c, int st, int* nst) {
#ifdef RBT_PROFILE
  c->n_bins++;
#endif
  if (AU) rbt_cd_assert_uniform(c);
  // Pure integer arithmetic, no boolean temporaries: the compiler materialises a wave-uniform bool through the vector unit
  // (v_cndmask + v_readfirstlane), which costs more than the decision itself. value and rms are below 2^31, so the sign
  // bit of their difference is the MPS/LPS decision.
  const uint32_t s = (uint32_t)st >> 1, nmps = ~(uint32_t)st & 1u;
  const uint32_t lps = (uint32_t)rbt_lps(&c->cs, (int)s, (int)((c->range >> 6) & 3));
  const uint32_t tr = (uint32_t)rbt_trans(&c->cs, (int)s);
  const uint32_t rm = c->range - lps, rms = rm << RBT_CD_SCALE;
  const uint32_t d = c->value - rms, mf = d >> 31;     // mf = 1 on the MPS path
  const uint32_t value = d < c->value ? d : c->value;
  const uint32_t range = mf ? rm : lps;
  *nst = (int)(((tr >> (mf << 3)) & 255u) ^ nmps);
  const int sh = __builtin_clz(range) - 23;            // 0 when range >= 256, at most 6
  c->range = range << sh;
  c->value = value << sh;
  c->avail -= sh;
  if (__builtin_expect(c->avail < 7, 0)) rbt_cd_refill(c);
  return (int)(nmps ^ mf);
}
// contexts outside residual_coding (ctx < CTX_LAST_X): all of them live in context register 0
RBT_DEV int rbt_cd_bin(RbtCabacDec* c, int ctx) {
#ifdef RBT_HOSTEMU
  int nst, b = rbt_cd_core<true>(c, rbt_ctx_get(&c->cs, ctx), &nst);
  rbt_ctx_set(&c->cs, ctx, nst);
#else
  ctx = RBT_UNI(ctx);
  int nst, b = rbt_cd_core<true>(c, __builtin_amdgcn_readlane(c->cs.st0, ctx), &nst);
  c->cs.st0 = rbt_writelane(c->cs.st0, nst, ctx);
#endif
  return b;
}
// class-specific entry points: the register holding the context is known at the call site
#ifdef RBT_HOSTEMU
#define RBT_CD_BIN_REG(NAME, REG, BASE) RBT_DEV int NAME(RbtCabacDec* c, int lane) { return rbt_cd_bin(c, (BASE) + lane); }
#else
#define RBT_CD_BIN_REG(NAME, REG, BASE) RBT_DEV int NAME(RbtCabacDec* c, int lane) { \
  int nst, b = rbt_cd_core<false>(c, __builtin_amdgcn_readlane(c->cs.REG, lane), &nst); \
  c->cs.REG = rbt_writelane(c->cs.REG, nst, lane); \
  return b; }
#endif
RBT_CD_BIN_REG(rbt_cd_bin_sig, st1, CTX_SIG)          // lane = sigCtx (0..43)
RBT_CD_BIN_REG(rbt_cd_bin_res2, st2, (lane < 4 ? CTX_CSBF : CTX_GT1 - 4))   // lane = 0..3 csbf, 4..27 greater1, 28..33 greater2
RBT_DEV int rbt_cd_bin_csbf(RbtCabacDec* c, int i) { return rbt_cd_bin_res2(c, i); }
RBT_DEV int rbt_cd_bin_gt1(RbtCabacDec* c, int i) { return rbt_cd_bin_res2(c, 4 + i); }
RBT_DEV int rbt_cd_bin_gt2(RbtCabacDec* c, int i) { return rbt_cd_bin_res2(c, 28 + i); }
RBT_CD_BIN_REG(rbt_cd_bin_last, st3, CTX_LAST_X)      // lane = 0..17 x prefix, 18..35 y prefix
template <bool AU = true> RBT_DEV int rbt_cd_bypass(RbtCabacDec* c) {
#ifdef RBT_PROFILE
  c->n_byp++;
#endif
  if (AU) rbt_cd_assert_uniform(c);
  const uint32_t v = c->value << 1, d = v - (c->range << RBT_CD_SCALE);   // v < 2 * rs and rs < 2^31: bit 31 of d = (v < rs)
  c->value = d < v ? d : v;
  if (__builtin_expect(--c->avail < 7, 0)) rbt_cd_refill(c);
  return (int)((d >> 31) ^ 1u);
}
template <bool AU = true> RBT_DEV uint32_t rbt_cd_bypass_n(RbtCabacDec* c, int n) {
  if (AU) { rbt_cd_assert_uniform(c); n = RBT_UNI(n); }
  uint32_t r = 0; const uint32_t rs = c->range << RBT_CD_SCALE, inv = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
  while (n > 0) {                                      // up to 7 bins between refill checks
    const int k = n < 7 ? n : 7;
    for (int i = 0; i < k; i++) { const uint32_t v = c->value << 1, d = v - rs; c->value = d < v ? d : v; r = (r << 1) | (d >> 31); }
#ifdef RBT_PROFILE
    c->n_byp += (uint32_t)k;
#endif
    c->avail -= k; n -= k;
    if (__builtin_expect(c->avail < 7, 0)) rbt_cd_refill(c);
  }
  return r ^ inv;                                      // the loop collected the complemented bins
}
RBT_DEV int rbt_cd_terminate(RbtCabacDec* c) {
  rbt_cd_assert_uniform(c);
  c->range -= 2;
  if (c->value >= (c->range << RBT_CD_SCALE)) return 1;
  if (c->range < 256) { c->range <<= 1; c->value <<= 1; if (__builtin_expect(--c->avail < 7, 0)) rbt_cd_refill(c); }
  return 0;
}
// Copies the engine into a function-local object whose scalar fields are marked wave-uniform: the local lives in SGPRs
// (scalar ALU, scalar branches) for the duration of a hot loop, independent of where the enclosing parser state sits.
RBT_DEV void rbt_cd_localise(RbtCabacDec* d, const RbtCabacDec* c) {
  d->next_raw = c->next_raw;
  d->w = (const RBT_CONST_AS uint32_t*)rbt_uni_ptr((const uint32_t*)(uintptr_t)c->w); d->n_words = (uint32_t)RBT_UNI(c->n_words); d->widx = (uint32_t)RBT_UNI(c->widx);
  d->buf = ((uint64_t)(uint32_t)RBT_UNI((uint32_t)(c->buf >> 32)) << 32) | (uint32_t)RBT_UNI((uint32_t)c->buf);
  d->nbuf = RBT_UNI(c->nbuf); d->range = (uint32_t)RBT_UNI(c->range); d->value = (uint32_t)RBT_UNI(c->value); d->avail = RBT_UNI(c->avail);
  d->bits_total = (uint32_t)RBT_UNI(c->bits_total);
  d->cs = c->cs;
#ifdef RBT_PROFILE
  d->n_bins = c->n_bins; d->n_byp = c->n_byp;
#endif
}
RBT_DEV int rbt_cd_overrun(const RbtCabacDec* c) { return c->widx * 32u - (uint32_t)c->nbuf > c->bits_total + 96u + (uint32_t)c->avail; }

// ------------------------------------------------------------------------------------------------ encoder
// Byte-oriented arithmetic encoder (the formulation of HM's TEncBinCABAC: `low` carries up to 23 pending bits, whole bytes
// leave through a one-byte buffer plus a count of 0xFF bytes so that a late carry can still ripple). It produces the same
// bits as the bit-serial PutBit / outstanding-bits procedure of 9.3.4.5, which the oracle implements.
struct RbtCabacEnc {
  uint8_t* out; uint32_t cap, n;          // byte output (lane 0 stores)
  uint32_t low, range; int bits_left, n_buffered, buffered;
  RbtCtxStore cs;
  int overflow;
};
RBT_DEV void rbt_ce_put_byte(RbtCabacEnc* c, int b) {
  if (c->n < c->cap) { if (RBT_LANE0) c->out[c->n] = (uint8_t)b; } else c->overflow = 1;
  c->n++;
}
RBT_DEV void rbt_ce_write_out(RbtCabacEnc* c) {
  const int lead = (int)(c->low >> (24 - c->bits_left));
  c->bits_left += 8;
  c->low &= 0xFFFFFFFFu >> c->bits_left;
  if (lead == 0xFF) c->n_buffered++;
  else if (c->n_buffered > 0) {
    const int carry = lead >> 8;
    rbt_ce_put_byte(c, c->buffered + carry);
    c->buffered = lead & 0xFF;
    const int fill = (0xFF + carry) & 0xFF;
    while (c->n_buffered > 1) { rbt_ce_put_byte(c, fill); c->n_buffered--; }
  } else { c->n_buffered = 1; c->buffered = lead; }
}
RBT_DEV void rbt_ce_start(RbtCabacEnc* c) { c->low = 0; c->range = 510; c->bits_left = 23; c->n_buffered = 0; c->buffered = 0xFF; }
// encodes `bin` with context variable value st; returns the updated variable
RBT_DEV int rbt_ce_core(RbtCabacEnc* c, int st, int bin) {
  c->low = (uint32_t)RBT_UNI(c->low); c->range = (uint32_t)RBT_UNI(c->range); c->bits_left = RBT_UNI(c->bits_left);
  const uint32_t s = (uint32_t)st >> 1, nmps = ~(uint32_t)st & 1u;
  const uint32_t lps = (uint32_t)rbt_lps(&c->cs, (int)s, (int)((c->range >> 6) & 3));
  const uint32_t tr = (uint32_t)rbt_trans(&c->cs, (int)s);
  const uint32_t rm = c->range - lps, mf = ((uint32_t)bin ^ nmps) & 1u;      // mf = 1: bin is the MPS
  const uint32_t range = mf ? rm : lps;
  const int sh = __builtin_clz(range) - 23;                                  // 0 or 1 on the MPS path, up to 6 on the LPS path
  c->low = (c->low + (mf ? 0u : rm)) << sh;
  c->range = range << sh;
  c->bits_left -= sh;
  if (__builtin_expect(c->bits_left < 12, 0)) rbt_ce_write_out(c);
  return (int)(((tr >> (mf << 3)) & 255u) ^ nmps);
}
RBT_DEV void rbt_ce_bin(RbtCabacEnc* c, int ctx, int bin) {     // any context (generic register select)
  rbt_ctx_set(&c->cs, ctx, rbt_ce_core(c, rbt_ctx_get(&c->cs, ctx), bin));
}
#ifdef RBT_HOSTEMU
#define RBT_CE_BIN_REG(NAME, REG, BASE) RBT_DEV void NAME(RbtCabacEnc* c, int lane, int bin) { rbt_ce_bin(c, (BASE) + lane, bin); }
#else
#define RBT_CE_BIN_REG(NAME, REG, BASE) RBT_DEV void NAME(RbtCabacEnc* c, int lane, int bin) { \
  lane = RBT_UNI(lane); \
  const int nst = rbt_ce_core(c, __builtin_amdgcn_readlane(c->cs.REG, lane), bin); \
  c->cs.REG = rbt_writelane(c->cs.REG, nst, lane); }
#endif
RBT_CE_BIN_REG(rbt_ce_bin0, st0, 0)                                          // contexts below CTX_LAST_X
RBT_CE_BIN_REG(rbt_ce_bin_sig, st1, CTX_SIG)
RBT_CE_BIN_REG(rbt_ce_bin_res2, st2, (lane < 4 ? CTX_CSBF : CTX_GT1 - 4))
RBT_CE_BIN_REG(rbt_ce_bin_last, st3, CTX_LAST_X)
RBT_DEV void rbt_ce_bypass_n(RbtCabacEnc* c, uint32_t v, int n) {           // n bypass bins, MSB of the n-bit value first
  c->low = (uint32_t)RBT_UNI(c->low); c->range = (uint32_t)RBT_UNI(c->range); c->bits_left = RBT_UNI(c->bits_left);
  while (n > 8) {
    n -= 8;
    const uint32_t pat = v >> n;
    c->low = (c->low << 8) + c->range * pat; v -= pat << n;
    c->bits_left -= 8;
    if (c->bits_left < 12) rbt_ce_write_out(c);
  }
  c->low = (c->low << n) + c->range * v;
  c->bits_left -= n;
  if (c->bits_left < 12) rbt_ce_write_out(c);
}
RBT_DEV void rbt_ce_bypass(RbtCabacEnc* c, int bin) { rbt_ce_bypass_n(c, (uint32_t)(bin & 1), 1); }
RBT_DEV void rbt_ce_finish(RbtCabacEnc* c) {
  if (c->low >> (32 - c->bits_left)) {
    rbt_ce_put_byte(c, c->buffered + 1);
    while (c->n_buffered > 1) { rbt_ce_put_byte(c, 0x00); c->n_buffered--; }
    c->low -= 1u << (32 - c->bits_left);
  } else {
    if (c->n_buffered > 0) rbt_ce_put_byte(c, c->buffered);
    while (c->n_buffered > 1) { rbt_ce_put_byte(c, 0xFF); c->n_buffered--; }
  }
  // the remaining 24 - bits_left bits of low >> 8, then rbsp_stop_one_bit and zero bits up to the byte boundary
  int nb = 24 - c->bits_left; uint32_t bits = (c->low >> 8) & ((1u << nb) - 1u);
  bits = (bits << 1) | 1u; nb++;
  while (nb & 7) { bits <<= 1; nb++; }
  for (int k = nb - 8; k >= 0; k -= 8) rbt_ce_put_byte(c, (int)((bits >> k) & 0xFF));
}
// end_of_slice_segment_flag; bin = 1 also flushes the encoder and writes the trailing bits (9.3.2.5, 7.3.2.5)
RBT_DEV void rbt_ce_terminate(RbtCabacEnc* c, int bin) {
  c->range -= 2;
  if (bin) { c->low = (c->low + c->range) << 7; c->range = 2 << 7; c->bits_left -= 7; }
  else if (c->range >= 256) return;
  else { c->low <<= 1; c->range <<= 1; c->bits_left--; }
  if (c->bits_left < 12) rbt_ce_write_out(c);
  if (bin) rbt_ce_finish(c);
}
RBT_DEV void rbt_ce_align_zero(RbtCabacEnc* c) { (void)c; }     // rbt_ce_terminate(c, 1) already ends on a byte boundary
