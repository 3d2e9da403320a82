/* ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
 *
 * CPU restatement of the HEVC decode the reference performs through libavcodec's `hevc` decoder
 * (PCCTranscoder.cpp:428-448: av_read_frame / avcodec_send_packet / avcodec_receive_frame). libavcodec is a system
 * dependency absent from /root/reference, so this follows ITU-T H.265 (04/2013 tools) for the toolset
 * the V-PCC CTC streams use (cfg/hm/ctc-hm-geometry-ai.cfg, ctc-hm-attribute-ai.cfg, ctc-hm-occupancy-map-ai-main10.cfg):
 * Main/Main10 4:2:0, I and P slices, CTB 16..64, TU 4..32, transform skip, cu_transquant_bypass, AMP, merge/AMVP,
 * TMVP, sign data hiding, cu_qp_delta, deblocking, SAO, decoded-picture-hash SEI (MD5); dependent slice segments (7.3.6.1)
 * and wavefront streams (entropy_coding_sync_enabled_flag: what libx265 writes by default, and RBT-E1's wavefront mode) - context
 * variables of the CTB above-right at the start of a CTB row (9.3.1), end_of_subset_one_bit and a new arithmetic codeword per row
 * (entry point offsets are read and not needed by a decoder that reads the rows one after the other).
 * Not supported (rejected with an error): B slices, tiles, PCM, scaling lists, long-term reference pictures. Weighted prediction of P slices (what libx265 writes from its
 * preset "veryfast" up: the reference's own output as an input) is decoded since round 4 (pred_weight_table 7.3.6.3, 8.5.3.3.4.3).
 * PARITY: unpinned against libavcodec (not available here). Self-checks: MD5 SEI, encoder-recon == decoder output.
 */
#include <limits.h>
#include "hevc_dec.h"

#define DEC_ERR(...) do { fprintf(stderr, "[oracle hevc_dec] " __VA_ARGS__); fprintf(stderr, "\n"); } while (0)


struct oracle_hevc_decoder {
  hevc_sps sps[16];
  hevc_pps pps[64];
  hevc_frame** out; hevc_colinfo* col; int n_out, cap_out;
  hevc_meta* meta; hevc_frame* cur; int cur_idx; int cur_poc; int pic_open;
  int prev_tid0_poc;
  int md5_checked, md5_failed;
  int last_conf_win[4];
  int slice_idx_in_pic;
  /* slice-level state that dependent slice segments (7.3.6.1) and wavefront rows (9.3.1) inherit */
  hevc_slice_hdr last_sh; int last_slice_idx, have_last_sh;   /* header and slice index of the slice's independent segment */
  uint8_t ds_ctx[CTX_COUNT]; int ds_qp_y;                     /* context variables / QpY at the end of the previous slice segment (TableStateIdxDs) */
  uint8_t wpp_ctx[CTX_COUNT];                                 /* context variables after the second CTB of the CTB row above (TableStateIdxWpp) */
  uint8_t pending_md5[3][16]; int have_md5;
  /* scan tables: [scanIdx 0 diag,1 hor,2 ver][log2 1..3][pos] -> x | y<<4 */
  uint8_t scan[3][4][64];
  int tables_ready;
  int error;
};

/* ================================================================================================ scans (6.5.3-6.5.5) */
static void build_scans(oracle_hevc_decoder* d) {
  for (int l = 0; l <= 3; l++) {
    int n = 1 << l, i = 0, x = 0, y = 0, stop = 0;
    while (!stop) {
      while (y >= 0) { if (x < n && y < n) d->scan[0][l][i++] = (uint8_t)(x | (y << 4)); y--; x++; }
      y = x; x = 0; if (i >= n * n) stop = 1;
    }
    i = 0; for (y = 0; y < n; y++) for (x = 0; x < n; x++) d->scan[1][l][i++] = (uint8_t)(x | (y << 4));
    i = 0; for (x = 0; x < n; x++) for (y = 0; y < n; y++) d->scan[2][l][i++] = (uint8_t)(x | (y << 4));
  }
  d->tables_ready = 1;
}

/* ================================================================================================ NAL / RBSP */
static size_t find_start(const uint8_t* p, size_t n, size_t from, int* sc_len) {
  for (size_t i = from; i + 3 <= n; i++)
    if (p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 1) { *sc_len = 3; return i; }
  return n;
}
static void unescape(const uint8_t* p, size_t n, bytebuf* out) {
  out->n = 0; int z = 0;
  for (size_t i = 0; i < n; i++) {
    if (z >= 2 && p[i] == 3) { z = 0; continue; }
    z = p[i] == 0 ? z + 1 : 0;
    bb_put(out, p[i]);
  }
}

/* ================================================================================================ parameter sets */
static void skip_ptl(bitreader* b, int max_sub_layers_minus1) {
  br_u(b, 8); br_u(b, 32); br_u(b, 4); br_u(b, 32); br_u(b, 11); br_u(b, 1); /* general: 88 bits */
  br_u(b, 8);
  int pp[8], lp[8];
  for (int i = 0; i < max_sub_layers_minus1; i++) { pp[i] = br_bit(b); lp[i] = br_bit(b); }
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) br_u(b, 2);
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    if (pp[i]) { br_u(b, 32); br_u(b, 32); br_u(b, 24); }
    if (lp[i]) br_u(b, 8);
  }
}
static int parse_st_rps(bitreader* b, hevc_sps* s, int idx, int in_slice_header) {
  int inter = idx ? br_bit(b) : 0;
  if (inter) {
    int delta_idx = 1;
    if (in_slice_header) delta_idx = (int)br_ue(b) + 1;
    int ref = idx - delta_idx;
    if (ref < 0) return -1;
    int sign = br_bit(b); int absd = (int)br_ue(b) + 1; int drps = (1 - 2 * sign) * absd;
    int nref = s->st_rps[ref].num, nneg = s->st_rps[ref].num_neg, npos = s->st_rps[ref].num_pos;
    int used[33], use_delta[33];
    for (int j = 0; j <= nref; j++) { used[j] = br_bit(b); use_delta[j] = 1; if (!used[j]) use_delta[j] = br_bit(b); }
    const int* rd = s->st_rps[ref].delta_poc;   /* [0..nneg) negatives, [nneg..nneg+npos) positives */
    int dp[16], du[16], i = 0;
    for (int j = npos - 1; j >= 0; j--) { int v = rd[nneg + j] + drps; if (v < 0 && use_delta[nneg + j]) { dp[i] = v; du[i++] = used[nneg + j]; } }
    if (drps < 0 && use_delta[nref]) { dp[i] = drps; du[i++] = used[nref]; }
    for (int j = 0; j < nneg; j++) { int v = rd[j] + drps; if (v < 0 && use_delta[j]) { dp[i] = v; du[i++] = used[j]; } }
    int nn = i;
    for (int j = nneg - 1; j >= 0; j--) { int v = rd[j] + drps; if (v > 0 && use_delta[j]) { dp[i] = v; du[i++] = used[j]; } }
    if (drps > 0 && use_delta[nref]) { dp[i] = drps; du[i++] = used[nref]; }
    for (int j = 0; j < npos; j++) { int v = rd[nneg + j] + drps; if (v > 0 && use_delta[nneg + j]) { dp[i] = v; du[i++] = used[nneg + j]; } }
    if (i > 16) return -1;
    s->st_rps[idx].num_neg = nn; s->st_rps[idx].num_pos = i - nn; s->st_rps[idx].num = i;
    memcpy(s->st_rps[idx].delta_poc, dp, sizeof(int) * i); memcpy(s->st_rps[idx].used, du, sizeof(int) * i);
  } else {
    int nn = (int)br_ue(b), np = (int)br_ue(b);
    if (nn + np > 16) return -1;
    int poc = 0;
    for (int i = 0; i < nn; i++) { poc -= (int)br_ue(b) + 1; s->st_rps[idx].delta_poc[i] = poc; s->st_rps[idx].used[i] = br_bit(b); }
    poc = 0;
    for (int i = 0; i < np; i++) { poc += (int)br_ue(b) + 1; s->st_rps[idx].delta_poc[nn + i] = poc; s->st_rps[idx].used[nn + i] = br_bit(b); }
    s->st_rps[idx].num_neg = nn; s->st_rps[idx].num_pos = np; s->st_rps[idx].num = nn + np;
  }
  return 0;
}
static int ceil_log2(unsigned v) { int n = 0; while ((1u << n) < v) n++; return n; }

static int parse_sps(oracle_hevc_decoder* d, bitreader* b) {
  hevc_sps s; memset(&s, 0, sizeof(s));
  s.vps_id = br_u(b, 4); int msl = br_u(b, 3); s.max_sub_layers = msl + 1; br_bit(b);
  skip_ptl(b, msl);
  s.sps_id = br_ue(b); if (s.sps_id > 15) return -1;
  s.chroma_format_idc = br_ue(b);
  if (s.chroma_format_idc != 1) { DEC_ERR("only 4:2:0 supported (chroma_format_idc=%d)", s.chroma_format_idc); return -1; }
  s.width = br_ue(b); s.height = br_ue(b);
  if (br_bit(b)) for (int i = 0; i < 4; i++) s.conf_win[i] = br_ue(b);
  s.bit_depth = 8 + br_ue(b); s.bit_depth_c = 8 + br_ue(b);
  if (s.bit_depth != s.bit_depth_c || s.bit_depth > 12) { DEC_ERR("unsupported bit depth"); return -1; }
  s.log2_max_poc_lsb = 4 + br_ue(b);
  int sub_info = br_bit(b);
  for (int i = sub_info ? 0 : msl; i <= msl; i++) { s.max_dec_pic_buffering = br_ue(b) + 1; s.num_reorder = br_ue(b); s.max_latency = br_ue(b); }
  s.log2_min_cb = 3 + br_ue(b); s.log2_diff_max_min_cb = br_ue(b); s.log2_ctb = s.log2_min_cb + s.log2_diff_max_min_cb;
  s.log2_min_tb = 2 + br_ue(b); s.log2_diff_max_min_tb = br_ue(b); s.log2_max_tb = s.log2_min_tb + s.log2_diff_max_min_tb;
  s.max_th_depth_inter = br_ue(b); s.max_th_depth_intra = br_ue(b);
  s.scaling_list_enabled = br_bit(b);
  if (s.scaling_list_enabled) { DEC_ERR("scaling lists unsupported"); return -1; }
  s.amp_enabled = br_bit(b); s.sao_enabled = br_bit(b); s.pcm_enabled = br_bit(b);
  if (s.pcm_enabled) { DEC_ERR("PCM unsupported"); return -1; }
  s.num_st_rps = br_ue(b); if (s.num_st_rps > 64) return -1;
  for (int i = 0; i < s.num_st_rps; i++) if (parse_st_rps(b, &s, i, 0)) return -1;
  s.long_term_ref_pics_present = br_bit(b);
  if (s.long_term_ref_pics_present) { DEC_ERR("long-term reference pictures unsupported"); return -1; }
  s.temporal_mvp_enabled = br_bit(b); s.strong_intra_smoothing = br_bit(b);
  /* vui / extensions are not needed for reconstruction */
  if (s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_max_tb > 5 || s.width > HEVC_MAX_W || s.height > HEVC_MAX_H ||
      (s.width & ((1 << s.log2_min_cb) - 1)) || (s.height & ((1 << s.log2_min_cb) - 1))) { DEC_ERR("bad SPS geometry"); return -1; }
  s.pic_w_ctb = (s.width + (1 << s.log2_ctb) - 1) >> s.log2_ctb; s.pic_h_ctb = (s.height + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
  s.valid = 1; d->sps[s.sps_id] = s;
  return 0;
}
static int parse_pps(oracle_hevc_decoder* d, bitreader* b) {
  hevc_pps p; memset(&p, 0, sizeof(p));
  p.pps_id = br_ue(b); p.sps_id = br_ue(b); if (p.pps_id > 63 || p.sps_id > 15) return -1;
  p.dependent_slice_segments_enabled = br_bit(b); p.output_flag_present = br_bit(b); p.num_extra_slice_header_bits = br_u(b, 3);
  p.sign_data_hiding = br_bit(b); p.cabac_init_present = br_bit(b);
  p.num_ref_idx_default[0] = br_ue(b) + 1; p.num_ref_idx_default[1] = br_ue(b) + 1;
  p.init_qp = 26 + br_se(b);
  p.constrained_intra_pred = br_bit(b); p.transform_skip_enabled = br_bit(b);
  p.cu_qp_delta_enabled = br_bit(b); if (p.cu_qp_delta_enabled) p.diff_cu_qp_delta_depth = br_ue(b);
  p.cb_qp_offset = br_se(b); p.cr_qp_offset = br_se(b); p.slice_chroma_qp_offsets_present = br_bit(b);
  p.weighted_pred = br_bit(b); p.weighted_bipred = br_bit(b);
  p.transquant_bypass_enabled = br_bit(b); p.tiles_enabled = br_bit(b); p.entropy_coding_sync = br_bit(b);
  if (p.tiles_enabled) { DEC_ERR("tiles unsupported"); return -1; }

  p.loop_filter_across_slices = br_bit(b);
  p.deblocking_control_present = br_bit(b);
  if (p.deblocking_control_present) {
    p.deblocking_override_enabled = br_bit(b); p.pps_deblocking_disabled = br_bit(b);
    if (!p.pps_deblocking_disabled) { p.beta_offset_div2 = br_se(b); p.tc_offset_div2 = br_se(b); }
  }
  if (br_bit(b)) { DEC_ERR("pps scaling list unsupported"); return -1; }
  p.lists_modification_present = br_bit(b); p.log2_parallel_merge_level = 2 + br_ue(b);
  p.slice_header_extension_present = br_bit(b);
  if (p.log2_parallel_merge_level != 2) { DEC_ERR("parallel merge level > 2 unsupported"); return -1; }
  p.valid = 1; d->pps[p.pps_id] = p;
  return 0;
}

/* ================================================================================================ slice header (7.3.6) */
static int parse_slice_header(oracle_hevc_decoder* d, bitreader* b, int nal_type, hevc_slice_hdr* h, hevc_sps** sps_out, hevc_pps** pps_out) {
  memset(h, 0, sizeof(*h)); h->nal_type = nal_type;
  h->first_slice_in_pic = br_bit(b);
  if (nal_type >= 16 && nal_type <= 23) h->no_output_of_prior_pics = br_bit(b);
  h->pps_id = br_ue(b); if (h->pps_id > 63 || !d->pps[h->pps_id].valid) { DEC_ERR("slice refers to missing PPS %d", h->pps_id); return -1; }
  hevc_pps* pps = &d->pps[h->pps_id]; hevc_sps* sps = &d->sps[pps->sps_id];
  for (int i = 0; i < 4; i++) d->last_conf_win[i] = sps->conf_win[i];
  if (!sps->valid) { DEC_ERR("missing SPS"); return -1; }
  *sps_out = sps; *pps_out = pps;
  if (!h->first_slice_in_pic) {
    if (pps->dependent_slice_segments_enabled) h->dependent = br_bit(b);
    h->segment_addr = br_u(b, ceil_log2(sps->pic_w_ctb * sps->pic_h_ctb));
  }
  if (h->dependent) {   /* everything up to the entry points is that of the slice's independent segment */
    if (!d->have_last_sh) { DEC_ERR("dependent slice segment without a slice"); return -1; }
    int addr = h->segment_addr; *h = d->last_sh; h->first_slice_in_pic = 0; h->dependent = 1; h->segment_addr = addr; h->nal_type = nal_type;
    goto entry_points;
  }
  for (int i = 0; i < pps->num_extra_slice_header_bits; i++) br_bit(b);
  h->slice_type = br_ue(b);
  if (h->slice_type == SLICE_B) { DEC_ERR("B slices unsupported"); return -1; }
  h->pic_output = 1; if (pps->output_flag_present) h->pic_output = br_bit(b);
  int idr = nal_type == NAL_IDR_W_RADL || nal_type == NAL_IDR_N_LP;
  if (!idr) {
    h->poc_lsb = br_u(b, sps->log2_max_poc_lsb);
    h->short_term_ref_pic_set_sps_flag = br_bit(b);
    int ri;
    if (!h->short_term_ref_pic_set_sps_flag) { if (parse_st_rps(b, sps, sps->num_st_rps, 1)) return -1; ri = sps->num_st_rps; }
    else { ri = sps->num_st_rps > 1 ? (int)br_u(b, ceil_log2(sps->num_st_rps)) : 0; }
    h->rps_num = sps->st_rps[ri].num;
    memcpy(h->rps_delta, sps->st_rps[ri].delta_poc, sizeof(int) * 16); memcpy(h->rps_used, sps->st_rps[ri].used, sizeof(int) * 16);
    h->st_rps_idx = ri;
    if (sps->temporal_mvp_enabled) h->temporal_mvp = br_bit(b);
  }
  if (sps->sao_enabled) { h->sao_luma = br_bit(b); h->sao_chroma = br_bit(b); }
  h->num_ref_idx[0] = pps->num_ref_idx_default[0];
  if (h->slice_type == SLICE_P) {
    if (br_bit(b)) h->num_ref_idx[0] = br_ue(b) + 1;
    int ntot = 0; for (int i = 0; i < h->rps_num; i++) ntot += h->rps_used[i];
    if (pps->lists_modification_present && ntot > 1) { if (br_bit(b)) { DEC_ERR("ref_pic_lists_modification unsupported"); return -1; } }
    if (pps->cabac_init_present) h->cabac_init_flag = br_bit(b);
    h->collocated_from_l0 = 1;
    if (h->temporal_mvp && h->num_ref_idx[0] > 1) h->collocated_ref_idx = br_ue(b);
    if (pps->weighted_pred) {   /* pred_weight_table() 7.3.6.3, semantics 7.4.7.3 (no high_precision_offsets: wpOffsetHalfRangeC = 128) */
      const int n = h->num_ref_idx[0]; if (n > 16) return -1;
      h->wp_luma_denom = (int)br_ue(b); h->wp_chroma_denom = h->wp_luma_denom + br_se(b);
      if (h->wp_luma_denom > 7 || h->wp_chroma_denom < 0 || h->wp_chroma_denom > 7) { DEC_ERR("pred_weight_table: weight denominator out of range"); return -1; }
      for (int i = 0; i < n; i++) h->wp_luma_flag[i] = br_bit(b);
      for (int i = 0; i < n; i++) h->wp_chroma_flag[i] = br_bit(b);
      for (int i = 0; i < n; i++) {
        h->wp_w[i][0] = 1 << h->wp_luma_denom; h->wp_o[i][0] = 0;
        h->wp_w[i][1] = h->wp_w[i][2] = 1 << h->wp_chroma_denom; h->wp_o[i][1] = h->wp_o[i][2] = 0;
        if (h->wp_luma_flag[i]) { const int dw = br_se(b), o = br_se(b); if (dw < -128 || dw > 127 || o < -128 || o > 127) return -1; h->wp_w[i][0] += dw; h->wp_o[i][0] = o; }
        if (h->wp_chroma_flag[i]) for (int j = 1; j < 3; j++) {
          const int dw = br_se(b), dof = br_se(b); if (dw < -128 || dw > 127 || dof < -512 || dof > 511) return -1;
          h->wp_w[i][j] += dw;
          h->wp_o[i][j] = clip3(-128, 127, 128 + dof - ((128 * h->wp_w[i][j]) >> h->wp_chroma_denom));
        }
      }
    }
    h->max_merge_cand = 5 - (int)br_ue(b);
    if (h->max_merge_cand < 1 || h->max_merge_cand > 5) return -1;
  }
  h->qp_delta = br_se(b);
  if (pps->slice_chroma_qp_offsets_present) { h->cb_qp_offset = br_se(b); h->cr_qp_offset = br_se(b); }
  h->deblocking_disabled = pps->pps_deblocking_disabled; h->beta_offset_div2 = pps->beta_offset_div2; h->tc_offset_div2 = pps->tc_offset_div2;
  int ovr = 0; if (pps->deblocking_override_enabled) ovr = br_bit(b);
  if (ovr) { h->deblocking_disabled = br_bit(b); if (!h->deblocking_disabled) { h->beta_offset_div2 = br_se(b); h->tc_offset_div2 = br_se(b); } }
  h->loop_filter_across_slices = pps->loop_filter_across_slices;
  if (pps->loop_filter_across_slices && (h->sao_luma || h->sao_chroma || !h->deblocking_disabled)) h->loop_filter_across_slices = br_bit(b);
  h->qp = pps->init_qp + h->qp_delta;
entry_points:
  h->num_entry_points = 0;
  if (pps->entropy_coding_sync) {   /* the offsets only matter to a decoder that starts the substreams in parallel: read and dropped */
    h->num_entry_points = (int)br_ue(b);
    if (h->num_entry_points > sps->pic_h_ctb) { DEC_ERR("slice header: %d entry points", h->num_entry_points); return -1; }
    if (h->num_entry_points > 0) { int len = (int)br_ue(b) + 1; if (len > 32) return -1; for (int i = 0; i < h->num_entry_points; i++) br_u(b, len); }
  }
  if (pps->slice_header_extension_present) { int n = br_ue(b); for (int i = 0; i < n; i++) br_u(b, 8); }
  if (!br_bit(b)) { DEC_ERR("slice header: alignment bit missing"); return -1; }
  while (!br_aligned(b)) br_bit(b);
  h->data_bit_offset = b->pos;
  return 0;
}

/* ================================================================================================ CABAC decoder (9.3.4.3) */
typedef struct {
  const uint8_t* d; size_t n; size_t bitpos;
  uint32_t range, offset;
  uint8_t st[CTX_COUNT];   /* pStateIdx << 1 | valMps */
} cabac_dec;
static inline int cb_read_bit(cabac_dec* c) {
  size_t by = c->bitpos >> 3; int v = by < c->n ? (c->d[by] >> (7 - (c->bitpos & 7))) & 1 : 0; c->bitpos++; return v;
}
static void cabac_init_ctx(uint8_t* st, int init_type, int qp) {
  qp = clip3(0, 51, qp);
  for (int i = 0; i < CTX_COUNT; i++) {
    int iv = k_ctx_init[init_type][i];
    int m = (iv >> 4) * 5 - 45, n = ((iv & 15) << 3) - 16;
    int pre = clip3(1, 126, ((m * qp) >> 4) + n);
    int mps = pre <= 63 ? 0 : 1;
    st[i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
  }
}
static void cabac_start(cabac_dec* c, const uint8_t* d, size_t n, size_t bitpos) {
  c->d = d; c->n = n; c->bitpos = bitpos; c->range = 510; c->offset = 0;
  for (int i = 0; i < 9; i++) c->offset = (c->offset << 1) | cb_read_bit(c);
}
static inline int cabac_bin(cabac_dec* c, int ctx) {
  int s = c->st[ctx] >> 1, mps = c->st[ctx] & 1, bin;
  uint32_t lps = k_range_lps[s][(c->range >> 6) & 3];
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = !mps; c->offset -= c->range; c->range = lps;
    if (s == 0) mps = 1 - mps;
    s = k_next_lps[s];
  } else { bin = mps; s = hevc_next_mps(s); }
  c->st[ctx] = (uint8_t)((s << 1) | mps);
  while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | cb_read_bit(c); }
  return bin;
}
static inline int cabac_bypass(cabac_dec* c) {
  c->offset = (c->offset << 1) | cb_read_bit(c);
  if (c->offset >= c->range) { c->offset -= c->range; return 1; }
  return 0;
}
static inline uint32_t cabac_bypass_n(cabac_dec* c, int n) { uint32_t v = 0; while (n--) v = (v << 1) | cabac_bypass(c); return v; }
static inline int cabac_terminate(cabac_dec* c) {
  c->range -= 2;
  if (c->offset >= c->range) return 1;
  while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | cb_read_bit(c); }
  return 0;
}

/* ================================================================================================ slice decoding state */
typedef struct {
  oracle_hevc_decoder* d; const hevc_sps* sps; const hevc_pps* pps; hevc_slice_hdr sh;
  cabac_dec c;
  hevc_frame* f; hevc_meta* m;
  const hevc_frame* ref[16]; const hevc_colinfo* refcol[16]; int ref_poc[16];
  int wp_on;                   /* P slice of a PPS with weighted_pred_flag: explicit weighted sample prediction with the slice header's table */
  int slice_idx;
  int qp_y, qp_y_prev;         /* current CU QpY; QpY of the last CU of the previous quantisation group */
  int is_cu_qp_delta_coded, cu_qp_delta_val, qp_pred;
  int last_pu_merge;
  hevc_mvpred mp;
  int ctb_x, ctb_y;
  int error;
  /* current CU */
  int cu_x, cu_y, cu_log2, cu_pred_mode, cu_part_mode, cu_tq_bypass;
  int intra_luma[4], intra_chroma;
  int max_trafo_depth;
} sdec;

static inline void set_rect8(uint8_t* a, int w4, int x, int y, int w, int h, int v) {
  for (int j = y >> 2; j < (y + h) >> 2; j++) memset(a + (size_t)j * w4 + (x >> 2), v, (size_t)(w >> 2));
}

/* ------------------------------------------------------------------------------------------------ SAO syntax (7.3.8.3) */
static void parse_sao(sdec* s, int rx, int ry) {
  hevc_meta* m = s->m; cabac_dec* c = &s->c;
  hevc_sao* p = &m->sao[ry * m->w_ctb + rx];
  memset(p, 0, sizeof(*p));
  if (!s->sh.sao_luma && !s->sh.sao_chroma) return;
  int merge_left = 0, merge_up = 0;
  if (rx > 0 && m->ctb_slice[ry * m->w_ctb + rx - 1] == s->slice_idx) merge_left = cabac_bin(c, CTX_SAO_MERGE);
  if (ry > 0 && !merge_left && m->ctb_slice[(ry - 1) * m->w_ctb + rx] == s->slice_idx) merge_up = cabac_bin(c, CTX_SAO_MERGE);
  if (merge_left) { *p = m->sao[ry * m->w_ctb + rx - 1]; return; }
  if (merge_up) { *p = m->sao[(ry - 1) * m->w_ctb + rx]; return; }
  int bd = s->sps->bit_depth;
  int cmax = (1 << (imin(bd, 10) - 5)) - 1;
  for (int ci = 0; ci < 3; ci++) {
    if ((ci == 0 && !s->sh.sao_luma) || (ci > 0 && !s->sh.sao_chroma)) continue;
    if (ci == 2) { p->type[2] = p->type[1]; }
    else { int t = 0; if (cabac_bin(c, CTX_SAO_TYPE)) t = cabac_bypass(c) ? 2 : 1; p->type[ci] = (uint8_t)t; }
    if (!p->type[ci]) continue;
    int absv[4];
    for (int i = 0; i < 4; i++) { int v = 0; while (v < cmax && cabac_bypass(c)) v++; absv[i] = v; }
    if (p->type[ci] == 1) {
      for (int i = 0; i < 4; i++) if (absv[i] && cabac_bypass(c)) absv[i] = -absv[i];
      p->band_pos[ci] = (uint8_t)cabac_bypass_n(c, 5);
    } else {
      absv[2] = -absv[2]; absv[3] = -absv[3];
      if (ci == 0) p->eo_class[0] = (uint8_t)cabac_bypass_n(c, 2);
      else if (ci == 1) p->eo_class[1] = (uint8_t)cabac_bypass_n(c, 2);
      else p->eo_class[2] = p->eo_class[1];
    }
    for (int i = 0; i < 4; i++) p->offset[ci][i] = (int8_t)(absv[i] * (1 << (bd - imin(bd, 10))));
  }
}

/* ------------------------------------------------------------------------------------------------ residual_coding (7.3.8.11) */
static void residual_coding(sdec* s, int log2, int c_idx, int scan_idx, int16_t* coeff, int* ts_flag) {
  cabac_dec* c = &s->c; oracle_hevc_decoder* d = s->d;
  int N = 1 << log2;
  memset(coeff, 0, sizeof(int16_t) * N * N);
  *ts_flag = 0;
  if (s->pps->transform_skip_enabled && !s->cu_tq_bypass && log2 <= 2) *ts_flag = cabac_bin(c, CTX_TRANSFORM_SKIP + (c_idx ? 1 : 0));
  /* last significant position */
  int ctx_off, ctx_shift;
  if (c_idx == 0) { ctx_off = 3 * (log2 - 2) + ((log2 - 1) >> 2); ctx_shift = (log2 + 1) >> 2; }
  else { ctx_off = 15; ctx_shift = log2 - 2; }
  int maxp = (log2 << 1) - 1;
  int px = 0, py = 0;
  while (px < maxp && cabac_bin(c, CTX_LAST_X + ctx_off + (px >> ctx_shift))) px++;
  while (py < maxp && cabac_bin(c, CTX_LAST_Y + ctx_off + (py >> ctx_shift))) py++;
  int lx = px, ly = py;
  if (px > 3) { int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + (int)cabac_bypass_n(c, nb); }
  if (py > 3) { int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + (int)cabac_bypass_n(c, nb); }
  if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
  /* locate last sub-block / position in scan order */
  const uint8_t* sb_scan = d->scan[scan_idx][log2 - 2];
  const uint8_t* pos_scan = d->scan[scan_idx][2];
  int n_sb = 1 << (2 * (log2 - 2));
  int last_sb = 0, last_pos = 0;
  { int sbx = lx >> 2, sby = ly >> 2, ix = lx & 3, iy = ly & 3;
    for (int i = 0; i < n_sb; i++) if ((sb_scan[i] & 15) == sbx && (sb_scan[i] >> 4) == sby) { last_sb = i; break; }
    for (int i = 0; i < 16; i++) if ((pos_scan[i] & 15) == ix && (pos_scan[i] >> 4) == iy) { last_pos = i; break; } }
  uint8_t csbf[8][8]; memset(csbf, 0, sizeof(csbf));
  int sbw = 1 << (log2 - 2);
  int greater1_ctx = 1; int first_sb_done = 0;
  int sign_hiding = s->pps->sign_data_hiding && !s->cu_tq_bypass;
  for (int i = last_sb; i >= 0; i--) {
    int xs = sb_scan[i] & 15, ys = sb_scan[i] >> 4;
    int infer_dc = 0, coded;
    int right = xs + 1 < sbw ? csbf[ys][xs + 1] : 0, below = ys + 1 < sbw ? csbf[ys + 1][xs] : 0;
    if (i < last_sb && i > 0) { coded = cabac_bin(c, CTX_CSBF + imin(right + below, 1) + (c_idx ? 2 : 0)); infer_dc = 1; }
    else coded = 1;
    csbf[ys][xs] = (uint8_t)coded;
    if (!coded) continue;
    int sig_pos[16], nsig = 0;
    int start = i == last_sb ? last_pos - 1 : 15;
    if (i == last_sb) sig_pos[nsig++] = last_pos;
    int prev_csbf = right | (below << 1);
    for (int n = start; n >= 0; n--) {
      int xp = pos_scan[n] & 15, yp = pos_scan[n] >> 4;
      int xc = (xs << 2) + xp, yc = (ys << 2) + yp;
      int sig;
      if (n > 0 || !infer_dc) {
        int sc;
        if (log2 == 2) sc = k_sig_ctx_4x4[(yc << 2) + xc];
        else if (xc + yc == 0) sc = 0;
        else {
          if (prev_csbf == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
          else if (prev_csbf == 1) sc = yp == 0 ? 2 : (yp == 1 ? 1 : 0);
          else if (prev_csbf == 2) sc = xp == 0 ? 2 : (xp == 1 ? 1 : 0);
          else sc = 2;
          if (c_idx == 0) { if (xs || ys) sc += 3; sc += log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21; }
          else sc += log2 == 3 ? 9 : 12;
        }
        sig = cabac_bin(c, CTX_SIG + (c_idx == 0 ? sc : 27 + sc));
        if (sig) infer_dc = 0;
      } else sig = 1;   /* inferred DC of a coded sub-block with no other significant coefficient */
      if (sig) sig_pos[nsig++] = n;
    }
    if (!nsig) continue;
    /* greater1 / greater2 */
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    if (first_sb_done && greater1_ctx == 0) ctx_set++;
    first_sb_done = 1;
    greater1_ctx = 1;
    int absl[16], g1[16]; int first_g1 = -1;
    int n8 = imin(nsig, 8);
    for (int k = 0; k < n8; k++) {
      g1[k] = cabac_bin(c, CTX_GT1 + (ctx_set << 2) + greater1_ctx + (c_idx ? 16 : 0));
      if (g1[k]) { greater1_ctx = 0; if (first_g1 < 0) first_g1 = k; }
      else if (greater1_ctx > 0 && greater1_ctx < 3) greater1_ctx++;
      absl[k] = 1 + g1[k];
    }
    for (int k = n8; k < nsig; k++) absl[k] = 1;
    if (first_g1 >= 0) { if (cabac_bin(c, CTX_GT2 + ctx_set + (c_idx ? 4 : 0))) absl[first_g1]++; }
    int hidden = sign_hiding && (sig_pos[0] - sig_pos[nsig - 1] > 3);
    int nsign = nsig - (hidden ? 1 : 0);
    uint32_t signs = cabac_bypass_n(c, nsign) << (16 - nsign);   /* bit 15 = first coefficient */
    /* remaining levels */
    int rice = 0, sum = 0;
    for (int k = 0; k < nsig; k++) {
      int base = k < 8 ? (k == first_g1 ? 3 : 2) : 1;
      if (absl[k] == base) {
        int pre = 0; while (pre < 32 && cabac_bypass(c)) pre++;
        int v;
        if (pre <= 3) v = (pre << rice) + (int)cabac_bypass_n(c, rice);
        else { int sl = pre - 3 + rice; if (sl > 30) { s->error = 1; return; } v = (((1 << (pre - 3)) + 3 - 1) << rice) + (int)cabac_bypass_n(c, sl); }
        absl[k] += v;
        if (absl[k] > 3 * (1 << rice)) rice = imin(rice + 1, 4);
      }
      sum += absl[k];
    }
    for (int k = 0; k < nsig; k++) {
      int neg;
      if (k < nsign) neg = (signs >> (15 - k)) & 1; else neg = (sum & 1);
      int n = sig_pos[k];
      int xc = (xs << 2) + (pos_scan[n] & 15), yc = (ys << 2) + (pos_scan[n] >> 4);
      int v = neg ? -absl[k] : absl[k];
      coeff[yc * N + xc] = (int16_t)clip3(-32768, 32767, v);
    }
  }
}

/* ------------------------------------------------------------------------------------------------ TU reconstruction */
static int chroma_qp_of(sdec* s, int c_idx) {
  int off = c_idx == 1 ? s->pps->cb_qp_offset + s->sh.cb_qp_offset : s->pps->cr_qp_offset + s->sh.cr_qp_offset;
  int bdo = 6 * (s->sps->bit_depth - 8);
  int qpi = clip3(-bdo, 57, s->qp_y + off);
  int qpc = qpi < 0 ? qpi : hevc_chroma_qp(qpi);
  return qpc + bdo;
}
static void recon_tb(sdec* s, int c_idx, int x0, int y0, int log2, int cbf, int intra_mode) {
  /* (x0,y0) in component samples */
  hevc_frame* f = s->f; int N = 1 << log2, pw = c_idx ? f->cw : f->w, bd = f->bit_depth, maxv = (1 << bd) - 1;
  if (s->cu_pred_mode == MODE_INTRA) hevc_intra_pred(f, s->m, c_idx, x0, y0, log2, intra_mode);
  if (!cbf) return;
  int16_t coeff[32 * 32], dq[32 * 32], res[32 * 32]; int ts = 0;
  int scan_idx = 0;
  if (s->cu_pred_mode == MODE_INTRA && (log2 == 2 || (log2 == 3 && c_idx == 0))) {
    if (intra_mode >= 6 && intra_mode <= 14) scan_idx = 2; else if (intra_mode >= 22 && intra_mode <= 30) scan_idx = 1;
  }
  residual_coding(s, log2, c_idx, scan_idx, coeff, &ts);
  if (s->error) return;
  if (s->cu_tq_bypass) memcpy(res, coeff, sizeof(int16_t) * N * N);
  else {
    int qp = c_idx ? chroma_qp_of(s, c_idx) : s->qp_y + 6 * (bd - 8);
    hevc_dequant(coeff, dq, log2, qp, bd);
    if (ts) hevc_inv_transform_skip(dq, res, log2, bd);
    else hevc_inv_transform(dq, res, log2, c_idx == 0 && log2 == 2 && s->cu_pred_mode == MODE_INTRA, bd);
  }
  uint16_t* p = f->p[c_idx] + (size_t)y0 * pw + x0;
  for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) p[(size_t)y * pw + x] = (uint16_t)clip3(0, maxv, p[(size_t)y * pw + x] + res[y * N + x]);
}

/* QpY prediction of a quantisation group (8.6.1) */
static int wrap_qp(sdec* s, int v) { int bdo = 6 * (s->sps->bit_depth - 8); return ((v + 52 + 2 * bdo) % (52 + bdo)) - bdo; }
static void start_quant_group(sdec* s, int xqg, int yqg) {
  const hevc_sps* sps = s->sps; hevc_meta* m = s->m;
  int ctb_mask = ~((1 << sps->log2_ctb) - 1);
  s->qp_y_prev = s->qp_y; s->is_cu_qp_delta_coded = 0; s->cu_qp_delta_val = 0;
  int qa = s->qp_y_prev, qb = s->qp_y_prev;
  if (xqg > 0 && ((xqg - 1) & ctb_mask) == (xqg & ctb_mask) && hevc_avail_cu(m, xqg, yqg, xqg - 1, yqg)) qa = m->qp[meta_idx(m, xqg - 1, yqg)];
  if (yqg > 0 && ((yqg - 1) & ctb_mask) == (yqg & ctb_mask) && hevc_avail_cu(m, xqg, yqg, xqg, yqg - 1)) qb = m->qp[meta_idx(m, xqg, yqg - 1)];
  s->qp_pred = (qa + qb + 1) >> 1;
}

static void transform_unit(sdec* s, int x0, int y0, int xb, int yb, int log2, int depth, int blk, int cbf_luma, int cbf_cb, int cbf_cr) {
  cabac_dec* c = &s->c; hevc_meta* m = s->m;
  int N = 1 << log2;
  int chroma_here = log2 > 2 || blk == 3;
  int any_chroma = cbf_cb || cbf_cr;
  if ((cbf_luma || (any_chroma)) && s->pps->cu_qp_delta_enabled && !s->is_cu_qp_delta_coded) {
    int v = 0; while (v < 5 && cabac_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0))) v++;
    if (v == 5) { int k = 0; while (k < 16 && cabac_bypass(c)) { v += 1 << k; k++; } v += (int)cabac_bypass_n(c, k); }
    if (v && cabac_bypass(c)) v = -v;
    s->is_cu_qp_delta_coded = 1; s->cu_qp_delta_val = v;
    s->qp_y = wrap_qp(s, s->qp_pred + v);
    set_rect8((uint8_t*)m->qp, m->w4, s->cu_x, s->cu_y, 1 << s->cu_log2, 1 << s->cu_log2, (uint8_t)(int8_t)s->qp_y);
  }
  /* luma */
  int part = 0;
  if (s->cu_part_mode == PART_NxN && s->cu_pred_mode == MODE_INTRA) part = ((y0 - s->cu_y) >= (1 << (s->cu_log2 - 1)) ? 2 : 0) + ((x0 - s->cu_x) >= (1 << (s->cu_log2 - 1)) ? 1 : 0);
  set_rect8(m->nz, m->w4, x0, y0, N, N, cbf_luma ? 1 : 0);
  for (int i = 0; i < N; i += 4) { m->edge_v[meta_idx(m, x0, y0 + i)] |= 1; m->edge_h[meta_idx(m, x0 + i, y0)] |= 1; }
  recon_tb(s, 0, x0, y0, log2, cbf_luma, s->intra_luma[part]);
  if (s->cu_pred_mode == MODE_INTRA) set_rect8(m->done, m->w4, x0, y0, N, N, 1);
  if (s->error) return;
  if (chroma_here) {
    int xc = (log2 > 2 ? x0 : xb) >> 1, yc = (log2 > 2 ? y0 : yb) >> 1, l2c = log2 > 2 ? log2 - 1 : 2;
    recon_tb(s, 1, xc, yc, l2c, cbf_cb, s->intra_chroma);
    if (s->error) return;
    recon_tb(s, 2, xc, yc, l2c, cbf_cr, s->intra_chroma);
  }
}

static void transform_tree(sdec* s, int x0, int y0, int xb, int yb, int log2, int depth, int blk, int pcb, int pcr) {
  cabac_dec* c = &s->c; const hevc_sps* sps = s->sps;
  if (s->error) return;
  int intra_split = s->cu_pred_mode == MODE_INTRA && s->cu_part_mode == PART_NxN;
  int inter_split = sps->max_th_depth_inter == 0 && s->cu_pred_mode != MODE_INTRA && s->cu_part_mode != PART_2Nx2N && depth == 0;
  int split;
  if (log2 <= sps->log2_max_tb && log2 > sps->log2_min_tb && depth < s->max_trafo_depth && !(intra_split && depth == 0))
    split = cabac_bin(c, CTX_SPLIT_TRANSFORM + 5 - log2);
  else split = (log2 > sps->log2_max_tb || (intra_split && depth == 0) || inter_split) ? 1 : 0;
  int cbf_cb = 0, cbf_cr = 0;
  if (log2 > 2) {
    if (depth == 0 || pcb) cbf_cb = cabac_bin(c, CTX_CBF_CHROMA + depth);
    if (depth == 0 || pcr) cbf_cr = cabac_bin(c, CTX_CBF_CHROMA + depth);
  } else { cbf_cb = pcb; cbf_cr = pcr; }
  if (split) {
    int h = 1 << (log2 - 1);
    transform_tree(s, x0, y0, x0, y0, log2 - 1, depth + 1, 0, cbf_cb, cbf_cr);
    transform_tree(s, x0 + h, y0, x0, y0, log2 - 1, depth + 1, 1, cbf_cb, cbf_cr);
    transform_tree(s, x0, y0 + h, x0, y0, log2 - 1, depth + 1, 2, cbf_cb, cbf_cr);
    transform_tree(s, x0 + h, y0 + h, x0, y0, log2 - 1, depth + 1, 3, cbf_cb, cbf_cr);
  } else {
    int cbf_luma = 1;
    if (s->cu_pred_mode == MODE_INTRA || depth != 0 || cbf_cb || cbf_cr) cbf_luma = cabac_bin(c, CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
    transform_unit(s, x0, y0, xb, yb, log2, depth, blk, cbf_luma, cbf_cb, cbf_cr);
  }
}

/* ------------------------------------------------------------------------------------------------ motion data */
static int parse_mvd_comp(cabac_dec* c, int gt0, int gt1) {
  if (!gt0) return 0;
  int v = 1;
  if (gt1) { int k = 1; v = 2; while (k < 24 && cabac_bypass(c)) { v += 1 << k; k++; } v += (int)cabac_bypass_n(c, k); }
  return cabac_bypass(c) ? -v : v;
}
static void prediction_unit(sdec* s, int xcb, int ycb, int x0, int y0, int w, int h, int part_idx, int skip) {
  cabac_dec* c = &s->c; hevc_meta* m = s->m;
  hevc_mvcand mv;
  int merge = skip ? 1 : cabac_bin(c, CTX_MERGE_FLAG);
  s->last_pu_merge = merge;
  if (merge) {
    int idx = 0;
    if (s->sh.max_merge_cand > 1) { idx = cabac_bin(c, CTX_MERGE_IDX); if (idx) while (idx < s->sh.max_merge_cand - 1 && cabac_bypass(c)) idx++; }
    s->mp.part_mode = s->cu_part_mode;
    mv = hevc_merge_candidate(&s->mp, x0, y0, w, h, part_idx, idx);
  } else {
    int ref_idx = 0;
    if (s->sh.num_ref_idx[0] > 1) {
      int mx = s->sh.num_ref_idx[0] - 1;
      while (ref_idx < mx) { int b = ref_idx < 2 ? cabac_bin(c, CTX_REF_IDX + ref_idx) : cabac_bypass(c); if (!b) break; ref_idx++; }
    }
    int gx0 = cabac_bin(c, CTX_MVD_GT0), gy0 = cabac_bin(c, CTX_MVD_GT0);
    int gx1 = gx0 ? cabac_bin(c, CTX_MVD_GT1) : 0, gy1 = gy0 ? cabac_bin(c, CTX_MVD_GT1) : 0;
    int dx = parse_mvd_comp(c, gx0, gx1), dy = parse_mvd_comp(c, gy0, gy1);
    int mvp = cabac_bin(c, CTX_MVP_FLAG);
    mv = hevc_amvp_candidate(&s->mp, x0, y0, w, h, ref_idx, mvp);
    mv.x = (int16_t)(mv.x + dx); mv.y = (int16_t)(mv.y + dy);
  }
  if (mv.ref < 0 || mv.ref >= s->sh.num_ref_idx[0] || !s->ref[mv.ref]) { DEC_ERR("bad reference index %d", mv.ref); s->error = 1; return; }
  for (int j = y0 >> 2; j < (y0 + h) >> 2; j++)
    for (int i = x0 >> 2; i < (x0 + w) >> 2; i++) {
      int k = j * m->w4 + i; m->mv[2 * k] = mv.x; m->mv[2 * k + 1] = mv.y; m->ref_idx[k] = (int8_t)mv.ref; m->pred_mode[k] = (uint8_t)(skip ? MODE_SKIP : MODE_INTER);
    }
  for (int i = 0; i < h; i += 4) m->edge_v[meta_idx(m, x0, y0 + i)] |= 2;
  for (int i = 0; i < w; i += 4) m->edge_h[meta_idx(m, x0 + i, y0)] |= 2;
  if (s->wp_on) {   /* explicit weighted sample prediction (8.5.3.3.4.3): log2WD = denominator + shift1 (14 - bitDepth), offsets at the sample bit depth */
    const int bd = s->f->bit_depth; hevc_wp wp;
    for (int c = 0; c < 3; c++) { wp.w[c] = s->sh.wp_w[mv.ref][c]; wp.o[c] = s->sh.wp_o[mv.ref][c] * (1 << (bd - 8)); }
    wp.shift[0] = s->sh.wp_luma_denom + 14 - bd; wp.shift[1] = s->sh.wp_chroma_denom + 14 - bd;
    hevc_inter_pred_wp(s->f, s->ref[mv.ref], x0, y0, w, h, mv.x, mv.y, &wp);
  } else hevc_inter_pred(s->f, s->ref[mv.ref], x0, y0, w, h, mv.x, mv.y);
}

/* ------------------------------------------------------------------------------------------------ coding unit (7.3.8.5) */
static void coding_unit(sdec* s, int x0, int y0, int log2, int depth) {
  cabac_dec* c = &s->c; hevc_meta* m = s->m; const hevc_sps* sps = s->sps;
  int N = 1 << log2;
  s->cu_x = x0; s->cu_y = y0; s->cu_log2 = log2; s->cu_tq_bypass = 0; s->cu_part_mode = PART_2Nx2N; s->cu_pred_mode = MODE_INTRA;
  if (s->pps->cu_qp_delta_enabled) s->qp_y = wrap_qp(s, s->qp_pred + s->cu_qp_delta_val);
  if (s->pps->transquant_bypass_enabled) s->cu_tq_bypass = cabac_bin(c, CTX_CU_TQ_BYPASS);
  int skip = 0;
  if (s->sh.slice_type != SLICE_I) {
    int cl = hevc_avail_cu(m, x0, y0, x0 - 1, y0) && m->pred_mode[meta_idx(m, x0 - 1, y0)] == MODE_SKIP;
    int ca = hevc_avail_cu(m, x0, y0, x0, y0 - 1) && m->pred_mode[meta_idx(m, x0, y0 - 1)] == MODE_SKIP;
    skip = cabac_bin(c, CTX_CU_SKIP + cl + ca);
  }
  set_rect8(m->cu_depth, m->w4, x0, y0, N, N, depth);
  set_rect8(m->tq_bypass, m->w4, x0, y0, N, N, s->cu_tq_bypass);
  set_rect8((uint8_t*)m->qp, m->w4, x0, y0, N, N, (uint8_t)(int8_t)s->qp_y);
  for (int i = 0; i < N; i += 4) { m->edge_v[meta_idx(m, x0, y0 + i)] |= 3; m->edge_h[meta_idx(m, x0 + i, y0)] |= 3; }
  if (skip) {
    s->cu_pred_mode = MODE_SKIP;
    prediction_unit(s, x0, y0, x0, y0, N, N, 0, 1);
    set_rect8(m->done, m->w4, x0, y0, N, N, 1);
    return;
  }
  if (s->sh.slice_type != SLICE_I) s->cu_pred_mode = cabac_bin(c, CTX_PRED_MODE) ? MODE_INTRA : MODE_INTER;
  if (s->cu_pred_mode == MODE_INTRA) {
    if (log2 == sps->log2_min_cb) s->cu_part_mode = cabac_bin(c, CTX_PART_MODE) ? PART_2Nx2N : PART_NxN;
    if (s->cu_part_mode == PART_NxN && log2 == 3 && sps->log2_min_tb > 2) { s->error = 1; return; }
  } else {
    if (cabac_bin(c, CTX_PART_MODE)) s->cu_part_mode = PART_2Nx2N;
    else if (log2 == sps->log2_min_cb) {
      if (log2 == 3) s->cu_part_mode = cabac_bin(c, CTX_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
      else if (cabac_bin(c, CTX_PART_MODE + 1)) s->cu_part_mode = PART_2NxN;
      else s->cu_part_mode = cabac_bin(c, CTX_PART_MODE + 2) ? PART_Nx2N : PART_NxN;
    } else if (!sps->amp_enabled) s->cu_part_mode = cabac_bin(c, CTX_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
    else {
      int hor = cabac_bin(c, CTX_PART_MODE + 1);
      if (cabac_bin(c, CTX_PART_MODE + 3)) s->cu_part_mode = hor ? PART_2NxN : PART_Nx2N;
      else { int b = cabac_bypass(c); s->cu_part_mode = hor ? (b ? PART_2NxnD : PART_2NxnU) : (b ? PART_nRx2N : PART_nLx2N); }
    }
  }
  if (s->cu_pred_mode == MODE_INTRA) {
    set_rect8(m->pred_mode, m->w4, x0, y0, N, N, MODE_INTRA);
    int np = s->cu_part_mode == PART_NxN ? 4 : 1, pb = N >> (np == 4);
    int prev[4], mpm_idx[4], rem[4];
    for (int i = 0; i < np; i++) prev[i] = cabac_bin(c, CTX_PREV_INTRA_LUMA);
    for (int i = 0; i < np; i++) {
      if (prev[i]) { mpm_idx[i] = cabac_bypass(c); if (mpm_idx[i]) mpm_idx[i] += cabac_bypass(c); }
      else rem[i] = (int)cabac_bypass_n(c, 5);
    }
    for (int i = 0; i < np; i++) {
      int xp = x0 + (i & 1) * pb, yp = y0 + (i >> 1) * pb;
      int cand[3]; hevc_intra_mpm(m, xp, yp, cand);
      int mode;
      if (prev[i]) mode = cand[mpm_idx[i]];
      else {
        if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
        if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
        if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
        mode = rem[i];
        for (int k = 0; k < 3; k++) if (mode >= cand[k]) mode++;
      }
      s->intra_luma[i] = mode;
      set_rect8(m->intra_mode, m->w4, xp, yp, pb, pb, mode);
    }
    int icp = 4;
    if (cabac_bin(c, CTX_INTRA_CHROMA)) icp = (int)cabac_bypass_n(c, 2);
    static const int cm[4] = {0, 26, 10, 1};
    if (icp == 4) s->intra_chroma = s->intra_luma[0];
    else s->intra_chroma = cm[icp] == s->intra_luma[0] ? 34 : cm[icp];
  } else {
    int h2 = N >> 1, q = N >> 2;
    switch (s->cu_part_mode) {
      case PART_2Nx2N: prediction_unit(s, x0, y0, x0, y0, N, N, 0, 0); break;
      case PART_2NxN: prediction_unit(s, x0, y0, x0, y0, N, h2, 0, 0); prediction_unit(s, x0, y0, x0, y0 + h2, N, h2, 1, 0); break;
      case PART_Nx2N: prediction_unit(s, x0, y0, x0, y0, h2, N, 0, 0); prediction_unit(s, x0, y0, x0 + h2, y0, h2, N, 1, 0); break;
      case PART_2NxnU: prediction_unit(s, x0, y0, x0, y0, N, q, 0, 0); prediction_unit(s, x0, y0, x0, y0 + q, N, N - q, 1, 0); break;
      case PART_2NxnD: prediction_unit(s, x0, y0, x0, y0, N, N - q, 0, 0); prediction_unit(s, x0, y0, x0, y0 + N - q, N, q, 1, 0); break;
      case PART_nLx2N: prediction_unit(s, x0, y0, x0, y0, q, N, 0, 0); prediction_unit(s, x0, y0, x0 + q, y0, N - q, N, 1, 0); break;
      case PART_nRx2N: prediction_unit(s, x0, y0, x0, y0, N - q, N, 0, 0); prediction_unit(s, x0, y0, x0 + N - q, y0, q, N, 1, 0); break;
      default:
        prediction_unit(s, x0, y0, x0, y0, h2, h2, 0, 0); prediction_unit(s, x0, y0, x0 + h2, y0, h2, h2, 1, 0);
        prediction_unit(s, x0, y0, x0, y0 + h2, h2, h2, 2, 0); prediction_unit(s, x0, y0, x0 + h2, y0 + h2, h2, h2, 3, 0);
    }
    if (s->error) return;
  }
  int rqt_root_cbf = 1;
  if (s->cu_pred_mode != MODE_INTRA && !(s->cu_part_mode == PART_2Nx2N && s->last_pu_merge)) rqt_root_cbf = cabac_bin(c, CTX_RQT_ROOT_CBF);
  if (rqt_root_cbf) {
    s->max_trafo_depth = s->cu_pred_mode == MODE_INTRA ? sps->max_th_depth_intra + (s->cu_part_mode == PART_NxN) : sps->max_th_depth_inter;
    transform_tree(s, x0, y0, x0, y0, log2, 0, 0, 0, 0);
  }
  set_rect8(m->done, m->w4, x0, y0, N, N, 1);
}

/* ------------------------------------------------------------------------------------------------ coding quadtree (7.3.8.4) */
static void coding_quadtree(sdec* s, int x0, int y0, int log2, int depth) {
  cabac_dec* c = &s->c; hevc_meta* m = s->m; const hevc_sps* sps = s->sps;
  if (s->error) return;
  int N = 1 << log2, split;
  if (x0 + N <= sps->width && y0 + N <= sps->height && log2 > sps->log2_min_cb) {
    int cl = hevc_avail_cu(m, x0, y0, x0 - 1, y0) && m->cu_depth[meta_idx(m, x0 - 1, y0)] > depth;
    int ca = hevc_avail_cu(m, x0, y0, x0, y0 - 1) && m->cu_depth[meta_idx(m, x0, y0 - 1)] > depth;
    split = cabac_bin(c, CTX_SPLIT_CU + cl + ca);
  } else split = log2 > sps->log2_min_cb;
  if (s->pps->cu_qp_delta_enabled && log2 >= sps->log2_ctb - s->pps->diff_cu_qp_delta_depth) start_quant_group(s, x0, y0);
  if (split) {
    int h = N >> 1;
    coding_quadtree(s, x0, y0, log2 - 1, depth + 1);
    if (x0 + h < sps->width) coding_quadtree(s, x0 + h, y0, log2 - 1, depth + 1);
    if (y0 + h < sps->height) coding_quadtree(s, x0, y0 + h, log2 - 1, depth + 1);
    if (x0 + h < sps->width && y0 + h < sps->height) coding_quadtree(s, x0 + h, y0 + h, log2 - 1, depth + 1);
  } else coding_unit(s, x0, y0, log2, depth);
}

/* ================================================================================================ picture management */
static void finish_picture(oracle_hevc_decoder* d) {
  if (!d->pic_open) return;
  hevc_frame* f = d->cur; hevc_meta* m = d->meta;
  /* collocated motion for later pictures (before loop filters; motion only) */
  hevc_colinfo* ci = &d->col[d->cur_idx];
  ci->poc = d->cur_poc; ci->w4 = m->w4; ci->h4 = m->h4;
  size_t n4 = (size_t)m->w4 * m->h4;
  ci->mv = (int16_t*)malloc(n4 * 4); ci->refpoc = (int32_t*)malloc(n4 * 4);
  memcpy(ci->mv, m->mv, n4 * 4);
  ci->imode = (uint8_t*)malloc(n4); for (size_t i = 0; i < n4; i++) ci->imode[i] = m->pred_mode[i] == MODE_INTRA ? m->intra_mode[i] : 255;   /* for a transcoder's re-encode (oracle_hevc_dec_imodes) */
  for (int y = 0; y < m->h4; y++) for (int x = 0; x < m->w4; x++) {
    size_t i = (size_t)y * m->w4 + x;
    if (m->pred_mode[i] == MODE_INTRA || m->pred_mode[i] == META_UNDECODED) ci->refpoc[i] = INT_MIN;
    else ci->refpoc[i] = m->slices[meta_slice_at(m, x * 4, y * 4)].ref_poc[m->ref_idx[i]];
  }
  hevc_deblock(f, m);
  int any_sao = 0; for (int i = 0; i < m->n_slices; i++) any_sao |= m->slices[i].sao_luma | m->slices[i].sao_chroma;
  if (any_sao) { hevc_frame* t = hevc_frame_alloc(f->w, f->h, f->bit_depth); hevc_sao_apply(t, f, m); hevc_frame_copy(f, t); hevc_frame_free(t); }
  if (d->have_md5) {
    d->md5_checked++;
    int bad = 0;
    for (int c = 0; c < 3; c++) { uint8_t h[16]; oracle_md5_plane(f->p[c], c ? f->cw : f->w, c ? f->ch : f->h, f->bit_depth, h); if (memcmp(h, d->pending_md5[c], 16)) bad = 1; }
    if (bad) { d->md5_failed++; DEC_ERR("MD5 mismatch on picture %d (POC %d)", d->cur_idx, d->cur_poc); }
    d->have_md5 = 0;
  }
  d->pic_open = 0;
}
static int start_picture(oracle_hevc_decoder* d, const hevc_sps* sps, const hevc_pps* pps, hevc_slice_hdr* h) {
  finish_picture(d);
  int idr = h->nal_type == NAL_IDR_W_RADL || h->nal_type == NAL_IDR_N_LP;
  if (idr) h->poc = 0;
  else {
    int max_lsb = 1 << sps->log2_max_poc_lsb, prev_lsb = d->prev_tid0_poc & (max_lsb - 1), prev_msb = d->prev_tid0_poc - prev_lsb, msb;
    if (h->poc_lsb < prev_lsb && prev_lsb - h->poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (h->poc_lsb > prev_lsb && h->poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    else msb = prev_msb;
    if (h->nal_type >= 16 && h->nal_type <= 23 && h->nal_type != NAL_CRA) msb = 0;
    h->poc = msb + h->poc_lsb;
  }
  if (!nal_keeps_poc_anchor(h->nal_type)) d->prev_tid0_poc = h->poc;
  d->cur_poc = h->poc;
  if (d->n_out == d->cap_out) {
    d->cap_out = d->cap_out ? d->cap_out * 2 : 64;
    d->out = (hevc_frame**)realloc(d->out, sizeof(void*) * d->cap_out); d->col = (hevc_colinfo*)realloc(d->col, sizeof(hevc_colinfo) * d->cap_out);
  }
  d->cur = hevc_frame_alloc(sps->width, sps->height, sps->bit_depth);
  d->cur_idx = d->n_out; d->out[d->n_out] = d->cur; memset(&d->col[d->n_out], 0, sizeof(hevc_colinfo)); d->col[d->n_out].poc = h->poc; d->n_out++;
  if (!d->meta || d->meta->w != sps->width || d->meta->h != sps->height || d->meta->log2_ctb != sps->log2_ctb) { hevc_meta_free(d->meta); d->meta = hevc_meta_alloc(sps->width, sps->height, sps->log2_ctb); }
  hevc_meta_reset(d->meta);
  d->meta->constrained_intra_pred = pps->constrained_intra_pred; d->meta->cb_qp_offset = pps->cb_qp_offset; d->meta->cr_qp_offset = pps->cr_qp_offset;
  d->meta->strong_intra_smoothing = sps->strong_intra_smoothing;
  d->pic_open = 1; d->slice_idx_in_pic = 0; d->have_last_sh = 0;
  return 0;
}

static int decode_slice(oracle_hevc_decoder* d, const uint8_t* rbsp, size_t n, int nal_type) {
  bitreader b = {rbsp, n, 16};
  sdec* s = (sdec*)calloc(1, sizeof(sdec)); hevc_sps* sps; hevc_pps* pps;
  if (parse_slice_header(d, &b, nal_type, &s->sh, &sps, &pps)) { free(s); return -1; }
  s->d = d; s->sps = sps; s->pps = pps;
  if (s->sh.first_slice_in_pic) { if (start_picture(d, sps, pps, &s->sh)) { free(s); return -1; } }
  else if (!d->pic_open) { DEC_ERR("slice without picture start"); free(s); return -1; }
  else { s->sh.poc = d->cur_poc; }
  s->f = d->cur; s->m = d->meta;
  hevc_meta* m = s->m;
  if (s->sh.dependent && s->sh.segment_addr == 0) { DEC_ERR("dependent slice segment at CTB 0"); free(s); return -1; }
  hevc_slice_meta dummy_sm;
  hevc_slice_meta* sm;
  if (s->sh.dependent) { s->slice_idx = d->last_slice_idx; sm = &dummy_sm; }   /* same slice: availability, loop filter and reference lists of its independent segment */
  else {
    if (m->n_slices >= 1024) { free(s); return -1; }
    s->slice_idx = m->n_slices++;
    sm = &m->slices[s->slice_idx];
    d->last_sh = s->sh; d->last_slice_idx = s->slice_idx; d->have_last_sh = 1;
  }
  memset(sm, 0, sizeof(*sm));
  sm->deblocking_disabled = (uint8_t)s->sh.deblocking_disabled; sm->loop_filter_across = (uint8_t)s->sh.loop_filter_across_slices;
  sm->sao_luma = (uint8_t)s->sh.sao_luma; sm->sao_chroma = (uint8_t)s->sh.sao_chroma;
  sm->beta_offset_div2 = (int8_t)s->sh.beta_offset_div2; sm->tc_offset_div2 = (int8_t)s->sh.tc_offset_div2; sm->slice_type = (int8_t)s->sh.slice_type;
  /* reference picture list 0 (8.3.2, 8.3.4): StCurrBefore then StCurrAfter, cyclic */
  if (s->sh.slice_type == SLICE_P) {
    int cand[16], nc = 0;
    for (int i = 0; i < s->sh.rps_num; i++) if (s->sh.rps_used[i]) cand[nc++] = d->cur_poc + s->sh.rps_delta[i];
    if (!nc) { DEC_ERR("P slice with empty reference picture set"); free(s); return -1; }
    for (int i = 0; i < s->sh.num_ref_idx[0] && i < 16; i++) {
      int poc = cand[i % nc]; s->ref_poc[i] = poc; sm->ref_poc[i] = poc; s->ref[i] = NULL;
      for (int k = d->n_out - 2; k >= 0; k--) if (d->col[k].poc == poc) { s->ref[i] = d->out[k]; s->refcol[i] = &d->col[k]; break; }
      if (!s->ref[i]) { DEC_ERR("missing reference picture POC %d", poc); free(s); return -1; }
    }
  }
  int init_type = s->sh.slice_type == SLICE_I ? 0 : (s->sh.cabac_init_flag ? 2 : 1);
  /* 9.3.1: a dependent slice segment goes on with the context variables (and QpY predictor, 8.6.1) the previous segment ended with; the wavefront rule
   * at the start of a CTB row (below) overrides this */
  if (s->sh.dependent) { memcpy(s->c.st, d->ds_ctx, CTX_COUNT); s->qp_y = d->ds_qp_y; }
  else { cabac_init_ctx(s->c.st, init_type, s->sh.qp); s->qp_y = s->sh.qp; }
  cabac_start(&s->c, rbsp, n, s->sh.data_bit_offset);
  s->qp_pred = s->sh.qp;
  s->wp_on = pps->weighted_pred && s->sh.slice_type == SLICE_P;
  s->mp.m = m; s->mp.max_merge_cand = s->sh.max_merge_cand; s->mp.num_ref_idx = s->sh.num_ref_idx[0]; s->mp.ref_poc = s->ref_poc;
  s->mp.cur_poc = d->cur_poc; s->mp.col = (s->sh.temporal_mvp && s->sh.slice_type == SLICE_P) ? s->refcol[s->sh.collocated_ref_idx] : NULL;
  s->mp.log2_ctb = sps->log2_ctb; s->mp.pic_w = sps->width; s->mp.pic_h = sps->height;
  int ctb_addr = s->sh.segment_addr, n_ctb = sps->pic_w_ctb * sps->pic_h_ctb, end = 0;
  while (!end) {
    if (ctb_addr >= n_ctb) { DEC_ERR("slice data runs past the picture"); s->error = 1; break; }
    int rx = ctb_addr % sps->pic_w_ctb, ry = ctb_addr / sps->pic_w_ctb;
    m->ctb_slice[ctb_addr] = (uint16_t)s->slice_idx;
    if (pps->entropy_coding_sync && rx == 0) {
      /* first CTB of a row (9.3.1): the context variables of the CTB above-right after it was parsed, when that CTB is available (6.4.1: inside the
       * picture and in the same slice), the initial ones otherwise; QpY prediction restarts from SliceQpY (8.6.1) */
      cabac_init_ctx(s->c.st, init_type, s->sh.qp);
      if (ry > 0 && sps->pic_w_ctb > 1 && m->ctb_slice[ctb_addr - sps->pic_w_ctb + 1] == s->slice_idx) memcpy(s->c.st, d->wpp_ctx, CTX_COUNT);
      s->qp_y = s->sh.qp;
    }
    parse_sao(s, rx, ry);
    coding_quadtree(s, rx << sps->log2_ctb, ry << sps->log2_ctb, sps->log2_ctb, 0);
    if (s->error) break;
    if (pps->entropy_coding_sync && rx == 1) memcpy(d->wpp_ctx, s->c.st, CTX_COUNT);   /* storage process after the second CTB of a row */
    end = cabac_terminate(&s->c);
    ctb_addr++;
    if (!end && pps->entropy_coding_sync && ctb_addr % sps->pic_w_ctb == 0) {
      /* end_of_subset_one_bit, byte_alignment(): the bit that ended the arithmetic codeword is the alignment bit (9.3.2.5); the next CTB row is
       * its own arithmetic codeword starting at the next byte */
      if (!cabac_terminate(&s->c)) { DEC_ERR("end_of_subset_one_bit is 0"); s->error = 1; break; }
      cabac_start(&s->c, rbsp, n, (s->c.bitpos + 7) & ~(size_t)7);
    }
    if (s->c.bitpos > n * 8 + 64) { DEC_ERR("CABAC read past the end of slice data"); s->error = 1; break; }
  }
  memcpy(d->ds_ctx, s->c.st, CTX_COUNT); d->ds_qp_y = s->qp_y;   /* storage process at the end of a slice segment (dependent_slice_segments_enabled_flag) */
  int err = s->error; free(s);
  if (err) { DEC_ERR("slice decode failed (ctb %d)", ctb_addr); return -1; }
  return 0;
}

static void parse_sei(oracle_hevc_decoder* d, const uint8_t* rbsp, size_t n) {
  size_t p = 2;
  while (p + 2 <= n) {
    int type = 0, size = 0;
    while (p < n && rbsp[p] == 0xFF) { type += 255; p++; } if (p >= n) return; type += rbsp[p++];
    while (p < n && rbsp[p] == 0xFF) { size += 255; p++; } if (p >= n) return; size += rbsp[p++];
    if (p + size > n) return;
    if (type == 132 && size >= 49 && rbsp[p] == 0) { for (int c = 0; c < 3; c++) memcpy(d->pending_md5[c], rbsp + p + 1 + 16 * c, 16); d->have_md5 = 1; }
    p += size;
    if (p < n && rbsp[p] == 0x80) break;
  }
}

/* ================================================================================================ API */
/* every slice segment header of an Annex-B stream as the oracle's parser reads it, 19 ints per slice (tests/test_slice_headers.py pins them against the
 * reference's TDecCavlc::parseSliceHeader, tests/golden/slices_*.json). Returns the slice count, -1 on a parse error. */
int oracle_slice_headers(const uint8_t* p, size_t n, int* out, int cap) {
  oracle_hevc_decoder* d = (oracle_hevc_decoder*)calloc(1, sizeof(*d));
  int k = 0; size_t i = 0;
  while (i + 3 < n) {
    if (!(p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 1)) { i++; continue; }
    size_t st = i + 3, e = st;
    while (e + 2 < n && !(p[e] == 0 && p[e + 1] == 0 && (p[e + 2] == 1 || (p[e + 2] == 0 && e + 3 < n && p[e + 3] == 1)))) e++;
    if (e + 2 >= n) e = n;
    int type = (p[st] >> 1) & 63;
    uint8_t* rb = (uint8_t*)malloc(e - st + 1); size_t m = 0; int z = 0;
    for (size_t q = st + 2; q < e; q++) { if (z >= 2 && p[q] == 3) { z = 0; continue; } z = p[q] == 0 ? z + 1 : 0; rb[m++] = p[q]; }
    bitreader b = {rb, m, 0};
    int rc = 0;
    if (type == NAL_SPS) rc = parse_sps(d, &b);
    else if (type == NAL_PPS) rc = parse_pps(d, &b);
    else if (type < 32) {
      hevc_slice_hdr h; hevc_sps* sps; hevc_pps* pps; memset(&h, 0, sizeof(h));
      rc = parse_slice_header(d, &b, type, &h, &sps, &pps);
      if (!rc && !h.dependent) {   /* PicOrderCntVal (8.3.1), as start_picture derives it; a dependent segment repeats its slice's */
        if (type == NAL_IDR_W_RADL || type == NAL_IDR_N_LP) h.poc = 0;
        else { int max_lsb = 1 << sps->log2_max_poc_lsb, prev_lsb = d->prev_tid0_poc & (max_lsb - 1), prev_msb = d->prev_tid0_poc - prev_lsb, msb = prev_msb;
          if (h.poc_lsb < prev_lsb && prev_lsb - h.poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb; else if (h.poc_lsb > prev_lsb && h.poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
          h.poc = msb + h.poc_lsb; }
      }
      if (!rc && k < cap) { int* o = out + 64 * k; int intra = h.slice_type == SLICE_I; o[18] = h.dependent;
        o[0] = type; o[1] = h.segment_addr; o[2] = h.slice_type; o[3] = h.poc; o[4] = h.temporal_mvp; o[5] = h.sao_luma; o[6] = h.sao_chroma; o[7] = intra ? 0 : h.num_ref_idx[0];
        o[8] = h.cabac_init_flag; o[9] = intra ? 0 : h.collocated_ref_idx; o[10] = intra ? 0 : h.max_merge_cand; o[11] = h.qp; o[12] = h.cb_qp_offset; o[13] = h.cr_qp_offset;
        o[14] = h.deblocking_disabled; o[15] = h.beta_offset_div2; o[16] = h.tc_offset_div2; o[17] = h.loop_filter_across_slices;
        { const int has_rps = type != NAL_IDR_W_RADL && type != NAL_IDR_N_LP; o[19] = has_rps ? h.rps_num : 0; for (int q = 0; q < 4; q++) { o[20 + 2 * q] = has_rps && q < h.rps_num ? h.rps_delta[q] : 0; o[21 + 2 * q] = has_rps && q < h.rps_num ? h.rps_used[q] : 0; } }
        { const int wp = !intra && pps->weighted_pred; o[28] = wp; o[29] = wp ? h.wp_luma_denom : 0; o[30] = wp ? h.wp_chroma_denom : 0; o[31] = 0;      /* pred_weight_table: per RefPicList0 entry (up to 4) flags, then weight and offset per component */
          for (int q = 0; q < 4; q++) { const int on = wp && q < h.num_ref_idx[0]; int* e = o + 32 + 8 * q; e[0] = on ? h.wp_luma_flag[q] : 0; e[1] = on ? h.wp_chroma_flag[q] : 0;
            for (int c = 0; c < 3; c++) { e[2 + 2 * c] = on ? h.wp_w[q][c] : 0; e[3 + 2 * c] = on ? h.wp_o[q][c] : 0; } } } }
      if (!rc) { k++; if (!nal_keeps_poc_anchor(type)) d->prev_tid0_poc = h.poc; if (!h.dependent) { d->last_sh = h; d->have_last_sh = 1; } }
    }
    free(rb);
    if (rc) { free(d); return -1; }
    i = e;
  }
  free(d);
  return k;
}
/* log2_max_poc_lsb, log2_ctb, sao, tmvp, num_st_rps of the first SPS of the stream (what the reference harness needs to be told, oracle/ref_harness.cpp) */
int oracle_sps_fields(const uint8_t* p, size_t n, int out[5]) {
  oracle_hevc_decoder* d = (oracle_hevc_decoder*)calloc(1, sizeof(*d));
  int rc = -1;
  for (size_t i = 0; i + 4 < n && rc; i++) if (p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 1 && ((p[i + 3] >> 1) & 63) == NAL_SPS) {
    size_t st = i + 3, e = st; while (e + 2 < n && !(p[e] == 0 && p[e + 1] == 0 && p[e + 2] <= 1)) e++;
    if (e + 2 >= n) e = n;
    uint8_t* rb = (uint8_t*)malloc(e - st + 1); size_t m = 0; int z = 0;
    for (size_t q = st + 2; q < e; q++) { if (z >= 2 && p[q] == 3) { z = 0; continue; } z = p[q] == 0 ? z + 1 : 0; rb[m++] = p[q]; }
    bitreader b = {rb, m, 0};
    if (!parse_sps(d, &b)) for (int k = 0; k < 16; k++) if (d->sps[k].valid) { hevc_sps* s = &d->sps[k]; out[0] = s->log2_max_poc_lsb; out[1] = s->log2_ctb; out[2] = s->sao_enabled; out[3] = s->temporal_mvp_enabled; out[4] = s->num_st_rps; rc = 0; break; }
    free(rb);
  }
  free(d);
  return rc;
}
oracle_hevc_decoder* oracle_hevc_dec_create(void) { oracle_hevc_decoder* d = (oracle_hevc_decoder*)calloc(1, sizeof(*d)); build_scans(d); return d; }
void oracle_hevc_dec_destroy(oracle_hevc_decoder* d) {
  if (!d) return;
  for (int i = 0; i < d->n_out; i++) { hevc_frame_free(d->out[i]); free(d->col[i].mv); free(d->col[i].refpoc); free(d->col[i].imode); }
  free(d->out); free(d->col); hevc_meta_free(d->meta); free(d);
}
int oracle_hevc_dec_decode(oracle_hevc_decoder* d, const uint8_t* p, size_t n) {
  bytebuf rb = {0, 0, 0}; int sc; int rc = 0;
  size_t pos = find_start(p, n, 0, &sc);
  while (pos < n) {
    size_t ns = pos + 3, next = find_start(p, n, ns, &sc), ne = next;
    while (ne > ns && p[ne - 1] == 0) ne--;     /* trailing_zero_8bits / 4-byte start code prefix */
    if (ne - ns >= 2) {
      int type = (p[ns] >> 1) & 0x3F;
      unescape(p + ns, ne - ns, &rb);
      bitreader b = {rb.d, rb.n, 16};
      if (type == NAL_SPS) { if (parse_sps(d, &b)) { rc = -1; break; } }
      else if (type == NAL_PPS) { if (parse_pps(d, &b)) { rc = -1; break; } }
      else if (type == NAL_SEI_SUFFIX) parse_sei(d, rb.d, rb.n);
      else if (type <= NAL_TRAIL_R || (type >= 16 && type <= 21)) { if (decode_slice(d, rb.d, rb.n, type)) { rc = -1; break; } }
      else if (type >= 2 && type <= 9) { DEC_ERR("unsupported VCL NAL type %d", type); rc = -1; break; }
    }
    pos = next;
  }
  finish_picture(d);
  free(rb.d);
  d->error = rc;
  return rc;
}
int oracle_hevc_dec_num_frames(const oracle_hevc_decoder* d) { return d->n_out; }
const hevc_frame* oracle_hevc_dec_frame(const oracle_hevc_decoder* d, int i) { return i >= 0 && i < d->n_out ? d->out[i] : NULL; }
/* per 4x4 unit of decoded picture i (row stride = ceil(width / 4)): the luma intra prediction mode the stream codes there, 255 where the unit is not intra coded */
const uint8_t* oracle_hevc_dec_imodes(const oracle_hevc_decoder* d, int i) { return i >= 0 && i < d->n_out ? d->col[i].imode : NULL; }
/* conformance window (luma samples: left, right, top, bottom) of the SPS the last picture used */
void oracle_hevc_dec_crop(const oracle_hevc_decoder* d, int out[4]) { for (int i = 0; i < 4; i++) out[i] = 2 * d->last_conf_win[i]; }
int oracle_hevc_dec_md5_checked(const oracle_hevc_decoder* d) { return d->md5_checked; }
int oracle_hevc_dec_md5_failed(const oracle_hevc_decoder* d) { return d->md5_failed; }
