// See rbt_hls.h. H.265 7.3.1.1 (NAL), 7.3.2.x (parameter sets), 7.3.6 (slice segment header), D.2.19 (picture hash).
#include <cstring>
#include "rbt_hls.h"
#include <cmath>
#include <algorithm>
#include <atomic>
#include <thread>

namespace rbt {

// Start codes and emulation-prevention bytes both begin with 00 00, so the scan jumps from one zero byte to the next
// with memchr and copies the spans in between with one insert each (the byte loop cost ~1.3 ms per MB on the critical path).
void split_annexb(const uint8_t* p, size_t n, std::vector<uint8_t>& rbsp, std::vector<Nal>& nals) {
  auto find = [&](size_t from) {
    while (from + 3 <= n) {
      const uint8_t* z = (const uint8_t*)memchr(p + from, 0, n - from - 2);
      if (!z) return n;
      size_t i = (size_t)(z - p);
      if (p[i + 1] == 0 && p[i + 2] == 1) return i;
      from = i + 1;
    }
    return n;
  };
  rbsp.reserve(rbsp.size() + n);
  size_t pos = find(0);
  while (pos < n) {
    size_t ns = pos + 3, next = find(ns), ne = next;
    while (ne > ns && p[ne - 1] == 0) ne--;
    if (ne - ns >= 2) {
      Nal nal; nal.type = (p[ns] >> 1) & 0x3F; nal.rbsp_off = rbsp.size();
      size_t i = ns;
      while (i < ne) {
        // next 00 00 03 inside [i, ne): everything before its 03 is payload
        size_t j = i, cut = ne;
        while (j + 2 < ne) {
          const uint8_t* z = (const uint8_t*)memchr(p + j, 0, ne - j - 2);
          if (!z) break;
          size_t k = (size_t)(z - p);
          if (p[k + 1] == 0 && p[k + 2] == 3) { cut = k + 2; break; }
          j = k + 1;
        }
        rbsp.insert(rbsp.end(), p + i, p + cut);
        if (cut < ne) nal.epb.push_back((uint32_t)(cut - ns));
        i = cut < ne ? cut + 1 : ne;                  // skip the emulation prevention byte
      }
      nal.rbsp_size = rbsp.size() - nal.rbsp_off;
      nals.push_back(nal);
    }
    pos = next;
  }
}

static void skip_ptl(BitReader& b, int msl) {
  b.u(8); b.u(32); b.u(4); b.u(32); b.u(11); b.u(1); b.u(8);
  int pp[8], lp[8];
  for (int i = 0; i < msl; i++) { pp[i] = b.bit(); lp[i] = b.bit(); }
  if (msl > 0) for (int i = msl; i < 8; i++) b.u(2);
  for (int i = 0; i < msl; i++) { if (pp[i]) { b.u(32); b.u(32); b.u(24); } if (lp[i]) b.u(8); }
}
static int ceil_log2(unsigned v) { int n = 0; while ((1u << n) < v) n++; return n; }

static int parse_st_rps(BitReader& b, Sps& s, int idx, bool in_slice_header) {
  Rps& r = s.st_rps[idx];
  int inter = idx ? b.bit() : 0;
  if (inter) {
    int delta_idx = in_slice_header ? (int)b.ue() + 1 : 1;
    int ref = idx - delta_idx; if (ref < 0) return -1;
    int sign = b.bit(), absd = (int)b.ue() + 1, drps = (1 - 2 * sign) * absd;
    const Rps& q = s.st_rps[ref];
    int used[33], use_delta[33];
    for (int j = 0; j <= q.num; j++) { used[j] = b.bit(); use_delta[j] = 1; if (!used[j]) use_delta[j] = b.bit(); }
    int dp[40], du[40], i = 0;
    for (int j = q.num_pos - 1; j >= 0; j--) { int v = q.delta_poc[q.num_neg + j] + drps; if (v < 0 && use_delta[q.num_neg + j]) { dp[i] = v; du[i++] = used[q.num_neg + j]; } }
    if (drps < 0 && use_delta[q.num]) { dp[i] = drps; du[i++] = used[q.num]; }
    for (int j = 0; j < q.num_neg; j++) { int v = q.delta_poc[j] + drps; if (v < 0 && use_delta[j]) { dp[i] = v; du[i++] = used[j]; } }
    int nn = i;
    for (int j = q.num_neg - 1; j >= 0; j--) { int v = q.delta_poc[j] + drps; if (v > 0 && use_delta[j]) { dp[i] = v; du[i++] = used[j]; } }
    if (drps > 0 && use_delta[q.num]) { dp[i] = drps; du[i++] = used[q.num]; }
    for (int j = 0; j < q.num_pos; j++) { int v = q.delta_poc[q.num_neg + j] + drps; if (v > 0 && use_delta[q.num_neg + j]) { dp[i] = v; du[i++] = used[q.num_neg + j]; } }
    if (i > 16) return -1;
    r.num_neg = nn; r.num_pos = i - nn; r.num = i;
    for (int k = 0; k < i; k++) { r.delta_poc[k] = dp[k]; r.used[k] = du[k]; }
  } else {
    int nn = (int)b.ue(), np = (int)b.ue();
    if (nn + np > 16) return -1;
    int poc = 0;
    for (int i = 0; i < nn; i++) { poc -= (int)b.ue() + 1; r.delta_poc[i] = poc; r.used[i] = b.bit(); }
    poc = 0;
    for (int i = 0; i < np; i++) { poc += (int)b.ue() + 1; r.delta_poc[nn + i] = poc; r.used[nn + i] = b.bit(); }
    r.num_neg = nn; r.num_pos = np; r.num = nn + np;
  }
  return 0;
}

int parse_sps(ParamSets& ps, const uint8_t* rbsp, size_t n, std::string& err) {
  BitReader b{rbsp, n, 16};
  Sps s;
  b.u(4); int msl = b.u(3); b.bit();
  skip_ptl(b, msl);
  s.sps_id = b.ue(); if (s.sps_id > 15) { err = "sps id"; return -2; }
  s.chroma_format_idc = b.ue();
  if (s.chroma_format_idc != 1) { err = "only 4:2:0 is supported"; return -3; }
  s.width = b.ue(); s.height = b.ue();
  if (b.bit()) for (int i = 0; i < 4; i++) s.conf_win[i] = b.ue();
  s.bit_depth = 8 + b.ue(); int bdc = 8 + b.ue();
  if (s.bit_depth != bdc || s.bit_depth > 12) { err = "unsupported bit depth"; return -3; }
  s.log2_max_poc_lsb = 4 + b.ue();
  int sub_info = b.bit();
  for (int i = sub_info ? 0 : msl; i <= msl; i++) { s.max_dec_pic_buffering = b.ue() + 1; b.ue(); b.ue(); }
  s.log2_min_cb = 3 + b.ue(); s.log2_diff_max_min_cb = b.ue(); s.log2_ctb = s.log2_min_cb + s.log2_diff_max_min_cb;
  s.log2_min_tb = 2 + b.ue(); s.log2_diff_max_min_tb = b.ue(); s.log2_max_tb = s.log2_min_tb + s.log2_diff_max_min_tb;
  s.max_th_depth_inter = b.ue(); s.max_th_depth_intra = b.ue();
  if (b.bit()) { err = "scaling lists are not supported"; return -3; }
  s.amp = b.bit(); s.sao = b.bit();
  if (b.bit()) { err = "PCM is not supported"; return -3; }
  s.num_st_rps = b.ue(); if (s.num_st_rps > 64) { err = "num_short_term_ref_pic_sets"; return -2; }
  for (int i = 0; i < s.num_st_rps; i++) if (parse_st_rps(b, s, i, false)) { err = "short-term RPS"; return -2; }
  if (b.bit()) { err = "long-term reference pictures are not supported"; return -3; }
  s.temporal_mvp = b.bit(); s.strong_intra = b.bit();
  if (s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_max_tb > 5 || s.width <= 0 || s.height <= 0 || s.width > 8192 || s.height > 8192 ||
      (s.width & ((1 << s.log2_min_cb) - 1)) || (s.height & ((1 << s.log2_min_cb) - 1))) { err = "unsupported picture / block geometry"; return -3; }
  s.w_ctb = (s.width + (1 << s.log2_ctb) - 1) >> s.log2_ctb; s.h_ctb = (s.height + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
  s.valid = true; ps.sps[s.sps_id] = s;
  return 0;
}

int parse_pps(ParamSets& ps, const uint8_t* rbsp, size_t n, std::string& err) {
  BitReader b{rbsp, n, 16};
  Pps p;
  p.pps_id = b.ue(); p.sps_id = b.ue(); if (p.pps_id > 63 || p.sps_id > 15) { err = "pps id"; return -2; }
  p.dependent_slice_segments = b.bit(); p.output_flag_present = b.bit(); p.num_extra_slice_header_bits = b.u(3);
  p.sign_data_hiding = b.bit(); p.cabac_init_present = b.bit();
  p.num_ref_idx_default = b.ue() + 1; b.ue();
  p.init_qp = 26 + b.se(); p.constrained_intra_pred = b.bit(); p.transform_skip = b.bit();
  p.cu_qp_delta = b.bit(); if (p.cu_qp_delta) p.diff_cu_qp_delta_depth = b.ue();
  p.cb_qp_offset = b.se(); p.cr_qp_offset = b.se(); p.slice_chroma_qp_offsets_present = b.bit();
  p.weighted_pred = b.bit(); b.bit();                  // weighted_bipred_flag concerns B slices, which are refused where they appear
  p.transquant_bypass = b.bit(); int tiles = b.bit(); p.entropy_coding_sync = b.bit();
  if (tiles) { err = "tiles are not supported"; return -3; }
  p.loop_filter_across_slices = b.bit(); p.deblocking_control_present = b.bit();
  if (p.deblocking_control_present) {
    p.deblocking_override_enabled = b.bit(); p.pps_deblocking_disabled = b.bit();
    if (!p.pps_deblocking_disabled) { p.beta_offset_div2 = b.se(); p.tc_offset_div2 = b.se(); }
  }
  if (b.bit()) { err = "scaling lists are not supported"; return -3; }
  p.lists_modification_present = b.bit();
  if (b.ue() != 0) { err = "parallel merge level > 2 is not supported"; return -3; }
  p.slice_header_extension_present = b.bit();
  p.valid = true; ps.pps[p.pps_id] = p;
  return 0;
}

int slice_poc(const Sps& s, int nal_type, int poc_lsb, int& prev_tid0_poc) {
  int poc = 0;
  if (nal_type != NAL_IDR_W_RADL && nal_type != NAL_IDR_N_LP) {
    const int max_lsb = 1 << s.log2_max_poc_lsb, prev_lsb = prev_tid0_poc & (max_lsb - 1), prev_msb = prev_tid0_poc - prev_lsb; int msb = prev_msb;
    if (poc_lsb < prev_lsb && prev_lsb - poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (poc_lsb > prev_lsb && poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    if (nal_type >= 16 && nal_type <= 18) msb = 0;                                      // BLA pictures
    poc = msb + poc_lsb;
  }
  const bool leading_or_slnr = (nal_type <= 14 && (nal_type & 1) == 0) || (nal_type >= 6 && nal_type <= 9);
  if (!leading_or_slnr) prev_tid0_poc = poc;
  return poc;
}

int parse_slice_header(ParamSets& ps, const uint8_t* rbsp, size_t n, int nal_type, SliceHdr& h, std::string& err, const SliceHdr* head) {
  BitReader b{rbsp, n, 16};
  h = SliceHdr(); h.nal_type = nal_type;
  h.first_slice_in_pic = b.bit();
  if (nal_type >= 16 && nal_type <= 23) b.bit();
  h.pps_id = b.ue(); if (h.pps_id > 63 || !ps.pps[h.pps_id].valid) { err = "slice refers to a missing PPS"; return -2; }
  const Pps& p = ps.pps[h.pps_id]; Sps& s = ps.sps[p.sps_id];
  if (!s.valid) { err = "slice refers to a missing SPS"; return -2; }
  int dependent = 0;
  if (!h.first_slice_in_pic) { if (p.dependent_slice_segments) dependent = b.bit(); h.segment_addr = b.u(ceil_log2(s.w_ctb * s.h_ctb)); }
  if (dependent) {   // 7.3.6.1: everything up to the entry points is that of the slice's independent segment
    if (!head || h.segment_addr == 0) { err = "dependent slice segment without a slice"; return -2; }
    const int addr = h.segment_addr; h = *head; h.nal_type = nal_type; h.first_slice_in_pic = 0; h.segment_addr = addr; h.dependent = 1;
  } else {
  for (int i = 0; i < p.num_extra_slice_header_bits; i++) b.bit();
  h.slice_type = b.ue();
  if (h.slice_type == RBT_SLICE_B) { err = "B slices are not supported"; return -3; }
  if (h.slice_type > 2) { err = "slice_type"; return -2; }
  if (p.output_flag_present) b.bit();
  bool idr = nal_type == NAL_IDR_W_RADL || nal_type == NAL_IDR_N_LP;
  if (!idr) {
    h.poc_lsb = b.u(s.log2_max_poc_lsb);
    int sps_flag = b.bit(), ri;
    if (!sps_flag) { if (parse_st_rps(b, s, s.num_st_rps, true)) { err = "slice RPS"; return -2; } ri = s.num_st_rps; }
    else ri = s.num_st_rps > 1 ? (int)b.u(ceil_log2(s.num_st_rps)) : 0;
    h.rps = s.st_rps[ri];
    if (s.temporal_mvp) h.temporal_mvp = b.bit();
  }
  if (s.sao) { h.sao_luma = b.bit(); h.sao_chroma = b.bit(); }
  h.num_ref_idx = p.num_ref_idx_default;
  if (h.slice_type == RBT_SLICE_P) {
    if (b.bit()) h.num_ref_idx = b.ue() + 1;
    int ntot = 0; for (int i = 0; i < h.rps.num; i++) ntot += h.rps.used[i];
    if (p.lists_modification_present && ntot > 1 && b.bit()) { err = "ref_pic_lists_modification is not supported"; return -3; }
    if (p.cabac_init_present) h.cabac_init_flag = b.bit();
    if (h.temporal_mvp && h.num_ref_idx > 1) h.collocated_ref_idx = b.ue();
    if (p.weighted_pred) {   // pred_weight_table() 7.3.6.3; derivations 7.4.7.3 without high_precision_offsets (wpOffsetHalfRangeC = 128)
      if (h.num_ref_idx > RBT_MAX_REFS) { err = "slice header range"; return -2; }
      h.wp_on = 1;
      h.wp_luma_denom = (int)b.ue(); h.wp_chroma_denom = h.wp_luma_denom + b.se();
      if (h.wp_luma_denom > 7 || h.wp_chroma_denom < 0 || h.wp_chroma_denom > 7) { err = "pred_weight_table: weight denominator out of range"; return -2; }
      for (int i = 0; i < h.num_ref_idx; i++) h.wp_luma_flag[i] = b.bit();
      for (int i = 0; i < h.num_ref_idx; i++) h.wp_chroma_flag[i] = b.bit();
      for (int i = 0; i < h.num_ref_idx; i++) {
        h.wp_w[i][0] = 1 << h.wp_luma_denom; h.wp_w[i][1] = h.wp_w[i][2] = 1 << h.wp_chroma_denom;
        if (h.wp_luma_flag[i]) { const int dw = b.se(), o = b.se(); if (dw < -128 || dw > 127 || o < -128 || o > 127) { err = "pred_weight_table range"; return -2; } h.wp_w[i][0] += dw; h.wp_o[i][0] = o; }
        if (h.wp_chroma_flag[i]) for (int j = 1; j < 3; j++) {
          const int dw = b.se(), dof = b.se(); if (dw < -128 || dw > 127 || dof < -512 || dof > 511) { err = "pred_weight_table range"; return -2; }
          h.wp_w[i][j] += dw;
          h.wp_o[i][j] = std::min(127, std::max(-128, 128 + dof - ((128 * h.wp_w[i][j]) >> h.wp_chroma_denom)));
        }
      }
    }
    h.max_merge_cand = 5 - (int)b.ue();
    if (h.max_merge_cand < 1 || h.max_merge_cand > 5 || h.num_ref_idx > RBT_MAX_REFS || h.collocated_ref_idx >= h.num_ref_idx) { err = "slice header range"; return -2; }
  }
  h.qp = p.init_qp + b.se();
  if (p.slice_chroma_qp_offsets_present) { h.cb_qp_offset = b.se(); h.cr_qp_offset = b.se(); }
  h.deblocking_disabled = p.pps_deblocking_disabled; h.beta_offset_div2 = p.beta_offset_div2; h.tc_offset_div2 = p.tc_offset_div2;
  int ovr = p.deblocking_override_enabled ? b.bit() : 0;
  if (ovr) { h.deblocking_disabled = b.bit(); if (!h.deblocking_disabled) { h.beta_offset_div2 = b.se(); h.tc_offset_div2 = b.se(); } }
  h.lf_across = p.loop_filter_across_slices;
  if (p.loop_filter_across_slices && (h.sao_luma || h.sao_chroma || !h.deblocking_disabled)) h.lf_across = b.bit();
  }
  h.num_entry_points = 0; h.entry_sizes.clear();                 // (a dependent segment starts as a copy of its slice's first header: not its entry points)
  if (p.entropy_coding_sync) {   // the offsets let the decoder give every CTB row of the segment a wave of its own (rbt_decode.cpp)
    h.num_entry_points = (int)b.ue();
    if (h.num_entry_points > s.h_ctb) { err = "slice header: entry points"; return -2; }
    if (h.num_entry_points > 0) { const int len = (int)b.ue() + 1; if (len > 32) { err = "slice header: entry points"; return -2; } for (int i = 0; i < h.num_entry_points; i++) h.entry_sizes.push_back((uint32_t)b.u(len) + 1u); }
  }
  if (p.slice_header_extension_present) { int k = b.ue(); for (int i = 0; i < k; i++) b.u(8); }
  if (!b.bit()) { err = "slice header alignment"; return -2; }
  while (!b.aligned()) b.bit();
  if (b.pos / 8 > n) { err = "truncated slice header"; return -2; }
  h.data_byte_offset = b.pos / 8;
  return 0;
}

bool parse_md5_sei(const uint8_t* rbsp, size_t n, uint8_t md5[3][16]) {
  size_t p = 2;
  while (p + 2 <= n) {
    int type = 0, size = 0;
    while (p < n && rbsp[p] == 0xFF) { type += 255; p++; } if (p >= n) return false; type += rbsp[p++];
    while (p < n && rbsp[p] == 0xFF) { size += 255; p++; } if (p >= n) return false; size += rbsp[p++];
    if (p + size > n) return false;
    if (type == 132 && size >= 49 && rbsp[p] == 0) { for (int c = 0; c < 3; c++) memcpy(md5[c], rbsp + p + 1 + 16 * c, 16); return true; }
    p += size;
    if (p < n && rbsp[p] == 0x80) break;
  }
  return false;
}

void fill_stream_cfg(const Sps& s, const Pps& p, RbtStreamCfg& c) {
  memset(&c, 0, sizeof(c));
  c.w = s.width; c.h = s.height; c.cw = s.width / 2; c.ch = s.height / 2; c.w4 = (s.width + 3) / 4; c.h4 = (s.height + 3) / 4; c.w_ctb = s.w_ctb; c.h_ctb = s.h_ctb;
  c.bit_depth = (int8_t)s.bit_depth; c.log2_ctb = (int8_t)s.log2_ctb; c.log2_min_cb = (int8_t)s.log2_min_cb; c.log2_min_tb = (int8_t)s.log2_min_tb; c.log2_max_tb = (int8_t)s.log2_max_tb;
  c.th_depth_inter = (int8_t)s.max_th_depth_inter; c.th_depth_intra = (int8_t)s.max_th_depth_intra; c.diff_cu_qp_delta_depth = (int8_t)p.diff_cu_qp_delta_depth;
  c.amp = (uint8_t)s.amp; c.sao = (uint8_t)s.sao; c.strong_intra = (uint8_t)s.strong_intra; c.tmvp = (uint8_t)s.temporal_mvp; c.sign_hiding = (uint8_t)p.sign_data_hiding;
  c.cabac_init_present = (uint8_t)p.cabac_init_present; c.cip = (uint8_t)p.constrained_intra_pred; c.transform_skip = (uint8_t)p.transform_skip;
  c.cu_qp_delta = (uint8_t)p.cu_qp_delta; c.tq_bypass_enabled = (uint8_t)p.transquant_bypass; c.cb_qp_offset = (int8_t)p.cb_qp_offset; c.cr_qp_offset = (int8_t)p.cr_qp_offset;
}

// ------------------------------------------------------------------------------------------------ writer
void append_nal(std::vector<uint8_t>& out, int type, const uint8_t* rbsp, size_t n, bool long_start_code) {
  if (long_start_code) out.push_back(0);
  out.push_back(0); out.push_back(0); out.push_back(1);
  out.push_back((uint8_t)(type << 1)); out.push_back(1);
  // emulation prevention (7.4.2): 00 00 0x (x <= 3) gets a 03 in front of its third byte. Zero bytes are found with memchr
  // and the spans in between are appended in one go.
  size_t i = 0; int z = 0;                             // z: zero bytes at the end of what has been emitted (max 2 matter)
  while (i < n) {
    if (z >= 2 && rbsp[i] <= 3) { out.push_back(3); z = 0; }
    if (rbsp[i] == 0) { out.push_back(0); z++; i++; continue; }
    // non-zero byte: copy up to the next zero byte
    const uint8_t* q = (const uint8_t*)memchr(rbsp + i, 0, n - i);
    size_t e = q ? (size_t)(q - rbsp) : n;
    out.insert(out.end(), rbsp + i, rbsp + e);
    i = e; z = 0;
  }
}
static void write_ptl(BitWriter& w, int bit_depth) {
  int profile = bit_depth > 8 ? 2 : 1;
  w.u(0, 2); w.u(0, 1); w.u(profile, 5);
  for (int i = 0; i < 32; i++) w.bit(i == profile || (profile == 1 && i == 2));
  w.bit(1); w.bit(0); w.bit(0); w.bit(1);
  w.u(0, 32); w.u(0, 11); w.bit(0);
  w.u(153, 8);
}
void write_param_sets(std::vector<uint8_t>& out, const Sps& s, const Pps& p) {
  BitWriter v;
  v.u(0, 4); v.u(3, 2); v.u(0, 6); v.u(0, 3); v.bit(1); v.u(0xFFFF, 16);
  write_ptl(v, s.bit_depth);
  v.bit(1); v.ue(s.max_dec_pic_buffering - 1); v.ue(0); v.ue(0);
  v.u(0, 6); v.ue(0); v.bit(0); v.bit(0); v.trailing();
  append_nal(out, NAL_VPS, v.b.data(), v.b.size(), true);
  BitWriter w;
  w.u(0, 4); w.u(0, 3); w.bit(1);
  write_ptl(w, s.bit_depth);
  w.ue(0); w.ue(1); w.ue(s.width); w.ue(s.height);
  if (s.conf_win[0] | s.conf_win[1] | s.conf_win[2] | s.conf_win[3]) { w.bit(1); for (int i = 0; i < 4; i++) w.ue(s.conf_win[i]); } else w.bit(0);
  w.ue(s.bit_depth - 8); w.ue(s.bit_depth - 8); w.ue(s.log2_max_poc_lsb - 4);
  w.bit(1); w.ue(s.max_dec_pic_buffering - 1); w.ue(0); w.ue(0);
  w.ue(s.log2_min_cb - 3); w.ue(s.log2_diff_max_min_cb); w.ue(s.log2_min_tb - 2); w.ue(s.log2_diff_max_min_tb);
  w.ue(s.max_th_depth_inter); w.ue(s.max_th_depth_intra);
  w.bit(0); w.bit(s.amp); w.bit(s.sao); w.bit(0);
  w.ue(s.num_st_rps);
  for (int i = 0; i < s.num_st_rps; i++) { if (i) w.bit(0); w.ue(i + 1); w.ue(0); for (int k = 0; k <= i; k++) { w.ue(0); w.bit(1); } }
  w.bit(0); w.bit(s.temporal_mvp); w.bit(s.strong_intra);
  w.bit(0); w.bit(0); w.trailing();
  append_nal(out, NAL_SPS, w.b.data(), w.b.size(), true);
  BitWriter q;
  q.ue(0); q.ue(0); q.bit(p.dependent_slice_segments); q.bit(0); q.u(0, 3);
  q.bit(p.sign_data_hiding); q.bit(p.cabac_init_present);
  q.ue(p.num_ref_idx_default - 1); q.ue(0);
  q.se(p.init_qp - 26); q.bit(p.constrained_intra_pred); q.bit(p.transform_skip);
  q.bit(p.cu_qp_delta); if (p.cu_qp_delta) q.ue(p.diff_cu_qp_delta_depth);
  q.se(p.cb_qp_offset); q.se(p.cr_qp_offset); q.bit(p.slice_chroma_qp_offsets_present);
  q.bit(0); q.bit(0);
  q.bit(p.transquant_bypass); q.bit(0); q.bit(p.entropy_coding_sync);
  q.bit(p.loop_filter_across_slices);
  q.bit(p.deblocking_control_present);
  if (p.deblocking_control_present) {
    q.bit(p.deblocking_override_enabled); q.bit(p.pps_deblocking_disabled);
    if (!p.pps_deblocking_disabled) { q.se(p.beta_offset_div2); q.se(p.tc_offset_div2); }
  }
  q.bit(0); q.bit(0); q.ue(0); q.bit(0); q.bit(0); q.trailing();
  append_nal(out, NAL_PPS, q.b.data(), q.b.size(), true);
}
void write_slice_header(BitWriter& w, const Sps& s, const Pps& p, const SliceHdr& h, bool is_idr, int st_rps_idx) {
  w.bit(h.first_slice_in_pic);
  if (is_idr) w.bit(0);
  w.ue(0);
  if (!h.first_slice_in_pic) { if (p.dependent_slice_segments) w.bit(h.dependent); w.u(h.segment_addr, ceil_log2(s.w_ctb * s.h_ctb)); }
  if (h.dependent) {   // 7.3.6.1: nothing but the entry points (none: every segment this encoder writes is one CTB row) and the alignment
    if (p.entropy_coding_sync) w.ue(0);
    w.bit(1); w.align_zero();
    return;
  }
  w.ue(h.slice_type);
  if (!is_idr) {
    w.u(h.poc & ((1 << s.log2_max_poc_lsb) - 1), s.log2_max_poc_lsb);
    w.bit(1);
    if (s.num_st_rps > 1) w.u(st_rps_idx, ceil_log2(s.num_st_rps));
    if (s.temporal_mvp) w.bit(h.temporal_mvp);
  }
  if (s.sao) { w.bit(h.sao_luma); w.bit(h.sao_chroma); }
  if (h.slice_type == RBT_SLICE_P) {
    int ovr = h.num_ref_idx != p.num_ref_idx_default;
    w.bit(ovr); if (ovr) w.ue(h.num_ref_idx - 1);
    if (p.cabac_init_present) w.bit(h.cabac_init_flag);
    if (h.temporal_mvp && h.num_ref_idx > 1) w.ue(h.collocated_ref_idx);
    w.ue(5 - h.max_merge_cand);
  }
  w.se(h.qp - p.init_qp);
  if (p.slice_chroma_qp_offsets_present) { w.se(h.cb_qp_offset); w.se(h.cr_qp_offset); }
  if (p.deblocking_override_enabled) {
    int ovr = h.deblocking_disabled != p.pps_deblocking_disabled || h.beta_offset_div2 != p.beta_offset_div2 || h.tc_offset_div2 != p.tc_offset_div2;
    w.bit(ovr);
    if (ovr) { w.bit(h.deblocking_disabled); if (!h.deblocking_disabled) { w.se(h.beta_offset_div2); w.se(h.tc_offset_div2); } }
  }
  if (p.loop_filter_across_slices && (h.sao_luma || h.sao_chroma || !h.deblocking_disabled)) w.bit(h.lf_across);
  if (p.entropy_coding_sync) w.ue(0);   // num_entry_point_offsets
  w.bit(1); w.align_zero();
}

// ------------------------------------------------------------------------------------------------ MD5 (RFC 1321)
namespace {
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u; uint64_t len = 0; uint8_t buf[64]; int nbuf = 0;
  static const uint32_t* K() { static uint32_t k[64]; static bool init = false; if (!init) { for (int i = 0; i < 64; i++) k[i] = (uint32_t)std::floor(std::fabs(std::sin((double)(i + 1))) * 4294967296.0); init = true; } return k; }
  void block(const uint8_t* p) {
    static const uint8_t S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    const uint32_t* Kt = K(); uint32_t M[16];
    for (int i = 0; i < 16; i++) M[i] = p[4 * i] | (p[4 * i + 1] << 8) | (p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; i++) {
      uint32_t F; int g;
      if (i < 16) { F = (B & C) | (~B & D); g = i; } else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; } else { F = C ^ (B | ~D); g = (7 * i) & 15; }
      F = F + A + Kt[i] + M[g]; A = D; D = C; C = B; B = B + ((F << S[i]) | (F >> (32 - S[i])));
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) {
      if (nbuf == 0 && n >= 64) { block(p); p += 64; n -= 64; continue; }
      size_t k = 64 - nbuf; if (k > n) k = n;
      memcpy(buf + nbuf, p, k); nbuf += (int)k; p += k; n -= k;
      if (nbuf == 64) { block(buf); nbuf = 0; }
    }
  }
  void finish(uint8_t out[16]) {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1); pad = 0; while (nbuf != 56) update(&pad, 1);
    uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i)); update(l, 8);
    uint32_t v[4] = {a, b, c, d}; for (int i = 0; i < 16; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
  }
};
}  // namespace
void md5_plane_u16(const uint16_t* p, int w, int h, int bit_depth, uint8_t out[16]) {
  Md5 m;
  if (bit_depth <= 8) { std::vector<uint8_t> row(w); for (int y = 0; y < h; y++) { for (int x = 0; x < w; x++) row[x] = (uint8_t)p[(size_t)y * w + x]; m.update(row.data(), w); } }
  else m.update((const uint8_t*)p, (size_t)w * h * 2);   // little-endian host: samples are already 2 bytes LSB first
  m.finish(out);
}

// The hashes of many planes at once: MD5 is a serial chain of ~5 cycles per step (~0.6 GB/s on one core, whatever the code looks like), and a 32-frame GOF of 1280x1280
// maps is 630 MB of samples - one chain per plane on the host's cores instead (a 16-core share hashes the GOF in ~70 ms instead of 1.1 s).
void md5_planes_u16(const Md5PlaneJob* jobs, size_t n) {
  unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
  size_t total = 0; for (size_t i = 0; i < n; i++) total += (size_t)jobs[i].w * jobs[i].h;
  const size_t nt = total < (1u << 20) ? 1 : std::min<size_t>(std::min<size_t>(hw, 32), n);   // small pictures: not worth starting threads for
  if (nt <= 1) { for (size_t i = 0; i < n; i++) md5_plane_u16(jobs[i].p, jobs[i].w, jobs[i].h, jobs[i].bit_depth, jobs[i].out); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++) th.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n;) md5_plane_u16(jobs[i].p, jobs[i].w, jobs[i].h, jobs[i].bit_depth, jobs[i].out); });
  for (auto& t : th) t.join();
}

}  // namespace rbt
