// Host-side HEVC high-level syntax: Annex-B NAL splitting, emulation-prevention removal, VPS/SPS/PPS and slice
// segment header parsing (H.265 7.3.1-7.3.6), POC and reference picture list derivation (8.3), plus the writer side
// the encoder uses. Tiny serial work per GOF; everything below the slice header runs on the GPU.
// Replaces what avformat_open_input / av_read_frame / the hevc parser do for the reference (PCCTranscoder.cpp:755-823, :428).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../csrc/rbt_types.h"

namespace rbt {

enum { NAL_TRAIL_N = 0, NAL_TRAIL_R = 1, NAL_IDR_W_RADL = 19, NAL_IDR_N_LP = 20, NAL_CRA = 21, NAL_VPS = 32, NAL_SPS = 33, NAL_PPS = 34,
       NAL_AUD = 35, NAL_SEI_PREFIX = 39, NAL_SEI_SUFFIX = 40 };

struct BitReader {
  const uint8_t* d; size_t n; size_t pos;
  int bit() { if ((pos >> 3) >= n) { pos++; return 0; } int v = (d[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return v; }
  uint32_t u(int k) { uint32_t v = 0; while (k--) v = (v << 1) | bit(); return v; }
  uint32_t ue() { int z = 0; while (!bit() && z < 32) z++; return z ? ((1u << z) - 1 + u(z)) : 0; }
  int32_t se() { uint32_t k = ue(); return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1); }
  bool aligned() const { return (pos & 7) == 0; }
};
struct BitWriter {
  std::vector<uint8_t> b; uint32_t acc = 0; int nacc = 0;
  void bit(int v) { acc = (acc << 1) | (v & 1); if (++nacc == 8) { b.push_back((uint8_t)acc); acc = 0; nacc = 0; } }
  void u(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bit((v >> i) & 1); }
  void ue(uint32_t v) { uint32_t k = v + 1; int len = 0; while ((k >> len) > 1) len++; u(0, len); u(k, len + 1); }
  void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
  void trailing() { bit(1); while (nacc) bit(0); }
  void align_zero() { while (nacc) bit(0); }
};

struct Rps { int num_neg = 0, num_pos = 0, num = 0; int delta_poc[16] = {0}; int used[16] = {0}; };

struct Sps {
  bool valid = false;
  int sps_id = 0, chroma_format_idc = 1, width = 0, height = 0, conf_win[4] = {0, 0, 0, 0};
  int bit_depth = 8, log2_max_poc_lsb = 4, max_dec_pic_buffering = 1;
  int log2_min_cb = 3, log2_diff_max_min_cb = 0, log2_ctb = 3, log2_min_tb = 2, log2_diff_max_min_tb = 0, log2_max_tb = 2;
  int max_th_depth_inter = 0, max_th_depth_intra = 0;
  int amp = 0, sao = 0, temporal_mvp = 0, strong_intra = 0;
  int num_st_rps = 0; Rps st_rps[65];
  int w_ctb = 0, h_ctb = 0;
};
struct Pps {
  bool valid = false;
  int pps_id = 0, sps_id = 0, entropy_coding_sync = 0, dependent_slice_segments = 0, output_flag_present = 0, num_extra_slice_header_bits = 0;
  int sign_data_hiding = 0, cabac_init_present = 0, num_ref_idx_default = 1, init_qp = 26, constrained_intra_pred = 0, transform_skip = 0;
  int cu_qp_delta = 0, diff_cu_qp_delta_depth = 0, cb_qp_offset = 0, cr_qp_offset = 0, slice_chroma_qp_offsets_present = 0;
  int transquant_bypass = 0, loop_filter_across_slices = 0, deblocking_control_present = 0, deblocking_override_enabled = 0;
  int pps_deblocking_disabled = 0, beta_offset_div2 = 0, tc_offset_div2 = 0, lists_modification_present = 0, slice_header_extension_present = 0;
  int weighted_pred = 0;               // weighted_pred_flag: P slices carry a pred_weight_table (libx265 from preset "veryfast" up)
};
struct SliceHdr {
  int nal_type = 0, first_slice_in_pic = 0, pps_id = 0, segment_addr = 0, slice_type = RBT_SLICE_I, dependent = 0, num_entry_points = 0;
  int poc_lsb = 0, poc = 0; Rps rps; int temporal_mvp = 0, sao_luma = 0, sao_chroma = 0, num_ref_idx = 1, cabac_init_flag = 0;
  int collocated_ref_idx = 0, max_merge_cand = 5, qp = 26, cb_qp_offset = 0, cr_qp_offset = 0;
  int deblocking_disabled = 0, beta_offset_div2 = 0, tc_offset_div2 = 0, lf_across = 0;
  // pred_weight_table (7.3.6.3 / 7.4.7.3) of a P slice under weighted_pred_flag: per RefPicList0 entry the flags, LumaWeightL0 / ChromaWeightL0 and luma_offset_l0 / ChromaOffsetL0
  int wp_on = 0, wp_luma_denom = 0, wp_chroma_denom = 0, wp_luma_flag[RBT_MAX_REFS] = {}, wp_chroma_flag[RBT_MAX_REFS] = {}, wp_w[RBT_MAX_REFS][3] = {}, wp_o[RBT_MAX_REFS][3] = {};
  size_t data_byte_offset = 0;      // of slice_segment_data() inside the RBSP
  std::vector<uint32_t> entry_sizes; // entry_point_offset_minus1[i] + 1: bytes of substream i as sent (emulation prevention bytes included)
};

struct Nal { int type; size_t rbsp_off, rbsp_size; std::vector<uint32_t> epb; };   // inside the unescaped batch buffer; epb: where emulation prevention bytes were
                                                                                    // removed, as offsets into the NAL unit AS SENT (entry point offsets count them, 7.4.7.1)

// Splits an Annex-B stream and appends the unescaped NAL units (2-byte header included) to `rbsp`.
void split_annexb(const uint8_t* p, size_t n, std::vector<uint8_t>& rbsp, std::vector<Nal>& nals);

struct ParamSets { Sps sps[16]; Pps pps[64]; };
int parse_sps(ParamSets& ps, const uint8_t* rbsp, size_t n, std::string& err);   // rbsp starts at the NAL header
int parse_pps(ParamSets& ps, const uint8_t* rbsp, size_t n, std::string& err);
// `head`: the header of the slice's independent segment (what a dependent slice segment repeats), nullptr when there is none yet
int parse_slice_header(ParamSets& ps, const uint8_t* rbsp, size_t n, int nal_type, SliceHdr& h, std::string& err, const SliceHdr* head = nullptr);
bool parse_md5_sei(const uint8_t* rbsp, size_t n, uint8_t md5[3][16]);
// PicOrderCntVal of a picture (8.3.1) from slice_pic_order_cnt_lsb; `prev_tid0_poc` is the POC anchor (prevTid0Pic), moved on unless the picture is a RASL / RADL
// or sub-layer non-reference picture (the even NAL types up to 14: HM codes the P pictures of the CTC structure as TRAIL_N, cfg/hm/ctc-hm-geometry-ai.cfg:29)
int slice_poc(const Sps& s, int nal_type, int poc_lsb, int& prev_tid0_poc);

void fill_stream_cfg(const Sps& s, const Pps& p, RbtStreamCfg& c);

// ---- writer side (encoder) ----
void append_nal(std::vector<uint8_t>& out, int type, const uint8_t* rbsp, size_t n, bool long_start_code);
void write_param_sets(std::vector<uint8_t>& out, const Sps& s, const Pps& p);
// slice segment header up to and including byte_alignment(); is_idr pictures carry no POC/RPS syntax
void write_slice_header(BitWriter& w, const Sps& s, const Pps& p, const SliceHdr& h, bool is_idr, int st_rps_idx);

void md5_plane_u16(const uint16_t* p, int w, int h, int bit_depth, uint8_t out[16]);
struct Md5PlaneJob { const uint16_t* p; int w, h, bit_depth; uint8_t* out; };
void md5_planes_u16(const Md5PlaneJob* jobs, size_t n);   // the same for many planes, one chain per host thread

}  // namespace rbt
