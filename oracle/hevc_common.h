/* ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
 *
 * Shared types for the CPU restatement of the HEVC decode / encode the reference obtains from libavcodec
 * (decode: PCCTranscoder.cpp:428-448) and libx265 (encode: PCCTranscoder.cpp:548-592). Third-party algorithm:
 * ITU-T H.265 (v1 Main / Main10 tools used by the V-PCC CTC streams, cfg/hm/ctc-hm-*-ai.cfg).
 */
#ifndef ORACLE_HEVC_COMMON_H
#define ORACLE_HEVC_COMMON_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hevc_tables.h"

#define HEVC_MAX_W 8192
#define HEVC_MAX_H 8192

enum { NAL_TRAIL_N = 0, NAL_TRAIL_R = 1, NAL_IDR_W_RADL = 19, NAL_IDR_N_LP = 20, NAL_CRA = 21,
       NAL_VPS = 32, NAL_SPS = 33, NAL_PPS = 34, NAL_AUD = 35, NAL_EOS = 36, NAL_EOB = 37, NAL_FD = 38,
       NAL_SEI_PREFIX = 39, NAL_SEI_SUFFIX = 40 };
enum { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2 };
/* 8.3.1: prevTid0Pic is the previous picture with TemporalId 0 that is not a RASL / RADL picture or a sub-layer non-reference picture (the even types up to
 * RSV_VCL_N14, Table 7-1) - such pictures leave the POC anchor alone (HM marks the P pictures of the CTC structure TRAIL_N) */
static inline int nal_keeps_poc_anchor(int nal_type) { return (nal_type <= 14 && (nal_type & 1) == 0) || (nal_type >= 6 && nal_type <= 9); }
enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };
enum { MODE_INTER = 0, MODE_INTRA = 1, MODE_SKIP = 2 };

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }

/* ---------------------------------------------------------------- pictures */
typedef struct {
  int w, h;          /* luma size (cropped == coded in this restatement's encoder; decoder keeps coded size) */
  int cw, ch;        /* chroma size (4:2:0) */
  int bit_depth;
  uint16_t* p[3];    /* planes, stride == plane width */
} hevc_frame;

static inline hevc_frame* hevc_frame_alloc(int w, int h, int bit_depth) {
  hevc_frame* f = (hevc_frame*)calloc(1, sizeof(*f));
  f->w = w; f->h = h; f->cw = w / 2; f->ch = h / 2; f->bit_depth = bit_depth;
  f->p[0] = (uint16_t*)calloc((size_t)w * h, 2);
  f->p[1] = (uint16_t*)calloc((size_t)f->cw * f->ch, 2);
  f->p[2] = (uint16_t*)calloc((size_t)f->cw * f->ch, 2);
  return f;
}
static inline void hevc_frame_free(hevc_frame* f) {
  if (!f) return;
  free(f->p[0]); free(f->p[1]); free(f->p[2]); free(f);
}
static inline void hevc_frame_copy(hevc_frame* d, const hevc_frame* s) {
  memcpy(d->p[0], s->p[0], (size_t)s->w * s->h * 2);
  memcpy(d->p[1], s->p[1], (size_t)s->cw * s->ch * 2);
  memcpy(d->p[2], s->p[2], (size_t)s->cw * s->ch * 2);
}

/* ---------------------------------------------------------------- byte buffers */
typedef struct { uint8_t* d; size_t n, cap; } bytebuf;
static inline void bb_reserve(bytebuf* b, size_t extra) {
  if (b->n + extra > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->n + extra) nc *= 2;
    b->d = (uint8_t*)realloc(b->d, nc); b->cap = nc;
  }
}
static inline void bb_put(bytebuf* b, uint8_t v) { bb_reserve(b, 1); b->d[b->n++] = v; }
static inline void bb_append(bytebuf* b, const uint8_t* p, size_t n) { bb_reserve(b, n); memcpy(b->d + b->n, p, n); b->n += n; }

/* ---------------------------------------------------------------- RBSP bit reader (7.2) */
typedef struct { const uint8_t* d; size_t n; size_t pos; /* bit position */ } bitreader;
static inline int br_bit(bitreader* b) {
  if ((b->pos >> 3) >= b->n) { b->pos++; return 0; }
  int v = (b->d[b->pos >> 3] >> (7 - (b->pos & 7))) & 1; b->pos++; return v;
}
static inline uint32_t br_u(bitreader* b, int n) { uint32_t v = 0; while (n--) v = (v << 1) | br_bit(b); return v; }
static inline uint32_t br_ue(bitreader* b) {
  int z = 0; while (!br_bit(b) && z < 32) z++;
  return z ? ((1u << z) - 1 + br_u(b, z)) : 0;
}
static inline int32_t br_se(bitreader* b) { uint32_t k = br_ue(b); return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1); }
static inline int br_aligned(bitreader* b) { return (b->pos & 7) == 0; }
static inline size_t br_bits_left(bitreader* b) { return b->n * 8 > b->pos ? b->n * 8 - b->pos : 0; }

/* ---------------------------------------------------------------- bit writer */
typedef struct { bytebuf bb; uint32_t acc; int nacc; } bitwriter;
static inline void bw_bit(bitwriter* w, int v) {
  w->acc = (w->acc << 1) | (v & 1);
  if (++w->nacc == 8) { bb_put(&w->bb, (uint8_t)w->acc); w->acc = 0; w->nacc = 0; }
}
static inline void bw_u(bitwriter* w, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bw_bit(w, (v >> i) & 1); }
static inline void bw_ue(bitwriter* w, uint32_t v) {
  uint32_t k = v + 1; int len = 0; while ((k >> len) > 1) len++;
  bw_u(w, 0, len); bw_u(w, k, len + 1);
}
static inline void bw_se(bitwriter* w, int32_t v) { bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static inline void bw_trailing(bitwriter* w) { bw_bit(w, 1); while (w->nacc) bw_bit(w, 0); }
static inline void bw_align_zero(bitwriter* w) { while (w->nacc) bw_bit(w, 0); }

/* ---------------------------------------------------------------- parameter sets (7.3.2) */
typedef struct {
  int valid;
  int sps_id, vps_id;
  int chroma_format_idc;
  int width, height;              /* pic_width/height_in_luma_samples (coded) */
  int conf_win[4];                /* left,right,top,bottom in chroma units */
  int bit_depth, bit_depth_c;
  int log2_max_poc_lsb;
  int max_dec_pic_buffering, num_reorder, max_latency;
  int log2_min_cb, log2_diff_max_min_cb, log2_ctb;
  int log2_min_tb, log2_diff_max_min_tb, log2_max_tb;
  int max_th_depth_inter, max_th_depth_intra;
  int scaling_list_enabled;
  int amp_enabled, sao_enabled, pcm_enabled;
  int pcm_bit_depth, pcm_bit_depth_c, log2_min_pcm, log2_max_pcm, pcm_loop_filter_disabled;
  int num_st_rps;
  struct { int num_neg, num_pos; int delta_poc[16]; int used[16]; int num; } st_rps[65];
  int long_term_ref_pics_present;
  int temporal_mvp_enabled, strong_intra_smoothing;
  int pic_w_ctb, pic_h_ctb, pic_w_mincb, pic_h_mincb;
  int max_sub_layers;
} hevc_sps;

typedef struct {
  int valid;
  int pps_id, sps_id;
  int dependent_slice_segments_enabled, output_flag_present, num_extra_slice_header_bits;
  int sign_data_hiding, cabac_init_present;
  int num_ref_idx_default[2];
  int init_qp;
  int constrained_intra_pred, transform_skip_enabled;
  int cu_qp_delta_enabled, diff_cu_qp_delta_depth;
  int cb_qp_offset, cr_qp_offset, slice_chroma_qp_offsets_present;
  int weighted_pred, weighted_bipred;
  int transquant_bypass_enabled;
  int tiles_enabled, entropy_coding_sync;
  int loop_filter_across_slices;
  int deblocking_control_present, deblocking_override_enabled, pps_deblocking_disabled;
  int beta_offset_div2, tc_offset_div2;
  int lists_modification_present, log2_parallel_merge_level, slice_header_extension_present;
} hevc_pps;

typedef struct {
  int first_slice_in_pic, no_output_of_prior_pics, pps_id, dependent, segment_addr;
  int slice_type, pic_output;
  int poc_lsb, poc;
  int short_term_ref_pic_set_sps_flag, st_rps_idx;
  int rps_num, rps_delta[16], rps_used[16];    /* the RPS in effect (delta POC rel. to current) */
  int temporal_mvp, sao_luma, sao_chroma;
  int num_ref_idx[2];
  int mvd_l1_zero, cabac_init_flag, collocated_from_l0, collocated_ref_idx;
  int max_merge_cand;
  int qp_delta, cb_qp_offset, cr_qp_offset;
  int deblocking_disabled, beta_offset_div2, tc_offset_div2;
  int loop_filter_across_slices;
  int num_entry_points;
  int qp;                                      /* SliceQpY */
  /* pred_weight_table (7.3.6.3, P slices of a PPS with weighted_pred_flag): the syntax elements per entry of RefPicList0 and what 7.4.7.3 derives from them */
  int wp_luma_denom, wp_chroma_denom;          /* luma_log2_weight_denom, ChromaLog2WeightDenom */
  int wp_luma_flag[16], wp_chroma_flag[16];
  int wp_w[16][3], wp_o[16][3];                /* LumaWeightL0 / ChromaWeightL0, luma_offset_l0 / ChromaOffsetL0 (before the shift by BitDepth - 8) */
  int nal_type;
  size_t data_bit_offset;                      /* bit offset of slice_segment_data() in the RBSP */
} hevc_slice_hdr;

/* ---------------------------------------------------------------- MD5 (oracle/md5.c) */
void oracle_md5(const uint8_t* data, size_t n, uint8_t out[16]);
void oracle_md5_plane(const uint16_t* p, int w, int h, int bit_depth, uint8_t out[16]);

#endif
