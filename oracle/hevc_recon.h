/* ORACLE — test infrastructure only. Picture-reconstruction primitives shared by the oracle decoder and encoder:
 * scaling + inverse transforms (H.265 8.6), intra prediction (8.4.4.2), inter prediction (8.5.3.3),
 * deblocking (8.7.2), SAO (8.7.3). The reference gets these from libavcodec/libx265 (PCCTranscoder.cpp:428-448,561). */
#ifndef ORACLE_HEVC_RECON_H
#define ORACLE_HEVC_RECON_H
#include "hevc_common.h"

#define META_UNDECODED 0xFF

typedef struct {
  uint8_t deblocking_disabled, loop_filter_across, sao_luma, sao_chroma;
  int8_t beta_offset_div2, tc_offset_div2;
  int8_t slice_type;
  int32_t ref_poc[16];    /* POC of RefPicList0 entries */
} hevc_slice_meta;

typedef struct {
  uint8_t type[3];        /* 0 off, 1 band, 2 edge */
  uint8_t band_pos[3];
  uint8_t eo_class[3];
  int8_t offset[3][4];
} hevc_sao;

typedef struct {
  int w, h, w4, h4;             /* luma size; size in 4x4 units */
  int log2_ctb, w_ctb, h_ctb;
  uint8_t* pred_mode;           /* per 4x4: MODE_* or META_UNDECODED */
  uint8_t* done;                /* per 4x4: samples reconstructed (pre-loop-filter) */
  uint8_t* intra_mode;          /* per 4x4: luma intra pred mode */
  uint8_t* cu_depth;            /* per 4x4 */
  int8_t* qp;                   /* per 4x4: QpY */
  uint8_t* tq_bypass;           /* per 4x4 */
  uint8_t* nz;                  /* per 4x4: luma TB covering it has non-zero coefficients */
  uint8_t* edge_v;              /* per 4x4: bit0 TU edge, bit1 PU edge on its left boundary */
  uint8_t* edge_h;              /* per 4x4: same for its top boundary */
  int16_t* mv;                  /* per 4x4: x,y (list 0) */
  int8_t* ref_idx;              /* per 4x4: -1 = none */
  uint16_t* ctb_slice;          /* per CTB: index into slices[] */
  hevc_sao* sao;                /* per CTB */
  hevc_slice_meta slices[1024];
  int n_slices;
  int constrained_intra_pred, pcm_loop_filter_disabled;
  int cb_qp_offset, cr_qp_offset;
  int strong_intra_smoothing;
} hevc_meta;

hevc_meta* hevc_meta_alloc(int w, int h, int log2_ctb);
void hevc_meta_reset(hevc_meta* m);
void hevc_meta_free(hevc_meta* m);

static inline int meta_idx(const hevc_meta* m, int x, int y) { return (y >> 2) * m->w4 + (x >> 2); }
static inline int meta_slice_at(const hevc_meta* m, int x, int y) {
  return m->ctb_slice[(y >> m->log2_ctb) * m->w_ctb + (x >> m->log2_ctb)];
}
/* z-scan availability (6.4.1) realised through the decode-order maps */
int hevc_avail_intra(const hevc_meta* m, int xc, int yc, int xn, int yn);
int hevc_avail_cu(const hevc_meta* m, int xc, int yc, int xn, int yn);

/* transforms / scaling */
void hevc_dequant(const int16_t* lvl, int16_t* d, int log2, int qp, int bit_depth);
void hevc_inv_transform(const int16_t* d, int16_t* res, int log2, int is_dst, int bit_depth);
void hevc_inv_transform_skip(const int16_t* d, int16_t* res, int log2, int bit_depth);
void hevc_fwd_transform(const int16_t* res, int16_t* coef, int log2, int is_dst, int bit_depth);
/* returns number of non-zero levels */
int hevc_quant(const int16_t* coef, int16_t* lvl, int log2, int qp, int bit_depth, int is_intra);

/* intra prediction of one TB of component c_idx at component coords (x0,y0); writes into frame */
void hevc_intra_pred(hevc_frame* f, const hevc_meta* m, int c_idx, int x0, int y0, int log2, int mode);
/* same but returns the prediction in pred[] (n*n) without touching the frame */
void hevc_intra_pred_buf(const hevc_frame* f, const hevc_meta* m, int c_idx, int x0, int y0, int log2, int mode, uint16_t* pred);

/* most-probable-mode candidates of the luma PB at (xp,yp) (8.4.2) */
void hevc_intra_mpm(const hevc_meta* m, int xp, int yp, int cand[3]);

/* uni-directional inter prediction of a luma block + its chroma, writes final clipped samples into f */
void hevc_inter_pred(hevc_frame* f, const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy);
/* the same with explicit weighted sample prediction (8.5.3.3.4.3, uni-prediction): per component c the weight w[c], the offset o[c] already shifted to the sample bit depth
 * and log2WD = the weight denominator + 14 - bitDepth in shift[c != 0]; wp == NULL: default weighting */
typedef struct { int w[3], o[3], shift[2]; } hevc_wp;
void hevc_inter_pred_wp(hevc_frame* f, const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy, const hevc_wp* wp);

/* the luma block hevc_inter_pred would write, into out[w*h] (w, h <= 64) */
void hevc_mc_luma_buf(const hevc_frame* ref, int x0, int y0, int w, int h, int mvx, int mvy, uint16_t* out);

/* motion vector prediction (8.5.3.2.2 - 8.5.3.2.9), list 0 only */
typedef struct { int16_t x, y; int ref; } hevc_mvcand;
typedef struct { int16_t* mv; int32_t* refpoc; int poc; int w4, h4; uint8_t* imode; } hevc_colinfo;   /* refpoc INT_MIN = not inter */
typedef struct {
  const hevc_meta* m; int part_mode; int max_merge_cand; int num_ref_idx; const int* ref_poc; int cur_poc;
  const hevc_colinfo* col;       /* collocated picture motion, NULL when slice_temporal_mvp_enabled_flag == 0 */
  int log2_ctb, pic_w, pic_h;
} hevc_mvpred;
hevc_mvcand hevc_merge_candidate(const hevc_mvpred* s, int xpb, int ypb, int w, int h, int part_idx, int merge_idx);
hevc_mvcand hevc_amvp_candidate(const hevc_mvpred* s, int xpb, int ypb, int w, int h, int ref_idx, int mvp_flag);

void hevc_deblock(hevc_frame* f, const hevc_meta* m);
/* SAO: reads deblocked picture src, writes dst (may not alias) */
void hevc_sao_apply(hevc_frame* dst, const hevc_frame* src, const hevc_meta* m);

#endif
