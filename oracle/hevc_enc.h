/* ORACLE — test infrastructure only. Encoder API of the CPU restatement (see hevc_enc.c). */
#ifndef ORACLE_HEVC_ENC_H
#define ORACLE_HEVC_ENC_H
#include "hevc_common.h"
#include "hevc_recon.h"

typedef struct {
  int width, height, bit_depth;
  int qp;                   /* CQP as x265 `qp=`: P slices use qp, I slices qp + i_qp_offset (x265 ipratio 1.4 -> -3) */
  int i_qp_offset;
  int gop;                  /* 1: all intra (occupancy, PCCTranscoder.cpp:835);  2: I,P pairs (PCCTranscoder.cpp:849) */
  int lossless;             /* x265 lossless=1 (PCCTranscoder.cpp:841): cu_transquant_bypass for every CU */
  int log2_ctb;             /* 4..6 */
  int ctb_rows_per_slice;   /* 0 = one slice per picture */
  int md5_sei;              /* emit decoded-picture-hash SEI */
  uint32_t stress_seed;     /* 0 = product decisions; != 0 = random-syntax generator for decoder test streams */
  int conf_win_right, conf_win_bottom;   /* conformance window offsets in chroma sample units (7.4.3.2.1): width / height are the CODED size */
  int hm_like;              /* 1 = "HM-like" decisions: the coding tools of the CTC input streams (cfg/hm/ctc-hm-geometry-ai.cfg: CTU 64, TU 4..32
                               :10-16, motion search :33-34, TransformSkip :47, SAO :68, AMP :69) chosen by a deterministic search instead of
                               RBT-E1's restricted set; produces the benchmark's R5 input (tests/golden/make_hm_gof.py). Not mirrored on the GPU. */
  int p_qp_offset;          /* hm_like: QP offset of P pictures (GOP table QPoffset: geometry -3, attribute 0; :29) */
  /* RBT-E1 as the second half of a transcoder: the luma intra modes the INPUT stream coded, per frame and 4x4 unit (255 = not intra; row stride hint_w4,
   * hint_h4 rows; NULL = none, the encoder searches on its own). Where given, the intra analysis tries planar, DC and the input's modes at the four
   * quarters of a block instead of searching the 35 modes (the input's encoder chose them with full rate-distortion optimisation). */
  const uint8_t* const* hint_modes; int hint_w4, hint_h4;
  /* Occupancy-aware coding of geometry / attribute maps (what dependencies/hm-modification/HM-16.20+SCM-8.8_with_RDO.patch does to HM's
   * distortion, TComRdCost.cpp xGetSSE*: `(org - cur) * (occupancy != 0)`): per frame a byte per 4x4 luma unit, != 0 where the decoder will make a point of some
   * sample of the unit (the atlas' occupancy map at the precision the OUTPUT carries). occ4_w x occ4_h units, row stride occ4_w; units beyond count as
   * unoccupied; NULL = off. A transform block without an occupied sample carries no residual (a 16x16 inter CU without one is a skip); in a partly occupied
   * block the other samples ask for the mean residual of the occupied ones; unoccupied samples are left out of the distortion of the transform-unit and
   * transform-skip decisions and of the SAO statistics. Mode and split decisions look at every sample: the closer the unoccupied area stays to the
   * source's padding, the better the occupied blocks next to it predict (measured: masking them as well costs 35 % more geometry bytes and 0.3 dB D1). */
  const uint8_t* const* occ4; int occ4_w, occ4_h;
  /* Stream structure of the CTC's HM encoder (cfg/hm/ctc-hm-geometry-ai.cfg:21-30, same in ctc-hm-attribute-ai.cfg: IntraPeriod -1, GOPSize 2, Frame1: P ref -1,
   * Frame2: I with RPS -2, ReWriteParamSetsFlag at IRAPs only): 0 = closed groups (every I picture an IDR with parameter sets, POC restarts);
   * 1 = ONE IDR_W_RADL with the parameter sets, every later picture a trailing picture with POC running on: I pictures are TRAIL_R with slice_type I and the
   * GOP table's reference picture set, pictures nobody references are TRAIL_N (HM: TemporalLayerNonReferenceFlag from the GOP entry's m_refPic), P pictures
   * reference POC - 1 (and POC - 2 with two references); 2 = the same with every trailing picture a TRAIL_R. The SPS carries the sets {-1}, {-2} (and {-1,-2}).
   * Random-syntax streams in this structure (stress_seed) also code sets explicitly in the slice header, with and without inter-set prediction. */
  int ctc_gop;
  int log2_max_poc_lsb;     /* 0 = 8 (HM's default); 4..16: pic_order_cnt_lsb wraps every 2^n pictures */
  int first_idx;            /* ctc_gop: index inside the stream of frames[0] (frames are independent of each other but for the headers: a long stream can be made
                               in pieces by parallel workers and concatenated; first_idx must start a group) */
  int weighted_pred;        /* random-syntax streams (stress_seed): PPS weighted_pred_flag with a pred_weight_table of random weights and offsets in every P slice
                               (what libx265 writes from its preset "veryfast" up); the weighting is applied to the generator's own prediction */
  int tools_off;            /* RBT-E1 decision tools to leave out: 1 SATD block costs, 2 closed-loop mode choice, 4 rounding by level and position, 16 coded trial of the two cheapest modes (the library's RBT_ET_* bits; oracle_transcode_params.preset) */
} oracle_enc_params;

/* Encodes n frames; appends an Annex-B stream to out. If recon != NULL it receives n newly allocated reconstructed
 * pictures (post loop filter). Returns 0 on success. */
int oracle_hevc_encode(const oracle_enc_params* p, const hevc_frame* const* frames, int n, bytebuf* out, hevc_frame** recon);

#endif
