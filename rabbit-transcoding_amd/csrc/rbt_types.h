// Data layout in HBM shared by the host orchestrator and the kernels.
//
// One RbtFrame per coded picture of a batch (a batch = every picture of the sub-bitstreams handed to one call).
// All per-picture state lives in HBM for the whole call (288 GB: a 32-frame GOF of the CTC streams needs ~2 GB):
//   pix[3]    uint16 planes, stride = plane width: reconstruction, deblocked in place
//   out[3]    uint16 planes: SAO output (== pix when SAO is off for the picture)
//   coef[3]   int16 planes, same geometry as pix: transform coefficient LEVELS at their TB position
//   per 4x4 luma unit maps (w4 x h4): pm, edges, dm, qp, mv, ref
//   per CTB: sao parameters, slice index, command list (decode) / CU decisions (encode)
#pragma once
#include <stdint.h>

enum { RBT_SLICE_B = 0, RBT_SLICE_P = 1, RBT_SLICE_I = 2 };
enum { RBT_PART_2Nx2N = 0, RBT_PART_2NxN, RBT_PART_Nx2N, RBT_PART_NxN, RBT_PART_2NxnU, RBT_PART_2NxnD, RBT_PART_nLx2N, RBT_PART_nRx2N };
enum { RBT_MODE_INTER = 0, RBT_MODE_INTRA = 1, RBT_MODE_SKIP = 2, RBT_MODE_NONE = 3 };

// pm map bits
#define RBT_PM_MODE_MASK 3
#define RBT_PM_TQ_BYPASS 4
#define RBT_PM_NZ 8
// edges map bits: vertical edge on the unit's left boundary (TU 1, PU 2), horizontal edge on its top boundary (TU 4, PU 8)
#define RBT_EV_TU 1
#define RBT_EV_PU 2
#define RBT_EH_TU 4
#define RBT_EH_PU 8

#define RBT_MAX_REFS 4
// command list capacity of a CTB = 2 * (ctb/4)^2: every 4x4 TU plus every 8x4 PU (RbtFrame::cmd_cap)

struct RbtStreamCfg {            // SPS + PPS fields the kernels need
  int32_t w, h, cw, ch, w4, h4, w_ctb, h_ctb;
  int8_t bit_depth, log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb, th_depth_inter, th_depth_intra, diff_cu_qp_delta_depth;
  uint8_t amp, sao, strong_intra, tmvp, sign_hiding, cabac_init_present, cip, transform_skip;
  uint8_t cu_qp_delta, tq_bypass_enabled, pad0, pad1;
  int8_t cb_qp_offset, cr_qp_offset, pad2, pad3;
};

struct alignas(4) RbtSao { uint8_t type[3], band_pos[3], eo_class[3]; int8_t offset[3][4]; uint8_t pad[3]; };   // 24 bytes

struct RbtCmd {                  // 16 bytes, one per CU / PU / TU in decode order inside a CTB
  uint8_t type;                  // 1 PU, 2 TU
  uint8_t x4, y4;                // position inside the CTB in 4-luma-sample units
  uint8_t log2;                  // TU: log2 luma TB size
  uint8_t a, b, c, d;            // PU: w4, h4, ref_idx, -   TU: flags, intra_luma, intra_chroma, -
  int16_t mvx, mvy;              // PU
  int8_t qp[3];                  // TU: qP for Y, Cb, Cr (including QpBdOffset)
  uint8_t pad;
};
#define RBT_CMD_PU 1
#define RBT_CMD_TU 2
#define RBT_TU_CBF_Y 1
#define RBT_TU_CBF_CB 2
#define RBT_TU_CBF_CR 4
#define RBT_TU_TS_Y 8
#define RBT_TU_TS_CB 16
#define RBT_TU_TS_CR 32
#define RBT_TU_CHROMA 64
#define RBT_TU_INTRA 128

struct RbtFrame {
  RbtStreamCfg cfg;
  uint16_t* pix[3];
  uint16_t* out[3];
  int16_t* coef[3];
  uint8_t* pm;                   // per 4x4: mode | tq_bypass | nz
  uint8_t* edges;                // per 4x4
  uint8_t* dm;                   // per 4x4: cu depth (bits 6-7) | luma intra pred mode (bits 0-5)
  int8_t* qp;                    // per 4x4: QpY
  int16_t* mv;                   // per 4x4: x, y
  int8_t* ref;                   // per 4x4: ref_idx (list 0), -1 none
  int32_t* refpoc;               // per 4x4: POC of the reference (for TMVP / deblocking), filled after parsing
  RbtSao* sao;                   // per CTB
  uint16_t* ctb_slice;           // per CTB: index into the batch slice table
  RbtCmd* cmds;                  // per CTB: cmd_cap records
  int32_t cmd_cap;
  uint32_t* cmd_count;           // per CTB
  int32_t poc;
  int32_t level;                 // dependency level inside the batch (0: no references inside the batch)
  int32_t n_slices, first_slice;
  int32_t error;                 // set by kernels (non-zero = corrupt / unsupported stream)
  uint32_t* ctb_done;            // per CTB two words (luma chain, Cb/Cr chain): set when that half of the CTB is reconstructed (k_recon_level)
  // wavefront streams (decoder): what the wave of one CTB row hands to the wave of the row below: CTBs parsed (zeroed per job), the context variables after
  // the second CTB (256 bytes per row), and the row's bottom line (prow_line_bytes per row, a multiple of 256 so that no cache line is shared with anything the
  // reading wave writes itself: mv int32[w4] | RbtSao[w_ctb] | slice u16[w_ctb] | pm, dm, ref bytes[w4] each)
  uint32_t* prow_done; uint8_t* prow_ctx; uint8_t* prow_line; int32_t prow_line_bytes;
  // ---- encoder side (RBT-E1) ----
  const uint16_t* src[3];        // source planes (the decoder's `out` planes or the pooled occupancy map)
  uint8_t* cu_log2;              // per 8x8 unit: log2 size of the coding unit covering it
  uint8_t* cu_mode;              // per 8x8 unit: luma intra prediction mode of that CU
  uint8_t* cu_flags;             // per 8x8 unit: RBT_CU_* bits of that CU
  uint8_t* cu_ts;                // per 8x8 unit (encoder, streams with transform skip): transform_skip_flag of the 4x4 luma blocks 0..3 of an 8x8 CU coded as four
  int32_t w8, h8;
  int32_t lossless;              // every CU cu_transquant_bypass (x265 lossless=1, PCCTranscoder.cpp:841)
  int32_t ref_frame;             // P pictures: batch index of the reference picture (zero-motion merge), else -1
  int32_t ref_poc;
  // wavefront mode (one dependent slice segment per CTB row): CTBs finished per row by the closed-loop intra stage [0, h_ctb) and by the entropy
  // coder [h_ctb, 2 h_ctb), then the next row to hand out in each of the two stages, zeroed per job; the entropy coder's context variables after the
  // second CTB of each row, 256 bytes per row
  uint32_t* row_done;
  uint8_t* row_ctx;
  // transcoder: the decoded INPUT picture's per-4x4 maps (pm: prediction mode, dm: luma intra mode in bits 0..5), hint_w4 x hint_h4 units, or nullptr.
  // Where given, the intra analysis tries planar, DC and the input stream's modes at a block's four quarters instead of searching (en_analyse_ctb).
  const uint8_t* hint_pm; const uint8_t* hint_dm;
  int32_t hint_w4, hint_h4;
  // occupancy-aware coding (oracle_enc_params.occ4): one byte per 4x4 luma unit of the picture, != 0 where the decoder makes a point of some sample of
  // the unit or of a unit next to it (k_occ_units, from the occupancy map the output carries); occ4_w x occ4_h units, units beyond are unoccupied; nullptr = every sample counts
  const uint8_t* occ4; int32_t occ4_w, occ4_h;
  int32_t enc_tools;             // RBT_ET_* decision tools of RBT-E1 (all on unless a development switch RBT_ENC_SATD / _REFINE / _RQ = 0 says otherwise: oracle/hevc_enc.c)
  int32_t pad_et;
};
#define RBT_ET_SATD 1      // block costs of the intra analysis by SATD (en_analyse_ctb)
#define RBT_ET_REFINE 2    // closed-loop choice of the luma intra mode (en_refine_mode)
#define RBT_ET_RQ 4        // rounding offset of the intra quantiser by level and position (en_rq_offset)
#define RBT_ET_RDM 16      // the two cheapest candidates of the closed-loop mode choice coded as one transform block each, the cheaper kept (en_intra_ctb; oracle e1_mode_trial)
#define RBT_CU_CBF_Y 1
#define RBT_CU_CBF_CB 2
#define RBT_CU_CBF_CR 4
#define RBT_CU_SKIP 8
// intra CU coded as four transform units (RBT-E1 codes an intra CU as one TU or four): in CUs of 16 and 32 every 8x8 unit then carries the cbf bits of
// the TU that covers it; in an 8x8 CU the luma cbf of the 4x4 TUs 1..3 sits in the three bits above (TU 0 in RBT_CU_CBF_Y), Cb / Cr are the CU's
#define RBT_CU_TU_SPLIT 16
#define RBT_CU_CBF_Y1 32

struct RbtSlice;
struct RbtFrame;
// D2 metric (csrc/rbt_pcc.h): one cloud - points, bit volume of its voxels, hash map voxel -> lowest point index (keys: voxel id + 1, 0 = empty)
struct RbtD2Set { const int16_t* xyz; int n; const uint32_t* vol; const uint32_t* keys; const uint32_t* vals; int lg; };
// One slice segment of a merged entropy-decoding launch (slices of several batches in one grid: rbt_kernels.h launch_parse_tasks)
struct RbtParseTask { RbtFrame* frames; RbtSlice* slices; const uint8_t* rbsp; int32_t slice; int32_t pad; };   // handed to waves in list order (a row task waits for the task of the row above it)
// One picture of a merged reconstruction launch (pictures of several batches on the same wavefront: launch_recon_refs)
// order: the picture's CTBs in dependency order (anti-diagonals x + 2y ascending), x | y << 16 each (launch_recon_level)
struct RbtFrameRef { RbtFrame* frames; const RbtSlice* slices; const uint32_t* order; int32_t frame; int32_t pad; };

struct RbtSlice {                // one per slice segment, parsed on the host (7.3.6)
  int32_t frame;                 // index into the batch frame table
  uint32_t data_off, data_size;  // slice_segment_data() inside the batch RBSP buffer (emulation prevention removed)
  int32_t ctb_addr;              // slice_segment_address
  int8_t slice_type, qp, cb_qp_offset, cr_qp_offset;
  uint8_t sao_luma, sao_chroma, deblocking_disabled, lf_across;
  int8_t beta_offset_div2, tc_offset_div2;
  uint8_t temporal_mvp, cabac_init_flag, max_merge_cand, num_ref_idx, collocated_ref_idx;
  uint8_t dependent;             // dependent slice segment (7.3.6.1): continues the slice of the segment before it; every other header field repeats that slice's
  uint8_t wpp;                   // entropy_coding_sync_enabled_flag of the PPS: CTB rows are separate arithmetic codewords with inherited context variables (9.3.1)
  uint8_t row_task;              // decoder, wavefront streams: this segment starts a CTB row and is parsed by a wave of its own, which takes the state of the row
                                 // above (context variables after its second CTB, bottom line of its units) from the wave that parses that row (rbt_parse.h)
  int32_t next_seg;              // decoder: the next dependent segment of the same slice that is NOT a row task (-1: none); the same wave goes on with it
  int32_t head;                  // decoder: index of the independent segment that heads this segment's slice (= own index for an independent one)
  int32_t end_addr;              // decoder: the CTB address this entry has to end at: where the next slice segment (or substream entry) of the picture starts, or the picture's CTB
                                 // count - the host knows the starts, only the parser finds the ends: an entry that ends anywhere else leaves a hole or runs into its successor
  int32_t ctb_limit;             // decoder: > 0: this entry is ONE substream (CTB row) of a segment with entry points: stop after that many CTBs, at end_of_subset_one_bit
  int32_t ref_frame[RBT_MAX_REFS];   // batch frame index of RefPicList0[i]
  int32_t ref_poc[RBT_MAX_REFS];
  // explicit weighted sample prediction (8.5.3.3.4.3) of a P slice under weighted_pred_flag, per RefPicList0 entry and component: weight, offset at the sample bit depth;
  // wp_shift = log2WD = weight denominator + 14 - bitDepth for luma / chroma; wp_on = 0: default weighting
  int16_t wp_w[RBT_MAX_REFS][3], wp_o[RBT_MAX_REFS][3];
  int8_t wp_shift[2]; uint8_t wp_on, wp_pad;
  int32_t poc;
  uint32_t n_ctbs_decoded;       // out: CTBs the slice covered
  // ---- encoder side ----
  int32_t n_ctbs;                // CTBs of this slice segment (encoder input)
  uint32_t out_off, out_cap;     // slice_segment_data() bytes inside the batch output buffer
  uint32_t out_size;             // out: bytes written (out_size > out_cap = overflow)
};
