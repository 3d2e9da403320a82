/* ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
 *
 * Normative ITU-T H.265 constants used by the CPU restatement of the transcode hot path.
 * The reference delegates all HEVC arithmetic to libavcodec/libx265 (PCCTranscoder.cpp:428-448,548-592);
 * the tables that ARE in the reference tree (dependencies/PccLibHevcParser) are cross-checked against this
 * file by tests/test_oracle_tables.py (golden: tests/golden/hevc_rom_tables.json):
 *   DCT/DST matrices      source/PccHevcTComRom.cpp:471-616
 *   quant/dequant scales  source/PccHevcTComRom.cpp:457-465
 *   chroma QP map         source/PccHevcTComRom.cpp:635-642
 *   CABAC init values     include/PccHevcContextTables.h:186-573
 * CABAC engine tables (rangeTabLPS, state transitions), intra angles, interpolation filters, deblock tc/beta
 * tables are restated from H.265 (9.3.4.3, 8.4.4.2.6, 8.5.3.3.3, 8.7.2.5.3); they are absent from the reference.
 */
#ifndef ORACLE_HEVC_TABLES_H
#define ORACLE_HEVC_TABLES_H
#include <stdint.h>

/* cos(m*pi/64) scaled: every entry of the 4/8/16/32-point HEVC core transforms is +-one of these. */
static const int8_t k_dct_angle[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                       61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};

/* T[k][n] of the N-point transform, N = 4,8,16,32: angle index m = k*(2n+1)*(32/N) mod 128. */
static inline int hevc_dct_coef(int N, int k, int n) {
  int m = (k * (2 * n + 1) * (32 / N)) & 127;
  if (m <= 32) return k_dct_angle[m];
  if (m <= 64) return -k_dct_angle[64 - m];
  if (m <= 96) return -k_dct_angle[m - 64];
  return k_dct_angle[128 - m];
}

static const int8_t k_dst4[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

static const int k_quant_scale[6]   = {26214, 23302, 20560, 18396, 16384, 14564};
static const int k_dequant_scale[6] = {40, 45, 51, 57, 64, 72};

/* qPi -> QpC for ChromaArrayType == 1 (Table 8-10) */
static inline int hevc_chroma_qp(int qpi) {
  static const int8_t t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  if (qpi < 30) return qpi;
  if (qpi >= 44) return qpi - 6;
  return t[qpi - 30];
}

/* ---- CABAC engine (9.3.4.3) ---- */
static const uint8_t k_range_lps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195},
    {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},
    {85, 104, 123, 142},  {81, 99, 117, 135},   {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},
    {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},     {56, 69, 81, 94},     {53, 65, 77, 89},
    {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},     {41, 50, 59, 69},
    {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},
    {23, 28, 33, 39},     {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},
    {18, 22, 26, 30},     {17, 21, 25, 28},     {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},
    {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},     {12, 14, 17, 20},     {11, 14, 16, 19},
    {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},      {9, 11, 12, 14},
    {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
static const uint8_t k_next_lps[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12,
                                       13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                       24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33,
                                       33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};
static inline int hevc_next_mps(int s) { return s >= 62 ? s : s + 1; }

/* ---- CABAC context layout (own layout; initValues per initType 0 (I), 1, 2 — Tables 9-5..9-37) ---- */
enum {
  CTX_SAO_MERGE = 0,                       /* 1 */
  CTX_SAO_TYPE = 1,                        /* 1 */
  CTX_SPLIT_CU = 2,                        /* 3 */
  CTX_CU_TQ_BYPASS = 5,                    /* 1 */
  CTX_CU_SKIP = 6,                         /* 3 */
  CTX_PRED_MODE = 9,                       /* 1 */
  CTX_PART_MODE = 10,                      /* 4 */
  CTX_PREV_INTRA_LUMA = 14,                /* 1 */
  CTX_INTRA_CHROMA = 15,                   /* 1 */
  CTX_RQT_ROOT_CBF = 16,                   /* 1 */
  CTX_MERGE_FLAG = 17,                     /* 1 */
  CTX_MERGE_IDX = 18,                      /* 1 */
  CTX_INTER_PRED_IDC = 19,                 /* 5 */
  CTX_REF_IDX = 24,                        /* 2 */
  CTX_MVP_FLAG = 26,                       /* 1 */
  CTX_SPLIT_TRANSFORM = 27,                /* 3 */
  CTX_CBF_LUMA = 30,                       /* 2 */
  CTX_CBF_CHROMA = 32,                     /* 5 */
  CTX_MVD_GT0 = 37,                        /* 1 */
  CTX_MVD_GT1 = 38,                        /* 1 */
  CTX_CU_QP_DELTA = 39,                    /* 2 */
  CTX_TRANSFORM_SKIP = 41,                 /* 2: luma, chroma */
  CTX_LAST_X = 43,                         /* 18: 15 luma + 3 chroma */
  CTX_LAST_Y = 61,                         /* 18 */
  CTX_CSBF = 79,                           /* 4 */
  CTX_SIG = 83,                            /* 44: 27 luma, 15 chroma, 2 transform-skip-context (unused) */
  CTX_GT1 = 127,                           /* 24 */
  CTX_GT2 = 151,                           /* 6 */
  CTX_COUNT = 157
};

/* row 0 = initType 0 (I slices), row 1 = initType 1, row 2 = initType 2 */
static const uint8_t k_ctx_init[3][CTX_COUNT] = {
    {/* sao_merge */ 153, /* sao_type */ 200, /* split_cu */ 139, 141, 157, /* tq_bypass */ 154,
     /* cu_skip */ 154, 154, 154, /* pred_mode */ 154, /* part_mode */ 184, 154, 154, 154,
     /* prev_intra */ 184, /* intra_chroma */ 63, /* rqt_root */ 154, /* merge_flag */ 154, /* merge_idx */ 154,
     /* inter_pred_idc */ 154, 154, 154, 154, 154, /* ref_idx */ 154, 154, /* mvp */ 154,
     /* split_tf */ 153, 138, 138, /* cbf_luma */ 111, 141, /* cbf_chroma */ 94, 138, 182, 154, 154,
     /* mvd */ 154, 154, /* dqp */ 154, 154, /* ts */ 139, 139,
     /* last_x */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
     /* last_y */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
     /* csbf */ 91, 171, 134, 141,
     /* sig luma */ 111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179,
     153, 125, 107, 125, 141, 179, 153, 125,
     /* sig chroma */ 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
     /* sig ts */ 141, 111,
     /* gt1 */ 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140,
     227, 122, 197,
     /* gt2 */ 138, 153, 136, 167, 152, 152},
    {/* sao_merge */ 153, /* sao_type */ 185, /* split_cu */ 107, 139, 126, /* tq_bypass */ 154,
     /* cu_skip */ 197, 185, 201, /* pred_mode */ 149, /* part_mode */ 154, 139, 154, 154,
     /* prev_intra */ 154, /* intra_chroma */ 152, /* rqt_root */ 79, /* merge_flag */ 110, /* merge_idx */ 122,
     /* inter_pred_idc */ 95, 79, 63, 31, 31, /* ref_idx */ 153, 153, /* mvp */ 168,
     /* split_tf */ 124, 138, 94, /* cbf_luma */ 153, 111, /* cbf_chroma */ 149, 107, 167, 154, 154,
     /* mvd */ 140, 198, /* dqp */ 154, 154, /* ts */ 139, 139,
     /* last_x */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
     /* last_y */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
     /* csbf */ 121, 140, 61, 154,
     /* sig luma */ 155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136,
     153, 154, 166, 183, 140, 136, 153, 154,
     /* sig chroma */ 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
     /* sig ts */ 140, 140,
     /* gt1 */ 154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167,
     154, 167, 137, 182,
     /* gt2 */ 107, 167, 91, 122, 107, 167},
    {/* sao_merge */ 153, /* sao_type */ 160, /* split_cu */ 107, 139, 126, /* tq_bypass */ 154,
     /* cu_skip */ 197, 185, 201, /* pred_mode */ 134, /* part_mode */ 154, 139, 154, 154,
     /* prev_intra */ 183, /* intra_chroma */ 152, /* rqt_root */ 79, /* merge_flag */ 154, /* merge_idx */ 137,
     /* inter_pred_idc */ 95, 79, 63, 31, 31, /* ref_idx */ 153, 153, /* mvp */ 168,
     /* split_tf */ 224, 167, 122, /* cbf_luma */ 153, 111, /* cbf_chroma */ 149, 92, 167, 154, 154,
     /* mvd */ 169, 198, /* dqp */ 154, 154, /* ts */ 139, 139,
     /* last_x */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
     /* last_y */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
     /* csbf */ 121, 140, 61, 154,
     /* sig luma */ 170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136,
     153, 154, 166, 183, 140, 136, 153, 154,
     /* sig chroma */ 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140,
     /* sig ts */ 140, 140,
     /* gt1 */ 154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167,
     154, 152, 167, 182,
     /* gt2 */ 107, 167, 91, 107, 107, 167}};

/* sig_coeff_flag ctxIdxMap for 4x4 TBs (9.3.4.2.5) */
static const uint8_t k_sig_ctx_4x4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};

/* ---- intra prediction (8.4.4.2.6) ---- */
static const int8_t k_intra_angle[35] = {0,  0,  32,  26,  21,  17,  13,  9,   5,   2,   0,   -2,
                                         -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9,
                                         -5, -2, 0,   2,   5,   9,   13,  17,  21,  26,  32};
static const int16_t k_intra_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256,
                                              -315,  -390,  -482, -630, -910, -1638, -4096}; /* modes 11..25 */

/* ---- fractional sample interpolation (8.5.3.3.3) ---- */
static const int8_t k_luma_filter[4][8]   = {{0, 0, 0, 64, 0, 0, 0, 0},
                                             {-1, 4, -10, 58, 17, -5, 1, 0},
                                             {-1, 4, -11, 40, 40, -11, 4, -1},
                                             {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t k_chroma_filter[8][4] = {{0, 64, 0, 0},    {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                             {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

/* ---- deblocking (8.7.2.5.3) ---- */
static const uint8_t k_beta_table[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,
                                         8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
                                         34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t k_tc_table[54]   = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,
                                         1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2,  2,  3,  3,  3,  3,  4,
                                         4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

#endif
