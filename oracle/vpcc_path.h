/* ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
 * Flat C entry points (ctypes-friendly) of the CPU restatement of the transcode hot path. */
#ifndef ORACLE_VPCC_PATH_H
#define ORACLE_VPCC_PATH_H
#include <stddef.h>
#include <stdint.h>
#include "hevc_enc.h"

typedef struct {
  int w, h, bit_depth, n_frames;
  uint16_t* data;             /* n_frames x planar 4:2:0 (Y w*h, Cb, Cr), malloc'd */
  int md5_checked, md5_failed;
} oracle_video;

typedef struct {
  int video_type;             /* PCCVideoType: 0 occupancy, 1 geometry, 19 attribute (PCCBitstreamCommon.h:79-118) */
  int qp;                     /* geometryQP_ / attributeQP_ / occupancyMapQP_ (PCCTranscoderParameters.h:58-80) */
  int occupancy_precision;    /* occupancyPrecision_: 4 -> 2x2 OR-pool (PCCTranscoder.cpp:466), else untouched */
  int log2_ctb;               /* implementation parameter of the RBT-E1 encoder (0 = default 5) */
  int ctb_rows_per_slice;     /* implementation parameter (0 = one slice per picture) */
  int md5_sei;
  int occupancy_rd;           /* geometry / attribute streams handed to oracle_transcode_data behind an occupancy stream: occupancy-aware coding (oracle_enc_params.occ4,
                                 SURVEY.md 8 row F4) with the occupancy map that stream comes out with; ignored by oracle_transcode_substream, which sees one stream */
  int preset;                 /* 0 = every decision tool of RBT-E1; 1 = "fast": the open-loop decisions only (oracle_enc_params.tools_off = 23); the library's rbt_stream_params.preset */
} oracle_transcode_params;

/* PCCVideoBitstream::sampleStreamToByteStream / byteStreamToSampleStream (PCCVideoBitstream.cpp:85-172), HEVC case,
 * precision 4, no emulation-prevention handling (the defaults the transcoder uses, PCCTranscoder.cpp:152,517) */
int oracle_sample_to_byte_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out);
int oracle_byte_to_sample_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out);

/* resize_frame2 (PCCTranscoder.cpp:594-646) == resizeOccupancyMap (:341-372): out = any(in block > 0) ? 1 : 0 */
void oracle_or_pool(const uint16_t* in, int w, int h, int factor, uint16_t* out);

int oracle_decode(const uint8_t* annexb, size_t n, oracle_video* out);
/* encode: yuv = n_frames x planar 4:2:0 uint16; params as oracle_enc_params fields */
int oracle_encode(int w, int h, int bit_depth, int qp, int i_qp_offset, int gop, int lossless, int log2_ctb, int rows_per_slice,
                  int md5_sei, uint32_t stress_seed, const uint16_t* yuv, int n_frames, uint8_t** out, size_t* n_out, uint16_t* recon);
int oracle_encode_ex(const oracle_enc_params* params, const uint16_t* yuv, int n_frames, uint8_t** out, size_t* n_out, uint16_t* recon);
/* PCCTranscoder::transcodeVideo (PCCTranscoder.cpp:374-546) on an Annex-B sub-bitstream */
int oracle_transcode_substream(const uint8_t* annexb, size_t n, const oracle_transcode_params* p, uint8_t** out, size_t* n_out);
/* PCCTranscoder::transcodeData (PCCTranscoder.cpp:145-168): occupancy only when occupancy_precision == 4, then geometry, attribute */
int oracle_transcode_data(int n, const uint8_t* const* in, const size_t* n_in, const oracle_transcode_params* p, uint8_t** out, size_t* n_out);
/* The V3C sample stream around transcodeData, as PccAppTranscoder's decompressVideo walks it (PccAppTranscoder.cpp:277-349): read (PCCBitstreamReader.cpp:51-70,
 * :1369-1387), GOF by GOF (:72-96), video units replaced by their transcodes (payload = sub-bitstream in sample stream form, PCCBitstream.cpp:88-111),
 * all units written as one sample stream (PCCBitstreamWriter.cpp:57-91, :1492-1507). V3C_VPS / V3C_AD units are carried over as bytes.
 * geometry / attribute: params of those two videos; occupancy transcoded (qp 8, lossless) only when occupancy_precision == 4. */
/* (occupancy_rd: the geometry / attribute units of a GOF are coded with the occupancy map its occupancy unit comes out with, oracle_transcode_params.occupancy_rd) */
int oracle_v3c_transcode(const uint8_t* in, size_t n, int occupancy_precision, int geometry_qp, int attribute_qp, int forced_precision_bytes,
                         int log2_ctb, int ctb_rows_per_slice, int md5_sei, int occupancy_rd, int preset, uint8_t** out, size_t* n_out);
void oracle_free(void* p);
#endif
