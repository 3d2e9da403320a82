/* ORACLE — test infrastructure only. Decoder API of the CPU restatement (see hevc_dec.c). */
#ifndef ORACLE_HEVC_DEC_H
#define ORACLE_HEVC_DEC_H
#include "hevc_common.h"
#include "hevc_recon.h"

typedef struct oracle_hevc_decoder oracle_hevc_decoder;

oracle_hevc_decoder* oracle_hevc_dec_create(void);
void oracle_hevc_dec_destroy(oracle_hevc_decoder* d);
/* Decodes a complete Annex-B elementary stream. Returns 0 on success (<0: error). Frames are kept in decode order
 * (== output order for the low-delay I/P structures this path handles). */
int oracle_hevc_dec_decode(oracle_hevc_decoder* d, const uint8_t* annexb, size_t n);
int oracle_hevc_dec_num_frames(const oracle_hevc_decoder* d);
const hevc_frame* oracle_hevc_dec_frame(const oracle_hevc_decoder* d, int i);
const uint8_t* oracle_hevc_dec_imodes(const oracle_hevc_decoder* d, int i);   /* per 4x4 unit: coded luma intra mode, 255 = not intra */
/* number of pictures whose MD5 SEI was checked / failed */
void oracle_hevc_dec_crop(const oracle_hevc_decoder* d, int out[4]);   /* luma samples to drop: left, right, top, bottom */
int oracle_hevc_dec_md5_checked(const oracle_hevc_decoder* d);
int oracle_hevc_dec_md5_failed(const oracle_hevc_decoder* d);
int oracle_slice_headers(const uint8_t* annexb, size_t n, int* out, int cap);   /* 18 ints per slice segment header, see hevc_dec.c */
#endif
