// Encode half + the transcode pipeline (decode -> pool -> encode). See rbt_transcode.cpp.
#pragma once
#include <string>
#include "../../include/rbt.h"
namespace rbt {
int transcode_gof(rbt_stats& st, std::string& err, int n, const uint8_t* const* in, const size_t* n_in, const rbt_stream_params* p, uint8_t** out, size_t* n_out);
struct GofJob;
// gof_rule: apply transcodeData's rule (PCCTranscoder.cpp:150): an occupancy stream is only transcoded when occupancyPrecision == 4
GofJob* gof_submit(int slot, int depth, int n, const uint8_t* const* in, const size_t* n_in, const rbt_stream_params* p, bool gof_rule);
int gof_wait(GofJob* j, rbt_stats& st, std::string& err, uint8_t** out, size_t* n_out);   // consumes the job
void gof_abandon(GofJob* j);
size_t gof_memory(const GofJob* j);             // device memory the job's build took (its arenas)
int encode_yuv(rbt_stats& st, std::string& err, const uint16_t* yuv, int w, int h, int bd, int n_frames, int qp, int gop, int lossless, int log2_ctb, int rows, int md5, uint8_t** out, size_t* n_out);
int or_pool_host(const uint16_t* plane, int w, int h, int factor, uint16_t* out);
}
