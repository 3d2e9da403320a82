#include "rbt_transcode.h"
namespace rbt {
int transcode_gof(rbt_stats&, std::string& err, int, const uint8_t* const*, const size_t*, const rbt_stream_params*, uint8_t**, size_t*) { err = "not built yet"; return RBT_ERR_PARAM; }
int encode_yuv(rbt_stats&, std::string& err, const uint16_t*, int, int, int, int, int, int, int, int, int, int, uint8_t**, size_t*) { err = "not built yet"; return RBT_ERR_PARAM; }
int or_pool_host(const uint16_t*, int, int, int, uint16_t*) { return RBT_ERR_PARAM; }
}
