// C ABI of librbt.so (include/rbt.h). Drop-in for PCCTranscoder::transcodeVideo (PCCTranscoder.cpp:374-546).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include "rbt_batch.h"
#include "rbt_transcode.h"
#include "rbt_pcc.h"
#include "rbt_internal.h"

struct rbt_ctx { int device, rank, world; rbt_stats stats; std::string last_err; };
struct rbt_job { rbt::GofJob* j; rbt_ctx* owner; };

// Job slots, pipeline depth and the lock are per DEVICE (the 16 HIP streams a job's lanes map onto are the device's, rbt_kernels.hip):
// contexts on different devices run concurrently, contexts on one device share its slots and are serialised, as are the calls on one
// context (the reference's transcoder is called serially too).
struct DevState { rbt_job* jobs[RBT_MAX_JOBS] = {}; int depth = 4; std::mutex mu; };
static DevState g_dev[16];
static_assert(RBT_MAX_JOBS == rbtk::RBT_JOB_SLOTS, "job slots");
// every context entry point starts with an empty error text (rbt_last_error describes the LAST call; the pointer it returns is valid until the next call on the context)
#define RBT_ENTER(ctx) DevState& D = g_dev[(ctx)->device]; std::lock_guard<std::mutex> lk(D.mu); (ctx)->last_err.clear(); if (rbtk::dev_select((ctx)->device)) return RBT_ERR_NO_DEVICE
// no exception crosses the C ABI: allocation failures of the host side (std::vector, std::string, the hashing threads) come back as error codes
#define RBT_CATCH catch (const std::bad_alloc&) { return RBT_ERR_NOMEM; } catch (...) { return RBT_ERR_NO_DEVICE; }

extern "C" {

const char* rbt_version(void) { return "rabbit-transcoding_amd 0.1 (RBT-E1 encoder, gfx950)"; }

const char* rbt_last_error(rbt_ctx* ctx) { return ctx ? ctx->last_err.c_str() : ""; }
void rbt_internal_set_error(rbt_ctx* ctx, const char* text) { if (ctx) { try { ctx->last_err = text ? text : ""; } catch (...) {} } }
const char* rbt_strerror(int code) {
  switch (code) {
    case RBT_OK: return "ok";
    case RBT_ERR_NO_DEVICE: return "no usable HIP device or HIP runtime failure (this library has no CPU fallback)";
    case RBT_ERR_BITSTREAM: return "corrupt or truncated HEVC bitstream";
    case RBT_ERR_UNSUPPORTED: return "bitstream uses a coding tool outside the supported V-PCC CTC toolset";
    case RBT_ERR_PARAM: return "invalid parameter";
    case RBT_ERR_NOMEM: return "out of memory";
    case RBT_ERR_MD5: return "decoded picture hash mismatch on the input stream";
    case RBT_ERR_BUSY: return "too many transcodes in flight";
    default: return "unknown error";
  }
}
void rbt_free(void* p) { free(p); }

int rbt_create(rbt_ctx** ctx, int device, int world_rank, int world_size) try {
  if (!ctx) return RBT_ERR_PARAM;
  *ctx = nullptr;
  if (world_size < 1 || world_rank < 0 || world_rank >= world_size) return RBT_ERR_PARAM;
  if (device < 0 || device >= 16) return RBT_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(g_dev[device].mu);
  if (rbtk::dev_init(device)) return RBT_ERR_NO_DEVICE;
  rbt_ctx* c = new rbt_ctx(); c->device = device; c->rank = world_rank; c->world = world_size; memset(&c->stats, 0, sizeof(c->stats));
  *ctx = c;
  return RBT_OK;
} RBT_CATCH
void rbt_destroy(rbt_ctx* ctx) {
  if (ctx) {
    DevState& D = g_dev[ctx->device]; std::lock_guard<std::mutex> lk(D.mu);
    if (!rbtk::dev_select(ctx->device)) {
      for (int s = 0; s < RBT_MAX_JOBS; s++) if (D.jobs[s] && D.jobs[s]->owner == ctx) { rbt::gof_abandon(D.jobs[s]->j); delete D.jobs[s]; D.jobs[s] = nullptr; }
      rbtk::dev_release_pool();
    }
  }
  delete ctx;
}
// GOF sharding rule of the multi-GPU transcoder: GOF g of a sequence belongs to rank g mod world_size. A host that
// walks the sequence GOF by GOF (PccAppTranscoder.cpp:307-341) on every rank skips the GOFs its context does not own.
int rbt_owns_gof(const rbt_ctx* ctx, int gof_index) { return ctx && gof_index >= 0 && gof_index % ctx->world == ctx->rank; }
int rbt_world(const rbt_ctx* ctx, int* rank, int* size) { if (!ctx) return RBT_ERR_PARAM; if (rank) *rank = ctx->rank; if (size) *size = ctx->world; return RBT_OK; }
int rbt_get_stats(rbt_ctx* ctx, rbt_stats* out) { if (!ctx || !out) return RBT_ERR_PARAM; *out = ctx->stats; return RBT_OK; }

int rbt_decode(rbt_ctx* ctx, const uint8_t* annexb, size_t n, int verify_md5, rbt_video* out) try {
  if (!ctx || !annexb || !out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  memset(out, 0, sizeof(*out));
  rbt::DecodeBatch b; rbt::StreamIn in{annexb, n};
  int rc = rbt::decode_build(b, &in, 1);
  if (!rc) rc = rbt::decode_run(b);
  if (!rc) rc = rbt::decode_fetch(b, 0, out, verify_md5 != 0);
  if (rc) { ctx->last_err = b.err; free(out->data); memset(out, 0, sizeof(*out)); return rc; }
  ctx->stats.k_parse_ms = rbtk::timer_ms(rbt::T_PARSE); ctx->stats.k_recon_ms = rbtk::timer_ms(rbt::T_RECON);
  if (verify_md5 && out->md5_failed) return RBT_ERR_MD5;
  return RBT_OK;
} RBT_CATCH

// PCCVideoBitstream.cpp:174-184
static size_t end_of_nalu(const uint8_t* d, size_t size, size_t start) {
  if (size < start + 4) return size;
  for (size_t i = start; i < size - 4; i++)
    if (d[i] == 0 && d[i + 1] == 0 && (d[i + 2] == 1 || (d[i + 2] == 0 && d[i + 3] == 1))) return i;
  return size;
}
// PCCVideoBitstream::byteStreamToSampleStream (PCCVideoBitstream.cpp:85-112), precision 4, no emulation prevention handling
int rbt_byte_to_sample_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out) try {
  if (!in || !out || !n_out || n < 4) return RBT_ERR_PARAM;
  std::vector<uint8_t> v; v.reserve(n + 64);
  size_t start = 0, end = 0;
  do {
    size_t sc = in[start + 2] == 0 ? 4 : 3;
    end = end_of_nalu(in, n, start + sc);
    size_t sz = end - (start + sc);
    for (int i = 0; i < 4; i++) v.push_back((uint8_t)(sz >> (8 * (3 - i))));
    v.insert(v.end(), in + start + sc, in + end);
    start = end;
  } while (end < n);
  *out = (uint8_t*)malloc(v.size() ? v.size() : 1); if (!*out) return RBT_ERR_NOMEM;
  memcpy(*out, v.data(), v.size()); *n_out = v.size();
  return RBT_OK;
} RBT_CATCH
// PCCVideoBitstream::sampleStreamToByteStream (PCCVideoBitstream.cpp:114-172), HEVC, precision 4
int rbt_sample_to_byte_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out) try {
  if (!in || !out || !n_out || n < 4) return RBT_ERR_PARAM;
  std::vector<uint8_t> v; v.reserve(n + 64);
  size_t sc = 4, start = 0, end = 0;
  do {
    uint32_t sz = 0; for (int i = 0; i < 4; i++) sz = (sz << 8) + in[start + i];
    end = start + 4 + sz;
    if (end > n) return RBT_ERR_BITSTREAM;
    for (size_t i = 0; i + 1 < sc; i++) v.push_back(0);
    v.push_back(1);
    v.insert(v.end(), in + start + 4, in + end);
    start = end;
    if (start + 4 < n) { int type = (in[start + 4] & 126) >> 1; sc = (type >= 32 && type < 41) ? 4 : 3; }   // the reference resets newFrame before testing it (:146)
  } while (end < n);
  *out = (uint8_t*)malloc(v.size() ? v.size() : 1); if (!*out) return RBT_ERR_NOMEM;
  memcpy(*out, v.data(), v.size()); *n_out = v.size();
  return RBT_OK;
} RBT_CATCH

static int submit(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, rbt_job** job, bool gof_rule);
// transcodeVideo re-encodes whatever it is handed (an occupancy stream with occupancyPrecision != 4 is re-encoded without pooling)
int rbt_transcode_substream(rbt_ctx* ctx, const uint8_t* annexb_in, size_t n_in, const rbt_stream_params* p, uint8_t** annexb_out, size_t* n_out) try {
  if (!ctx || !annexb_in || !p || !annexb_out || !n_out) return RBT_ERR_PARAM;
  rbt_job* job = nullptr;
  int rc = submit(ctx, 1, &annexb_in, &n_in, p, &job, false);
  if (rc) return rc;
  return rbt_wait_gof(ctx, job, annexb_out, n_out);
} RBT_CATCH
int rbt_transcode_gof(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, uint8_t** annexb_out, size_t* n_out) try {
  if (!ctx || n < 1 || n > RBT_MAX_STREAMS || !annexb_in || !n_in || !p || !annexb_out || !n_out) return RBT_ERR_PARAM;
  rbt_job* job = nullptr;
  int rc = rbt_submit_gof(ctx, n, annexb_in, n_in, p, &job);
  if (rc) return rc;
  return rbt_wait_gof(ctx, job, annexb_out, n_out);
} RBT_CATCH
int rbt_submit_gof(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, rbt_job** job) try { return submit(ctx, n, annexb_in, n_in, p, job, true); } RBT_CATCH
static int submit(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, rbt_job** job, bool gof_rule) {
  if (!ctx || n < 1 || n > RBT_MAX_STREAMS || !annexb_in || !n_in || !p || !job) return RBT_ERR_PARAM;
  *job = nullptr;
  RBT_ENTER(ctx);
  int slot = -1;
  for (int s = 0; s < D.depth && slot < 0; s++) if (!D.jobs[s]) slot = s;
  if (slot < 0) return RBT_ERR_BUSY;
  rbt_job* j = new rbt_job{rbt::gof_submit(slot, D.depth, n, annexb_in, n_in, p, gof_rule), ctx};
  D.jobs[slot] = j; *job = j;
  return RBT_OK;                       // errors of the build surface in rbt_wait_gof, which also releases the job
}
int rbt_set_depth(rbt_ctx* ctx, int max_in_flight) try {
  if (!ctx || max_in_flight < 1 || max_in_flight > RBT_MAX_JOBS) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  for (int s = 0; s < RBT_MAX_JOBS; s++) if (D.jobs[s]) return RBT_ERR_BUSY;
  D.depth = max_in_flight;
  return RBT_OK;
} RBT_CATCH
int rbt_get_depth(rbt_ctx* ctx) try {
  if (!ctx) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return D.depth;
} RBT_CATCH
// Measured on one MI355X (tools/short_run_sweep.sh, tools/walk_shape_sweep.sh): a long walk keeps max_jobs jobs of 2 GOFs in flight; a walk shorter than 48 GOFs
// is all ramp-up and drain and runs as at most 7 jobs (2 up to 12 GOFs), which then own several hardware queues each. Same rule as gof_shard.job_shape.
int rbt_job_shape(int n_gofs, int max_jobs, int* gofs_per_job, int* jobs_in_flight) try {
  if (n_gofs < 0 || max_jobs < 1 || !gofs_per_job || !jobs_in_flight) return RBT_ERR_PARAM;
  if (max_jobs > RBT_MAX_JOBS) max_jobs = RBT_MAX_JOBS;
  if (n_gofs >= 96) { *gofs_per_job = 3; *jobs_in_flight = max_jobs; return RBT_OK; }      // round 3 (XCD-aware tile order, less HBM traffic): 16 x 3 GOFs 905-909 fps, 16 x 2 874-882, 12 x 4 897
  if (n_gofs >= 48) { *gofs_per_job = 2; *jobs_in_flight = max_jobs; return RBT_OK; }
  const int jobs = n_gofs <= 12 ? 2 : 7;
  int g = (n_gofs + jobs - 1) / jobs; if (g < 1) g = 1;
  int d = (n_gofs + g - 1) / g; if (d > max_jobs) d = max_jobs; if (d < 1) d = 1;
  *gofs_per_job = g; *jobs_in_flight = d;
  return RBT_OK;
} RBT_CATCH
int rbt_preset_from_name(const char* name) try {
  if (!name || !*name) return RBT_PRESET_DEFAULT;
  for (const char* f : {"ultrafast", "superfast"}) if (!strcmp(name, f)) return RBT_PRESET_FAST;
  for (const char* d : {"veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo"}) if (!strcmp(name, d)) return RBT_PRESET_DEFAULT;
  return RBT_ERR_PARAM;
} RBT_CATCH
int rbt_device_memory(rbt_ctx* ctx, rbt_memory* out) try {
  if (!ctx || !out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  memset(out, 0, sizeof(*out));
  if (rbtk::dev_mem_info(&out->free_bytes, &out->total_bytes, &out->cached_bytes, &out->in_use_bytes)) return RBT_ERR_NO_DEVICE;
  out->reserve_bytes = rbtk::dev_reserve_bytes();
  return RBT_OK;
} RBT_CATCH
int rbt_job_memory(rbt_ctx* ctx, const rbt_job* job, size_t* bytes) try {
  if (!ctx || !job || !bytes) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  bool mine = false; for (int s = 0; s < RBT_MAX_JOBS; s++) if (D.jobs[s] == job && job->owner == ctx) mine = true;
  if (!mine) return RBT_ERR_PARAM;
  *bytes = rbt::gof_memory(job->j);
  return RBT_OK;
} RBT_CATCH
int rbt_trim(rbt_ctx* ctx) try {
  if (!ctx) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  for (int s = 0; s < RBT_MAX_JOBS; s++) if (D.jobs[s]) return RBT_ERR_BUSY;
  rbtk::dev_release_pool();
  return RBT_OK;
} RBT_CATCH
int rbt_wait_gof(rbt_ctx* ctx, rbt_job* job, uint8_t** annexb_out, size_t* n_out) try {
  if (!ctx || !job || !annexb_out || !n_out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  int slot = -1;
  for (int s = 0; s < RBT_MAX_JOBS; s++) if (D.jobs[s] == job) slot = s;
  if (slot < 0 || job->owner != ctx) return RBT_ERR_PARAM;
  int rc = rbt::gof_wait(job->j, ctx->stats, ctx->last_err, annexb_out, n_out);
  D.jobs[slot] = nullptr; delete job;
  return rc;
} RBT_CATCH
int rbt_encode(rbt_ctx* ctx, const uint16_t* yuv, int width, int height, int bit_depth, int n_frames, int qp, int gop, int lossless,
               int log2_ctb, int ctb_rows_per_slice, int md5_sei, uint8_t** annexb_out, size_t* n_out) try {
  if (!ctx || !yuv || !annexb_out || !n_out || n_frames < 1) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return rbt::encode_yuv(ctx->stats, ctx->last_err, yuv, width, height, bit_depth, n_frames, qp, gop, lossless, log2_ctb, ctb_rows_per_slice, md5_sei, annexb_out, n_out);
} RBT_CATCH
int rbt_or_pool(rbt_ctx* ctx, const uint16_t* plane, int width, int height, int factor, uint16_t* out) try {
  if (!ctx || !plane || !out || factor < 1 || width % factor || height % factor) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return rbt::or_pool_host(plane, width, height, factor, out);
} RBT_CATCH

int rbt_selftest_transform32(rbt_ctx* ctx, const int16_t* blocks, int n_blocks, int bit_depth, uint32_t* n_mismatch) try {
  if (!ctx || !blocks || n_blocks < 1 || bit_depth < 8 || bit_depth > 12 || !n_mismatch) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return rbtk::selftest_transform32(blocks, n_blocks, bit_depth, n_mismatch) ? RBT_ERR_NO_DEVICE : RBT_OK;
} RBT_CATCH
int rbt_reconstruct(rbt_ctx* ctx, const rbt_atlas_params* atlas, const rbt_patch* patches, int n_patches, const uint16_t* occ_luma, const uint16_t* geo_d0,
                    const uint16_t* geo_d1, int geo_bit_depth, const uint16_t* attr_t0, const uint16_t* attr_t1, int attr_bit_depth, rbt_cloud* out) try {
  if (!ctx || !atlas || (!patches && n_patches) || !occ_luma || !geo_d0 || !out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  int rc = rbt::pcc_reconstruct(ctx->last_err, atlas, patches, n_patches, occ_luma, geo_d0, geo_d1, geo_bit_depth, attr_t0, attr_t1, attr_bit_depth, out);
  if (rc) rbt_cloud_free(out);
  return rc;
} RBT_CATCH
void rbt_cloud_free(rbt_cloud* c) { if (!c) return; free(c->xyz); free(c->yuv); free(c->occupancy_map); free(c->block_to_patch); memset(c, 0, sizeof(*c)); }
int rbt_d1(rbt_ctx* ctx, const int16_t* xyz_a, int n_a, const int16_t* xyz_b, int n_b, int peak, rbt_d1_result* out) try {
  if (!ctx || !xyz_a || !xyz_b || !out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return rbt::pcc_d1(ctx->last_err, xyz_a, n_a, xyz_b, n_b, peak, out);
} RBT_CATCH
int rbt_d2(rbt_ctx* ctx, const int16_t* xyz_a, const int16_t* normals_a, int n_a, const int16_t* xyz_b, int n_b, int peak, rbt_d2_result* out) try {
  if (!ctx || !xyz_a || !normals_a || !xyz_b || !out) return RBT_ERR_PARAM;
  RBT_ENTER(ctx);
  return rbt::pcc_d2(ctx->last_err, xyz_a, normals_a, n_a, xyz_b, n_b, peak, out);
} RBT_CATCH

}  // extern "C"
