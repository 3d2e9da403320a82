/* rbt.h — C ABI of librbt.so, the MI355X-native replacement of the V-PCC transcoding hot path.
 *
 * Drop-in boundary: the reference's
 *     void PCCTranscoder::transcodeVideo(PCCVideoBitstream& videoBitstream, PCCVideoType type)
 *     (source/lib/PccLibTranscoder/include/PCCTranscoder.h:105, source/PCCTranscoder.cpp:374-546)
 * which decodes one HEVC sub-bitstream of a GOF with libavcodec, OR-pools the occupancy luma plane when
 * occupancyPrecision == 4 (resize_frame2, PCCTranscoder.cpp:594-646) and re-encodes it with libx265
 * (setEncoderOptions :825-904, encodeVideo :548-592). rbt_transcode_substream() has the same contract on plain
 * pointers: Annex-B in, Annex-B out, parameters from PCCTranscoderParameters (PCCTranscoderParameters.h:58-80).
 * INTEGRATION.md shows the ten-line patch of transcodeVideo that calls it.
 *
 * Everything runs on the GPU selected at rbt_create(); there is no CPU fallback: every entry point returns
 * RBT_ERR_NO_DEVICE if no HIP device is usable.
 */
#ifndef RBT_H
#define RBT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rbt_ctx rbt_ctx;

enum {
  RBT_OK = 0,
  RBT_ERR_NO_DEVICE = -1,     /* no usable HIP device / HIP runtime error */
  RBT_ERR_BITSTREAM = -2,     /* corrupt or truncated input */
  RBT_ERR_UNSUPPORTED = -3,   /* stream uses a tool outside the V-PCC CTC toolset (B slices, tiles, PCM, ...) */
  RBT_ERR_PARAM = -4,
  RBT_ERR_NOMEM = -5,
  RBT_ERR_MD5 = -6,           /* decoded picture hash SEI mismatch on the input stream */
  RBT_ERR_BUSY = -7           /* rbt_submit_gof: the announced number of transcodes is already in flight */
};

/* PCCVideoType values the reference passes (PCCBitstreamCommon.h:79-118) */
enum { RBT_VIDEO_OCCUPANCY = 0, RBT_VIDEO_GEOMETRY = 1, RBT_VIDEO_ATTRIBUTE = 19 };

typedef struct {
  int video_type;            /* RBT_VIDEO_* (transcodeVideo's `type`) */
  int qp;                    /* geometryQP_ / attributeQP_ / occupancyMapQP_ (PCCTranscoderParameters.h:58-80) */
  int occupancy_precision;   /* occupancyPrecision_: 4 => 2x2 OR-pool of the occupancy map (PCCTranscoder.cpp:466) */
  int log2_ctb;              /* encoder CTB size, 0 = default (5) */
  int ctb_rows_per_slice;    /* encoder slice structure: n > 0 = independent slices of n CTB rows; 0 = one slice per picture; -1 = wavefront mode, one slice per
                              * picture coded as one dependent slice segment per CTB row with entropy_coding_sync (rows predict from each other and inherit
                              * context variables: ~11 % fewer bytes than 1 at the same PSNR; what the CTC rate points use, gof_shard.DEFAULT_ROWS).
                              * All streams of one call must agree on wavefront mode or not. */
  int md5_sei;               /* emit decoded-picture-hash SEI in the output (the reconstructed pictures come to the host and are hashed on its cores: +35 % on a blocking GOF) */
  int verify_md5;            /* check the input stream's MD5 SEI (a device-to-host copy of every picture, hashed on the host's cores: +25 % on a blocking GOF) */
  int occupancy_rd;          /* geometry / attribute streams handed to rbt_transcode_gof / rbt_submit_gof behind an occupancy stream that is transcoded in the same call
                              * (occupancy_precision 4): occupancy-aware coding (what dependencies/hm-modification/HM-16.20+SCM-8.8_with_RDO.patch does to
                              * HM's distortion, TComRdCost.cpp xGetSSE*). The occupancy map the output carries tells which 4x4 units the decoder makes points of; with one unit
                              * of margin around them, transform blocks outside carry no residual, partly occupied blocks code what their occupied samples ask for, and the
                              * unoccupied samples stay out of the encoder's distortion terms. Measured on the benchmark GOF at R3: 67 % fewer geometry and 30 % fewer
                              * attribute bytes, mean D1 over the GOF's four base atlases -0.08 dB (67.14 -> 67.06 dB; a single frame scatters by +-0.3 dB either way, frame 0: -0.05 dB), PSNR of the occupied samples -0.04 / -0.02 dB (tests/test_gpu_transcode.py holds both within 0.1 dB). The pictures outside the occupied area are then whatever
                              * prediction leaves there. Entries come GOF by GOF, occupancy first; ignored where the call holds no such occupancy stream, by
                              * rbt_transcode_substream (one stream) and for lossless streams; not together with verify_md5 (RBT_ERR_PARAM). 0 = off: every sample counts. */
  int preset;                /* RBT_PRESET_*: what the reference's `preset` (PCCTranscoderParameters.h:58, handed to libx265 at PCCTranscoder.cpp:877,883) selects here.
                              * RBT_PRESET_DEFAULT (0, x265 "veryfast" and slower): every decision tool of RBT-E1. RBT_PRESET_FAST (1, "ultrafast", "superfast"):
                              * the open-loop decisions only - no SATD block costs, no closed-loop mode choice, no coded mode trial, fixed rounding: 16 % more bytes at the same
                              * QP (benchmark GOF: out / in 0.382 instead of 0.329, D1 67.80 instead of 67.94 dB), 2-3 % faster (805 against 779 point-cloud frames/s for a
                              * 20-GOF run).
                              * rbt_preset_from_name maps the reference's strings. */
} rbt_stream_params;
enum { RBT_PRESET_DEFAULT = 0, RBT_PRESET_FAST = 1 };

typedef struct {             /* decoded video returned by rbt_decode (host memory, rbt_free) */
  int width, height, bit_depth, n_frames;
  uint16_t* data;            /* n_frames x planar 4:2:0: Y (w*h), Cb, Cr */
  int md5_checked, md5_failed;
} rbt_video;

typedef struct {             /* timings of the last call, milliseconds */
  double host_parse_ms, h2d_ms, gpu_ms, d2h_ms, host_pack_ms, total_ms;
  double k_parse_ms, k_recon_ms, k_filter_ms, k_analyse_ms, k_encode_ms, k_entropy_ms;
  uint64_t algorithmic_bytes; /* SURVEY.md 8(d) accounting of the call */
} rbt_stats;

/* One context per host thread and GPU. `device` is the HIP device the context's work runs on; streams, timers, job slots and the
 * recycling pool of device memory are per device, so contexts on different devices may be used concurrently from different threads
 * (contexts on the same device share its 16 HIP streams and job slots and are serialised by the library).
 * world_rank / world_size describe the multi-GPU job the context is part of (one process per GPU): the library owns
 * the GOF sharding rule (rbt_owns_gof), the NAL gather itself runs in the host program over RCCL. 0 <= world_rank < world_size. */
int rbt_create(rbt_ctx** ctx, int device, int world_rank, int world_size);
/* GOF g of a sequence is transcoded by rank g mod world_size; a host that walks the sequence GOF by GOF (PccAppTranscoder.cpp:307-341)
 * on every rank skips what its context does not own. Returns 1 / 0. */
int rbt_owns_gof(const rbt_ctx* ctx, int gof_index);
int rbt_world(const rbt_ctx* ctx, int* world_rank, int* world_size);
void rbt_destroy(rbt_ctx* ctx);
const char* rbt_strerror(int code);
const char* rbt_last_error(rbt_ctx* ctx);   /* what the LAST call on this context had to say if it failed (e.g. which syntax element it rejected); "" if it succeeded or gave no
                                              * detail: every entry point that takes the context starts by clearing the text. The pointer is valid until the next call on the context. */
void rbt_free(void* p);
const char* rbt_version(void);

/* transcodeVideo: one Annex-B HEVC sub-bitstream of a GOF -> re-encoded Annex-B stream (malloc'd, rbt_free). */
int rbt_transcode_substream(rbt_ctx* ctx, const uint8_t* annexb_in, size_t n_in, const rbt_stream_params* p, uint8_t** annexb_out, size_t* n_out);

/* transcodeData (PCCTranscoder.cpp:145-168): the sub-bitstreams of one GOF in one call, each as its own pipeline on its own
 * HIP stream. As in the reference (:150), an occupancy stream is only transcoded when occupancy_precision == 4; with any other
 * precision it is returned unchanged (rbt_transcode_substream, like transcodeVideo, re-encodes whatever it is handed).
 * Entries that name the same input buffer (same pointer and size) with different parameters are decoded once and re-encoded once
 * per entry: the rate fan-out of BASELINE.json configs[4] (R1..R5 from one R5 input) on one GPU. More than three streams (the sub-bitstreams of several GOFs: one GOF leaves most of an MI355X idle) are grouped
 * by video type into three pipelines. Outputs come back in input order. */
#define RBT_MAX_STREAMS 96
int rbt_transcode_gof(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, uint8_t** annexb_out, size_t* n_out);

/* rbt_transcode_gof in two halves, for the caller that walks a sequence GOF by GOF (PccAppTranscoder.cpp:307-341 calls
 * transcode -> transcodeData, PCCTranscoder.cpp:66-70 / :145-168, once per GOF): submit builds the batch and enqueues every
 * kernel of the GOF, wait collects the streams.
 * Several transcodes may be in flight; they use disjoint HIP streams, so the entropy decoding of GOF i+1 (a few hundred lone
 * waves) runs underneath the entropy decoding, reconstruction and re-encode of GOF i. The input buffers may be released as soon
 * as submit returns. Jobs may be waited for in any order; every submitted job must be waited for (rbt_destroy drains what is
 * left). Results are identical to rbt_transcode_gof's. Calls on one device are serialised by the library (a wait blocks a
 * concurrent submit from another thread on the same GPU).
 * rbt_set_depth announces how many jobs the caller keeps in flight (1..RBT_MAX_JOBS, default 4; per device; refused with
 * RBT_ERR_BUSY while jobs are in flight): the library has 16 HIP streams (more hardware queues slow every queue down on
 * MI355X), so up to 4 jobs get four streams each, 5 get three, up to 8 two, up to 16 one (pipelines that share a stream run
 * their entropy decoding and their reconstruction in merged launches). rbt_submit_gof returns RBT_ERR_BUSY when that many
 * jobs are already in flight. */
#define RBT_MAX_JOBS 16
typedef struct rbt_job rbt_job;
int rbt_set_depth(rbt_ctx* ctx, int max_in_flight);
int rbt_get_depth(rbt_ctx* ctx);   /* the announced depth (> 0) or RBT_ERR_PARAM */
/* How to cut a walk of n_gofs GOFs into jobs on one GPU, as measured on 1280x1280 maps: 16 jobs of 3 GOFs for a long walk (96 GOFs and more; 2 GOFs from 48: the
 * arenas of 48 GOFs in flight are 198 GB at that size (4.12 GB per GOF since round 4, rbt_job_memory; 64 in flight fit too and run no faster). The shape looks at the length of the walk only; rbt_transcode_v3c bounds the jobs it keeps in flight by the memory the
 * first job took (rbt_job_memory against rbt_device_memory) and falls back to one GOF per job, one job at a time, when a job still fails with RBT_ERR_NOMEM; a caller
 * that drives rbt_submit_gof itself does the same or passes its own shape); a walk shorter than 48 GOFs is all ramp-up and
 * drain and does better as at most 7 jobs (2 jobs up to 12 GOFs) of ceil(n / jobs) GOFs, which then own several hardware queues each. max_jobs caps the jobs in flight. */
int rbt_job_shape(int n_gofs, int max_jobs, int* gofs_per_job, int* jobs_in_flight);
/* The reference's `preset` string (an x265 preset name, PCCTranscoderParameters.h:58) as RBT_PRESET_*: "ultrafast", "superfast" -> RBT_PRESET_FAST; "veryfast"
 * (what the reference's scripts pass, transcode.sh:18), "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo" and NULL / "" -> RBT_PRESET_DEFAULT (even x265's
 * "veryfast" decides by rate-distortion cost, which RBT_PRESET_FAST does not); anything else -> RBT_ERR_PARAM. */
int rbt_preset_from_name(const char* name);
int rbt_submit_gof(rbt_ctx* ctx, int n, const uint8_t* const* annexb_in, const size_t* n_in, const rbt_stream_params* p, rbt_job** job);
/* Device memory as the library sees it, for callers that size their own pipelines (rbt_transcode_v3c does: jobs in flight are bounded by it). free_bytes / total_bytes: the
 * driver's figures; cached_bytes: arenas of collected jobs kept for the next job of the same shape (handed back before an allocation fails; rbt_trim); in_use_bytes: arenas
 * of jobs in flight; reserve_bytes: what a new arena leaves free for the HIP runtime's own allocations (scratch memory of the hardware queues: 3 GB unless the environment
 * variable RBT_HBM_RESERVE_MB says otherwise) - an arena that would cut into it fails with RBT_ERR_NOMEM and the context stays usable. */
typedef struct { size_t total_bytes, free_bytes, cached_bytes, in_use_bytes, reserve_bytes; } rbt_memory;
int rbt_device_memory(rbt_ctx* ctx, rbt_memory* out);
/* Device memory a submitted job holds (its decoder and encoder arenas); what the next job of the same shape will take. */
int rbt_job_memory(rbt_ctx* ctx, const rbt_job* job, size_t* bytes);
int rbt_wait_gof(rbt_ctx* ctx, rbt_job* job, uint8_t** annexb_out, size_t* n_out);
/* The library keeps the device memory of collected jobs for the next job of the same shape (hipMalloc / hipFree of GOF-sized arenas cost milliseconds
 * and hipFree drains the device). rbt_trim hands that cache back to the driver: worth calling when the workload changes shape (other picture sizes, other
 * numbers of streams per job), so that the old shapes do not crowd the 288 GB. RBT_ERR_BUSY while jobs are in flight on the device. */
int rbt_trim(rbt_ctx* ctx);

/* The two halves exposed on their own (SURVEY.md 8(b) alternative seam; used by the parity tests).
 * Picture sizes: any even width / height. Sizes that are not multiples of 8 (all-intra) / 16 (gop 2) are coded padded with a
 * conformance window in the SPS (as libx265 does for the reference); rbt_decode returns the cropped pictures. */
int rbt_decode(rbt_ctx* ctx, const uint8_t* annexb, size_t n, int verify_md5, rbt_video* out);
int rbt_encode(rbt_ctx* ctx, const uint16_t* yuv, int width, int height, int bit_depth, int n_frames, int qp, int gop, int lossless,
               int log2_ctb, int ctb_rows_per_slice, int md5_sei, uint8_t** annexb_out, size_t* n_out);

/* resize_frame2 (PCCTranscoder.cpp:594-646) on a host plane (tests): out[v][u] = any(in block > 0) */
int rbt_or_pool(rbt_ctx* ctx, const uint16_t* plane, int width, int height, int factor, uint16_t* out);

/* PCCVideoBitstream::sampleStreamToByteStream / byteStreamToSampleStream (PCCVideoBitstream.cpp:85-172), host side */
int rbt_sample_to_byte_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out);
int rbt_byte_to_sample_stream(const uint8_t* in, size_t n, uint8_t** out, size_t* n_out);

int rbt_get_stats(rbt_ctx* ctx, rbt_stats* out);
/* Device self-test of the 32-point transform stages on the matrix cores (v_mfma_i32_32x32x32_i8) against the vector-ALU form of the same stages:
 * n_blocks blocks of 32 x 32 int16 (coefficients for the inverse, residuals for the forward transform); *n_mismatch = differing output samples. */
int rbt_selftest_transform32(rbt_ctx* ctx, const int16_t* blocks, int n_blocks, int bit_depth, uint32_t* n_mismatch);

/* ---- decoder-side verification stage: what turns transcoded maps into the D1 figure of the metric ----
 * Replaces PCCCodec::generateOccupancyMap (PCCCodec.cpp:1584-1606), generateBlockToPatchFromOccupancyMapVideo (:1725-1763),
 * generatePointCloud / generatePoints (:517-978, :327-515) and the colour fetch of colorPointCloud (:1308-1449) for the configuration of the
 * CTC streams (two geometry maps with absolute D1; no EOM, raw patches, point-local reconstruction or patch border filtering; one tile),
 * and the point-to-point part of QualityMetrics::compute (PCCMetrics.cpp:75-231, :44-48, :299-309). */
typedef struct {             /* the fields of PCCPatch the reconstruction reads (PCCPatch.h) */
  int32_t u0, v0, size_u0, size_v0;      /* position / size in the atlas in occupancy-resolution blocks */
  int32_t u1, v1, d1;                    /* 3-D offset along the tangent, bitangent and normal axis */
  int32_t normal_axis, tangent_axis, bitangent_axis;   /* 0..2, all different */
  int32_t projection_mode;               /* 0: depth + d1, 1: d1 - depth */
  int32_t orientation;                   /* PATCH_ORIENTATION_* (PCCCommon.h:128-137) */
  int32_t lod_x, lod_y;
} rbt_patch;
typedef struct {
  int32_t width, height;                 /* atlas frame size (multiples of occupancy_resolution) */
  int32_t occupancy_resolution;          /* 16 in the CTC (cfg/common/ctc-common.cfg) */
  int32_t occupancy_precision;           /* atlas size / occupancy video size: 2 at R5, 4 after the transcoder's OR-pool */
  int32_t map_count, absolute_d1, remove_duplicate_points, threshold_lossy_om;
  /* Geometry smoothing of the decoder's post-processing, which the CTC switches on (cfg/common/ctc-common.cfg:57-60: flagGeometrySmoothing 1, gridSmoothing 1, gridSize 8,
   * thresholdSmoothing 64; PCCDecoder.cpp:434-437 -> PCCCodec::smoothPointCloudPostprocess, PCCCodec.cpp:52-145, :980-1104): boundary points (identifyBoundaryPoints, :266-325)
   * next to cells that hold points of more than one patch move towards the tri-linear blend of the 8 surrounding cell centroids. geometry_smoothing 0 = off (the cloud as
   * generatePointCloud leaves it); grid_size 2..255, even (the reference's cfg uses 8). Exact for cells of fewer than 16384 points (the reference sums centroids in float). */
  int32_t geometry_smoothing, grid_size, threshold_smoothing;
} rbt_atlas_params;
typedef struct {             /* host memory, released with rbt_cloud_free */
  int n_points;
  int16_t* xyz;              /* 3 per point, in the order PCCCodec::generatePointCloud emits them */
  uint16_t* yuv;             /* 3 per point: the attribute samples at the point's pixel (chroma at the co-sited 4:2:0 sample; the reference
                                converts to 4:4:4 first, PccLibColorConverter, out of scope) */
  uint8_t* occupancy_map;    /* width x height: the up-scaled, binarised occupancy map */
  uint32_t* block_to_patch;  /* (width / res) x (height / res): patch index + 1 */
  int n_smoothed;            /* points the geometry smoothing moved */
} rbt_cloud;
/* occ_luma: occupancy video luma, (width / precision) x (height / precision); geo_d0 / geo_d1: luma of the near / far geometry map, width x
 * height samples of geo_bit_depth bits (mapped to 8 bits like PCCImage::set, PCCImage.h:107-124); attr_t0 / attr_t1: planar 4:2:0 attribute
 * pictures or NULL. */
int rbt_reconstruct(rbt_ctx* ctx, const rbt_atlas_params* atlas, const rbt_patch* patches, int n_patches, const uint16_t* occ_luma, const uint16_t* geo_d0,
                    const uint16_t* geo_d1, int geo_bit_depth, const uint16_t* attr_t0, const uint16_t* attr_t1, int attr_bit_depth, rbt_cloud* out);
void rbt_cloud_free(rbt_cloud* c);
typedef struct {
  int n_a, n_b;                          /* points after merging duplicates (PCCMetricsParameters.cpp:50) */
  uint64_t sse_ab, sse_ba, max_ab, max_ba;   /* sum / maximum of squared nearest-neighbour distances, A -> B and B -> A (exact integers) */
  float mse_ab, mse_ba, psnr_ab, psnr_ba, psnr;   /* psnr = 10 log10(3 peak^2 / max(mse_ab, mse_ba)) (PCCMetrics.cpp:44-48, :299-309) */
} rbt_d1_result;
/* point-to-point (D1) metric between two clouds; coordinates 0..1023 (peak = 1023 in the CTC). */
int rbt_d1(rbt_ctx* ctx, const int16_t* xyz_a, int n_a, const int16_t* xyz_b, int n_b, int peak, rbt_d1_result* out);
/* point-to-plane (D2) metric, the other half of BASELINE's "D1/D2 geom PSNR": QualityMetrics::compute with computeC2p_ (PCCMetrics.cpp:100-124, :213-215), symmetric (:299-309),
 * between a source cloud A that comes with normals (three per point, fixed point Q14: 16384 = 1.0) and a decoded cloud B that gets its normals from A the way
 * PCCMetrics::compute arranges it (:371-376: copyNormals on the source, scaleNormals on the reconstruction, PCCPointSet.cpp:2322-2380: every source point gives its normal to
 * the points of B nearest to it, a point of B that got none takes the mean of the source points nearest to it). Per point of one cloud: the mean, over the other cloud's points
 * at the nearest distance, of the squared projection of the difference on that point's normal. Where the reference depends on the order its kd-tree returns equidistant points
 * in, this is defined instead: duplicates are merged first and a merged point keeps the normal of its lowest-index duplicate; ALL points at exactly the nearest squared
 * distance count (the reference looks at up to 30 results). */
typedef struct {
  int n_a, n_b;
  double sse_ab, sse_ba, max_ab, max_ba;              /* sum / maximum of the per-point values, A -> B and B -> A */
  float mse_ab, mse_ba, psnr_ab, psnr_ba, psnr;       /* psnr = 10 log10(3 peak^2 / max(mse_ab, mse_ba)) */
} rbt_d2_result;
int rbt_d2(rbt_ctx* ctx, const int16_t* xyz_a, const int16_t* normals_a, int n_a, const int16_t* xyz_b, int n_b, int peak, rbt_d2_result* out);

/* ---- V3C sample stream: the container either side of the path ----
 * What PccAppTranscoder's decompressVideo does around transcodeData (PccAppTranscoder.cpp:277-349): read the sample stream (PCCBitstreamReader::read,
 * PCCBitstreamReader.cpp:51-70, C.2: one header byte with the unit size precision, then size + unit), walk it GOF by GOF (a GOF starts at each
 * V3C_VPS unit, :72-96), transcode the video units of every GOF, collect the units of all GOFs (PCCBitstreamWriter::encode, PCCBitstreamWriter.cpp:96-237) and
 * write them as ONE sample stream whose unit size precision follows the largest unit (PCCBitstreamWriter::write, :57-91).
 * The library does not parse the V3C parameter set or the atlas sub-bitstream: the reference reads both into its context and writes them back
 * (with the end-of-tile patch type its reader dropped put back, PCCTranscoder::addEndTile :906-914), which for a stream its own writer made reproduces
 * the bytes; here V3C_VPS and V3C_AD units are copied. The payload of an OVD / GVD / AVD unit is the video sub-bitstream in sample stream form
 * (PCCBitstream::readVideoStream, PCCBitstream.cpp:88-97; unit size - 4, PCCBitstreamReader.cpp:225). */
enum { RBT_V3C_VPS = 0, RBT_V3C_AD = 1, RBT_V3C_OVD = 2, RBT_V3C_GVD = 3, RBT_V3C_AVD = 4 };   /* V3CUnitType, PCCBitstreamCommon.h:133-137 */
typedef struct {
  int type;                  /* vuh_unit_type (first 5 bits of the unit, PCCBitstreamReader.cpp:1381-1383) */
  int gof;                   /* GOF the unit belongs to (units in front of the first V3C_VPS: 0) */
  int parameter_set_id, atlas_id;                     /* v3cUnitHeader, PCCBitstreamReader.cpp:182-211; 0 where the unit type has none */
  int attribute_index, attribute_dimension_index, map_index, auxiliary_video;
  int video_type;            /* RBT_VIDEO_* the unit's sub-bitstream is filed under where transcodeData looks for it (videoSubStream, :98-158): OVD -> occupancy,
                                GVD without auxiliary video -> geometry, AVD without auxiliary video, partition 0 -> attribute; -1 for every other unit */
  size_t offset, size;       /* the unit (4-byte header + payload) inside the input */
} rbt_v3c_unit;
/* Lists the units of a sample stream (host only, no GPU needed; *units is malloc'd, rbt_free). RBT_ERR_BITSTREAM if a unit overruns the input. */
int rbt_v3c_index(const uint8_t* in, size_t n, rbt_v3c_unit** units, int* n_units);
/* PCCBitstreamWriter::write (:57-91) + sampleStreamV3CHeader / sampleStreamV3CUnit (:1492-1507): precision = min(max(ceil(ceilLog2(largest unit) / 8), 1), 8)
 * bytes, at least forced_precision_bytes (forcedSsvhUnitSizePrecisionBytes_); then every unit behind its size. Host only. */
int rbt_v3c_write(const uint8_t* const* unit, const size_t* unit_size, int n_units, int forced_precision_bytes, uint8_t** out, size_t* n_out);
/* PCCBitstreamStat as decompressVideo prints it for the input and the output file (PccAppTranscoder.cpp:351-352, PCCBitstream.h:48-154), from the bytes alone. Host only. */
typedef struct {
  int n_units, n_gofs, unit_size_precision_bytes;
  uint64_t header;                                    /* sample stream header + the size fields of all units (PCCBitstreamReader.cpp:51-70) */
  uint64_t unit_size[5];                              /* V3CUnitSize[type]: unit headers + payloads */
  uint64_t occupancy_video, geometry_video, geometry_aux_video, attribute_video, attribute_aux_video;   /* videoBinSize: payloads of the video units */
  uint64_t total_metadata, total_geometry, total_attribute, total;   /* getTotalMetadata() + header, getTotalGeometry(), getTotalAttribute(), their sum (:138-148) */
} rbt_v3c_stat;
int rbt_v3c_stats(const uint8_t* in, size_t n, rbt_v3c_stat* out);
typedef struct {
  int occupancy_precision;   /* occupancyPrecision_: the occupancy video is transcoded (2x2 OR-pool, lossless) only when 4 (PCCTranscoder.cpp:150) */
  int geometry_qp, attribute_qp;                      /* geometryQP_, attributeQP_ */
  int forced_unit_size_precision_bytes;               /* forcedSsvhUnitSizePrecisionBytes_, 0 = none */
  int log2_ctb, ctb_rows_per_slice, md5_sei, verify_md5;   /* as in rbt_stream_params */
  int gofs_per_job;          /* GOFs handed to the GPU per job; 0 = by rbt_job_shape from the number of GOFs this context owns, which also lowers the announced depth for
                              * the duration of the call when the walk is short (the depth announced with rbt_set_depth is the cap and is restored) */
  int occupancy_rd;          /* occupancy-aware coding of the geometry / attribute units of every GOF (rbt_stream_params.occupancy_rd): with the occupancy map that GOF's
                              * occupancy unit comes out with, when occupancy_precision is 4 */
  int preset;                /* RBT_PRESET_* for the geometry / attribute units (rbt_stream_params.preset) */
} rbt_v3c_params;
/* The whole walk: index, per GOF the video units through rbt_submit_gof / rbt_wait_gof with as many jobs in flight as rbt_set_depth announced (fewer for a short walk
 * with gofs_per_job = 0), write.
 * Video units transcodeData does not look at (auxiliary video, attribute partitions beyond the first) are copied; a GOF with several
 * geometry or attribute map streams (multipleMapStreamsPresentFlag) is refused (RBT_ERR_UNSUPPORTED) - the reference looks for VIDEO_GEOMETRY / VIDEO_ATTRIBUTE, which such a GOF does not have.
 * In a multi-GPU job (rbt_create with world_size > 1) the output holds the GOFs this rank owns (rbt_owns_gof) and nothing else: rank 0 of the host
 * program gathers the partial streams and merges them with rbt_v3c_index + rbt_v3c_write (gof_shard.transcode_v3c does). */
int rbt_transcode_v3c(rbt_ctx* ctx, const uint8_t* in, size_t n, const rbt_v3c_params* p, uint8_t** out, size_t* n_out);
/* The same walk with the output handed over GOF by GOF, in GOF order, as soon as the job that holds a GOF has been collected (the jobs behind it are still running): the
 * sink gets the units of one GOF this context owns - carried-over units point into `in`, transcoded ones into memory that is released when the sink returns. A sink that
 * writes a sample stream itself chooses the unit size precision up front (PCCBitstreamWriter::write derives it from the largest unit of the whole file, which a streaming
 * writer does not know yet: forcedSsvhUnitSizePrecisionBytes_ = 4 is what fits every file). A non-zero return of the sink ends the walk (RBT_ERR_PARAM; jobs in flight are drained). */
typedef int (*rbt_v3c_sink)(void* user, int gof, int n_units, const uint8_t* const* unit, const size_t* unit_size);
int rbt_transcode_v3c_stream(rbt_ctx* ctx, const uint8_t* in, size_t n, const rbt_v3c_params* p, rbt_v3c_sink sink, void* user);

#ifdef __cplusplus
}
#endif
#endif
