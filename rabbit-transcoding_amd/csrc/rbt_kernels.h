// Launch interface between the host orchestrator (host/*.cpp) and the kernels. The product implementation is
// rbt_kernels.hip (HIP, gfx950). tests/hostemu/ provides a serial stand-in of the same interface for debugging the
// kernel bodies in a GPU-less container; it is never linked into librbt.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include "rbt_types.h"
#include "../../include/rbt.h"

struct RbtPccParams;
struct RbtSmooth;
namespace rbtk {
int dev_init(int device);                 // 0 = ok; creates the device's streams / events on first use and selects the device for this thread
int dev_select(int device);               // makes an initialised device the calling thread's current one (every C-ABI entry point calls it)
// Independent sub-bitstreams run on separate HIP streams so that the short pipelines (occupancy, geometry) overlap the
// long entropy-decoding chain of the attribute stream. All calls below act on the currently selected stream.
// The HIP runtime multiplexes streams onto 4 hardware queues by default; two streams sharing a queue serialise, and a copy
// from pageable memory blocks the host until its queue has drained (measured: 285 ms). A job with four streams uses one per
// sub-bitstream pipeline, and the last one is the auxiliary stream of the longest pipeline.
// Several GOFs can be in flight (rbt_submit_gof). Host code names streams by "lane" = job slot * 4 + pipeline (timers are kept
// per lane); map_lane binds a lane to one of the 16 HIP streams. dev_init asks the HIP runtime for 16 hardware queues
// (GPU_MAX_HW_QUEUES) when it is the first HIP user of the process. Measured on MI355X: with 24 / 32 queues a lone GOF takes
// 342 / 441 ms instead of 276 ms (the queues are time-sliced), so 16 it is; deeper pipelines give each job fewer streams.
enum { RBT_AUX_STREAM = 3, RBT_STREAMS_PER_JOB = 4, RBT_JOB_SLOTS = 16, RBT_N_LANES = RBT_STREAMS_PER_JOB * RBT_JOB_SLOTS, RBT_N_STREAMS = 16 };
void map_lane(int lane, int stream);
void set_stream(int i);
void stream_wait(int waiter, int signaller);
int stream_mark(int signaller);                // remembers the point reached on `signaller`; stream_wait_mark makes later work of `waiter` start after it
void stream_wait_mark(int waiter, int mark);   // work enqueued on `waiter` from now on starts after everything enqueued on `signaller` so far
const char* dev_name();
void* dev_alloc(size_t n);                // nullptr on failure
void dev_free(void* p);                   // returns the block to a recycling pool
void dev_release_pool();                  // hands pooled blocks back to the driver (rbt_destroy)
size_t dev_reserve_bytes();               // HBM a new arena must leave free for the runtime's own allocations (RBT_HBM_RESERVE_MB, read once)
// free / total: the driver's view (hipMemGetInfo); cached: blocks the recycling pool holds (given back before an allocation fails); live: blocks handed out
int dev_mem_info(size_t* free_b, size_t* total_b, size_t* cached_b, size_t* live_b);
size_t dev_alloc_total();                 // bytes dev_alloc has handed out on this thread so far (a job's footprint = the difference around its build)
int h2d(void* d, const void* h, size_t n);
int d2h(void* h, const void* d, size_t n);
int dev_memset(void* d, int v, size_t n);
int dev_sync();                           // returns non-zero on a device error
void timer_begin(int id);                 // GPU-side timers on the launch stream (hipEvent)
void timer_end(int id);
double timer_ms(int id);                  // valid after dev_sync()

// decode
// save == nullptr: every slice is parsed to its end; else resumable (one RbtParseSave of parse_save_bytes() per slice of the
// batch, zero-initialised): each launch advances every unfinished slice up to CTB row row_limit
// pictures of several batches on one wavefront (every anti-diagonal is one launch over all of them)
void launch_recon_refs(const RbtFrameRef* refs, int n_frames, int max_w_ctb, int max_h_ctb);
// One launch per dependency level instead of one per anti-diagonal: every workgroup takes a ticket (atomic counter, *ticket zeroed beforehand), tickets walk the
// pictures' CTBs in dependency order, a CTB waits for the done flags of its left and above-right neighbours (RbtFrame::ctb_done, zeroed beforehand).
// max_ctbs = CTBs of the largest picture. Tickets are handed out in start order, so every flag a workgroup waits for belongs to one that already runs.
void launch_recon_level(const RbtFrameRef* refs, int n_frames, int max_ctbs, uint32_t* ticket);
// One launch per dependency level with a device-side READY queue: persistent workgroups take the next ready CTB, a finished CTB counts itself in at its successors
// (RbtFrame::ctb_done[2 * addr], zeroed beforehand) and queues those that are complete. qmem: recon_queue_words(total) zeroed 32-bit words; total = CTBs of all pictures of
// the launch (each picture < 2^18 CTBs, n_frames < 8192); n_wgs workgroups (0: chosen from the pictures).
void launch_recon_queue(const RbtFrameRef* refs, int n_frames, uint32_t total_ctbs, uint32_t* qmem, int n_wgs);
inline size_t recon_queue_words(size_t total_ctbs) { return 16 + total_ctbs; }
// slices of several batches in one launch (pipelines that share a HIP stream: their parsers then run side by side)
void launch_parse_tasks(const RbtParseTask* tasks, int n_tasks, int max_w4, uint32_t* ticket = nullptr);   // ticket: a zeroed word; hands the list out in start order (needed when it holds row tasks)
// max_w4: width of the widest picture of the launch in 4-sample units (<= 2048; selects the LDS footprint of the parser)
void launch_parse(RbtFrame* frames, RbtSlice* slices, const uint8_t* rbsp, const int32_t* slice_list, int n_slices, int max_w4, void* save = nullptr, int row_limit = 0, uint32_t* ticket = nullptr);
size_t parse_save_bytes();
void launch_recon(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int y_begin = 0, int y_end = 1 << 30);
void launch_deblock(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_units);
void launch_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_luma_samples);
// SAO with one workgroup per CTB (rbt_sao_ctb: the CTB's parameters read once, neighbourhood tests only on the CTB's border); max_ctbs: CTBs of the largest picture
void launch_sao_ctb(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs);
// deblocking + SAO in one launch through LDS tiles (pictures whose out planes are not their pix planes: those with SAO); max_w / max_h: the largest picture of the list
void launch_loopfilter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w, int max_h);

// encode (RBT-E1)
void launch_pool(const uint16_t* in, int stride, int w, int h, int factor, uint16_t* out, uint16_t* out_cb, uint16_t* out_cr, int chroma_value);   // w x h region of a plane with row stride `stride`
// n pictures of one size, outputs out + k * out_step (Y, then Cb, Cr behind it), inputs anywhere
void launch_pool_many(const uint16_t* const* in, int n, int stride, int w, int h, int factor, uint16_t* out, size_t out_step, int chroma_value);
// occupancy-aware coding: RbtFrame::occ4 maps (w4 x h4 bytes each, back to back) of n occupancy pictures (ow x oh luma samples, row stride ow, in_step samples apart) for W x H pictures
void launch_occ_units(const uint16_t* occ, size_t in_step, int n, int ow, int oh, int W, int w4, int h4, uint8_t* out);
void launch_pad(const uint16_t* in, int stride, int x0, int y0, int w, int h, uint16_t* out, int dw, int dh);
void launch_enc_analyse(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs);
// row_mode 1: every slice is one CTB row, rows are independent and each wave walks its row; 2: wavefront mode (rows of a picture wait for each other, k_enc_intra_wave);
// otherwise CTBs are scheduled on anti-diagonals like the decoder's reconstruction
void launch_enc_intra(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int row_mode, int max_log2_ctb, uint32_t* ticket = nullptr);
void launch_enc_inter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs);
// SAO parameters from source-vs-reconstruction statistics and their application for every CTB of the listed pictures (en_sao_ctb); deblock_inside: the pictures have not
// been through launch_deblock, the kernel deblocks each CTB and its halo in LDS first (RBT_FUSED_ENC_LF=1)
void launch_enc_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs, int max_log2_ctb, int deblock_inside);
void launch_entropy(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* slice_list, int n_slices, int max_log2_ctb);
// wavefront mode: every slice segment (one per CTB row) of the listed pictures; rows of a picture hand their context variables down (rbt_kernels.hip k_entropy_wave)
void set_jobs_in_flight(int depth);   // hint: how many transcode jobs the caller keeps in flight on the selected device (sizes the wavefront launches); kept per device
int jobs_in_flight();                 // ... as last announced for the selected device
void launch_entropy_wave(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int max_log2_ctb, uint32_t* ticket);
// gathers the slice data of every slice segment into one contiguous buffer (dst_off = exclusive prefix sum of out_size)
void launch_pack(const uint8_t* out, const RbtSlice* slices, const uint32_t* dst_off, uint8_t* packed, int n_slices);
// matrix-core transform stages against the vector-ALU stages on n blocks of 32 x 32 int16 (0 = ok; *n_bad = differing samples)
int selftest_transform32(const int16_t* blocks, int n, int bd, uint32_t* n_bad);
// verification stage (rbt_pcc.h). items: one (patch << 16 | block inside the patch) word per patch block, in the reference's visiting order
void launch_pcc_occmap(const RbtPccParams* P, const uint16_t* occ, uint8_t* om);
void launch_pcc_owner(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, uint32_t* b2p);
void launch_pcc_count(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint32_t* b2p, uint32_t* counts);
void launch_scan_u32(const uint32_t* in, uint32_t* out, int n);      // out[i] = sum of in[0..i), out[n] = total
void launch_pcc_emit(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint16_t* t0, const uint16_t* t1,
                     const uint32_t* b2p, const uint32_t* offsets, int16_t* xyz, uint16_t* yuv, const uint8_t* om = nullptr, uint32_t* meta = nullptr);   // meta (with om): per point, patch index | boundary point << 31
// geometry smoothing (rbt_pcc.h RbtSmooth): the largest coordinate of the cloud (*out zeroed beforehand), then the three passes mark / accum / filter over dense cell arrays
void launch_sm_max(const int16_t* xyz, int n_points, uint32_t* out);
void launch_sm_passes(const RbtSmooth* G, int16_t* xyz, const uint32_t* meta);
void launch_vol_set(const int16_t* xyz, int n, uint32_t* vol, uint8_t* first, uint32_t* n_unique);
void launch_vol_nn(const int16_t* xyz, const uint8_t* first, int n, const uint32_t* vol_other, unsigned long long* sse, uint32_t* max_d2);
// D2 (csrc/rbt_pcc.h): bit volume + voxel -> lowest index map of a cloud; normals of the reconstruction (sum, count) from the source's; point-to-plane sums
void launch_d2_insert(const int16_t* xyz, int n, uint32_t* vol, uint32_t* keys, uint32_t* vals, int lg);
void launch_d2_give(const RbtD2Set* A, const int16_t* normals_a, const RbtD2Set* B, long long* acc_b, int32_t* cnt_b);
void launch_d2_take(const RbtD2Set* B, const RbtD2Set* A, const int16_t* normals_a, long long* acc_b, int32_t* cnt_b);
void launch_d2_dist(const RbtD2Set* P, const RbtD2Set* Q, const long long* acc_q, const int32_t* cnt_q, const int16_t* normals_q, double* out);
}  // namespace rbtk
