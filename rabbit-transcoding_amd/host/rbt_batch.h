// Host orchestration of one call: a "batch" is every picture of the sub-bitstreams handed over together (one GOF's
// geometry + attribute + occupancy streams in rbt_transcode_gof, a single stream in rbt_transcode_substream).
// The host parses NAL units, parameter sets and slice headers, lays the batch out in one HBM arena, uploads the
// unescaped slice data once, and enqueues the kernels; pictures never leave HBM between decode and re-encode.
#pragma once
#include <string>
#include <vector>
#include "../../include/rbt.h"
#include "../csrc/rbt_kernels.h"
#include "rbt_hls.h"

namespace rbt {

struct StreamIn { const uint8_t* p; size_t n; };
struct FrameInfo { int stream; int nal_type; bool has_md5; uint8_t md5[3][16]; bool sao; };

enum { T_PARSE = 0, T_RECON = 1, T_FILTER = 2, T_ANALYSE = 3, T_ENCODE = 4, T_ENTROPY = 5, T_ALL = 6, T_POOL = 7, T_INTER = 8, T_ENTROPY_I = 9, T_COUNT = 10 };

struct Arena {           // bump allocator over one device allocation
  uint8_t* base = nullptr; size_t size = 0, used = 0;
  size_t reserve(size_t n) { size_t o = (used + 255) & ~(size_t)255; used = o + n; return o; }
};

struct DecodeBatch {
  std::vector<uint8_t> rbsp;
  std::vector<RbtFrame> frames; std::vector<RbtSlice> slices; std::vector<FrameInfo> info;
  std::vector<int> stream_first, stream_count;
  std::vector<Sps> stream_sps; std::vector<Pps> stream_pps;
  std::vector<std::vector<int>> level_frames;
  bool ordered_parse = false, lists_uploaded = false;
  std::vector<char> alias_taken;       // per stream: its pictures' dead buffers (coefficient levels, pre-SAO samples) have been handed to an encoder stream (setup_encode)
  bool recon_external = false;         // ... and reconstructed by merged per-level launches (launch_recon_refs + decode_launch_filters)
  bool parse_external = false;         // the batch's slices are parsed by a merged launch of the caller (launch_parse_tasks)
  std::vector<int32_t> lists_keep;     // host staging of the index lists, alive until the copy has completed
  std::vector<size_t> fr_off;          // offset of each level's frame list inside d_lists
  std::vector<size_t> sl_off, sl_cnt;  // slice list of each level inside d_lists
  std::vector<uint32_t> order_keep; std::vector<size_t> order_off;   // CTB dependency order per picture (host staging, offset per frame)
  std::vector<RbtFrameRef> refs_keep; std::vector<size_t> refs_off;   // the pictures of each level as RbtFrameRef (launch_recon_level)
  bool has_row_tasks = false;      // some segment of a wavefront stream is parsed by a wave of its own (ordered hand-out of the parse tasks, no banded parsing)
  uint32_t* d_queue = nullptr; std::vector<size_t> queue_off; std::vector<uint32_t> queue_total; std::vector<int> queue_wgs;   // ready queues of the levels (launch_recon_queue): offset in words, CTBs, workgroups
  uint32_t* d_order = nullptr; RbtFrameRef* d_refs = nullptr; uint32_t* d_tickets = nullptr;   // one ticket counter per level + spare ones for merged launches
  void* d_save = nullptr;              // RbtParseSave per slice (resumable parsing), zero-initialised; nullptr when not requested
  bool want_save = false;              // set before decode_build to reserve d_save
  void* arena = nullptr; size_t arena_size = 0;
  RbtFrame* d_frames = nullptr; RbtSlice* d_slices = nullptr; uint8_t* d_rbsp = nullptr; int32_t* d_lists = nullptr;
  std::string err; int err_code = 0;
  ~DecodeBatch() { rbtk::dev_free(arena); }
};

int decode_build(DecodeBatch& b, const StreamIn* streams, int n);
bool recon_by_diagonals();           // per-diagonal launches instead of the flag kernel (see rbt_decode.cpp)
int recon_mode();                    // 0 = one launch per anti-diagonal, 1 = one launch per level with neighbour flags, 2 = one launch per level with a ready queue (rbt_decode.cpp)
int recon_queue_width(const RbtStreamCfg& c);   // CTBs of a picture that can be reconstructed side by side, on average (workgroups the ready-queue launch gets per picture)
void recon_set_depth(int depth);     // jobs the caller keeps in flight (rbt_set_depth)
int decode_launch(DecodeBatch& b);   // enqueue every decode kernel of the batch on the current stream (no wait)
int decode_launch_parse(DecodeBatch& b);            // index lists + entropy decoding
int decode_upload_lists(DecodeBatch& b);            // index lists only (first half of decode_launch_parse)
int decode_max_w4(const DecodeBatch& b);            // width of the widest picture in 4-sample units
// Entropy decoding in `chunks` row bands with the reconstruction of each finished band of the level-0 pictures enqueued on
// stream `aux` underneath the parsing of the next band; ends with level 0 complete (filters included) on the current stream.
int decode_launch_chunked(DecodeBatch& b, int chunks, int main_stream, int aux_stream);
void decode_launch_level(DecodeBatch& b, size_t l);  // reconstruction + loop filters of dependency level l
void decode_launch_filters(DecodeBatch& b, size_t l); // loop filters of level l only
int decode_finish(DecodeBatch& b);   // wait for the batch's stream and check the per-picture error words
int decode_run(DecodeBatch& b);      // launch + finish
int decode_fetch(DecodeBatch& b, int stream, rbt_video* out, bool verify_md5);

size_t frame_samples(const RbtStreamCfg& c);

}  // namespace rbt
