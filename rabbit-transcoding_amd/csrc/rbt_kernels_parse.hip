// Kernels and launchers of the slice parser (rbt_parse.h). A translation unit of its own because of its optimisation level: the
// parser is the latency chain of a GOF (one lone wave per slice, 240 ms for the largest) and is compiled with -O3; everything
// else (rbt_kernels.hip) is compiled with -Os - with many GOFs in flight the kernels of different stages share the instruction
// caches, and the smaller code is worth 3 % of throughput (measured: 914 -> 942 fps) while -Os costs the parser 2.5 % of speed.
#include <hip/hip_runtime.h>
#include "rbt_kernels.h"
#include "rbt_parse.h"

namespace rbtk {
hipStream_t current_stream();            // rbt_kernels.hip: the stream the host code selected (set_stream)
#define g_stream current_stream()

// one wave per slice segment: wave-uniform CABAC parse (rbt_parse.h)
// (CAP4: capacity of the parser's line buffers in 4-sample units; the variant fixes the LDS footprint of the workgroup)
// ticket != nullptr: the list is handed to the waves in the order they start (a row task of a wavefront stream waits for the task of the CTB row above
// it, which is earlier in the list: with start order = list order the wave it waits for is always running)
__device__ __forceinline__ uint32_t parse_index(uint32_t* ticket) {
  if (!ticket) return blockIdx.x;
  int t = 0;
  if ((threadIdx.x & 63) == 0) t = (int)atomicAdd(ticket, 1u);
  return (uint32_t)__builtin_amdgcn_readfirstlane(t);
}
template <int CAP4>
__global__ void __launch_bounds__(64) k_parse(RbtFrame* frames, RbtSlice* slices, const uint8_t* rbsp, const int32_t* slice_list, RbtParseSave* save, int row_limit, uint32_t* ticket) {
  __shared__ alignas(16) uint32_t lds[(RBT_PARSE_LDS_BYTES(CAP4) + 3) / 4];
  rbt_parse_slice(frames, slices, slice_list[parse_index(ticket)], rbsp, RBT_LDS_CAST(RbtParseLds, lds), CAP4, save, row_limit);
}
// the same over slices of several batches (each task names its batch's tables)
#ifndef RBT_PARSE_TASKS_ATTR
#define RBT_PARSE_TASKS_ATTR
#endif
template <int CAP4>
__global__ void __launch_bounds__(64) RBT_PARSE_TASKS_ATTR k_parse_tasks(const RbtParseTask* tasks, uint32_t* ticket) {
  __shared__ alignas(16) uint32_t lds[(RBT_PARSE_LDS_BYTES(CAP4) + 3) / 4];
  const RbtParseTask t = tasks[parse_index(ticket)];
  rbt_parse_slice(t.frames, t.slices, t.slice, t.rbsp, RBT_LDS_CAST(RbtParseLds, lds), CAP4, nullptr, 0);
}
void launch_parse(RbtFrame* frames, RbtSlice* slices, const uint8_t* rbsp, const int32_t* slice_list, int n_slices, int max_w4, void* save, int row_limit, uint32_t* ticket) {
  if (n_slices <= 0) return;
  if (max_w4 <= RBT_PARSE_CAP4_S) hipLaunchKernelGGL(k_parse<RBT_PARSE_CAP4_S>, dim3(n_slices), dim3(64), 0, g_stream, frames, slices, rbsp, slice_list, (RbtParseSave*)save, row_limit, ticket);
  else if (max_w4 <= RBT_PARSE_CAP4_M) hipLaunchKernelGGL(k_parse<RBT_PARSE_CAP4_M>, dim3(n_slices), dim3(64), 0, g_stream, frames, slices, rbsp, slice_list, (RbtParseSave*)save, row_limit, ticket);
  else hipLaunchKernelGGL(k_parse<RBT_PARSE_CAP4_L>, dim3(n_slices), dim3(64), 0, g_stream, frames, slices, rbsp, slice_list, (RbtParseSave*)save, row_limit, ticket);
}
void launch_parse_tasks(const RbtParseTask* tasks, int n_tasks, int max_w4, uint32_t* ticket) {
  if (n_tasks <= 0) return;
  if (max_w4 <= RBT_PARSE_CAP4_S) hipLaunchKernelGGL(k_parse_tasks<RBT_PARSE_CAP4_S>, dim3(n_tasks), dim3(64), 0, g_stream, tasks, ticket);
  else if (max_w4 <= RBT_PARSE_CAP4_M) hipLaunchKernelGGL(k_parse_tasks<RBT_PARSE_CAP4_M>, dim3(n_tasks), dim3(64), 0, g_stream, tasks, ticket);
  else hipLaunchKernelGGL(k_parse_tasks<RBT_PARSE_CAP4_L>, dim3(n_tasks), dim3(64), 0, g_stream, tasks, ticket);
}
size_t parse_save_bytes() { return sizeof(RbtParseSave); }
}  // namespace rbtk
