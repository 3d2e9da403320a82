// Launch interface between the host orchestrator (host/*.cpp) and the kernels. The product implementation is
// rbt_kernels.hip (HIP, gfx950). tests/hostemu/ provides a serial stand-in of the same interface for debugging the
// kernel bodies in a GPU-less container; it is never linked into librbt.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include "rbt_types.h"

namespace rbtk {
int dev_init(int device);                 // 0 = ok
const char* dev_name();
void* dev_alloc(size_t n);                // nullptr on failure
void dev_free(void* p);
int h2d(void* d, const void* h, size_t n);
int d2h(void* h, const void* d, size_t n);
int dev_memset(void* d, int v, size_t n);
int dev_sync();                           // returns non-zero on a device error
void timer_begin(int id);                 // GPU-side timers on the launch stream (hipEvent)
void timer_end(int id);
double timer_ms(int id);                  // valid after dev_sync()

// decode
void launch_parse(RbtFrame* frames, RbtSlice* slices, const uint8_t* rbsp, const int32_t* slice_list, int n_slices);
void launch_recon(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb);
void launch_deblock(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_units);
void launch_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_luma_samples);
}  // namespace rbtk
