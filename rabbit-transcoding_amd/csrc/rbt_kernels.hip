// HIP kernels + launchers for gfx950 (MI355X). See rbt_kernels.h. Kernels of a call are enqueued back-to-back on the streams of
// the calling context's device, the host synchronises once per phase that needs results.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "rbt_kernels.h"
#include "rbt_recon.h"
#include "rbt_filter.h"
#include "rbt_encode.h"
#include "rbt_pcc.h"

namespace rbtk {
// Everything the host code touches on a GPU is per DEVICE: 16 HIP streams, the timer / dependency events, the lane -> stream
// map and the recycling pool of device allocations. A host thread works on the device it selected last (dev_select, called
// by every C-ABI entry point with its context's device), so contexts on different devices run concurrently from different
// threads; contexts on the same device share that device's streams (rbt_api.cpp serialises them with the device's mutex).
enum { RBT_MAX_DEVICES = 16, RBT_POOL_KEEP = 160 };
struct PoolBlock { void* p; size_t n; };
struct Dev {
  int id = -1;
  hipStream_t streams[RBT_N_STREAMS] = {};
  hipEvent_t dep_ev[64]; int dep_next = 0;
  int map[RBT_N_LANES];                 // lane -> HIP stream
  hipEvent_t ev[RBT_N_LANES][16][2];
  char name[256] = "";
  std::vector<PoolBlock> pool_free, pool_live; std::mutex pool_mu;
  int depth = 1, wave_div = 2;          // jobs the caller keeps in flight on THIS device (set_jobs_in_flight) and what follows from it (contexts on other devices have their own)
};
static Dev* g_devs[RBT_MAX_DEVICES] = {};
static std::mutex g_devs_mu;
static thread_local Dev* t_dev = nullptr;      // device selected by this host thread
static thread_local int t_cur = 0;             // ... and its current lane
static thread_local char t_err[256] = "";
static inline int lane_of(int i) { return ((i % RBT_N_LANES) + RBT_N_LANES) % RBT_N_LANES; }
static inline hipStream_t stream_of(int lane) { return t_dev->streams[t_dev->map[lane_of(lane)]]; }
#define g_stream (t_dev->streams[t_dev->map[t_cur]])
hipStream_t current_stream() { return g_stream; }   // for rbt_kernels_parse.hip

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(t_err, sizeof t_err, "%s: %s", #x, hipGetErrorString(e_)); return -1; } } while (0)

int dev_init(int device) {
  // one hardware queue per HIP stream (default: 4 queues shared by all streams); only honoured before the runtime initialises
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { snprintf(t_err, sizeof t_err, "no HIP device"); return -1; }
  if (device < 0 || device >= n || device >= RBT_MAX_DEVICES) { snprintf(t_err, sizeof t_err, "device %d out of range (%d devices)", device, n); return -1; }
  std::lock_guard<std::mutex> lk(g_devs_mu);
  HIPCHK(hipSetDevice(device));
  if (!g_devs[device]) {
    Dev* d = new Dev(); d->id = device;
    for (int i = 0; i < RBT_N_LANES; i++) d->map[i] = i % RBT_N_STREAMS;
    hipDeviceProp_t p; HIPCHK(hipGetDeviceProperties(&p, device));
    snprintf(d->name, sizeof d->name, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    for (int k = 0; k < RBT_N_STREAMS; k++) HIPCHK(hipStreamCreateWithFlags(&d->streams[k], hipStreamNonBlocking));
    for (int i = 0; i < 64; i++) HIPCHK(hipEventCreateWithFlags(&d->dep_ev[i], hipEventDisableTiming));
    for (int k = 0; k < RBT_N_LANES; k++) for (int i = 0; i < 16; i++) { HIPCHK(hipEventCreate(&d->ev[k][i][0])); HIPCHK(hipEventCreate(&d->ev[k][i][1])); }
    g_devs[device] = d;
  }
  t_dev = g_devs[device]; t_cur = 0;
  return 0;
}
int dev_select(int device) {
  if (device < 0 || device >= RBT_MAX_DEVICES || !g_devs[device]) return -1;
  if (t_dev != g_devs[device]) { HIPCHK(hipSetDevice(device)); t_dev = g_devs[device]; t_cur = 0; }
  return 0;
}
const char* dev_name() { return t_dev ? t_dev->name : ""; }
void set_stream(int i) { t_cur = lane_of(i); }
void map_lane(int lane, int stream) { t_dev->map[lane_of(lane)] = ((stream % RBT_N_STREAMS) + RBT_N_STREAMS) % RBT_N_STREAMS; }
int stream_mark(int signaller) {
  int id = t_dev->dep_next; t_dev->dep_next = (t_dev->dep_next + 1) % 64;
  (void)hipEventRecord(t_dev->dep_ev[id], stream_of(signaller));
  return id;
}
void stream_wait_mark(int waiter, int mark) { (void)hipStreamWaitEvent(stream_of(waiter), t_dev->dep_ev[mark & 63], 0); }
void stream_wait(int waiter, int signaller) {
  hipEvent_t e = t_dev->dep_ev[t_dev->dep_next]; t_dev->dep_next = (t_dev->dep_next + 1) % 64;
  (void)hipEventRecord(e, stream_of(signaller));
  (void)hipStreamWaitEvent(stream_of(waiter), e, 0);
}
// Device allocations are recycled: hipMalloc / hipFree of GOF-sized arenas cost milliseconds each (hipFree also drains the
// device), and a transcoder calls with the same sizes over and over. Freed blocks go to a small best-fit pool; at most
// RBT_POOL_KEEP blocks are kept, the rest is returned to the driver.
// HBM a new arena leaves free. The runtime allocates too, and when IT finds nothing left the process is aborted (HSA_STATUS_ERROR_OUT_OF_RESOURCES, seen in round 3 with
// ~280 GB of cached arenas): scratch memory of a hardware queue the first time a kernel with a private segment runs on it. What this library's kernels need: the largest
// private segment is 68 bytes per lane (the slice parsers; intra analysis 40, entropy coders 52, encoder SAO 28, intra coder 8 - llvm-readelf --notes of the code objects),
// i.e. 64 x 68 B rounded up to 5 KB per wave, x 256 CUs x 32 wave slots = 42 MB per queue, x 16 queues = 0.66 GB; plus code objects, signals and the copy engines' staging
// (measured on MI355X with tools/scratch_probe.py: the figure is in DESIGN.md 5). Default 3 GB - four times the computed need; RBT_HBM_RESERVE_MB overrides (read once).
static size_t g_reserve = (size_t)-1;
size_t dev_reserve_bytes() {
  if (g_reserve == (size_t)-1) { const char* e = getenv("RBT_HBM_RESERVE_MB"); long long mb = e && *e ? atoll(e) : 3072; if (mb < 0) mb = 0; g_reserve = (size_t)mb << 20; }
  return g_reserve;
}
static thread_local size_t t_alloc_total = 0;
size_t dev_alloc_total() { return t_alloc_total; }
int dev_mem_info(size_t* free_b, size_t* total_b, size_t* cached_b, size_t* live_b) {
  Dev* D = t_dev; if (!D) return -1;
  size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return -1; }
  std::lock_guard<std::mutex> lk(D->pool_mu);
  size_t c = 0, l = 0; for (auto& f : D->pool_free) c += f.n; for (auto& f : D->pool_live) l += f.n;
  if (free_b) *free_b = fr; if (total_b) *total_b = tot; if (cached_b) *cached_b = c; if (live_b) *live_b = l;
  return 0;
}
void* dev_alloc(size_t n) {
  if (!n) n = 1;
  Dev* D = t_dev; if (!D) return nullptr;
  std::lock_guard<std::mutex> lk(D->pool_mu);
  int best = -1;
  for (size_t i = 0; i < D->pool_free.size(); i++)
    if (D->pool_free[i].n >= n && D->pool_free[i].n <= n + n / 4 + (1u << 20) && (best < 0 || D->pool_free[i].n < D->pool_free[(size_t)best].n)) best = (int)i;
  PoolBlock b;
  if (best >= 0) { b = D->pool_free[(size_t)best]; D->pool_free.erase(D->pool_free.begin() + best); }
  else {
    b.p = nullptr; b.n = n;
    // Arenas must not take the last of the HBM (dev_reserve_bytes above): a new block of a megabyte or more leaves the reserve free - cached blocks go back to the driver
    // first, and if that is not enough the call fails (RBT_ERR_NOMEM) and the context stays usable (tests/test_gpu_memory.py).
    if (n >= ((size_t)1 << 20)) {
      const size_t reserve = dev_reserve_bytes();
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < n + reserve) {
        for (auto& f : D->pool_free) (void)hipFree(f.p);
        D->pool_free.clear();
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < n + reserve) return nullptr;
      }
    }
    if (hipMalloc(&b.p, n) != hipSuccess) {
      (void)hipGetLastError();                                        // the failed allocation is handled here: it must not surface later as a "kernel" error
      for (auto& f : D->pool_free) (void)hipFree(f.p);                 // give everything back and try once more
      D->pool_free.clear();
      if (hipMalloc(&b.p, n) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
  }
  D->pool_live.push_back(b);
  t_alloc_total += b.n;
  return b.p;
}
void dev_free(void* p) {
  if (!p) return;
  // the block may belong to another device than the one this thread works on (a job destroyed from another context's call)
  for (int k = -1; k < RBT_MAX_DEVICES; k++) {
    Dev* D = k < 0 ? t_dev : g_devs[k]; if (!D || (k >= 0 && D == t_dev)) continue;
    std::lock_guard<std::mutex> lk(D->pool_mu);
    for (size_t i = 0; i < D->pool_live.size(); i++) if (D->pool_live[i].p == p) {
      D->pool_free.push_back(D->pool_live[i]); D->pool_live.erase(D->pool_live.begin() + (long)i);
      while (D->pool_free.size() > RBT_POOL_KEEP) {                      // drop the smallest block
        size_t s = 0; for (size_t j = 1; j < D->pool_free.size(); j++) if (D->pool_free[j].n < D->pool_free[s].n) s = j;
        (void)hipFree(D->pool_free[s].p); D->pool_free.erase(D->pool_free.begin() + (long)s);
      }
      return;
    }
  }
  (void)hipFree(p);
}
void dev_release_pool() {
  Dev* D = t_dev; if (!D) return;
  std::lock_guard<std::mutex> lk(D->pool_mu);
  for (auto& f : D->pool_free) (void)hipFree(f.p);
  D->pool_free.clear();
}
int h2d(void* d, const void* h, size_t n) { HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g_stream)); return 0; }
int d2h(void* h, const void* d, size_t n) { HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); return 0; }
int dev_memset(void* d, int v, size_t n) { HIPCHK(hipMemsetAsync(d, v, n, g_stream)); return 0; }
int dev_sync() { HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipGetLastError()); return 0; }
void timer_begin(int id) { (void)hipEventRecord(t_dev->ev[t_cur][id][0], g_stream); }
void timer_end(int id) { (void)hipEventRecord(t_dev->ev[t_cur][id][1], g_stream); }
double timer_ms(int id) { float ms = 0; if (hipEventElapsedTime(&ms, t_dev->ev[t_cur][id][0], t_dev->ev[t_cur][id][1]) != hipSuccess) return 0; return ms; }

// ---------------------------------------------------------------------------------------------- decode kernels
// (the slice parser's kernels live in rbt_kernels_parse.hip: that file is compiled for speed, this one for size)
// one workgroup (two waves: luma chain, Cb/Cr chain - rbt_recon.h RbtReconRole) per CTB on anti-diagonal d (x + 2y == d):
// left, above-left, above and above-right CTBs are complete
__device__ __forceinline__ void recon_ctb_roles(RbtFrame* frames, const RbtSlice* slices, int fi, int addr, RBT_LDS_AS RbtReconCtbLds* L) {
  if (threadIdx.x < 64) rbt_recon_ctb<RC_ROLE_LUMA>(frames, slices, fi, addr, &L->t, &L->role[0]);
  else rbt_recon_ctb<RC_ROLE_CHROMA>(frames, slices, fi, addr, &L->t, &L->role[1]);
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_recon_diag(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int d, int y_first) {
  __shared__ RbtReconCtbLds lds;
  int fi = frame_list[blockIdx.y];
  const RbtStreamCfg* g = &frames[fi].cfg;
  int y = y_first + blockIdx.x, x = d - 2 * y;
  if (y >= g->h_ctb || x < 0 || x >= g->w_ctb) return;
  int addr = y * g->w_ctb + x;
  if (frames[fi].ctb_slice[addr] == 0xFFFF) return;     // CTB not covered by any decoded slice
  recon_ctb_roles(frames, slices, fi, addr, RBT_LDS_CAST(RbtReconCtbLds, &lds));
}
// the same over pictures of several batches
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_recon_diag_refs(const RbtFrameRef* refs, int d, int y_first) {
  __shared__ RbtReconCtbLds lds;
  const RbtFrameRef r = refs[blockIdx.y];
  const RbtStreamCfg* g = &r.frames[r.frame].cfg;
  int y = y_first + blockIdx.x, x = d - 2 * y;
  if (y >= g->h_ctb || x < 0 || x >= g->w_ctb) return;
  int addr = y * g->w_ctb + x;
  if (r.frames[r.frame].ctb_slice[addr] == 0xFFFF) return;     // CTB not covered by any decoded slice
  recon_ctb_roles(r.frames, r.slices, r.frame, addr, RBT_LDS_CAST(RbtReconCtbLds, &lds));
}
// ---- one launch per dependency level: CTB-to-CTB hand-off through done flags in HBM ----
// Producer (each of the two waves for its half of the CTB): stores -> agent-scope release (writes the XCD's L2 back) -> s_waitcnt vmcnt(0) -> relaxed flag store.
// Consumer: relaxed agent-scope poll of the flag (served by L2, never a stale L1 line) -> agent-scope acquire (invalidates this CU's L1) -> its own loads.
// (MI355X_MICROARCH.md, inter-workgroup visibility: per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores.)
// Every wave reaches its flag store on every path - a corrupt picture, an uncovered CTB or a wait that ran out of patience sets the picture's error word and goes on -
// so the grid always drains.
__device__ __forceinline__ void recon_wait_flag(const uint32_t* flag, RbtFrame* f) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(32);                               // ~2k cycles between polls: a CTB takes 100+ us, and a polling wave should leave the issue slots to working ones
    if (++spins > (1 << 20)) { f->error = 90; break; }      // seconds: the predecessor runs on this GPU already (ticket order), something is broken
  }
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_recon_level(const RbtFrameRef* refs, int n_frames, uint32_t* ticket) {
  __shared__ RbtReconCtbLds lds;
  __shared__ uint32_t s_ticket;
  if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t t = s_ticket;
  const RbtFrameRef r = refs[t % (uint32_t)n_frames];
  RbtFrame* f = &r.frames[r.frame];
  const int w = f->cfg.w_ctb, n_ctb = w * f->cfg.h_ctb, i = (int)(t / (uint32_t)n_frames);
  if (i >= n_ctb) return;                                     // this picture has fewer CTBs than the largest one of the launch
  const int xy = (int)r.order[i], x = xy & 0xFFFF, y = xy >> 16, addr = y * w + x, role = (int)threadIdx.x >> 6;
  uint32_t* done = f->ctb_done;
  // left neighbour; above-right neighbour (it waited for the one above, which waited for the one above-left) or, in the last column, the one above
  if (x > 0) recon_wait_flag(&done[2 * (addr - 1) + role], f);
  if (y > 0) recon_wait_flag(&done[2 * (addr - w + (x + 1 < w ? 1 : 0)) + role], f);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (f->ctb_slice[addr] != 0xFFFF) recon_ctb_roles(r.frames, r.slices, r.frame, addr, RBT_LDS_CAST(RbtReconCtbLds, &lds));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(&done[2 * addr + role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- one launch per dependency level with a device-side ready queue (round 4) ----
// k_recon_level above gives every CTB a workgroup that waits for its neighbours' flags: whatever is resident beyond the CTBs that can run holds LDS and wave slots (with 16
// jobs in flight that cost more than the shorter level gained), and the static ticket order is only the order of readiness when every CTB takes the same time (the
// per-diagonal launches show sigma > mean). Here a FEW persistent workgroups - about as many as the level's wavefront is wide - each take the next entry of a queue that is
// filled in the order CTBs BECOME ready: a finished CTB adds itself to its successors' arrival counts (ctb_done[2 * addr]) and appends the ones whose count is complete.
// qmem: [0] head (next slot to hand out), [1] tail (next slot to fill, counted from n_frames), [16 + s] slot n_frames + s = (picture << 18 | CTB) + 1; the first n_frames
// slots are the pictures' CTB 0 and need no memory. A workgroup that has reserved slot s waits until it is filled - by a workgroup that runs and waits for nobody, so the
// grid drains however few workgroups are resident; all waits are bounded. Hand-off as above: producer stores -> agent-scope release -> s_waitcnt vmcnt(0) -> relaxed
// atomics; consumer relaxed agent-scope poll -> agent-scope acquire -> its own loads (the arrival counts are a chain of relaxed agent-scope atomics: every producer's
// stores have reached memory before its count does, and the consumer invalidates after it has seen the last of them).
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_recon_queue(const RbtFrameRef* refs, int n_frames, uint32_t total, uint32_t* qmem) {
  __shared__ RbtReconCtbLds lds;
  __shared__ uint32_t s_task;
  uint32_t* head = qmem; uint32_t* tail = qmem + 1; uint32_t* q = qmem + 16;
  for (;;) {
    if (threadIdx.x == 0) {
      uint32_t task = 0;
      const uint32_t slot = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < total) {
        if (slot < (uint32_t)n_frames) task = (slot << 18) + 1;
        else {
          const uint32_t* p = &q[slot - (uint32_t)n_frames]; int spins = 0;
          while ((task = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 20)) { refs[0].frames[refs[0].frame].error = 92; break; }     // seconds: whoever fills the slot is running; something is broken - leave, the level ends incomplete with the error set
          }
        }
      }
      s_task = task;
    }
    __syncthreads();
    const uint32_t task = s_task;
    if (!task) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int fr = (int)((task - 1) >> 18), addr = (int)((task - 1) & 0x3FFFF);
    const RbtFrameRef r = refs[fr];
    RbtFrame* f = &r.frames[r.frame];
    if (f->ctb_slice[addr] != 0xFFFF) recon_ctb_roles(r.frames, r.slices, r.frame, addr, RBT_LDS_CAST(RbtReconCtbLds, &lds));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // both halves of the CTB are in memory (and nobody reads s_task or the tile any more)
    if (threadIdx.x == 0) {
      const int w = f->cfg.w_ctb, h = f->cfg.h_ctb, x = addr % w, y = addr / w;
      int succ[3]; const int ns = rc_ctb_successors(w, h, x, y, succ);
      for (int k = 0; k < ns; k++) {
        const int a2 = succ[k], need = rc_ctb_need(a2 % w, a2 / w);
        if ((int)__hip_atomic_fetch_add(&f->ctb_done[2 * a2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == need) {
          const uint32_t s2 = __hip_atomic_fetch_add(tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&q[s2], ((uint32_t)fr << 18 | (uint32_t)a2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}
// XCD-aware tile order for the kernels whose workgroups are independent. Workgroups go to the 8 XCDs round-robin in launch order, so neighbouring tiles of a picture - which
// share cache lines (a 32-sample CTB row is half a 128-byte line) and halo rows - would sit behind eight different L2s and every shared line would be fetched from memory once
// per L2. Tile = the b-th workgroup's position in a CONTIGUOUS eighth of the picture's tiles instead: workgroup b (XCD b % 8) takes tile (b % 8) * (n / 8) + b / 8, the
// remainder of an n that is no multiple of 8 keeps its place. Only when the grid's x extent is a multiple of 8 (then the XCD of a workgroup is blockIdx.x % 8 in every row of the grid).
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int per = n >> 3;
  if ((gridDim.x & 7) || per == 0 || b >= per * 8) return b;
  return (b & 7) * per + (b >> 3);
}
__global__ void __launch_bounds__(256) k_deblock(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int dir) {
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  const int n_edge = rbt_deblock_edge_count(&f->cfg, dir), n_blk = (n_edge + 255) >> 8;      // the units that can carry an edge of this direction (every second column / row)
  if ((int)blockIdx.x >= n_blk) return;
  const int e = xcd_tile(blockIdx.x, n_blk) * 256 + threadIdx.x;
  if (e >= n_edge) return;
  rbt_deblock_unit(f, slices, rbt_deblock_edge_unit(&f->cfg, dir, e), dir);
}
// deblocking (both edge directions) + SAO of one 64x64 tile through LDS (rbt_filter.h rbt_loopfilter_tile): pictures with SAO, whose output is a plane of its own
__global__ void __launch_bounds__(256) k_loopfilter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  __shared__ RbtLoopLds lds;
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  const int tw = (f->cfg.w + RBT_LF_TILE - 1) / RBT_LF_TILE, th = (f->cfg.h + RBT_LF_TILE - 1) / RBT_LF_TILE;
  if ((int)blockIdx.x >= tw * th) return;
  rbt_loopfilter_tile(f, slices, xcd_tile(blockIdx.x, tw * th), RBT_LDS_CAST(RbtLoopLds, &lds));
}
__global__ void __launch_bounds__(256) k_sao_ctb(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  const int n_ctb = f->cfg.w_ctb * f->cfg.h_ctb;
  if ((int)blockIdx.x >= n_ctb) return;
  rbt_sao_ctb(f, slices, xcd_tile(blockIdx.x, n_ctb));
}
__global__ void __launch_bounds__(256) k_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  int c = blockIdx.z, pw = c ? f->cfg.cw : f->cfg.w, ph = c ? f->cfg.ch : f->cfg.h;
  const int n_blk = (pw * ph + 255) >> 8;
  if ((int)blockIdx.x >= n_blk) return;
  int i = xcd_tile(blockIdx.x, n_blk) * 256 + threadIdx.x;
  if (i >= pw * ph) return;
  rbt_sao_sample(f, slices, c, i % pw, i / pw);
}

// CTB rows [y_begin, y_end): the rows above y_begin are complete, the anti-diagonals that touch the range run in order
void launch_recon(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int y_begin, int y_end) {
  if (n_frames <= 0) return;
  if (y_end > max_h_ctb) y_end = max_h_ctb;
  if (y_begin >= y_end) return;
  for (int d = 2 * y_begin; d <= max_w_ctb - 1 + 2 * (y_end - 1); d++) {
    int y_hi = d / 2; if (y_hi > y_end - 1) y_hi = y_end - 1;              // x = d - 2y >= 0
    int y_lo = (d - (max_w_ctb - 1) + 1) / 2; if (y_lo < y_begin) y_lo = y_begin;   // x <= w - 1
    if (y_lo > y_hi) continue;
    hipLaunchKernelGGL(k_recon_diag, dim3(y_hi - y_lo + 1, n_frames), dim3(128), 0, g_stream, frames, slices, frame_list, d, y_lo);
  }
}
void launch_recon_refs(const RbtFrameRef* refs, int n_frames, int max_w_ctb, int max_h_ctb) {
  if (n_frames <= 0) return;
  for (int d = 0; d <= max_w_ctb - 1 + 2 * (max_h_ctb - 1); d++) {
    int y_hi = d / 2; if (y_hi > max_h_ctb - 1) y_hi = max_h_ctb - 1;
    int y_lo = (d - (max_w_ctb - 1) + 1) / 2; if (y_lo < 0) y_lo = 0;
    if (y_lo > y_hi) continue;
    hipLaunchKernelGGL(k_recon_diag_refs, dim3(y_hi - y_lo + 1, n_frames), dim3(128), 0, g_stream, refs, d, y_lo);
  }
}
void launch_recon_level(const RbtFrameRef* refs, int n_frames, int max_ctbs, uint32_t* ticket) {
  if (n_frames <= 0 || max_ctbs <= 0) return;
  hipLaunchKernelGGL(k_recon_level, dim3((unsigned)n_frames * (unsigned)max_ctbs), dim3(128), 0, g_stream, refs, n_frames, ticket);
}
// Workgroups of the queue kernel: about what the level's wavefronts are wide. A picture of w x h CTBs has w + 2 (h - 1) anti-diagonals x + 2y = d, so on average
// w h / (w + 2h - 2) of its CTBs can run side by side (1280x1280 with 64x64 CTBs: 400 / 58 = 7); RBT_RECON_QUEUE_WIDTH=<percent> scales it (experiments).
static int g_queue_width_pct = -1;
void launch_recon_queue(const RbtFrameRef* refs, int n_frames, uint32_t total_ctbs, uint32_t* qmem, int n_wgs) {
  if (n_frames <= 0 || total_ctbs == 0) return;
  if (g_queue_width_pct < 0) { const char* e = getenv("RBT_RECON_QUEUE_WIDTH"); g_queue_width_pct = e && atoi(e) > 0 ? atoi(e) : 100; }
  long long n = (long long)n_wgs * g_queue_width_pct / 100;
  if (n < n_frames) n = n_frames; if (n > (long long)total_ctbs) n = total_ctbs; if (n > 65535 * 16) n = 65535 * 16;
  hipLaunchKernelGGL(k_recon_queue, dim3((unsigned)n), dim3(128), 0, g_stream, refs, n_frames, total_ctbs, qmem);
}
void launch_deblock(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_units) {
  if (n_frames <= 0) return;
  for (int dir = 0; dir < 2; dir++)
    hipLaunchKernelGGL(k_deblock, dim3((max_units / 2 + 1024 + 255) / 256, n_frames), dim3(256), 0, g_stream, frames, slices, frame_list, dir);   // edge units: at most units / 2 + max(w4, h4) / 2, pictures are at most 2048 units wide / high
}
void launch_loopfilter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w, int max_h) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(k_loopfilter, dim3(((max_w + RBT_LF_TILE - 1) / RBT_LF_TILE) * ((max_h + RBT_LF_TILE - 1) / RBT_LF_TILE), n_frames), dim3(256), 0, g_stream, frames, slices, frame_list);
}
void launch_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_luma_samples) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(k_sao, dim3((max_luma_samples + 255) / 256, n_frames, 3), dim3(256), 0, g_stream, frames, slices, frame_list);
}
void launch_sao_ctb(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs) {
  if (n_frames <= 0 || max_ctbs <= 0) return;
  hipLaunchKernelGGL(k_sao_ctb, dim3((unsigned)max_ctbs, n_frames), dim3(256), 0, g_stream, frames, slices, frame_list);
}

// ---------------------------------------------------------------------------------------------- encode kernels
__global__ void __launch_bounds__(256) k_pool(const uint16_t* in, int w, int factor, uint16_t* out, int ow, int oh, uint16_t* out_cb, uint16_t* out_cr, int chroma_value) {   // w = row stride of `in`
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ow * oh) en_pool_sample(in, w, factor, out, ow, i % ow, i / ow);
  if (i < (ow / 2) * (oh / 2)) { out_cb[i] = (uint16_t)chroma_value; out_cr[i] = (uint16_t)chroma_value; }
}
// up to 32 pictures of the same size in one launch (blockIdx.y = picture; the input pointers travel in the kernel arguments, the outputs are evenly spaced)
struct PoolIn { const uint16_t* p[32]; };
__global__ void __launch_bounds__(256) k_pool_many(PoolIn in, int w, int factor, uint16_t* out, size_t out_step, int ow, int oh, int chroma_value) {
  const uint16_t* src = in.p[blockIdx.y];
  uint16_t* y = out + out_step * blockIdx.y; uint16_t* cb = y + (size_t)ow * oh; uint16_t* cr = cb + (size_t)(ow / 2) * (oh / 2);
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ow * oh) en_pool_sample(src, w, factor, y, ow, i % ow, i / ow);
  if (i < (ow / 2) * (oh / 2)) { cb[i] = (uint16_t)chroma_value; cr[i] = (uint16_t)chroma_value; }
}
__global__ void __launch_bounds__(64) k_enc_analyse(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  __shared__ RbtAnalyseLds lds;
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  if ((int)blockIdx.x >= f->cfg.w_ctb * f->cfg.h_ctb) return;
  en_analyse_ctb(f, slices, xcd_tile(blockIdx.x, f->cfg.w_ctb * f->cfg.h_ctb), RBT_LDS_CAST(RbtAnalyseLds, &lds));
}
// (TL2: log2 of the largest CTB of the launch; fixes the size of the reconstruction tile in LDS)
template <int TL2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_enc_intra_rows(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  __shared__ RbtEncTileLdsT<TL2> lds;
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  int row = blockIdx.x;
  if (row >= f->cfg.h_ctb) return;
  rc_stage_tables(&RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds)->rc);
  for (int x = 0; x < f->cfg.w_ctb; x++) en_intra_ctb<TL2>(f, slices, row * f->cfg.w_ctb + x, RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds), x > 0);
}
// Wavefront mode (one slice per picture, one dependent slice segment per CTB row): rows of a picture predict from each other, so CTB x of row r is coded
// once row r - 1 has finished CTB x + 1 (its above-right neighbour; the last CTB of the row for the last column). With that two-CTB lag at most w_ctb / 2
// rows of a picture are in progress at any time, so a picture gets K = ceil(w_ctb / 2) waves instead of one per row (which would sit on their LDS
// waiting for most of their lives). A wave takes its picture from a launch ticket (start order) and then rows from the picture's own counter, the next
// row not handed out yet, until none is left: rows are handed out in order to waves that are running, so the row a wave waits for is always being worked
// on by a running wave and the launch makes progress however few of its waves the GPU holds at a time.
#ifndef WAVE_PUBLISH_EVERY
#define WAVE_PUBLISH_EVERY 2
#endif
__device__ __forceinline__ int wave_next_row(uint32_t* counter) {
  int r = 0;
  if ((threadIdx.x & 63) == 0) r = (int)atomicAdd(counter, 1u);
  return __builtin_amdgcn_readfirstlane(r);
}
template <int TL2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_enc_intra_wave(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, uint32_t* ticket) {
  __shared__ RbtEncTileLdsT<TL2> lds;
  const uint32_t t = (uint32_t)wave_next_row(ticket);
  RbtFrame* f = &frames[frame_list[t % (uint32_t)n_frames]];
  const int w = f->cfg.w_ctb, h = f->cfg.h_ctb;
  rc_stage_tables(&RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds)->rc);
  uint32_t* done = f->row_done;
  for (int row = wave_next_row(&done[2 * h]); row < h; row = wave_next_row(&done[2 * h])) {
    uint32_t seen = 0;
    for (int x = 0; x < w; x++) {
      if (row > 0) seen = rbt_flag_wait_seen(&done[row - 1], (uint32_t)(x + 2 < w ? x + 2 : w), seen, &f->error);
      en_intra_ctb<TL2>(f, slices, row * w + x, RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds), x > 0);
      if (((x + 1) & (WAVE_PUBLISH_EVERY - 1)) == 0 || x + 1 == w) RBT_FLAG_PUBLISH(&done[row], x + 1);   // a release (L2 write-back) per few CTBs, not per CTB
    }
  }
}
template <int TL2>
__global__ void __launch_bounds__(64) k_enc_intra_diag(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int d) {
  __shared__ RbtEncTileLdsT<TL2> lds;
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  int y = blockIdx.x, x = d - 2 * y;
  if (y >= f->cfg.h_ctb || x < 0 || x >= f->cfg.w_ctb) return;
  rc_stage_tables(&RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds)->rc);
  en_intra_ctb<TL2>(f, slices, y * f->cfg.w_ctb + x, RBT_LDS_CAST(RbtEncTileLdsT<TL2>, &lds), 0);
}
__global__ void __launch_bounds__(64) k_enc_inter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  __shared__ RbtEncLds lds;
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  if ((int)blockIdx.x >= f->cfg.w_ctb * f->cfg.h_ctb) return;
  rc_stage_tables(&RBT_LDS_CAST(RbtEncLds, &lds)->rc);
  en_inter_ctb(frames, f, slices, xcd_tile(blockIdx.x, f->cfg.w_ctb * f->cfg.h_ctb), RBT_LDS_CAST(RbtEncLds, &lds));
}
template <int TL2, bool REGION>
__global__ void __launch_bounds__(64) k_enc_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list) {
  __shared__ RbtSaoLds lds;
  __shared__ RbtSaoRegionT<REGION ? TL2 : 2> reg;      // REGION: the CTB + halo, deblocked here (RBT_FUSED_ENC_LF=1); otherwise a stub
  RbtFrame* f = &frames[frame_list[blockIdx.y]];
  if ((int)blockIdx.x >= f->cfg.w_ctb * f->cfg.h_ctb) return;
  en_sao_ctb<REGION>(f, slices, xcd_tile(blockIdx.x, f->cfg.w_ctb * f->cfg.h_ctb), RBT_LDS_CAST(RbtSaoLds, &lds), RBT_LDS_CAST(uint16_t, reg.ry), RBT_LDS_CAST(uint16_t, reg.rc[0]), RBT_LDS_CAST(uint16_t, reg.rc[1]));
}
template <int TL2>
__global__ void __launch_bounds__(64) k_entropy(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* slice_list) {
  __shared__ alignas(16) uint32_t lds[(RBT_ENTROPY_LDS_BYTES(TL2) + 3) / 4];
  en_entropy_slice(frames, slices, slice_list[blockIdx.x], out, RBT_LDS_CAST(RbtEntropyLds, lds));
}
// Wavefront mode: the segment of CTB row r starts from the context variables row r - 1 had after its second CTB (en_entropy_slice waits for them); as in
// k_enc_intra_wave a picture gets K = ceil(w_ctb / 2) waves, each taking the next row not handed out yet (segment of row r = the picture's first_slice + r).
template <int TL2>
__global__ void __launch_bounds__(64) k_entropy_wave(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* frame_list, int n_frames, uint32_t* ticket) {
  __shared__ alignas(16) uint32_t lds[(RBT_ENTROPY_LDS_BYTES(TL2) + 3) / 4];
  const uint32_t t = (uint32_t)wave_next_row(ticket);
  const RbtFrame* f = &frames[frame_list[t % (uint32_t)n_frames]];
  const int h = f->cfg.h_ctb, first = f->first_slice;
  uint32_t* next = &f->row_done[2 * h + 1];
  for (int row = wave_next_row(next); row < h; row = wave_next_row(next)) en_entropy_slice(frames, slices, first + row, out, RBT_LDS_CAST(RbtEntropyLds, lds));
}

__global__ void __launch_bounds__(256) k_pack(const uint8_t* out, const RbtSlice* slices, const uint32_t* dst_off, uint8_t* packed) {
  const RbtSlice* sl = &slices[blockIdx.x];
  const uint8_t* src = out + sl->out_off; uint8_t* dst = packed + dst_off[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < sl->out_size; i += 256) dst[i] = src[i];
}
void launch_pack(const uint8_t* out, const RbtSlice* slices, const uint32_t* dst_off, uint8_t* packed, int n_slices) {
  if (n_slices <= 0) return;
  hipLaunchKernelGGL(k_pack, dim3(n_slices), dim3(256), 0, g_stream, out, slices, dst_off, packed);
}
// copies the w x h region at (x0,y0) of a plane with row stride `stride` into a dw x dh plane, repeating the last column / row
// (source pictures of sizes the coded picture size has to be padded up to, conformance window 7.4.3.2.1)
__global__ void __launch_bounds__(256) k_pad(const uint16_t* in, int stride, int x0, int y0, int w, int h, uint16_t* out, int dw, int dh) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= dw * dh) return;
  int x = i % dw, y = i / dw;
  out[i] = in[(size_t)(y0 + (y < h ? y : h - 1)) * stride + x0 + (x < w ? x : w - 1)];
}
void launch_pad(const uint16_t* in, int stride, int x0, int y0, int w, int h, uint16_t* out, int dw, int dh) {
  hipLaunchKernelGGL(k_pad, dim3((dw * dh + 255) / 256), dim3(256), 0, g_stream, in, stride, x0, y0, w, h, out, dw, dh);
}
void launch_pool(const uint16_t* in, int stride, int w, int h, int factor, uint16_t* out, uint16_t* out_cb, uint16_t* out_cr, int chroma_value) {
  int ow = w / factor, oh = h / factor;
  hipLaunchKernelGGL(k_pool, dim3((ow * oh + 255) / 256), dim3(256), 0, g_stream, in, stride, factor, out, ow, oh, out_cb, out_cr, chroma_value);
}
void launch_pool_many(const uint16_t* const* in, int n, int stride, int w, int h, int factor, uint16_t* out, size_t out_step, int chroma_value) {
  const int ow = w / factor, oh = h / factor;
  for (int k = 0; k < n; k += 32) {
    PoolIn a; const int m = n - k < 32 ? n - k : 32;
    for (int i = 0; i < 32; i++) a.p[i] = in[k + (i < m ? i : 0)];
    hipLaunchKernelGGL(k_pool_many, dim3((ow * oh + 255) / 256, m), dim3(256), 0, g_stream, a, stride, factor, out + out_step * k, out_step, ow, oh, chroma_value);
  }
}
__global__ void __launch_bounds__(256) k_occ_units(const uint16_t* occ, size_t in_step, int ow, int oh, int s, int w4, int h4, uint8_t* out) {
  const int u = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (u < w4 * h4) out[(size_t)k * w4 * h4 + u] = (uint8_t)en_occ_unit_value(occ + in_step * k, ow, oh, s, w4, h4, u % w4, u / w4);
}
void launch_occ_units(const uint16_t* occ, size_t in_step, int n, int ow, int oh, int W, int w4, int h4, uint8_t* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_occ_units, dim3((w4 * h4 + 255) / 256, n), dim3(256), 0, g_stream, occ, in_step, ow, oh, W / ow, w4, h4, out);
}
void launch_enc_analyse(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(k_enc_analyse, dim3(max_ctbs, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
}
// Waves per picture of the wavefront kernels. w_ctb / 2 is all a picture can use (two-CTB lag between rows) and gives the shortest launch; while the first
// rows ramp up and the last ramp down those waves wait, holding their LDS. With many jobs in flight the GPU is full anyway and idle residents only take
// room from the other jobs' kernels: an eighth of the width then (measured with 16 GOFs in flight, 1280 x 1280, 40 CTBs wide: 625 / 642 / 660 / 631 / 548
// frames/s with 20 / 10 / 5 / 4 / 2 waves per picture). set_jobs_in_flight; RBT_WAVE_DIV overrides the divisor.
void set_jobs_in_flight(int depth) { static int env = -1; if (env < 0) { const char* e = getenv("RBT_WAVE_DIV"); env = e ? atoi(e) : 0; } if (!t_dev) return; t_dev->depth = depth; t_dev->wave_div = env > 0 ? env : (depth > 4 ? 8 : 2); }
int jobs_in_flight() { return t_dev ? t_dev->depth : 1; }
static int wave_rows_in_flight(int max_w_ctb, int max_h_ctb) { const int g_wave_div = t_dev ? t_dev->wave_div : 2; int K = (max_w_ctb + g_wave_div - 1) / g_wave_div; if (K < 1) K = 1; return K > max_h_ctb ? max_h_ctb : K; }
void launch_enc_intra(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int row_mode, int max_log2_ctb, uint32_t* ticket) {
  if (n_frames <= 0) return;
  const bool small = max_log2_ctb <= 5;
  if (row_mode == 2) {
    const int K = wave_rows_in_flight(max_w_ctb, max_h_ctb);
    if (small) hipLaunchKernelGGL(k_enc_intra_wave<5>, dim3(K * n_frames), dim3(64), 0, g_stream, frames, slices, frame_list, n_frames, ticket);
    else hipLaunchKernelGGL(k_enc_intra_wave<6>, dim3(K * n_frames), dim3(64), 0, g_stream, frames, slices, frame_list, n_frames, ticket);
    return;
  }
  if (row_mode) {
    if (small) hipLaunchKernelGGL(k_enc_intra_rows<5>, dim3(max_h_ctb, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
    else hipLaunchKernelGGL(k_enc_intra_rows<6>, dim3(max_h_ctb, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
    return;
  }
  int n_diag = max_w_ctb + 2 * (max_h_ctb - 1);
  for (int d = 0; d < n_diag; d++) {
    int rows = d / 2 + 1; if (rows > max_h_ctb) rows = max_h_ctb;
    if (small) hipLaunchKernelGGL(k_enc_intra_diag<5>, dim3(rows, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list, d);
    else hipLaunchKernelGGL(k_enc_intra_diag<6>, dim3(rows, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list, d);
  }
}
void launch_enc_inter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(k_enc_inter, dim3(max_ctbs, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
}
void launch_enc_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_ctbs, int max_log2_ctb, int deblock_inside) {
  if (n_frames <= 0) return;
  if (!deblock_inside) hipLaunchKernelGGL((k_enc_sao<5, false>), dim3(max_ctbs, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
  else if (max_log2_ctb <= 5) hipLaunchKernelGGL((k_enc_sao<5, true>), dim3(max_ctbs, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
  else hipLaunchKernelGGL((k_enc_sao<6, true>), dim3(max_ctbs, n_frames), dim3(64), 0, g_stream, frames, slices, frame_list);
}
void launch_entropy(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* slice_list, int n_slices, int max_log2_ctb) {
  if (n_slices <= 0) return;
  if (max_log2_ctb <= 5) hipLaunchKernelGGL(k_entropy<5>, dim3(n_slices), dim3(64), 0, g_stream, frames, slices, out, slice_list);
  else hipLaunchKernelGGL(k_entropy<6>, dim3(n_slices), dim3(64), 0, g_stream, frames, slices, out, slice_list);
}
void launch_entropy_wave(RbtFrame* frames, RbtSlice* slices, uint8_t* out, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int max_log2_ctb, uint32_t* ticket) {
  if (n_frames <= 0) return;
  const int K = wave_rows_in_flight(max_w_ctb, max_h_ctb);
  if (max_log2_ctb <= 5) hipLaunchKernelGGL(k_entropy_wave<5>, dim3(K * n_frames), dim3(64), 0, g_stream, frames, slices, out, frame_list, n_frames, ticket);
  else hipLaunchKernelGGL(k_entropy_wave<6>, dim3(K * n_frames), dim3(64), 0, g_stream, frames, slices, out, frame_list, n_frames, ticket);
}
// ---------------------------------------------------------------------------------------------- self-test
// The 32-point transform stages on the matrix cores (rbt_mfma.h) against the vector-ALU form of the same stages, on the device, for every block
// of `in` (n blocks of 32 x 32 int16): mismatches are counted in *bad. mode 0: inverse (sh = 20 - bd), 1: forward.
__global__ void __launch_bounds__(64) k_selftest_t32(const int16_t* in, int n, int bd, uint32_t* bad) {
  __shared__ RbtReconLdsCore lds;
  RBT_LDS_AS RbtReconLdsCore* l = RBT_LDS_CAST(RbtReconLdsCore, &lds);
  rc_stage_tables(l);
  uint32_t miss = 0;
  for (int b = blockIdx.x; b < n; b += gridDim.x) for (int mode = 0; mode < 2; mode++) {
    int16_t want[16];
    for (int i = threadIdx.x; i < 1024; i += 64) l->res[i] = in[(size_t)b * 1024 + i];
    RBT_SYNC_LDS();
    if (mode == 0) rc_inv_transform_n<5>(0, 20 - bd, l); else en_fwd_transform_n<5>(0, bd, l);
    for (int k = 0; k < 16; k++) want[k] = l->res[threadIdx.x + 64 * k];
    RBT_SYNC_LDS();
    for (int i = threadIdx.x; i < 1024; i += 64) l->res[i] = in[(size_t)b * 1024 + i];
    RBT_SYNC_LDS();
    if (mode == 0) rc_inv_transform_32(20 - bd, l); else en_fwd_transform_32(bd, l);
    for (int k = 0; k < 16; k++) miss += want[k] != l->res[threadIdx.x + 64 * k];
    RBT_SYNC_LDS();
  }
  if (miss) atomicAdd(bad, miss);
}
int selftest_transform32(const int16_t* blocks, int n, int bd, uint32_t* n_bad) {
  int16_t* d_in = (int16_t*)dev_alloc((size_t)n * 2048); uint32_t* d_bad = (uint32_t*)dev_alloc(4);
  if (!d_in || !d_bad) { dev_free(d_in); dev_free(d_bad); return -1; }
  int rc = h2d(d_in, blocks, (size_t)n * 2048) | dev_memset(d_bad, 0, 4);
  if (!rc) { hipLaunchKernelGGL(k_selftest_t32, dim3(n < 256 ? n : 256), dim3(64), 0, g_stream, d_in, n, bd, d_bad); rc = d2h(n_bad, d_bad, 4) | dev_sync(); }
  dev_free(d_in); dev_free(d_bad);
  return rc;
}

// ---------------------------------------------------------------------------------------------- verification stage (rbt_pcc.h)
__global__ void __launch_bounds__(256) k_pcc_occmap(RbtPccParams P, const uint16_t* occ, uint8_t* om) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < P.w * P.h) { const int u = i % P.w, v = i / P.w; om[i] = occ[(size_t)(v / P.prec) * P.ow + u / P.prec] > P.threshold; }
}
// one workgroup per (patch, block): a block belongs to the LAST patch with an occupied pixel in it (the reference overwrites in patch order)
__global__ void __launch_bounds__(256) k_pcc_owner(RbtPccParams P, const rbt_patch* patches, const uint32_t* items, const uint16_t* occ, uint32_t* b2p) {
  const uint32_t it = items[blockIdx.x]; const int pi = (int)(it >> 16), blk = (int)(it & 0xFFFF);
  const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0;
  int any = 0;
  for (int q = threadIdx.x; q < P.res * P.res; q += 256) any |= pc_pixel_occupied_video(&P, p, occ, ub, vb, q);
  any = __syncthreads_or(any);
  if (any && threadIdx.x == 0) atomicMax(&b2p[pc_block2canvas(p, ub, vb, P.bw)], (uint32_t)pi + 1);
}
__device__ __forceinline__ int pcc_block_scan(int v, int* total) {     // exclusive prefix sum over the 256 threads of the workgroup
  __shared__ int sh[256];
  sh[threadIdx.x] = v; __syncthreads();
  for (int o = 1; o < 256; o <<= 1) { int t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
  *total = sh[255];
  const int r = sh[threadIdx.x] - v; __syncthreads();
  return r;
}
__global__ void __launch_bounds__(256) k_pcc_count(RbtPccParams P, const rbt_patch* patches, const uint32_t* items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint32_t* b2p, uint32_t* counts) {
  const uint32_t it = items[blockIdx.x]; const int pi = (int)(it >> 16), blk = (int)(it & 0xFFFF);
  const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0;
  int n = 0;
  if (b2p[pc_block2canvas(p, ub, vb, P.bw)] == (uint32_t)pi + 1)
    for (int q = threadIdx.x; q < P.res * P.res; q += 256) n += pc_pixel_points(&P, p, occ, d0, d1, nullptr, nullptr, ub, vb, q, nullptr, nullptr);
  int total; pcc_block_scan(n, &total);
  if (threadIdx.x == 0) counts[blockIdx.x] = (uint32_t)total;
}
__global__ void __launch_bounds__(256) k_pcc_emit(RbtPccParams P, const rbt_patch* patches, const uint32_t* items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint16_t* t0, const uint16_t* t1,
                                                  const uint32_t* b2p, const uint32_t* offsets, int16_t* xyz, uint16_t* yuv, const uint8_t* om, uint32_t* meta) {
  const uint32_t it = items[blockIdx.x]; const int pi = (int)(it >> 16), blk = (int)(it & 0xFFFF);
  const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0;
  if (b2p[pc_block2canvas(p, ub, vb, P.bw)] != (uint32_t)pi + 1) return;         // uniform over the workgroup
  uint32_t base = offsets[blockIdx.x];
  for (int q0 = 0; q0 < P.res * P.res; q0 += 256) {                                 // pixels in raster order inside the block (v1, u1)
    const int q = q0 + threadIdx.x;
    int16_t pts[6]; uint16_t col[6];
    const int n = q < P.res * P.res ? pc_pixel_points(&P, p, occ, d0, d1, t0, t1, ub, vb, q, pts, col) : 0;
    int total; const int off = pcc_block_scan(n, &total);
    for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) { xyz[3 * (size_t)(base + off + i) + c] = pts[3 * i + c]; yuv[3 * (size_t)(base + off + i) + c] = col[3 * i + c]; }
    if (meta && n) {                                                               // geometry smoothing: the point's patch and whether its pixel is a boundary pixel
      int x, y; pc_patch2canvas(p, P.res, ub * P.res + q % P.res, vb * P.res + q / P.res, &x, &y);
      const uint32_t m = (uint32_t)pi | ((uint32_t)pc_boundary_point(om, x, y, P.w, P.h) << 31);
      for (int i = 0; i < n; i++) meta[base + off + i] = m;
    }
    base += (uint32_t)total;
  }
}
// geometry smoothing (rbt_pcc.h): one lane per point; max of all coordinates first (the grid's extent)
__global__ void __launch_bounds__(256) k_sm_max(const int16_t* xyz, int n3, uint32_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int v = i < n3 ? xyz[i] : 0; if (v < 0) v = 0;
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(out, (uint32_t)v);
}
__global__ void __launch_bounds__(256) k_sm_mark(RbtSmooth G, const int16_t* xyz, const uint32_t* meta) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < G.n_points) pc_sm_mark(&G, xyz, meta, i); }
__global__ void __launch_bounds__(256) k_sm_accum(RbtSmooth G, const int16_t* xyz, const uint32_t* meta) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < G.n_points) pc_sm_accum(&G, xyz, meta, i); }
__global__ void __launch_bounds__(256) k_sm_filter(RbtSmooth G, int16_t* xyz, const uint32_t* meta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < G.n_points && pc_sm_filter(&G, xyz, meta, i)) atomicAdd(G.moved, 1u);
}
// exclusive prefix sum of n <= a few 10^4 counts in one workgroup: every thread sums a contiguous chunk, the chunk sums are scanned in LDS
__global__ void __launch_bounds__(1024) k_scan_u32(const uint32_t* in, uint32_t* out, int n) {
  __shared__ uint32_t sh[1024];
  const int chunk = (n + 1023) / 1024, b = threadIdx.x * chunk, e = min(n, b + chunk);
  uint32_t s = 0; for (int i = b; i < e; i++) s += in[i];
  sh[threadIdx.x] = s; __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) { uint32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
  uint32_t run = sh[threadIdx.x] - s;
  for (int i = b; i < e; i++) { out[i] = run; run += in[i]; }
  if (threadIdx.x == 1023) out[n] = sh[1023];
}
__global__ void __launch_bounds__(256) k_vol_set(const int16_t* xyz, int n, uint32_t* vol, uint8_t* first, uint32_t* n_unique) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  const uint32_t bit = 1u << (x & 31), old = atomicOr(&vol[pc_voxel_word(x, y, z)], bit);
  const int f = !(old & bit);                                                       // the first point at this position represents its duplicates
  first[i] = (uint8_t)f;
  if (f) atomicAdd(n_unique, 1u);
}
__global__ void __launch_bounds__(256) k_vol_nn(const int16_t* xyz, const uint8_t* first, int n, const uint32_t* vol_other, unsigned long long* sse, uint32_t* max_d2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t d = 0;
  if (i < n && first[i]) d = pc_nearest_d2(vol_other, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  unsigned long long s = d; uint32_t m = d;
  for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o, 64); const uint32_t t = __shfl_down(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && s) { atomicAdd(sse, s); atomicMax(max_d2, m); }
}
__global__ void __launch_bounds__(256) k_d2_insert(const int16_t* xyz, int n, uint32_t* vol, uint32_t* keys, uint32_t* vals, int lg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  atomicOr(&vol[pc_voxel_word(x, y, z)], 1u << (x & 31));
  pc_hash_insert(keys, vals, lg, pc_voxel_id(x, y, z), (uint32_t)i);
}
__global__ void __launch_bounds__(256) k_d2_give(RbtD2Set A, const int16_t* normals_a, RbtD2Set B, long long* acc_b, int32_t* cnt_b) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < A.n) pc_d2_give(&A, normals_a, &B, acc_b, cnt_b, i); }
__global__ void __launch_bounds__(256) k_d2_take(RbtD2Set B, RbtD2Set A, const int16_t* normals_a, long long* acc_b, int32_t* cnt_b) { const int j = blockIdx.x * 256 + threadIdx.x; if (j < B.n) pc_d2_take(&B, &A, normals_a, acc_b, cnt_b, j); }
// out: [0] sum of the values (double), [1] their maximum (double, compared as its bit pattern: the values are >= 0), [2] number of representatives (u64)
__global__ void __launch_bounds__(256) k_d2_dist(RbtD2Set P, RbtD2Set Q, const long long* acc_q, const int32_t* cnt_q, const int16_t* normals_q, double* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = i < P.n ? pc_d2_value(&P, &Q, acc_q, cnt_q, normals_q, i) : -1.0;
  unsigned long long cnt = v >= 0; if (v < 0) v = 0;
  double s = v, m = v;
  for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o, 64); const double t = __shfl_down(m, o, 64); m = t > m ? t : m; cnt += __shfl_down(cnt, o, 64); }
  if ((threadIdx.x & 63) == 0 && cnt) { atomicAdd(&out[0], s); atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(m)); atomicAdd((unsigned long long*)&out[2], cnt); }
}
void launch_d2_insert(const int16_t* xyz, int n, uint32_t* vol, uint32_t* keys, uint32_t* vals, int lg) { if (n > 0) hipLaunchKernelGGL(k_d2_insert, dim3((n + 255) / 256), dim3(256), 0, g_stream, xyz, n, vol, keys, vals, lg); }
void launch_d2_give(const RbtD2Set* A, const int16_t* normals_a, const RbtD2Set* B, long long* acc_b, int32_t* cnt_b) { if (A->n > 0) hipLaunchKernelGGL(k_d2_give, dim3((A->n + 255) / 256), dim3(256), 0, g_stream, *A, normals_a, *B, acc_b, cnt_b); }
void launch_d2_take(const RbtD2Set* B, const RbtD2Set* A, const int16_t* normals_a, long long* acc_b, int32_t* cnt_b) { if (B->n > 0) hipLaunchKernelGGL(k_d2_take, dim3((B->n + 255) / 256), dim3(256), 0, g_stream, *B, *A, normals_a, acc_b, cnt_b); }
void launch_d2_dist(const RbtD2Set* P, const RbtD2Set* Q, const long long* acc_q, const int32_t* cnt_q, const int16_t* normals_q, double* out) { if (P->n > 0) hipLaunchKernelGGL(k_d2_dist, dim3((P->n + 255) / 256), dim3(256), 0, g_stream, *P, *Q, acc_q, cnt_q, normals_q, out); }
void launch_pcc_occmap(const RbtPccParams* P, const uint16_t* occ, uint8_t* om) { hipLaunchKernelGGL(k_pcc_occmap, dim3((P->w * P->h + 255) / 256), dim3(256), 0, g_stream, *P, occ, om); }
void launch_pcc_owner(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, uint32_t* b2p) {
  if (n_items > 0) hipLaunchKernelGGL(k_pcc_owner, dim3(n_items), dim3(256), 0, g_stream, *P, patches, items, occ, b2p);
}
void launch_pcc_count(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint32_t* b2p, uint32_t* counts) {
  if (n_items > 0) hipLaunchKernelGGL(k_pcc_count, dim3(n_items), dim3(256), 0, g_stream, *P, patches, items, occ, d0, d1, b2p, counts);
}
void launch_scan_u32(const uint32_t* in, uint32_t* out, int n) { hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, g_stream, in, out, n); }
void launch_pcc_emit(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint16_t* t0, const uint16_t* t1,
                     const uint32_t* b2p, const uint32_t* offsets, int16_t* xyz, uint16_t* yuv, const uint8_t* om, uint32_t* meta) {
  if (n_items > 0) hipLaunchKernelGGL(k_pcc_emit, dim3(n_items), dim3(256), 0, g_stream, *P, patches, items, occ, d0, d1, t0, t1, b2p, offsets, xyz, yuv, om, meta);
}
void launch_sm_max(const int16_t* xyz, int n_points, uint32_t* out) { if (n_points > 0) hipLaunchKernelGGL(k_sm_max, dim3((3 * n_points + 255) / 256), dim3(256), 0, g_stream, xyz, 3 * n_points, out); }
void launch_sm_passes(const RbtSmooth* G, int16_t* xyz, const uint32_t* meta) {
  if (G->n_points <= 0) return;
  const dim3 grid((G->n_points + 255) / 256);
  hipLaunchKernelGGL(k_sm_mark, grid, dim3(256), 0, g_stream, *G, xyz, meta);
  hipLaunchKernelGGL(k_sm_accum, grid, dim3(256), 0, g_stream, *G, xyz, meta);
  hipLaunchKernelGGL(k_sm_filter, grid, dim3(256), 0, g_stream, *G, xyz, meta);
}
void launch_vol_set(const int16_t* xyz, int n, uint32_t* vol, uint8_t* first, uint32_t* n_unique) { if (n > 0) hipLaunchKernelGGL(k_vol_set, dim3((n + 255) / 256), dim3(256), 0, g_stream, xyz, n, vol, first, n_unique); }
void launch_vol_nn(const int16_t* xyz, const uint8_t* first, int n, const uint32_t* vol_other, unsigned long long* sse, uint32_t* max_d2) {
  if (n > 0) hipLaunchKernelGGL(k_vol_nn, dim3((n + 255) / 256), dim3(256), 0, g_stream, xyz, first, n, vol_other, sse, max_d2);
}
}  // namespace rbtk
