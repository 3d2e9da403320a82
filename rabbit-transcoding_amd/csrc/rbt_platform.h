// rabbit-transcoding_amd — MI355X-native V-PCC video transcoding hot path.
// Kernel-side platform glue. The product is built with hipcc for gfx950 only (RBT_HOSTEMU undefined).
//
// RBT_HOSTEMU is a TEST-ONLY build mode (tests/hostemu/, never shipped, never a fallback of the product library):
// it compiles the same kernel bodies as serial host code so their logic can be debugged against the oracle in a
// container that has no GPU. Kernel bodies are written in "phase" style: wave-/block-uniform scalar code plus
// RBT_PAR_FOR loops separated by RBT_SYNC(); on the GPU a PAR_FOR is a thread-strided loop, in host emulation a
// plain loop executed once per block.
#pragma once
#include <stdint.h>
#include <stddef.h>

// MEASUREMENT builds only (tools/ablate.sh, never the product): RBT_ABLATE is a mask of phases LEFT OUT of the CTB chains, so that a counter pass shows what each
// one costs in instructions; what such a build computes is wrong by design. 0 in the product: every `if (RBT_ABLATE & bit)` folds away.
// Reconstruction: 1 residual (scaling + inverse transform), 2 reference-sample gather, 4 reference smoothing, 8 mode set-up (DC sum / angular reference array),
// 16 prediction + store, 32 the CTB's fetch into LDS, 64 prediction units (motion compensation), 128 the residual pass of inter blocks. Analysis: 0x100 source fetch, 0x200 gather + smoothing per block, 0x400 the input's modes (hints),
// 0x800 candidate set-up, 0x1000 candidate SAD, 0x2000 SATD of the chosen mode. Intra coding (decisions only: the output stays a valid stream): 0x4000 the closed-loop mode
// choice, 0x8000 its coded trial, 0x10000 the four-way form of a CU's luma.
#ifndef RBT_ABLATE
#define RBT_ABLATE 0
#endif

#ifdef RBT_HOSTEMU
#include <string.h>
#define RBT_DEV static inline
#define RBT_CONST static const
#define RBT_PAR_FOR(i, n) for (int i = 0; i < (int)(n); i++)
#define RBT_BLK_FOR(i, n) for (int i = 0; i < (int)(n); i++)
#define RBT_SYNC() do { } while (0)
#define RBT_SYNC_LDS() do { } while (0)
#define RBT_LANE0 1
#define RBT_NTHREADS 1
#define RBT_LDS_AS
#define RBT_CONST_AS
#define RBT_LDS_CAST(T, p) (p)
#define RBT_UNI(x) (x)
// Per-lane ("vector") values inside wave-uniform code: on the GPU one register whose lane p holds element p; in host
// emulation an array. RBT_VFOR(p, n) runs its body for p = 0..n-1 (n <= 64): lane p on the GPU, a loop on the host.
#define RBT_VEC(T, name) T name[64]
#define RBT_V(name, p) name[p]
#define RBT_VFOR(p, n) for (int p = 0; p < (int)(n); p++)
#define RBT_VBALLOT(out, p, n, expr) do { uint64_t b_ = 0; for (int p = 0; p < (int)(n); p++) if (expr) b_ |= 1ull << p; (out) = b_; } while (0)
#define RBT_VGET(name, lane) name[lane]
#define RBT_VSET(name, lane, v) name[lane] = (v)
#else
#include <hip/hip_runtime.h>
#define RBT_VEC(T, name) T name
#define RBT_V(name, p) name
#define RBT_VFOR(p, n) for (int p = (int)threadIdx.x & 63, once_ = 1; once_ && p < (int)(n); once_ = 0)
#define RBT_VBALLOT(out, p, n, expr) do { int p = (int)threadIdx.x & 63; (out) = __ballot(p < (int)(n) && (expr)); } while (0)
#define RBT_VGET(name, lane) __builtin_amdgcn_readlane(name, lane)
#define RBT_VSET(name, lane, v) name = rbt_writelane(name, v, lane)
// v_writelane_b32 through the LLVM intrinsic (this clang has no builtin for it)
__device__ int rbt_llvm_writelane(int v, int lane, int old) __asm("llvm.amdgcn.writelane");
static __device__ __forceinline__ int rbt_writelane(int old, int v, int lane) { return rbt_llvm_writelane(v, lane, old); }
#define RBT_DEV static __device__ __forceinline__
#define RBT_CONST static __device__ const
// (wave-local: kernels that use it run single-wave workgroups, or one wave per role - k_recon_diag)
#define RBT_PAR_FOR(i, n) for (int i = (int)threadIdx.x & 63; i < (int)(n); i += 64)
// the same over all threads of a workgroup of any size (separated by RBT_SYNC())
#define RBT_BLK_FOR(i, n) for (int i = (int)threadIdx.x; i < (int)(n); i += (int)blockDim.x)
#define RBT_SYNC() __syncthreads()
// single-wave workgroups only: orders LDS traffic between the lanes of the wave without waiting for outstanding global
// stores (a __syncthreads() would wait for every store round trip to HBM)
// (Round 4 tried a compiler-only barrier here - the LDS executes one wave's instructions in issue order, and all 221 GPU parity tests passed that way - and measured no
// gain, 988-1005 against 983-999 fps: the compiler waits before the first use of the data anyway. The explicit wait stays.)
#define RBT_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define RBT_LANE0 ((threadIdx.x & 63) == 0)
#define RBT_NTHREADS ((int)blockDim.x)
// LDS objects are always reached through address_space(3) pointers (ds_* instructions). A generic (flat) pointer into
// LDS is unsafe on gfx950: the compiler may fold part of an index into the instruction's immediate offset, and a base
// register that underflows the LDS aperture (object at LDS offset 0, negative partial index) is treated as a global
// address and faults with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
#define RBT_LDS_AS __attribute__((address_space(3)))
// Read-only kernel inputs that are addressed wave-uniformly (the slice data) are read through the constant address space:
// the loads become scalar (s_load_dword, counted by lgkmcnt) instead of flat vector loads. Vector loads share vmcnt with
// the stores on gfx9-family parts, so waiting for one loaded word also waits for every store still in flight - each wait
// then costs a full store round trip to HBM.
#define RBT_CONST_AS __attribute__((address_space(4)))
#define RBT_LDS_CAST(T, p) ((RBT_LDS_AS T*)(uintptr_t)(p))
// Marks a value as wave-uniform so the compiler keeps it in SGPRs / issues it on the scalar unit. Only for values that
// ARE uniform by construction (the entropy kernels run every lane of the wave on identical data).
#ifdef RBT_NO_UNI
#define RBT_UNI(x) (x)
#else
#define RBT_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif
#endif

// integer add on an LDS word from any lane of the workgroup (ds_add_u32); a plain add in the serial host emulation
#ifdef RBT_HOSTEMU
#define RBT_LDS_ADD(p, v) (*(p) += (v))
#else
#define RBT_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif

// Progress counters between workgroups of one launch (wavefront dependencies between CTB rows). The producer makes its global stores visible to the
// whole device (release fence at agent scope: per-XCD L2 write-back) and waits for them before it publishes the counter; the consumer polls the counter
// with a relaxed agent-scope load, sleeping in between so that a waiting wave leaves the issue slots to working ones, and then invalidates its caches
// (acquire). The poll is bounded: a wave that ran out of patience reports through *err and goes on, so the grid always drains. In the serial host
// emulation the producer has always run already.
#ifdef RBT_HOSTEMU
#define RBT_FLAG_PUBLISH(p, v) (*(p) = (uint32_t)(v))
RBT_DEV void rbt_flag_wait(const uint32_t* p, uint32_t need, int32_t* err) { if (*p < need) *err = 91; }
RBT_DEV uint32_t rbt_flag_wait_seen(const uint32_t* p, uint32_t need, uint32_t seen, int32_t* err) { if (*p < need) *err = 91; (void)seen; return *p; }
#else
#define RBT_FLAG_PUBLISH(p, v) do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  if (RBT_LANE0) __hip_atomic_store((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
RBT_DEV void rbt_flag_wait(const uint32_t* p, uint32_t need, int32_t* err) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(32);
    if (++spins > (1 << 20)) { *err = 91; break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// the same for a consumer that follows one counter: `seen` is the value its last wait returned (everything published up to it is visible already), so
// nothing is polled or invalidated until more is needed; returns the value to remember
RBT_DEV uint32_t rbt_flag_wait_seen(const uint32_t* p, uint32_t need, uint32_t seen, int32_t* err) {
  if (need <= seen) return seen;
  int spins = 0; uint32_t v;
  while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
    __builtin_amdgcn_s_sleep(32);
    if (++spins > (1 << 20)) { *err = 91; v = need; break; }
    if ((spins & 63) == 0 && __hip_atomic_load((const uint32_t*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { v = need; break; }   // the picture is bad already: do not wait for what may never come
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
#endif

// the picture's error word as other waves have left it (a wave that waited in vain, or for a row whose wave gave up, must not go on with what it did not get)
RBT_DEV int32_t rbt_err_peek(const int32_t* err) {
#ifdef RBT_HOSTEMU
  return *err;
#else
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#endif
}
// Bit-field read of a packed wave-uniform word. On the GPU it is one scalar instruction the compiler may neither hoist nor
// keep alive: rarely used parameters then cost one SGPR per word instead of one (spilled) SGPR per field.
template <int SH, int N> RBT_DEV uint32_t rbt_bfe(uint32_t w) {
#ifdef RBT_HOSTEMU
  return (w >> SH) & ((1u << N) - 1u);
#else
  uint32_t v; asm volatile("s_bfe_u32 %0, %1, %2" : "=s"(v) : "s"(w), "n"(SH | (N << 16)) : "scc"); return v;
#endif
}
template <int SH, int N> RBT_DEV int rbt_bfe_i(uint32_t w) {
#ifdef RBT_HOSTEMU
  return (int)(w << (32 - SH - N)) >> (32 - N);
#else
  int v; asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(v) : "s"(w), "n"(SH | (N << 16)) : "scc"); return v;
#endif
}
template <class T> RBT_DEV T* rbt_uni_ptr(T* p) {
#ifdef RBT_HOSTEMU
  return p;
#else
  uintptr_t a = (uintptr_t)p;
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return (T*)(((uintptr_t)hi << 32) | lo);
#endif
}
RBT_DEV int rbt_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
RBT_DEV int rbt_min(int a, int b) { return a < b ? a : b; }
RBT_DEV int rbt_max(int a, int b) { return a > b ? a : b; }
RBT_DEV int rbt_abs(int a) { return a < 0 ? -a : a; }
