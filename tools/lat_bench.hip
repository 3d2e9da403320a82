// Micro-benchmark: issue cost of instruction patterns for a LONE wave on gfx950 (what the serial CABAC chain pays).
// Build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/lat_bench tools/lat_bench.hip ; prints cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 20000
__global__ void k(uint64_t* out, int seed) {
  int lane = threadIdx.x & 63;
  int tab = lane * 7 + seed;
  uint64_t t0, t1; int r = 0;
  // 0: dependent SALU chain, 8 adds per iteration
  { int a = __builtin_amdgcn_readfirstlane(seed);
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1" : "+s"(a) : : "scc"); }
    t1 = __builtin_readcyclecounter(); r += a; if (lane == 0) out[0] = t1 - t0; }
  // 1: independent SALU, 8 adds to 4 registers
  { int a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
    a = __builtin_amdgcn_readfirstlane(a); b = __builtin_amdgcn_readfirstlane(b); c = __builtin_amdgcn_readfirstlane(c); d = __builtin_amdgcn_readfirstlane(d);
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc"); }
    t1 = __builtin_readcyclecounter(); r += a + b + c + d; if (lane == 0) out[1] = t1 - t0; }
  // 2: dependent VALU chain, 8 adds
  { int a = lane;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(a)); }
    t1 = __builtin_readcyclecounter(); r += a; if (lane == 0) out[2] = t1 - t0; }
  // 3: readlane -> salu -> readlane chain (4 pairs): s = readlane(tab, s & 63); s = s + 1
  { int s = __builtin_amdgcn_readfirstlane(seed) & 63;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("s_and_b32 %0, %0, 63\n s_nop 0\n v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n s_nop 0\n v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n s_nop 0\n v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n s_nop 0\n v_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(tab) : "scc"); }
    t1 = __builtin_readcyclecounter(); r += s; if (lane == 0) out[3] = t1 - t0; }
  // 4: taken branch: loop with only the loop-back branch + 1 add
  { int a = __builtin_amdgcn_readfirstlane(seed);
    t0 = __builtin_readcyclecounter();
    asm volatile("s_mov_b32 s40, %1\n 1: s_add_i32 %0, %0, 1\n s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b" : "+s"(a) : "n"(N_IT) : "s40", "scc");
    t1 = __builtin_readcyclecounter(); r += a; if (lane == 0) out[4] = t1 - t0; }
  // 5: same loop body with 2 extra forward taken branches
  { int a = __builtin_amdgcn_readfirstlane(seed);
    t0 = __builtin_readcyclecounter();
    asm volatile("s_mov_b32 s40, %1\n 1: s_add_i32 %0, %0, 1\n s_branch 2f\n s_nop 0\n 2: s_add_i32 %0, %0, 1\n s_branch 3f\n s_nop 0\n 3: s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b" : "+s"(a) : "n"(N_IT) : "s40", "scc");
    t1 = __builtin_readcyclecounter(); r += a; if (lane == 0) out[5] = t1 - t0; }
  // 6: writelane + readlane round trip (4 pairs)
  { int s = __builtin_amdgcn_readfirstlane(seed) & 63; int v = tab;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("s_mov_b32 m0, 5\n v_writelane_b32 %1, %0, m0\n v_readlane_b32 %0, %1, 5\n s_add_i32 %0, %0, 1\n v_writelane_b32 %1, %0, m0\n v_readlane_b32 %0, %1, 5\n s_add_i32 %0, %0, 1\n v_writelane_b32 %1, %0, m0\n v_readlane_b32 %0, %1, 5\n s_add_i32 %0, %0, 1\n v_writelane_b32 %1, %0, m0\n v_readlane_b32 %0, %1, 5\n s_add_i32 %0, %0, 1" : "+s"(s), "+v"(v) : : "m0", "scc"); }
    t1 = __builtin_readcyclecounter(); r += s + v; if (lane == 0) out[6] = t1 - t0; }
  // 7: not-taken branch cost: 8 x (cmp + cbranch not taken)
  { int a = __builtin_amdgcn_readfirstlane(seed) | 1;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; i++) { asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 9f\n 9:" : "+s"(a) : : "scc"); }
    t1 = __builtin_readcyclecounter(); r += a; if (lane == 0) out[7] = t1 - t0; }
  if (lane == 0) out[15] = r;
}
int main() {
  uint64_t* d; hipMalloc(&d, 16 * 8); hipMemset(d, 0, 128);
  k<<<1, 64>>>(d, 3); hipDeviceSynchronize(); k<<<1, 64>>>(d, 3); hipDeviceSynchronize();
  uint64_t h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
  const char* names[8] = {"8 dependent s_add", "8 independent s_add", "8 dependent v_add", "4 x (s_and, s_nop, v_readlane) dependent", "loop: add, sub, cmp, taken cbranch", "same + 2 taken s_branch (+2 add)", "4 x (v_writelane, v_readlane, s_add)", "4 x (s_cmp, not-taken cbranch)"};
  for (int i = 0; i < 8; i++) printf("%-45s %8.2f cycles/iter\n", names[i], (double)h[i] / N_IT);
  return 0;
}
