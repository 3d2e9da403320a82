// Probe of the A / B operand lane maps of v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data (the guide documents the bf16 maps only):
// which k does byte j of lane half h carry? Result on MI355X: BOTH candidate maps give the exact product - the instruction sums over the 32 (lane half, byte)
// slots and any assignment of k to slots is right as long as A and B use the same one, which csrc/rbt_mfma.h does (RBT_MFMA_K for both operands). hipcc --offload-arch=gfx950 tools/mfma_i8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// hyp: 0: k = 16h + j ; 1: k = 8h + (j&7) + 16*(j>>3)
__device__ int kmap(int hyp, int h, int j) { return hyp == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3); }
__global__ void k(const signed char* A, const signed char* B, int* D, int hyp) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v4i v; signed char b[16]; } a, b;
  for (int j = 0; j < 16; j++) { int kk = kmap(hyp, h, j); a.b[j] = A[r * 32 + kk]; b.b[j] = B[kk * 32 + r]; }
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
  for (int q = 0; q < 16; q++) { int row = (q & 3) + 8 * (q >> 2) + 4 * h; D[row * 32 + r] = c[q]; }
}
int main() {
  signed char A[1024], B[1024]; int D[1024], R[1024];
  srand(1); for (int i = 0; i < 1024; i++) { A[i] = rand() % 255 - 127; B[i] = rand() % 255 - 127; }
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int q = 0; q < 32; q++) s += A[i * 32 + q] * B[q * 32 + j]; R[i * 32 + j] = s; }
  signed char *dA, *dB; int* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; hyp++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp); hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
    printf("hyp %d: %s\n", hyp, memcmp(D, R, 4096) ? "MISMATCH" : "exact");
  }
  return 0;
}
