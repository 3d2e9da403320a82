// 32-point transform stages on the matrix cores (gfx950 v_mfma_i32_32x32x32_i8), exact integer arithmetic.
// The four 32x32 stages of the codec - inverse rows / columns (8.6.4.2) and the encoder's forward rows / columns - are all one 32x32x32 product
// of the int8 DCT matrix (or its transpose) with a 32x32 block of int16 data followed by a rounding shift. The int16 operand is split into three
// signed-byte planes, x = p2 * 16384 + p1 * 128 + p0 with p0, p1 in 0..127 and p2 in -2..1, so three MFMAs accumulate the exact product in int32
// (|sum| <= 32 * 90 * 32768 < 2^27): acc = ((M . p2) * 128 + M . p1) * 128 + M . p0. One wave, operands and result in LDS:
// 16 LDS reads per operand and lane, 3 matrix instructions and 16 stores per stage instead of 16 outputs x 32 multiply-adds x 2 LDS reads per lane.
// Smaller transforms stay on the vector ALU (a 16x16 block fills a quarter of the 32x32 tile; SURVEY.md: MFMA only for the dense 32x32 tiles).
// The host emulation (tests/hostemu) has no matrix cores and keeps the scalar form; both are exact, so they agree bit for bit.
#pragma once
#include "rbt_platform.h"
#ifndef RBT_HOSTEMU
typedef int rbt_v4i __attribute__((ext_vector_type(4)));
typedef int rbt_v16i __attribute__((ext_vector_type(16)));
// byte j (0..15) of lane half h carries k = RBT_MFMA_K(h, j) of the A row / B column the lane holds. The instruction sums over the 32 (half, byte)
// slots, so any assignment of k to slots is exact as long as both operands use the same one (tools/mfma_i8_probe.hip; checked on the device by
// rbt_selftest_transform32); this one makes the int16 operand's row reads contiguous.
#ifndef RBT_MFMA_K
#define RBT_MFMA_K(h, j) (16 * (h) + (j))
#endif
RBT_DEV void mf_planes(const int* x, rbt_v4i* p2, rbt_v4i* p1, rbt_v4i* p0) {     // 16 sign-extended int16 values -> three byte planes
#pragma unroll
  for (int d = 0; d < 4; d++) {
    unsigned a = 0, b = 0, c = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { const int v = x[4 * d + q]; a |= ((unsigned)(v >> 14) & 255u) << (8 * q); b |= ((unsigned)(v >> 7) & 127u) << (8 * q); c |= ((unsigned)v & 127u) << (8 * q); }
    (*p2)[d] = (int)a; (*p1)[d] = (int)b; (*p0)[d] = (int)c;
  }
}
// out[i][j] = (sum_k L[i][k] * R[k][j] + (1 << (shift - 1))) >> shift, clipped to int16 when clip16; 32 x 32, row pitch 32, one wave, all lanes.
// M8_LEFT: L[i][k] = m8[i * s_free + k * s_sum] (int8), R = x; otherwise L = x and R[k][j] = m8[k * s_sum + j * s_free].
template <bool M8_LEFT> RBT_DEV void mf_mm32(const RBT_LDS_AS int8_t* m8, int s_free, int s_sum, const RBT_LDS_AS int16_t* x, RBT_LDS_AS int16_t* out, int shift, int clip16) {
  const int l = (int)threadIdx.x & 63, r = l & 31, h = l >> 5;
  rbt_v4i m; int xv[16];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    unsigned w = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) w |= ((unsigned)(uint8_t)m8[r * s_free + RBT_MFMA_K(h, 4 * d + q) * s_sum]) << (8 * q);
    m[d] = (int)w;
  }
#pragma unroll
  for (int j = 0; j < 16; j++) xv[j] = M8_LEFT ? x[RBT_MFMA_K(h, j) * 32 + r] : x[r * 32 + RBT_MFMA_K(h, j)];
  rbt_v4i p2, p1, p0; mf_planes(xv, &p2, &p1, &p0);
  rbt_v16i acc = {0};
  acc = M8_LEFT ? __builtin_amdgcn_mfma_i32_32x32x32_i8(m, p2, acc, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(p2, m, acc, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] <<= 7;
  acc = M8_LEFT ? __builtin_amdgcn_mfma_i32_32x32x32_i8(m, p1, acc, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(p1, m, acc, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] <<= 7;
  acc = M8_LEFT ? __builtin_amdgcn_mfma_i32_32x32x32_i8(m, p0, acc, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(p0, m, acc, 0, 0, 0);
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
#pragma unroll
  for (int q = 0; q < 16; q++) {                       // C/D map: column = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
    int v = (acc[q] + add) >> shift;
    if (clip16) v = rbt_clip3(-32768, 32767, v);
    out[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = (int16_t)v;
  }
  RBT_SYNC_LDS();
}
#endif
