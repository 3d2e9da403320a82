// Decoder-side verification stage on the GPU: occupancy-masked reprojection of the decoded geometry maps
// to 3-D points with their colours, and the point-to-point (D1) metric. Replaces, for the configuration the CTC streams use (two geometry maps
// with absolute D1, no EOM / raw patches / point-local reconstruction, one tile per atlas frame):
//   PCCCodec::generateOccupancyMap                        source/lib/PccLibCommon/source/PCCCodec.cpp:1584-1606
//   PCCCodec::generateBlockToPatchFromOccupancyMapVideo   :1725-1763
//   PCCCodec::generatePointCloud + generatePoints         :517-978 (:556-570, :628-668, :781-838), :327-515 (:497-513)
//   PCCCodec::colorPointCloud                             :1308-1449 (:1417-1421)
//   PCCPatch::patch2Canvas / patchBlock2CanvasBlock / generatePoint   PCCPatch.cpp:192-305, PCCPatch.h:177-207
//   PCCImage::set                                         PCCImage.h:107-124 (video sample -> 8-bit map value)
//   QualityMetrics::compute (point-to-point)              source/lib/PccLibMetrics/source/PCCMetrics.cpp:75-231, :44-48, :299-309
// Data layout: the candidate points are enumerated exactly as the reference's loops visit them - patch, block (v0, u0), pixel (v1, u1),
// map - so one workgroup handles one (patch, block) pair with one lane per pixel, and a prefix sum over the per-block point counts
// gives every point the index it has in the reference's output: the point list is identical, order included.
// D1 works on a 1024^3-bit occupancy volume per cloud (128 MB of the 288 GB): setting a voxel's bit tells whether the point is a duplicate
// (duplicates are merged before the comparison, PCCMetricsParameters.cpp:50), and the nearest neighbour of a query is found by probing
// the other cloud's volume in shells of growing Chebyshev radius - a few dozen bit tests for surfaces a voxel or two apart.
#pragma once
#include "rbt_platform.h"
#include "../../include/rbt.h"

struct RbtPccParams {            // rbt_atlas_params + derived sizes
  int32_t w, h, res, prec, map_count, absolute_d1, remove_dup, threshold, geo_bd, attr_bd, bw, bh, ow, n_patches, has_attr, pad;
};
enum { RBT_OR_DEFAULT = 0, RBT_OR_SWAP, RBT_OR_ROT180, RBT_OR_MIRROR, RBT_OR_MROT180, RBT_OR_ROT270, RBT_OR_MROT90, RBT_OR_ROT90 };   // PCCCommon.h:128-137

RBT_DEV void pc_patch2canvas(const rbt_patch* p, int res, int u, int v, int* x, int* y) {      // PCCPatch.cpp:192-251
  const int su = p->size_u0 * res, sv = p->size_v0 * res, ox = p->u0 * res, oy = p->v0 * res;
  switch (p->orientation) {
    case RBT_OR_ROT90: *x = (sv - 1 - v) + ox; *y = u + oy; break;
    case RBT_OR_ROT180: *x = (su - 1 - u) + ox; *y = (sv - 1 - v) + oy; break;
    case RBT_OR_ROT270: *x = v + ox; *y = (su - 1 - u) + oy; break;
    case RBT_OR_MIRROR: *x = (su - 1 - u) + ox; *y = v + oy; break;
    case RBT_OR_MROT90: *x = (sv - 1 - v) + ox; *y = (su - 1 - u) + oy; break;
    case RBT_OR_MROT180: *x = u + ox; *y = (sv - 1 - v) + oy; break;
    case RBT_OR_SWAP: *x = v + ox; *y = u + oy; break;
    default: *x = u + ox; *y = v + oy; break;
  }
}
RBT_DEV int pc_block2canvas(const rbt_patch* p, int ub, int vb, int bw) {                    // PCCPatch.cpp:253-305 (bounds are checked on the host)
  int x, y;
  switch (p->orientation) {
    case RBT_OR_ROT90: x = (p->size_v0 - 1 - vb) + p->u0; y = ub + p->v0; break;
    case RBT_OR_ROT180: x = (p->size_u0 - 1 - ub) + p->u0; y = (p->size_v0 - 1 - vb) + p->v0; break;
    case RBT_OR_ROT270: x = vb + p->u0; y = (p->size_u0 - 1 - ub) + p->v0; break;
    case RBT_OR_MIRROR: x = (p->size_u0 - 1 - ub) + p->u0; y = vb + p->v0; break;
    case RBT_OR_MROT90: x = (p->size_v0 - 1 - vb) + p->u0; y = (p->size_u0 - 1 - ub) + p->v0; break;
    case RBT_OR_MROT180: x = ub + p->u0; y = (p->size_v0 - 1 - vb) + p->v0; break;
    case RBT_OR_SWAP: x = vb + p->u0; y = ub + p->v0; break;
    default: x = ub + p->u0; y = vb + p->v0; break;
  }
  return x + bw * y;
}
RBT_DEV int pc_to8(int v, int bd) { const int sh = bd - 8; if (sh <= 0) return v; const int r = (v + (1 << (sh - 1))) >> sh; return r > 255 ? 255 : r; }   // PCCImage.h:107-124
RBT_DEV void pc_gen_point(const rbt_patch* p, int u, int v, int depth, int16_t* out) {       // PCCPatch.h:177-207
  int n;
  if (p->projection_mode == 0) n = depth + p->d1; else { n = p->d1 - depth; if (n < 0) n = 0; }
  out[p->normal_axis] = (int16_t)n; out[p->tangent_axis] = (int16_t)(u * p->lod_x + p->u1); out[p->bitangent_axis] = (int16_t)(v * p->lod_y + p->v1);
}
// work item = (patch, block) pair `item` (items[] = patch index << 16 | block index inside the patch), lane = pixel inside the block
RBT_DEV int pc_pixel_occupied_video(const RbtPccParams* P, const rbt_patch* p, const uint16_t* occ, int ub, int vb, int q) {   // :1748-1755
  int x, y; pc_patch2canvas(p, P->res, ub * P->res + q % P->res, vb * P->res + q / P->res, &x, &y);
  return occ[(size_t)(y / P->prec) * P->ow + x / P->prec] > P->threshold;   // the frame generateOccupancyMap binarised in place (PCCCodec.cpp:1599-1600), not the raw sample
}
// the points pixel q of block (ub, vb) of patch p contributes (0..2); pts / col may be null (count only)
RBT_DEV int pc_pixel_points(const RbtPccParams* P, const rbt_patch* p, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint16_t* t0, const uint16_t* t1,
                            int ub, int vb, int q, int16_t* pts, uint16_t* col) {
  const int u = ub * P->res + q % P->res, v = vb * P->res + q / P->res;
  int x, y; pc_patch2canvas(p, P->res, u, v, &x, &y);
  if (!(occ[(size_t)(y / P->prec) * P->ow + x / P->prec] > P->threshold)) return 0;            // generateOccupancyMap :1584-1606 folded in
  int16_t a[3], b[3];
  pc_gen_point(p, u, v, pc_to8(d0[(size_t)y * P->w + x], P->geo_bd), a);
  int n = 1;
  if (P->map_count > 1) {
    if (P->absolute_d1) pc_gen_point(p, u, v, pc_to8(d1[(size_t)y * P->w + x], P->geo_bd), b);
    else { b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; const int dv = pc_to8(d1[(size_t)y * P->w + x], P->geo_bd); b[p->normal_axis] = (int16_t)(b[p->normal_axis] + (p->projection_mode == 0 ? dv : -dv)); }
    if (!(P->remove_dup && a[0] == b[0] && a[1] == b[1] && a[2] == b[2])) n = 2;
  }
  if (pts) {
    const int cw = P->w / 2; const size_t ys = (size_t)P->w * P->h, cs = (size_t)cw * (P->h / 2), co = (size_t)(y / 2) * cw + x / 2;
    for (int i = 0; i < n; i++) {
      const int16_t* s = i ? b : a; pts[3 * i] = s[0]; pts[3 * i + 1] = s[1]; pts[3 * i + 2] = s[2];
      const uint16_t* t = i ? t1 : t0;
      if (P->has_attr) { col[3 * i] = t[(size_t)y * P->w + x]; col[3 * i + 1] = t[ys + co]; col[3 * i + 2] = t[ys + cs + co]; }
      else col[3 * i] = col[3 * i + 1] = col[3 * i + 2] = (uint16_t)(1 << (P->attr_bd - 1));
    }
  }
  return n;
}

// ---- D1 on bit volumes ----
#define RBT_PCC_BITS 10                                   // coordinates 0..1023 (peak 1023, PCCMetricsParameters.cpp:49)
#define RBT_PCC_DIM (1 << RBT_PCC_BITS)
RBT_DEV size_t pc_voxel_word(int x, int y, int z) { return ((((size_t)z << RBT_PCC_BITS) + y) << (RBT_PCC_BITS - 5)) + (x >> 5); }
RBT_DEV int pc_voxel_set(const uint32_t* vol, int x, int y, int z) { return (vol[pc_voxel_word(x, y, z)] >> (x & 31)) & 1; }
// squared distance from (x,y,z) to the nearest set voxel of vol (vol is not empty)
RBT_DEV uint32_t pc_nearest_d2(const uint32_t* vol, int x, int y, int z) {
  if (pc_voxel_set(vol, x, y, z)) return 0;
  uint32_t best = 0xFFFFFFFFu;
  for (int r = 1; r < RBT_PCC_DIM; r++) {
    for (int dz = -r; dz <= r; dz++) {
      const int zz = z + dz; if (zz < 0 || zz >= RBT_PCC_DIM) continue;
      for (int dy = -r; dy <= r; dy++) {
        const int yy = y + dy; if (yy < 0 || yy >= RBT_PCC_DIM) continue;
        const int face = (dz == -r || dz == r || dy == -r || dy == r);         // on a face of the shell: the whole row, else its two ends
        const uint32_t base = (uint32_t)(dz * dz + dy * dy);
        if (base >= best) continue;
        for (int dx = -r; dx <= r; dx += face ? 1 : 2 * r) {
          const int xx = x + dx; if (xx < 0 || xx >= RBT_PCC_DIM) continue;
          if (pc_voxel_set(vol, xx, yy, zz)) { const uint32_t d = base + (uint32_t)(dx * dx); if (d < best) best = d; }
        }
      }
    }
    // everything outside shell r is at least r + 1 away
    if (best <= (uint32_t)((r + 1) * (r + 1))) break;
  }
  return best;
}
