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

// ---- D2 (point-to-plane): QualityMetrics::compute with computeC2p_ (PCCMetrics.cpp:100-124, :213-215), normals by copyNormals / scaleNormals (:371-376, PCCPointSet.cpp:2322-2380) ----
// Duplicates are merged as for D1; a merged point is stood for by its lowest original index (hash map voxel -> index), its normal is that point's. "The points at the same
// distance" are all points at exactly the nearest squared distance: the shell of voxels at that distance is walked in the other cloud's bit volume. Normals are Q14 integers;
// the reconstruction's normals are kept as integer sum and count (scaleNormals' mean is formed where it is used), so the only floating point is the final per-point value.
RBT_DEV uint32_t pc_voxel_id(int x, int y, int z) { return ((uint32_t)z << (2 * RBT_PCC_BITS)) | ((uint32_t)y << RBT_PCC_BITS) | (uint32_t)x; }
RBT_DEV uint32_t pc_hash_slot(uint32_t id, int lg) { return (id * 2654435761u) >> (32 - lg); }
// voxel -> lowest point index; keys: voxel id + 1 (0 = empty), vals start at 0xFFFFFFFF
RBT_DEV void pc_hash_insert(uint32_t* keys, uint32_t* vals, int lg, uint32_t id, uint32_t index) {
  const uint32_t mask = (1u << lg) - 1;
  for (uint32_t s = pc_hash_slot(id, lg);; s = (s + 1) & mask) {
#ifdef RBT_HOSTEMU
    if (keys[s] == 0) keys[s] = id + 1;
    if (keys[s] == id + 1) { if (index < vals[s]) vals[s] = index; return; }
#else
    const uint32_t old = atomicCAS(&keys[s], 0u, id + 1);
    if (old == 0 || old == id + 1) { atomicMin(&vals[s], index); return; }
#endif
  }
}
RBT_DEV uint32_t pc_hash_find(const uint32_t* keys, const uint32_t* vals, int lg, uint32_t id) {
  const uint32_t mask = (1u << lg) - 1;
  for (uint32_t s = pc_hash_slot(id, lg);; s = (s + 1) & mask) { if (keys[s] == id + 1) return vals[s]; if (keys[s] == 0) return 0xFFFFFFFFu; }
}
RBT_DEV int pc_isqrt(uint32_t v) { int r = (int)__builtin_sqrtf((float)v); while ((uint32_t)(r * r) > v) r--; while ((uint32_t)((r + 1) * (r + 1)) <= v) r++; return r; }
// every set voxel of vol at squared distance exactly d2 from (x,y,z): f(voxel id)
template <class F> RBT_DEV void pc_for_ties(const uint32_t* vol, int x, int y, int z, uint32_t d2, F f) {
  const int r = pc_isqrt(d2);
  for (int dz = -r; dz <= r; dz++) {
    const int zz = z + dz; if (zz < 0 || zz >= RBT_PCC_DIM) continue;
    for (int dy = -r; dy <= r; dy++) {
      const int yy = y + dy; if (yy < 0 || yy >= RBT_PCC_DIM) continue;
      const int rem = (int)d2 - dz * dz - dy * dy; if (rem < 0) continue;
      const int dx = pc_isqrt((uint32_t)rem); if (dx * dx != rem) continue;
      if (x + dx < RBT_PCC_DIM && pc_voxel_set(vol, x + dx, yy, zz)) f(pc_voxel_id(x + dx, yy, zz));
      if (dx && x - dx >= 0 && pc_voxel_set(vol, x - dx, yy, zz)) f(pc_voxel_id(x - dx, yy, zz));
    }
  }
}
RBT_DEV int pc_d2_is_rep(const RbtD2Set* s, int i) { return pc_hash_find(s->keys, s->vals, s->lg, pc_voxel_id(s->xyz[3 * i], s->xyz[3 * i + 1], s->xyz[3 * i + 2])) == (uint32_t)i; }
// scaleNormals, first half: source point i (a representative) gives its normal to the points of B nearest to it
RBT_DEV void pc_d2_give(const RbtD2Set* A, const int16_t* normals_a, const RbtD2Set* B, long long* acc_b, int32_t* cnt_b, int i) {
  if (!pc_d2_is_rep(A, i)) return;
  const int x = A->xyz[3 * i], y = A->xyz[3 * i + 1], z = A->xyz[3 * i + 2];
  const uint32_t d2 = pc_nearest_d2(B->vol, x, y, z);
  pc_for_ties(B->vol, x, y, z, d2, [&](uint32_t id) {
    const uint32_t j = pc_hash_find(B->keys, B->vals, B->lg, id);
#ifdef RBT_HOSTEMU
    for (int c = 0; c < 3; c++) acc_b[3 * j + c] += normals_a[3 * i + c];
    cnt_b[j]++;
#else
    for (int c = 0; c < 3; c++) atomicAdd((unsigned long long*)&acc_b[3 * j + c], (unsigned long long)(long long)normals_a[3 * i + c]);
    atomicAdd(&cnt_b[j], 1);
#endif
  });
}
// second half: a point of B that got no normal takes the mean of the source points nearest to it (sum and count)
RBT_DEV void pc_d2_take(const RbtD2Set* B, const RbtD2Set* A, const int16_t* normals_a, long long* acc_b, int32_t* cnt_b, int j) {
  if (!pc_d2_is_rep(B, j) || cnt_b[j] != 0) return;
  const int x = B->xyz[3 * j], y = B->xyz[3 * j + 1], z = B->xyz[3 * j + 2];
  const uint32_t d2 = pc_nearest_d2(A->vol, x, y, z);
  long long s0 = 0, s1 = 0, s2 = 0; int n = 0;
  pc_for_ties(A->vol, x, y, z, d2, [&](uint32_t id) { const uint32_t i = pc_hash_find(A->keys, A->vals, A->lg, id); s0 += normals_a[3 * i]; s1 += normals_a[3 * i + 1]; s2 += normals_a[3 * i + 2]; n++; });
  acc_b[3 * j] = s0; acc_b[3 * j + 1] = s1; acc_b[3 * j + 2] = s2; cnt_b[j] = n;
}
// point-to-plane value of point i of P against Q: mean over Q's points at the nearest distance of ((p - q) . normal(q))^2; Q's normals are acc_q / cnt_q (Q14), or
// normals_q with count 1 where acc_q is null. Returns -1 for a point that is not a representative.
RBT_DEV double pc_d2_value(const RbtD2Set* P, const RbtD2Set* Q, const long long* acc_q, const int32_t* cnt_q, const int16_t* normals_q, int i) {
  if (!pc_d2_is_rep(P, i)) return -1.0;
  const int x = P->xyz[3 * i], y = P->xyz[3 * i + 1], z = P->xyz[3 * i + 2];
  const uint32_t d2 = pc_nearest_d2(Q->vol, x, y, z);
  double sum = 0; int n = 0;
  pc_for_ties(Q->vol, x, y, z, d2, [&](uint32_t id) {
    const uint32_t j = pc_hash_find(Q->keys, Q->vals, Q->lg, id);
    const int ex = x - (int)(id & (RBT_PCC_DIM - 1)), ey = y - (int)((id >> RBT_PCC_BITS) & (RBT_PCC_DIM - 1)), ez = z - (int)(id >> (2 * RBT_PCC_BITS));
    long long dot; int cnt;
    if (acc_q) { dot = ex * acc_q[3 * j] + ey * acc_q[3 * j + 1] + ez * acc_q[3 * j + 2]; cnt = cnt_q[j]; }
    else { dot = (long long)ex * normals_q[3 * j] + (long long)ey * normals_q[3 * j + 1] + (long long)ez * normals_q[3 * j + 2]; cnt = 1; }
    const double v = (double)dot / (double)cnt;
    sum += v * v; n++;
  });
  return sum / n / (16384.0 * 16384.0);
}

// ---- geometry smoothing (rbt_atlas_params.geometry_smoothing): PCCCodec::identifyBoundaryPoints (PCCCodec.cpp:266-325) and smoothPointCloudPostprocess with gridSmoothing
// (:52-145, addGridCentroid :980-998, gridFiltering :1000-1063, smoothPointCloudGrid :1065-1104) ----
// The reference walks the point list three times and numbers the cells it meets; nothing in its result depends on that numbering or on the order of the points (centroid
// sums are exact in float below 2^24, "a second patch in the cell" is min != max of the patch indices), so here every pass is one lane per point over dense w^3 cell arrays:
//   mark   boundary points flag the 8 cells around them                       (cellIndex != -1)
//   accum  every point adds itself to its own cell if that is flagged          (atomicAdd of coordinates and count, atomicMin / atomicMax of patch index + 1)
//   filter boundary points next to a cell with two patches move to the tri-linear blend of the 8 cell centroids when they are far enough from it - the reference's
//          float / double arithmetic operation for operation (no contraction into fused multiply-adds), so the moved points are bit-identical
// identifyBoundaryPoints for an occupied pixel (x, y) of the up-scaled occupancy map: 1 when the point gets boundary point type 1
RBT_DEV int pc_boundary_point(const uint8_t* om, int x, int y, int W, int H) {
#define RBT_OM(xx, yy) om[(size_t)(yy) * W + (xx)]
  int t = 0;
  if (y > 0 && y < H - 1 && (RBT_OM(x, y - 1) == 0 || RBT_OM(x, y + 1) == 0)) t = 1;
  if (!t && x > 0 && x < W - 1 && (RBT_OM(x + 1, y) == 0 || RBT_OM(x - 1, y) == 0)) t = 1;
  if (!t && y > 0 && y < H - 1 && x > 0 && (RBT_OM(x - 1, y - 1) == 0 || RBT_OM(x - 1, y + 1) == 0)) t = 1;
  if (!t && y > 0 && y < H - 1 && x < W - 1 && (RBT_OM(x + 1, y - 1) == 0 || RBT_OM(x + 1, y + 1) == 0)) t = 1;
  if (y == 0 || y == H - 1 || x == 0 || x == W - 1) t = 1;
  if (!t) {                                                                   // second layer: the ring two away
    for (int ix = -2; ix <= 2 && !t; ix++) for (int iy = -2; iy <= 2 && !t; iy++)
      if ((ix > 1 || ix < -1 || iy > 1 || iy < -1) && y + iy >= 0 && y + iy < H && x + ix >= 0 && x + ix < W && RBT_OM(x + ix, y + iy) == 0) t = 1;
    if (y == 1 || y == H - 2 || x == 1 || x == W - 2) t = 1;
  }
#undef RBT_OM
  return t;
}
struct RbtSmooth {               // grid of the smoothing pass and its dense cell arrays (w^3 cells)
  int32_t g, w, disth, th, threshold, n_points;
  uint8_t* flag; uint32_t* sum; uint32_t* cnt; uint32_t* pmin; uint32_t* pmax;     // sum: 3 per cell; pmin starts at 0xFFFFFFFF
  uint32_t* moved;
};
RBT_DEV int pc_sm_skip(const RbtSmooth* G, int x, int y, int z) { return x < G->disth || y < G->disth || z < G->disth || G->th <= x + G->disth || G->th <= y + G->disth || G->th <= z + G->disth; }
// per point: meta = patch index | boundary flag << 31
RBT_DEV void pc_sm_mark(const RbtSmooth* G, const int16_t* xyz, const uint32_t* meta, int i) {
  if (!(meta[i] >> 31)) return;
  const int P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  if (pc_sm_skip(G, P[0], P[1], P[2])) return;
  int Q[3]; for (int k = 0; k < 3; k++) Q[k] = P[k] / G->g + ((P[k] % G->g < G->g / 2) ? -1 : 0);
  for (int iz = 0; iz < 2; iz++) for (int iy = 0; iy < 2; iy++) for (int ix = 0; ix < 2; ix++) G->flag[(size_t)(Q[0] + ix) + (size_t)(Q[1] + iy) * G->w + (size_t)(Q[2] + iz) * G->w * G->w] = 1;
}
RBT_DEV void pc_sm_accum(const RbtSmooth* G, const int16_t* xyz, const uint32_t* meta, int i) {
  const int P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  if (pc_sm_skip(G, P[0], P[1], P[2])) return;
  const size_t c = (size_t)(P[0] / G->g) + (size_t)(P[1] / G->g) * G->w + (size_t)(P[2] / G->g) * G->w * G->w;
  if (!G->flag[c]) return;
  const uint32_t pidx = (meta[i] & 0x7FFFFFFFu) + 1;
#ifdef RBT_HOSTEMU
  for (int k = 0; k < 3; k++) G->sum[3 * c + k] += (uint32_t)P[k];
  G->cnt[c]++; if (pidx < G->pmin[c]) G->pmin[c] = pidx; if (pidx > G->pmax[c]) G->pmax[c] = pidx;
#else
  for (int k = 0; k < 3; k++) atomicAdd(&G->sum[3 * c + k], (uint32_t)P[k]);
  atomicAdd(&G->cnt[c], 1u); atomicMin(&G->pmin[c], pidx); atomicMax(&G->pmax[c], pidx);
#endif
}
// gridFiltering + the decision of smoothPointCloudGrid for boundary point i: writes the new position and returns 1 when the point moves
RBT_DEV int pc_sm_filter(const RbtSmooth* G, int16_t* xyz, const uint32_t* meta, int i) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (!(meta[i] >> 31)) return 0;
  const int P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  if (pc_sm_skip(G, P[0], P[1], P[2])) return 0;
  const int g = G->g, w = G->w, half = g / 2, g2 = g * 2;
  int S[3], Wt[3]; size_t idx[8]; int other = 0;
  for (int k = 0; k < 3; k++) { const int P2 = P[k] / g, P3 = P[k] - P2 * g; S[k] = P2 + (P3 < half ? -1 : 0); Wt[k] = (P[k] - S[k] * g - half) * 2 + 1; }
  for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
    const size_t c = (size_t)(S[0] + dx) + (size_t)(S[1] + dy) * w + (size_t)(S[2] + dz) * w * w; idx[dz * 4 + dy * 2 + dx] = c;
    if ((G->cnt[c] & 0xFFFFu) != 0 && G->pmin[c] != G->pmax[c]) other = 1;                     // doSmooth && count != 0 (the reference counts in 16 bits)
  }
  if (!other) return 0;
  const double cur[3] = {(double)P[0], (double)P[1], (double)P[2]};
  const int Q[3] = {g2 - Wt[0], g2 - Wt[1], g2 - Wt[2]};
  int cnt = 0; double c4[3] = {0.0, 0.0, 0.0};
  for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
    const size_t c = idx[dz * 4 + dy * 2 + dx]; const int n = (int)(G->cnt[c] & 0xFFFFu);
    const int abc = (dx ? Wt[0] : Q[0]) * (dy ? Wt[1] : Q[1]) * (dz ? Wt[2] : Q[2]);
    for (int k = 0; k < 3; k++) {
      // the cell centre as the reference holds it: a float sum divided by the count in float (division in double and rounding once is the correctly rounded float quotient)
      const double centre = n > 0 ? (double)(float)((double)(float)G->sum[3 * c + k] / (double)(float)n) : cur[k];
      c4[k] = c4[k] + centre * (double)abc;
    }
    cnt += abc * n;
  }
  const double vol = (double)(g2 * g2 * g2);
  cnt /= g2 * g2 * g2;
  if (cnt == 0) return 0;                                                                    // (the reference divides by zero here and its comparison is false)
  double centroid[3], d[3];
  for (int k = 0; k < 3; k++) { centroid[k] = (c4[k] / vol) * (double)cnt; d[k] = cur[k] * (double)cnt - centroid[k]; }
  const double dist2 = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / (double)cnt + 0.5;
  if (!(dist2 >= (double)((G->threshold > cnt ? G->threshold : cnt) * 2))) return 0;
  for (int k = 0; k < 3; k++) xyz[3 * i + k] = (int16_t)(double)(long long)(centroid[k] / (double)cnt + 0.5);
  return 1;
}

