/* ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
 *
 * CPU restatement of the decoder-side verification stage, for the configuration the CTC streams use
 * (two geometry maps in one interleaved video with absolute D1, no EOM, no raw patches, no point-local reconstruction, no patch border
 * filtering, one tile per atlas frame, removeDuplicatePoints on):
 *   PCCImage::set                                        source/lib/PccLibCommon/include/PCCImage.h:90-131    (10-bit video sample -> 8-bit map value)
 *   PCCCodec::generateOccupancyMap                       source/lib/PccLibCommon/source/PCCCodec.cpp:1584-1606
 *   PCCCodec::generateBlockToPatchFromOccupancyMapVideo  :1725-1763
 *   PCCPatch::patch2Canvas / patchBlock2CanvasBlock      source/lib/PccLibCommon/source/PCCPatch.cpp:192-251 / :253-305
 *   PCCPatch::generatePoint / generateNormalCoordinate   include/PCCPatch.h:177-207
 *   PCCCodec::generatePointCloud                         :517-978 (occupancy upscale :556-570, patch / block / pixel loops :628-668, the
 *                                                        non-EOM branch :781-838) with generatePoints :327-515 (the last branch :497-513)
 *   PCCCodec::colorPointCloud                            :1308-1449 (the `f < mapCount` fetch :1417-1421)
 *   QualityMetrics::compute                              source/lib/PccLibMetrics/source/PCCMetrics.cpp:75-231 (point-to-point part), getPSNR :44-48,
 *                                                        symmetric result :299-309, duplicate points merged first (PCCMetricsParameters.cpp:50)
 *   PCCCodec::identifyBoundaryPoints                     :266-325, called per point at :962-972
 *   PCCCodec::smoothPointCloudPostprocess (gridSmoothing) :52-145 with addGridCentroid :980-998, gridFiltering :1000-1063, smoothPointCloudGrid :1065-1104
 *                                                        (oracle_atlas.geometry_smoothing; call site PCCDecoder.cpp:434-437, parameters cfg/common/ctc-common.cfg:57-60)
 * PARITY UNPINNED: PccLibCommon / PccLibMetrics need a cmake-generated PCCConfig.h (plus TBB and nanoflann) and the reference holds no
 * fixtures for this stage; the restatement follows the source text. The 4:2:0 -> 4:4:4 conversion of the attribute video in front of
 * colorPointCloud is PccLibColorConverter (out of scope): chroma is fetched at the co-sited half-resolution sample here.
 */
#include "pcc_recon.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* PCCCommon.h:128-137 */
enum { OR_DEFAULT = 0, OR_SWAP, OR_ROT180, OR_MIRROR, OR_MROT180, OR_ROT270, OR_MROT90, OR_ROT90 };

/* PCCPatch::patch2Canvas (PCCPatch.cpp:192-251); returns 0 when (x,y) leaves the canvas (the reference exits) */
static int patch2canvas(const oracle_patch* p, int res, int u, int v, int cw, int ch, int* x, int* y) {
  const int su = p->size_u0 * res, sv = p->size_v0 * res, ox = p->u0 * res, oy = p->v0 * res;
  switch (p->orientation) {
    case OR_DEFAULT: *x = u + ox; *y = v + oy; break;
    case OR_ROT90: *x = (sv - 1 - v) + ox; *y = u + oy; break;
    case OR_ROT180: *x = (su - 1 - u) + ox; *y = (sv - 1 - v) + oy; break;
    case OR_ROT270: *x = v + ox; *y = (su - 1 - u) + oy; break;
    case OR_MIRROR: *x = (su - 1 - u) + ox; *y = v + oy; break;
    case OR_MROT90: *x = (sv - 1 - v) + ox; *y = (su - 1 - u) + oy; break;
    case OR_MROT180: *x = u + ox; *y = (sv - 1 - v) + oy; break;
    case OR_SWAP: *x = v + ox; *y = u + oy; break;    /* == MROT270 */
    default: return 0;
  }
  return *x >= 0 && *y >= 0 && *x < cw && *y < ch;
}
/* PCCPatch::patchBlock2CanvasBlock (PCCPatch.cpp:253-305): -1 outside */
static int block2canvas(const oracle_patch* p, int ub, int vb, int bw, int bh) {
  int x, y;
  switch (p->orientation) {
    case OR_DEFAULT: x = ub + p->u0; y = vb + p->v0; break;
    case OR_ROT90: x = (p->size_v0 - 1 - vb) + p->u0; y = ub + p->v0; break;
    case OR_ROT180: x = (p->size_u0 - 1 - ub) + p->u0; y = (p->size_v0 - 1 - vb) + p->v0; break;
    case OR_ROT270: x = vb + p->u0; y = (p->size_u0 - 1 - ub) + p->v0; break;
    case OR_MIRROR: x = (p->size_u0 - 1 - ub) + p->u0; y = vb + p->v0; break;
    case OR_MROT90: x = (p->size_v0 - 1 - vb) + p->u0; y = (p->size_u0 - 1 - ub) + p->v0; break;
    case OR_MROT180: x = ub + p->u0; y = (p->size_v0 - 1 - vb) + p->v0; break;
    case OR_SWAP: x = vb + p->u0; y = ub + p->v0; break;
    default: return -1;
  }
  if (x < 0 || y < 0 || x >= bw || y >= bh) return -1;
  return x + bw * y;
}
/* PCCImage::set (PCCImage.h:107-124): video sample of `bd` bits -> the 8-bit value the decoder works with */
static int to8(int v, int bd) { int sh = bd - 8; if (sh <= 0) return v; int r = (v + (1 << (sh - 1))) >> sh; return r > 255 ? 255 : r; }
/* PCCPatch::generatePoint (PCCPatch.h:177-207) */
static void gen_point(const oracle_patch* p, int u, int v, int depth, int16_t out[3]) {
  int n;
  if (p->projection_mode == 0) n = depth + p->d1; else { n = p->d1 - depth; if (n < 0) n = 0; }
  out[p->normal_axis] = (int16_t)n;
  out[p->tangent_axis] = (int16_t)(u * p->lod_x + p->u1);
  out[p->bitangent_axis] = (int16_t)(v * p->lod_y + p->v1);
}

/* PCCCodec::identifyBoundaryPoints (:266-325) for an occupied pixel: 1 when the point ends with boundary point type 1. First layer: an unoccupied pixel among the 8
 * neighbours (each test only for pixels off the respective picture border, as written) or the pixel on the picture border; second layer: an unoccupied pixel in the ring two
 * away, or the pixel one off the picture border. */
static int boundary_point(const uint8_t* om, int x, int y, int W, int H) {
#define OM(xx, yy) om[(size_t)(yy) * W + (xx)]
  int t = 0;
  if (y > 0 && y < H - 1) if (OM(x, y - 1) == 0 || OM(x, y + 1) == 0) t = 1;
  if (x > 0 && x < W - 1 && !t) if (OM(x + 1, y) == 0 || OM(x - 1, y) == 0) t = 1;
  if (y > 0 && y < H - 1 && x > 0 && !t) if (OM(x - 1, y - 1) == 0 || OM(x - 1, y + 1) == 0) t = 1;
  if (y > 0 && y < H - 1 && x < W - 1 && !t) if (OM(x + 1, y - 1) == 0 || OM(x + 1, y + 1) == 0) t = 1;
  if (y == 0 || y == H - 1 || x == 0 || x == W - 1) t = 1;
  if (!t) {
    for (int ix = -2; ix <= 2 && !t; ix++) for (int iy = -2; iy <= 2 && !t; iy++)
      if ((ix > 1 || ix < -1 || iy > 1 || iy < -1) && y + iy >= 0 && y + iy < H && x + ix >= 0 && x + ix < W && OM(x + ix, y + iy) == 0) t = 1;
    if (y == 1 || y == H - 2 || x == 1 || x == W - 2) t = 1;
  }
#undef OM
  return t;
}
/* PCCCodec::smoothPointCloudPostprocess with gridSmoothing (:52-145) + smoothPointCloudGrid (:1065-1104), arithmetic types as in the reference: cell centres are float
 * sums divided by a uint16_t count, the filter works in double. Returns the number of points moved. */
static int smooth_grid(int16_t* xyz, int n, const uint8_t* btype, const uint32_t* part, int g, int threshold) {
  int maxv = xyz[0] > xyz[1] ? xyz[0] : xyz[1]; if (xyz[2] > maxv) maxv = xyz[2];          /* bounding box :70-80: only max_ is used (:81) */
  int bmax[3] = {xyz[0], xyz[1], xyz[2]};
  for (int j = 0; j < n; j++) for (int k = 0; k < 3; k++) if (xyz[3 * j + k] > bmax[k]) bmax[k] = xyz[3 * j + k];
  maxv = bmax[0] > bmax[1] ? bmax[0] : bmax[1]; if (bmax[2] > maxv) maxv = bmax[2];
  const int w = (maxv + g - 1) / g, disth = g / 2 > 1 ? g / 2 : 1, th = g * w;
  if (w <= 0) return 0;
  const size_t w3 = (size_t)w * w * w;
  int* cell = (int*)malloc(w3 * sizeof(int)); for (size_t i = 0; i < w3; i++) cell[i] = -1;
  int n_cells = 0;
#define SKIP(P) ((P)[0] < disth || (P)[1] < disth || (P)[2] < disth || th <= (P)[0] + disth || th <= (P)[1] + disth || th <= (P)[2] + disth)
  for (int i = 0; i < n; i++) if (btype[i] == 1) {                                      /* boundary cells :88-113 */
    const int P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (SKIP(P)) continue;
    int Q[3]; for (int k = 0; k < 3; k++) Q[k] = P[k] / g + ((P[k] % g < g / 2) ? -1 : 0);
    for (int ix = 0; ix < 2; ix++) for (int iy = 0; iy < 2; iy++) for (int iz = 0; iz < 2; iz++) {
      const size_t id = (size_t)(Q[0] + ix) + (size_t)(Q[1] + iy) * w + (size_t)(Q[2] + iz) * w * w;
      if (cell[id] == -1) cell[id] = n_cells++;
    }
  }
  float* centre = (float*)calloc((size_t)(n_cells ? n_cells : 1) * 3, sizeof(float)); uint16_t* count = (uint16_t*)calloc((size_t)(n_cells ? n_cells : 1), 2);
  uint32_t* cpart = (uint32_t*)calloc((size_t)(n_cells ? n_cells : 1), 4); uint8_t* do_smooth = (uint8_t*)calloc((size_t)(n_cells ? n_cells : 1), 1);
  for (int j = 0; j < n; j++) {                                                         /* centroids :122-137, addGridCentroid :980-998 */
    const int P[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
    if (SKIP(P)) continue;
    const size_t id = (size_t)(P[0] / g) + (size_t)(P[1] / g) * w + (size_t)(P[2] / g) * w * w;
    const int c = cell[id]; if (c == -1) continue;
    const uint32_t pidx = part[j] + 1;
    if (count[c] == 0) { cpart[c] = pidx; centre[3 * c] = centre[3 * c + 1] = centre[3 * c + 2] = 0.f; do_smooth[c] = 0; }
    else if (!do_smooth[c] && cpart[c] != pidx) do_smooth[c] = 1;
    for (int k = 0; k < 3; k++) centre[3 * c + k] += (float)P[k];
    count[c]++;
  }
  for (int c = 0; c < n_cells; c++) if (count[c] != 0) for (int k = 0; k < 3; k++) centre[3 * c + k] /= (float)count[c];   /* :138-140 */
  int moved = 0;
  for (int ci = 0; ci < n; ci++) {                                                      /* smoothPointCloudGrid :1065-1104 */
    const int P[3] = {xyz[3 * ci], xyz[3 * ci + 1], xyz[3 * ci + 2]};
    if (SKIP(P) || btype[ci] != 1) continue;
    /* gridFiltering :1000-1063 */
    const int half = g / 2;
    int S[3], Wt[3], idx[2][2][2], other = 0;
    for (int k = 0; k < 3; k++) { const int P2 = P[k] / g, P3 = P[k] - P2 * g; S[k] = P2 + (P3 < half ? -1 : 0); }
    for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
      const int tmp = (S[0] + dx) + (S[1] + dy) * w + (S[2] + dz) * w * w; idx[dz][dy][dx] = tmp;
      if (do_smooth[cell[tmp]] && count[cell[tmp]] != 0) other = 1;
    }
    if (!other) continue;
    double c3[2][2][2][3]; const double cur[3] = {(double)P[0], (double)P[1], (double)P[2]};
    const int g2 = g * 2;
    for (int k = 0; k < 3; k++) Wt[k] = (P[k] - S[k] * g - half) * 2 + 1;
    for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
      const int c = cell[idx[dz][dy][dx]];
      for (int k = 0; k < 3; k++) c3[dz][dy][dx][k] = count[c] > 0 ? (double)centre[3 * c + k] : cur[k];
    }
    const int Q[3] = {g2 - Wt[0], g2 - Wt[1], g2 - Wt[2]};
    int cnt = 0; double c4[3] = {0.0, 0.0, 0.0};
    for (int dz = 0, cc = Q[2]; dz < 2; dz++, cc = Wt[2]) for (int dy = 0, bb = Q[1]; dy < 2; dy++, bb = Wt[1]) for (int dx = 0, aa = Q[0]; dx < 2; dx++, aa = Wt[0]) {
      for (int k = 0; k < 3; k++) { c3[dz][dy][dx][k] *= (double)(aa * bb * cc); c4[k] += c3[dz][dy][dx][k]; }
      cnt += aa * bb * cc * (int)count[cell[idx[dz][dy][dx]]];
    }
    for (int k = 0; k < 3; k++) c4[k] /= (double)(g2 * g2 * g2);
    cnt /= g2 * g2 * g2;
    double centroid[3]; for (int k = 0; k < 3; k++) centroid[k] = c4[k] * (double)cnt;
    if (cnt == 0) continue;                                                             /* the reference divides by zero here: NaN, and the comparison below is false */
    double d[3]; for (int k = 0; k < 3; k++) d[k] = cur[k] * (double)cnt - centroid[k];
    const double dist2 = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / (double)cnt + 0.5;
    if (dist2 >= (double)((threshold > cnt ? threshold : cnt) * 2)) {
      for (int k = 0; k < 3; k++) { const double v = centroid[k] / (double)cnt + 0.5; xyz[3 * ci + k] = (int16_t)(double)(int64_t)v; }
      moved++;
    }
  }
#undef SKIP
  free(cell); free(centre); free(count); free(cpart); free(do_smooth);
  return moved;
}

int oracle_reconstruct(const oracle_atlas* a, const oracle_patch* patches, int n_patches, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, int geo_bd,
                       const uint16_t* t0, const uint16_t* t1, int attr_bd, oracle_cloud* out) {
  const int W = a->width, H = a->height, res = a->occupancy_resolution, prec = a->occupancy_precision, ow = W / prec;
  const int bw = W / res, bh = H / res;
  memset(out, 0, sizeof(*out));
  if (W % res || H % res || W % prec || H % prec || res < 1 || prec < 1) return -1;
  /* generateOccupancyMap (:1584-1606): nearest-neighbour upscale, binarised against thresholdLossyOM */
  uint8_t* om = (uint8_t*)calloc((size_t)W * H, 1);
  for (int v = 0; v < H; v++) for (int u = 0; u < W; u++) om[(size_t)v * W + u] = occ[(size_t)(v / prec) * ow + u / prec] > a->threshold_lossy_om;
  /* generateBlockToPatchFromOccupancyMapVideo (:1725-1763): a block belongs to the LAST patch that has an occupied pixel in it */
  uint32_t* b2p = (uint32_t*)calloc((size_t)bw * bh, sizeof(uint32_t));
  for (int pi = 0; pi < n_patches; pi++) {
    const oracle_patch* p = &patches[pi];
    for (int v0 = 0; v0 < p->size_v0; v0++) for (int u0 = 0; u0 < p->size_u0; u0++) {
      int bi = block2canvas(p, u0, v0, bw, bh), nz = 0;
      if (bi < 0) { free(om); free(b2p); return -2; }
      for (int v1 = 0; v1 < res; v1++) for (int u1 = 0; u1 < res; u1++) {
        int x, y;
        if (!patch2canvas(p, res, u0 * res + u1, v0 * res + v1, W, H, &x, &y)) { free(om); free(b2p); return -2; }
        nz += occ[(size_t)(y / prec) * ow + x / prec] > a->threshold_lossy_om;   /* generateOccupancyMap has binarised the video frame in place by now (PCCCodec.cpp:1599-1600, call order PCCDecoder.cpp:363-376): :1754 reads 0 / 1 */
      }
      if (nz > 0) b2p[bi] = (uint32_t)pi + 1;
    }
  }
  /* generatePointCloud (:628-838) */
  size_t cap = 1024, n = 0;
  int16_t* xyz = (int16_t*)malloc(cap * 6); uint16_t* col = (uint16_t*)malloc(cap * 6);
  uint8_t* btype = (uint8_t*)malloc(cap); uint32_t* part = (uint32_t*)malloc(cap * 4);      /* getBoundaryPointType == 1 (identifyBoundaryPoints on the point's pixel, :962-972), partition[] (:819) */
  const int cw = W / 2;
  for (int pi = 0; pi < n_patches; pi++) {
    const oracle_patch* p = &patches[pi];
    for (int v0 = 0; v0 < p->size_v0; v0++) for (int u0 = 0; u0 < p->size_u0; u0++) {
      if (b2p[block2canvas(p, u0, v0, bw, bh)] != (uint32_t)pi + 1) continue;
      for (int v1 = 0; v1 < res; v1++) for (int u1 = 0; u1 < res; u1++) {
        int u = u0 * res + u1, v = v0 * res + v1, x, y;
        patch2canvas(p, res, u, v, W, H, &x, &y);
        if (!om[(size_t)y * W + x]) continue;
        int16_t pt[2][3];
        gen_point(p, u, v, to8(d0[(size_t)y * W + x], geo_bd), pt[0]);
        int np = 1;
        if (a->map_count > 1) {   /* generatePoints :497-513 */
          if (a->absolute_d1) gen_point(p, u, v, to8(d1[(size_t)y * W + x], geo_bd), pt[1]);
          else { memcpy(pt[1], pt[0], 6); int dv = to8(d1[(size_t)y * W + x], geo_bd); pt[1][p->normal_axis] = (int16_t)(pt[1][p->normal_axis] + (p->projection_mode == 0 ? dv : -dv)); }
          np = 2;
        }
        for (int i = 0; i < np; i++) {
          if (a->remove_duplicate_points && i > 0 && !memcmp(pt[i], pt[0], 6)) continue;   /* :793-794 */
          if (n == cap) { cap *= 2; xyz = (int16_t*)realloc(xyz, cap * 6); col = (uint16_t*)realloc(col, cap * 6); btype = (uint8_t*)realloc(btype, cap); part = (uint32_t*)realloc(part, cap * 4); }
          memcpy(xyz + 3 * n, pt[i], 6);
          btype[n] = (uint8_t)boundary_point(om, x, y, W, H); part[n] = (uint32_t)pi;
          const uint16_t* t = i == 0 ? t0 : t1;    /* colorPointCloud :1417-1421: frame shift + f, pixel (x,y) */
          if (t) { col[3 * n] = t[(size_t)y * W + x]; col[3 * n + 1] = t[(size_t)W * H + (size_t)(y / 2) * cw + x / 2]; col[3 * n + 2] = t[(size_t)W * H + (size_t)cw * (H / 2) + (size_t)(y / 2) * cw + x / 2]; }
          else col[3 * n] = col[3 * n + 1] = col[3 * n + 2] = (uint16_t)(1 << (attr_bd - 1));
          n++;
        }
      }
    }
  }
  out->n = (int)n; out->xyz = xyz; out->yuv = col; out->occupancy_map = om; out->block_to_patch = b2p;
  if (a->geometry_smoothing && n > 0) {
    if (a->grid_size < 2 || a->grid_size > 255) { free(btype); free(part); return -3; }
    out->n_smoothed = smooth_grid(xyz, (int)n, btype, part, a->grid_size, a->threshold_smoothing);
  }
  free(btype); free(part);
  return 0;
}
void oracle_cloud_free(oracle_cloud* c) { free(c->xyz); free(c->yuv); free(c->occupancy_map); free(c->block_to_patch); memset(c, 0, sizeof(*c)); }

/* ---- D1 (point-to-point) ---- */
static int cmp_pt(const void* a, const void* b) {
  const int16_t* p = (const int16_t*)a; const int16_t* q = (const int16_t*)b;
  if (p[2] != q[2]) return p[2] - q[2];
  if (p[1] != q[1]) return p[1] - q[1];
  return p[0] - q[0];
}
/* duplicate points are merged before the comparison (dropDuplicates_ = 2, PCCMetrics.cpp:355-360): sort + unique */
static int16_t* unique_points(const int16_t* p, int n, int* nu) {
  int16_t* q = (int16_t*)malloc((size_t)(n ? n : 1) * 6); memcpy(q, p, (size_t)n * 6);
  qsort(q, (size_t)n, 6, cmp_pt);
  int m = 0;
  for (int i = 0; i < n; i++) if (!m || memcmp(q + 3 * (m - 1), q + 3 * i, 6)) { memmove(q + 3 * m, q + 3 * i, 6); m++; }
  *nu = m; return q;
}
/* sum over the points of A of the squared distance to the nearest point of B: uniform grid of 8^3 cells over B, rings of cells
 * around the query until no closer point can exist (the reference uses a kd-tree, PCCMetrics.cpp:85-97; the nearest distance is the same) */
static uint64_t sse_a_to_b(const int16_t* A, int na, const int16_t* B, int nb, uint64_t* max_d2) {
  enum { CS = 3, G = 1 << (16 - CS) };     /* cells of 8 units; coordinates are taken modulo nothing: they must be 0..65535 >> 3 */
  int gmax = 0;
  for (int i = 0; i < nb; i++) for (int c = 0; c < 3; c++) { int g = B[3 * i + c] >> CS; if (g > gmax) gmax = g; }
  for (int i = 0; i < na; i++) for (int c = 0; c < 3; c++) { int g = A[3 * i + c] >> CS; if (g > gmax) gmax = g; }
  const int D = gmax + 1; (void)G;
  int* start = (int*)calloc((size_t)D * D * D + 1, sizeof(int)); int* order = (int*)malloc(sizeof(int) * (size_t)(nb ? nb : 1));
#define CELL(x, y, z) ((((size_t)(z) * D) + (y)) * D + (x))
  for (int i = 0; i < nb; i++) start[CELL(B[3 * i] >> CS, B[3 * i + 1] >> CS, B[3 * i + 2] >> CS) + 1]++;
  for (size_t i = 0; i < (size_t)D * D * D; i++) start[i + 1] += start[i];
  int* fill = (int*)malloc(sizeof(int) * (size_t)D * D * D); memcpy(fill, start, sizeof(int) * (size_t)D * D * D);
  for (int i = 0; i < nb; i++) order[fill[CELL(B[3 * i] >> CS, B[3 * i + 1] >> CS, B[3 * i + 2] >> CS)]++] = i;
  uint64_t sse = 0, mx = 0;
  for (int i = 0; i < na; i++) {
    const int ax = A[3 * i], ay = A[3 * i + 1], az = A[3 * i + 2], cx = ax >> CS, cy = ay >> CS, cz = az >> CS;
    int64_t best = INT64_MAX;
    for (int r = 0; r < D + 1; r++) {
      for (int z = cz - r; z <= cz + r; z++) for (int y = cy - r; y <= cy + r; y++) for (int x = cx - r; x <= cx + r; x++) {
        if (x < 0 || y < 0 || z < 0 || x >= D || y >= D || z >= D) continue;
        if (abs(x - cx) != r && abs(y - cy) != r && abs(z - cz) != r) continue;   /* the ring only */
        for (int k = start[CELL(x, y, z)]; k < start[CELL(x, y, z) + 1]; k++) {
          const int16_t* b = B + 3 * order[k]; int64_t dx = b[0] - ax, dy = b[1] - ay, dz = b[2] - az, d = dx * dx + dy * dy + dz * dz;
          if (d < best) best = d;
        }
      }
      /* every point in a cell outside ring r is at least r * 8 away along one axis (the query lies inside its own cell) */
      if (best != INT64_MAX && best <= (int64_t)(r << CS) * (r << CS)) break;
    }
    sse += (uint64_t)best; if ((uint64_t)best > mx) mx = (uint64_t)best;
  }
#undef CELL
  free(start); free(order); free(fill);
  *max_d2 = mx;
  return sse;
}
int oracle_d1(const int16_t* a, int na, const int16_t* b, int nb, int peak, oracle_d1_result* out) {
  memset(out, 0, sizeof(*out));
  if (na <= 0 || nb <= 0) return -1;
  for (int i = 0; i < 3 * na; i++) if (a[i] < 0) return -1;
  for (int i = 0; i < 3 * nb; i++) if (b[i] < 0) return -1;
  int ua, ub; int16_t* A = unique_points(a, na, &ua); int16_t* B = unique_points(b, nb, &ub);
  out->n_a = ua; out->n_b = ub;
  out->sse_ab = sse_a_to_b(A, ua, B, ub, &out->max_ab); out->sse_ba = sse_a_to_b(B, ub, A, ua, &out->max_ba);
  free(A); free(B);
  /* QualityMetrics::compute :204-206 (float mse, getPSNR with factor 3), symmetric = the worse direction (:299-309) */
  float mse_ab = (float)((double)out->sse_ab / ua), mse_ba = (float)((double)out->sse_ba / ub);
  out->mse_ab = mse_ab; out->mse_ba = mse_ba;
  float p = (float)peak, m = mse_ab > mse_ba ? mse_ab : mse_ba;
  out->psnr_ab = 10 * log10f(3 * p * p / mse_ab); out->psnr_ba = 10 * log10f(3 * p * p / mse_ba); out->psnr = 10 * log10f(3 * p * p / m);
  return 0;
}

/* ---- D2 (point-to-plane) ---- */
typedef struct { int n; int16_t* p; int* orig; int D; int* start; int* order; } pset;   /* merged points (sorted), the original index each one stands for, cell grid */
static const int16_t* g_sort_pts;
static int cmp_idx(const void* a, const void* b) {
  int i = *(const int*)a, j = *(const int*)b, c = cmp_pt(g_sort_pts + 3 * i, g_sort_pts + 3 * j);
  return c ? c : i - j;
}
#define PCELL(s, x, y, z) ((((size_t)(z) * (s)->D) + (y)) * (s)->D + (x))
static void pset_build(pset* s, const int16_t* p, int n, int gmax) {
  int* idx = (int*)malloc(sizeof(int) * (size_t)n); for (int i = 0; i < n; i++) idx[i] = i;
  g_sort_pts = p; qsort(idx, (size_t)n, sizeof(int), cmp_idx);
  s->p = (int16_t*)malloc((size_t)n * 6); s->orig = (int*)malloc(sizeof(int) * (size_t)n); s->n = 0;
  for (int k = 0; k < n; k++) if (!s->n || memcmp(s->p + 3 * (s->n - 1), p + 3 * idx[k], 6)) { memcpy(s->p + 3 * s->n, p + 3 * idx[k], 6); s->orig[s->n++] = idx[k]; }   /* ties sorted by index: the lowest stands for the point */
  free(idx);
  s->D = gmax + 1;
  size_t nc = (size_t)s->D * s->D * s->D;
  s->start = (int*)calloc(nc + 1, sizeof(int)); s->order = (int*)malloc(sizeof(int) * (size_t)s->n);
  for (int i = 0; i < s->n; i++) s->start[PCELL(s, s->p[3 * i] >> 3, s->p[3 * i + 1] >> 3, s->p[3 * i + 2] >> 3) + 1]++;
  for (size_t i = 0; i < nc; i++) s->start[i + 1] += s->start[i];
  int* fill = (int*)malloc(sizeof(int) * nc); memcpy(fill, s->start, sizeof(int) * nc);
  for (int i = 0; i < s->n; i++) s->order[fill[PCELL(s, s->p[3 * i] >> 3, s->p[3 * i + 1] >> 3, s->p[3 * i + 2] >> 3)]++] = i;
  free(fill);
}
static void pset_free(pset* s) { free(s->p); free(s->orig); free(s->start); free(s->order); }
/* every point of s at the nearest squared distance from q: their indices into s->p (up to cap), returns how many; *d2 = that distance */
static int pset_nearest_ties(const pset* s, const int16_t* q, int* out, int cap, int64_t* d2) {
  const int cx = q[0] >> 3, cy = q[1] >> 3, cz = q[2] >> 3; int64_t best = INT64_MAX; int n = 0;
  for (int r = 0; r < s->D + 1; r++) {
    for (int z = cz - r; z <= cz + r; z++) for (int y = cy - r; y <= cy + r; y++) for (int x = cx - r; x <= cx + r; x++) {
      if (x < 0 || y < 0 || z < 0 || x >= s->D || y >= s->D || z >= s->D) continue;
      if (abs(x - cx) != r && abs(y - cy) != r && abs(z - cz) != r) continue;
      for (int k = s->start[PCELL(s, x, y, z)]; k < s->start[PCELL(s, x, y, z) + 1]; k++) {
        const int16_t* b = s->p + 3 * s->order[k]; int64_t dx = b[0] - q[0], dy = b[1] - q[1], dz = b[2] - q[2], d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; n = 0; }
        if (d == best && n < cap) out[n++] = s->order[k];
      }
    }
    if (best != INT64_MAX && best < (int64_t)(r << 3) * (r << 3)) break;   /* strictly: a point in a later ring may still TIE at r * 8 */
  }
  *d2 = best;
  return n;
}
/* sum over the points of A of the mean, over B's points at the nearest distance, of the squared projection of (a - b) on b's normal; B's normals are acc / cnt (Q14) */
static double d2_a_to_b(const pset* A, const pset* B, const int64_t* accB, const int32_t* cntB, double* mx) {
  double sse = 0, m = 0; int ties[4096];
  for (int i = 0; i < A->n; i++) {
    int64_t d2; int nt = pset_nearest_ties(B, A->p + 3 * i, ties, 4096, &d2); double sum = 0;
    for (int t = 0; t < nt; t++) {
      const int j = ties[t]; const int16_t* b = B->p + 3 * j; const int16_t* a = A->p + 3 * i;
      const int64_t dot = (int64_t)(a[0] - b[0]) * accB[3 * j] + (int64_t)(a[1] - b[1]) * accB[3 * j + 1] + (int64_t)(a[2] - b[2]) * accB[3 * j + 2];
      const double v = (double)dot / (double)cntB[j];
      sum += v * v;
    }
    const double dist = sum / nt / (16384.0 * 16384.0);
    sse += dist; if (dist > m) m = dist;
  }
  *mx = m;
  return sse;
}
int oracle_d2(const int16_t* a, const int16_t* normals_a, int na, const int16_t* b, int nb, int peak, oracle_d2_result* out) {
  memset(out, 0, sizeof(*out));
  if (na <= 0 || nb <= 0) return -1;
  int gmax = 0;
  for (int i = 0; i < 3 * na; i++) { if (a[i] < 0) return -1; if ((a[i] >> 3) > gmax) gmax = a[i] >> 3; }
  for (int i = 0; i < 3 * nb; i++) { if (b[i] < 0) return -1; if ((b[i] >> 3) > gmax) gmax = b[i] >> 3; }
  pset A, B; pset_build(&A, a, na, gmax); pset_build(&B, b, nb, gmax);
  out->n_a = A.n; out->n_b = B.n;
  /* normals: A's are given (copyNormals); B's by scaleNormals - every source point gives its normal to the points of B nearest to it, a point of B that got none
   * takes the mean of the source points nearest to it. Kept as integer sum and count: the mean is formed where it is used. */
  int64_t* accA = (int64_t*)calloc((size_t)A.n * 3, sizeof(int64_t)); int32_t* cntA = (int32_t*)calloc((size_t)A.n, sizeof(int32_t));
  int64_t* accB = (int64_t*)calloc((size_t)B.n * 3, sizeof(int64_t)); int32_t* cntB = (int32_t*)calloc((size_t)B.n, sizeof(int32_t));
  for (int i = 0; i < A.n; i++) { for (int c = 0; c < 3; c++) accA[3 * i + c] = normals_a[3 * A.orig[i] + c]; cntA[i] = 1; }
  int ties[4096]; int64_t d2;
  for (int i = 0; i < A.n; i++) {
    int nt = pset_nearest_ties(&B, A.p + 3 * i, ties, 4096, &d2);
    for (int t = 0; t < nt; t++) { for (int c = 0; c < 3; c++) accB[3 * ties[t] + c] += accA[3 * i + c]; cntB[ties[t]]++; }
  }
  for (int j = 0; j < B.n; j++) if (!cntB[j]) {
    int nt = pset_nearest_ties(&A, B.p + 3 * j, ties, 4096, &d2);
    for (int t = 0; t < nt; t++) { for (int c = 0; c < 3; c++) accB[3 * j + c] += accA[3 * ties[t] + c]; cntB[j]++; }
  }
  out->sse_ab = d2_a_to_b(&A, &B, accB, cntB, &out->max_ab);
  out->sse_ba = d2_a_to_b(&B, &A, accA, cntA, &out->max_ba);
  free(accA); free(cntA); free(accB); free(cntB);
  float mse_ab = (float)(out->sse_ab / A.n), mse_ba = (float)(out->sse_ba / B.n);
  out->mse_ab = mse_ab; out->mse_ba = mse_ba;
  float p = (float)peak, m = mse_ab > mse_ba ? mse_ab : mse_ba;
  out->psnr_ab = 10 * log10f(3 * p * p / mse_ab); out->psnr_ba = 10 * log10f(3 * p * p / mse_ba); out->psnr = 10 * log10f(3 * p * p / m);
  pset_free(&A); pset_free(&B);
  return 0;
}
