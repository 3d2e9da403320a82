// Host side of the verification stage (csrc/rbt_pcc.h): argument checks, uploads, launches, the point list back to the host.
// Replaces the decoder-side loops of PCCCodec::generatePointCloud (PCCCodec.cpp:517-978) and QualityMetrics::compute (PCCMetrics.cpp:75-231).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rbt_batch.h"
#include "rbt_pcc.h"
#include "../csrc/rbt_pcc.h"

namespace rbt {
namespace {
struct DevBuf { void* p = nullptr; ~DevBuf() { rbtk::dev_free(p); } bool alloc(size_t n) { p = rbtk::dev_alloc(n); return p != nullptr; } template <class T> T* as() { return (T*)p; } };
int block_xy(const rbt_patch& p, int ub, int vb, int* x, int* y) {       // PCCPatch::patchBlock2CanvasBlock
  switch (p.orientation) {
    case RBT_OR_DEFAULT: *x = ub + p.u0; *y = vb + p.v0; return 1;
    case RBT_OR_ROT90: *x = (p.size_v0 - 1 - vb) + p.u0; *y = ub + p.v0; return 1;
    case RBT_OR_ROT180: *x = (p.size_u0 - 1 - ub) + p.u0; *y = (p.size_v0 - 1 - vb) + p.v0; return 1;
    case RBT_OR_ROT270: *x = vb + p.u0; *y = (p.size_u0 - 1 - ub) + p.v0; return 1;
    case RBT_OR_MIRROR: *x = (p.size_u0 - 1 - ub) + p.u0; *y = vb + p.v0; return 1;
    case RBT_OR_MROT90: *x = (p.size_v0 - 1 - vb) + p.u0; *y = (p.size_u0 - 1 - ub) + p.v0; return 1;
    case RBT_OR_MROT180: *x = ub + p.u0; *y = (p.size_v0 - 1 - vb) + p.v0; return 1;
    case RBT_OR_SWAP: *x = vb + p.u0; *y = ub + p.v0; return 1;
  }
  return 0;
}
}  // namespace

int pcc_reconstruct(std::string& err, const rbt_atlas_params* a, const rbt_patch* patches, int n_patches, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, int geo_bd,
                    const uint16_t* t0, const uint16_t* t1, int attr_bd, rbt_cloud* out) {
  memset(out, 0, sizeof(*out));
  const int W = a->width, H = a->height, res = a->occupancy_resolution, prec = a->occupancy_precision;
  if (W <= 0 || H <= 0 || res < 1 || prec < 1 || W % res || H % res || W % prec || H % prec || W % 2 || H % 2 || W > 8192 || H > 8192 || n_patches < 0 || n_patches > 65535 ||
      geo_bd < 8 || geo_bd > 16 || attr_bd < 8 || attr_bd > 16 || a->map_count < 1 || a->map_count > 2 || (a->map_count > 1 && !d1) || ((t0 != nullptr) != (t1 != nullptr) && a->map_count > 1) ||
      (a->geometry_smoothing && (a->grid_size < 2 || a->grid_size > 255 || a->threshold_smoothing < 0))) {
    err = "bad atlas parameters"; return RBT_ERR_PARAM; }
  const bool smooth = a->geometry_smoothing != 0;
  RbtPccParams P; memset(&P, 0, sizeof(P));
  P.w = W; P.h = H; P.res = res; P.prec = prec; P.map_count = a->map_count; P.absolute_d1 = a->absolute_d1; P.remove_dup = a->remove_duplicate_points; P.threshold = a->threshold_lossy_om;
  P.geo_bd = geo_bd; P.attr_bd = attr_bd; P.bw = W / res; P.bh = H / res; P.ow = W / prec; P.n_patches = n_patches; P.has_attr = t0 != nullptr;
  // every patch block must lie on the canvas (the reference exits otherwise, PCCPatch.cpp:238-245); items in the reference's visiting order
  std::vector<uint32_t> items;
  for (int pi = 0; pi < n_patches; pi++) {
    const rbt_patch& p = patches[pi];
    const bool swapped = p.orientation == RBT_OR_ROT90 || p.orientation == RBT_OR_ROT270 || p.orientation == RBT_OR_MROT90 || p.orientation == RBT_OR_SWAP;
    if (p.size_u0 < 1 || p.size_v0 < 1 || (long long)p.size_u0 * p.size_v0 > 65535 || p.orientation < 0 || p.orientation > RBT_OR_ROT90 || p.u0 < 0 || p.v0 < 0 ||
        p.u0 + (swapped ? p.size_v0 : p.size_u0) > P.bw || p.v0 + (swapped ? p.size_u0 : p.size_v0) > P.bh ||
        p.normal_axis < 0 || p.normal_axis > 2 || p.tangent_axis < 0 || p.tangent_axis > 2 || p.bitangent_axis < 0 || p.bitangent_axis > 2 ||
        p.normal_axis == p.tangent_axis || p.normal_axis == p.bitangent_axis || p.tangent_axis == p.bitangent_axis) { err = "patch outside the atlas or malformed"; return RBT_ERR_PARAM; }
    for (int vb = 0; vb < p.size_v0; vb++) for (int ub = 0; ub < p.size_u0; ub++) { int x, y; block_xy(p, ub, vb, &x, &y); if (x < 0 || y < 0 || x >= P.bw || y >= P.bh) { err = "patch block outside the atlas"; return RBT_ERR_PARAM; }
      items.push_back((uint32_t)pi << 16 | (uint32_t)(vb * p.size_u0 + ub)); }
  }
  const int n_items = (int)items.size();
  const size_t ys = (size_t)W * H, os = (size_t)(W / prec) * (H / prec), fs = ys * 3 / 2;
  DevBuf b_occ, b_d0, b_d1, b_t0, b_t1, b_patches, b_items, b_b2p, b_counts, b_off, b_om, b_xyz, b_yuv, b_meta, b_cells, b_scal;
  if (!b_occ.alloc(os * 2) || !b_d0.alloc(ys * 2) || !b_d1.alloc(ys * 2) || !b_patches.alloc(sizeof(rbt_patch) * (size_t)(n_patches ? n_patches : 1)) || !b_items.alloc(4 * (size_t)(n_items ? n_items : 1)) ||
      !b_b2p.alloc(4 * (size_t)P.bw * P.bh) || !b_counts.alloc(4 * (size_t)(n_items + 1)) || !b_off.alloc(4 * (size_t)(n_items + 1)) || !b_om.alloc(ys) ||
      (t0 && (!b_t0.alloc(fs * 2) || !b_t1.alloc(fs * 2)))) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
  int bad = rbtk::h2d(b_occ.p, occ, os * 2) | rbtk::h2d(b_d0.p, d0, ys * 2) | rbtk::h2d(b_d1.p, d1 ? d1 : d0, ys * 2);
  if (n_patches) bad |= rbtk::h2d(b_patches.p, patches, sizeof(rbt_patch) * (size_t)n_patches);
  if (n_items) bad |= rbtk::h2d(b_items.p, items.data(), 4 * (size_t)n_items);
  if (t0) bad |= rbtk::h2d(b_t0.p, t0, fs * 2) | rbtk::h2d(b_t1.p, t1 ? t1 : t0, fs * 2);
  bad |= rbtk::dev_memset(b_b2p.p, 0, 4 * (size_t)P.bw * P.bh);
  if (bad) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  rbtk::launch_pcc_occmap(&P, b_occ.as<uint16_t>(), b_om.as<uint8_t>());
  rbtk::launch_pcc_owner(&P, b_patches.as<rbt_patch>(), b_items.as<uint32_t>(), n_items, b_occ.as<uint16_t>(), b_b2p.as<uint32_t>());
  rbtk::launch_pcc_count(&P, b_patches.as<rbt_patch>(), b_items.as<uint32_t>(), n_items, b_occ.as<uint16_t>(), b_d0.as<uint16_t>(), b_d1.as<uint16_t>(), b_b2p.as<uint32_t>(), b_counts.as<uint32_t>());
  rbtk::launch_scan_u32(b_counts.as<uint32_t>(), b_off.as<uint32_t>(), n_items);
  uint32_t total = 0;
  if (rbtk::d2h(&total, b_off.as<uint32_t>() + n_items, 4)) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  if (!b_xyz.alloc(6 * (size_t)(total ? total : 1)) || !b_yuv.alloc(6 * (size_t)(total ? total : 1)) || (smooth && (!b_meta.alloc(4 * (size_t)(total ? total : 1)) || !b_scal.alloc(64)))) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
  rbtk::launch_pcc_emit(&P, b_patches.as<rbt_patch>(), b_items.as<uint32_t>(), n_items, b_occ.as<uint16_t>(), b_d0.as<uint16_t>(), b_d1.as<uint16_t>(), b_t0.as<uint16_t>(), b_t1.as<uint16_t>(),
                        b_b2p.as<uint32_t>(), b_off.as<uint32_t>(), b_xyz.as<int16_t>(), b_yuv.as<uint16_t>(), smooth ? b_om.as<uint8_t>() : nullptr, smooth ? b_meta.as<uint32_t>() : nullptr);
  out->n_points = (int)total;
  // geometry smoothing (PCCCodec::smoothPointCloudPostprocess with gridSmoothing, PCCCodec.cpp:52-145): the grid spans the largest coordinate of the cloud (:70-83)
  if (smooth && total) {
    uint32_t* scal = b_scal.as<uint32_t>();                        // [0] largest coordinate, [1] points moved
    uint32_t maxv = 0;
    if (rbtk::dev_memset(scal, 0, 64)) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
    rbtk::launch_sm_max(b_xyz.as<int16_t>(), (int)total, scal);
    if (rbtk::d2h(&maxv, scal, 4)) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
    RbtSmooth G; memset(&G, 0, sizeof(G));
    G.g = a->grid_size; G.w = ((int)maxv + G.g - 1) / G.g; G.disth = std::max(G.g / 2, 1); G.th = G.g * G.w; G.threshold = a->threshold_smoothing; G.n_points = (int)total;
    if (G.w > 0) {
      const size_t w3 = (size_t)G.w * G.w * G.w, o_sum = (w3 + 255) & ~(size_t)255, o_cnt = o_sum + 12 * w3, o_min = o_cnt + 4 * w3, o_max = o_min + 4 * w3, bytes = o_max + 4 * w3;
      if (!b_cells.alloc(bytes)) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
      uint8_t* base = b_cells.as<uint8_t>();
      G.flag = base; G.sum = (uint32_t*)(base + o_sum); G.cnt = (uint32_t*)(base + o_cnt); G.pmin = (uint32_t*)(base + o_min); G.pmax = (uint32_t*)(base + o_max); G.moved = scal + 1;
      if (rbtk::dev_memset(base, 0, bytes) || rbtk::dev_memset(G.pmin, 0xFF, 4 * w3)) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
      rbtk::launch_sm_passes(&G, b_xyz.as<int16_t>(), b_meta.as<uint32_t>());
      uint32_t moved = 0;
      if (rbtk::d2h(&moved, scal + 1, 4)) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
      out->n_smoothed = (int)moved;
    }
  }
  out->xyz = (int16_t*)malloc(6 * (size_t)(total ? total : 1)); out->yuv = (uint16_t*)malloc(6 * (size_t)(total ? total : 1));
  out->occupancy_map = (uint8_t*)malloc(ys); out->block_to_patch = (uint32_t*)malloc(4 * (size_t)P.bw * P.bh);
  if (!out->xyz || !out->yuv || !out->occupancy_map || !out->block_to_patch) { err = "out of memory"; return RBT_ERR_NOMEM; }
  bad = rbtk::d2h(out->occupancy_map, b_om.p, ys) | rbtk::d2h(out->block_to_patch, b_b2p.p, 4 * (size_t)P.bw * P.bh);
  if (total) bad |= rbtk::d2h(out->xyz, b_xyz.p, 6 * (size_t)total) | rbtk::d2h(out->yuv, b_yuv.p, 6 * (size_t)total);
  if (bad || rbtk::dev_sync()) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  return RBT_OK;
}

int pcc_d1(std::string& err, const int16_t* a, int na, const int16_t* b, int nb, int peak, rbt_d1_result* out) {
  memset(out, 0, sizeof(*out));
  if (na <= 0 || nb <= 0 || peak <= 0) { err = "empty point cloud"; return RBT_ERR_PARAM; }
  for (int i = 0; i < 3 * na; i++) if (a[i] < 0 || a[i] >= RBT_PCC_DIM) { err = "coordinate outside 0..1023"; return RBT_ERR_PARAM; }
  for (int i = 0; i < 3 * nb; i++) if (b[i] < 0 || b[i] >= RBT_PCC_DIM) { err = "coordinate outside 0..1023"; return RBT_ERR_PARAM; }
  const size_t vol_bytes = (size_t)1 << (3 * RBT_PCC_BITS - 3);
  DevBuf va, vb, pa, pb, fa, fb, acc;
  if (!va.alloc(vol_bytes) || !vb.alloc(vol_bytes) || !pa.alloc(6 * (size_t)na) || !pb.alloc(6 * (size_t)nb) || !fa.alloc((size_t)na) || !fb.alloc((size_t)nb) || !acc.alloc(64)) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
  int bad = rbtk::dev_memset(va.p, 0, vol_bytes) | rbtk::dev_memset(vb.p, 0, vol_bytes) | rbtk::dev_memset(acc.p, 0, 64) | rbtk::h2d(pa.p, a, 6 * (size_t)na) | rbtk::h2d(pb.p, b, 6 * (size_t)nb);
  if (bad) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  // acc: [0] sse_ab (u64), [1] sse_ba (u64), then u32: [4] unique a, [5] unique b, [6] max ab, [7] max ba
  unsigned long long* a64 = acc.as<unsigned long long>(); uint32_t* a32 = acc.as<uint32_t>();
  rbtk::launch_vol_set(pa.as<int16_t>(), na, va.as<uint32_t>(), fa.as<uint8_t>(), a32 + 4);
  rbtk::launch_vol_set(pb.as<int16_t>(), nb, vb.as<uint32_t>(), fb.as<uint8_t>(), a32 + 5);
  rbtk::launch_vol_nn(pa.as<int16_t>(), fa.as<uint8_t>(), na, vb.as<uint32_t>(), a64, a32 + 6);
  rbtk::launch_vol_nn(pb.as<int16_t>(), fb.as<uint8_t>(), nb, va.as<uint32_t>(), a64 + 1, a32 + 7);
  uint64_t h[8];
  if (rbtk::d2h(h, acc.p, 64) || rbtk::dev_sync()) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  const uint32_t* h32 = (const uint32_t*)h;
  out->sse_ab = h[0]; out->sse_ba = h[1]; out->n_a = (int)h32[4]; out->n_b = (int)h32[5]; out->max_ab = h32[6]; out->max_ba = h32[7];
  // QualityMetrics::compute :204-206: float mse, getPSNR with factor 3; symmetric = the worse direction (:299-309)
  out->mse_ab = (float)((double)out->sse_ab / out->n_a); out->mse_ba = (float)((double)out->sse_ba / out->n_b);
  const float p = (float)peak, m = out->mse_ab > out->mse_ba ? out->mse_ab : out->mse_ba;
  out->psnr_ab = 10 * log10f(3 * p * p / out->mse_ab); out->psnr_ba = 10 * log10f(3 * p * p / out->mse_ba); out->psnr = 10 * log10f(3 * p * p / m);
  return RBT_OK;
}

// QualityMetrics::compute with computeC2p_ (PCCMetrics.cpp:100-124, :213-215) both ways (:299-309), normals as PCCMetrics::compute arranges them (:371-376): csrc/rbt_pcc.h
int pcc_d2(std::string& err, const int16_t* a, const int16_t* normals_a, int na, const int16_t* b, int nb, int peak, rbt_d2_result* out) {
  memset(out, 0, sizeof(*out));
  if (na <= 0 || nb <= 0 || peak <= 0 || !normals_a) { err = "empty point cloud or no normals"; return RBT_ERR_PARAM; }
  for (int i = 0; i < 3 * na; i++) if (a[i] < 0 || a[i] >= RBT_PCC_DIM) { err = "coordinate outside 0..1023"; return RBT_ERR_PARAM; }
  for (int i = 0; i < 3 * nb; i++) if (b[i] < 0 || b[i] >= RBT_PCC_DIM) { err = "coordinate outside 0..1023"; return RBT_ERR_PARAM; }
  const size_t vol_bytes = (size_t)1 << (3 * RBT_PCC_BITS - 3);
  auto lg_of = [](int n) { int lg = 4; while (((size_t)1 << lg) < 2 * (size_t)n) lg++; return lg; };
  const int lga = lg_of(na), lgb = lg_of(nb);
  DevBuf va, vb, pa, pb, na_, ka, kb, wa, wb, accb, cntb, res;
  if (!va.alloc(vol_bytes) || !vb.alloc(vol_bytes) || !pa.alloc(6 * (size_t)na) || !pb.alloc(6 * (size_t)nb) || !na_.alloc(6 * (size_t)na) || !ka.alloc((size_t)4 << lga) || !wa.alloc((size_t)4 << lga) ||
      !kb.alloc((size_t)4 << lgb) || !wb.alloc((size_t)4 << lgb) || !accb.alloc(24 * (size_t)nb) || !cntb.alloc(4 * (size_t)nb) || !res.alloc(64)) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
  int bad = rbtk::dev_memset(va.p, 0, vol_bytes) | rbtk::dev_memset(vb.p, 0, vol_bytes) | rbtk::dev_memset(ka.p, 0, (size_t)4 << lga) | rbtk::dev_memset(kb.p, 0, (size_t)4 << lgb) |
            rbtk::dev_memset(wa.p, 0xFF, (size_t)4 << lga) | rbtk::dev_memset(wb.p, 0xFF, (size_t)4 << lgb) | rbtk::dev_memset(accb.p, 0, 24 * (size_t)nb) | rbtk::dev_memset(cntb.p, 0, 4 * (size_t)nb) |
            rbtk::dev_memset(res.p, 0, 64) | rbtk::h2d(pa.p, a, 6 * (size_t)na) | rbtk::h2d(pb.p, b, 6 * (size_t)nb) | rbtk::h2d(na_.p, normals_a, 6 * (size_t)na);
  if (bad) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  rbtk::launch_d2_insert(pa.as<int16_t>(), na, va.as<uint32_t>(), ka.as<uint32_t>(), wa.as<uint32_t>(), lga);
  rbtk::launch_d2_insert(pb.as<int16_t>(), nb, vb.as<uint32_t>(), kb.as<uint32_t>(), wb.as<uint32_t>(), lgb);
  RbtD2Set A{pa.as<int16_t>(), na, va.as<uint32_t>(), ka.as<uint32_t>(), wa.as<uint32_t>(), lga}, B{pb.as<int16_t>(), nb, vb.as<uint32_t>(), kb.as<uint32_t>(), wb.as<uint32_t>(), lgb};
  rbtk::launch_d2_give(&A, na_.as<int16_t>(), &B, accb.as<long long>(), cntb.as<int32_t>());
  rbtk::launch_d2_take(&B, &A, na_.as<int16_t>(), accb.as<long long>(), cntb.as<int32_t>());
  double* r = res.as<double>();
  rbtk::launch_d2_dist(&A, &B, accb.as<long long>(), cntb.as<int32_t>(), nullptr, r);
  rbtk::launch_d2_dist(&B, &A, nullptr, nullptr, na_.as<int16_t>(), r + 4);
  double h[8];
  if (rbtk::d2h(h, res.p, 64) || rbtk::dev_sync()) { err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  uint64_t cnt_a, cnt_b; memcpy(&cnt_a, &h[2], 8); memcpy(&cnt_b, &h[6], 8);
  out->n_a = (int)cnt_a; out->n_b = (int)cnt_b; out->sse_ab = h[0]; out->max_ab = h[1]; out->sse_ba = h[4]; out->max_ba = h[5];
  out->mse_ab = (float)(out->sse_ab / out->n_a); out->mse_ba = (float)(out->sse_ba / out->n_b);
  const float p = (float)peak, m = out->mse_ab > out->mse_ba ? out->mse_ab : out->mse_ba;
  out->psnr_ab = 10 * log10f(3 * p * p / out->mse_ab); out->psnr_ba = 10 * log10f(3 * p * p / out->mse_ba); out->psnr = 10 * log10f(3 * p * p / m);
  return RBT_OK;
}
}  // namespace rbt
