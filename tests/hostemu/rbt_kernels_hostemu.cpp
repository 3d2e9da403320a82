// TEST-ONLY serial stand-in of csrc/rbt_kernels.h: runs the kernel bodies (rbt_parse.h, rbt_recon.h, rbt_filter.h,
// rbt_encode.h) as plain host code so their logic can be debugged against the oracle in a container without a GPU.
// Built into tests/hostemu/librbt_hostemu.so by tests/hostemu/Makefile. It is NOT part of the product: librbt.so is
// linked against csrc/rbt_kernels.hip only and has no CPU path.
#ifndef RBT_HOSTEMU
#error "build with -DRBT_HOSTEMU"
#endif
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "../../rabbit-transcoding_amd/csrc/rbt_kernels.h"
#include "../../rabbit-transcoding_amd/csrc/rbt_parse.h"
#include "../../rabbit-transcoding_amd/csrc/rbt_recon.h"
#include "../../rabbit-transcoding_amd/csrc/rbt_filter.h"
#include "../../rabbit-transcoding_amd/csrc/rbt_encode.h"
#include "../../rabbit-transcoding_amd/csrc/rbt_pcc.h"
#include "../../rabbit-transcoding_amd/host/rbt_hls.h"

namespace rbtk {
static double g_t[32][2];
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int dev_init(int) { return 0; }
int dev_select(int) { return 0; }
void set_stream(int) {}
void map_lane(int, int) {}
void stream_wait(int, int) {}
int stream_mark(int) { return 0; }
void stream_wait_mark(int, int) {}
const char* dev_name() { return "host emulation (test only)"; }
// poisoned: the product recycles device arenas, so whatever a kernel body reads without having written it is another job's data - the parity tests run on a pattern that shows it
// RBT_HOSTEMU_HBM_MB (read at every call, so a test can change it): a device memory of that many megabytes - what is handed out is counted, an allocation that does not fit
// next to the reserve fails like the product's dev_alloc does (RBT_ERR_NOMEM paths of the host code: tests/test_memory_bound.py)
static std::mutex g_mem_mu; static std::map<void*, size_t> g_mem_live; static size_t g_mem_used = 0; static thread_local size_t t_alloc_total = 0;
static size_t hbm_total() { const char* e = getenv("RBT_HOSTEMU_HBM_MB"); return e && *e ? (size_t)atoll(e) << 20 : (size_t)288 << 30; }
size_t dev_reserve_bytes() { const char* e = getenv("RBT_HBM_RESERVE_MB"); return e && *e ? (size_t)atoll(e) << 20 : 0; }
size_t dev_alloc_total() { return t_alloc_total; }
int dev_mem_info(size_t* free_b, size_t* total_b, size_t* cached_b, size_t* live_b) {
  std::lock_guard<std::mutex> lk(g_mem_mu); const size_t tot = hbm_total();
  if (free_b) *free_b = tot > g_mem_used ? tot - g_mem_used : 0; if (total_b) *total_b = tot; if (cached_b) *cached_b = 0; if (live_b) *live_b = g_mem_used;
  return 0;
}
void* dev_alloc(size_t n) {
  if (!n) n = 1;
  { std::lock_guard<std::mutex> lk(g_mem_mu); if (n >= ((size_t)1 << 20) && g_mem_used + n + dev_reserve_bytes() > hbm_total()) return nullptr; }
  void* p = malloc(n); if (!p) return nullptr;
  memset(p, 0xA5, n);
  std::lock_guard<std::mutex> lk(g_mem_mu); g_mem_live[p] = n; g_mem_used += n; t_alloc_total += n;
  return p;
}
void dev_free(void* p) { if (!p) return; { std::lock_guard<std::mutex> lk(g_mem_mu); auto it = g_mem_live.find(p); if (it != g_mem_live.end()) { g_mem_used -= it->second; g_mem_live.erase(it); } } free(p); }
void dev_release_pool() {}
int h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); return 0; }
int d2h(void* h, const void* d, size_t n) { memcpy(h, d, n); return 0; }
int dev_memset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static int g_depth = 1;
void set_jobs_in_flight(int d) { g_depth = d; }
int jobs_in_flight() { return g_depth; }
int dev_sync() { return 0; }
void timer_begin(int id) { g_t[id][0] = now_ms(); }
void timer_end(int id) { g_t[id][1] = now_ms(); }
double timer_ms(int id) { return g_t[id][1] - g_t[id][0]; }

void launch_parse(RbtFrame* frames, RbtSlice* slices, const uint8_t* rbsp, const int32_t* slice_list, int n_slices, int max_w4, void* save, int row_limit, uint32_t*) {
  alignas(16) static uint32_t plds[(RBT_PARSE_LDS_BYTES(RBT_PARSE_CAP4_L) + 3) / 4];
  const int cap4 = max_w4 <= RBT_PARSE_CAP4_S ? RBT_PARSE_CAP4_S : max_w4 <= RBT_PARSE_CAP4_M ? RBT_PARSE_CAP4_M : RBT_PARSE_CAP4_L;
  for (int i = 0; i < n_slices; i++) rbt_parse_slice(frames, slices, slice_list[i], rbsp, (RbtParseLds*)plds, cap4, (RbtParseSave*)save, row_limit);
}
void launch_parse_tasks(const RbtParseTask* tasks, int n_tasks, int max_w4, uint32_t*) {
  alignas(16) static uint32_t plds[(RBT_PARSE_LDS_BYTES(RBT_PARSE_CAP4_L) + 3) / 4];
  const int cap4 = max_w4 <= RBT_PARSE_CAP4_S ? RBT_PARSE_CAP4_S : max_w4 <= RBT_PARSE_CAP4_M ? RBT_PARSE_CAP4_M : RBT_PARSE_CAP4_L;
  for (int i = 0; i < n_tasks; i++) rbt_parse_slice(tasks[i].frames, tasks[i].slices, tasks[i].slice, tasks[i].rbsp, (RbtParseLds*)plds, cap4, nullptr, 0);
}
size_t parse_save_bytes() { return sizeof(RbtParseSave); }
void launch_recon_refs(const RbtFrameRef* refs, int n_frames, int max_w_ctb, int max_h_ctb) {
  static RbtReconCtbLds lds;
  for (int d = 0; d <= max_w_ctb - 1 + 2 * (max_h_ctb - 1); d++)
    for (int k = 0; k < n_frames; k++) {
      RbtFrame* frames = refs[k].frames; int fi = refs[k].frame; const RbtStreamCfg* g = &frames[fi].cfg;
      for (int y = 0; y < g->h_ctb; y++) {
        int x = d - 2 * y; if (x < 0 || x >= g->w_ctb) continue;
        int addr = y * g->w_ctb + x;
        if (frames[fi].ctb_slice[addr] == 0xFFFF) continue;
        rbt_recon_ctb<RC_ROLE_LUMA>(frames, refs[k].slices, fi, addr, &lds.t, &lds.role[0]);     // the two roles are independent:
        rbt_recon_ctb<RC_ROLE_CHROMA>(frames, refs[k].slices, fi, addr, &lds.t, &lds.role[1]);   // one after the other emulates them
      }
    }
}
// RBT_HOSTEMU_CMD_STATS=1: what the decoder's command lists hold (tools/cmd_stats.py): per transform unit size x prediction kind x coded planes, printed at exit
static long long g_cmd_stats[2][6][3][8]; static bool g_cmd_stats_on = false;
static void cmd_stats_print() {
  for (int in = 0; in < 2; in++) for (int l = 2; l < 6; l++) for (int m = 0; m < 3; m++) for (int c = 0; c < 8; c++)
    if (g_cmd_stats[in][l][m][c]) fprintf(stderr, "cmd_stats intra=%d log2=%d mode=%s cbf_y=%d chroma=%d cbf_c=%d n=%lld\n", in, l, m == 0 ? "planar" : m == 1 ? "dc" : "angular", c & 1, (c >> 1) & 1, (c >> 2) & 1, g_cmd_stats[in][l][m][c]);
}
static void cmd_stats_ctb(const RbtFrame* f, int addr) {
  static int init = 0;
  if (!init) { init = 1; const char* e = getenv("RBT_HOSTEMU_CMD_STATS"); g_cmd_stats_on = e && *e == '1'; if (g_cmd_stats_on) atexit(cmd_stats_print); }
  if (!g_cmd_stats_on) return;
  const RbtCmd* c = f->cmds + (size_t)addr * f->cmd_cap; uint32_t n = f->cmd_count[addr];
  for (uint32_t k = 0; k < n; k++) if (c[k].type == RBT_CMD_TU) {
    const int fl = c[k].a, in = (fl & RBT_TU_INTRA) != 0, m = c[k].b == 0 ? 0 : c[k].b == 1 ? 1 : 2;
    g_cmd_stats[in][c[k].log2][m][(fl & RBT_TU_CBF_Y ? 1 : 0) | (fl & RBT_TU_CHROMA ? 2 : 0) | (fl & (RBT_TU_CBF_CB | RBT_TU_CBF_CR) ? 4 : 0)]++;
  }
}
void launch_recon_level(const RbtFrameRef* refs, int n_frames, int max_ctbs, uint32_t*) {
  static RbtReconCtbLds lds;
  for (int i = 0; i < max_ctbs; i++) for (int k = 0; k < n_frames; k++) {      // ticket order: CTB i of every picture, then CTB i + 1
    RbtFrame* frames = refs[k].frames; int fi = refs[k].frame; const RbtStreamCfg* g = &frames[fi].cfg;
    if (i >= g->w_ctb * g->h_ctb) continue;
    int xy = (int)refs[k].order[i], addr = (xy >> 16) * g->w_ctb + (xy & 0xFFFF);
    if (frames[fi].ctb_slice[addr] == 0xFFFF) continue;
    cmd_stats_ctb(&frames[fi], addr);
    rbt_recon_ctb<RC_ROLE_LUMA>(frames, refs[k].slices, fi, addr, &lds.t, &lds.role[0]);
    rbt_recon_ctb<RC_ROLE_CHROMA>(frames, refs[k].slices, fi, addr, &lds.t, &lds.role[1]);
  }
}
void launch_recon_queue(const RbtFrameRef* refs, int n_frames, uint32_t total, uint32_t* qmem, int) {
  // the kernel's queue discipline with one worker: seeds first, then whatever becomes ready, in that order; arrival counts in ctb_done as on the device.
  // Ends short (pictures incomplete, caught by the callers' comparisons) if the graph ever failed to make a CTB ready.
  static RbtReconCtbLds lds;
  uint32_t* q = qmem + 16; uint32_t head = 0, tail = 0;
  while (head < total) {
    uint32_t task;
    if (head < (uint32_t)n_frames) task = (head << 18) + 1; else { if (head - (uint32_t)n_frames >= tail) break; task = q[head - (uint32_t)n_frames]; }
    head++;
    const int fr = (int)((task - 1) >> 18), addr = (int)((task - 1) & 0x3FFFF);
    RbtFrame* frames = refs[fr].frames; const int fi = refs[fr].frame; RbtFrame* f = &frames[fi];
    if (f->ctb_slice[addr] != 0xFFFF) {
      rbt_recon_ctb<RC_ROLE_LUMA>(frames, refs[fr].slices, fi, addr, &lds.t, &lds.role[0]);
      rbt_recon_ctb<RC_ROLE_CHROMA>(frames, refs[fr].slices, fi, addr, &lds.t, &lds.role[1]);
    }
    const int w = f->cfg.w_ctb, h = f->cfg.h_ctb; int succ[3]; const int ns = rc_ctb_successors(w, h, addr % w, addr / w, succ);
    for (int k = 0; k < ns; k++) if ((int)++f->ctb_done[2 * succ[k]] == rc_ctb_need(succ[k] % w, succ[k] / w)) q[tail++] = ((uint32_t)fr << 18 | (uint32_t)succ[k]) + 1;
  }
  qmem[0] = head; qmem[1] = tail;
}
void launch_recon(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int max_w_ctb, int max_h_ctb, int y_begin, int y_end) {
  static RbtReconCtbLds lds;
  if (y_end > max_h_ctb) y_end = max_h_ctb;
  for (int d = 2 * y_begin; d <= max_w_ctb - 1 + 2 * (y_end - 1); d++)
    for (int k = 0; k < n_frames; k++) {
      int fi = frame_list[k]; const RbtStreamCfg* g = &frames[fi].cfg;
      for (int y = y_begin; y < y_end && y < g->h_ctb; y++) {
        int x = d - 2 * y; if (x < 0 || x >= g->w_ctb) continue;
        int addr = y * g->w_ctb + x;
        if (frames[fi].ctb_slice[addr] == 0xFFFF) continue;
        rbt_recon_ctb<RC_ROLE_LUMA>(frames, slices, fi, addr, &lds.t, &lds.role[0]);
        rbt_recon_ctb<RC_ROLE_CHROMA>(frames, slices, fi, addr, &lds.t, &lds.role[1]);
      }
    }
}
void launch_deblock(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int) {
  for (int dir = 0; dir < 2; dir++)
    for (int k = 0; k < n_frames; k++) { RbtFrame* f = &frames[frame_list[k]]; for (int e = 0; e < rbt_deblock_edge_count(&f->cfg, dir); e++) rbt_deblock_unit(f, slices, rbt_deblock_edge_unit(&f->cfg, dir, e), dir); }
}
void launch_loopfilter(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int, int) {
  static RbtLoopLds lds;
  for (int k = 0; k < n_frames; k++) { RbtFrame* f = &frames[frame_list[k]]; const int nt = ((f->cfg.w + RBT_LF_TILE - 1) / RBT_LF_TILE) * ((f->cfg.h + RBT_LF_TILE - 1) / RBT_LF_TILE);
    for (int t = 0; t < nt; t++) rbt_loopfilter_tile(f, slices, t, &lds); }
}
void launch_sao_ctb(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int) {
  for (int k = 0; k < n_frames; k++) { RbtFrame* f = &frames[frame_list[k]]; for (int ctb = 0; ctb < f->cfg.w_ctb * f->cfg.h_ctb; ctb++) rbt_sao_ctb(f, slices, ctb); }
}
void launch_sao(RbtFrame* frames, const RbtSlice* slices, const int32_t* frame_list, int n_frames, int) {
  for (int k = 0; k < n_frames; k++) {
    RbtFrame* f = &frames[frame_list[k]];
    for (int c = 0; c < 3; c++) { int pw = c ? f->cfg.cw : f->cfg.w, ph = c ? f->cfg.ch : f->cfg.h; for (int i = 0; i < pw * ph; i++) rbt_sao_sample(f, slices, c, i % pw, i / pw); }
  }
}
#include "rbt_kernels_hostemu_enc.inc"
int selftest_transform32(const int16_t*, int, int, uint32_t* n_bad) { *n_bad = 0; return 0; }   // no matrix cores here
// verification stage: the same per-element routines, visited serially
void launch_pcc_occmap(const RbtPccParams* P, const uint16_t* occ, uint8_t* om) { for (int i = 0; i < P->w * P->h; i++) om[i] = occ[(size_t)(i / P->w / P->prec) * P->ow + (i % P->w) / P->prec] > P->threshold; }
void launch_pcc_owner(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, uint32_t* b2p) {
  for (int k = 0; k < n_items; k++) { const int pi = (int)(items[k] >> 16), blk = (int)(items[k] & 0xFFFF); const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0; int any = 0;
    for (int q = 0; q < P->res * P->res; q++) any |= pc_pixel_occupied_video(P, p, occ, ub, vb, q);
    if (any) { uint32_t* d = &b2p[pc_block2canvas(p, ub, vb, P->bw)]; if ((uint32_t)pi + 1 > *d) *d = (uint32_t)pi + 1; } }
}
void launch_pcc_count(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint32_t* b2p, uint32_t* counts) {
  for (int k = 0; k < n_items; k++) { const int pi = (int)(items[k] >> 16), blk = (int)(items[k] & 0xFFFF); const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0; uint32_t n = 0;
    if (b2p[pc_block2canvas(p, ub, vb, P->bw)] == (uint32_t)pi + 1) for (int q = 0; q < P->res * P->res; q++) n += (uint32_t)pc_pixel_points(P, p, occ, d0, d1, nullptr, nullptr, ub, vb, q, nullptr, nullptr);
    counts[k] = n; }
}
void launch_scan_u32(const uint32_t* in, uint32_t* out, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) { out[i] = r; r += in[i]; } out[n] = r; }
void launch_pcc_emit(const RbtPccParams* P, const rbt_patch* patches, const uint32_t* items, int n_items, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, const uint16_t* t0, const uint16_t* t1,
                     const uint32_t* b2p, const uint32_t* offsets, int16_t* xyz, uint16_t* yuv, const uint8_t* om, uint32_t* meta) {
  for (int k = 0; k < n_items; k++) { const int pi = (int)(items[k] >> 16), blk = (int)(items[k] & 0xFFFF); const rbt_patch* p = &patches[pi]; const int ub = blk % p->size_u0, vb = blk / p->size_u0;
    if (b2p[pc_block2canvas(p, ub, vb, P->bw)] != (uint32_t)pi + 1) continue;
    size_t o = offsets[k];
    for (int q = 0; q < P->res * P->res; q++) {
      const int n = pc_pixel_points(P, p, occ, d0, d1, t0, t1, ub, vb, q, xyz + 3 * o, yuv + 3 * o);
      if (meta && n) { int x, y; pc_patch2canvas(p, P->res, ub * P->res + q % P->res, vb * P->res + q / P->res, &x, &y);
        const uint32_t m = (uint32_t)pi | ((uint32_t)pc_boundary_point(om, x, y, P->w, P->h) << 31); for (int i = 0; i < n; i++) meta[o + i] = m; }
      o += (size_t)n; } }
}
void launch_sm_max(const int16_t* xyz, int n_points, uint32_t* out) { for (int i = 0; i < 3 * n_points; i++) if (xyz[i] > 0 && (uint32_t)xyz[i] > *out) *out = (uint32_t)xyz[i]; }
void launch_sm_passes(const RbtSmooth* G, int16_t* xyz, const uint32_t* meta) {
  for (int i = 0; i < G->n_points; i++) pc_sm_mark(G, xyz, meta, i);
  for (int i = 0; i < G->n_points; i++) pc_sm_accum(G, xyz, meta, i);
  for (int i = 0; i < G->n_points; i++) if (pc_sm_filter(G, xyz, meta, i)) (*G->moved)++;
}
void launch_vol_set(const int16_t* xyz, int n, uint32_t* vol, uint8_t* first, uint32_t* n_unique) {
  for (int i = 0; i < n; i++) { const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2]; uint32_t* w = &vol[pc_voxel_word(x, y, z)]; const uint32_t bit = 1u << (x & 31);
    first[i] = !(*w & bit); *w |= bit; if (first[i]) (*n_unique)++; }
}
void launch_d2_insert(const int16_t* xyz, int n, uint32_t* vol, uint32_t* keys, uint32_t* vals, int lg) {
  for (int i = 0; i < n; i++) { const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2]; vol[pc_voxel_word(x, y, z)] |= 1u << (x & 31); pc_hash_insert(keys, vals, lg, pc_voxel_id(x, y, z), (uint32_t)i); }
}
void launch_d2_give(const RbtD2Set* A, const int16_t* normals_a, const RbtD2Set* B, long long* acc_b, int32_t* cnt_b) { for (int i = 0; i < A->n; i++) pc_d2_give(A, normals_a, B, acc_b, cnt_b, i); }
void launch_d2_take(const RbtD2Set* B, const RbtD2Set* A, const int16_t* normals_a, long long* acc_b, int32_t* cnt_b) { for (int j = 0; j < B->n; j++) pc_d2_take(B, A, normals_a, acc_b, cnt_b, j); }
void launch_d2_dist(const RbtD2Set* P, const RbtD2Set* Q, const long long* acc_q, const int32_t* cnt_q, const int16_t* normals_q, double* out) {
  for (int i = 0; i < P->n; i++) { const double v = pc_d2_value(P, Q, acc_q, cnt_q, normals_q, i); if (v < 0) continue; out[0] += v; if (v > out[1]) out[1] = v; ((unsigned long long*)out)[2]++; }
}
void launch_vol_nn(const int16_t* xyz, const uint8_t* first, int n, const uint32_t* vol_other, unsigned long long* sse, uint32_t* max_d2) {
  for (int i = 0; i < n; i++) if (first[i]) { const uint32_t d = pc_nearest_d2(vol_other, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]); *sse += d; if (d > *max_d2) *max_d2 = d; }
}
}  // namespace rbtk

// ---- accessors of the PRODUCT's normative tables (csrc/rbt_tables.h) for tests/test_tables_product.py ----
extern "C" int rbt_hostemu_table(const char* name, int i, int j, int k) {
  if (!strcmp(name, "dct32")) return k_dct32[i][j];
  if (!strcmp(name, "dst4")) return k_dst4[i][j];
  if (!strcmp(name, "quant_scale")) return k_quant_scale[i];
  if (!strcmp(name, "dequant_scale")) return k_dequant_scale[i];
  if (!strcmp(name, "chroma_qp")) return rbt_chroma_qp(i);
  if (!strcmp(name, "range_lps")) return k_range_lps[i][j];
  if (!strcmp(name, "next_lps")) return k_next_lps[i];
  if (!strcmp(name, "ctx_count")) return RBT_CTX_COUNT;
  if (!strcmp(name, "ctx_init")) return k_ctx_init[i][j];
  if (!strcmp(name, "sig_ctx_4x4")) return k_sig_ctx_4x4[i];
  if (!strcmp(name, "intra_angle")) return k_intra_angle[i];
  if (!strcmp(name, "intra_inv_angle")) return k_intra_inv_angle[i];
  if (!strcmp(name, "luma_filter")) return k_luma_filter[i][j];
  if (!strcmp(name, "chroma_filter")) return k_chroma_filter[i][j];
  if (!strcmp(name, "beta")) return k_beta_table[i];
  if (!strcmp(name, "tc")) return k_tc_table[i];
  if (!strcmp(name, "scan")) return k_scan[i][j][k];     // [scan_idx][log2 block size in 4x4 units... see rbt_tables.h][pos] = x | y << 4
  if (!strcmp(name, "group_idx")) return k_group_idx[i];
  if (!strcmp(name, "min_in_group")) return k_min_in_group[i];
  return -99999;
}

// ---- the PRODUCT's host-side slice segment header parser (host/rbt_hls.cpp) on every slice of an Annex-B stream, 18 ints per slice, for
// tests/test_slice_headers.py (pinned against the reference's TDecCavlc::parseSliceHeader through tests/golden/slices_*.json). Returns the slice count.
extern "C" int rbt_hostemu_slice_headers(const uint8_t* annexb, size_t n, int* out, int cap) {
  std::vector<uint8_t> rbsp; std::vector<rbt::Nal> nals; rbt::split_annexb(annexb, n, rbsp, nals);
  rbt::ParamSets* ps = new rbt::ParamSets(); std::string err; int k = 0, prev_poc = 0;
  for (auto& nal : nals) {
    const uint8_t* r = rbsp.data() + nal.rbsp_off;
    if (nal.type == 33) rbt::parse_sps(*ps, r, nal.rbsp_size, err);
    else if (nal.type == 34) rbt::parse_pps(*ps, r, nal.rbsp_size, err);
    else if (nal.type < 32) {
      static thread_local rbt::SliceHdr head; static thread_local bool have_head = false; if (k == 0) have_head = false;
      rbt::SliceHdr h; if (rbt::parse_slice_header(*ps, r, nal.rbsp_size, nal.type, h, err, have_head ? &head : nullptr)) { delete ps; return -1; }
      if (h.dependent) { h.poc = head.poc; h.poc_lsb = head.poc_lsb; }
      { // PicOrderCntVal (8.3.1) by the function host/rbt_decode.cpp uses
        const rbt::Sps& sp = ps->sps[ps->pps[h.pps_id].sps_id];
        if (h.dependent) h.poc = head.poc; else h.poc = rbt::slice_poc(sp, nal.type, h.poc_lsb, prev_poc); }
      if (!h.dependent) { head = h; have_head = true; }
      if (k < cap) { int* o = out + 64 * k; const int intra = h.slice_type == RBT_SLICE_I; o[18] = h.dependent;
        o[0] = nal.type; o[1] = h.segment_addr; o[2] = h.slice_type; o[3] = h.poc; o[4] = h.temporal_mvp; o[5] = h.sao_luma; o[6] = h.sao_chroma; o[7] = intra ? 0 : h.num_ref_idx;
        o[8] = h.cabac_init_flag; o[9] = intra ? 0 : h.collocated_ref_idx; o[10] = intra ? 0 : h.max_merge_cand; o[11] = h.qp; o[12] = h.cb_qp_offset; o[13] = h.cr_qp_offset;
        o[14] = h.deblocking_disabled; o[15] = h.beta_offset_div2; o[16] = h.tc_offset_div2; o[17] = h.lf_across;
        const bool has_rps = nal.type != 19 && nal.type != 20; o[19] = has_rps ? h.rps.num : 0; for (int q = 0; q < 4; q++) { o[20 + 2 * q] = has_rps && q < h.rps.num ? h.rps.delta_poc[q] : 0; o[21 + 2 * q] = has_rps && q < h.rps.num ? h.rps.used[q] : 0; }
        { const int wp = !intra && h.wp_on; o[28] = wp; o[29] = wp ? h.wp_luma_denom : 0; o[30] = wp ? h.wp_chroma_denom : 0; o[31] = 0;
          for (int q = 0; q < 4; q++) { const int on = wp && q < h.num_ref_idx; int* e = o + 32 + 8 * q; e[0] = on ? h.wp_luma_flag[q] : 0; e[1] = on ? h.wp_chroma_flag[q] : 0;
            for (int c = 0; c < 3; c++) { e[2 + 2 * c] = on ? h.wp_w[q][c] : 0; e[3 + 2 * c] = on ? h.wp_o[q][c] : 0; } } } }
      k++;
    }
  }
  delete ps;
  return k;
}
