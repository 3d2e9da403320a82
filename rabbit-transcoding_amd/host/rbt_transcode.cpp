// Encode half of the hot path and the transcode pipeline.
// Replaces initEncoder / setEncoderOptions / encodeVideo / resize_frame2 of PCCTranscoder (PCCTranscoder.cpp:683-753,
// :825-904, :548-592, :594-646) and the decode -> pool -> encode loop of transcodeVideo (:428-510).
#include <algorithm>
#include <memory>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include "rbt_batch.h"
#include "rbt_transcode.h"

namespace rbt {

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct EncStreamDesc {
  int w, h, bd, n_frames, qp, i_qp_offset, gop, lossless, log2_ctb, rows, md5;   // w, h: display size (any even numbers)
  std::vector<const uint8_t*> hint_pm, hint_dm; int hint_w4 = 0, hint_h4 = 0;       // per frame: the decoded input picture's 4x4 maps (device), empty = no hints
  std::vector<const uint8_t*> occ4; int occ4_w = 0, occ4_h = 0;                      // per frame: occupancy of the 4x4 luma units (device, RbtFrame::occ4), empty = every sample counts
  int sao = 0;                           // SAO on (every stream that is not lossless, unless RBT_ENC_SAO=0)
  int tools_off = 0;                     // RBT_ET_* decision tools left out (rbt_stream_params.preset)
  std::vector<const uint16_t*> src[3];   // device planes per frame
  // Arena sharing (round 4): per frame, buffers of the DECODED input picture that are dead by the time this picture is encoded and have the encoder's geometry - its
  // coefficient levels (read by the reconstruction of that picture only) and its pre-SAO samples (read by that picture's own loop filters only): the encoder keeps its own
  // levels and reconstruction there instead of in memory of its own (9.8 MB per 1280x1280 picture, 1.26 GB per GOF). Empty / null = the encoder allocates.
  std::vector<uint16_t*> alias_pix; std::vector<int16_t*> alias_coef;
  int src_stride = 0, src_x0 = 0, src_y0 = 0;   // the planes are views: luma row stride (0 = w) and origin of the w x h region (luma samples)
};
// Pictures are coded at the display size rounded up to 8 (all-intra) or 16 (I,P pairs: 16x16 inter CUs) and the padding is
// signalled as the conformance window (7.4.3.2.1), like libx265 does for the reference (PCCTranscoder.cpp:706).
struct PadJob { const uint16_t* in; int stride, x0, y0, w, h; uint16_t* out; int dw, dh; };
struct EncodeBatch {
  std::vector<EncStreamDesc> desc;
  std::vector<Sps> sps; std::vector<Pps> pps;
  std::vector<RbtFrame> frames; std::vector<RbtSlice> slices; std::vector<int> frame_stream, frame_is_idr;
  std::vector<int> stream_first;
  void* arena = nullptr; size_t arena_size = 0;
  RbtFrame* d_frames = nullptr; RbtSlice* d_slices = nullptr; int32_t* d_lists = nullptr; uint8_t* d_out = nullptr; uint8_t* d_packed = nullptr; uint32_t* d_dst = nullptr;
  size_t out_total = 0;
  std::vector<PadJob> pad_jobs;                 // source pictures that have to be copied into padded planes before the encoder reads them
  std::vector<std::vector<uint16_t>> cs_keep;   // host staging of the ctb->slice maps, alive until the copies have completed
  uint8_t* d_zero = nullptr; size_t zero_bytes = 0; bool wpp = false;   // launch tickets (3 words) + row progress of the wavefront mode
  int main_stream = 0, aux_stream = -1;          // aux_stream >= 0: the intra part was enqueued there (its timers live there)
  std::vector<int32_t> lists_keep; size_t off_i = 0, off_ideb = 0, off_p = 0, off_sl = 0, off_sl_p = 0, off_isao = 0, off_psao = 0, off_pdeb = 0; int n_pdeb = 0, n_i = 0, n_ideb = 0, n_p = 0, n_sl_i = 0, n_sl_p = 0, n_isao = 0, n_psao = 0;   // index lists (encode_upload_lists)
  std::string err;
  ~EncodeBatch() { rbtk::dev_free(arena); }
};

// RBT-E1 codes SAO unless RBT_ENC_SAO=0 (development switch, read by the oracle the same way)
// ... and transform skip for the 4x4 luma blocks unless RBT_ENC_TS=0
static int e1_ts_on() { static int v = -1; if (v < 0) { const char* e = getenv("RBT_ENC_TS"); v = !e || atoi(e) != 0; } return v; }
static int e1_sao_on() { static int v = -1; if (v < 0) { const char* e = getenv("RBT_ENC_SAO"); v = !e || atoi(e) != 0; } return v; }
// ... and the decision tools of round 3 (RBT_ET_*): SATD block costs, closed-loop mode choice, level-dependent rounding, coded mode trial unless RBT_ENC_SATD / _REFINE / _RQ / _RDM = 0
static int e1_tools(int lossless) {
  static int v = -1;
  if (v < 0) { auto on = [](const char* n) { const char* e = getenv(n); return !e || atoi(e) != 0; }; v = (on("RBT_ENC_SATD") ? RBT_ET_SATD : 0) | (on("RBT_ENC_REFINE") ? RBT_ET_REFINE : 0) | (on("RBT_ENC_RQ") ? RBT_ET_RQ : 0) | (on("RBT_ENC_RDM") ? RBT_ET_RDM : 0); }
  return lossless ? v & ~(RBT_ET_RQ | RBT_ET_RDM) : v;
}
// Every picture takes the in-place deblocking launches before the SAO kernel; RBT_FUSED_ENC_LF=1 deblocks inside the SAO kernel instead (en_sao_ctb: the CTB and its halo
// in LDS; same samples, ~2 GB less HBM traffic per GOF). Off by default: the driver's 20-GOF run is 3 % slower with it, 5 % with the decoder's fused form as well
// (tools/lf_probe.sh: 763 / 741 / 725 point-cloud frames/s) - the path is not HBM-bound, and the LDS-free filter launches fill gaps the LDS-holding kernels leave.
static int e1_fused_lf() { static int v = -1; if (v < 0) { const char* e = getenv("RBT_FUSED_ENC_LF"); v = e && atoi(e) != 0; } return v; }
static int coded_size(int v, int gop) { int al = gop > 1 ? 16 : 8; return (v + al - 1) / al * al; }
static void make_param_sets(const EncStreamDesc& d, Sps& s, Pps& p) {
  s = Sps(); p = Pps();
  const int cw = coded_size(d.w, d.gop), ch = coded_size(d.h, d.gop);
  s.valid = true; s.width = cw; s.height = ch; s.conf_win[1] = (cw - d.w) / 2; s.conf_win[3] = (ch - d.h) / 2; s.bit_depth = d.bd; s.log2_max_poc_lsb = 8; s.max_dec_pic_buffering = 3;
  s.log2_ctb = d.log2_ctb ? d.log2_ctb : 5; s.log2_min_cb = 3; s.log2_diff_max_min_cb = s.log2_ctb - 3;
  s.log2_min_tb = 2; s.log2_max_tb = std::min(5, s.log2_ctb); s.log2_diff_max_min_tb = s.log2_max_tb - 2;
  s.num_st_rps = 1; s.sao = d.sao; s.max_th_depth_intra = 1;   // an intra CU is one transform unit or four (oracle/hevc_enc.c setup_stream)
  s.w_ctb = (cw + (1 << s.log2_ctb) - 1) >> s.log2_ctb; s.h_ctb = (ch + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
  p.valid = true; p.num_ref_idx_default = 1; p.init_qp = std::min(51, std::max(0, d.qp)); p.loop_filter_across_slices = 1;
  p.transform_skip = !d.lossless && e1_ts_on();   // the 4x4 luma blocks are coded with or without the transform, whichever is cheaper (oracle/hevc_enc.c hm_tb_finish)
  if (d.lossless) { p.transquant_bypass = 1; p.deblocking_control_present = 1; p.pps_deblocking_disabled = 1; p.loop_filter_across_slices = 0; }
  if (d.rows < 0) { p.entropy_coding_sync = 1; p.dependent_slice_segments = 1; }   // wavefront rows, one dependent slice segment each (oracle/hevc_enc.c setup_stream)
}

static int encode_build(EncodeBatch& b) {
  size_t ns = b.desc.size();
  b.sps.resize(ns); b.pps.resize(ns); b.stream_first.resize(ns);
  for (size_t si = 0; si < ns; si++) {
    const EncStreamDesc& d = b.desc[si];
    if (d.w % 2 || d.h % 2 || d.w <= 0 || d.h <= 0 || d.w > 8192 || d.h > 8192) { b.err = "picture size must be even and at most 8192"; return RBT_ERR_UNSUPPORTED; }
    if (d.log2_ctb && (d.log2_ctb < 4 || d.log2_ctb > 6)) { b.err = "log2_ctb must be 4..6"; return RBT_ERR_PARAM; }
    if ((d.rows < 0) != (b.desc[0].rows < 0)) { b.err = "streams of one call must agree on the wavefront mode (ctb_rows_per_slice < 0)"; return RBT_ERR_PARAM; }
    b.desc[si].sao = !d.lossless && e1_sao_on();
    make_param_sets(b.desc[si], b.sps[si], b.pps[si]);
    const Sps& s = b.sps[si]; const Pps& p = b.pps[si];
    b.stream_first[si] = (int)b.frames.size();
    for (int i = 0; i < d.n_frames; i++) {
      bool is_i = d.gop <= 1 || (i % d.gop) == 0;
      RbtFrame f; memset(&f, 0, sizeof(f));
      fill_stream_cfg(s, p, f.cfg);
      f.poc = is_i ? 0 : (i % d.gop); f.level = is_i ? 0 : 1; f.first_slice = (int)b.slices.size();
      f.w8 = s.width / 8; f.h8 = s.height / 8; f.lossless = d.lossless; f.enc_tools = e1_tools(d.lossless) & ~d.tools_off; f.ref_frame = is_i ? -1 : (int)b.frames.size() - 1; f.ref_poc = is_i ? 0 : f.poc - 1;
      for (int c = 0; c < 3; c++) f.src[c] = d.src[c][i];
      if (!d.hint_dm.empty()) { f.hint_pm = d.hint_pm[i]; f.hint_dm = d.hint_dm[i]; f.hint_w4 = d.hint_w4; f.hint_h4 = d.hint_h4; }
      if (!d.occ4.empty() && !d.lossless) { f.occ4 = d.occ4[i]; f.occ4_w = d.occ4_w; f.occ4_h = d.occ4_h; }
      int n_ctb = s.w_ctb * s.h_ctb, step = d.rows > 0 ? d.rows * s.w_ctb : (d.rows < 0 ? s.w_ctb : n_ctb);
      for (int addr = 0; addr < n_ctb; addr += step) {
        RbtSlice sl; memset(&sl, 0, sizeof(sl));
        sl.next_seg = -1; sl.wpp = (uint8_t)(d.rows < 0); sl.dependent = (uint8_t)(d.rows < 0 && addr > 0);
        sl.frame = (int)b.frames.size(); sl.ctb_addr = addr; sl.n_ctbs = std::min(step, n_ctb - addr);
        sl.slice_type = (int8_t)(is_i ? RBT_SLICE_I : RBT_SLICE_P);
        sl.qp = (int8_t)std::min(51, std::max(0, is_i ? d.qp + d.i_qp_offset : d.qp));
        sl.deblocking_disabled = (uint8_t)p.pps_deblocking_disabled; sl.lf_across = (uint8_t)p.loop_filter_across_slices;
        sl.max_merge_cand = 1; sl.num_ref_idx = 1; sl.poc = f.poc; sl.sao_luma = sl.sao_chroma = (uint8_t)b.desc[si].sao;
        if (!is_i) { sl.ref_frame[0] = f.ref_frame; sl.ref_poc[0] = f.ref_poc; }
        // slice data buffer, by the slice's QP, sized for full-range noise (the most a picture can cost: log2(2^bit_depth / Qstep) bits per sample, 1.5 samples per luma
        // sample in 4:2:0): 3 bytes per luma sample of the slice (the raw samples at 2 bytes each) below QP 16 and for lossless streams, 2 below QP 34 (Qstep >= 4: at most
        // 8 bits x 1.5), 1 from there on (Qstep >= 32: 5 bits x 1.5), + slack. A slice that does not fit ends the call with an error (RbtSlice::out_size = 0xFFFFFFFF), it
        // is never cut short. Round 3 gave every slice 3: 7.6 of the 17 MB an encoded 1280x1280 picture took.
        size_t rows = (size_t)(sl.n_ctbs + s.w_ctb - 1) / s.w_ctb;
        const int bytes_per_sample = (d.lossless || sl.qp < 16) ? 3 : (sl.qp < 34 ? 2 : 1);
        sl.out_cap = (uint32_t)(rows * ((size_t)s.width << s.log2_ctb) * bytes_per_sample + 4096);
        f.n_slices++;
        b.slices.push_back(sl);
      }
      b.frames.push_back(f); b.frame_stream.push_back((int)si); b.frame_is_idr.push_back(is_i);
    }
  }
  if (b.slices.size() >= 0xFFFF) { b.err = "too many slice segments in one call"; return RBT_ERR_UNSUPPORTED; }
  // ---- HBM layout ----
  Arena a; size_t nf = b.frames.size();
  // wavefront bookkeeping: three launch tickets, then per picture 2 * h_ctb progress words and two next-row counters (zeroed before the first kernel of every job)
  bool any_wpp = false; for (auto& d : b.desc) any_wpp |= d.rows < 0;
  std::vector<size_t> o_rowdone(nf, 0), o_rowctx(nf, 0);
  size_t o_zero = a.reserve(64), zero_bytes = 64;
  if (any_wpp) for (size_t i = 0; i < nf; i++) { size_t n = ((size_t)b.frames[i].cfg.h_ctb * 2 + 2) * sizeof(uint32_t); o_rowdone[i] = a.reserve(n); zero_bytes = o_rowdone[i] + n - o_zero; }
  if (any_wpp) for (size_t i = 0; i < nf; i++) o_rowctx[i] = a.reserve((size_t)b.frames[i].cfg.h_ctb * 256);
  std::vector<size_t> o_src(nf, (size_t)-1), o_pix(nf), o_sout(nf), o_sao(nf), o_coef(nf), o_pm(nf), o_edges(nf), o_qp(nf), o_mv(nf), o_ref(nf), o_refpoc(nf), o_cs(nf), o_cul(nf), o_cum(nf), o_cuf(nf), o_cut(nf);
  for (size_t i = 0; i < nf; i++) {
    const RbtStreamCfg& c = b.frames[i].cfg; size_t u = (size_t)c.w4 * c.h4, nc = (size_t)c.w_ctb * c.h_ctb, u8 = (size_t)b.frames[i].w8 * b.frames[i].h8;
    { const EncStreamDesc& d = b.desc[b.frame_stream[i]];
      if (c.w != d.w || c.h != d.h || d.src_x0 || d.src_y0 || (d.src_stride && d.src_stride != d.w)) o_src[i] = a.reserve(frame_samples(c) * 2); }
    { const EncStreamDesc& d = b.desc[b.frame_stream[i]]; const size_t k = i - (size_t)b.stream_first[b.frame_stream[i]];
      o_pix[i] = (!d.alias_pix.empty() && d.alias_pix[k]) ? (size_t)-1 : a.reserve(frame_samples(c) * 2);
      o_coef[i] = (!d.alias_coef.empty() && d.alias_coef[k]) ? (size_t)-1 : a.reserve(frame_samples(c) * 2); }
    o_sout[i] = b.desc[b.frame_stream[i]].sao ? a.reserve(frame_samples(c) * 2) : o_pix[i]; o_sao[i] = a.reserve(nc * sizeof(RbtSao));      // (no SAO: out = pix, wherever that is)
    o_pm[i] = a.reserve(u); o_edges[i] = a.reserve(u); o_qp[i] = a.reserve(u); o_mv[i] = a.reserve(u * 4); o_ref[i] = a.reserve(u); o_refpoc[i] = a.reserve(u * 4);
    o_cul[i] = a.reserve(u8); o_cum[i] = a.reserve(u8); o_cuf[i] = a.reserve(u8); o_cut[i] = a.reserve(u8);
  }
  // the CTB -> slice maps of all pictures sit back to back: one upload per batch instead of one per picture (a copy is a queue entry of its own)
  size_t cs_words = 0; for (size_t i = 0; i < nf; i++) { o_cs[i] = cs_words; cs_words += (size_t)b.frames[i].cfg.w_ctb * b.frames[i].cfg.h_ctb; }
  const size_t o_cs_all = a.reserve(cs_words * 2);
  size_t o_frames = a.reserve(nf * sizeof(RbtFrame)), o_slices = a.reserve(b.slices.size() * sizeof(RbtSlice));
  size_t o_lists = a.reserve((nf + b.slices.size()) * 3 * sizeof(int32_t)), o_dst = a.reserve(b.slices.size() * sizeof(uint32_t));
  size_t out_cap_total = 0; for (auto& sl : b.slices) { sl.out_off = (uint32_t)out_cap_total; out_cap_total += sl.out_cap; }
  if (out_cap_total >= 0xFFFFFFFFull) { b.err = "output buffer too large for one call"; return RBT_ERR_UNSUPPORTED; }
  size_t o_out = a.reserve(out_cap_total), o_packed = a.reserve(out_cap_total / 2 + 65536);
  b.arena_size = a.reserve(0);
  b.arena = rbtk::dev_alloc(b.arena_size);
  if (!b.arena) { b.err = "device allocation failed"; return RBT_ERR_NOMEM; }
  uint8_t* base = (uint8_t*)b.arena;
  b.cs_keep.assign(1, std::vector<uint16_t>(cs_words, 0));
  for (size_t i = 0; i < nf; i++) {
    uint16_t* cs_host = b.cs_keep[0].data() + o_cs[i];
    RbtFrame& f = b.frames[i]; const RbtStreamCfg& c = f.cfg; size_t ys = (size_t)c.w * c.h, cs = (size_t)c.cw * c.ch;
    const EncStreamDesc& ds = b.desc[b.frame_stream[i]]; const size_t ks = i - (size_t)b.stream_first[b.frame_stream[i]];
    f.pix[0] = o_pix[i] == (size_t)-1 ? ds.alias_pix[ks] : (uint16_t*)(base + o_pix[i]); f.pix[1] = f.pix[0] + ys; f.pix[2] = f.pix[1] + cs;
    f.out[0] = o_sout[i] == (size_t)-1 ? f.pix[0] : (uint16_t*)(base + o_sout[i]); f.out[1] = f.out[0] + ys; f.out[2] = f.out[1] + cs; f.sao = (RbtSao*)(base + o_sao[i]);
    if (o_src[i] != (size_t)-1) {
      const EncStreamDesc& d = b.desc[b.frame_stream[i]]; const int st = d.src_stride ? d.src_stride : d.w;
      uint16_t* pl[3] = {(uint16_t*)(base + o_src[i]), nullptr, nullptr}; pl[1] = pl[0] + ys; pl[2] = pl[1] + cs;
      for (int k = 0; k < 3; k++) { const int sh = k ? 1 : 0;
        b.pad_jobs.push_back(PadJob{f.src[k], st >> sh, d.src_x0 >> sh, d.src_y0 >> sh, d.w >> sh, d.h >> sh, pl[k], c.w >> sh, c.h >> sh}); f.src[k] = pl[k]; }
    }
    f.coef[0] = o_coef[i] == (size_t)-1 ? ds.alias_coef[ks] : (int16_t*)(base + o_coef[i]); f.coef[1] = f.coef[0] + ys; f.coef[2] = f.coef[1] + cs;
    f.pm = base + o_pm[i]; f.edges = base + o_edges[i]; f.qp = (int8_t*)(base + o_qp[i]); f.mv = (int16_t*)(base + o_mv[i]); f.ref = (int8_t*)(base + o_ref[i]);
    f.refpoc = (int32_t*)(base + o_refpoc[i]); f.ctb_slice = (uint16_t*)(base + o_cs_all) + o_cs[i];
    f.cu_log2 = base + o_cul[i]; f.cu_mode = base + o_cum[i]; f.cu_flags = base + o_cuf[i]; f.cu_ts = base + o_cut[i];
    if (any_wpp) { f.row_done = (uint32_t*)(base + o_rowdone[i]); f.row_ctx = base + o_rowctx[i]; }
    // per CTB: the SLICE it belongs to (index of the slice's independent segment): availability, QP and loop filter flags are per slice
    { int head = f.first_slice;
      for (int k = 0; k < f.n_slices; k++) { const RbtSlice& sl = b.slices[f.first_slice + k]; if (!sl.dependent) head = f.first_slice + k; for (int q = 0; q < sl.n_ctbs; q++) cs_host[sl.ctb_addr + q] = (uint16_t)head; } }
  }
  if (rbtk::h2d(base + o_cs_all, b.cs_keep[0].data(), cs_words * 2)) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  b.d_frames = (RbtFrame*)(base + o_frames); b.d_slices = (RbtSlice*)(base + o_slices); b.d_lists = (int32_t*)(base + o_lists); b.d_dst = (uint32_t*)(base + o_dst);
  b.d_out = base + o_out; b.d_packed = base + o_packed; b.out_total = out_cap_total;
  b.d_zero = base + o_zero; b.zero_bytes = zero_bytes; b.wpp = any_wpp;
  if (rbtk::h2d(b.d_frames, b.frames.data(), nf * sizeof(RbtFrame)) || rbtk::h2d(b.d_slices, b.slices.data(), b.slices.size() * sizeof(RbtSlice))) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  return 0;
}

// runs the kernels and packs one Annex-B stream per input stream
// The encoder of one stream is enqueued in three steps so that nothing on the host waits for the GPU in between:
// encode_upload_lists (host -> device copies, issued while the stream is still empty), encode_launch (kernels up to the
// entropy coder, may be enqueued behind the decoder's kernels on the same stream), encode_finish (one sync, packing, NALs).
static int encode_upload_lists(EncodeBatch& b) {
  size_t nf = b.frames.size(), ns = b.slices.size();
  std::vector<int32_t>& lists = b.lists_keep; lists.clear(); b.n_i = b.n_ideb = b.n_p = 0;
  b.off_i = lists.size(); for (size_t i = 0; i < nf; i++) if (b.frame_is_idr[i]) { lists.push_back((int)i); b.n_i++; }
  // with RBT_FUSED_ENC_LF=1 pictures with SAO are deblocked inside the SAO kernel (en_sao_ctb: the CTB and a halo in LDS) and only pictures without it (RBT_ENC_SAO=0) take the in-place deblocking launches
  b.off_ideb = lists.size(); for (size_t i = 0; i < nf; i++) if (b.frame_is_idr[i] && !b.frames[i].lossless && (!b.desc[b.frame_stream[i]].sao || !e1_fused_lf())) { lists.push_back((int)i); b.n_ideb++; }
  b.n_pdeb = 0; b.off_pdeb = lists.size(); for (size_t i = 0; i < nf; i++) if (!b.frame_is_idr[i] && !b.frames[i].lossless && (!b.desc[b.frame_stream[i]].sao || !e1_fused_lf())) { lists.push_back((int)i); b.n_pdeb++; }
  b.off_p = lists.size(); for (size_t i = 0; i < nf; i++) if (!b.frame_is_idr[i]) { lists.push_back((int)i); b.n_p++; }
  b.n_isao = b.n_psao = 0;
  b.off_isao = lists.size(); for (size_t i = 0; i < nf; i++) if (b.frame_is_idr[i] && b.desc[b.frame_stream[i]].sao) { lists.push_back((int)i); b.n_isao++; }
  b.off_psao = lists.size(); for (size_t i = 0; i < nf; i++) if (!b.frame_is_idr[i] && b.desc[b.frame_stream[i]].sao) { lists.push_back((int)i); b.n_psao++; }
  // slice segments of the intra pictures first, then the rest: the two groups are entropy-coded by separate launches
  b.n_sl_i = b.n_sl_p = 0;
  b.off_sl = lists.size(); for (size_t i = 0; i < ns; i++) if (b.frame_is_idr[b.slices[i].frame]) { lists.push_back((int)i); b.n_sl_i++; }
  b.off_sl_p = lists.size(); for (size_t i = 0; i < ns; i++) if (!b.frame_is_idr[b.slices[i].frame]) { lists.push_back((int)i); b.n_sl_p++; }
  if (rbtk::h2d(b.d_lists, lists.data(), lists.size() * sizeof(int32_t))) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  return 0;
}
// intra pictures (analysis, closed-loop intra coding, deblocking): they only read their own source pictures
static void encode_launch_intra(EncodeBatch& b) {
  size_t nf = b.frames.size();
  for (const PadJob& j : b.pad_jobs) rbtk::launch_pad(j.in, j.stride, j.x0, j.y0, j.w, j.h, j.out, j.dw, j.dh);
  int mw = 0, mh = 0, mu = 0, mc = 0, row_mode = 1, ml2 = 0, ml = 0;
  for (size_t i = 0; i < nf; i++) {
    const RbtStreamCfg& c = b.frames[i].cfg; ml = std::max(ml, c.w * c.h); ml2 = std::max(ml2, (int)c.log2_ctb); mw = std::max(mw, c.w_ctb); mh = std::max(mh, c.h_ctb); mu = std::max(mu, c.w4 * c.h4); mc = std::max(mc, c.w_ctb * c.h_ctb);
    if (b.desc[b.frame_stream[i]].rows != 1) row_mode = 0;
  }
  if (b.wpp) { row_mode = 2; rbtk::dev_memset(b.d_zero, 0, b.zero_bytes); }   // every stream of the batch is in wavefront mode (encode_build checks)
  rbtk::timer_begin(T_ANALYSE);
  rbtk::launch_enc_analyse(b.d_frames, b.d_slices, b.d_lists + b.off_i, b.n_i, mc);
  rbtk::timer_end(T_ANALYSE);
  rbtk::timer_begin(T_ENCODE);
  rbtk::launch_enc_intra(b.d_frames, b.d_slices, b.d_lists + b.off_i, b.n_i, mw, mh, row_mode, ml2, (uint32_t*)b.d_zero);
  rbtk::launch_deblock(b.d_frames, b.d_slices, b.d_lists + b.off_ideb, b.n_ideb, mu);
  rbtk::launch_enc_sao(b.d_frames, b.d_slices, b.d_lists + b.off_isao, b.n_isao, mc, ml2, e1_fused_lf());   // decides and applies
  rbtk::timer_end(T_ENCODE);
}
// entropy coding of the intra pictures' slices: needs nothing but their levels and CU data
static int max_log2_ctb(const EncodeBatch& b) { int m = 0; for (auto& f : b.frames) m = std::max(m, (int)f.cfg.log2_ctb); return m; }
static int max_w_ctb(const EncodeBatch& b) { int m = 0; for (auto& f : b.frames) m = std::max(m, (int)f.cfg.w_ctb); return m; }
static int max_h_ctb(const EncodeBatch& b) { int m = 0; for (auto& f : b.frames) m = std::max(m, (int)f.cfg.h_ctb); return m; }
static void encode_launch_entropy_intra(EncodeBatch& b) {
  rbtk::timer_begin(T_ENTROPY_I);
  if (b.wpp) rbtk::launch_entropy_wave(b.d_frames, b.d_slices, b.d_out, b.d_lists + b.off_i, b.n_i, max_w_ctb(b), max_h_ctb(b), max_log2_ctb(b), (uint32_t*)b.d_zero + 1);
  else rbtk::launch_entropy(b.d_frames, b.d_slices, b.d_out, b.d_lists + b.off_sl, b.n_sl_i, max_log2_ctb(b));
  rbtk::timer_end(T_ENTROPY_I);
}
// inter pictures (need the reconstructed intra pictures and their own sources), then the entropy coder for every slice
static void encode_launch_rest(EncodeBatch& b) {
  size_t nf = b.frames.size();
  int mu = 0, mc = 0, ml = 0;
  for (size_t i = 0; i < nf; i++) { const RbtStreamCfg& c = b.frames[i].cfg; mu = std::max(mu, c.w4 * c.h4); mc = std::max(mc, c.w_ctb * c.h_ctb); ml = std::max(ml, c.w * c.h); }
  rbtk::timer_begin(T_INTER);
  rbtk::launch_enc_inter(b.d_frames, b.d_slices, b.d_lists + b.off_p, b.n_p, mc);
  rbtk::launch_deblock(b.d_frames, b.d_slices, b.d_lists + b.off_pdeb, b.n_pdeb, mu);
  rbtk::launch_enc_sao(b.d_frames, b.d_slices, b.d_lists + b.off_psao, b.n_psao, mc, max_log2_ctb(b), e1_fused_lf());
  rbtk::timer_end(T_INTER);
  rbtk::timer_begin(T_ENTROPY);
  if (b.wpp) rbtk::launch_entropy_wave(b.d_frames, b.d_slices, b.d_out, b.d_lists + b.off_p, b.n_p, max_w_ctb(b), max_h_ctb(b), max_log2_ctb(b), (uint32_t*)b.d_zero + 2);
  else rbtk::launch_entropy(b.d_frames, b.d_slices, b.d_out, b.d_lists + b.off_sl_p, b.n_sl_p, max_log2_ctb(b));
  rbtk::timer_end(T_ENTROPY);
}
static int encode_launch(EncodeBatch& b) { encode_launch_intra(b); encode_launch_entropy_intra(b); encode_launch_rest(b); return 0; }
static int encode_finish(EncodeBatch& b, std::vector<std::vector<uint8_t>>& outs, rbt_stats& st) {
  size_t nf = b.frames.size(), ns = b.slices.size();
  if (rbtk::d2h(b.slices.data(), b.d_slices, ns * sizeof(RbtSlice))) { b.err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  std::vector<uint32_t> dst(ns); size_t total = 0;
  for (size_t i = 0; i < ns; i++) { if (b.slices[i].out_size > b.slices[i].out_cap) { b.err = "slice data exceeds its buffer"; return RBT_ERR_NOMEM; } dst[i] = (uint32_t)total; total += b.slices[i].out_size; }
  if (total > b.out_total / 2 + 65536) { b.err = "packed output exceeds its buffer"; return RBT_ERR_NOMEM; }
  std::vector<uint8_t> packed(total);
  if (rbtk::h2d(b.d_dst, dst.data(), ns * sizeof(uint32_t))) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  rbtk::launch_pack(b.d_out, b.d_slices, b.d_dst, b.d_packed, (int)ns);
  double t0 = now_ms();
  if (total && rbtk::d2h(packed.data(), b.d_packed, total)) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  if (rbtk::dev_sync()) { b.err = "kernel execution failed"; return RBT_ERR_NO_DEVICE; }
  st.d2h_ms += now_ms() - t0;
  // ---- NAL packing (parameter sets, slice headers, emulation prevention) ----
  double t1 = now_ms();
  outs.assign(b.desc.size(), {});
  // decoded picture hash SEI (md5_sei): the reconstructed pictures come to the host once and their planes are hashed side by side (md5_planes_u16)
  std::vector<uint16_t> rec; std::vector<size_t> rec_off(nf, 0); std::vector<uint8_t> hashes(nf * 48);
  { size_t tot = 0; for (size_t i = 0; i < nf; i++) if (b.desc[b.frame_stream[i]].md5) { rec_off[i] = tot; tot += frame_samples(b.frames[i].cfg); }
    if (tot) {
      rec.resize(tot); std::vector<Md5PlaneJob> jobs;
      for (size_t i = 0; i < nf; i++) if (b.desc[b.frame_stream[i]].md5) {
        const RbtFrame& f = b.frames[i]; const RbtStreamCfg& c = f.cfg; uint16_t* r = rec.data() + rec_off[i];
        if (rbtk::d2h(r, f.out[0], frame_samples(c) * 2)) { b.err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
        jobs.push_back({r, c.w, c.h, c.bit_depth, &hashes[i * 48]});
        jobs.push_back({r + (size_t)c.w * c.h, c.cw, c.ch, c.bit_depth, &hashes[i * 48 + 16]});
        jobs.push_back({r + (size_t)c.w * c.h + (size_t)c.cw * c.ch, c.cw, c.ch, c.bit_depth, &hashes[i * 48 + 32]});
      }
      md5_planes_u16(jobs.data(), jobs.size());
    } }
  for (size_t i = 0; i < nf; i++) {
    int si = b.frame_stream[i]; const Sps& s = b.sps[si]; const Pps& p = b.pps[si]; std::vector<uint8_t>& out = outs[si]; const RbtFrame& f = b.frames[i];
    bool idr = b.frame_is_idr[i] != 0;
    if (idr) write_param_sets(out, s, p);
    for (int k = 0; k < f.n_slices; k++) {
      const RbtSlice& sl = b.slices[f.first_slice + k];
      SliceHdr h; h.first_slice_in_pic = k == 0; h.dependent = sl.dependent; h.segment_addr = sl.ctb_addr; h.slice_type = sl.slice_type; h.poc = sl.poc; h.num_ref_idx = 1; h.max_merge_cand = 1; h.qp = sl.qp;
      h.sao_luma = sl.sao_luma; h.sao_chroma = sl.sao_chroma; h.deblocking_disabled = sl.deblocking_disabled; h.beta_offset_div2 = p.beta_offset_div2; h.tc_offset_div2 = p.tc_offset_div2; h.lf_across = sl.lf_across;
      BitWriter w; write_slice_header(w, s, p, h, idr, 0);
      w.b.insert(w.b.end(), packed.begin() + dst[f.first_slice + k], packed.begin() + dst[f.first_slice + k] + sl.out_size);
      append_nal(out, idr ? NAL_IDR_W_RADL : NAL_TRAIL_R, w.b.data(), w.b.size(), k == 0);
    }
    if (b.desc[si].md5) {
      uint8_t sei[52]; sei[0] = 132; sei[1] = 49; sei[2] = 0;
      memcpy(sei + 3, &hashes[i * 48], 48);
      sei[51] = 0x80;
      append_nal(out, NAL_SEI_SUFFIX, sei, 52, false);
    }
  }
  st.host_pack_ms += now_ms() - t1;
  { int cur = b.main_stream; if (b.aux_stream >= 0) rbtk::set_stream(b.aux_stream);
    st.k_analyse_ms += rbtk::timer_ms(T_ANALYSE); st.k_encode_ms += rbtk::timer_ms(T_ENCODE); st.k_entropy_ms += rbtk::timer_ms(T_ENTROPY_I);
    if (b.aux_stream >= 0) rbtk::set_stream(cur);
    st.k_encode_ms += rbtk::timer_ms(T_INTER); st.k_entropy_ms += rbtk::timer_ms(T_ENTROPY); }
  return 0;
}

static int encode_run(EncodeBatch& b, std::vector<std::vector<uint8_t>>& outs, rbt_stats& st) {
  int rc = encode_upload_lists(b);
  if (!rc) rc = encode_launch(b);
  return rc ? rc : encode_finish(b, outs, st);
}

static int hand_out(const std::vector<std::vector<uint8_t>>& outs, uint8_t** out, size_t* n_out) {
  for (size_t i = 0; i < outs.size(); i++) {
    out[i] = (uint8_t*)malloc(outs[i].size() ? outs[i].size() : 1);
    if (!out[i]) { for (size_t k = 0; k < i; k++) { free(out[k]); out[k] = nullptr; } return RBT_ERR_NOMEM; }
    memcpy(out[i], outs[i].data(), outs[i].size()); n_out[i] = outs[i].size();
  }
  return 0;
}

// Pool + encoder setup for the one stream of `db` (PCCTranscoder.cpp:466, :825-904).
// Row-band parsing of the longest pipeline (resumable parser + reconstruction of finished bands underneath the rest of the
// parse). Measured on the 32-frame GOF: every band ends with the slowest slice OF THAT BAND, and the sum of those maxima
// exceeds the single maximum by more than the hidden reconstruction saves (4 bands: 311.9 ms, 2 bands: 302.6 ms, off:
// 297.3 ms). Off by default; RBT_PARSE_BANDS=<n> in the environment turns it on for experiments.
static int parse_bands() { static int v = -1; if (v < 0) { const char* e = getenv("RBT_PARSE_BANDS"); v = e ? atoi(e) : 1; if (v < 1) v = 1; if (v > 16) v = 16; } return v; }
struct PoolJob { const uint16_t* in; int stride, w, h; uint16_t *y, *cb, *cr; int grey; };
// occupancy-aware coding: the occupancy maps per 4x4 luma unit (RbtFrame::occ4) that the geometry / attribute streams of a GOF are coded with, made from the pooled
// occupancy pictures of that GOF's occupancy stream by one launch on the occupancy pipeline's stream
struct OccSource { const uint16_t* occ = nullptr; size_t in_step = 0; int n = 0, ow = 0, oh = 0, pipeline = -1; };   // the pooled luma planes of one occupancy stream
struct OccJob { int source; uint8_t* maps; int W, w4, h4; };                                                           // one launch_occ_units: the maps of one consumer stream
// pool_jobs != nullptr: the OR-pool launches are recorded instead of issued (the decoder's kernels are not enqueued yet)
// stream `si` of the decode batch becomes stream `ei` of the encode batch (several target rate points may re-encode one decoded
// stream: BASELINE.json configs[4], rate fan-out)
static int setup_encode(DecodeBatch& db, int si, int ei, const rbt_stream_params& p, EncodeBatch& eb, std::vector<void*>& pooled, std::string& err, std::vector<PoolJob>* pool_jobs = nullptr, const OccSource* occ = nullptr, int occ_index = -1, std::vector<OccJob>* occ_jobs = nullptr) {
  if ((int)eb.desc.size() <= ei) eb.desc.resize((size_t)ei + 1);
  EncStreamDesc& d = eb.desc[ei]; int first = db.stream_first[si], cnt = db.stream_count[si];
  const RbtStreamCfg& c = db.frames[first].cfg; const Sps& isps = db.stream_sps[si];
  // what a player shows of the input: the coded picture minus its conformance window
  const int cl = 2 * isps.conf_win[0], ct = 2 * isps.conf_win[2], dw = c.w - cl - 2 * isps.conf_win[1], dh = c.h - ct - 2 * isps.conf_win[3];
  if (dw <= 0 || dh <= 0) { err = "empty conformance window"; return RBT_ERR_BITSTREAM; }
  if (p.preset != RBT_PRESET_DEFAULT && p.preset != RBT_PRESET_FAST) { err = "preset must be RBT_PRESET_DEFAULT or RBT_PRESET_FAST"; return RBT_ERR_PARAM; }
  d.bd = c.bit_depth; d.n_frames = cnt; d.qp = p.qp; d.log2_ctb = p.log2_ctb; d.rows = p.ctb_rows_per_slice; d.md5 = p.md5_sei; d.tools_off = p.preset == RBT_PRESET_FAST ? (RBT_ET_SATD | RBT_ET_REFINE | RBT_ET_RQ | RBT_ET_RDM) : 0;
  for (int k = 0; k < 3; k++) d.src[k].resize(cnt);
  auto view = [&](int k, int q) { return (const uint16_t*)db.frames[first + k].out[q]; };
  if (p.video_type == RBT_VIDEO_OCCUPANCY) {
    int factor = p.occupancy_precision / 2; if (factor < 1) factor = 1;
    d.gop = 1; d.lossless = 1; d.i_qp_offset = 0; d.w = dw / factor; d.h = dh / factor;
    if (p.occupancy_precision == 4) {
      if (dw % 4 || dh % 4) { err = "occupancy map size must be a multiple of 4 to pool"; return RBT_ERR_UNSUPPORTED; }
      size_t ys = (size_t)d.w * d.h, cs = (size_t)(d.w / 2) * (d.h / 2);
      uint16_t* buf = (uint16_t*)rbtk::dev_alloc((ys + 2 * cs) * 2 * (size_t)cnt);
      if (!buf) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
      pooled.push_back(buf);
      if (!pool_jobs) rbtk::timer_begin(T_POOL);
      for (int k = 0; k < cnt; k++) {
        uint16_t* y = buf + (ys + 2 * cs) * (size_t)k;
        const uint16_t* in = view(k, 0) + (size_t)ct * c.w + cl;
        // the reference leaves the pooled chroma planes unwritten (PCCTranscoder.cpp:638-641); mid-grey here
        if (pool_jobs) pool_jobs->push_back(PoolJob{in, c.w, dw, dh, y, y + ys, y + ys + cs, 1 << (c.bit_depth - 1)});
        else rbtk::launch_pool(in, c.w, dw, dh, 2, y, y + ys, y + ys + cs, 1 << (c.bit_depth - 1));
        d.src[0][k] = y; d.src[1][k] = y + ys; d.src[2][k] = y + ys + cs;
      }
      if (!pool_jobs) rbtk::timer_end(T_POOL);
    } else { d.src_stride = c.w; d.src_x0 = cl; d.src_y0 = ct; for (int k = 0; k < cnt; k++) for (int q = 0; q < 3; q++) d.src[q][k] = view(k, q); }
  } else {
    d.gop = 2; d.lossless = 0; d.i_qp_offset = -3; d.w = dw; d.h = dh;
    d.src_stride = c.w; d.src_x0 = cl; d.src_y0 = ct;
    for (int k = 0; k < cnt; k++) for (int q = 0; q < 3; q++) d.src[q][k] = view(k, q);
    // arena sharing: the encoder's levels and reconstruction of picture k live in the decoded picture k's dead buffers when the two pictures have one geometry (coded size
    // = the input's coded size, no window offset) and no other target rate of a fan-out took them already. RBT_ARENA_SHARE=0 switches it off.
    { static const int share = [] { const char* e = getenv("RBT_ARENA_SHARE"); return !e || atoi(e) != 0; }();
      if (db.alias_taken.size() < db.stream_first.size()) db.alias_taken.resize(db.stream_first.size(), 0);
      if (share && cl == 0 && ct == 0 && coded_size(dw, 2) == c.w && coded_size(dh, 2) == c.h && !db.alias_taken[si]) {
        db.alias_taken[si] = 1;
        d.alias_pix.assign(cnt, nullptr); d.alias_coef.assign(cnt, nullptr);
        for (int k = 0; k < cnt; k++) {
          const RbtFrame& fr = db.frames[first + k];
          if (fr.cfg.w != c.w || fr.cfg.h != c.h) continue;
          d.alias_coef[k] = fr.coef[0];
          if (fr.out[0] != fr.pix[0]) d.alias_pix[k] = fr.pix[0];        // pictures with SAO: the deblocked samples are dead once the SAO output exists
        }
      } }
    // the input stream's intra modes come along as hints for the re-encode (same sample grid: not with a window offset at the left / top; oracle/vpcc_path.c)
    if (cl == 0 && ct == 0) {
      d.hint_pm.resize(cnt); d.hint_dm.resize(cnt); d.hint_w4 = c.w4; d.hint_h4 = c.h4;
      for (int k = 0; k < cnt; k++) { d.hint_pm[k] = db.frames[first + k].pm; d.hint_dm[k] = db.frames[first + k].dm; }
    }
    // occupancy-aware coding (oracle/vpcc_path.c transcode_substream_occ): picture k belongs to occupancy frame k * n_occ / cnt; the occupancy video must be the
    // atlas scaled down by a whole factor, else every sample counts
    if (occ && occ_jobs && occ->n > 0 && cnt % occ->n == 0 && dw % occ->ow == 0 && dh % occ->oh == 0 && dw / occ->ow == dh / occ->oh) {
      const int w4 = (dw + 3) / 4, h4 = (dh + 3) / 4;
      uint8_t* maps = (uint8_t*)rbtk::dev_alloc((size_t)occ->n * w4 * h4);
      if (!maps) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
      pooled.push_back(maps); occ_jobs->push_back(OccJob{occ_index, maps, dw, w4, h4});
      d.occ4.resize(cnt); d.occ4_w = w4; d.occ4_h = h4;
      for (int k = 0; k < cnt; k++) d.occ4[k] = maps + (size_t)((size_t)k * occ->n / cnt) * w4 * h4;
    }
  }
  return 0;
}

// The sub-bitstreams of a GOF are independent (PCCTranscoder.cpp:122-165 transcodes them one after the other), so each
// gets its own decode/encode batch on its own HIP stream: every decode chain is enqueued up front, longest first, and
// the host then walks the streams shortest first, so the pool / re-encode of the short streams (occupancy, geometry)
// runs underneath the entropy-decoding chain of the longest one (attribute) instead of behind it.
// One transcode call in flight. rbt_transcode_gof = submit + wait; rbt_submit_gof / rbt_wait_gof expose the two halves so that
// a caller can keep two GOFs in flight (job slots use disjoint HIP streams): the next GOF's entropy decoding then runs
// underneath the previous GOF's reconstruction and re-encode.
struct GofJob {
  int n = 0, slot = 0, ng = 0, rc = 0;
  std::vector<std::vector<int>> groups; std::vector<int> order;
  std::vector<DecodeBatch> db; std::vector<EncodeBatch> eb; std::vector<char> chained;
  std::vector<void*> pooled;
  bool has_aux = true;
  std::vector<int> phys;                       // HIP stream of each pipeline
  std::vector<char> parse_timed;               // pipeline recorded its own T_PARSE timer
  std::vector<std::vector<RbtParseTask>> tasks_keep;   // host staging of merged parse launches, alive until the job is collected
  std::vector<std::vector<RbtFrameRef>> refs_keep;     // ... and of merged reconstruction launches
  std::vector<char> recon_timed;
  std::vector<rbt_stream_params> params; std::vector<size_t> n_in;
  std::vector<std::vector<uint8_t>> passthrough;   // transcodeData (PCCTranscoder.cpp:150): occupancy is only transcoded when occupancyPrecision == 4; else the stream stays as it is
  std::vector<char> is_pass;
  std::vector<std::vector<int>> dec_of;            // per pipeline: decode stream of each of its (encode) streams - identical inputs are decoded once
  rbt_stats st; std::string err; double t_all = 0, t_gpu = 0; size_t dev_bytes = 0;
  ~GofJob() { for (void* q : pooled) rbtk::dev_free(q); }
};
static int job_stream(const GofJob& j, int pipeline) { return j.slot * rbtk::RBT_STREAMS_PER_JOB + pipeline; }

// Streams of a job. The 16 HIP streams are shared out by the pipeline depth the caller announced (rbt_set_depth): up to 4 jobs
// in flight get four streams each (three pipelines + the auxiliary stream), 5 get three (no auxiliary stream), up to 8 get two
// (the longest pipeline alone, the others behind each other on the second).
static void bind_streams(GofJob& j, int depth) {
  const int spj = depth <= 4 ? 4 : depth == 5 ? 3 : depth <= 8 ? 2 : 1, base = j.slot * spj;
  j.has_aux = spj == 4;
  j.phys.assign(j.ng, 0);
  for (int k = 0; k < j.ng; k++) { j.phys[j.order[k]] = base + (k == 0 || spj == 1 ? 0 : 1 + (k - 1) % (std::min(spj, 3) - 1)); rbtk::map_lane(job_stream(j, j.order[k]), j.phys[j.order[k]]); }
  rbtk::map_lane(job_stream(j, rbtk::RBT_AUX_STREAM), base + spj - 1);
}

size_t gof_memory(const GofJob* j) { return j ? j->dev_bytes : 0; }
GofJob* gof_submit(int slot, int depth, int n, const uint8_t* const* in, const size_t* n_in, const rbt_stream_params* p, bool gof_rule) {
  GofJob* J = new GofJob(); GofJob& j = *J;
  struct Footprint { GofJob& j; size_t a0; ~Footprint() { j.dev_bytes = rbtk::dev_alloc_total() - a0; } } footprint{j, rbtk::dev_alloc_total()};
  j.t_all = now_ms(); j.n = n; j.slot = slot; memset(&j.st, 0, sizeof(j.st));
  j.params.assign(p, p + n); j.n_in.assign(n_in, n_in + n);
  rbt_stats& st = j.st; std::string& err = j.err;
  // Pipelines: up to three sub-bitstreams get one pipeline (= HIP stream) each. A call with more (several GOFs at once:
  // one GOF leaves most of the GPU idle) groups them by video type, so that the slices of all attribute streams parse
  // in one launch, all geometry streams in another, ...
  std::vector<std::vector<int>>& groups = j.groups;
  j.passthrough.resize(n); j.is_pass.assign(n, 0);
  int n_live = 0;
  for (int i = 0; i < n; i++) {
    if (gof_rule && p[i].video_type == RBT_VIDEO_OCCUPANCY && p[i].occupancy_precision != 4) { j.is_pass[i] = 1; j.passthrough[i].assign(in[i], in[i] + n_in[i]); }
    else n_live++;
  }
  if (n_live <= rbtk::RBT_AUX_STREAM) { for (int i = 0; i < n; i++) if (!j.is_pass[i]) groups.push_back({i}); }
  else {
    const int types[3] = {RBT_VIDEO_ATTRIBUTE, RBT_VIDEO_GEOMETRY, RBT_VIDEO_OCCUPANCY};
    for (int t = 0; t < 3; t++) { std::vector<int> g; for (int i = 0; i < n; i++) if (!j.is_pass[i] && p[i].video_type == types[t]) g.push_back(i); if (!g.empty()) groups.push_back(g); }
    std::vector<int> rest; for (int i = 0; i < n; i++) if (!j.is_pass[i] && p[i].video_type != types[0] && p[i].video_type != types[1] && p[i].video_type != types[2]) rest.push_back(i);
    if (!rest.empty()) { if (groups.size() < 3) groups.push_back(rest); else groups.back().insert(groups.back().end(), rest.begin(), rest.end()); }
  }
  const int ng = j.ng = (int)groups.size();
  // Streams of a pipeline that are the same input (same buffer: one sub-bitstream re-encoded at several rate points) are decoded once
  j.dec_of.resize(ng);
  std::vector<std::vector<int>> uniq(ng);          // per pipeline: the stream indices that are decoded
  for (int g = 0; g < ng; g++) for (int i : groups[g]) {
    int d = -1;
    for (size_t q = 0; q < uniq[g].size(); q++) if (in[uniq[g][q]] == in[i] && n_in[uniq[g][q]] == n_in[i] && !p[i].verify_md5 && !p[uniq[g][q]].verify_md5) { d = (int)q; break; }
    if (d < 0) { d = (int)uniq[g].size(); uniq[g].push_back(i); }
    j.dec_of[g].push_back(d);
  }
  auto bytes_of = [&](int g) { size_t t = 0; for (int i : uniq[g]) t += n_in[i]; return t; };
  j.db.resize(ng); j.eb.resize(ng); j.chained.assign(ng, 0);
  std::vector<DecodeBatch>& db = j.db; std::vector<EncodeBatch>& eb = j.eb; std::vector<char>& chained = j.chained; std::vector<void*>& pooled = j.pooled;
  std::vector<int>& order = j.order; order.resize(ng); for (int i = 0; i < ng; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bytes_of(a) > bytes_of(b); });
  const int aux = job_stream(j, rbtk::RBT_AUX_STREAM);
  bind_streams(j, depth);
  recon_set_depth(depth); rbtk::set_jobs_in_flight(depth);
  // ---- phase A, longest pipeline first: build decoder and encoder batches and upload them; then enqueue decode -> pool ->
  // encode on the stream without a host round trip in between (PCCTranscoder.cpp:428-448, :466, :825-904). Every upload of
  // the job is issued before its first kernel: a copy from pageable memory blocks the host until the stream has reached it,
  // and pipelines may share a stream. Pipelines that ask for the input MD5 check keep the decoder / encoder split, because
  // the check needs the decoded pictures on the host first.
  j.t_gpu = now_ms();
  int rc = 0;
  std::vector<std::vector<PoolJob>> pool_jobs(ng);
  // occupancy-aware coding (rbt_stream_params.occupancy_rd): entry i is coded with the occupancy map of the nearest occupancy entry in front of it, if this call pools it
  std::vector<int> occ_of(n, -1), group_of(n, -1), pos_of(n, -1); bool any_occ_rd = false;
  for (int g = 0; g < ng; g++) for (size_t q = 0; q < groups[g].size(); q++) { group_of[groups[g][q]] = g; pos_of[groups[g][q]] = (int)q; }
  for (int i = 0, last = -1; i < n; i++) {
    if (p[i].video_type == RBT_VIDEO_OCCUPANCY) last = (gof_rule && !j.is_pass[i] && p[i].occupancy_precision == 4) ? i : -1;
    else if (gof_rule && p[i].occupancy_rd && last >= 0) { occ_of[i] = last; any_occ_rd = true; if (p[i].verify_md5 || p[last].verify_md5) { err = "occupancy_rd cannot be combined with verify_md5"; rc = RBT_ERR_PARAM; } }
  }
  std::vector<OccSource> occ_src(n); std::vector<OccJob> occ_jobs;
  std::vector<char> verify_of(ng, 0);
  for (int k = 0; k < ng && !rc; k++) {
    const int gi = order[k], sid = job_stream(j, gi); const std::vector<int>& gs = groups[gi]; rbtk::set_stream(sid);
    std::vector<StreamIn> sins; bool verify = false;
    for (int i : uniq[gi]) sins.push_back(StreamIn{in[i], n_in[i]});
    for (int i : gs) verify |= p[i].verify_md5 != 0;
    verify_of[gi] = verify;
    double t0 = now_ms();
    db[gi].want_save = parse_bands() > 1 && k == 0 && j.has_aux && ng <= rbtk::RBT_AUX_STREAM && !verify;
    rc = decode_build(db[gi], sins.data(), (int)sins.size());
    st.host_parse_ms += now_ms() - t0;
    if (!rc) rc = decode_upload_lists(db[gi]);
    if (rc) { err = db[gi].err; break; }
  }
  // encoder set-up: the pipelines whose occupancy streams others are coded with first (their pooled planes are what the maps are made of)
  for (int pass = 0; pass < 2 && !rc; pass++) for (int k = 0; k < ng && !rc; k++) {
    const int gi = order[k], sid = job_stream(j, gi); const std::vector<int>& gs = groups[gi]; rbtk::set_stream(sid);
    bool feeds = false; for (int i : gs) for (int c = 0; c < n; c++) feeds |= occ_of[c] == i;
    if (feeds != (pass == 0) || verify_of[gi]) continue;
    for (size_t q = 0; q < gs.size() && !rc; q++) {
      const int i = gs[q], io = occ_of[i];
      rc = setup_encode(db[gi], j.dec_of[gi][q], (int)q, p[i], eb[gi], pooled, err, &pool_jobs[gi], io >= 0 ? &occ_src[io] : nullptr, io, &occ_jobs);
      if (!rc && feeds && p[i].video_type == RBT_VIDEO_OCCUPANCY) {
        const EncStreamDesc& d = eb[gi].desc[q];
        if (d.n_frames > 0) { occ_src[i].occ = d.src[0][0]; occ_src[i].in_step = d.n_frames > 1 ? (size_t)(d.src[0][1] - d.src[0][0]) : 0; occ_src[i].n = d.n_frames; occ_src[i].ow = d.w; occ_src[i].oh = d.h; occ_src[i].pipeline = gi; }
      }
    }
    if (!rc) { rc = encode_build(eb[gi]); if (!rc) rc = encode_upload_lists(eb[gi]); if (rc) err = eb[gi].err; }
    if (rc) break;
    chained[gi] = 1;
  }
  (void)group_of; (void)pos_of;
  // Pipelines that share a HIP stream would parse one after the other; their slices go into one merged launch instead, so
  // that all parsers of the stream run side by side and only the (short) tails of the pipelines follow each other.
  j.parse_timed.assign(ng, 1); j.recon_timed.assign(ng, 1);
  for (int k = 0; k < ng && !rc; k++) {
    const int lead = order[k];
    if (!chained[lead] || db[lead].parse_external) continue;
    std::vector<int> grp;
    for (int q = k; q < ng; q++) { const int gi = order[q]; if (j.phys[gi] == j.phys[lead] && chained[gi] && !db[gi].ordered_parse && !db[gi].d_save) grp.push_back(gi); }
    if (grp.size() < 2 || grp[0] != lead) continue;
    j.tasks_keep.emplace_back(); std::vector<RbtParseTask>& tasks = j.tasks_keep.back(); int mw4 = 0;
    std::vector<uint32_t> task_bytes; bool any_row_tasks = false;
    for (int gi : grp) {
      for (size_t i = 0; i < db[gi].slices.size(); i++) { tasks.push_back(RbtParseTask{db[gi].d_frames, db[gi].d_slices, db[gi].d_rbsp, db[gi].lists_keep[i], 0}); task_bytes.push_back(db[gi].slices[(size_t)db[gi].lists_keep[i]].data_size); }
      mw4 = std::max(mw4, decode_max_w4(db[gi])); db[gi].parse_external = true; j.parse_timed[gi] = gi == lead; any_row_tasks |= db[gi].has_row_tasks;
    }
    // Longest first: a launch's waves start in list order, and when more slices are in flight than the GPU holds waves (several jobs' launches side by side) the ones that
    // wait should be the short ones - the launch lasts as long as its largest slice (an attribute IDR) however late that one starts. Not for lists with row tasks of
    // wavefront streams: there a task must come after the task of the row above it.
    if (!any_row_tasks) {
      std::vector<size_t> ord(tasks.size()); for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return task_bytes[a] > task_bytes[b]; });
      std::vector<RbtParseTask> sorted; sorted.reserve(tasks.size()); for (size_t i : ord) sorted.push_back(tasks[i]);
      tasks.swap(sorted);
    }
    // ... and the pictures of one dependency level of all of them go on one wavefront (58 launches per level instead of 58
    // per level and pipeline, one after the other)
    size_t n_levels = 0; for (int gi : grp) n_levels = std::max(n_levels, db[gi].level_frames.size());
    j.refs_keep.emplace_back(); std::vector<RbtFrameRef>& refs = j.refs_keep.back();
    std::vector<size_t> lv_off(n_levels + 1, 0); std::vector<int> lv_w(n_levels, 0), lv_h(n_levels, 0);
    std::vector<size_t> lq_off(n_levels + 1, 0); std::vector<uint32_t> lq_total(n_levels, 0); std::vector<int> lq_wgs(n_levels, 0);      // ready queues of the merged levels
    for (size_t l = 0; l < n_levels; l++) {
      lv_off[l] = refs.size();
      for (int gi : grp) if (l < db[gi].level_frames.size()) for (int fi : db[gi].level_frames[l]) {
        refs.push_back(RbtFrameRef{db[gi].d_frames, db[gi].d_slices, db[gi].d_order + db[gi].order_off[fi], fi, 0});
        const RbtStreamCfg& c = db[gi].frames[fi].cfg;
        lv_w[l] = std::max(lv_w[l], (int)c.w_ctb); lv_h[l] = std::max(lv_h[l], (int)c.h_ctb);
        lq_total[l] += (uint32_t)(c.w_ctb * c.h_ctb); lq_wgs[l] += recon_queue_width(c);
      }
      lq_off[l + 1] = lq_off[l] + rbtk::recon_queue_words(lq_total[l]);
    }
    lv_off[n_levels] = refs.size();
    int rmode = recon_mode();
    for (size_t l = 0; l < n_levels; l++) if (lv_off[l + 1] - lv_off[l] >= 8192 || (size_t)lv_w[l] * lv_h[l] >= ((size_t)1 << 18)) { if (rmode == 2) rmode = 0; }
    RbtParseTask* d_tasks = (RbtParseTask*)rbtk::dev_alloc(tasks.size() * sizeof(RbtParseTask));
    RbtFrameRef* d_refs = (RbtFrameRef*)rbtk::dev_alloc(refs.size() * sizeof(RbtFrameRef));
    if (d_tasks) pooled.push_back(d_tasks);
    if (d_refs) pooled.push_back(d_refs);
    uint32_t* d_queue = rmode == 2 ? (uint32_t*)rbtk::dev_alloc(lq_off[n_levels] * 4) : nullptr;
    if (d_queue) pooled.push_back(d_queue);
    if (!d_tasks || !d_refs || (rmode == 2 && !d_queue)) { err = "device allocation failed"; rc = RBT_ERR_NOMEM; break; }
    rbtk::set_stream(job_stream(j, lead));
    if (d_queue && rbtk::dev_memset(d_queue, 0, lq_off[n_levels] * 4)) { err = "device transfer failed"; rc = RBT_ERR_NO_DEVICE; break; }
    if (rbtk::h2d(d_tasks, tasks.data(), tasks.size() * sizeof(RbtParseTask)) || rbtk::h2d(d_refs, refs.data(), refs.size() * sizeof(RbtFrameRef))) { err = "device transfer failed"; rc = RBT_ERR_NO_DEVICE; break; }
    bool row_tasks = false; for (int gi : grp) row_tasks |= db[gi].has_row_tasks;
    rbtk::timer_begin(T_PARSE); rbtk::launch_parse_tasks(d_tasks, (int)tasks.size(), mw4, row_tasks ? db[lead].d_tickets + 96 : nullptr); rbtk::timer_end(T_PARSE);
    rbtk::timer_begin(T_RECON);
    for (size_t l = 0; l < n_levels; l++) {
      // the merged launch of level l uses the lead batch's spare ticket counter 32 + l (its own levels use 0..31)
      if (rmode == 0) rbtk::launch_recon_refs(d_refs + lv_off[l], (int)(lv_off[l + 1] - lv_off[l]), lv_w[l], lv_h[l]);
      else if (rmode == 1) rbtk::launch_recon_level(d_refs + lv_off[l], (int)(lv_off[l + 1] - lv_off[l]), lv_w[l] * lv_h[l], db[lead].d_tickets + 32 + l);
      else rbtk::launch_recon_queue(d_refs + lv_off[l], (int)(lv_off[l + 1] - lv_off[l]), lq_total[l], d_queue + lq_off[l], lq_wgs[l]);
      for (int gi : grp) if (l < db[gi].level_frames.size()) decode_launch_filters(db[gi], l);
    }
    rbtk::timer_end(T_RECON);
    for (int gi : grp) { db[gi].recon_external = true; j.recon_timed[gi] = gi == lead; }
  }
  // enqueue order: longest pipeline first - except that a pipeline whose occupancy maps others wait for goes in front of them (an event has to be recorded before it is waited for)
  std::vector<int> lorder; std::vector<char> feeds_any(ng, 0), consumes(ng, 0);
  for (const OccJob& oj : occ_jobs) feeds_any[occ_src[oj.source].pipeline] = 1;
  for (int g = 0; g < ng; g++) for (const EncStreamDesc& d : eb[g].desc) consumes[g] |= !d.occ4.empty();
  for (int k = 0; k < ng; k++) if (feeds_any[order[k]]) lorder.push_back(order[k]);
  for (int k = 0; k < ng; k++) if (!feeds_any[order[k]]) lorder.push_back(order[k]);
  std::vector<int> occ_marks;
  for (int k = 0; k < ng && !rc; k++) {
    const int gi = lorder[k], sid = job_stream(j, gi); rbtk::set_stream(sid);
    const std::vector<PoolJob>& jobs = pool_jobs[gi];
    if (!chained[gi]) { rc = decode_launch(db[gi]); if (rc) { err = db[gi].err; break; } continue; }
    // Intra pictures of the output only read the decoded pictures they are re-encoded from. When those are complete
    // before the last dependency level of the decoder, analysis + intra coding run on an auxiliary stream underneath the
    // remaining reconstruction levels.
    EncodeBatch& e = eb[gi]; e.main_stream = sid;
    size_t n_levels = db[gi].level_frames.size(), fork_level = 0;
    for (size_t q = 0; q < e.frames.size(); q++) if (e.frame_is_idr[q]) {
      const int si = e.frame_stream[q], local = (int)q - e.stream_first[si], ds = j.dec_of[gi][si];
      fork_level = std::max(fork_level, (size_t)db[gi].frames[db[gi].stream_first[ds] + local].level);
    }
    int intra_done = 0;
    const bool fork = gi == order[0] && j.has_aux && ng <= rbtk::RBT_AUX_STREAM && jobs.empty() && e.pad_jobs.empty() && fork_level + 1 < n_levels;   // longest pipeline only: one spare stream
    const bool banded = db[gi].d_save != nullptr && !db[gi].ordered_parse;
    rc = banded ? decode_launch_chunked(db[gi], parse_bands(), sid, aux) : decode_launch_parse(db[gi]);
    if (rc) { err = db[gi].err; break; }
    if (!db[gi].recon_external) rbtk::timer_begin(T_RECON);
    for (size_t l = 0; l < n_levels && !db[gi].recon_external; l++) {
      if (!(banded && l == 0)) decode_launch_level(db[gi], l);
      if (fork && l == fork_level) {
        e.aux_stream = aux;
        rbtk::stream_wait(e.aux_stream, sid);
        if (consumes[gi]) for (int m : occ_marks) rbtk::stream_wait_mark(e.aux_stream, m);      // occupancy-aware coding: the maps are made on the occupancy pipeline's stream
        rbtk::set_stream(e.aux_stream); encode_launch_intra(e); intra_done = rbtk::stream_mark(e.aux_stream); encode_launch_entropy_intra(e); rbtk::set_stream(sid);
      }
    }
    if (!db[gi].recon_external) rbtk::timer_end(T_RECON);
    if (!jobs.empty()) {
      // the pictures of one stream are of one size and their pooled copies evenly spaced (setup_encode): one launch per run of such jobs
      rbtk::timer_begin(T_POOL);
      for (size_t a0 = 0; a0 < jobs.size();) {
        size_t a1 = a0 + 1; const PoolJob& p0 = jobs[a0];
        const size_t step = a1 < jobs.size() ? (size_t)(jobs[a1].y - p0.y) : 0;
        while (a1 < jobs.size() && jobs[a1].stride == p0.stride && jobs[a1].w == p0.w && jobs[a1].h == p0.h && jobs[a1].grey == p0.grey && (size_t)(jobs[a1].y - p0.y) == step * (a1 - a0)) a1++;
        std::vector<const uint16_t*> ins; for (size_t q = a0; q < a1; q++) ins.push_back(jobs[q].in);
        rbtk::launch_pool_many(ins.data(), (int)ins.size(), p0.stride, p0.w, p0.h, 2, p0.y, step, p0.grey);
        a0 = a1;
      }
      rbtk::timer_end(T_POOL);
    }
    if (feeds_any[gi]) {
      for (const OccJob& oj : occ_jobs) { const OccSource& os = occ_src[oj.source]; if (os.pipeline == gi) rbtk::launch_occ_units(os.occ, os.in_step, os.n, os.ow, os.oh, oj.W, oj.w4, oj.h4, oj.maps); }
      occ_marks.push_back(rbtk::stream_mark(sid));
    }
    if (consumes[gi]) for (int m : occ_marks) rbtk::stream_wait_mark(sid, m);
    if (fork) rbtk::stream_wait_mark(sid, intra_done); else { encode_launch_intra(e); encode_launch_entropy_intra(e); }
    encode_launch_rest(e);
    if (fork) rbtk::stream_wait(sid, e.aux_stream);      // the intra pictures' entropy coding on the auxiliary stream
  }
  j.rc = rc;
  return J;
}

// phase B, shortest pipeline first: one sync per stream, then slice sizes -> pack -> NAL assembly. Consumes the job.
int gof_wait(GofJob* J, rbt_stats& st_out, std::string& err_out, uint8_t** out, size_t* n_out) {
  std::unique_ptr<GofJob> guard(J); GofJob& j = *J;
  const int n = j.n, ng = j.ng; int rc = j.rc;
  rbt_stats& st = j.st; std::string& err = j.err;
  std::vector<DecodeBatch>& db = j.db; std::vector<EncodeBatch>& eb = j.eb; const rbt_stream_params* p = j.params.data();
  for (int i = 0; i < n; i++) { out[i] = nullptr; n_out[i] = 0; }
  std::vector<std::vector<uint8_t>> outs(n);
  for (int k = ng - 1; k >= 0; k--) {
    const int gi = j.order[k], sid = job_stream(j, gi); const std::vector<int>& gs = j.groups[gi]; rbtk::set_stream(sid);
    if (rc) { rbtk::dev_sync(); continue; }              // drain the remaining streams before their arenas are released
    if (db[gi].frames.empty()) continue;
    rc = decode_finish(db[gi]);
    if (rc) { err = db[gi].err; continue; }
    if (j.parse_timed.empty() || j.parse_timed[gi]) st.k_parse_ms += rbtk::timer_ms(T_PARSE);
    if (j.recon_timed.empty() || j.recon_timed[gi]) st.k_recon_ms += rbtk::timer_ms(T_RECON);
    std::vector<std::vector<uint8_t>> o1;
    if (j.chained[gi]) rc = encode_finish(eb[gi], o1, st);
    else {
      for (size_t q = 0; q < gs.size() && !rc; q++) if (p[gs[q]].verify_md5) {
        rbt_video v; rc = decode_fetch(db[gi], j.dec_of[gi][q], &v, true); free(v.data);
        if (rc) { err = "fetch failed"; break; }
        if (v.md5_failed) { err = "input MD5 mismatch"; rc = RBT_ERR_MD5; }
      }
      for (size_t q = 0; q < gs.size() && !rc; q++) rc = setup_encode(db[gi], j.dec_of[gi][q], (int)q, p[gs[q]], eb[gi], j.pooled, err);
      if (rc) continue;
      eb[gi].main_stream = sid;
      rc = encode_build(eb[gi]);
      if (!rc) rc = encode_run(eb[gi], o1, st);
    }
    if (rc) { if (err.empty()) err = eb[gi].err; continue; }
    for (size_t q = 0; q < gs.size(); q++) outs[gs[q]].swap(o1[q]);
  }
  rbtk::set_stream(0);
  if (rc) { err_out = err; st_out = st; return rc; }
  for (int i = 0; i < n; i++) if (j.is_pass[i]) outs[i].swap(j.passthrough[i]);
  st.gpu_ms = now_ms() - j.t_gpu;
  rc = hand_out(outs, out, n_out);
  // SURVEY.md 8(d) algorithmic traffic: per coded picture of S samples (2 bytes each): decode writes S, P pictures read
  // their reference once; encode reads the source S, writes the reconstruction S (I) and reads the reference (P)
  uint64_t bytes = 0;
  for (int g = 0; g < ng; g++) {
    for (size_t k = 0; k < db[g].frames.size(); k++) { uint64_t s2 = frame_samples(db[g].frames[k].cfg) * 2; bytes += s2 + (db[g].frames[k].level ? s2 : 0); }
    for (size_t k = 0; k < eb[g].frames.size(); k++) { uint64_t s2 = frame_samples(eb[g].frames[k].cfg) * 2; bytes += s2 + s2; }
  }
  for (int i = 0; i < n; i++) bytes += j.n_in[i] + n_out[i];
  st.algorithmic_bytes = bytes;
  st.total_ms = now_ms() - j.t_all;
  st_out = st; err_out = err;
  return rc;
}
void gof_abandon(GofJob* J) {   // a job nobody will wait for: drain its streams, then free it
  if (!J) return;
  for (int g = 0; g < J->ng; g++) { rbtk::set_stream(job_stream(*J, g)); rbtk::dev_sync(); }
  rbtk::set_stream(job_stream(*J, rbtk::RBT_AUX_STREAM)); rbtk::dev_sync(); rbtk::set_stream(0);
  delete J;
}
int transcode_gof(rbt_stats& st, std::string& err, int n, const uint8_t* const* in, const size_t* n_in, const rbt_stream_params* p, uint8_t** out, size_t* n_out) {
  return gof_wait(gof_submit(0, 1, n, in, n_in, p, true), st, err, out, n_out);
}

int encode_yuv(rbt_stats& st, std::string& err, const uint16_t* yuv, int w, int h, int bd, int n_frames, int qp, int gop, int lossless, int log2_ctb, int rows, int md5,
               uint8_t** out, size_t* n_out) {
  memset(&st, 0, sizeof(st));
  *out = nullptr; *n_out = 0;
  if (w <= 0 || h <= 0 || w % 2 || h % 2 || bd < 8 || bd > 12) { err = "bad picture format"; return RBT_ERR_PARAM; }
  size_t ys = (size_t)w * h, cs = (size_t)(w / 2) * (h / 2), fs = ys + 2 * cs;
  uint16_t* buf = (uint16_t*)rbtk::dev_alloc(fs * 2 * (size_t)n_frames);
  if (!buf) { err = "device allocation failed"; return RBT_ERR_NOMEM; }
  struct G { void* p; ~G() { rbtk::dev_free(p); } } g{buf};
  if (rbtk::h2d(buf, yuv, fs * 2 * (size_t)n_frames)) { err = "device transfer failed"; return RBT_ERR_NO_DEVICE; }
  EncodeBatch eb; eb.desc.resize(1);
  EncStreamDesc& d = eb.desc[0];
  d.w = w; d.h = h; d.bd = bd; d.n_frames = n_frames; d.qp = qp; d.i_qp_offset = lossless ? 0 : -3; d.gop = gop; d.lossless = lossless; d.log2_ctb = log2_ctb; d.rows = rows; d.md5 = md5;
  for (int k = 0; k < 3; k++) d.src[k].resize(n_frames);
  for (int i = 0; i < n_frames; i++) { uint16_t* y = buf + fs * (size_t)i; d.src[0][i] = y; d.src[1][i] = y + ys; d.src[2][i] = y + ys + cs; }
  int rc = encode_build(eb);
  std::vector<std::vector<uint8_t>> outs;
  if (!rc) rc = encode_run(eb, outs, st);
  if (rc) { err = eb.err; return rc; }
  return hand_out(outs, out, n_out);
}

int or_pool_host(const uint16_t* plane, int w, int h, int factor, uint16_t* out) {
  int ow = w / factor, oh = h / factor;
  size_t in_n = (size_t)w * h, out_n = (size_t)ow * oh;
  uint16_t* buf = (uint16_t*)rbtk::dev_alloc((in_n + out_n * 2) * 2);
  if (!buf) return RBT_ERR_NOMEM;
  struct G { void* p; ~G() { rbtk::dev_free(p); } } g{buf};
  if (rbtk::h2d(buf, plane, in_n * 2)) return RBT_ERR_NO_DEVICE;
  rbtk::launch_pool(buf, w, w, h, factor, buf + in_n, buf + in_n + out_n, buf + in_n + out_n + out_n / 4, 0);
  if (rbtk::d2h(out, buf + in_n, out_n * 2)) return RBT_ERR_NO_DEVICE;
  return 0;
}

}  // namespace rbt
