// Decode half of the hot path: replaces initDecoder + the av_read_frame / avcodec_send_packet / avcodec_receive_frame
// loop of PCCTranscoder::transcodeVideo (PCCTranscoder.cpp:404, :428-448).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "rbt_batch.h"

namespace rbt {

size_t frame_samples(const RbtStreamCfg& c) { return (size_t)c.w * c.h + 2 * (size_t)c.cw * c.ch; }

struct DpbEntry { int poc, frame; };

int decode_build(DecodeBatch& b, const StreamIn* streams, int n_streams) {
  b.stream_first.assign(n_streams, 0); b.stream_count.assign(n_streams, 0);
  b.stream_sps.resize(n_streams); b.stream_pps.resize(n_streams);
  // ---- host parse ----
  for (int si = 0; si < n_streams; si++) {
    std::vector<Nal> nals; size_t rbsp_start = b.rbsp.size();
    split_annexb(streams[si].p, streams[si].n, b.rbsp, nals);
    (void)rbsp_start;
    ParamSets* ps = new ParamSets();
    std::vector<DpbEntry> dpb; int prev_tid0_poc = 0, cur = -1;
    SliceHdr head_hdr; int head_idx = -1, last_seg = -1;   // the independent segment of the current slice, the last segment of its chain
    b.stream_first[si] = (int)b.frames.size();
    for (const Nal& nal : nals) {
      const uint8_t* r = b.rbsp.data() + nal.rbsp_off; int rc = 0;
      if (nal.type == NAL_SPS) rc = parse_sps(*ps, r, nal.rbsp_size, b.err);
      else if (nal.type == NAL_PPS) rc = parse_pps(*ps, r, nal.rbsp_size, b.err);
      else if (nal.type == NAL_SEI_SUFFIX) { if (cur >= 0) { FrameInfo& fi = b.info[cur]; if (parse_md5_sei(r, nal.rbsp_size, fi.md5)) fi.has_md5 = true; } }
      else if (nal.type <= NAL_TRAIL_R || (nal.type >= 16 && nal.type <= 21)) {
        SliceHdr h; rc = parse_slice_header(*ps, r, nal.rbsp_size, nal.type, h, b.err, head_idx >= 0 ? &head_hdr : nullptr);
        if (rc) { delete ps; b.err_code = rc == -3 ? RBT_ERR_UNSUPPORTED : RBT_ERR_BITSTREAM; return b.err_code; }
        const Pps& pps = ps->pps[h.pps_id]; const Sps& sps = ps->sps[pps.sps_id];
        if (h.first_slice_in_pic) {
          h.poc = slice_poc(sps, nal.type, h.poc_lsb, prev_tid0_poc);
          RbtFrame f; memset(&f, 0, sizeof(f));
          fill_stream_cfg(sps, pps, f.cfg); f.poc = h.poc; f.first_slice = (int)b.slices.size();
          int ctb = 1 << sps.log2_ctb; f.cmd_cap = 2 * (ctb / 4) * (ctb / 4);
          if (b.stream_count[si] > 0 && memcmp(&b.frames[b.stream_first[si]].cfg, &f.cfg, sizeof(f.cfg)) != 0) {
            // parameter sets may be repeated but must not change inside one sub-bitstream of a GOF
            const RbtStreamCfg& c0 = b.frames[b.stream_first[si]].cfg;
            if (c0.w != f.cfg.w || c0.h != f.cfg.h || c0.bit_depth != f.cfg.bit_depth) { delete ps; b.err = "picture size changes inside a stream"; return b.err_code = RBT_ERR_UNSUPPORTED; }
          }
          cur = (int)b.frames.size(); b.frames.push_back(f);
          FrameInfo fi; memset(&fi, 0, sizeof(fi)); fi.stream = si; fi.nal_type = nal.type; b.info.push_back(fi);
          b.stream_count[si]++; b.stream_sps[si] = sps; b.stream_pps[si] = pps;
          dpb.insert(dpb.begin(), DpbEntry{h.poc, cur});
          head_idx = -1;
        } else if (cur < 0) { delete ps; b.err = "slice segment before the first picture"; return b.err_code = RBT_ERR_BITSTREAM; }
        else h.poc = b.frames[cur].poc;
        // slice segments of a picture come in increasing address order (7.4.7.1), the first one at 0: with that, segment i has to cover exactly the CTBs up to the
        // start of segment i + 1 (RbtSlice::end_addr, checked by the parser where the segment ends) - no overlap, no hole, nothing for a row task to wait for in vain
        { const int n_ctb = sps.w_ctb * sps.h_ctb, prev = b.frames[cur].n_slices ? b.slices.back().ctb_addr : -1;
          if ((int)h.segment_addr <= prev || (int)h.segment_addr >= n_ctb || (prev < 0 && h.segment_addr != 0)) { delete ps; b.err = "slice segment address out of order"; return b.err_code = RBT_ERR_BITSTREAM; }
          if (prev >= 0) b.slices.back().end_addr = (int32_t)h.segment_addr; }
        RbtSlice s; memset(&s, 0, sizeof(s));
        s.frame = cur; s.data_off = (uint32_t)(nal.rbsp_off + h.data_byte_offset);
        if (h.data_byte_offset > nal.rbsp_size) { delete ps; b.err = "empty slice data"; return b.err_code = RBT_ERR_BITSTREAM; }
        s.data_size = (uint32_t)(nal.rbsp_size - h.data_byte_offset); s.ctb_addr = h.segment_addr;
        s.slice_type = (int8_t)h.slice_type; s.qp = (int8_t)h.qp; s.cb_qp_offset = (int8_t)h.cb_qp_offset; s.cr_qp_offset = (int8_t)h.cr_qp_offset;
        s.sao_luma = (uint8_t)h.sao_luma; s.sao_chroma = (uint8_t)h.sao_chroma; s.deblocking_disabled = (uint8_t)h.deblocking_disabled; s.lf_across = (uint8_t)h.lf_across;
        s.beta_offset_div2 = (int8_t)h.beta_offset_div2; s.tc_offset_div2 = (int8_t)h.tc_offset_div2; s.temporal_mvp = (uint8_t)h.temporal_mvp;
        s.cabac_init_flag = (uint8_t)h.cabac_init_flag; s.max_merge_cand = (uint8_t)h.max_merge_cand; s.num_ref_idx = (uint8_t)h.num_ref_idx;
        s.collocated_ref_idx = (uint8_t)h.collocated_ref_idx; s.poc = h.poc;
        if (h.slice_type == RBT_SLICE_P && h.wp_on) {
          s.wp_on = 1; s.wp_shift[0] = (int8_t)(h.wp_luma_denom + 14 - sps.bit_depth); s.wp_shift[1] = (int8_t)(h.wp_chroma_denom + 14 - sps.bit_depth);
          for (int i = 0; i < h.num_ref_idx && i < RBT_MAX_REFS; i++) for (int c = 0; c < 3; c++) { s.wp_w[i][c] = (int16_t)h.wp_w[i][c]; s.wp_o[i][c] = (int16_t)(h.wp_o[i][c] * (1 << (sps.bit_depth - 8))); }
        }
        if (h.qp < -6 * (sps.bit_depth - 8) || h.qp > 51) { delete ps; b.err = "slice QP out of range"; return b.err_code = RBT_ERR_BITSTREAM; }
        if (h.slice_type == RBT_SLICE_P) {
          int cand[16], nc = 0;
          for (int i = 0; i < h.rps.num; i++) if (h.rps.used[i]) cand[nc++] = h.poc + h.rps.delta_poc[i];
          if (!nc) { delete ps; b.err = "P slice with an empty reference picture set"; return b.err_code = RBT_ERR_BITSTREAM; }
          for (int i = 0; i < h.num_ref_idx; i++) {
            int poc = cand[i % nc], found = -1;
            for (size_t k = 1; k < dpb.size(); k++) if (dpb[k].poc == poc) { found = dpb[k].frame; break; }   // dpb[0] is the current picture
            if (found < 0) { delete ps; b.err = "missing reference picture"; return b.err_code = RBT_ERR_BITSTREAM; }
            s.ref_frame[i] = found; s.ref_poc[i] = poc;
            b.frames[cur].level = std::max(b.frames[cur].level, b.frames[found].level + 1);
            if (h.temporal_mvp && i == h.collocated_ref_idx && b.frames[found].level > 0) b.ordered_parse = true;
          }
        }
        if (h.sao_luma || h.sao_chroma) b.info[cur].sao = true;
        s.wpp = (uint8_t)pps.entropy_coding_sync; s.next_seg = -1; s.head = (int32_t)b.slices.size();
        // a segment of a wavefront stream that starts a CTB row gets a wave of its own (it takes what it needs of the row above from that row's wave:
        // rbt_parse.h); RBT_WPP_PARALLEL=0 keeps one wave per slice
        static const int row_tasks = [] { const char* e = getenv("RBT_WPP_PARALLEL"); return !e || atoi(e) != 0; }();
        s.row_task = (uint8_t)(pps.entropy_coding_sync && row_tasks && h.dependent && h.segment_addr % sps.w_ctb == 0);   // (an independent segment needs nothing of the row above: it is a wave of its own anyway)
        if (s.row_task) b.has_row_tasks = true;
        if (h.dependent) {
          // same slice as the segment before it: parsed by the wave that parsed that one (next_seg chain) unless it starts a row of a wavefront stream
          if (head_idx < 0 || last_seg < 0) { delete ps; b.err = "dependent slice segment without a slice"; return b.err_code = RBT_ERR_BITSTREAM; }
          s.dependent = 1; s.head = head_idx;
          if (!s.row_task) b.slices[last_seg].next_seg = (int32_t)b.slices.size();
        } else { head_hdr = h; head_idx = (int)b.slices.size(); }
        last_seg = (int)b.slices.size();
        b.frames[cur].n_slices++;
        // A segment of a wavefront stream that spans several CTB rows names its substreams by entry points (x265's form): every substream after the first starts
        // a row and becomes a row task of its own. The offsets count bytes AS SENT (emulation prevention bytes included, 7.4.7.1): mapped through the
        // positions where such bytes were removed. Offsets that do not add up leave the segment to one wave, which finds the rows by itself.
        std::vector<RbtSlice> rows;
        if (s.wpp && row_tasks && !h.entry_sizes.empty()) {
          auto sent_to_rbsp = [&](size_t e) { size_t k = 0; while (k < nal.epb.size() && nal.epb[k] < e) k++; return e - k; };   // offset in the NAL as sent -> offset in its RBSP
          size_t e0 = h.data_byte_offset; { size_t k = 0; while (k < nal.epb.size() && (size_t)nal.epb[k] - k < h.data_byte_offset) k++; e0 += k; }
          const int wc = sps.w_ctb, first_row_ctbs = wc - h.segment_addr % wc; bool ok = true;
          std::vector<size_t> u(1, h.data_byte_offset); size_t e = e0;
          for (uint32_t sz : h.entry_sizes) { e += sz; const size_t x = sent_to_rbsp(e); if (x <= u.back() || x >= nal.rbsp_size) { ok = false; break; } u.push_back(x); }
          if (ok && (size_t)h.segment_addr + first_row_ctbs + (h.entry_sizes.size() - 1) * (size_t)wc < (size_t)wc * sps.h_ctb) {
            s.data_size = (uint32_t)(u[1] - u[0]); s.ctb_limit = first_row_ctbs; s.end_addr = (int32_t)h.segment_addr + first_row_ctbs;
            for (size_t i = 1; i < u.size(); i++) {
              RbtSlice t = s; t.dependent = 1; t.row_task = 1; t.next_seg = -1; t.head = s.head;
              t.ctb_addr = h.segment_addr + first_row_ctbs + (int)(i - 1) * wc;
              t.data_off = (uint32_t)(nal.rbsp_off + u[i]); t.data_size = (uint32_t)((i + 1 < u.size() ? u[i + 1] : nal.rbsp_size) - u[i]);
              t.ctb_limit = i + 1 < u.size() ? wc : 0;               // the last substream ends with the segment
              t.end_addr = t.ctb_limit ? t.ctb_addr + wc : 0;        // (the last one: where the next segment starts, filled in when that is known)
              rows.push_back(t);
            }
            b.has_row_tasks = true;
          }
        }
        b.slices.push_back(s);
        for (const RbtSlice& t : rows) { last_seg = (int)b.slices.size(); b.frames[cur].n_slices++; b.slices.push_back(t); }
      } else if (nal.type >= 2 && nal.type <= 9) { rc = -3; b.err = "unsupported VCL NAL unit type"; }
      if (rc) { delete ps; b.err_code = rc == -3 ? RBT_ERR_UNSUPPORTED : RBT_ERR_BITSTREAM; return b.err_code; }
    }
    delete ps;
  }
  if (b.frames.empty()) { b.err = "no pictures in the input"; return b.err_code = RBT_ERR_BITSTREAM; }
  for (size_t i = 0; i < b.slices.size(); i++) if (!b.slices[i].end_addr) { const RbtStreamCfg& c = b.frames[b.slices[i].frame].cfg; b.slices[i].end_addr = c.w_ctb * c.h_ctb; }   // the last entry of a picture ends with the picture
  if (b.has_row_tasks) b.want_save = false;                              // banded (resumable) parsing re-launches the list; row tasks need one ordered launch
  if (b.slices.size() >= 0xFFFF) { b.err = "too many slice segments"; return b.err_code = RBT_ERR_UNSUPPORTED; }
  int n_levels = 0; for (auto& f : b.frames) n_levels = std::max(n_levels, f.level + 1);
  b.level_frames.assign(n_levels, {});
  for (size_t i = 0; i < b.frames.size(); i++) b.level_frames[b.frames[i].level].push_back((int)i);

  // ---- HBM layout: [zero region | pm region (0x03) | ctb_slice region (0xFF) | rest] ----
  Arena a;
  size_t nf = b.frames.size();
  std::vector<size_t> o_coef(nf), o_edges(nf), o_cnt(nf), o_done(nf), o_pm(nf), o_cs(nf), o_pix(nf), o_out(nf), o_dm(nf), o_qp(nf), o_mv(nf), o_ref(nf), o_refpoc(nf), o_sao(nf), o_cmds(nf);
  for (size_t i = 0; i < nf; i++) { const RbtStreamCfg& c = b.frames[i].cfg; size_t u = (size_t)c.w4 * c.h4, nc = (size_t)c.w_ctb * c.h_ctb;
    o_coef[i] = a.reserve(frame_samples(c) * 2); o_edges[i] = a.reserve(u); o_cnt[i] = a.reserve(nc * 4); o_done[i] = a.reserve(nc * 8); }
  size_t o_tickets = a.reserve(128 * 4);
  // ready queues of the reconstruction levels (launch_recon_queue), zeroed with the rest of this region
  b.queue_off.clear(); b.queue_total.clear(); b.queue_wgs.clear();
  size_t queue_words = 0;
  for (auto& lf : b.level_frames) {
    uint32_t tot = 0; int wg = 0;
    for (int fi : lf) { const RbtStreamCfg& c = b.frames[fi].cfg; tot += (uint32_t)(c.w_ctb * c.h_ctb); wg += recon_queue_width(c); }
    b.queue_off.push_back(queue_words); b.queue_total.push_back(tot); b.queue_wgs.push_back(wg); queue_words += rbtk::recon_queue_words(tot);
  }
  size_t o_queue = a.reserve(queue_words * 4);
  // wavefront streams: progress counters of the CTB rows (zeroed with the rest of this region)
  std::vector<size_t> o_prow_done(nf, 0), o_prow_ctx(nf, 0); std::vector<int> wpp_frame(nf, 0);
  for (size_t i = 0; i < nf; i++) { wpp_frame[i] = b.stream_pps[b.info[i].stream].entropy_coding_sync; if (wpp_frame[i]) o_prow_done[i] = a.reserve((size_t)b.frames[i].cfg.h_ctb * 4); }
  size_t o_save = b.want_save ? a.reserve(b.slices.size() * rbtk::parse_save_bytes()) : 0;
  size_t zero_end = a.reserve(0);
  for (size_t i = 0; i < nf; i++) { const RbtStreamCfg& c = b.frames[i].cfg; o_pm[i] = a.reserve((size_t)c.w4 * c.h4); }
  size_t pm_begin = o_pm[0], pm_end = a.reserve(0);
  for (size_t i = 0; i < nf; i++) { const RbtStreamCfg& c = b.frames[i].cfg; o_cs[i] = a.reserve((size_t)c.w_ctb * c.h_ctb * 2); }
  size_t cs_begin = o_cs[0], cs_end = a.reserve(0);
  for (size_t i = 0; i < nf; i++) {
    const RbtStreamCfg& c = b.frames[i].cfg; size_t u = (size_t)c.w4 * c.h4, nc = (size_t)c.w_ctb * c.h_ctb;
    o_pix[i] = a.reserve(frame_samples(c) * 2);
    o_out[i] = b.info[i].sao ? a.reserve(frame_samples(c) * 2) : o_pix[i];
    o_dm[i] = a.reserve(u); o_qp[i] = a.reserve(u); o_mv[i] = a.reserve(u * 4); o_ref[i] = a.reserve(u); o_refpoc[i] = a.reserve(u * 4);
    o_sao[i] = a.reserve(nc * sizeof(RbtSao)); o_cmds[i] = a.reserve(nc * (size_t)b.frames[i].cmd_cap * sizeof(RbtCmd));
  }
  std::vector<size_t> o_prow_line(nf, 0);
  const size_t prow_begin = a.reserve(0);     // the rows' hand-over records (one block, zeroed per job: recycled pool memory must not pass for the state of a row nobody parsed)
  for (size_t i = 0; i < nf; i++) if (wpp_frame[i]) {
    const RbtStreamCfg& c = b.frames[i].cfg;
    b.frames[i].prow_line_bytes = (int32_t)(((size_t)c.w4 * 7 + (size_t)c.w_ctb * (2 + sizeof(RbtSao)) + 255) & ~(size_t)255);
    o_prow_ctx[i] = a.reserve((size_t)c.h_ctb * 256 + 256); o_prow_line[i] = a.reserve((size_t)c.h_ctb * b.frames[i].prow_line_bytes + 256);
  }
  const size_t prow_end = a.reserve(0);
  size_t o_frames = a.reserve(nf * sizeof(RbtFrame)), o_slices = a.reserve(b.slices.size() * sizeof(RbtSlice));
  size_t o_rbsp = a.reserve(b.rbsp.size() + 64), o_lists = a.reserve((nf + b.slices.size()) * 2 * sizeof(int32_t));
  // CTB dependency order (anti-diagonals x + 2y ascending, top to bottom inside one) per distinct picture geometry, and the pictures of every level as RbtFrameRef
  b.order_keep.clear(); b.order_off.assign(nf, 0);
  { std::vector<std::pair<std::pair<int, int>, size_t>> seen;
    for (size_t i = 0; i < nf; i++) { const int w = b.frames[i].cfg.w_ctb, h = b.frames[i].cfg.h_ctb; size_t off = (size_t)-1;
      for (auto& s_ : seen) if (s_.first == std::make_pair(w, h)) off = s_.second;
      if (off == (size_t)-1) { off = b.order_keep.size(); seen.push_back({{w, h}, off});
        for (int d = 0; d <= w - 1 + 2 * (h - 1); d++) for (int y = 0; y < h; y++) { int x = d - 2 * y; if (x >= 0 && x < w) b.order_keep.push_back((uint32_t)x | ((uint32_t)y << 16)); } }
      b.order_off[i] = off; } }
  size_t o_order = a.reserve(b.order_keep.size() * 4), o_refs = a.reserve(nf * sizeof(RbtFrameRef));
  b.arena_size = a.reserve(0);
  b.arena = rbtk::dev_alloc(b.arena_size);
  if (!b.arena) { b.err = "device allocation failed"; return b.err_code = RBT_ERR_NOMEM; }
  uint8_t* base = (uint8_t*)b.arena;
  for (size_t i = 0; i < nf; i++) {
    RbtFrame& f = b.frames[i]; const RbtStreamCfg& c = f.cfg; size_t ys = (size_t)c.w * c.h, cs = (size_t)c.cw * c.ch;
    auto planes = [&](size_t off, uint16_t** p) { p[0] = (uint16_t*)(base + off); p[1] = p[0] + ys; p[2] = p[1] + cs; };
    planes(o_pix[i], f.pix); planes(o_out[i], f.out);
    f.coef[0] = (int16_t*)(base + o_coef[i]); f.coef[1] = f.coef[0] + ys; f.coef[2] = f.coef[1] + cs;
    f.pm = base + o_pm[i]; f.edges = base + o_edges[i]; f.dm = base + o_dm[i]; f.qp = (int8_t*)(base + o_qp[i]); f.mv = (int16_t*)(base + o_mv[i]);
    f.ref = (int8_t*)(base + o_ref[i]); f.refpoc = (int32_t*)(base + o_refpoc[i]); f.sao = (RbtSao*)(base + o_sao[i]); f.ctb_slice = (uint16_t*)(base + o_cs[i]);
    f.cmds = (RbtCmd*)(base + o_cmds[i]); f.cmd_count = (uint32_t*)(base + o_cnt[i]); f.ctb_done = (uint32_t*)(base + o_done[i]);
    if (wpp_frame[i]) { f.prow_done = (uint32_t*)(base + o_prow_done[i]); f.prow_ctx = (uint8_t*)(((uintptr_t)(base + o_prow_ctx[i]) + 255) & ~(uintptr_t)255);
                        f.prow_line = (uint8_t*)(((uintptr_t)(base + o_prow_line[i]) + 255) & ~(uintptr_t)255); }
  }
  b.d_order = (uint32_t*)(base + o_order); b.d_refs = (RbtFrameRef*)(base + o_refs); b.d_tickets = (uint32_t*)(base + o_tickets); b.d_queue = (uint32_t*)(base + o_queue);
  b.refs_keep.clear(); b.refs_off.clear();
  for (auto& lf : b.level_frames) { b.refs_off.push_back(b.refs_keep.size()); for (int fi : lf) b.refs_keep.push_back(RbtFrameRef{(RbtFrame*)(base + o_frames), (const RbtSlice*)(base + o_slices), b.d_order + b.order_off[fi], fi, 0}); }
  if (b.level_frames.size() > 32) { b.err = "too many dependency levels"; return b.err_code = RBT_ERR_UNSUPPORTED; }
  b.d_save = b.want_save ? (void*)(base + o_save) : nullptr;
  b.d_frames = (RbtFrame*)(base + o_frames); b.d_slices = (RbtSlice*)(base + o_slices); b.d_rbsp = base + o_rbsp; b.d_lists = (int32_t*)(base + o_lists);
  if (rbtk::dev_memset(base, 0, zero_end) || (prow_end > prow_begin && rbtk::dev_memset(base + prow_begin, 0, prow_end - prow_begin)) || rbtk::dev_memset(base + pm_begin, RBT_MODE_NONE, pm_end - pm_begin) || rbtk::dev_memset(base + cs_begin, 0xFF, cs_end - cs_begin) ||
      rbtk::h2d(b.d_frames, b.frames.data(), nf * sizeof(RbtFrame)) || rbtk::h2d(b.d_slices, b.slices.data(), b.slices.size() * sizeof(RbtSlice)) ||
      rbtk::h2d(b.d_rbsp, b.rbsp.data(), b.rbsp.size()) || rbtk::h2d(b.d_order, b.order_keep.data(), b.order_keep.size() * 4) ||
      rbtk::h2d(b.d_refs, b.refs_keep.data(), b.refs_keep.size() * sizeof(RbtFrameRef))) { b.err = "device transfer failed"; return b.err_code = RBT_ERR_NO_DEVICE; }
  return 0;
}

// Reconstruction of a dependency level: one launch with CTB-to-CTB done flags (launch_recon_level), or one launch per anti-diagonal.
// Measured on MI355X (HM-like 32-frame GOF): the flag kernel shortens the level itself by 13-19 % (blocking call 352.1 -> 349.3 ms, reconstruction 64 -> 52 ms per job),
// but with 16 GOFs in flight its resident waiting workgroups hold LDS and wave slots the other jobs' analysis / intra-coding kernels need (those go from 25-27 / 31 ms
// to 40 / 49 ms per job): 598 against 635 frames/s. So: flags while few jobs are in flight (<= 4: the GPU is mostly idle), diagonals beyond.
// RBT_RECON_DIAG=1 / RBT_RECON_LEVEL=1 force one or the other.
void recon_set_depth(int depth) { rbtk::set_jobs_in_flight(depth); }      // kept with the device (rbt_kernels.hip Dev): contexts on different devices do not share it
// Round 4: one launch per level with a device-side READY queue (launch_recon_queue): a few persistent workgroups - about as many as the level's wavefronts are wide -
// take CTBs in the order they become ready. RBT_RECON_QUEUE=1 / =0 forces it on / off.
int recon_mode() {
  static int force = -1;
  if (force < 0) { const char* d = getenv("RBT_RECON_DIAG"); const char* l = getenv("RBT_RECON_LEVEL"); const char* q = getenv("RBT_RECON_QUEUE");
                   force = d && atoi(d) ? 1 : (l && atoi(l) ? 2 : (q && *q ? (atoi(q) ? 3 : 4) : 0)); }
  if (force == 1) return 0;
  if (force == 2) return 1;
  if (force == 3) return 2;
  return rbtk::jobs_in_flight() > 4 ? 0 : 1;                     // (force 4: the queue off, the round-3 rule)
}
bool recon_by_diagonals() { return recon_mode() == 0; }
int recon_queue_width(const RbtStreamCfg& c) { const int w = c.w_ctb, h = c.h_ctb, d = w + 2 * h - 2; return d > 0 ? std::max(1, (w * h + d - 1) / d) : 1; }

int decode_run(DecodeBatch& b) { int rc = decode_launch(b); return rc ? rc : decode_finish(b); }

int decode_launch(DecodeBatch& b) {
  int rc = decode_launch_parse(b);
  if (rc) return rc;
  rbtk::timer_begin(T_RECON);
  for (size_t l = 0; l < b.level_frames.size(); l++) decode_launch_level(b, l);
  rbtk::timer_end(T_RECON);
  return 0;
}
static void build_lists(DecodeBatch& b) {
  std::vector<size_t>& sl_off = b.sl_off; std::vector<size_t>& sl_cnt = b.sl_cnt; sl_off.clear(); sl_cnt.clear();
  // index lists: slices grouped by level, frames grouped by level
  std::vector<int32_t>& lists = b.lists_keep; lists.clear(); b.fr_off.clear();
  for (auto& lf : b.level_frames) {
    sl_off.push_back(lists.size());
    for (int fi : lf) for (int k = 0; k < b.frames[fi].n_slices; k++) lists.push_back(b.frames[fi].first_slice + k);
    sl_cnt.push_back(lists.size() - sl_off.back());
  }
  for (auto& lf : b.level_frames) { b.fr_off.push_back(lists.size()); for (int fi : lf) lists.push_back(fi); }
}
static int max_w4(const DecodeBatch& b) { int m = 0; for (auto& f : b.frames) m = std::max(m, (int)f.cfg.w4); return m; }
int decode_max_w4(const DecodeBatch& b) { return max_w4(b); }
int decode_upload_lists(DecodeBatch& b) {
  if (b.lists_uploaded) return 0;
  b.lists_uploaded = true;
  build_lists(b);
  if (b.lists_keep.size() > (b.frames.size() + b.slices.size()) * 2) { b.err = "internal: list overflow"; return b.err_code = RBT_ERR_PARAM; }
  if (rbtk::h2d(b.d_lists, b.lists_keep.data(), b.lists_keep.size() * sizeof(int32_t))) { b.err = "device transfer failed"; return b.err_code = RBT_ERR_NO_DEVICE; }
  return 0;
}
int decode_launch_parse(DecodeBatch& b) {
  int rc = decode_upload_lists(b);
  if (rc) return rc;
  if (b.parse_external) return 0;
  const std::vector<size_t>& sl_off = b.sl_off; const std::vector<size_t>& sl_cnt = b.sl_cnt;
  rbtk::timer_begin(T_PARSE);
  uint32_t* tk = b.has_row_tasks ? b.d_tickets + 64 : nullptr;          // row tasks wait for earlier list entries: hand the list out in start order
  if (b.ordered_parse) { for (size_t l = 0; l < b.level_frames.size(); l++) rbtk::launch_parse(b.d_frames, b.d_slices, b.d_rbsp, b.d_lists + sl_off[l], (int)sl_cnt[l], max_w4(b), nullptr, 0, tk ? tk + l : nullptr); }
  else rbtk::launch_parse(b.d_frames, b.d_slices, b.d_rbsp, b.d_lists, (int)b.slices.size(), max_w4(b), nullptr, 0, tk);
  rbtk::timer_end(T_PARSE);
  return 0;
}
void decode_launch_filters(DecodeBatch& b, size_t l) {
  const std::vector<int>& lf = b.level_frames[l];
  int mu = 0, ml = 0, mc = 0;
  for (int fi : lf) { const RbtStreamCfg& c = b.frames[fi].cfg; mu = std::max(mu, c.w4 * c.h4); ml = std::max(ml, c.w * c.h); mc = std::max(mc, c.w_ctb * c.h_ctb); }
  // SAO: one workgroup per CTB (round 4: a third of the per-sample kernel's instructions); RBT_SAO_PER_SAMPLE=1 keeps the round-1 form
  static const int sao_per_sample = [] { const char* e = getenv("RBT_SAO_PER_SAMPLE"); return e && atoi(e) != 0; }();
  // Deblocking in place, one launch per edge direction, then SAO from `pix` to `out`. RBT_FUSED_LF=1 (round 3) gives pictures with SAO ONE launch through LDS tiles instead
  // (rbt_loopfilter_tile: one read of the reconstruction, one write of the output; same samples): 1-2 % faster for a lone GOF, 3-6 % slower with 16+ GOFs in flight, where the
  // kernels queue for LDS (the reconstruction holds a CTB there) and the LDS-free filter launches fill the gaps - measured in tools/lf_probe.sh, so it is off by default.
  // Runs of pictures with / without SAO inside the level's list get their own launches.
  static const int fused = [] { const char* e = getenv("RBT_FUSED_LF"); return e && atoi(e) != 0; }();
  size_t k = 0;
  while (k < lf.size()) {
    const bool sao = b.info[lf[k]].sao;
    size_t e = k; while (e < lf.size() && b.info[lf[e]].sao == sao) e++;
    int mw_ = 0, mh_ = 0; for (size_t q = k; q < e; q++) { mw_ = std::max(mw_, (int)b.frames[lf[q]].cfg.w); mh_ = std::max(mh_, (int)b.frames[lf[q]].cfg.h); }
    if (sao && fused) rbtk::launch_loopfilter(b.d_frames, b.d_slices, b.d_lists + b.fr_off[l] + k, (int)(e - k), mw_, mh_);
    else {
      rbtk::launch_deblock(b.d_frames, b.d_slices, b.d_lists + b.fr_off[l] + k, (int)(e - k), mu);
      if (sao && sao_per_sample) rbtk::launch_sao(b.d_frames, b.d_slices, b.d_lists + b.fr_off[l] + k, (int)(e - k), ml);
      else if (sao) rbtk::launch_sao_ctb(b.d_frames, b.d_slices, b.d_lists + b.fr_off[l] + k, (int)(e - k), mc);
    }
    k = e;
  }
}
int decode_launch_chunked(DecodeBatch& b, int chunks, int main_stream, int aux_stream) {
  int rc = decode_upload_lists(b);
  if (rc) return rc;
  const std::vector<int>& lf = b.level_frames[0];
  int mw = 0, mh = 0, max_h_all = 0;
  for (int fi : lf) { const RbtStreamCfg& c = b.frames[fi].cfg; mw = std::max(mw, c.w_ctb); mh = std::max(mh, c.h_ctb); }
  for (auto& f : b.frames) max_h_all = std::max(max_h_all, (int)f.cfg.h_ctb);
  rbtk::timer_begin(T_PARSE);
  int y_prev = 0;
  for (int c = 0; c < chunks; c++) {
    int y_lim = c + 1 == chunks ? (1 << 30) : (max_h_all * (c + 1) + chunks - 1) / chunks;
    rbtk::set_stream(main_stream);
    rbtk::launch_parse(b.d_frames, b.d_slices, b.d_rbsp, b.d_lists, (int)b.slices.size(), max_w4(b), b.d_save, y_lim);
    if (c + 1 == chunks) rbtk::timer_end(T_PARSE);
    rbtk::stream_wait(aux_stream, main_stream);
    rbtk::set_stream(aux_stream);
    rbtk::launch_recon(b.d_frames, b.d_slices, b.d_lists + b.fr_off[0], (int)lf.size(), mw, mh, y_prev, y_lim);
    y_prev = y_lim;
  }
  rbtk::set_stream(main_stream);
  rbtk::stream_wait(main_stream, aux_stream);
  decode_launch_filters(b, 0);
  return 0;
}
void decode_launch_level(DecodeBatch& b, size_t l) {
  const std::vector<int>& lf = b.level_frames[l];
  int mw = 0, mh = 0;
  for (int fi : lf) { const RbtStreamCfg& c = b.frames[fi].cfg; mw = std::max(mw, c.w_ctb); mh = std::max(mh, c.h_ctb); }
  const int mode = (lf.size() >= 8192 || (size_t)mw * mh >= ((size_t)1 << 18)) && recon_mode() == 2 ? 0 : recon_mode();      // the queue packs (picture, CTB) into 31 bits
  if (mode == 0) rbtk::launch_recon(b.d_frames, b.d_slices, b.d_lists + b.fr_off[l], (int)lf.size(), mw, mh);
  else if (mode == 1) rbtk::launch_recon_level(b.d_refs + b.refs_off[l], (int)lf.size(), mw * mh, b.d_tickets + l);
  else rbtk::launch_recon_queue(b.d_refs + b.refs_off[l], (int)lf.size(), b.queue_total[l], b.d_queue + b.queue_off[l], b.queue_wgs[l]);
  decode_launch_filters(b, l);
}

int decode_finish(DecodeBatch& b) {
  size_t nf = b.frames.size();
  if (rbtk::dev_sync()) { b.err = "kernel execution failed"; return b.err_code = RBT_ERR_NO_DEVICE; }
  // per-picture error words and slice coverage
  std::vector<RbtFrame> fr(nf);
  if (rbtk::d2h(fr.data(), b.d_frames, nf * sizeof(RbtFrame))) { b.err = "device transfer failed"; return b.err_code = RBT_ERR_NO_DEVICE; }
  for (size_t i = 0; i < nf; i++) if (fr[i].error) { b.err = "slice data decoding failed (code " + std::to_string(fr[i].error) + ")"; return b.err_code = RBT_ERR_BITSTREAM; }
  // every CTB of every picture must have been decoded by exactly one slice segment (a truncated stream leaves pictures with holes)
  std::vector<RbtSlice> sl(b.slices.size());
  if (rbtk::d2h(sl.data(), b.d_slices, sl.size() * sizeof(RbtSlice))) { b.err = "device transfer failed"; return b.err_code = RBT_ERR_NO_DEVICE; }
  std::vector<uint64_t> covered(nf, 0);
  for (auto& s_ : sl) covered[(size_t)s_.frame] += s_.n_ctbs_decoded;
  for (size_t i = 0; i < nf; i++) if (covered[i] != (uint64_t)b.frames[i].cfg.w_ctb * b.frames[i].cfg.h_ctb) { b.err = "picture " + std::to_string(i) + " is not completely covered by slice data"; return b.err_code = RBT_ERR_BITSTREAM; }
  return 0;
}

int decode_fetch(DecodeBatch& b, int stream, rbt_video* out, bool verify_md5) {
  memset(out, 0, sizeof(*out));
  int first = b.stream_first[stream], n = b.stream_count[stream];
  if (n <= 0) return RBT_ERR_BITSTREAM;
  const RbtStreamCfg& c = b.frames[first].cfg; const Sps& sps = b.stream_sps[stream];
  // output = coded picture minus the conformance window (7.4.3.2.1); the picture hash covers the whole coded picture
  const int cl = 2 * sps.conf_win[0], ct = 2 * sps.conf_win[2], dw = c.w - cl - 2 * sps.conf_win[1], dh = c.h - ct - 2 * sps.conf_win[3];
  if (dw <= 0 || dh <= 0) return RBT_ERR_BITSTREAM;
  const bool crop = dw != c.w || dh != c.h;
  size_t fs = frame_samples(c), ofs = (size_t)dw * dh + 2 * (size_t)(dw / 2) * (dh / 2);
  out->width = dw; out->height = dh; out->bit_depth = c.bit_depth; out->n_frames = n;
  out->data = (uint16_t*)malloc(ofs * 2 * (size_t)n);
  if (!out->data) return RBT_ERR_NOMEM;
  // with a conformance window the whole coded pictures are kept next to the cropped output while their hashes are checked (all planes side by side, md5_planes_u16)
  const bool keep = crop && verify_md5;
  std::vector<uint16_t> full(crop ? (keep ? fs * (size_t)n : fs) : 0);
  std::vector<Md5PlaneJob> jobs; std::vector<uint8_t> hashes((size_t)n * 48);
  for (int i = 0; i < n; i++) {
    const RbtFrame& f = b.frames[first + i];
    uint16_t* dst = out->data + ofs * (size_t)i;
    uint16_t* p = crop ? full.data() + (keep ? fs * (size_t)i : 0) : dst;
    if (rbtk::d2h(p, f.out[0], fs * 2)) return RBT_ERR_NO_DEVICE;
    if (verify_md5 && b.info[first + i].has_md5) {
      jobs.push_back({p, c.w, c.h, c.bit_depth, &hashes[(size_t)i * 48]});
      jobs.push_back({p + (size_t)c.w * c.h, c.cw, c.ch, c.bit_depth, &hashes[(size_t)i * 48 + 16]});
      jobs.push_back({p + (size_t)c.w * c.h + (size_t)c.cw * c.ch, c.cw, c.ch, c.bit_depth, &hashes[(size_t)i * 48 + 32]});
    }
    if (crop) {
      const uint16_t* full_i = p;
      const uint16_t* src = full_i; uint16_t* d = dst;
      for (int k = 0; k < 3; k++) { const int sh = k ? 1 : 0, pw = c.w >> sh, ph = c.h >> sh, ow = dw >> sh, oh = dh >> sh;
        for (int y = 0; y < oh; y++) memcpy(d + (size_t)y * ow, src + (size_t)(y + (ct >> sh)) * pw + (cl >> sh), (size_t)ow * 2);
        src += (size_t)pw * ph; d += (size_t)ow * oh; }
    }
  }
  md5_planes_u16(jobs.data(), jobs.size());
  for (int i = 0; i < n; i++) if (verify_md5 && b.info[first + i].has_md5) {
    out->md5_checked++;
    if (memcmp(&hashes[(size_t)i * 48], b.info[first + i].md5[0], 16) || memcmp(&hashes[(size_t)i * 48 + 16], b.info[first + i].md5[1], 16) || memcmp(&hashes[(size_t)i * 48 + 32], b.info[first + i].md5[2], 16)) out->md5_failed++;
  }
  return 0;
}

}  // namespace rbt
