// Host-side use of the C ABI from C++ (the reference's language), as INTEGRATION.md describes it for the application's GOF loop
// (PccAppTranscoder.cpp:307-341): GOFs are handed to rbt_submit_gof ahead of rbt_wait_gof, `depth` of them in flight.
//
//   rbt_pipeline <in.gofs> <out.gofs> [depth] [geometryQP] [attributeQP]
//   rbt_pipeline --v3c <in.bin> <out.bin> [depth] [geometryQP] [attributeQP] [occupancyPrecision]
//       the file-level form: a V3C sample stream in, a V3C sample stream out (rbt_transcode_v3c = the loop of PccAppTranscoder.cpp:277-349), then the
//       PCCBitstreamStat-style totals of both files (rbt_v3c_stats)
//
// File format (little endian, test harness only): u32 n_gofs, then per GOF three sub-bitstreams in the order occupancy, geometry,
// attribute, each as u32 size + Annex-B bytes. The output file has the same layout with the re-encoded streams.
// Build: g++ -std=c++17 -O2 -I include examples/rbt_pipeline.cpp -L rabbit-transcoding_amd -lrbt -Wl,-rpath,<dir> -o rbt_pipeline
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>
#include "rbt.h"

struct Gof { std::vector<uint8_t> s[3]; };

static bool read_all(const char* path, std::vector<Gof>& gofs) {
  FILE* f = fopen(path, "rb"); if (!f) return false;
  uint32_t n = 0; if (fread(&n, 4, 1, f) != 1) { fclose(f); return false; }
  gofs.resize(n);
  for (auto& g : gofs) for (int k = 0; k < 3; k++) {
    uint32_t sz = 0; if (fread(&sz, 4, 1, f) != 1) { fclose(f); return false; }
    g.s[k].resize(sz); if (sz && fread(g.s[k].data(), 1, sz, f) != sz) { fclose(f); return false; }
  }
  fclose(f); return true;
}

static int v3c_main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s --v3c in.bin out.bin [depth] [geometryQP] [attributeQP] [occupancyPrecision]\n", argv[0]); return 2; }
  const int depth = argc > 4 ? atoi(argv[4]) : 8;
  rbt_v3c_params vp; memset(&vp, 0, sizeof(vp));
  vp.geometry_qp = argc > 5 ? atoi(argv[5]) : 24; vp.attribute_qp = argc > 6 ? atoi(argv[6]) : 32; vp.occupancy_precision = argc > 7 ? atoi(argv[7]) : 4;
  vp.ctb_rows_per_slice = -1; vp.gofs_per_job = 0;   // job shape by the length of the walk (rbt_job_shape), `depth` is the cap
  std::vector<uint8_t> in;
  FILE* f = fopen(argv[2], "rb"); if (!f) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); in.resize(sz > 0 ? (size_t)sz : 0);
  if (sz > 0 && fread(in.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 2; }
  fclose(f);
  rbt_ctx* ctx = nullptr;
  int rc = rbt_create(&ctx, 0, 0, 1);
  if (rc != RBT_OK) { fprintf(stderr, "rbt_create: %s\n", rbt_strerror(rc)); return 1; }
  uint8_t* out = nullptr; size_t n = 0;
  if ((rc = rbt_set_depth(ctx, depth)) == RBT_OK) rc = rbt_transcode_v3c(ctx, in.data(), in.size(), &vp, &out, &n);
  if (rc != RBT_OK) { fprintf(stderr, "rbt_transcode_v3c: %s %s\n", rbt_strerror(rc), rbt_last_error(ctx)); rbt_destroy(ctx); return 1; }
  rbt_destroy(ctx);
  f = fopen(argv[3], "wb"); if (!f) return 2;
  fwrite(out, 1, n, f); fclose(f);
  rbt_v3c_stat a, b;
  if (rbt_v3c_stats(in.data(), in.size(), &a) == RBT_OK && rbt_v3c_stats(out, n, &b) == RBT_OK)
    printf("%d GOFs, %d units; metadata %llu -> %llu B, geometry %llu -> %llu B, attribute %llu -> %llu B, total %llu -> %llu B\n", a.n_gofs, a.n_units,
           (unsigned long long)a.total_metadata, (unsigned long long)b.total_metadata, (unsigned long long)a.total_geometry, (unsigned long long)b.total_geometry,
           (unsigned long long)a.total_attribute, (unsigned long long)b.total_attribute, (unsigned long long)a.total, (unsigned long long)b.total);
  rbt_free(out);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--v3c")) return v3c_main(argc, argv);
  if (argc < 3) { fprintf(stderr, "usage: %s in.gofs out.gofs [depth] [geometryQP] [attributeQP]\n", argv[0]); return 2; }
  const int depth = argc > 3 ? atoi(argv[3]) : 8, geo_qp = argc > 4 ? atoi(argv[4]) : 24, att_qp = argc > 5 ? atoi(argv[5]) : 32;
  std::vector<Gof> in; if (!read_all(argv[1], in)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  rbt_ctx* ctx = nullptr;
  int rc = rbt_create(&ctx, 0, 0, 1);
  if (rc != RBT_OK) { fprintf(stderr, "rbt_create: %s\n", rbt_strerror(rc)); return 1; }   // no GPU: there is no CPU path
  if ((rc = rbt_set_depth(ctx, depth)) != RBT_OK) { fprintf(stderr, "rbt_set_depth: %s\n", rbt_strerror(rc)); return 1; }
  // PCCTranscoderParameters -> rbt_stream_params (R3 by default: geometryQP 24, attributeQP 32, occupancyPrecision 4)
  rbt_stream_params p[3] = {{RBT_VIDEO_OCCUPANCY, 8, 4, 0, -1, 1, 0}, {RBT_VIDEO_GEOMETRY, geo_qp, 4, 0, -1, 1, 0}, {RBT_VIDEO_ATTRIBUTE, att_qp, 4, 0, -1, 1, 0}};   // ctb_rows_per_slice -1: wavefront rows
  std::vector<Gof> out(in.size());
  std::deque<std::pair<rbt_job*, size_t>> inflight;
  auto collect = [&]() -> int {
    rbt_job* job = inflight.front().first; size_t g = inflight.front().second; inflight.pop_front();
    uint8_t* o[3]; size_t n[3];
    int r = rbt_wait_gof(ctx, job, o, n);
    if (r != RBT_OK) { fprintf(stderr, "GOF %zu: %s\n", g, rbt_strerror(r)); return r; }
    for (int k = 0; k < 3; k++) { out[g].s[k].assign(o[k], o[k] + n[k]); rbt_free(o[k]); }
    return RBT_OK;
  };
  for (size_t g = 0; g < in.size() && rc == RBT_OK; g++) {
    if ((int)inflight.size() == depth) rc = collect();
    if (rc != RBT_OK) break;
    const uint8_t* ptr[3] = {in[g].s[0].data(), in[g].s[1].data(), in[g].s[2].data()};
    size_t len[3] = {in[g].s[0].size(), in[g].s[1].size(), in[g].s[2].size()};
    rbt_job* job = nullptr;
    rc = rbt_submit_gof(ctx, 3, ptr, len, p, &job);
    if (rc != RBT_OK) { fprintf(stderr, "submit GOF %zu: %s\n", g, rbt_strerror(rc)); break; }
    inflight.push_back({job, g});
  }
  while (!inflight.empty()) { int r = collect(); if (rc == RBT_OK) rc = r; }
  rbt_destroy(ctx);
  if (rc != RBT_OK) return 1;
  FILE* f = fopen(argv[2], "wb"); if (!f) return 2;
  uint32_t n = (uint32_t)out.size(); fwrite(&n, 4, 1, f);
  for (auto& g : out) for (int k = 0; k < 3; k++) { uint32_t sz = (uint32_t)g.s[k].size(); fwrite(&sz, 4, 1, f); fwrite(g.s[k].data(), 1, sz, f); }
  fclose(f);
  printf("%zu GOFs transcoded, %d in flight\n", out.size(), depth);
  return 0;
}
