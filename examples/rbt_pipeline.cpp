// Host-side use of the C ABI from C++ (the reference's language), as INTEGRATION.md describes it for the application's GOF loop
// (PccAppTranscoder.cpp:307-341): GOFs are handed to rbt_submit_gof ahead of rbt_wait_gof, `depth` of them in flight.
//
//   rbt_pipeline <in.gofs> <out.gofs> [depth] [geometryQP] [attributeQP]
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

int main(int argc, char** argv) {
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
