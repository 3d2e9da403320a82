// Multi-GPU host of the transcoder in C++ (the reference's language) over the C ABI: one process per GPU, GOFs sharded by the library's own rule, the re-encoded NAL units
// gathered on rank 0 with RCCL over xGMI (north star: "host side in C++ calling HIP through a thin C-ABI ... RCCL over xGMI gathering the re-encoded NAL units").
//
//   rbt_multi_gpu --ranks N <in.bin> <out.bin> [depth] [geometryQP] [attributeQP] [occupancyPrecision] [occupancyRd]
//
// in.bin / out.bin: V3C sample streams (what PccAppTranscoder reads and writes, PccAppTranscoder.cpp:289, :345-348).
//
// The parent never touches a GPU: it starts N fresh child processes (fork + exec of this binary with --child r) and waits for them. Child r:
//   rbt_create(device r, rank r, world N)      the context knows which GOFs it owns (rbt_owns_gof: GOF g -> rank g mod N)
//   rbt_transcode_v3c                          the whole file in, a sample stream with ITS GOFs out (the walk of PccAppTranscoder.cpp:277-349, several GOFs in flight)
//   ncclAllGather of the part sizes            (8 bytes per rank)
//   ncclGroupStart { ncclSend | ncclRecv }     every rank sends its part to rank 0, which posts one receive per rank: RCCL has no gatherv; <= a few MB per GOF, so the xGMI
//   ncclGroupEnd                               links (7 x ~153 GB/s per GPU) are idle in comparison - one exchange per file, never one per frame
//   rank 0: rbt_v3c_index on every part, units interleaved back into GOF order, rbt_v3c_write once over all units (PCCBitstreamWriter::write is called once, :343-348)
// The RCCL unique id travels from rank 0 to the others through a file in the parent's scratch directory (no MPI, no launcher).
//
// Build (rabbit-transcoding_amd/Makefile, target rbt_multi_gpu): hipcc -std=c++17 -O2 -I include examples/rbt_multi_gpu.cpp -L rabbit-transcoding_amd -lrbt -lrccl
// Not run on more than one GPU in this project's environment (one-GPU boxes only): --ranks 1 is tested against rbt_transcode_v3c (tests/test_cpp_host.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "rbt.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(e_)); return 1; } } while (0)
static int g_rank = 0;

static bool read_file(const char* path, std::vector<uint8_t>& v) {
  FILE* f = fopen(path, "rb"); if (!f) return false;
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); v.resize(sz > 0 ? (size_t)sz : 0);
  bool ok = sz <= 0 || fread(v.data(), 1, (size_t)sz, f) == (size_t)sz;
  fclose(f); return ok;
}

// the parts of all ranks (part r holds the GOFs r, r + world, ... in order) -> the stream one rank would have written
static int merge_parts(const std::vector<std::vector<uint8_t>>& parts, int forced_precision, uint8_t** out, size_t* n_out) {
  const int world = (int)parts.size();
  std::vector<std::vector<std::vector<std::pair<const uint8_t*, size_t>>>> per(world);     // per rank, per local GOF: its units
  size_t n_gofs = 0;
  for (int r = 0; r < world; r++) {
    rbt_v3c_unit* u = nullptr; int nu = 0;
    int rc = rbt_v3c_index(parts[r].data(), parts[r].size(), &u, &nu);
    if (rc != RBT_OK) return rc;
    for (int k = 0; k < nu; k++) { if ((size_t)u[k].gof >= per[r].size()) per[r].resize((size_t)u[k].gof + 1); per[r][u[k].gof].push_back({parts[r].data() + u[k].offset, u[k].size}); }
    rbt_free(u);
    n_gofs += per[r].size();
  }
  std::vector<const uint8_t*> up; std::vector<size_t> us;
  for (size_t g = 0; g < n_gofs; g++) {
    const size_t r = g % world, k = g / world;
    if (k >= per[r].size()) return RBT_ERR_BITSTREAM;                                    // not a round-robin split of one stream
    for (auto& q : per[r][k]) { up.push_back(q.first); us.push_back(q.second); }
  }
  return rbt_v3c_write(up.data(), us.data(), (int)up.size(), forced_precision, out, n_out);
}

static int child_main(int rank, int world, const char* id_path, int argc, char** argv) {
  g_rank = rank;
  const char* in_path = argv[0]; const char* out_path = argv[1];
  const int depth = argc > 2 ? atoi(argv[2]) : 8;
  rbt_v3c_params vp; memset(&vp, 0, sizeof(vp));
  vp.geometry_qp = argc > 3 ? atoi(argv[3]) : 24; vp.attribute_qp = argc > 4 ? atoi(argv[4]) : 32; vp.occupancy_precision = argc > 5 ? atoi(argv[5]) : 4;
  vp.occupancy_rd = argc > 6 ? atoi(argv[6]) : 0; vp.ctb_rows_per_slice = -1; vp.gofs_per_job = 0;
  std::vector<uint8_t> in;
  if (!read_file(in_path, in)) { fprintf(stderr, "rank %d: cannot read %s\n", rank, in_path); return 2; }
  const auto t0 = std::chrono::steady_clock::now();
  // ---- this rank's GOFs through the library
  rbt_ctx* ctx = nullptr;
  int rc = rbt_create(&ctx, rank, rank, world);
  if (rc != RBT_OK) { fprintf(stderr, "rank %d: rbt_create: %s\n", rank, rbt_strerror(rc)); return 1; }   // no GPU: there is no CPU path
  uint8_t* part = nullptr; size_t n_part = 0;
  if ((rc = rbt_set_depth(ctx, depth)) == RBT_OK) rc = rbt_transcode_v3c(ctx, in.data(), in.size(), &vp, &part, &n_part);
  if (rc != RBT_OK) { fprintf(stderr, "rank %d: rbt_transcode_v3c: %s %s\n", rank, rbt_strerror(rc), rbt_last_error(ctx)); rbt_destroy(ctx); return 1; }
  rbt_destroy(ctx);
  const auto t1 = std::chrono::steady_clock::now();
  // ---- gather on rank 0 over RCCL
  CHECK_HIP(hipSetDevice(rank));
  ncclUniqueId id;
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(&id));
    const std::string tmp = std::string(id_path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb"); if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank 0: cannot write %s\n", tmp.c_str()); return 1; }
    fclose(f); rename(tmp.c_str(), id_path);                                            // appears complete or not at all
  } else {
    FILE* f = nullptr;
    for (int tries = 0; tries < 6000 && !(f = fopen(id_path, "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    if (!f || fread(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank %d: no RCCL id from rank 0\n", rank); return 1; }
    fclose(f);
  }
  ncclComm_t comm; hipStream_t stream;
  CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
  CHECK_HIP(hipStreamCreate(&stream));
  unsigned long long my_size = n_part, *d_sizes = nullptr, *d_mine = nullptr;
  CHECK_HIP(hipMalloc(&d_sizes, 8 * (size_t)world)); CHECK_HIP(hipMalloc(&d_mine, 8));
  CHECK_HIP(hipMemcpyAsync(d_mine, &my_size, 8, hipMemcpyHostToDevice, stream));
  CHECK_NCCL(ncclAllGather(d_mine, d_sizes, 1, ncclUint64, comm, stream));
  std::vector<unsigned long long> sizes(world);
  CHECK_HIP(hipMemcpyAsync(sizes.data(), d_sizes, 8 * (size_t)world, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  uint8_t* d_part = nullptr; CHECK_HIP(hipMalloc(&d_part, n_part ? n_part : 1));
  CHECK_HIP(hipMemcpyAsync(d_part, part, n_part, hipMemcpyHostToDevice, stream));
  std::vector<uint8_t*> d_recv(world, nullptr);
  if (rank == 0) for (int r = 1; r < world; r++) CHECK_HIP(hipMalloc(&d_recv[r], sizes[r] ? sizes[r] : 1));
  CHECK_NCCL(ncclGroupStart());
  if (rank != 0) CHECK_NCCL(ncclSend(d_part, n_part, ncclUint8, 0, comm, stream));
  else for (int r = 1; r < world; r++) CHECK_NCCL(ncclRecv(d_recv[r], sizes[r], ncclUint8, r, comm, stream));
  CHECK_NCCL(ncclGroupEnd());
  CHECK_HIP(hipStreamSynchronize(stream));
  int ret = 0;
  if (rank == 0) {
    std::vector<std::vector<uint8_t>> parts(world);
    parts[0].assign(part, part + n_part);
    for (int r = 1; r < world; r++) { parts[r].resize(sizes[r]); CHECK_HIP(hipMemcpy(parts[r].data(), d_recv[r], sizes[r], hipMemcpyDeviceToHost)); CHECK_HIP(hipFree(d_recv[r])); }
    uint8_t* out = nullptr; size_t n = 0;
    rc = merge_parts(parts, vp.forced_unit_size_precision_bytes, &out, &n);
    if (rc != RBT_OK) { fprintf(stderr, "merge: %s\n", rbt_strerror(rc)); ret = 1; }
    else {
      FILE* f = fopen(out_path, "wb");
      if (!f || fwrite(out, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", out_path); ret = 2; }
      if (f) fclose(f);
      const auto t2 = std::chrono::steady_clock::now();
      rbt_v3c_stat a, b;
      if (rbt_v3c_stats(in.data(), in.size(), &a) == RBT_OK && rbt_v3c_stats(out, n, &b) == RBT_OK)
        printf("%d ranks, %d GOFs: %llu -> %llu bytes; transcode %.1f ms (rank 0), gather + merge + write %.1f ms\n", world, a.n_gofs, (unsigned long long)a.total, (unsigned long long)b.total,
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
      rbt_free(out);
    }
  }
  rbt_free(part);
  (void)hipFree(d_part); (void)hipFree(d_sizes); (void)hipFree(d_mine); (void)hipStreamDestroy(stream);
  ncclCommDestroy(comm);
  return ret;
}

int main(int argc, char** argv) {
  if (argc >= 6 && !strcmp(argv[1], "--child")) return child_main(atoi(argv[2]), atoi(argv[3]), argv[4], argc - 5, argv + 5);
  if (argc < 5 || strcmp(argv[1], "--ranks")) { fprintf(stderr, "usage: %s --ranks N in.bin out.bin [depth] [geometryQP] [attributeQP] [occupancyPrecision] [occupancyRd]\n", argv[0]); return 2; }
  const int world = atoi(argv[2]);
  if (world < 1 || world > 64) { fprintf(stderr, "--ranks 1..64\n"); return 2; }
  // the parent does nothing that initialises a GPU: every rank is a fresh process
  char dir[] = "/tmp/rbt_multi_gpu_XXXXXX";
  if (!mkdtemp(dir)) { perror("mkdtemp"); return 2; }
  const std::string id_path = std::string(dir) + "/rccl_id";
  std::vector<pid_t> pids;
  for (int r = 0; r < world; r++) {
    const pid_t pid = fork();
    if (pid < 0) { perror("fork"); return 2; }
    if (pid == 0) {
      std::vector<std::string> a = {argv[0], "--child", std::to_string(r), std::to_string(world), id_path};
      for (int i = 3; i < argc; i++) a.push_back(argv[i]);
      std::vector<char*> av; for (auto& s : a) av.push_back((char*)s.c_str()); av.push_back(nullptr);
      execv(argv[0], av.data());
      perror("execv"); _exit(127);
    }
    pids.push_back(pid);
  }
  int rc = 0;
  for (pid_t p : pids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
  unlink(id_path.c_str()); rmdir(dir);
  return rc;
}
