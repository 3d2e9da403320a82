// Multi-GPU host of the transcoder in C++ (the reference's language) over the C ABI: one process per GPU, GOFs sharded by the library's own rule, the re-encoded NAL units
// gathered on rank 0 with RCCL over xGMI (north star: "host side in C++ calling HIP through a thin C-ABI ... RCCL over xGMI gathering the re-encoded NAL units").
//
//   rbt_multi_gpu --ranks N [--timeout S] <in.bin> <out.bin> [depth] [geometryQP] [attributeQP] [occupancyPrecision] [occupancyRd]
//
// in.bin / out.bin: V3C sample streams (what PccAppTranscoder reads and writes, PccAppTranscoder.cpp:289, :345-348).
//
// The parent never touches a GPU: it starts N fresh child processes (fork + exec of this binary with --child r) and supervises them. Child r:
//   rbt_create(device r, rank r, world N)      the context knows which GOFs it owns (rbt_owns_gof: GOF g -> rank g mod N)
//   rbt_transcode_v3c                          the whole file in, a sample stream with ITS GOFs out (the walk of PccAppTranscoder.cpp:277-349, several GOFs in flight)
//   status file                                "ok" or "fail <why>", published in the scratch directory BEFORE anything of RCCL: a rank enters the communicator only once
//                                              EVERY rank has said ok - one failed rank (no device, damaged input) makes all ranks exit non-zero without a collective to hang in
//   ncclAllGather of the part sizes            (8 bytes per rank)
//   ncclGroupStart { ncclSend | ncclRecv }     every rank sends its part to rank 0, which posts one receive per rank: RCCL has no gatherv; <= a few MB per GOF, so the xGMI
//   ncclGroupEnd                               links (7 x ~153 GB/s per GPU) are idle in comparison - one exchange per file, never one per frame. No return between the two.
//   rank 0: rbt_v3c_index on every part, units interleaved back into GOF order, rbt_v3c_write once over all units (PCCBitstreamWriter::write is called once, :343-348);
//           the output file appears complete or not at all (PccAppTranscoder.cpp:343-348: rank 0 writes, or nothing does)
// The RCCL unique id travels from rank 0 to the others through a file in the same directory (no MPI, no launcher); rank 0 makes it after the status round, so the
// ranks reach ncclCommInitRank together however long their transcodes took.
// The parent reaps children as they end (waitpid(-1)); on the first failure, or when --timeout (default 900 s) runs out, it raises the abort flag the children poll
// while they wait, gives them a moment and kills the rest: a rank blocked inside a collective whose peer died does not keep the job alive.
//
// Profiling: run the profiler on a RANK, never on the --ranks launcher (the launcher execs its children; a profiler's preloaded library would initialise the GPU
// in it first):  rocprofv3 --kernel-trace --stats -d out -- rbt_multi_gpu --child 0 1 /tmp/scratch_dir in.bin out.bin
//
// Build (rabbit-transcoding_amd/Makefile, target rbt_multi_gpu): hipcc -std=c++17 -O2 -I include examples/rbt_multi_gpu.cpp -L rabbit-transcoding_amd -lrbt -lrccl
// Tested: one rank on the GPU box against the oracle, a damaged input with one rank (tests/test_cpp_host.py, -m gpu); the status round, the abort flag and the
// parent's deadline without a device (--selftest, same file). More than one rank on real GPUs has not run in this project's environment (one-GPU boxes).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/stat.h>
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

static int g_rank = 0;
using clk = std::chrono::steady_clock;

static bool read_file(const char* path, std::vector<uint8_t>& v) {
  FILE* f = fopen(path, "rb"); if (!f) return false;
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); v.resize(sz > 0 ? (size_t)sz : 0);
  bool ok = sz <= 0 || fread(v.data(), 1, (size_t)sz, f) == (size_t)sz;
  fclose(f); return ok;
}
// a file that appears complete or not at all
static bool publish(const std::string& path, const void* p, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb"); if (!f) return false;
  const bool ok = fwrite(p, 1, n, f) == n;
  return (fclose(f) == 0) && ok && rename(tmp.c_str(), path.c_str()) == 0;
}
static bool exists(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0; }
static std::string slurp(const std::string& path) { std::vector<uint8_t> v; return read_file(path.c_str(), v) ? std::string(v.begin(), v.end()) : std::string(); }

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

// The status round. Every rank says how its transcode went; nobody goes on to RCCL unless all said ok. Returns 0 when all ranks are ok, 1 otherwise (a peer
// failed, the parent raised the abort flag, or the deadline passed: the reason is printed).
static int status_round(const std::string& dir, int rank, int world, bool ok, const std::string& why, double timeout_s) {
  const std::string mine = ok ? "ok" : "fail " + why;
  if (!publish(dir + "/status." + std::to_string(rank), mine.data(), mine.size())) { fprintf(stderr, "rank %d: cannot publish its status in %s\n", rank, dir.c_str()); return 1; }
  if (!ok) return 1;
  const auto deadline = clk::now() + std::chrono::duration<double>(timeout_s);
  for (int r = 0; r < world; r++) {
    const std::string p = dir + "/status." + std::to_string(r);
    while (!exists(p)) {
      if (exists(dir + "/abort")) { fprintf(stderr, "rank %d: job aborted (%s)\n", rank, slurp(dir + "/abort").c_str()); return 1; }
      if (clk::now() > deadline) { fprintf(stderr, "rank %d: no status from rank %d within %.0f s\n", rank, r, timeout_s); return 1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    const std::string s = slurp(p);
    if (s != "ok") { fprintf(stderr, "rank %d: rank %d failed (%s): leaving without the gather\n", rank, r, s.c_str()); return 1; }
  }
  return 0;
}

struct Gather {      // device buffers and handles of the RCCL phase: released on every way out
  ncclComm_t comm = nullptr; hipStream_t stream = nullptr; unsigned long long *d_sizes = nullptr, *d_mine = nullptr; uint8_t* d_part = nullptr; std::vector<uint8_t*> d_recv;
  ~Gather() {
    for (uint8_t* p : d_recv) if (p) (void)hipFree(p);
    if (d_part) (void)hipFree(d_part); if (d_sizes) (void)hipFree(d_sizes); if (d_mine) (void)hipFree(d_mine);
    if (stream) (void)hipStreamDestroy(stream);
    if (comm) ncclCommDestroy(comm);
  }
};
#define TRY_HIP(x) do { if (!err) { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); err = 1; } } } while (0)
#define TRY_NCCL(x) do { if (!err) { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(e_)); err = 1; } } } while (0)

// parts of all ranks onto rank 0 (parts[r], rank 0 only). Every step is skipped once something failed (err), except that a group that was started is always ended.
static int gather_parts(const std::string& dir, int rank, int world, const uint8_t* part, size_t n_part, double timeout_s, std::vector<std::vector<uint8_t>>& parts) {
  int err = 0; Gather g; g.d_recv.assign(world, nullptr);
  TRY_HIP(hipSetDevice(rank));
  ncclUniqueId id; const std::string id_path = dir + "/rccl_id";
  if (rank == 0) { TRY_NCCL(ncclGetUniqueId(&id)); if (!err && !publish(id_path, &id, sizeof(id))) { fprintf(stderr, "rank 0: cannot write %s\n", id_path.c_str()); err = 1; } }
  else {
    const auto deadline = clk::now() + std::chrono::duration<double>(timeout_s);
    while (!err && !exists(id_path)) {
      if (exists(dir + "/abort") || clk::now() > deadline) { fprintf(stderr, "rank %d: no RCCL id from rank 0\n", rank); err = 1; }
      else std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (!err) { const std::string s = slurp(id_path); if (s.size() != sizeof(id)) { fprintf(stderr, "rank %d: RCCL id file of %zu bytes\n", rank, s.size()); err = 1; } else memcpy(&id, s.data(), sizeof(id)); }
  }
  if (err) return 1;
  TRY_NCCL(ncclCommInitRank(&g.comm, world, id, rank));
  TRY_HIP(hipStreamCreate(&g.stream));
  unsigned long long my_size = n_part;
  TRY_HIP(hipMalloc(&g.d_sizes, 8 * (size_t)world)); TRY_HIP(hipMalloc(&g.d_mine, 8));
  TRY_HIP(hipMemcpyAsync(g.d_mine, &my_size, 8, hipMemcpyHostToDevice, g.stream));
  TRY_NCCL(ncclAllGather(g.d_mine, g.d_sizes, 1, ncclUint64, g.comm, g.stream));
  std::vector<unsigned long long> sizes(world, 0);
  TRY_HIP(hipMemcpyAsync(sizes.data(), g.d_sizes, 8 * (size_t)world, hipMemcpyDeviceToHost, g.stream));
  TRY_HIP(hipStreamSynchronize(g.stream));
  TRY_HIP(hipMalloc(&g.d_part, n_part ? n_part : 1));
  TRY_HIP(hipMemcpyAsync(g.d_part, part, n_part, hipMemcpyHostToDevice, g.stream));
  if (rank == 0) for (int r = 1; r < world; r++) TRY_HIP(hipMalloc(&g.d_recv[r], sizes[r] ? sizes[r] : 1));
  if (err) return 1;                                                                     // nothing of the exchange has been posted yet: the peers learn of it from the parent (abort flag, kill)
  {
    ncclResult_t gs = ncclGroupStart();
    if (gs != ncclSuccess) { fprintf(stderr, "rank %d: ncclGroupStart: %s\n", rank, ncclGetErrorString(gs)); return 1; }
    if (rank != 0) TRY_NCCL(ncclSend(g.d_part, n_part, ncclUint8, 0, g.comm, g.stream));
    else for (int r = 1; r < world; r++) TRY_NCCL(ncclRecv(g.d_recv[r], sizes[r], ncclUint8, r, g.comm, g.stream));
    ncclResult_t ge = ncclGroupEnd();                                                   // always: a group left open would leave the peers' matching calls pending
    if (ge != ncclSuccess) { fprintf(stderr, "rank %d: ncclGroupEnd: %s\n", rank, ncclGetErrorString(ge)); err = 1; }
  }
  TRY_HIP(hipStreamSynchronize(g.stream));
  if (!err && rank == 0) {
    parts.assign(world, {});
    parts[0].assign(part, part + n_part);
    for (int r = 1; r < world; r++) { parts[r].resize(sizes[r]); TRY_HIP(hipMemcpy(parts[r].data(), g.d_recv[r], sizes[r], hipMemcpyDeviceToHost)); }
  }
  return err;
}

// --selftest MODE (test hook, no device needed): the ranks skip the library and RCCL and only run the status round; MODE "fail:R" rank R reports a failure,
// "hang:R" rank R never reports, "ok" everybody reports ok
static int selftest_child(int rank, int world, const std::string& dir, const char* mode, double timeout_s) {
  int who = -1; if (const char* c = strchr(mode, ':')) who = atoi(c + 1);
  if (!strncmp(mode, "hang", 4) && rank == who) { for (;;) { if (exists(dir + "/abort")) return 1; std::this_thread::sleep_for(std::chrono::milliseconds(5)); } }
  const bool ok = !(!strncmp(mode, "fail", 4) && rank == who);
  return status_round(dir, rank, world, ok, "selftest", timeout_s);
}

static int child_main(int rank, int world, const char* dir_c, int argc, char** argv) {
  g_rank = rank;
  const std::string dir = dir_c;
  double timeout_s = 600; const char* selftest = nullptr;
  while (argc >= 2 && argv[0][0] == '-') {
    if (!strcmp(argv[0], "--timeout")) timeout_s = atof(argv[1]); else if (!strcmp(argv[0], "--selftest")) selftest = argv[1]; else break;
    argc -= 2; argv += 2;
  }
  if (selftest) return selftest_child(rank, world, dir, selftest, timeout_s);
  if (argc < 2) { fprintf(stderr, "rank %d: missing file names\n", rank); return 2; }
  const char* in_path = argv[0]; const char* out_path = argv[1];
  const int depth = argc > 2 ? atoi(argv[2]) : 8;
  rbt_v3c_params vp; memset(&vp, 0, sizeof(vp));
  vp.geometry_qp = argc > 3 ? atoi(argv[3]) : 24; vp.attribute_qp = argc > 4 ? atoi(argv[4]) : 32; vp.occupancy_precision = argc > 5 ? atoi(argv[5]) : 4;
  vp.occupancy_rd = argc > 6 ? atoi(argv[6]) : 0; vp.ctb_rows_per_slice = -1; vp.gofs_per_job = 0;
  const auto t0 = clk::now();
  // ---- this rank's GOFs through the library; whatever goes wrong ends in the status round, never in an early exit the peers would wait for
  std::vector<uint8_t> in; std::string why; uint8_t* part = nullptr; size_t n_part = 0;
  if (!read_file(in_path, in)) why = std::string("cannot read ") + in_path;
  else {
    rbt_ctx* ctx = nullptr;
    int rc = rbt_create(&ctx, rank, rank, world);
    if (rc != RBT_OK) why = std::string("rbt_create: ") + rbt_strerror(rc);                                         // no GPU: there is no CPU path
    else {
      if ((rc = rbt_set_depth(ctx, depth)) == RBT_OK) rc = rbt_transcode_v3c(ctx, in.data(), in.size(), &vp, &part, &n_part);
      if (rc != RBT_OK) why = std::string("rbt_transcode_v3c: ") + rbt_strerror(rc) + " " + rbt_last_error(ctx);
      rbt_destroy(ctx);
    }
  }
  if (!why.empty()) fprintf(stderr, "rank %d: %s\n", rank, why.c_str());
  if (status_round(dir, rank, world, why.empty(), why, timeout_s)) { rbt_free(part); return 1; }
  const auto t1 = clk::now();
  // ---- gather on rank 0 over RCCL (all ranks are alive and past their transcode here)
  std::vector<std::vector<uint8_t>> parts;
  int ret = gather_parts(dir, rank, world, part, n_part, timeout_s, parts);
  rbt_free(part);
  if (!ret && rank == 0) {
    uint8_t* out = nullptr; size_t n = 0;
    const int rc = merge_parts(parts, vp.forced_unit_size_precision_bytes, &out, &n);
    if (rc != RBT_OK) { fprintf(stderr, "merge: %s\n", rbt_strerror(rc)); ret = 1; }
    else {
      if (!publish(out_path, out, n)) { fprintf(stderr, "cannot write %s\n", out_path); ret = 2; }
      const auto t2 = clk::now();
      rbt_v3c_stat a, b;
      if (!ret && rbt_v3c_stats(in.data(), in.size(), &a) == RBT_OK && rbt_v3c_stats(out, n, &b) == RBT_OK)
        printf("%d ranks, %d GOFs: %llu -> %llu bytes; transcode %.1f ms (rank 0), gather + merge + write %.1f ms\n", world, a.n_gofs, (unsigned long long)a.total, (unsigned long long)b.total,
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
      rbt_free(out);
    }
  }
  return ret;
}

int main(int argc, char** argv) {
  if (argc >= 5 && !strcmp(argv[1], "--child")) return child_main(atoi(argv[2]), atoi(argv[3]), argv[4], argc - 5, argv + 5);
  if (argc < 5 || strcmp(argv[1], "--ranks")) { fprintf(stderr, "usage: %s --ranks N [--timeout S] in.bin out.bin [depth] [geometryQP] [attributeQP] [occupancyPrecision] [occupancyRd]\n", argv[0]); return 2; }
  const int world = atoi(argv[2]);
  if (world < 1 || world > 64) { fprintf(stderr, "--ranks 1..64\n"); return 2; }
  double timeout_s = 900;
  for (int i = 3; i + 1 < argc && argv[i][0] == '-'; i += 2) if (!strcmp(argv[i], "--timeout")) timeout_s = atof(argv[i + 1]);
  // the parent does nothing that initialises a GPU: every rank is a fresh process
  char dir[] = "/tmp/rbt_multi_gpu_XXXXXX";
  if (!mkdtemp(dir)) { perror("mkdtemp"); return 2; }
  const std::string sdir = dir;
  std::vector<pid_t> pids;
  for (int r = 0; r < world; r++) {
    const pid_t pid = fork();
    if (pid < 0) { perror("fork"); for (pid_t p : pids) kill(p, SIGKILL); return 2; }
    if (pid == 0) {
      std::vector<std::string> a = {argv[0], "--child", std::to_string(r), std::to_string(world), sdir};
      for (int i = 3; i < argc; i++) a.push_back(argv[i]);
      std::vector<char*> av; for (auto& s : a) av.push_back((char*)s.c_str()); av.push_back(nullptr);
      execv(argv[0], av.data());
      perror("execv"); _exit(127);
    }
    pids.push_back(pid);
  }
  // supervise: reap whoever ends; the first failure (or the deadline) raises the abort flag, the rest get a moment to leave on their own and are then killed
  int rc = 0; size_t left = pids.size();
  const auto deadline = clk::now() + std::chrono::duration<double>(timeout_s);
  auto kill_at = clk::time_point::max();
  while (left) {
    int st = 0; const pid_t p = waitpid(-1, &st, WNOHANG);
    if (p > 0) {
      left--;
      for (pid_t& q : pids) if (q == p) q = -1;
      if (!WIFEXITED(st) || WEXITSTATUS(st)) {
        if (!rc) { const std::string why = "a rank ended with status " + std::to_string(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st)); publish(sdir + "/abort", why.data(), why.size()); kill_at = clk::now() + std::chrono::seconds(3); }
        rc = 1;
      }
      continue;
    }
    if (p < 0) break;
    if (!rc && clk::now() > deadline) { const std::string why = "deadline of " + std::to_string((int)timeout_s) + " s passed"; fprintf(stderr, "rbt_multi_gpu: %s\n", why.c_str()); publish(sdir + "/abort", why.data(), why.size()); kill_at = clk::now() + std::chrono::seconds(3); rc = 1; }
    if (clk::now() > kill_at) { for (pid_t q : pids) if (q > 0) kill(q, SIGKILL); kill_at = clk::time_point::max(); }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  for (int r = 0; r < world; r++) { unlink((sdir + "/status." + std::to_string(r)).c_str()); unlink((sdir + "/status." + std::to_string(r) + ".tmp").c_str()); }
  unlink((sdir + "/rccl_id").c_str()); unlink((sdir + "/rccl_id.tmp").c_str()); unlink((sdir + "/abort").c_str()); unlink((sdir + "/abort.tmp").c_str()); rmdir(dir);
  return rc;
}
