#!/usr/bin/env python3
"""Benchmark of the V-PCC transcoding hot path on MI355X (BASELINE.json metric: transcoded point-cloud frames/s).

Workload (config.workload): one 32-frame GOF of 1280x1280 V-PCC maps — 64 geometry + 64 attribute pictures
(yuv420p10, I/P pairs) and 32 occupancy pictures (640x640, 8 bit, lossless) — at "R5" (QP 16 / 22, occupancy
precision 2), transcoded to R3 (geometryQP 24, attributeQP 32, occupancyPrecision 4). There is no 8i data and no HM
here, so the maps are synthetic (tests/synth.py) and the R5 input is the committed fixture tests/golden/hm_r5_1280x1280_f32_*.annexb:
those maps coded with the toolset of the CTC input streams (cfg/hm/ctc-hm-*-ai.cfg: CTU 64, TU 4..32, 35 intra modes, quarter-sample motion
search, AMP, transform skip, SAO, hash SEI; one slice per picture) by the oracle's HM-like encoder (tests/golden/make_hm_gof.py, run in the
build container). --input e1 codes the input with this library's own encoder instead (other sizes). One "step" = one GOF through rbt_submit_gof + rbt_wait_gof (together:
rbt_transcode_gof); --in-flight GOFs (default 16) are submitted ahead of the one being collected, as a transcoder walking a sequence does.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pc-frames F]          (N > 1 without a launcher: starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: weak scaling — every rank walks its own sequence of K GOFs (one process per GPU), the re-encoded NAL units of every GOF are
gathered on rank 0 with RCCL inside the timed region. Extra legs: sequence_walk (configs[3]: 300 frames GOF-sharded, strong scaling,
stitched output checked against the unsharded walk) and rate_fanout (configs[4]: R1..R5, one target rate per rank).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before anything initialises HIP: librbt shares 16 streams among the jobs in flight
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def make_gof_maps(w, h, n_pc, seed):
    import synth
    return synth.make_gof_maps(w, h, n_pc, seed)


def launch_ranks(n):
    """python bench.py --gpus N without a launcher: start the N ranks as child processes (one per GPU) and pass rank 0's line through.
    This parent never touches the GPU (no HIP call, no torch.cuda, librbt not loaded), so nothing that initialised a GPU is ever
    re-executed; the children are ordinary new processes."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # supervise (round-3 review: a rank that fails must not leave its peers in a barrier or a collective for ever, nor this parent in wait()): the first rank that
    # ends non-zero - or the deadline - gives the others a few seconds to end by themselves, then they are killed; the exit code says so
    deadline = time.time() + float(os.environ.get("RBT_BENCH_RANK_TIMEOUT", "3000"))
    rc, kill_at = 0, None
    while any(p_.poll() is None for p_ in procs):
        for r, p_ in enumerate(procs):
            if p_.poll() not in (None, 0) and not rc:
                rc = abs(p_.returncode) or 1
                print(f"bench.py: rank {r} ended with status {p_.returncode}: stopping the other ranks", file=sys.stderr)
                kill_at = time.time() + 5
        if not rc and time.time() > deadline:
            rc = 124
            print("bench.py: the ranks did not finish before the deadline (RBT_BENCH_RANK_TIMEOUT): stopping them", file=sys.stderr)
            kill_at = time.time()
        if kill_at is not None and time.time() >= kill_at:
            for p_ in procs:
                if p_.poll() is None: p_.kill()
            kill_at = None
        time.sleep(0.05)
    for p_ in procs:
        if p_.returncode and not rc: rc = abs(p_.returncode)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--pc-frames", type=int, default=32)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=1280)
    ap.add_argument("--cpu-sample", type=int, default=10, help="point-cloud frames of the CPU baseline sample (0 = skip; 10 frames are ~27 s of one core with the round-3 encoder)")
    ap.add_argument("--multi-gof", type=int, default=8, help="also time G GOFs per call (extra field multi_gof; 0/1 = skip)")
    ap.add_argument("--in-flight", type=int, default=16, help="GOFs in flight (rbt_submit_gof ahead of rbt_wait_gof), 1..16; 1 = blocking calls")
    ap.add_argument("--gofs-per-job", type=int, default=0, help="GOFs handed over per rbt_submit_gof call (a step stays one GOF; the K steps are spread evenly over ceil(K / G) jobs). "
                    "0 = choose by the length of the run (gof_shard.job_shape): 3 for a long run (16 jobs x 3 GOFs keep the GPU full; 2 below 96 steps); a run shorter than 48 steps is all ramp-up and "
                    "drain and does better with few jobs (7, or 2 up to 12 steps) that own several hardware queues each than with many that own one)")
    ap.add_argument("--sweep", type=int, default=64, help="also time K GOFs at every in-flight depth 1..4 (extra field in_flight_sweep; 0/1 = skip)")
    ap.add_argument("--steady-steps", type=int, default=256, help="when --steps is smaller: also time a walk of this many GOFs (extra top-level field steady_state_fps_256; 0 = skip)")
    ap.add_argument("--quality", type=int, default=1, help="report picture PSNR of the output vs the input (extra field quality; 0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0 (with --backend gloo)")
    ap.add_argument("--save-input", default=None, help="write the generated R5 input streams to this .npz file and exit")
    ap.add_argument("--load-input", default=None, help="read the R5 input streams from a file written by --save-input (keeps the input "
                    "encoder's kernels out of a profile of the transcode step); the file must come from the same size / seed")
    ap.add_argument("--input", default="auto", choices=["auto", "hm", "hm-closed", "e1"], help="R5 input: hm = the committed HM-like fixture in the stream structure of the CTC's HM encoder "
                    "(tests/golden/hm_r5ctc_*.annexb: CTC toolset AND structure - one IDR per sub-bitstream, then TRAIL_N P / TRAIL_R intra pictures with reference picture sets, POC 0..63; coded by the "
                    "oracle's HM-like encoder, tests/golden/make_hm_gof.py --ctc 1); hm-closed = the same maps and decisions as closed (IDR, P) pairs with repeated parameter sets (hm_r5_*.annexb, the "
                    "input of rounds 2-3); e1 = coded on the fly by this library's own encoder (any size); auto = hm when the fixture fits")
    ap.add_argument("--rows", type=int, default=-1, help="encoder slice structure (rbt_stream_params.ctb_rows_per_slice): -1 = wavefront mode, one slice per picture coded as one "
                    "dependent slice segment per CTB row (entropy_coding_sync); 1 = one independent slice per CTB row; 0 = one slice per picture")
    ap.add_argument("--walk-frames", type=int, default=300, help="also walk a sequence of this many point-cloud frames GOF-sharded over the ranks "
                    "(BASELINE.json configs[3]: 300 = 9 x 32 + 12; extra field sequence_walk; 0 = skip)")
    ap.add_argument("--fanout-gofs", type=int, default=2, help="also transcode this many GOFs to every rate point R1..R5, one target rate per rank "
                    "(BASELINE.json configs[4]; extra field rate_fanout; 0 = skip)")
    ap.add_argument("--v3c-input", default=None, help="a V3C sample stream file (e.g. a real longdress_r5.bin; SURVEY.md 8(d)): transcoded unchanged, whole file, GOF-sharded over "
                    "the ranks, with --geometry-qp / --attribute-qp / --occupancy-precision (extra field v3c_file; the headline stays on the synthetic GOF)")
    ap.add_argument("--v3c-output", default=None, help="where rank 0 writes the transcoded file of --v3c-input")
    ap.add_argument("--geometry-qp", type=int, default=24)
    ap.add_argument("--attribute-qp", type=int, default=32)
    ap.add_argument("--occupancy-precision", type=int, default=4)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    st_ = os.environ.get("RBT_BENCH_SELFTEST", "")          # test hook of the launcher's supervision (tests/test_gof_shard.py): "fail:R" rank R ends with status 3, the others never end
    if st_.startswith("fail:") and world > 1:
        if rank == int(st_[5:]): sys.exit(3)
        time.sleep(3600)
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=float(os.environ.get("RBT_BENCH_COLLECTIVE_TIMEOUT", "900")))     # a collective whose peer is gone fails after this, it does not wait for ever
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend=args.backend, timeout=tmo)
    import rbt_lib
    R = rbt_lib.module()
    dev = 0 if args.share_device else local_rank
    tdev = f"cuda:{dev}" if args.backend == "nccl" else "cpu"          # where the collectives' tensors live
    ctx = R.Context(device=dev, rank=rank, world=world)   # raises without a GPU: no CPU fallback

    w, h, n_pc = args.width, args.height, args.pc_frames
    gs = rbt_lib.module_file("gof_shard")
    # R5 input. Default: the committed fixture - the synthetic maps below coded with the CTC toolset (SAO, transform skip, AMP, quarter-sample
    # motion, TU trees, 35 intra modes; one slice per picture, CTB 64) by the oracle's HM-like encoder in the build container. Every rank
    # walks the same sequence (weak scaling: K GOFs per rank).
    fixture = None
    man_path = os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")
    if args.input != "e1" and not args.load_input and os.path.exists(man_path):
        mans = json.load(open(man_path))
        man = mans.get(f"{w}x{h}_f32" + ("" if args.input == "hm-closed" else "_ctc")) or (mans.get(f"{w}x{h}_f32") if args.input == "auto" else None)
        if man and n_pc <= 32:      # a prefix of a sub-bitstream is a sub-bitstream in either structure: the first n_pc point-cloud frames
            fixture = {k: gs.first_pictures(open(os.path.join(ROOT, "tests", "golden", v["file"]), "rb").read(), n_pc * (1 if k == "occ" else 2)) for k, v in man["streams"].items()}
    if args.input in ("hm", "hm-closed") and fixture is None:
        sys.exit(f"--input {args.input}: no committed fixture for this size (tests/golden/make_hm_gof.py)")
    ctc_structure = bool(fixture) and not gs.is_closed_pairs(fixture["geo"])
    input_kind = ("hm-like fixture (oracle HM-like encoder: SAO, TS, AMP, quarter-pel ME, TU trees, 35 intra modes), " +
                  ("stream structure of the CTC's HM encoder: one IDR per sub-bitstream, TRAIL_N P / TRAIL_R intra pictures with reference picture sets, POC running on, parameter sets at the IDR only"
                   if ctc_structure else "closed (IDR, P) pairs with repeated parameter sets")) if fixture else "RBT-E1 (this library's encoder, CTB 64, one slice per picture)"
    if fixture:
        sg, sa, so = fixture["geo"], fixture["attr"], fixture["occ"]
        geo = attr = occ = None
    else:
        geo, attr, occ = make_gof_maps(w, h, n_pc, 1051)
    if fixture:
        pass
    elif args.load_input:
        z = np.load(args.load_input)
        sg, sa, so = z["sg"].tobytes(), z["sa"].tobytes(), z["so"].tobytes()
    else:
        sg = ctx.encode(geo, w, h, 10, 16, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
        sa = ctx.encode(attr, w, h, 10, 22, gop=2, log2_ctb=6, rows_per_slice=0, md5_sei=0)
        so = ctx.encode(occ, w // 2, h // 2, 8, 8, gop=1, lossless=1, log2_ctb=6, rows_per_slice=0, md5_sei=0)
    if args.save_input:
        np.savez(args.save_input, sg=np.frombuffer(sg, np.uint8), sa=np.frombuffer(sa, np.uint8), so=np.frombuffer(so, np.uint8))
        return
    P = R.StreamParams
    params = [P(R.RBT_VIDEO_OCCUPANCY, 8, 4, 5, args.rows, 0, 0), P(R.RBT_VIDEO_GEOMETRY, 24, 4, 5, args.rows, 0, 0), P(R.RBT_VIDEO_ATTRIBUTE, 32, 4, 5, args.rows, 0, 0)]
    streams = [so, sg, sa]

    # One step = one GOF through the hot path. The steps are issued the way a transcoder walks a sequence: rbt_submit_gof
    # for GOF i+D-1 before rbt_wait_gof for GOF i (D = --in-flight GOFs in flight on disjoint HIP streams; D = 1 is the
    # blocking rbt_transcode_gof). Every one of the K timed steps is submitted and collected inside the timed region.
    if args.gofs_per_job <= 0:
        args.gofs_per_job = gs.job_shape(args.steps)[0]       # 2 for a long run, ceil(K / 7) for one shorter than 48 steps, ceil(K / 2) up to 12
    D = max(1, min(args.in_flight, 16, (args.steps + max(1, args.gofs_per_job) - 1) // max(1, args.gofs_per_job)))   # never announce a deeper pipeline than the run has steps: shallower pipelines get more streams per job
    stats_acc = {}

    host_t = {"submit": 0.0, "wait": 0.0}

    def collect(job, acc):
        c0 = time.perf_counter()
        outs = ctx.wait_gof(job)
        host_t["wait"] += time.perf_counter() - c0
        if acc is not None:
            for k, v in ctx.stats().items():
                acc[k] = acc.get(k, 0.0) + v
        if world > 1:
            gs = rbt_lib.module_file("gof_shard")
            gs.gather_streams(outs, device=tdev)
        return outs

    # G GOFs per job (--gofs-per-job): with 16 jobs in flight every job owns one hardware queue, and a GPU that still has room is better used by giving each
    # queue's launches the pictures of two GOFs than by queueing a 17th job (MI355X time-slices more than 16 queues). Distinct buffers per GOF (the library
    # decodes entries that name the same buffer once), handed over in one rbt_submit_gof call; a step stays ONE GOF.
    G = max(1, args.gofs_per_job)
    job_cache = {}

    def job_of(g):
        if g not in job_cache:
            job_cache[g] = ([bytes(bytearray(x)) for _ in range(g) for x in streams] if g > 1 else streams, list(params) * g)
        return job_cache[g]

    arena_probe = []

    def run(n_steps, depth, acc, g=None):
        g = g or G
        sizes = gs.spread(n_steps, g)                                           # exactly n_steps GOFs, spread evenly over the jobs (20 steps, G = 3: 3 3 3 3 3 3 2)
        q, outs = [], None
        for sz in sizes:
            js, jp = job_of(sz)
            if len(q) == depth: outs = collect(q.pop(0), acc)
            c0 = time.perf_counter()
            q.append(ctx.submit_gof(js, jp))
            host_t["submit"] += time.perf_counter() - c0
            if not arena_probe: arena_probe.append(ctx.job_memory(q[-1]) / sz)      # device memory one GOF of this workload holds while it is in flight (decoder + encoder arenas; rbt_job_memory), asked once
        while q: outs = collect(q.pop(0), acc)
        return outs[:3] if outs else outs

    def sync():
        if world > 1:
            import torch
            dist.barrier()
            if args.backend == "nccl": torch.cuda.synchronize()

    ctx.set_depth(D)
    # set-up, untimed: one job per slot, so that every slot's device arenas exist (librbt recycles them) even when W < D
    primed = D if (args.warmup < D * G and D > 1) else 0
    if primed: run(primed * G, D, None)
    run(args.warmup, D, None)
    sync()
    host_t["submit"] = host_t["wait"] = 0.0
    t0 = time.perf_counter()
    outs = run(args.steps, D, stats_acc)
    sync()
    elapsed = time.perf_counter() - t0
    host_submit_ms, host_wait_ms = 1000 * host_t["submit"] / args.steps, 1000 * host_t["wait"] / args.steps
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps = args.steps
    fps = world * n_pc * steps / elapsed
    n_jobs = (steps + G - 1) // G                              # kernel timings are per job = per launch group (a job's launches cover its GOFs)
    gpj = steps / n_jobs                                       # GOFs per job of the timed run (G unless K is not a multiple)
    st = {k: v / n_jobs for k, v in stats_acc.items()}

    # dominant kernel group of the path (GPU-side hipEvent timings taken on the launch stream inside librbt)
    groups = {"cabac_parse": st["k_parse_ms"], "reconstruct+loopfilter": st["k_recon_ms"], "intra_analysis": st["k_analyse_ms"],
              "encode_recon": st["k_encode_ms"], "cabac_encode": st["k_entropy_ms"]}
    dom = max(groups, key=groups.get)
    in_bytes = sum(len(s_) for s_ in streams)
    out_bytes = sum(len(o) for o in outs)
    fsz = lambda ww, hh: ww * hh * 3  # noqa: E731  (4:2:0, 2 bytes per sample)
    pics = {"geo": (2 * n_pc, fsz(w, h)), "attr": (2 * n_pc, fsz(w, h)), "occ_in": (n_pc, fsz(w // 2, h // 2)), "occ_out": (n_pc, fsz(w // 4, h // 4))}
    dec_pix = pics["geo"][0] * pics["geo"][1] + pics["attr"][0] * pics["attr"][1] + pics["occ_in"][0] * pics["occ_in"][1]
    enc_pix = pics["geo"][0] * pics["geo"][1] + pics["attr"][0] * pics["attr"][1] + pics["occ_out"][0] * pics["occ_out"][1]
    # algorithmic bytes per launch group (DESIGN.md "Measurement"): what each stage must move at minimum
    alg = {"cabac_parse": in_bytes + dec_pix,                 # slice data in, one coefficient level per sample out
           "reconstruct+loopfilter": 2 * dec_pix + dec_pix // 2,   # levels in, samples out, P pictures read their reference
           "intra_analysis": enc_pix // 2,                     # I-picture source samples
           "encode_recon": 3 * enc_pix + enc_pix // 2,         # source in, levels + reconstruction out, P reference in
           "cabac_encode": enc_pix + out_bytes}                # levels in, slice data out
    achieved = gpj * alg[dom] / (groups[dom] * 1e-3) / 1e9 if groups[dom] > 0 else 0.0      # a launch covers the GOFs of its job
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs of this same command: profiles/r02_pmc_traffic.json, tools/refresh_profiles.py); counters cannot be read live from inside the benchmark
    traffic = None
    try:
        pmc_file = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")     # HM-like input, this round's code; else the files of the rounds before (the round-1 file belongs to the RBT-E1-coded input)
        if not os.path.exists(pmc_file): pmc_file = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if not os.path.exists(pmc_file): pmc_file = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
        if not fixture or not os.path.exists(pmc_file): pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        pmc = json.load(open(pmc_file))["kernels"]
        kmap = {"cabac_parse": ["k_parse"], "intra_analysis": ["k_enc_analyse"], "cabac_encode": ["k_entropy"]}
        if dom in kmap and n_pc == 32 and (w, h) == (1280, 1280):
            traffic = int(gpj * sum((pmc[k]["FETCH_SIZE_KB"] + pmc[k]["WRITE_SIZE_KB"]) * 1024 for k in kmap[dom]))   # counters were taken on one GOF
    except Exception:
        traffic = None
    # the practical bound of the path is the seriality of entropy decoding: bits per second through one slice's chain
    def nal_sizes(b):
        pos, out = [], []
        i = b.find(b"\x00\x00\x01")
        while i >= 0:
            pos.append(i); i = b.find(b"\x00\x00\x01", i + 3)
        for a, e in zip(pos, pos[1:] + [len(b)]):
            if ((b[a + 3] >> 1) & 63) < 32: out.append(e - a - 3)     # VCL NAL units = slice segments
        return out
    sl_sizes = [n_ for s_ in streams for n_ in nal_sizes(s_)]
    # with more than 8 GOFs in flight the entropy decoding of a GOF is one launch (k_parse_tasks), whose duration is that of its largest slice
    cabac = {"slices_per_gof": len(sl_sizes), "largest_slice_kbit": round(max(sl_sizes) * 8 / 1000, 1),
             "Mbit_per_s_through_largest_slice": round(max(sl_sizes) * 8 / 1e6 / (st["k_parse_ms"] * 1e-3), 2) if D > 8 and st["k_parse_ms"] > 0 else None,
             "note": "largest slice's bits / duration of the entropy-decoding launch that contains it (about 1.17 bins per bit)"}
    path_achieved = st["algorithmic_bytes"] / gpj / (elapsed / steps) / 1e9   # whole path: SURVEY.md 8(d) bytes of one GOF over the time one GOF takes
    # The roofline this path lives under is not HBM but instruction issue: a CU issues at most one scalar and one vector instruction per cycle (each SIMD gets a turn every
    # 4 cycles), 256 CUs x 2.4 GHz = 614 G/s of each kind, and the codec's serial work runs on the scalar unit on purpose. Instructions per GOF from the
    # committed counter pass of this code (rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU, profiles/r0N_pmc_sq.json) x GOFs per second of the timed run.
    issue = None
    try:
        sq_file = next(f for f in (os.path.join(ROOT, "profiles", n) for n in ("r04_pmc_sq.json", "r03_pmc_sq.json")) if os.path.exists(f))
        if fixture and n_pc == 32 and (w, h) == (1280, 1280):
            ks = json.load(open(sq_file))["kernels"]
            salu, valu = sum(v["SQ_INSTS_SALU"] for v in ks.values()), sum(v["SQ_INSTS_VALU"] for v in ks.values())
            gofs_per_s = world * steps / elapsed; peak = 256 * 2.4e9
            issue = {"salu_G_per_gof": round(salu / 1e9, 2), "valu_G_per_gof": round(valu / 1e9, 2), "peak_G_per_s_each": round(peak / 1e9), "scalar_issue_frac": round(salu * gofs_per_s / world / peak, 3),
                     "vector_issue_frac": round(valu * gofs_per_s / world / peak, 3), "counters": os.path.basename(sq_file),
                     "note": "one scalar + one vector instruction per CU and cycle at most; waves here are dependency chains (a lone wave issues every ~4.6 cycles), and two scalar-heavy waves on one SIMD already contend for its scalar turn"}
    except Exception:
        issue = None

    # configs[3]: a sequence of --walk-frames point-cloud frames (300 = 9 GOFs of 32 + one of 12), GOF g on rank g mod world, D GOFs in flight
    # per GPU, re-encoded NAL units gathered on rank 0 (strong scaling: the sequence is fixed). Rank 0 then walks the whole sequence alone
    # and checks that the stitched output is identical.
    walk = None
    ctx.trim()        # the legs below run other job shapes than the headline loop: its arenas (32 GOFs' worth) go back to the driver first
    if args.walk_frames > 0 and n_pc > 1:
        seq = gs.make_sequence(streams, args.walk_frames, n_pc)
        WD = min(16, args.in_flight)
        gs.transcode_sequence(ctx, seq, params, rank=rank, world=world, depth=WD, device=tdev, gofs_per_job=0)          # untimed pass: arenas of this shape exist
        sync()
        w0 = time.perf_counter()
        stitched = gs.transcode_sequence(ctx, seq, params, rank=rank, world=world, depth=WD, device=tdev, gofs_per_job=0)
        sync()
        wt = time.perf_counter() - w0
        if world > 1:
            import torch
            t = torch.tensor([wt], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wt = float(t.item())
        if rank == 0:
            c1 = R.Context(device=dev, rank=0, world=1) if world > 1 else ctx
            alone = gs.transcode_sequence(c1, seq, params, depth=WD)
            if c1 is not ctx: c1.close()
            walk = {"frames": args.walk_frames, "gofs": [len(gs.access_units(g[0])) for g in seq], "ranks": world, "value": round(args.walk_frames / wt, 3), "unit": "point-cloud frames/s",
                    "seconds": round(wt, 4), "scaling": "strong", "out_bytes": sum(len(s_) for g in stitched for s_ in g), "stitched_equals_unsharded": stitched == alone,
                    "job_shape": dict(zip(("gofs_per_job", "jobs_in_flight"), gs.job_shape(len(gs.gofs_of_rank(len(seq), 0, world)), WD)))}
        # the same sequence as a V3C sample stream, file in -> file out (rbt_transcode_v3c: the loop of PccAppTranscoder.cpp:277-349 around transcodeData;
        # includes the sample stream <-> byte stream conversions and the container write on the host; sharded like the walk above, partial files merged on rank 0)
        data = gs.wrap_v3c(R, seq, lib=ctx.L)
        gq, aq, prec = params[1].qp, params[2].qp, params[0].occupancy_precision
        kw = dict(occupancy_precision=prec, rows_per_slice=args.rows, gofs_per_job=0)
        gs.transcode_v3c(ctx, R, data, gq, aq, rank=rank, world=world, depth=WD, device=tdev, **dict(kw))
        sync()
        c0 = time.perf_counter()
        merged = gs.transcode_v3c(ctx, R, data, gq, aq, rank=rank, world=world, depth=WD, device=tdev, **dict(kw))
        sync()
        ct = time.perf_counter() - c0
        if world > 1:
            import torch
            t = torch.tensor([ct], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ct = float(t.item())
        if rank == 0:
            walk["container"] = {"value": round(args.walk_frames / ct, 3), "unit": "point-cloud frames/s", "seconds": round(ct, 4), "bytes_in": len(data), "bytes_out": len(merged),
                                 "stat_in": {k: v for k, v in R.v3c_stats(data, ctx.L).items() if k.startswith("total")}, "stat_out": {k: v for k, v in R.v3c_stats(merged, ctx.L).items() if k.startswith("total")},
                                 "video_units_equal_walk": gs.unwrap_v3c(R, merged, lib=ctx.L, annexb=False) == [[R.byte_to_sample_stream(s_, ctx.L) for s_ in g] for g in stitched]}
    # a V3C sample stream file handed over at run time: file in -> file out, as PccAppTranscoder does it (rbt_transcode_v3c), sharded like the walk
    v3c_file = None
    if args.v3c_input:
        ctx.trim()
        data = open(args.v3c_input, "rb").read()
        try:
            idx = R.v3c_index(data, ctx.L)
            # point-cloud frames = pictures of the geometry video / maps per frame (2 in the CTC): pictures counted as slice segments with first_slice_segment_in_pic_flag
            def pictures(payload):
                bs, n_pic, i = R.sample_to_byte_stream(payload, ctx.L), 0, 0
                i = bs.find(b"\x00\x00\x01")
                while i >= 0:
                    if ((bs[i + 3] >> 1) & 63) < 32 and i + 5 < len(bs) and bs[i + 5] & 0x80: n_pic += 1
                    i = bs.find(b"\x00\x00\x01", i + 3)
                return n_pic
            n_pics = sum(pictures(data[u["offset"] + 4:u["offset"] + u["size"]]) for u in idx if u["video_type"] == R.RBT_VIDEO_GEOMETRY)
            kw = dict(occupancy_precision=args.occupancy_precision, rows_per_slice=args.rows, gofs_per_job=0)
            WD = min(16, args.in_flight)
            gs.transcode_v3c(ctx, R, data, args.geometry_qp, args.attribute_qp, rank=rank, world=world, depth=WD, device=tdev, **dict(kw))     # untimed pass
            sync()
            v0 = time.perf_counter()
            merged = gs.transcode_v3c(ctx, R, data, args.geometry_qp, args.attribute_qp, rank=rank, world=world, depth=WD, device=tdev, **dict(kw))
            sync()
            vt = time.perf_counter() - v0
            if world > 1:
                import torch
                t = torch.tensor([vt], dtype=torch.float64, device=tdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                vt = float(t.item())
            if rank == 0:
                if args.v3c_output: open(args.v3c_output, "wb").write(merged)
                v3c_file = {"file": os.path.basename(args.v3c_input), "gofs": idx[-1]["gof"] + 1 if idx else 0, "geometry_pictures": n_pics, "frames": n_pics // 2, "ranks": world,
                            "value": round(n_pics / 2 / vt, 3), "unit": "point-cloud frames/s (two geometry maps per frame)", "seconds": round(vt, 4), "geometry_qp": args.geometry_qp,
                            "attribute_qp": args.attribute_qp, "occupancy_precision": args.occupancy_precision,
                            "stat_in": {k: v for k, v in R.v3c_stats(data, ctx.L).items() if k.startswith("total")}, "stat_out": {k: v for k, v in R.v3c_stats(merged, ctx.L).items() if k.startswith("total")}}
        except R.RbtError as e:
            if world > 1: raise
            v3c_file = {"file": os.path.basename(args.v3c_input), "error": str(e)}
    # configs[4]: every rate point R1..R5 from the R5 input, target rate i on rank i mod world (decode replicated; a rank that holds several
    # rates hands each GOF over once and the library decodes it once)
    fanout = None
    ctx.trim()
    if args.fanout_gofs > 0:
        fseq = [streams] * args.fanout_gofs
        gs.transcode_fanout(ctx, R, fseq, rank=rank, world=world, depth=D, device=tdev, rows_per_slice=args.rows)      # untimed: same shape, so the timed pass allocates nothing
        sync()
        f0 = time.perf_counter()
        fan = gs.transcode_fanout(ctx, R, fseq, rank=rank, world=world, depth=D, device=tdev, rows_per_slice=args.rows)
        sync()
        ft = time.perf_counter() - f0
        if world > 1:
            import torch
            t = torch.tensor([ft], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ft = float(t.item())
        if rank == 0:
            fanout = {"gofs": args.fanout_gofs, "rates": {f"R{r}": {"geoQP": gs.RATE_POINTS[r][0], "attrQP": gs.RATE_POINTS[r][1], "occPrecision": gs.RATE_POINTS[r][2],
                                                                    "out_bytes_per_gof": sum(len(s_) for s_ in fan[r][0])} for r in sorted(fan)},
                      "active_ranks": min(world, 5), "value": round(5 * args.fanout_gofs * n_pc / ft, 3), "unit": "output point-cloud frames/s (5 rate points)", "seconds": round(ft, 4),
                      "R3_equals_single_rate_call": fan[3][0] == outs}
    ctx.set_depth(D)

    # informative extra (not the headline): G GOFs handed over in one call. One GOF's critical path is a few hundred serial
    # waves, so the GPU has room for several at once; a sequence of GOFs (configs[3]) can use that inside each GPU.
    multi = None
    ctx.trim()
    if world == 1 and args.multi_gof > 1:
        MG = args.multi_gof
        ms, mp = [bytes(bytearray(s_)) for _ in range(MG) for s_ in streams], params * MG     # distinct buffers: identical ones would be decoded once
        ctx.transcode_gof(ms, mp)
        m0 = time.perf_counter()
        for _ in range(2): mo = ctx.transcode_gof(ms, mp)
        mt = (time.perf_counter() - m0) / 2
        assert all(mo[3 * g + q] == outs[q] for g in range(MG) for q in range(3))
        multi = {"gofs_per_call": MG, "value": round(MG * n_pc / mt, 3), "unit": "point-cloud frames/s", "ms_per_call": round(1000 * mt, 3)}

    # informative extra: the same loop with ONE GOF per job at every depth (D = 1 is the blocking call: its ms_per_gof is the latency of one GOF)
    sweep = None
    ctx.trim()
    if world == 1 and args.sweep > 1:
        sweep = []
        for d in (1, 2, 4, 8, 16):
            ctx.set_depth(d)
            run(d, d, None, 1)
            p0 = time.perf_counter()
            so_ = run(args.sweep, d, None, 1)
            pt = (time.perf_counter() - p0) / args.sweep
            assert so_ == outs
            sweep.append({"in_flight": d, "gofs": args.sweep, "value": round(n_pc / pt, 3), "ms_per_gof": round(1000 * pt, 3)})
        ctx.set_depth(D)
    # the headline of a short run (the driver's --steps 20) is all ramp-up and drain: the steady state of a long walk beside it (the same job shape rule as a 256-step run)
    steady = None
    ctx.trim()
    if world == 1 and args.steady_steps > 0 and args.steps < args.steady_steps:
        g_s = gs.job_shape(args.steady_steps)[0]; d_s = max(1, min(args.in_flight, 16, (args.steady_steps + g_s - 1) // g_s))
        ctx.set_depth(d_s)
        run(d_s * g_s, d_s, None, g_s)
        p0 = time.perf_counter(); run(args.steady_steps, d_s, None, g_s); pt = time.perf_counter() - p0
        steady = {"steps": args.steady_steps, "value": round(n_pc * args.steady_steps / pt, 3), "unit": "point-cloud frames/s", "jobs_in_flight": d_s, "gofs_per_job": g_s}
        ctx.set_depth(D)

    # informative: what the re-encode did to the pictures (luma PSNR of the R3 output's pictures against the R5 input's,
    # both decoded by this library), and that the occupancy output is exactly the 2x2 OR-pool of the input occupancy
    quality = None
    ctx.trim()        # every leg above ran another job shape and left its arenas cached
    if rank == 0 and world == 1 and args.quality:
        def psnr_y(a_stream, b_stream, ww, hh, peak):
            da, db = ctx.decode(a_stream)[0], ctx.decode(b_stream)[0]
            ya, yb = da[:, :ww * hh].astype(np.float64), db[:, :ww * hh].astype(np.float64)
            mse = float(np.mean((ya - yb) ** 2))
            return round(10 * np.log10(peak * peak / mse), 2) if mse > 0 else float("inf")
        oi, oo = ctx.decode(so)[0], ctx.decode(outs[0])[0]
        ow, oh = w // 2, h // 2
        pooled = (oi[:, :ow * oh].reshape(-1, oh // 2, 2, ow // 2, 2).max(axis=(2, 4)) > 0)
        # D1 half of the metric (PCCMetrics.cpp:75-231, peak 1023): point-cloud frame 0 rebuilt on the GPU (rbt_reconstruct, the decoder-side
        # stage PCCCodec::generatePointCloud) from the synthetic atlas and (a) the uncoded source maps, (b) the decoded R5 input, (c) the decoded R3
        # output; D1 of (b) and (c) against (a) - what PccAppMetrics reports against the original cloud - and of (c) against (b)
        d1 = None
        try:
            import synth
            src = synth.make_maps(w, h, 1051)
            pats = synth.atlas_patches(R, w, h, 1051)
            first = lambda s_, k_, maps=1: gs.first_pictures(s_, k_ * maps)      # the first k_ point-cloud frames of a sub-bitstream (maps: pictures per frame)

            smoothed_pts = []

            def cloud(occ_plane, prec, g2, smooth=True):
                # decoded clouds go through the decoder's geometry smoothing, which the CTC switches on (cfg/common/ctc-common.cfg:57-60: gridSmoothing, gridSize 8, thresholdSmoothing 64;
                # PCCDecoder.cpp:434-437): what PccAppDecoder + PccAppMetrics would score. smooth=False: the cloud as generatePointCloud leaves it (the figure of rounds 2-3)
                c_ = ctx.reconstruct(R.AtlasParams(w, h, 16, prec, 2, 1, 1, 0, 1 if smooth else 0, 8, 64), pats, occ_plane, g2[0][: w * h].reshape(h, w), g2[1][: w * h].reshape(h, w), 10)[0]
                if smooth: smoothed_pts.append(int(ctx.n_smoothed))
                return c_
            c_src, n_src = synth.source_normals(R, ctx.reconstruct, w, h, 1051, src["occ_full"], src["geo"])     # the source cloud with one normal per point: its patch's projection axis
            c_in = cloud(ctx.decode(first(so, 1))[0][0][: (w // 2) * (h // 2)].reshape(h // 2, w // 2), 2, ctx.decode(first(sg, 1, 2))[0])
            c_out = cloud(ctx.decode(first(outs[0], 1))[0][0][: (w // 4) * (h // 4)].reshape(h // 4, w // 4), 4, ctx.decode(first(outs[1], 1, 2))[0])
            r_in, r_out, r_io = ctx.d1(c_src, c_in), ctx.d1(c_src, c_out), ctx.d1(c_in, c_out)
            p_in, p_out = ctx.d2(c_src, n_src, c_in), ctx.d2(c_src, n_src, c_out)          # D2 (point-to-plane, PCCMetrics.cpp:100-124): rbt_d2
            n_sm_in, n_sm_out = smoothed_pts[0], smoothed_pts[1]
            c_out_plain = cloud(ctx.decode(first(outs[0], 1))[0][0][: (w // 4) * (h // 4)].reshape(h // 4, w // 4), 4, ctx.decode(first(outs[1], 1, 2))[0], smooth=False)
            r_out_plain, p_out_plain = ctx.d1(c_src, c_out_plain), ctx.d2(c_src, n_src, c_out_plain)
            d1_tools = (cloud, c_src, first, n_src)
            src_cache = {}

            def d1d2_frames(occ_stream, prec, geo_stream, nfr=min(4, n_pc)):
                """D1 / D2 of point-cloud frames 0..nfr-1 (the GOF's four base atlases, tests/synth.py make_gof_maps) against their source clouds: one frame's D1 scatters by
                +-0.3 dB with any change of the encoder (a depth error of 1 on a few thousand of 600 000 points), the mean of four is what comparisons should read"""
                ow_, oh_ = w // prec, h // prec
                occ_d, geo_d = ctx.decode(first(occ_stream, nfr))[0], ctx.decode(first(geo_stream, nfr, 2))[0]
                res = []
                for k in range(nfr):
                    if k not in src_cache:
                        sk = synth.make_maps(w, h, 1051 + k); pk = synth.atlas_patches(R, w, h, 1051 + k)
                        src_cache[k] = (pk,) + synth.source_normals(R, ctx.reconstruct, w, h, 1051 + k, sk["occ_full"], sk["geo"])
                    pk, cs, ns = src_cache[k]
                    ck = ctx.reconstruct(R.AtlasParams(w, h, 16, prec, 2, 1, 1, 0, 1, 8, 64), pk, occ_d[k][: ow_ * oh_].reshape(oh_, ow_), geo_d[2 * k][: w * h].reshape(h, w), geo_d[2 * k + 1][: w * h].reshape(h, w), 10)[0]
                    res.append((round(ctx.d1(cs, ck)["psnr"], 3), round(ctx.d2(cs, ns, ck)["psnr"], 3)))
                return res
            f4_in, f4_out = d1d2_frames(so, 2, sg), d1d2_frames(outs[0], 4, outs[1])
            d1 = {"points_source": int(c_src.shape[0]), "points_r5_input": int(c_in.shape[0]), "points_r3_output": int(c_out.shape[0]),
                  "d1_psnr_r5_input_vs_source_db": round(r_in["psnr"], 3), "d1_psnr_r3_output_vs_source_db": round(r_out["psnr"], 3), "d1_psnr_r3_output_vs_r5_input_db": round(r_io["psnr"], 3),
                  "d2_psnr_r5_input_vs_source_db": round(p_in["psnr"], 3), "d2_psnr_r3_output_vs_source_db": round(p_out["psnr"], 3),
                  "geometry_smoothing": {"grid_size": 8, "threshold": 64, "points_moved_r5_input": n_sm_in, "points_moved_r3_output": n_sm_out,
                                         "d1_psnr_r3_output_vs_source_db_without": round(r_out_plain["psnr"], 3), "d2_psnr_r3_output_vs_source_db_without": round(p_out_plain["psnr"], 3),
                                         "note": "decoded clouds are scored after the decoder's grid smoothing (CTC default on; rbt_atlas_params.geometry_smoothing), as PccAppDecoder + PccAppMetrics would; "
                                                 "'without': the cloud as generatePointCloud leaves it (what rounds 2-3 reported). The patches of the synthetic atlas are planes scattered in space: few of them meet, so the smoothing has little to move here"},
                  "frames_0_3": {"d1_r5_input_vs_source_db": [a for a, _ in f4_in], "d1_r3_output_vs_source_db": [a for a, _ in f4_out], "d2_r3_output_vs_source_db": [b for _, b in f4_out],
                                 "d1_mean_r5_input_db": round(sum(a for a, _ in f4_in) / len(f4_in), 3), "d1_mean_r3_output_db": round(sum(a for a, _ in f4_out) / len(f4_out), 3),
                                 "d2_mean_r3_output_db": round(sum(b for _, b in f4_out) / len(f4_out), 3),
                                 "note": "the GOF's four base atlases (frames 0..3); a single frame's D1 scatters by +-0.3 dB between encoder variants, compare the means"},
                  "note": "point-cloud frame 0, synthetic atlas (tests/synth.py atlas_patches), symmetric PSNR, peak 1023; D1 point-to-point (rbt_d1), D2 point-to-plane (rbt_d2) with "
                          "the source's normals = the projection axis of each point's patch (tests/synth.py source_normals), the decoded cloud's by scaleNormals"}
        except Exception as e:   # the metric stage is informative: never lose the benchmark line over it
            d1 = {"error": str(e)}
        # Occupancy-aware coding (rbt_stream_params.occupancy_rd) on top: the same GOF with the geometry / attribute maps coded for the samples the
        # decoder makes points of. Bytes, D1 of frame 0, luma PSNR of the OCCUPIED samples (output occupancy map, full resolution) against the R5 input's, and the rate
        # of a short run in the headline's job shape. The headline stays without it (the reference's libx265 path knows nothing of the occupancy map).
        occ_rd = None
        try:
            p_on = [P(R.RBT_VIDEO_OCCUPANCY, 8, 4, 5, args.rows, 0, 0, 0), P(R.RBT_VIDEO_GEOMETRY, 24, 4, 5, args.rows, 0, 0, 1), P(R.RBT_VIDEO_ATTRIBUTE, 32, 4, 5, args.rows, 0, 0, 1)]
            ctx.set_depth(1)
            o_on = ctx.transcode_gof(streams, p_on)
            m = (ctx.decode(o_on[0])[0][:, : (w // 4) * (h // 4)].reshape(-1, h // 4, w // 4) > 0).repeat(4, 1).repeat(4, 2).repeat(2, 0)      # two maps per point-cloud frame

            def psnr_occ(a_stream, b_stream):
                ya, yb = ctx.decode(a_stream)[0][:, : w * h].reshape(-1, h, w).astype(np.float64), ctx.decode(b_stream)[0][:, : w * h].reshape(-1, h, w).astype(np.float64)
                return round(10 * np.log10(1023.0 * 1023.0 / float(np.mean(((ya - yb) ** 2)[m]))), 2)
            occ_rd = {"out_bytes": sum(len(o) for o in o_on), "out_bytes_occupancy_geometry_attribute": [len(o) for o in o_on], "plain_out_bytes_occupancy_geometry_attribute": [len(o) for o in outs],
                      "geometry_bytes_saved": round(1 - len(o_on[1]) / len(outs[1]), 4), "attribute_bytes_saved": round(1 - len(o_on[2]) / len(outs[2]), 4),
                      "out_over_in": round(sum(len(o) for o in o_on) / in_bytes, 4),
                      "occupied_psnr_y_vs_r5_input_db": {"geometry": psnr_occ(sg, o_on[1]), "attribute": psnr_occ(sa, o_on[2])},
                      "plain_occupied_psnr_y_vs_r5_input_db": {"geometry": psnr_occ(sg, outs[1]), "attribute": psnr_occ(sa, outs[2])}}
            if d1 and "error" not in d1:
                cloud, c_src, first, n_src = d1_tools
                c_on = cloud(ctx.decode(first(o_on[0], 1))[0][0][: (w // 4) * (h // 4)].reshape(h // 4, w // 4), 4, ctx.decode(first(o_on[1], 1, 2))[0])
                occ_rd["d1_psnr_vs_source_db"] = round(ctx.d1(c_src, c_on)["psnr"], 3)
                occ_rd["plain_d1_psnr_vs_source_db"] = d1["d1_psnr_r3_output_vs_source_db"]
                occ_rd["d2_psnr_vs_source_db"] = round(ctx.d2(c_src, n_src, c_on)["psnr"], 3)
                occ_rd["plain_d2_psnr_vs_source_db"] = d1["d2_psnr_r3_output_vs_source_db"]
                f4_on = d1d2_frames(o_on[0], 4, o_on[1])
                occ_rd["frames_0_3"] = {"d1_vs_source_db": [a for a, _ in f4_on], "d2_vs_source_db": [b for _, b in f4_on], "d1_mean_db": round(sum(a for a, _ in f4_on) / len(f4_on), 3),
                                        "d2_mean_db": round(sum(b for _, b in f4_on) / len(f4_on), 3), "plain_d1_mean_db": d1["frames_0_3"]["d1_mean_r3_output_db"], "plain_d2_mean_db": d1["frames_0_3"]["d2_mean_r3_output_db"]}
            ctx.set_depth(D)
            k_on = min(args.steps, 32)
            job_cache.clear(); params_keep = list(params); params[:] = p_on
            run(D * G, D, None)        # untimed: one job per slot with this parameter set (the occupancy maps' arenas are first-use allocations: they stay outside)
            t1 = time.perf_counter(); run(k_on, D, None); dt = time.perf_counter() - t1
            params[:] = params_keep; job_cache.clear()
            occ_rd["value"] = round(n_pc * k_on / dt, 2); occ_rd["unit"] = "point-cloud frames/s"; occ_rd["steps"] = k_on
        except Exception as e:   # informative leg: never lose the benchmark line over it
            occ_rd = {"error": str(e)}
        # RBT_PRESET_FAST (rbt_stream_params.preset: what the reference's x265 preset strings "ultrafast" / "superfast" select): the open-loop decisions only
        preset_fast = None
        try:
            p_fast = [P(q.video_type, q.qp, q.occupancy_precision, q.log2_ctb, q.ctb_rows_per_slice, 0, 0, 0, R.RBT_PRESET_FAST) for q in params]
            ctx.set_depth(1)
            o_f = ctx.transcode_gof(streams, p_fast)
            preset_fast = {"out_bytes": sum(len(o) for o in o_f), "out_over_in": round(sum(len(o) for o in o_f) / in_bytes, 4),
                           "geometry_psnr_y_db": psnr_y(sg, o_f[1], w, h, 1023), "attribute_psnr_y_db": psnr_y(sa, o_f[2], w, h, 1023)}
            if d1 and "error" not in d1:
                cloud, c_src, first, n_src = d1_tools
                c_f = cloud(ctx.decode(first(o_f[0], 1))[0][0][: (w // 4) * (h // 4)].reshape(h // 4, w // 4), 4, ctx.decode(first(o_f[1], 1, 2))[0])
                preset_fast["d1_psnr_vs_source_db"] = round(ctx.d1(c_src, c_f)["psnr"], 3)
            ctx.set_depth(D)
            k_f = min(args.steps, 32)
            job_cache.clear(); params_keep = list(params); params[:] = p_fast
            run(D * G, D, None)
            t1 = time.perf_counter(); run(k_f, D, None); dt = time.perf_counter() - t1
            params[:] = params_keep; job_cache.clear()
            preset_fast["value"] = round(n_pc * k_f / dt, 2); preset_fast["unit"] = "point-cloud frames/s"; preset_fast["steps"] = k_f
        except Exception as e:   # informative leg
            preset_fast = {"error": str(e)}
        # the same-data anchor (tests/golden/make_anchor.py, made in the build container): the SOURCE maps of this GOF coded directly at the target QPs by the oracle's HM-like
        # mode - what BASELINE.md's R3 row is for 8i data - next to what the transcode of the R5 stream comes out at, both against the uncoded source maps
        anchor = None
        try:
            an = json.load(open(os.path.join(ROOT, "tests", "golden", "anchor_direct_encode.json")))
            if fixture and (an["width"], an["height"], an["frames"]) == (w, h, n_pc) and "R3" in an["rates"]:
                import synth
                g_src, a_src, _ = synth.make_gof_maps(w, h, n_pc, 1051)

                def psnr_src(maps, stream):
                    d_ = ctx.decode(stream)[0][:, : w * h].astype(np.float64)
                    return round(10 * np.log10(1023.0 * 1023.0 / float(np.mean((maps[:, : w * h].astype(np.float64) - d_) ** 2))), 3)
                anchor = {"direct_encode_R3": an["rates"]["R3"], "encoder": an["encoder"],
                          "transcode_R5_to_R3": {"bytes": {"occupancy": len(outs[0]), "geometry": len(outs[1]), "attribute": len(outs[2]), "total": out_bytes},
                                                 "geometry_psnr_y_vs_source_db": psnr_src(g_src, outs[1]), "attribute_psnr_y_vs_source_db": psnr_src(a_src, outs[2]),
                                                 "d1_psnr_frame0_vs_source_db": (d1 or {}).get("d1_psnr_r3_output_vs_source_db"), "d2_psnr_frame0_vs_source_db": (d1 or {}).get("d2_psnr_r3_output_vs_source_db")},
                          "r5_input": {"bytes": in_bytes, "geometry_psnr_y_vs_source_db": psnr_src(g_src, sg), "attribute_psnr_y_vs_source_db": psnr_src(a_src, sa)},
                          "note": "same maps, same QPs: the direct encode sees the uncoded source (first generation), the transcode the decoded R5 stream (second generation)"}
        except Exception as e:
            anchor = {"error": str(e)}
        quality = {"d1": d1, "anchor_direct_encode": anchor, "occupancy_rd": occ_rd, "preset_fast": preset_fast, "geometry_psnr_y_db": psnr_y(sg, outs[1], w, h, 1023), "attribute_psnr_y_db": psnr_y(sa, outs[2], w, h, 1023),
                   "occupancy_is_or_pool": bool(np.array_equal(oo[:, :(ow // 2) * (oh // 2)].reshape(-1, oh // 2, ow // 2) > 0, pooled)),
                   "note": "picture PSNR: R3 output pictures vs R5 input pictures"}

    cpu = None
    cpu_all = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import oracle_lib as O   # CPU checker, used here only as the timed CPU baseline ("port")
        k = min(args.cpu_sample, n_pc)
        # bounded sample of the same workload: the first k point-cloud frames of the same GOF
        pairs = [gs.frame_pieces(s_, m_) for s_, m_ in zip(streams, (1, 2, 2))]      # the pictures of a point-cloud frame reference nothing outside it (CTC structure: parameter sets put in front of each piece)
        sub = [gs.first_pictures(s_, k * m_) for s_, m_ in zip(streams, (1, 2, 2))]
        c0 = time.perf_counter()
        cpu_out = [O.transcode_substream(sub[0], 0, 8, md5_sei=0, rows_per_slice=args.rows), O.transcode_substream(sub[1], 1, 24, md5_sei=0, rows_per_slice=args.rows), O.transcode_substream(sub[2], 19, 32, md5_sei=0, rows_per_slice=args.rows)]
        ct = time.perf_counter() - c0
        cpu_parity = cpu_out == ctx.transcode_gof(sub, params)     # full-size parity for free: the GPU path on the same sample, byte for byte
        import external_tools
        x265 = "ffmpeg with libx265 present" if external_tools.have_libx265() else "libx265 / ffmpeg unavailable on this box"
        cpu = {"value": round(k / ct, 4), "unit": "point-cloud frames/s", "cores": 1, "kind": "port",
               "sample": f"first {k} point-cloud frames of the same GOF, oracle/liboracle.so (scalar C restatement; {x265}), {ct:.1f} s",
               "output_equals_gpu_path": bool(cpu_parity)}
        # the same restatement on the host cores of this GPU's share: the point-cloud frames of a GOF are independent (I/P pairs,
        # all-intra occupancy), so every worker process takes every N-th frame of the whole GOF
        import subprocess, tempfile
        ncore = max(1, min(16, os.cpu_count() or 1, n_pc))
        per = {"n": np.array(n_pc)}
        for q in range(n_pc):
            per[f"o{q}"] = np.frombuffer(pairs[0][q], np.uint8); per[f"g{q}"] = np.frombuffer(pairs[1][q], np.uint8); per[f"a{q}"] = np.frombuffer(pairs[2][q], np.uint8)
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "frames.npz"); np.savez(f, **per)
            worker = os.path.join(ROOT, "tests", "cpu_baseline_worker.py")
            procs = [subprocess.Popen([sys.executable, worker, f, str(i), str(ncore), str(args.rows)], stdout=subprocess.PIPE, text=True) for i in range(ncore)]
            res = [p_.communicate(timeout=600)[0].split() for p_ in procs]
        if all(p_.returncode == 0 for p_ in procs) and sum(int(r_[0]) for r_ in res) == n_pc:
            mt = max(float(r_[1]) for r_ in res)
            cpu_all = {"value": round(n_pc / mt, 4), "unit": "point-cloud frames/s", "cores": ncore, "kind": "port",
                       "sample": f"all {n_pc} point-cloud frames of the same GOF, one oracle process per core, slowest worker {mt:.1f} s"}

    if rank == 0:
        import external_tools
        ext = external_tools.describe()
        line = {"metric": "transcoded point-cloud frames/sec, R5->R3", "value": round(fps, 3), "unit": "point-cloud frames/s", "n_gpus": world,
                "steps": steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u16/i32", "data": "synthetic",
                "latency_ms_one_gof": (sweep[0]["ms_per_gof"] if sweep else None),                # one GOF alone (blocking call): the headline is a throughput with GOFs in flight, not this
                "steady_state_fps_256": (steady["value"] if steady else (round(fps, 3) if steps >= 256 else None)), "steady_state": steady,
                "config": {"workload": f"{n_pc}-frame GOF, {w}x{h} V-PCC maps (2x{n_pc} geometry + 2x{n_pc} attribute yuv420p10 I/P pairs, {n_pc} occupancy {w // 2}x{h // 2} lossless), "
                                       f"R5 (QP16/22, prec 2) -> R3 (QP24/32, prec 4), synthetic longdress-like atlas", "input": input_kind,
                           "encoder": ("RBT-E1, wavefront mode (one slice per picture, a dependent slice segment per CTB row, entropy_coding_sync)" if args.rows < 0 else f"RBT-E1, {args.rows or 'all'} CTB row(s) per slice")
                                      + ", 35 intra modes, SATD block costs, closed-loop mode choice with a coded trial of the two cheapest modes, rounding by level and position, one-or-four transform units per intra CU (4x4 luma blocks with the DST or transform skip), SAO, closed (I,P) pairs, CQP, preset default",
                           "gof_per_gpu": 1, "arena_MB_per_gof": round(arena_probe[0] / 1e6, 1) if arena_probe else None, "jobs_in_flight": D, "gofs_per_job": round(gpj, 3), "gofs_in_flight": round(D * gpj), "setup_jobs_before_warmup": primed, "in_bytes": in_bytes, "out_bytes": out_bytes, "parallelism": f"gof-shard x{world}"},
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                             "traffic": traffic, "kernel_ms": {k_: round(v, 3) for k_, v in groups.items()},
                             "path_achieved_GBs": round(path_achieved, 3), "path_frac": round(path_achieved / HBM_PEAK_GBS, 6), "instruction_issue": issue},
                "external_decoder": ext["external_decoder"], "libx265": ext["libx265"],      # third-party HEVC tools on this box (tests/external_tools.py): absent on every box so far - parity with one is unpinned
                "cabac": cabac, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "sequence_walk": walk, "v3c_file": v3c_file, "rate_fanout": fanout, "multi_gof": multi, "in_flight_sweep": sweep, "quality": quality,
                "host_ms": {"parse": round(st["host_parse_ms"], 3), "pack": round(st["host_pack_ms"], 3), "submit_call": round(host_submit_ms, 3), "wait_call": round(host_wait_ms, 3), "job_gpu_span": round(st["gpu_ms"], 3), "job_span": round(st["total_ms"], 3)}}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush(); sys.stdout.flush()
        os._exit(1)          # a failed rank leaves at once: interpreter shutdown would run the process group's destructor, which waits for peers that are waiting for this rank
