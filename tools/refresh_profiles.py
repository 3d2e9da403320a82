"""Turns the rocprofv3 outputs a gpurun call merged into gpurun_out/ into the committed summaries under profiles/ (every file there
comes from this script: profiles/<tag>_bench_line.json, _bench_kernel_stats.csv, _pmc_traffic.json, _pmc_sq.json, _timeline.txt).
Inputs (tools/profile_round.sh runs them on the GPU box): gpurun_out/bench_line.json (python bench.py), prof_kt/
(rocprofv3 --kernel-trace --stats --output-format csv of the bench command without its informative extras: --steps 32 --warmup 16 --gofs-per-job 2
--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0), prof_f/ and prof_w/ (--pmc FETCH_SIZE / WRITE_SIZE,
separate passes, one blocking step: --steps 1 --warmup 0 --in-flight 1), prof_sq/ (--pmc SQ instruction / matrix-core counters, same
step), prof_tl/ (--kernel-trace of one blocking step: --steps 1 --warmup 1 --in-flight 1)."""
import collections, csv, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
shutil.copy("gpurun_out/prof_kt/kt_kernel_stats.csv", f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy("gpurun_out/bench_line.json", f"profiles/{tag}_bench_line.json")
import os
if os.path.exists("gpurun_out/bench_line_driver.json"): shutil.copy("gpurun_out/bench_line_driver.json", f"profiles/{tag}_bench_line_driver.json")   # python bench.py --gpus 1 --steps 20 --warmup 5

def kname(n):   # "void rbtk::k_parse<384>(RbtFrame*, ...)" -> "k_parse"
    return n.split("(")[0].replace("void ", "").replace("rbtk::", "").split("<")[0]

def agg(path, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = kname(r["Kernel_Name"]); d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, --kernel-trace only), python3 bench.py --steps 1 --warmup 0 --in-flight 1 --cpu-sample 0 "
        "--multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 on MI355X: one transcode step of the HM-like 32-frame 1280x1280 GOF, nothing else in the "
        "process. Values are KB summed over every dispatch of the kernel in that step. FETCH_SIZE is reported as counted: on gfx950 it tallies 128-B requests at 64 B, "
        "i.e. exactly half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) - FETCH_SIZE_KB_x2 is that correction, an upper bound for the "
        "kernels here, whose accesses are mostly narrower than 16 B per lane (uncalibrated widths). kernel_ms / HBM_GBps: time of the dispatches in the FETCH_SIZE pass "
        "and (FETCH + WRITE) / time; MI355X peak 8000 GB/s.")

def traffic(fdir, wdir, dst, extra=""):
    F = agg(f"gpurun_out/{fdir}/f_counter_collection.csv", "FETCH_SIZE"); W = agg(f"gpurun_out/{wdir}/w_counter_collection.csv", "WRITE_SIZE")
    out = {"note": NOTE + extra, "kernels": {}}
    dur = collections.defaultdict(float)     # kernel time of the same (FETCH_SIZE) pass; counter collection serialises the dispatches
    for r in csv.DictReader(open(f"gpurun_out/{fdir}/f_kernel_trace.csv")): dur[kname(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tf = tw = 0.0
    for k in sorted(set(F) | set(W)):
        if k.startswith("__"): continue
        out["kernels"][k] = {"dispatches": F[k][0] or W[k][0], "FETCH_SIZE_KB": round(F[k][1], 1), "FETCH_SIZE_KB_x2": round(2 * F[k][1], 1), "WRITE_SIZE_KB": round(W[k][1], 1), "kernel_ms": round(dur[k], 3),
                             "HBM_GBps": round((F[k][1] + W[k][1]) * 1024 / (dur[k] * 1e-3) / 1e9, 1) if dur[k] > 0 else None}
        tf += F[k][1]; tw += W[k][1]
    out["total"] = {"FETCH_SIZE_GB": round(tf * 1024 / 1e9, 3), "WRITE_SIZE_GB": round(tw * 1024 / 1e9, 3), "FETCH_plus_WRITE_GB": round((tf + tw) * 1024 / 1e9, 3),
                    "FETCH_x2_plus_WRITE_GB": round((2 * tf + tw) * 1024 / 1e9, 3),
                    "filter_dispatches": sum(v["dispatches"] for k, v in out["kernels"].items() if k in ("k_deblock", "k_sao", "k_loopfilter", "k_enc_sao"))}
    json.dump(out, open(dst, "w"), indent=1)
traffic("prof_f", "prof_w", f"profiles/{tag}_pmc_traffic.json")
if os.path.exists("gpurun_out/prof_ff/f_counter_collection.csv"):
    traffic("prof_ff", "prof_wf", f"profiles/{tag}_pmc_traffic_fused_lf.json", " THIS FILE: the same step with RBT_FUSED_LF=1 RBT_FUSED_ENC_LF=1 (deblocking + SAO of the decoder in one launch "
            "through LDS tiles, the encoder's deblocking inside its SAO kernel) - not the default, see DESIGN.md 2.")

# SQ pass: instruction mix per kernel and what the matrix cores did (SURVEY / north star: MFMA only for the dense 32x32 transform tiles)
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_I8", "SQ_INSTS_VALU_MFMA_MOPS_I8", "SQ_VALU_MFMA_BUSY_CYCLES"]
sq = {"note": "rocprofv3 --pmc " + " ".join(names) + " (one pass, --kernel-trace only) of the same blocking step. Sums over every dispatch of the kernel. "
              "MFMA_I8 = v_mfma_i32_32x32x32_i8 instructions (csrc/rbt_mfma.h: three per 32-point transform stage), MOPS = their operations in units of 512; "
              "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of the kernel. The matrix cores only ever see the 32x32 transform blocks, a few per cent of the "
              "path's instructions: their busy fraction is evidence that the path is NOT a GEMM, not a utilisation target.", "kernels": {}}
try:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open("gpurun_out/prof_sq/sq_counter_collection.csv")): acc[kname(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(acc):
        if k.startswith("__"): continue
        v = acc[k]; e = {n: int(v.get(n, 0)) for n in names}
        e["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / v["SQ_BUSY_CYCLES"], 6) if v.get("SQ_BUSY_CYCLES") else None
        sq["kernels"][k] = e
    line = json.load(open("gpurun_out/bench_line.json"))
    if "k_parse" in sq["kernels"]:
        bits = 8 * line["config"]["in_bytes"]; e = sq["kernels"]["k_parse"]
        sq["parser_per_bit"] = {"slice_data_bits_per_gof": bits, "instructions_per_bit": round((e["SQ_INSTS_VALU"] + e["SQ_INSTS_SALU"]) / bits, 2),
                                "note": "VALU + SALU instructions of all 160 parser waves of the GOF per bit of input (about 1.17 bins per bit)"}
    json.dump(sq, open(f"profiles/{tag}_pmc_sq.json", "w"), indent=1)
    # the bench lines of this round carry roofline.instruction_issue computed from whatever counter file was committed when they ran: put this round's counts in
    salu = sum(v["SQ_INSTS_SALU"] for v in sq["kernels"].values()); valu = sum(v["SQ_INSTS_VALU"] for v in sq["kernels"].values())
    for name in (f"profiles/{tag}_bench_line.json", f"profiles/{tag}_bench_line_driver.json"):
        if not os.path.exists(name): continue
        d = json.loads(open(name).read().strip().splitlines()[-1])
        ii = d.get("roofline", {}).get("instruction_issue")
        if ii:
            gofs_per_s = d["steps"] / (d["ms_per_step"] * d["steps"] / 1000.0); peak = 256 * 2.4e9
            ii.update({"salu_G_per_gof": round(salu / 1e9, 2), "valu_G_per_gof": round(valu / 1e9, 2), "scalar_issue_frac": round(salu * gofs_per_s / peak, 3), "vector_issue_frac": round(valu * gofs_per_s / peak, 3),
                       "counters": f"{tag}_pmc_sq.json (same round; recomputed by tools/refresh_profiles.py)"})
            open(name, "w").write(json.dumps(d) + "\n")
except FileNotFoundError:
    print("no prof_sq pass: profiles/%s_pmc_sq.json not written" % tag)

rows = [r for r in csv.DictReader(open("gpurun_out/prof_tl/tl_kernel_trace.csv")) if "rbtk::" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
parses = [r for r in rows if "k_parse" in r["Kernel_Name"]]
t0 = min(r["s"] for r in parses[-3:]); last = [r for r in rows if r["s"] >= t0]; tend = max(r["e"] for r in last)
with open(f"profiles/{tag}_timeline.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --in-flight 1 --cpu-sample 0 ... (MI355X, HM-like input): one GOF alone (blocking call); timed step only, ms from the first kernel\n")
    o.write("# one HIP stream (queue) per sub-bitstream pipeline (occupancy, geometry, attribute) + the auxiliary stream of the longest one. step span %.2f ms\n" % ((tend - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in last: byq[r["Queue_Id"]].append(r)
    for q, rs in byq.items():
        o.write("queue %s dispatches %d\n" % (q, len(rs)))
        cur = None
        for r in rs + [None]:
            name = kname(r["Kernel_Name"]) if r else None
            if cur and cur[0] == name: cur[2] = r["e"]; cur[3] += 1; cur[4] += r["e"] - r["s"]
            else:
                if cur: o.write("   %-18s start %8.2f end %8.2f launches %4d busy %8.2f\n" % (cur[0], (cur[1] - t0) / 1e6, (cur[2] - t0) / 1e6, cur[3], cur[4] / 1e6))
                cur = [name, r["s"], r["e"], 1, r["e"] - r["s"]] if r else None
print(open(f"profiles/{tag}_timeline.txt").read())
