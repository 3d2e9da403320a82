#!/bin/bash
# One counter pass: VALU / SALU / LDS instructions per kernel of one blocking transcode step (what profile_round.sh's SQ pass takes), summed per kernel. Output: gpurun_out/sq_pass.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_sq1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/prof_sq1 -o sq -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --gofs-per-job 1 $QUIET > $O/prof_sq1.log 2>&1 || exit 5
python3 - $O/prof_sq1 > $O/sq_pass.txt <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
f = glob.glob(sys.argv[1] + "/**/sq_counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rbtk::", "").split("<")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] for k, v in acc.items() if not k.startswith("__"))
for k, v in sorted(acc.items(), key=lambda kv: -(kv[1]["SQ_INSTS_VALU"] + kv[1]["SQ_INSTS_SALU"])):
    if k.startswith("__"): continue
    print("%-20s valu %7.3f G salu %7.3f G lds %6.3f G waves %9d  share %5.1f %%" % (k, v["SQ_INSTS_VALU"] / 1e9, v["SQ_INSTS_SALU"] / 1e9, v["SQ_INSTS_LDS"] / 1e9, int(v["SQ_WAVES"]), 100 * (v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"]) / tot))
print("total %.3f G" % (tot / 1e9))
PY
cat $O/sq_pass.txt
