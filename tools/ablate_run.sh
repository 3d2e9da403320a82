#!/bin/bash
# GPU box: what each phase of the reconstruction chain and of the intra analysis costs in instructions - a counter pass over one GOF with every measurement build
# tools/ablate.sh made (RBT_ABLATE masks; decode only for the reconstruction masks 1..32, a transcode for the analysis masks 0x100..). Output: gpurun_out/ablate.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/ablate.txt
for spec in "$@"; do
  mask=${spec%%:*}; mode=${spec##*:}
  rm -rf $O/prof_ab
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/prof_ab -o ab -- python3 $R/tools/ablate_run.py $mask $mode > $O/prof_ab.log 2>&1 || { tail -5 $O/prof_ab.log; exit 5; }
  python3 - $O/prof_ab $mask $mode >> $O/ablate.txt <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
f = glob.glob(sys.argv[1] + "/**/ab_counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rbtk::", "").split("<")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in ("k_recon_level", "k_enc_analyse", "k_enc_intra_wave", "k_parse_tasks", "k_parse"):
    if k in acc: print("mask %-7s %-9s %-17s valu %7.4f G salu %7.4f G lds %7.4f G" % (sys.argv[2], sys.argv[3], k, acc[k]["SQ_INSTS_VALU"] / 1e9, acc[k]["SQ_INSTS_SALU"] / 1e9, acc[k]["SQ_INSTS_LDS"] / 1e9))
PY
  tail -4 $O/ablate.txt
done
