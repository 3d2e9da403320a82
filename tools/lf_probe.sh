#!/bin/bash
# Development probe (GPU box): kernel stats of a short benchmark run with the fused loop filter and with the decoder's unfused path (RBT_FUSED_LF=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export RBT_FUSED_LF=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lf$v -o lf -- python3 $R/bench.py --steps 32 --warmup 16 --gofs-per-job 2 $QUIET > $O/prof_lf$v.log 2>&1 || exit 2
  tail -1 $O/prof_lf$v.log | cut -c1-200
  head -14 $O/prof_lf$v/lf_kernel_stats.csv | cut -c1-160
done
