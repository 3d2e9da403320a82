#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/profile_round.sh'): the benchmark line and every rocprofv3 pass tools/refresh_profiles.py turns into profiles/<tag>_*.
# Counter passes are separate runs with --kernel-trace only (no other trace domain), the program itself after `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
if [ -z "$RBT_PROFILE_SKIP_BENCH" ]; then      # (set it to take the profiler passes only, next to bench lines that exist)
timeout -k 10 900 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err || exit 1
# the command the driver times at round end (a run that is all ramp-up and drain: DESIGN.md 5)
timeout -k 10 900 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver.json 2> $O/bench_line_driver.err || exit 1
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python3 $R/bench.py --steps 32 --warmup 16 --gofs-per-job 2 $QUIET > $O/prof_kt.log 2>&1 || exit 2
ONE="--steps 1 --warmup 0 --in-flight 1 --gofs-per-job 1 $QUIET"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/prof_f -o f -- python3 $R/bench.py $ONE > $O/prof_f.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/prof_w -o w -- python3 $R/bench.py $ONE > $O/prof_w.log 2>&1 || exit 4
# the same two passes with the fused loop filters (RBT_FUSED_LF=1 RBT_FUSED_ENC_LF=1: DESIGN.md 2), for profiles/<tag>_pmc_traffic_fused_lf.json
export RBT_FUSED_LF=1 RBT_FUSED_ENC_LF=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/prof_ff -o f -- python3 $R/bench.py $ONE > $O/prof_ff.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/prof_wf -o w -- python3 $R/bench.py $ONE > $O/prof_wf.log 2>&1 || exit 4
unset RBT_FUSED_LF RBT_FUSED_ENC_LF
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES -d $O/prof_sq -o sq -- python3 $R/bench.py $ONE > $O/prof_sq.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o tl -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight 1 --gofs-per-job 1 $QUIET > $O/prof_tl.log 2>&1 || exit 6
ls $O/prof_f $O/prof_w $O/prof_sq $O/prof_tl
