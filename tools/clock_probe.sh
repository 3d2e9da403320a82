#!/bin/bash
# Development probe (GPU box): engine clock and power while the benchmark's long run is going (rocm-smi sampled beside it)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/clock_samples.txt 2>&1 &
S=$!
sleep 3
timeout -k 10 200 python3 $R/bench.py --steps 256 --warmup 32 --gofs-per-job 2 $QUIET > $O/clk.json 2> $O/clk.err
sleep 2
timeout -k 10 100 python3 $R/bench.py --steps 6 --warmup 1 --in-flight 1 --gofs-per-job 1 $QUIET > $O/clk1.json 2> $O/clk1.err
wait $S
cat $O/clock_samples.txt | cut -c1-220
