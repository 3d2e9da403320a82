#!/bin/bash
# Development probe (GPU box): the benchmark under values of one environment variable, interleaved repeats. $1 = variable, $2.. = values
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
VAR=$1; shift
for shape in "--steps 20 --warmup 5" "--steps 256 --warmup 48" "--steps 4 --warmup 1 --in-flight 1 --gofs-per-job 1"; do for rep in 1 2 3; do for val in "$@"; do
  env $VAR=$val timeout -k 10 200 python3 $R/bench.py $shape $QUIET > $O/ab.json 2> $O/ab.err || exit 2
  python3 - "$VAR=$val" "$shape" $O/ab.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "fps", d["value"], {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()}, flush=True)
PY
done; done; done
